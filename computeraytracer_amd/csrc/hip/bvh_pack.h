// bvh_pack.h -- per-primitive device helpers shared by the builders (bvh_build.hip, bvh_direct.hip): padded boxes
// and the traversal records of a sorted primitive.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "build_internal.h"

namespace crt {

// Conservative padding: the slab test runs in fma form with rounded -o/d terms (intersect.h); 2^-18
// of the coordinate magnitude is ~64 ulp, far above its error, far below any feature size.
__device__ __forceinline__ void pad_box(float lo[3], float hi[3]) {
    for (int a = 0; a < 3; a++) {
        float m = fmaxf(fabsf(lo[a]), fabsf(hi[a]));
        float p = m * 3.8146973e-6f + (hi[a] - lo[a]) * 9.5367432e-7f + 1e-30f;
        lo[a] -= p; hi[a] += p;
    }
}

__device__ __forceinline__ void prim_box(const BuildInput &in, uint32_t g, float lo[3], float hi[3]) {
    if (g < in.n_spheres) {
        const crt_sphere &s = in.spheres[g];
        float r = fabsf(s.radius);
        for (int a = 0; a < 3; a++) { lo[a] = s.center[a] - r; hi[a] = s.center[a] + r; }
    } else if (g < in.n_spheres + in.n_quads) {
        const crt_quad &q = in.quads[g - in.n_spheres];
        for (int a = 0; a < 3; a++) {
            float c = q.corner[a], cu = c + q.u[a], cv = c + q.v[a], cuv = (c + q.u[a]) + q.v[a];
            lo[a] = fminf(fminf(c, cu), fminf(cv, cuv));
            hi[a] = fmaxf(fmaxf(c, cu), fmaxf(cv, cuv));
        }
    } else {
        // three 16-byte loads (v0, v1, v2 rows of the 96-byte record) instead of nine strided dwords
        const float4 *p = (const float4 *)(in.tris + (g - in.n_spheres - in.n_quads));
        const float4 v0 = p[0], v1 = p[1], v2 = p[2];
        lo[0] = fminf(fminf(v0.x, v1.x), v2.x); hi[0] = fmaxf(fmaxf(v0.x, v1.x), v2.x);
        lo[1] = fminf(fminf(v0.y, v1.y), v2.y); hi[1] = fmaxf(fmaxf(v0.y, v1.y), v2.y);
        lo[2] = fminf(fminf(v0.z, v1.z), v2.z); hi[2] = fmaxf(fmaxf(v0.z, v1.z), v2.z);
    }
    pad_box(lo, hi);
}

// Head of a triangle's 64-byte shading record (and, copied from there, of its traversal record): v0, e10 = v1 - v0,
// e20 = v2 - v0, the two IEEE subtractions of hit_tri (compute.comp:281-282).  A triangle of ZERO AREA (cross(e10, e20)
// == 0 exactly: a repeated vertex, as real meshes have them) gets e10 = e20 = 0 and is then never hit (hit_tri's
// determinant is 0).  With the uploaded edges its determinant e10 . (d x e20) is rounding noise of either sign, often above
// the shader's 1e-8 threshold, and the beta / gamma tests then pass for rays that run close to the triangle's LINE, at
// distances that need not lie inside the triangle's box: whether such a ray "hits" depends on whether the traversal
// happens to reach the leaf, i.e. on the tree (found by the random-scene tests: one ray in 4 x 10^8).  The oracle's
// canonical modes apply the same rule (oracle.cpp: hit_tri); its reference mode keeps the shader's behaviour.
__device__ __forceinline__ void tri_record_head(const float4 &v0, const float4 &v1, const float4 &v2, float4 &o0, float4 &o1, float &e20z) {
    float e1[3] = {v1.x - v0.x, v1.y - v0.y, v1.z - v0.z}, e2[3] = {v2.x - v0.x, v2.y - v0.y, v2.z - v0.z};
    const float cx = e1[1] * e2[2] - e1[2] * e2[1], cy = e1[2] * e2[0] - e1[0] * e2[2], cz = e1[0] * e2[1] - e1[1] * e2[0];
    if (cx == 0.0f && cy == 0.0f && cz == 0.0f) { e1[0] = e1[1] = e1[2] = 0.0f; e2[0] = e2[1] = e2[2] = 0.0f; }
    o0 = make_float4(v0.x, v0.y, v0.z, e1[0]);
    o1 = make_float4(e1[1], e1[2], e2[0], e2[1]);
    e20z = e2[2];
}

// Traversal record of sorted position s (primitive g) into its kind's array; returns the kind and the slot there.
// WANT_BOX: also the primitive's padded box.  Spheres and quads: prim_box's own arithmetic on the record read here anyway.
// Triangles: from the record's v0, v0 + e10, v0 + e20, i.e. corners at most an ulp from the uploaded ones, which is far
// inside the 64-ulp pad and is the triangle the intersection test (which works from v0, e10, e20) sees.
struct PackOut { float4 *tri_pack, *sphere_pack, *quad_pack; uint32_t *sphere_ids, *quad_ids, *tri_ids; };
template <bool WANT_BOX>
__device__ __forceinline__ void pack_prim(const BuildInput &in, uint32_t n, uint32_t s, uint32_t g, const uint32_t *rank_sphere, const uint32_t *rank_quad,
                                          const float4 *tri_shade, const PackOut &out, uint32_t &kind, uint32_t &slot, float lo[3], float hi[3]) {
    // rank arrays are null when the scene holds one kind of primitive: every sorted position is that kind's slot
    uint32_t rs = rank_sphere ? rank_sphere[s] : (in.n_spheres == n ? s : 0u), rq = rank_quad ? rank_quad[s] : (in.n_quads == n ? s : 0u);
    if (g < in.n_spheres) {
        const crt_sphere &sp = in.spheres[g];
        out.sphere_pack[rs] = make_float4(sp.center[0], sp.center[1], sp.center[2], sp.radius);
        out.sphere_ids[rs] = g;
        kind = CRT_ID_KIND_SPHERE; slot = rs;
    } else if (g < in.n_spheres + in.n_quads) {
        uint32_t i = g - in.n_spheres;
        const crt_quad &q = in.quads[i];
        float4 *o = out.quad_pack + 4 * (size_t)rq;
        o[0] = make_float4(q.normal[0], q.normal[1], q.normal[2], q.d);
        o[1] = make_float4(q.corner[0], q.corner[1], q.corner[2], q.u[0]);
        o[2] = make_float4(q.u[1], q.u[2], q.v[0], q.v[1]);
        o[3] = make_float4(q.v[2], q.w[0], q.w[1], q.w[2]);
        out.quad_ids[rq] = i;
        kind = CRT_ID_KIND_QUAD; slot = rq;
    } else {
        uint32_t i = g - in.n_spheres - in.n_quads;
        uint32_t rt = s - rs - rq;
        // v0, e10 = v1 - v0, e20 = v2 - v0 (the two IEEE subtractions of hit_tri, compute.comp:281-282) are the head of
        // the triangle's 64-byte shading record, written by k_prim_bounds: half a cache line instead of a 96-byte upload record
        const float4 *tp = tri_shade + 4 * (size_t)i;
        const float4 a = tp[0], b = tp[1], c = tp[2];
        float4 *o = out.tri_pack + 3 * (size_t)rt;
        o[0] = a;
        o[1] = b;
        o[2] = make_float4(c.x, __uint_as_float(CRT_MAKE_ID(CRT_ID_KIND_TRI, i)), 0.0f, 0.0f);
        out.tri_ids[rt] = i;
        kind = CRT_ID_KIND_TRI; slot = rt;
        if (WANT_BOX) {
            const float v0[3] = {a.x, a.y, a.z}, v1[3] = {a.x + a.w, a.y + b.x, a.z + b.y}, v2[3] = {a.x + b.z, a.y + b.w, a.z + c.x};
            for (int k = 0; k < 3; k++) { lo[k] = fminf(fminf(v0[k], v1[k]), v2[k]); hi[k] = fmaxf(fmaxf(v0[k], v1[k]), v2[k]); }
            pad_box(lo, hi);
        }
        return;
    }
    if (WANT_BOX) prim_box(in, g, lo, hi);
}

}  // namespace crt
