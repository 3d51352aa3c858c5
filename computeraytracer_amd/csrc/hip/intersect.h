// intersect.h -- closest-hit search over the build's pair BVH (R5-R9 of
// SURVEY.md section 8(a)) and reconstruction of the shader's HitRecord.
//
// Traversal keeps only (t, id).  Acceptance is order-independent: a candidate
// replaces the current hit when tNear < t and (t < best or (t == best and
// id < best_id)), so any conservative traversal order returns the same
// primitive as testing every primitive in ID order (the shader's strict '<',
// compute.comp:239,260,301, makes the first primitive tested win a tie).
#pragma once
#include "device_math.h"
#include "crt/types.h"

namespace crt {

struct DeviceScene {
    const float4 *nodes;        // pair p = nodes[4p .. 4p+3] = {lo0|link0, hi0|count0, lo1|link1, hi1|count1}
    uint32_t n_pairs;           // 0 = empty scene
    const float4 *wide;         // 8-wide quantised nodes (bvh_wide.hip), null unless built with CRT_BVH_WIDE
    const uint32_t *wide_links; // their leaf links, eight words per node
    uint32_t n_wide;
    const float4 *tri_pack;     // sorted slot s: {v0.xyz, e10.x} {e10.y, e10.z, e20.x, e20.y} {e20.z, id bits, -, -}
    const float4 *sphere_pack;  // sorted slot s: {center.xyz, radius}
    const float4 *quad_pack;    // sorted slot s: {normal.xyz, d} {corner.xyz, u.x} {u.y, u.z, v.x, v.y} {v.z, w.xyz}
    const float4 *tri_shade;    // input index i: {v0.xyz, e10.x} {e10.yz, e20.xy} {e20.z, materialIndex, u0, u1} {u2, v0, v1, v2} (64 B, one line)
    const uint32_t *sphere_ids; // sorted slot -> input index
    const uint32_t *quad_ids;
    const crt_sphere *spheres;  // as uploaded (input order): shading data
    const crt_quad *quads;
    const crt_tri *tris;
    uint32_t n_tris;            // 0: tri_shade holds no record
    const crt_material *materials;
    const uint4 *tex0;          // textures[0] as bilinear footprints: entry y * w + x = the RGBA8 texels (x, y), (x+1, y), (x, y+1),
                                // (x+1, y+1) with REPEAT wrap, so a sample is ONE 16-byte load (the shader only ever samples index 0,
                                // compute.comp:387); the scene's other textures follow it in the same allocation (tex_table)
    uint32_t tex0_w, tex0_h;
    uint32_t n_textures;
    const uint4 *tex_table;     // texture i: {first texel (offset from tex0 in texels), width, height, -}
    uint32_t ext;               // CRT_EXT_* build extensions (crt_extensions_set); 0 = the reference's behaviour
    const float4 *dbg_boxes;    // crt_debug_boxes_upload: the reference's BVHNode array (64-byte stride), or null
    uint32_t n_dbg_boxes;
};

#ifndef CRT_LDS_STACK
#define CRT_LDS_STACK 24
#endif
constexpr int kLdsStack = CRT_LDS_STACK;     // per-lane LDS stack entries; deeper levels spill to scratch
constexpr int kSpillStack = 96;               // LBVH pair trees are at most 95 deep (63 Morton bits + 32 index bits); SAH-refined ones are measured at
                                              // build time and refused beyond kMaxPairDepth = 96 (bvh_sah.hip); lds_depth may be as low as 1
constexpr uint32_t kIdMiss = 0xFFFFFFFFu;

CRT_DEV bool accept_hit(float t, uint32_t id, float best_t, uint32_t best_id) {
    if (!(kTNear < t)) return false;
    if (t < best_t) return true;
    return t == best_t && id < best_id;
}

// compute.comp:229-239 (distance only)
CRT_DEV bool sphere_t(V3 o, V3 d, float4 cr, float &t) {
    V3 c = mk3(cr.x, cr.y, cr.z);
    V3 rel = o - c;
    float b = dot3(rel, d);
    V3 qc = rel - b * d;
    float disc = cr.w * cr.w - dot3(qc, qc);
    if (disc < 0.0f) return false;
    t = -b - __builtin_sqrtf(disc);
    return true;
}

// compute.comp:253-268 (distance only)
CRT_DEV bool quad_t(V3 o, V3 d, const float4 *q, float &t, float &alpha, float &beta) {
    float4 a = q[0], b = q[1], c = q[2], e = q[3];
    V3 n = mk3(a.x, a.y, a.z);
    float denom = dot3(n, d);
    if (__builtin_fabsf(denom) < 1e-8f) return false;
    t = (a.w - dot3(n, o)) / denom;
    V3 isect = o + t * d;
    V3 planar = isect - mk3(b.x, b.y, b.z);
    V3 u = mk3(b.w, c.x, c.y), v = mk3(c.z, c.w, e.x), w = mk3(e.y, e.z, e.w);
    alpha = dot3(w, cross3(planar, v));
    beta = dot3(w, cross3(u, planar));
    if (!(alpha >= 0.0f && alpha <= 1.0f) || !(beta >= 0.0f && beta <= 1.0f)) return false;
    return true;
}

// compute.comp:280-301 (distance only)
CRT_DEV bool tri_t(V3 o, V3 d, V3 v0, V3 e10, V3 e20, float &t, float &beta, float &gamma, float &invDet) {
    V3 p = cross3(d, e20);
    float det = dot3(e10, p);
    if (__builtin_fabsf(det) < 1e-8f) return false;
    V3 r0 = o - v0;
    V3 q = cross3(r0, e10);
    beta = dot3(r0, p);
    gamma = dot3(d, q);
    if (beta < 0.0f || gamma < 0.0f || (beta + gamma) > det) return false;
    invDet = 1.0f / det;
    t = dot3(e20, q) * invDet;
    return true;
}

struct Counts { uint32_t nodes, prims; };

// ---- build extensions (crt_extensions_set; no counterpart in the reference, off by default) ----
CRT_DEV int wrapi(float f, int n) { int i = (int)f % n; return i < 0 ? i + n : i; }

// The texture a material samples: textures[0] as the shader does (compute.comp:387), or textures[textureIndex] with
// CRT_EXT_TEXTURE_INDEX (crt_scene_upload / crt_extensions_set refuse an index past the array).
// EXT = false: the kernels that run while no extension is set; they do not even test the flag (as a run-time branch with
// per-lane texture sizes behind it, the test cost k_wf_shade 7 %).
template <bool EXT>
CRT_DEV const uint4 *material_texture(const DeviceScene &sc, uint32_t texIndex, uint32_t &w, uint32_t &h) {
    w = sc.tex0_w; h = sc.tex0_h;
    if (!EXT || !(sc.ext & CRT_EXT_TEXTURE_INDEX)) return sc.tex0;
    const uint4 e = sc.tex_table[texIndex];
    w = e.y; h = e.z;
    return sc.tex0 + (size_t)e.x;
}

// Alpha channel through the colour fetch's sampler (shade.h sample_texture: bilinear, REPEAT, centres at +0.5).
CRT_DEV float texture_alpha(const uint4 *tex, uint32_t w, uint32_t h, float u, float v) {
    float x = u * (float)w - 0.5f, y = v * (float)h - 0.5f;
    float x0f = __builtin_floorf(x), y0f = __builtin_floorf(y);
    float fx = x - x0f, fy = y - y0f;
    int x0 = wrapi(x0f, (int)w), y0 = wrapi(y0f, (int)h);
    const uint4 q = tex[(size_t)y0 * w + (size_t)x0];      // the 2 x 2 footprint (x0, y0), (x1, y0), (x0, y1), (x1, y1)
    const float a00 = (float)(q.x >> 24) / 255.0f, a10 = (float)(q.y >> 24) / 255.0f, a01 = (float)(q.z >> 24) / 255.0f, a11 = (float)(q.w >> 24) / 255.0f;
    float top = a00 * (1.0f - fx) + a10 * fx;
    float bot = a01 * (1.0f - fx) + a11 * fx;
    return top * (1.0f - fy) + bot * fy;
}

// CRT_EXT_ALPHA_TEST: a triangle hit counts only where its material's texture is opaque (alpha >= 0.5 at the hit's
// (u, v), compute.comp:309-314 for the coordinates).  beta, gamma, invDet are tri_t's for this ray and triangle.
CRT_DEV bool tri_alpha_pass(const DeviceScene &sc, uint32_t id, float beta, float gamma, float invDet) {
    const float4 *p = sc.tri_shade + 4 * (size_t)(id & 0x3FFFFFFFu);
    const float4 c = p[2], e = p[3];
    const uint32_t texIndex = sc.materials[f2u(c.y)].textureIndex;
    if (texIndex == CRT_BAD_INDEX || sc.n_textures == 0) return true;
    beta = beta * invDet; gamma = gamma * invDet;
    const float alpha = 1.0f - beta - gamma;
    const float u = (alpha * c.z + beta * c.w) + gamma * e.x;
    const float v = (alpha * e.y + beta * e.z) + gamma * e.w;
    uint32_t w, h;
    const uint4 *tex = material_texture<true>(sc, texIndex, w, h);
    return !(texture_alpha(tex, w, h, u, v) < 0.5f);
}

// Slab test for culling only (R5).  fma form: t = b * (1/d) - o * (1/d).  Boxes are padded at build
// time and the far bound carries a relative pad, so rounding here never culls a primitive that an
// exhaustive test would accept.
CRT_DEV bool slab(float4 lo, float4 hi, V3 idir, V3 noid, float tfar, float &tn) {
    float ax = __builtin_fmaf(lo.x, idir.x, noid.x), bx = __builtin_fmaf(hi.x, idir.x, noid.x);
    float ay = __builtin_fmaf(lo.y, idir.y, noid.y), by = __builtin_fmaf(hi.y, idir.y, noid.y);
    float az = __builtin_fmaf(lo.z, idir.z, noid.z), bz = __builtin_fmaf(hi.z, idir.z, noid.z);
    float n0 = __builtin_fminf(ax, bx), f0 = __builtin_fmaxf(ax, bx);
    float n1 = __builtin_fminf(ay, by), f1 = __builtin_fmaxf(ay, by);
    float n2 = __builtin_fminf(az, bz), f2 = __builtin_fmaxf(az, bz);
    tn = __builtin_fmaxf(__builtin_fmaxf(n0, n1), __builtin_fmaxf(n2, 0.0f));
    float tf = __builtin_fminf(__builtin_fminf(f0, f1), __builtin_fminf(f2, tfar));
    return tn <= tf;
}

// EXT: compiled only into the kernels that run while some extension is set (as a run-time branch in the one kernel the
// alpha test cost the headline k_wf_extend 18 spilled VGPRs).
template <bool COUNT, bool EXT>
CRT_DEV void test_leaf(const DeviceScene &sc, V3 o, V3 d, uint32_t first, uint32_t count_kind, float &best_t, uint32_t &best_id, Counts &cnt) {
    uint32_t kind = count_kind >> 28, n = count_kind & 0x0FFFFFFFu;
    if (COUNT) cnt.prims += n;
    if (kind == CRT_ID_KIND_TRI) {
        // Two triangles per trip, both records requested before either is tested: a leaf step waits for one memory round
        // trip per PAIR instead of one per triangle (the traversal kernels are bound by such dependent waits, not by
        // arithmetic).  The asm statements keep the compiler from sinking a load behind a test's branch.
        for (uint32_t i = 0; i < n; i += 2) {
            const float4 *p = sc.tri_pack + 3 * (size_t)(first + i);
            const bool two = i + 1 < n;
            const float4 a = p[0], b = p[1], c = p[2];
            float4 a1 = a, b1 = b, c1 = c;
            if (two) { a1 = p[3]; b1 = p[4]; c1 = p[5]; }
            asm volatile("" :: "v"(a.x), "v"(a.y), "v"(a.z), "v"(a1.x), "v"(a1.y), "v"(a1.z), "v"(c1.y));
            float t, beta, gamma, invDet;
            if (tri_t(o, d, mk3(a.x, a.y, a.z), mk3(a.w, b.x, b.y), mk3(b.z, b.w, c.x), t, beta, gamma, invDet)) {
                uint32_t id = f2u(c.y);
                if (accept_hit(t, id, best_t, best_id) && (!EXT || !(sc.ext & CRT_EXT_ALPHA_TEST) || tri_alpha_pass(sc, id, beta, gamma, invDet))) { best_t = t; best_id = id; }
            }
            if (two && tri_t(o, d, mk3(a1.x, a1.y, a1.z), mk3(a1.w, b1.x, b1.y), mk3(b1.z, b1.w, c1.x), t, beta, gamma, invDet)) {
                uint32_t id = f2u(c1.y);
                if (accept_hit(t, id, best_t, best_id) && (!EXT || !(sc.ext & CRT_EXT_ALPHA_TEST) || tri_alpha_pass(sc, id, beta, gamma, invDet))) { best_t = t; best_id = id; }
            }
        }
    } else if (kind == CRT_ID_KIND_SPHERE) {
        for (uint32_t i = 0; i < n; i++) {
            float t;
            if (sphere_t(o, d, sc.sphere_pack[first + i], t) && kTNear < t && t <= best_t) {
                uint32_t id = CRT_MAKE_ID(CRT_ID_KIND_SPHERE, sc.sphere_ids[first + i]);
                if (accept_hit(t, id, best_t, best_id)) { best_t = t; best_id = id; }
            }
        }
    } else {
        for (uint32_t i = 0; i < n; i++) {
            float t, alpha, beta;
            if (quad_t(o, d, sc.quad_pack + 4 * (size_t)(first + i), t, alpha, beta) && kTNear < t && t <= best_t) {
                uint32_t id = CRT_MAKE_ID(CRT_ID_KIND_QUAD, sc.quad_ids[first + i]);
                if (accept_hit(t, id, best_t, best_id)) { best_t = t; best_id = id; }
            }
        }
    }
}

// R6: closest hit.  `stack` points at this lane's column of the block's LDS stack (stride = blockDim).
template <bool COUNT, bool EXT = false>
CRT_DEV void closest_hit(const DeviceScene &sc, V3 o, V3 d, uint32_t *stack, uint32_t stack_stride,
                         float &best_t, uint32_t &best_id, Counts &cnt) {
    best_t = kInf; best_id = kIdMiss;
    if (sc.n_pairs == 0) return;
    V3 idir;
    idir.x = __builtin_fabsf(d.x) > 1e-30f ? 1.0f / d.x : __builtin_copysignf(1e30f, d.x);
    idir.y = __builtin_fabsf(d.y) > 1e-30f ? 1.0f / d.y : __builtin_copysignf(1e30f, d.y);
    idir.z = __builtin_fabsf(d.z) > 1e-30f ? 1.0f / d.z : __builtin_copysignf(1e30f, d.z);
    V3 noid = mk3(-(o.x * idir.x), -(o.y * idir.y), -(o.z * idir.z));
    uint32_t spill[kSpillStack];
    int sp = 0;
    uint32_t cur = 0;
    float tfar = kInf;
    for (;;) {
        const float4 *p = sc.nodes + 4 * (size_t)cur;
        float4 lo0 = p[0], hi0 = p[1], lo1 = p[2], hi1 = p[3];
        if (COUNT) cnt.nodes += 2;
        float tn0, tn1;
        bool h0 = slab(lo0, hi0, idir, noid, tfar, tn0);
        bool h1 = slab(lo1, hi1, idir, noid, tfar, tn1);
        uint32_t c0 = f2u(hi0.w), c1 = f2u(hi1.w);
        if (h0 && c0 != 0) { test_leaf<COUNT, EXT>(sc, o, d, f2u(lo0.w), c0, best_t, best_id, cnt); h0 = false; }
        if (h1 && c1 != 0) { test_leaf<COUNT, EXT>(sc, o, d, f2u(lo1.w), c1, best_t, best_id, cnt); h1 = false; }
        tfar = best_t * 1.000004f;       // relative pad (2^-18) on the far bound
        uint32_t l0 = f2u(lo0.w), l1 = f2u(lo1.w);
        if (h0 && h1) {
            bool first0 = tn0 <= tn1;
            uint32_t nearc = first0 ? l0 : l1, farc = first0 ? l1 : l0;
            if (sp < kLdsStack) stack[sp * stack_stride] = farc; else if (sp < kLdsStack + kSpillStack) spill[sp - kLdsStack] = farc;
            sp++;
            cur = nearc;
        } else if (h0) cur = l0;
        else if (h1) cur = l1;
        else {
            if (sp == 0) break;
            sp--;
            cur = sp < kLdsStack ? stack[sp * stack_stride] : spill[sp - kLdsStack];
        }
    }
}

// The same traversal as closest_hit, cut into single steps so that a persistent wave can give a lane a new
// ray the moment its old one finishes (k_wf_extend) instead of idling until the slowest lane of the wave is done.
struct Traversal {
    V3 o, d, idir, noid;
    float best_t, tfar;
    uint32_t best_id, cur;
    int sp;
};

CRT_DEV void trav_begin(Traversal &tr, V3 o, V3 d) {
    tr.o = o; tr.d = d;
    tr.idir.x = __builtin_fabsf(d.x) > 1e-30f ? 1.0f / d.x : __builtin_copysignf(1e30f, d.x);
    tr.idir.y = __builtin_fabsf(d.y) > 1e-30f ? 1.0f / d.y : __builtin_copysignf(1e30f, d.y);
    tr.idir.z = __builtin_fabsf(d.z) > 1e-30f ? 1.0f / d.z : __builtin_copysignf(1e30f, d.z);
    tr.noid = mk3(-(o.x * tr.idir.x), -(o.y * tr.idir.y), -(o.z * tr.idir.z));
    tr.best_t = kInf; tr.tfar = kInf; tr.best_id = kIdMiss; tr.cur = 0; tr.sp = 0;
}

// Stack / cursor entries are tagged: an inner node is its pair index; a leaf is
// kLeafTag | kind << 29 | (count - 1) << 26 | first slot, so a leaf can wait in `cur` (or on the stack) until the
// wave decides to run a leaf-test phase with many lanes at once (while-while traversal).
constexpr uint32_t kLeafTag = 0x80000000u;
constexpr uint32_t kMaxLeafLink = 1u << 26;
CRT_DEV uint32_t make_entry(uint32_t link, uint32_t count_kind) {
    if (count_kind == 0) return link;
    return kLeafTag | ((count_kind >> 28) << 29) | (((count_kind & 0x0FFFFFFFu) - 1u) << 26) | link;
}

// Pops the next entry; false when the traversal is finished (best_t / best_id are final).
CRT_DEV bool trav_pop(Traversal &tr, const uint32_t *stack, uint32_t stack_stride, const uint32_t *spill, int lds_depth) {
    if (tr.sp == 0) return false;
    tr.sp--;
    tr.cur = tr.sp < lds_depth ? stack[tr.sp * stack_stride] : spill[tr.sp - lds_depth];
    return true;
}

// Inner phase: one pair visit (64-byte fetch, two slab tests, ordered descent).  tr.cur must be an inner entry.
template <bool COUNT>
CRT_DEV bool trav_inner(const DeviceScene &sc, Traversal &tr, uint32_t *stack, uint32_t stack_stride, uint32_t *spill, int lds_depth, Counts &cnt) {
    const float4 *p = sc.nodes + 4 * (size_t)tr.cur;
    float4 lo0 = p[0], hi0 = p[1], lo1 = p[2], hi1 = p[3];
    if (COUNT) cnt.nodes += 2;
    float tn0, tn1;
    uint32_t c0 = f2u(hi0.w), c1 = f2u(hi1.w);
    bool h0 = slab(lo0, hi0, tr.idir, tr.noid, tr.tfar, tn0) && c0 != (CRT_ID_KIND_TRI << 28);   // skip the empty pad leaf
    bool h1 = slab(lo1, hi1, tr.idir, tr.noid, tr.tfar, tn1) && c1 != (CRT_ID_KIND_TRI << 28);
    uint32_t e0 = make_entry(f2u(lo0.w), c0), e1 = make_entry(f2u(lo1.w), c1);
    if (h0 && h1) {
        bool first0 = tn0 <= tn1;
        uint32_t nearc = first0 ? e0 : e1, farc = first0 ? e1 : e0;
        if (tr.sp < lds_depth) stack[tr.sp * stack_stride] = farc; else if (tr.sp < lds_depth + kSpillStack) spill[tr.sp - lds_depth] = farc;
        tr.sp++;
        tr.cur = nearc;
        return true;
    }
    if (h0) { tr.cur = e0; return true; }
    if (h1) { tr.cur = e1; return true; }
    return trav_pop(tr, stack, stack_stride, spill, lds_depth);
}

// Leaf phase: test the primitives of the leaf entry in tr.cur, then pop.
template <bool COUNT, bool EXT = false>
CRT_DEV bool trav_leaf(const DeviceScene &sc, Traversal &tr, const uint32_t *stack, uint32_t stack_stride, const uint32_t *spill, int lds_depth, Counts &cnt) {
    const uint32_t e = tr.cur;
    const uint32_t kind = (e >> 29) & 3u, n = ((e >> 26) & 7u) + 1u, first = e & (kMaxLeafLink - 1u);
    test_leaf<COUNT, EXT>(sc, tr.o, tr.d, first, (kind << 28) | n, tr.best_t, tr.best_id, cnt);
    tr.tfar = tr.best_t * 1.000004f;       // relative pad (2^-18) on the far bound
    return trav_pop(tr, stack, stack_stride, spill, lds_depth);
}

// The shader's HitRecord (compute.comp:195-203), rebuilt for the winning primitive from the
// uploaded arrays with the same arithmetic hit_sphere/hit_quad/hit_tri use at the moment of the hit.
// (In hit_tri the invDet/beta/gamma are functions of the ray and the triangle only, so evaluating
// them again after traversal gives the same bits.)
struct HitRecord {
    V3 position, normal;
    float u, v;
    bool front;
    uint32_t materialIndex;
};

// compute.comp:207-210
CRT_DEV void set_face_normal(V3 d, V3 outward, HitRecord &rec) {
    rec.front = dot3(d, outward) < 0.0f;
    rec.normal = rec.front ? outward : neg3(outward);
}

CRT_DEV void sphere_uv(V3 outward, float &u, float &v) {   // compute.comp:247-250
    float theta = dm_acos(-outward.y);
    float phi = dm_atan2(-outward.z, outward.x) + kPi;
    u = phi / (2.0f * kPi);
    v = theta / kPi;
}

// The 64-byte shading record of a triangle (DeviceScene::tri_shade) in registers.  k_wf_shade fetches it with the four
// lanes of a quad working together (tri_rec_quad_fetch, PRE = true); for the other callers the record is read here.
struct TriRec { float4 a, b, c, e; };

// Position, face normal and material of the winning primitive.
template <bool PRE = false>
CRT_DEV void resolve_hit(const DeviceScene &sc, V3 o, V3 d, float t, uint32_t id, HitRecord &rec, const TriRec &pre = TriRec()) {
    uint32_t kind = id >> 30, idx = id & 0x3FFFFFFFu;
    rec.u = 0.0f; rec.v = 0.0f;
    rec.position = o + t * d;
    if (kind == CRT_ID_KIND_TRI) {                       // compute.comp:303-307
        const float4 *p = sc.tri_shade + 4 * (size_t)idx;
        const float4 a = PRE ? pre.a : p[0], b = PRE ? pre.b : p[1], c = PRE ? pre.c : p[2];
        const V3 e10 = mk3(a.w, b.x, b.y), e20 = mk3(b.z, b.w, c.x);      // v1 - v0, v2 - v0 (bvh_build.hip: k_tri_bounds)
        set_face_normal(d, cross3(e20, e10), rec);       // not normalised (sic)
        rec.materialIndex = f2u(c.y);
    } else if (kind == CRT_ID_KIND_SPHERE) {             // compute.comp:241-245
        float4 cr = *(const float4 *)(sc.spheres + idx);
        V3 outward = (rec.position - mk3(cr.x, cr.y, cr.z)) / cr.w;
        set_face_normal(d, outward, rec);
        rec.materialIndex = sc.spheres[idx].materialIndex;
    } else {                                             // compute.comp:270-273
        const crt_quad &q = sc.quads[idx];
        set_face_normal(d, ld3(q.normal), rec);
        rec.materialIndex = q.materialIndex;
    }
}

// Texture coordinates of the hit; only evaluated for a textured material (the values are the ones the
// shader computes at every hit and then ignores for untextured materials).
template <bool PRE = false>
CRT_DEV void resolve_uv(const DeviceScene &sc, V3 o, V3 d, uint32_t id, HitRecord &rec, const TriRec &pre = TriRec()) {
    uint32_t kind = id >> 30, idx = id & 0x3FFFFFFFu;
    if (kind == CRT_ID_KIND_TRI) {                       // compute.comp:309-314
        const float4 *p = sc.tri_shade + 4 * (size_t)idx;
        const float4 a = PRE ? pre.a : p[0], b = PRE ? pre.b : p[1], c = PRE ? pre.c : p[2], e = PRE ? pre.e : p[3];
        const V3 v0 = mk3(a.x, a.y, a.z), e10 = mk3(a.w, b.x, b.y), e20 = mk3(b.z, b.w, c.x);
        float tt, beta, gamma, invDet;
        tri_t(o, d, v0, e10, e20, tt, beta, gamma, invDet);
        beta = beta * invDet; gamma = gamma * invDet;
        float alpha = 1.0f - beta - gamma;
        rec.u = (alpha * c.z + beta * c.w) + gamma * e.x;
        rec.v = (alpha * e.y + beta * e.z) + gamma * e.w;
    } else if (kind == CRT_ID_KIND_SPHERE) {             // compute.comp:247-250
        float4 cr = *(const float4 *)(sc.spheres + idx);
        V3 outward = (rec.position - mk3(cr.x, cr.y, cr.z)) / cr.w;
        sphere_uv(outward, rec.u, rec.v);
    } else {                                             // compute.comp:264-266,274-275
        const crt_quad &q = sc.quads[idx];
        V3 planar = rec.position - ld3(q.corner);
        rec.u = dot3(ld3(q.w), cross3(planar, ld3(q.v)));
        rec.v = dot3(ld3(q.w), cross3(ld3(q.u), planar));
    }
}

}  // namespace crt
