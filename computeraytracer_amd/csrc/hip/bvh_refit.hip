// bvh_refit.hip -- crt_bvh_refit: the boxes of the existing 8-wide tree recomputed bottom-up from the primitives as they
// are now (after crt_tris_animate / crt_tris_update), for callers that move geometry a little every frame and do not want
// to pay for a rebuild (SURVEY.md 8(b), row "BVH": crt_bvh_refit/rebuild).  The reference builds once and never refits
// (engine/src/vk_engine.cpp:748-934), so there is no counterpart to cite; closest-hit results do not depend on the tree, so
// a refitted tree renders the same bits as a rebuilt one -- only its traversal cost differs, which the call reports.
//
// Works on any 8-wide tree of this library (collapsed from the pair BVH or built directly): nodes are in BFS order and the
// builders leave the per-level node counts on the device (BuildState::wide_levels_dev), topology = the header word of a
// node (child_base << 8 | inner mask) + its eight link words (leaf: kind, count, first slot in the kind's sorted array),
// primitive order = the *_ids arrays.  One launch for the triangles' shading records (k_tri_bounds, as in a build), then
// one launch per level, deepest first: eight lanes per node, one per slot; a leaf slot re-packs its primitives' traversal
// records from the moved primitives and takes their padded boxes, an inner slot takes its child's exact box from the
// level below; wide_node_finish (bvh_wide_node.h) reduces, quantises and writes the node's 48 plane bytes and origin.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <string>

#include "build_internal.h"
#include "bvh_common.h"
#include "bvh_pack.h"
#include "bvh_wide_node.h"

namespace crt {

struct WideRefit {
    BuildInput in;
    const uint32_t *level_count;
    float4 *wide; const uint32_t *links; float4 *box;
    const float4 *tri_shade; float4 *tri_pack, *sphere_pack, *quad_pack;
    const uint32_t *tri_ids, *sphere_ids, *quad_ids;
    float *cost;      // [0] += sum of node areas, [1] = area of the root
};

__global__ __launch_bounds__(256) void k_wide_refit(WideRefit A, uint32_t level) {
    const uint32_t count = A.level_count[level];
    uint32_t first_id = 0;
    for (uint32_t l = 0; l < level; l++) first_id += A.level_count[l];
    const uint32_t sl = threadIdx.x & 7u;
    float area_sum = 0.0f;
    for (uint32_t g0 = blockIdx.x * 32u; g0 < count; g0 += gridDim.x * 32u) {      // block-uniform trip count
        const uint32_t grp = g0 + (threadIdx.x >> 3);
        const bool active = grp < count;
        const uint32_t id = first_id + (active ? grp : 0u);
        const uint32_t hdr = ((const uint32_t *)(A.wide + (size_t)kWideStrideF4 * id))[3];
        const uint32_t imask = hdr & 0xFFu, child_base = hdr >> 8;
        const uint32_t link = A.links[8 * (size_t)id + sl];
        float clo[3] = {3.402823466e38f, 3.402823466e38f, 3.402823466e38f}, chi[3] = {-3.402823466e38f, -3.402823466e38f, -3.402823466e38f};
        bool valid = false;
        if (active) {
            if ((imask >> sl) & 1u) {
                const uint32_t cid = child_base + (uint32_t)__builtin_popcount(imask & ((1u << sl) - 1u));
                const float4 a = A.box[2 * (size_t)cid], b = A.box[2 * (size_t)cid + 1];
                clo[0] = a.x; clo[1] = a.y; clo[2] = a.z; chi[0] = b.x; chi[1] = b.y; chi[2] = b.z;
                valid = true;
            } else if (link != kWideEmptyLink) {
                const uint32_t kind = (link >> 29) & 3u, cnt = ((link >> 26) & 7u) + 1u, s0 = link & ((1u << 26) - 1u);
                for (uint32_t j = 0; j < cnt; j++) {
                    float lo[3], hi[3];
                    const uint32_t s = s0 + j;
                    if (kind == CRT_ID_KIND_TRI) {
                        // pack_prim's triangle branch: record head of the shading record, box from v0, v0 + e10, v0 + e20
                        const uint32_t i = A.tri_ids[s];
                        const float4 *tp = A.tri_shade + 4 * (size_t)i;
                        const float4 a = tp[0], b = tp[1], c = tp[2];
                        float4 *o = A.tri_pack + 3 * (size_t)s;
                        o[0] = a; o[1] = b;
                        o[2] = make_float4(c.x, __uint_as_float(CRT_MAKE_ID(CRT_ID_KIND_TRI, i)), 0.0f, 0.0f);
                        const float v0[3] = {a.x, a.y, a.z}, v1[3] = {a.x + a.w, a.y + b.x, a.z + b.y}, v2[3] = {a.x + b.z, a.y + b.w, a.z + c.x};
                        for (int k = 0; k < 3; k++) { lo[k] = fminf(fminf(v0[k], v1[k]), v2[k]); hi[k] = fmaxf(fmaxf(v0[k], v1[k]), v2[k]); }
                        pad_box(lo, hi);
                    } else if (kind == CRT_ID_KIND_SPHERE) {
                        const uint32_t g = A.sphere_ids[s];
                        const crt_sphere &sp = A.in.spheres[g];
                        A.sphere_pack[s] = make_float4(sp.center[0], sp.center[1], sp.center[2], sp.radius);
                        prim_box(A.in, g, lo, hi);
                    } else {
                        const uint32_t i = A.quad_ids[s];
                        const crt_quad &q = A.in.quads[i];
                        float4 *o = A.quad_pack + 4 * (size_t)s;
                        o[0] = make_float4(q.normal[0], q.normal[1], q.normal[2], q.d);
                        o[1] = make_float4(q.corner[0], q.corner[1], q.corner[2], q.u[0]);
                        o[2] = make_float4(q.u[1], q.u[2], q.v[0], q.v[1]);
                        o[3] = make_float4(q.v[2], q.w[0], q.w[1], q.w[2]);
                        prim_box(A.in, A.in.n_spheres + i, lo, hi);
                    }
#pragma unroll
                    for (int a = 0; a < 3; a++) { clo[a] = fminf(clo[a], lo[a]); chi[a] = fmaxf(chi[a], hi[a]); }
                }
                valid = true;
            }
        }
        float area = 0.0f;
        wide_node_finish(active, sl, id, hdr, clo, chi, valid, A.wide, A.box, &area);
        if (sl == 0u) area_sum += area;
    }
    if (A.cost) {
        for (int off = 32; off > 0; off >>= 1) area_sum += __shfl_xor(area_sum, off);
        if ((threadIdx.x & 63u) == 0u && area_sum != 0.0f) {
            atomicAdd(&A.cost[0], area_sum);
            if (level == 0u) A.cost[1] = area_sum;      // level 0 is the root alone
        }
    }
}

__global__ void k_refit_cost_init(float *cost) { if (threadIdx.x < 2) cost[threadIdx.x] = 0.0f; }

#define HIP_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { err = std::string(#expr) + ": " + hipGetErrorString(e_); return false; } } while (0)

// bvh_build.hip
void launch_tri_records(hipStream_t stream, const BuildInput &in, BuildState &st);

bool refit_wide(hipStream_t stream, const BuildInput &in, BuildState &st, float *cost_dev, std::string &err) {
    if (st.n_prims == 0) return true;
    if (st.n_wide == 0 || st.wide_levels == 0) { err = "no 8-wide tree to refit (build with CRT_BVH_WIDE or CRT_BVH_WIDE_DIRECT first)"; return false; }
    if (in.n_spheres + in.n_quads + in.n_tris != st.n_prims) { err = "the scene changed size since the build"; return false; }
    // exact node boxes: two float4 per node (the direct build keeps them here too)
    const size_t need = (size_t)st.wide_cap * 2 * sizeof(float4);
    if (st.nlo.bytes < need || !st.nlo.ptr) {
        if (st.nlo.ptr) { hipError_t e = hipFree(st.nlo.ptr); (void)e; st.nlo.ptr = nullptr; st.nlo.bytes = 0; }
        HIP_TRY(hipMalloc(&st.nlo.ptr, need < 256 ? 256 : need));
        st.nlo.bytes = need < 256 ? 256 : need;
    }
    launch_tri_records(stream, in, st);      // the moved triangles' 64-byte shading records (and centroids nobody reads)
    const uint32_t *links = (const uint32_t *)((const float4 *)st.wide.ptr + (size_t)kWideStrideF4 * st.wide_cap);
    WideRefit A{in, (const uint32_t *)st.wide_levels_dev.ptr, (float4 *)st.wide.ptr, links, (float4 *)st.nlo.ptr,
                (const float4 *)st.tri_shade.ptr, (float4 *)st.tri_pack.ptr, (float4 *)st.sphere_pack.ptr, (float4 *)st.quad_pack.ptr,
                (const uint32_t *)st.tri_ids.ptr, (const uint32_t *)st.sphere_ids.ptr, (const uint32_t *)st.quad_ids.ptr, cost_dev};
    if (cost_dev) hipLaunchKernelGGL(k_refit_cost_init, dim3(1), dim3(64), 0, stream, cost_dev);
    // levels are at most 8 times the one above and at most wide_cap nodes; the kernel is grid-stride and reads the real count
    for (uint32_t l = st.wide_levels; l-- > 0;) {
        uint64_t bound = 1;
        for (uint32_t k = 0; k < l && bound < st.wide_cap; k++) bound *= 8;
        const uint32_t nodes = (uint32_t)std::min<uint64_t>(bound, st.wide_cap);
        hipLaunchKernelGGL(k_wide_refit, dim3(std::min<uint32_t>((nodes + 31u) / 32u, 16384u)), dim3(256), 0, stream, A, l);
    }
    HIP_TRY(hipGetLastError());
    return true;
}

}  // namespace crt
