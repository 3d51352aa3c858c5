// bvh_build.hip -- device-side BVH construction (replaces the host builder
// raytracing/src/bounding_volume_hierarchy.cpp:5-130, rows B1-B3 of SURVEY.md
// section 8(a)).
//
// Pipeline (all on the context's stream; steps 1-7 without host round trips):
//   1. k_prim_bounds   padded AABB + centroid per primitive, scene centroid bounds (atomic min/max)
//   2. k_morton        Morton code of the centroid, 21 bits per axis on a grid of cubic cells (63-bit keys)
//   3. radix sort      (code, primitive) pairs, rocPRIM
//   4. kind ranks      exclusive scans of "is sphere" / "is quad" over the sorted order, rocPRIM
//   5. k_pack          traversal records per kind in sorted order (tri 48 B, sphere 16 B, quad 64 B)
//   6. k_hierarchy     Karras 2012: one thread per inner node finds its range, split and children
//   7. k_refit_leaves / k_refit_resume   bottom-up in stages that never leave workgroup scope: the second climber to
//                      reach a node unions its children's boxes and writes the node's 64-byte sibling pair
//                      (the traversal format)
//   8. optional: SAH refinement (bvh_sah.hip), collapse into 8-wide quantised nodes (bvh_wide.hip)
// The reference's tree is different (full-sweep SAH, threaded links); closest hits do not depend on
// the tree because acceptance is order-independent (intersect.h).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <string>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include "build_internal.h"
#include "bvh_common.h"
#include "bvh_pack.h"

namespace crt {

// rocPRIM's sort of (code, primitive) pairs; below a million keys it is a merge sort whose tile sizes can be set here.
// Measured at 225 k pairs (whole static build, 12 builds each): default 0.72-0.73 ms; sorted tiles of 4096 keys (256 x 16,
// 512 x 8, 1024 x 4: two merge rounds fewer) 0.73-0.75 ms; 2048-key tiles with 1024-key merges 0.76 ms.  The default stays.
#ifdef CRT_MS_SORT_ITEMS
using PairSortConfig = rocprim::radix_sort_config<rocprim::default_config,
    rocprim::merge_sort_config<512, CRT_MS_SORT_BLOCK, CRT_MS_SORT_ITEMS, 128, CRT_MS_MERGE_BLOCK, CRT_MS_MERGE_ITEMS>, rocprim::default_config, 1024 * 1024>;
#else
using PairSortConfig = rocprim::default_config;
#endif

__global__ void k_init_bounds(uint32_t *b) {
    if (threadIdx.x < 3) b[threadIdx.x] = 0xFFFFFFFFu;       // min
    else if (threadIdx.x < 6) b[threadIdx.x] = 0u;           // max
}

// Shading record of triangle i (input order, 64 bytes = half a cache line, never straddling one): everything k_wf_shade
// needs of a hit triangle in four 16-byte loads from ONE line: {v0.xyz, e10.x} {e10.yz, e20.xy} {e20.z, materialIndex,
// u0, u1} {u2, v0, v1, v2}.  The 96-byte upload record put materialIndex 92 bytes after v0 (a second line for every
// other triangle) and the shader's two edge subtractions in front of every hit; e10 / e20 are those same IEEE
// subtractions (compute.comp:281-282), done once here.

// Block-level min/max of the centroids (wave shuffles, then LDS), one atomic per block and axis: the six result words
// are a single hot line, so the number of atomics, not their width, sets the time of the kernels below.
__device__ __forceinline__ void reduce_centroid_bounds(uint32_t cmn[3], uint32_t cmx[3], uint32_t *scene_bounds) {
    __shared__ uint32_t s_red[6][4];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int a = 0; a < 3; a++) {
        uint32_t mn = cmn[a], mx = cmx[a];
        for (int off = 32; off > 0; off >>= 1) {
            mn = min(mn, (uint32_t)__shfl_xor((int)mn, off));
            mx = max(mx, (uint32_t)__shfl_xor((int)mx, off));
        }
        if (lane == 0) { s_red[a][wave] = mn; s_red[3 + a][wave] = mx; }
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        const uint32_t nw = (blockDim.x + 63) / 64;
        uint32_t v = s_red[threadIdx.x][0];
        for (uint32_t w = 1; w < nw; w++) v = threadIdx.x < 3 ? min(v, s_red[threadIdx.x][w]) : max(v, s_red[threadIdx.x][w]);
        if (threadIdx.x < 3) atomicMin(&scene_bounds[threadIdx.x], v); else atomicMax(&scene_bounds[threadIdx.x], v);
    }
}

// Spheres and quads (primitives [0, n) of the build's numbering; the triangles follow and go through k_tri_bounds).
// BOXES = false (CRT_BVH_WIDE_DIRECT, which takes the boxes from the packed records later): only the centroid is kept,
// in plo: 16 instead of 32 bytes written here and read by k_morton.
template <bool BOXES>
__global__ __launch_bounds__(256) void k_prim_bounds(BuildInput in, uint32_t n, float4 *plo, float4 *phi, uint32_t *scene_bounds) {
    // grid-stride: the grid is capped (build_bvh) so that the atomics stay a few thousand however large n is
    uint32_t cmn[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, cmx[3] = {0u, 0u, 0u};
    for (uint32_t g = blockIdx.x * blockDim.x + threadIdx.x; g < n; g += gridDim.x * blockDim.x) {
        float lo[3], hi[3];
        prim_box(in, g, lo, hi);
        if (BOXES) {
            plo[g] = make_float4(lo[0], lo[1], lo[2], 0.0f);
            phi[g] = make_float4(hi[0], hi[1], hi[2], 0.0f);
        } else plo[g] = make_float4(0.5f * lo[0] + 0.5f * hi[0], 0.5f * lo[1] + 0.5f * hi[1], 0.5f * lo[2] + 0.5f * hi[2], 0.0f);
        for (int a = 0; a < 3; a++) { const uint32_t e = enc_f(0.5f * lo[a] + 0.5f * hi[a]); cmn[a] = min(cmn[a], e); cmx[a] = max(cmx[a], e); }
    }
    reduce_centroid_bounds(cmn, cmx, scene_bounds);
}

// Triangles: box (or centroid), centroid bounds and the 64-byte shading record (layout above).  A wave
// moves its 64 records through LDS, 6 KiB in and 4 KiB out, so that global loads and stores are 16 bytes per lane over
// consecutive addresses and every line crosses the L1 once; one record per lane straight from global memory is six
// accesses at a 96-byte stride whose lines do not survive in the L1 between the first and the sixth (k_animate_tris has
// the numbers).  first = index of triangle 0 in the build's numbering (spheres + quads).
template <bool BOXES>
__global__ __launch_bounds__(256) void k_tri_bounds(const crt_tri *tris, uint32_t n_tris, uint32_t first, float4 *plo, float4 *phi, float4 *tri_shade, uint32_t *scene_bounds) {
    __shared__ float4 s_rec[4][64 * 6];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float4 *sw = s_rec[wave];
    uint32_t cmn[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, cmx[3] = {0u, 0u, 0u};
    const uint32_t n_chunks = (n_tris + 63u) / 64u, per_round = gridDim.x * 4u;
    for (uint32_t round = 0; round * per_round < n_chunks; round++) {      // same trip count for every wave of the block (barriers inside)
        const uint32_t chunk = round * per_round + blockIdx.x * 4u + wave;
        const uint32_t base = chunk * 64u;
        const uint32_t cnt = chunk < n_chunks ? min(64u, n_tris - base) : 0u;
        const float4 *src = (const float4 *)(tris + base);
#pragma unroll
        for (uint32_t k = 0; k < 6; k++) { const uint32_t idx = k * 64 + lane; if (idx < cnt * 6) sw[idx] = src[idx]; }
        __syncthreads();
        float4 o0, o1, o2, o3;
        if (lane < cnt) {
            const float4 v0 = sw[lane * 6], v1 = sw[lane * 6 + 1], v2 = sw[lane * 6 + 2], tu = sw[lane * 6 + 3], tv = sw[lane * 6 + 4], last = sw[lane * 6 + 5];
            float lo[3], hi[3];
            lo[0] = fminf(fminf(v0.x, v1.x), v2.x); hi[0] = fmaxf(fmaxf(v0.x, v1.x), v2.x);
            lo[1] = fminf(fminf(v0.y, v1.y), v2.y); hi[1] = fmaxf(fmaxf(v0.y, v1.y), v2.y);
            lo[2] = fminf(fminf(v0.z, v1.z), v2.z); hi[2] = fmaxf(fmaxf(v0.z, v1.z), v2.z);
            pad_box(lo, hi);
            const uint32_t g = first + base + lane;
            if (BOXES) {
                plo[g] = make_float4(lo[0], lo[1], lo[2], 0.0f);
                phi[g] = make_float4(hi[0], hi[1], hi[2], 0.0f);
            } else plo[g] = make_float4(0.5f * lo[0] + 0.5f * hi[0], 0.5f * lo[1] + 0.5f * hi[1], 0.5f * lo[2] + 0.5f * hi[2], 0.0f);
            for (int a = 0; a < 3; a++) { const uint32_t e = enc_f(0.5f * lo[a] + 0.5f * hi[a]); cmn[a] = min(cmn[a], e); cmx[a] = max(cmx[a], e); }
            float e20z;
            tri_record_head(v0, v1, v2, o0, o1, e20z);
            o2 = make_float4(e20z, last.w, tu.x, tu.y);
            o3 = make_float4(tu.z, tv.x, tv.y, tv.z);
        }
        __syncthreads();
        if (lane < cnt) { sw[lane * 4] = o0; sw[lane * 4 + 1] = o1; sw[lane * 4 + 2] = o2; sw[lane * 4 + 3] = o3; }
        __syncthreads();
        float4 *dst = tri_shade + 4 * (size_t)base;
#pragma unroll
        for (uint32_t k = 0; k < 4; k++) { const uint32_t idx = k * 64 + lane; if (idx < cnt * 4) dst[idx] = sw[idx]; }
        __syncthreads();
    }
    reduce_centroid_bounds(cmn, cmx, scene_bounds);
}

// Spread the low 21 bits of v to every third bit.
__device__ __forceinline__ uint64_t expand_bits21(uint64_t v) {
    v &= 0x1FFFFFull;
    v = (v | v << 32) & 0x1F00000000FFFFull;
    v = (v | v << 16) & 0x1F0000FF0000FFull;
    v = (v | v << 8) & 0x100F00F00F00F00Full;
    v = (v | v << 4) & 0x10C30C30C30C30C3ull;
    v = (v | v << 2) & 0x1249249249249249ull;
    return v;
}

// Morton code of the centroid, `bits` per axis (cells finer than the
// primitives' spacing keep equal codes, which the hierarchy can only split by index, rare).
// With idx_bits != 0 the key carries the primitive's index below the code (code << idx_bits | g): the sort then moves
// keys only, over the code's bits, and k_hierarchy recovers the order from the low bits.
__global__ void k_morton(uint32_t n, uint32_t bits, uint32_t idx_bits, const float4 *plo, const float4 *phi,
                         const uint32_t *scene_bounds, uint64_t *keys, uint32_t *vals) {
    uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n) return;
    float c[3];
    if (phi) {
        const float4 lo = plo[g], hi = phi[g];
        c[0] = 0.5f * lo.x + 0.5f * hi.x; c[1] = 0.5f * lo.y + 0.5f * hi.y; c[2] = 0.5f * lo.z + 0.5f * hi.z;
    } else { const float4 ce = plo[g]; c[0] = ce.x; c[1] = ce.y; c[2] = ce.z; }      // centroids only (k_prim_bounds<false>)
    const float cells = (float)(1u << bits);
    uint64_t q[3];
    // one scale for all axes (cubic Morton cells): normalising each axis by its own extent would spend as many splits
    // on the thin axis of a flat scene (a terrain, a city) as on the long ones and cut it into overlapping slabs
    float ext = 0.0f;
    for (int a = 0; a < 3; a++) ext = fmaxf(ext, dec_f(scene_bounds[3 + a]) - dec_f(scene_bounds[a]));
    for (int a = 0; a < 3; a++) {
        float mn = dec_f(scene_bounds[a]);
        float f = ext > 0.0f ? (c[a] - mn) / ext : 0.0f;
        q[a] = (uint64_t)fminf(fmaxf(f * cells, 0.0f), cells - 1.0f);
    }
    const uint64_t code = (expand_bits21(q[0]) << 2) | (expand_bits21(q[1]) << 1) | expand_bits21(q[2]);
    if (idx_bits) { keys[g] = code << idx_bits | g; return; }
    keys[g] = code;
    vals[g] = g;
}

__global__ void k_kind_flags(uint32_t n, const uint32_t *vals, uint32_t ns, uint32_t nq, uint32_t *is_sphere, uint32_t *is_quad) {
    uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    uint32_t g = vals[s];
    is_sphere[s] = g < ns ? 1u : 0u;
    is_quad[s] = (g >= ns && g < ns + nq) ? 1u : 0u;
}

// CRT_BVH_WIDE_DIRECT: the sorted primitive index is the low part of the key
__global__ void k_kind_flags_keys(uint32_t n, const uint64_t *keys, uint32_t idx_mask, uint32_t ns, uint32_t nq, uint32_t *is_sphere, uint32_t *is_quad) {
    uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    uint32_t g = (uint32_t)keys[s] & idx_mask;
    is_sphere[s] = g < ns ? 1u : 0u;
    is_quad[s] = (g >= ns && g < ns + nq) ? 1u : 0u;
}

// SAH path (and n == 1): records only; leaf_link[s] = slot inside the kind's array, leaf_count[s] = kind << 28 | 1.
__global__ void k_pack(BuildInput in, uint32_t n, const uint32_t *vals, const uint32_t *rank_sphere, const uint32_t *rank_quad,
                       const float4 *tri_shade, PackOut out, uint32_t *leaf_link, uint32_t *leaf_count) {
    uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    uint32_t kind, slot; float lo[3], hi[3];
    pack_prim<false>(in, n, s, vals[s], rank_sphere, rank_quad, tri_shade, out, kind, slot, lo, hi);
    leaf_link[s] = slot; leaf_count[s] = (kind << 28) | 1u;
}

// delta(i, j): length of the common prefix of keys i and j, ties broken by the index (Karras 2012, section 4).
// The keys around a workgroup's 256 nodes sit in LDS (kHierWin on either side): most probes of the searches below stay
// inside that window, and a wave is as slow as its lane with the longest chain of dependent probes.
constexpr int kHierWin = 384;
constexpr int kHierKeys = 256 + 2 * kHierWin;
__device__ __forceinline__ int delta(const uint64_t *keys, const uint64_t *s_keys, int win_lo, int n, uint64_t a, int i, int j) {
    if (j < 0 || j >= n) return -1;
    const int w = j - win_lo;
    const uint64_t b = (w >= 0 && w < kHierKeys) ? s_keys[w] : keys[j];
    if (a == b) return 64 + __clz((uint32_t)i ^ (uint32_t)j);
    return __clzll((long long)(a ^ b));
}

// Node numbering: inner i in [0, n-2], leaf j -> (n - 1) + j.  parent[] is indexed by that numbering.
// unpack_vals (packed keys only): vals[i] = the index in the key's low bits, written on the way.
__global__ __launch_bounds__(256) void k_hierarchy(int n, const uint64_t *keys, uint32_t *child_l, uint32_t *child_r, uint32_t *parent,
                                                   uint32_t *range_lo, uint32_t *range_hi, uint32_t *unpack_vals, uint32_t idx_mask) {
    __shared__ uint64_t s_keys[kHierKeys];
    const int win_lo = (int)(blockIdx.x * blockDim.x) - kHierWin;
    for (int t = threadIdx.x; t < kHierKeys; t += 256) { const int g = win_lo + t; s_keys[t] = (g >= 0 && g < n) ? keys[g] : 0ull; }
    __syncthreads();
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n - 1) return;
    const uint64_t ki = s_keys[i - win_lo];
    if (unpack_vals) {
        unpack_vals[i] = (uint32_t)ki & idx_mask;
        if (i == n - 2) unpack_vals[n - 1] = (uint32_t)keys[n - 1] & idx_mask;
    }
#define DELTA(j) delta(keys, s_keys, win_lo, n, ki, i, (j))
    int d = (DELTA(i + 1) - DELTA(i - 1)) >= 0 ? 1 : -1;
    int dmin = DELTA(i - d);
    int lmax = 2;
    while (DELTA(i + lmax * d) > dmin) lmax *= 2;
    int l = 0;
    for (int t = lmax / 2; t >= 1; t /= 2)
        if (DELTA(i + (l + t) * d) > dmin) l += t;
    int j = i + l * d;
    int dnode = DELTA(j);
    int s = 0;
    for (int t = (l + 1) / 2;; t = (t + 1) / 2) {
        if (DELTA(i + (s + t) * d) > dnode) s += t;
        if (t <= 1) break;
    }
#undef DELTA
    int gamma = i + s * d + min(d, 0);
    uint32_t left = (min(i, j) == gamma) ? (uint32_t)(n - 1 + gamma) : (uint32_t)gamma;
    uint32_t right = (max(i, j) == gamma + 1) ? (uint32_t)(n - 1 + gamma + 1) : (uint32_t)(gamma + 1);
    child_l[i] = left; child_r[i] = right;
    parent[left] = (uint32_t)i; parent[right] = (uint32_t)i;
    if (i == 0) parent[0] = 0xFFFFFFFFu;
    if (range_lo) { range_lo[i] = (uint32_t)min(i, j); range_hi[i] = (uint32_t)max(i, j); }     // sorted positions the node covers
}

// Per-subtree summary carried up by the refit: primitive count, SAH cost, kind (or kMixed), first leaf of its
// (contiguous) range in sorted order, and whether the subtree is emitted as one multi-primitive leaf.
constexpr uint32_t kMixed = 0xFFu;
constexpr uint32_t kCollapsed = 0x100u;
struct Sub { uint32_t count; float cost; uint32_t kind_flags; uint32_t first; };      // first = slot of the range's first primitive in its kind's array

__device__ __forceinline__ float4 sub_pack(Sub s) {
    return make_float4(__uint_as_float(s.count), s.cost, __uint_as_float(s.kind_flags), __uint_as_float(s.first));
}
__device__ __forceinline__ Sub sub_unpack(float4 v) {
    Sub s; s.count = __float_as_uint(v.x); s.cost = v.y; s.kind_flags = __float_as_uint(v.z); s.first = __float_as_uint(v.w); return s;
}

// Child record inside its parent's pair: an inner child links to its own pair, a leaf (single primitive or a
// collapsed subtree) to its first slot in the kind's sorted array.
__device__ __forceinline__ void write_child(float4 *pair_half, int n, uint32_t child, float4 lo, float4 hi, Sub sub) {
    uint32_t link, count;
    if (child >= (uint32_t)(n - 1) || (sub.kind_flags & kCollapsed)) { link = sub.first; count = ((sub.kind_flags & 0xFFu) << 28) | sub.count; }
    else { link = child; count = 0u; }
    pair_half[0] = make_float4(lo.x, lo.y, lo.z, __uint_as_float(link));
    pair_half[1] = make_float4(hi.x, hi.y, hi.z, __uint_as_float(count));
}

// Bottom-up refit.  A climber starts at a finished node; at each inner node the first arrival stops, the second
// (which therefore knows both subtrees are finished) computes the node and climbs on.  Hand-off of the sibling's
// data: stores, release fence, returning atomic; the winner fences again (acquire) before loading what the other
// climber stored.  The SCOPE of that hand-off is what this costs: an agent-scope release writes this XCD's L2
// back (the eight L2s are not coherent with each other), and with one such fence per thread and level a
// 10 M-triangle refit took 79 ms (47 ms with agent scope only on the ~4 % of nodes that span a workgroup boundary).
// So the refit runs in stages and never leaves workgroup scope:
//   stage 1   a workgroup owns kRefitLeaves consecutive leaves and climbs while the parent's range stays inside them: both
//             climbers of such a node are waves of this workgroup, on one CU behind one L1.  A climber whose parent
//             reaches outside parks its (finished) node in the workgroup's continuation list;
//   stage s   a workgroup owns 64 ranges of stage s-1, resumes their parked climbers and climbs while the parent
//             stays inside its 64x larger range.  The kernel boundary between stages makes the earlier stores visible.
// A range of the sorted leaves decomposes into at most 2 * depth maximal subtrees and the tree is at most 95 deep
// (up to 63 Morton bits + 32 index bits of tie-break), so a list of 256 entries per workgroup cannot overflow.
//
// SAH leaf collapse (CRT_BVH_LEAF_COLLAPSE): with C(leaf of n) = Ci * n * A and C(inner) = Ct * A + C(left) + C(right)
// (A = surface area; entering a node costs one pair fetch + two slab tests = Ct), a subtree of one kind with at most
// max_leaf primitives becomes a single leaf when that is not more expensive.  Its primitives are contiguous in the
// sorted arrays because an LBVH subtree covers a contiguous Morton range.
#ifndef CRT_REFIT_LEAVES
#define CRT_REFIT_LEAVES 512
#endif
constexpr uint32_t kRefitLeaves = CRT_REFIT_LEAVES;     // leaves per stage-1 workgroup (10 M triangles: 128 / 256 / 512 / 1024 build in 2.86 / 2.70 / 2.62 / 2.84 ms)
constexpr uint32_t kRefitFan = 64;         // ranges of the previous stage per workgroup of a later stage
constexpr uint32_t kRefitCap = 256;        // continuation entries per workgroup

struct RefitArgs {
    int n;
    const uint32_t *vals;
    BuildInput in; const uint32_t *rank_sphere, *rank_quad; const float4 *tri_shade; PackOut out;      // stage 1 packs the leaves' records on the way
    const uint32_t *child_l, *child_r, *parent, *range_lo, *range_hi;
    float4 *nlo, *nhi, *ninfo; uint32_t *flags; float4 *pairs;
    uint32_t max_leaf; float c_isect, c_trav;
};

__device__ __forceinline__ void refit_climb(const RefitArgs &A, uint32_t node, float4 lo, float4 hi, Sub sub,
                                            uint32_t blk_lo, uint32_t blk_hi, uint32_t *s_count, uint32_t *list_out) {
    const int n = A.n;
    for (;;) {
        const uint32_t p = A.parent[node];
        if (p == 0xFFFFFFFFu) return;                                  // node is the root
        if (A.range_lo[p] < blk_lo || A.range_hi[p] > blk_hi) {        // the parent belongs to a later stage
            const uint32_t k = atomicAdd(s_count, 1u);
            if (k < kRefitCap) list_out[k] = node;
            return;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        const uint32_t old = __hip_atomic_fetch_add(&A.flags[p], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (old == 0u) return;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        const uint32_t cl = A.child_l[p], cr = A.child_r[p];
        const uint32_t sib = cl == node ? cr : cl;
        const float4 slo = A.nlo[sib], shi = A.nhi[sib];
        const Sub ssub = sub_unpack(A.ninfo[sib]);
        const bool mine_left = cl == node;
        const float4 l_lo = mine_left ? lo : slo, l_hi = mine_left ? hi : shi;
        const float4 r_lo = mine_left ? slo : lo, r_hi = mine_left ? shi : hi;
        const Sub l_sub = mine_left ? sub : ssub, r_sub = mine_left ? ssub : sub;
        write_child(A.pairs + 4 * (size_t)p, n, cl, l_lo, l_hi, l_sub);
        write_child(A.pairs + 4 * (size_t)p + 2, n, cr, r_lo, r_hi, r_sub);
        lo = make_float4(fminf(l_lo.x, r_lo.x), fminf(l_lo.y, r_lo.y), fminf(l_lo.z, r_lo.z), 0.0f);
        hi = make_float4(fmaxf(l_hi.x, r_hi.x), fmaxf(l_hi.y, r_hi.y), fmaxf(l_hi.z, r_hi.z), 0.0f);
        const float area = half_area(lo, hi);
        const uint32_t lk = l_sub.kind_flags & 0xFFu, rk = r_sub.kind_flags & 0xFFu;
        sub.count = l_sub.count + r_sub.count;
        sub.first = l_sub.first;                       // the left child covers the lower part of the range
        const uint32_t kind = (lk == rk) ? lk : kMixed;
        const float as_inner = A.c_trav * area + l_sub.cost + r_sub.cost;
        const float as_leaf = A.c_isect * (float)sub.count * area;
        const bool collapse = kind != kMixed && sub.count <= A.max_leaf && as_leaf <= as_inner;
        sub.cost = collapse ? as_leaf : as_inner;
        sub.kind_flags = kind | (collapse ? kCollapsed : 0u);
        A.nlo[p] = lo; A.nhi[p] = hi; A.ninfo[p] = sub_pack(sub);
        node = p;
    }
}

// Stage 1: one thread per leaf, which first writes the leaf's traversal record (pack_prim).  Everything a climber hands to its sibling inside the workgroup's range stays in LDS
// (an inner node whose range lies in [blk_lo, blk_hi] has its own index in that interval: a Karras node sits at one
// end of its range), so a level costs one round of independent global loads (parent, range, children) plus the
// 128 bytes of the pair it emits; only parked nodes are written to the global node arrays for the next stage.
__global__ __launch_bounds__(kRefitLeaves) void k_refit_leaves(RefitArgs A, uint32_t *list_out, uint32_t *count_out) {
    __shared__ float4 s_lo[2 * kRefitLeaves], s_hi[2 * kRefitLeaves], s_info[2 * kRefitLeaves];      // first half: inner nodes, second half: leaves
    __shared__ uint32_t s_flag[kRefitLeaves];
    __shared__ uint32_t s_parent[kRefitLeaves], s_rlo[kRefitLeaves], s_rhi[kRefitLeaves], s_cl[kRefitLeaves], s_cr[kRefitLeaves];
    __shared__ uint32_t s_count;
    const uint32_t blk_lo = blockIdx.x * kRefitLeaves, blk_hi = blk_lo + kRefitLeaves - 1u;
    const uint32_t j = blk_lo + threadIdx.x;
    const int n = A.n;
    if (threadIdx.x == 0) s_count = 0;
    s_flag[threadIdx.x] = 0;
    // The topology of the inner nodes whose index lies in this block's interval, in one round of coalesced loads: every
    // node the climb below may enter is one of them, so a level costs LDS reads only (it used to be two dependent global
    // loads per level and climber, which made this kernel latency-bound: 698 us at 10 M leaves).
    if (j < (uint32_t)(n - 1)) {
        s_parent[threadIdx.x] = A.parent[j]; s_rlo[threadIdx.x] = A.range_lo[j]; s_rhi[threadIdx.x] = A.range_hi[j];
        s_cl[threadIdx.x] = A.child_l[j]; s_cr[threadIdx.x] = A.child_r[j];
    }
    __syncthreads();
    if (j < (uint32_t)n) {
        uint32_t node = (uint32_t)(n - 1) + j;
        uint32_t p = A.parent[node];
        // the leaf: its traversal record goes to its kind's array (k_pack's job on the SAH path) and its box comes from the
        // same read, instead of a second gather of 2 x 16 bytes out of two more cache lines
        uint32_t kind, slot; float blo[3], bhi[3];
        pack_prim<true>(A.in, (uint32_t)n, j, A.vals[j], A.rank_sphere, A.rank_quad, A.tri_shade, A.out, kind, slot, blo, bhi);
        float4 lo = make_float4(blo[0], blo[1], blo[2], 0.0f), hi = make_float4(bhi[0], bhi[1], bhi[2], 0.0f);
        Sub sub;
        sub.count = 1; sub.kind_flags = kind; sub.first = slot;
        sub.cost = A.c_isect * half_area(lo, hi);
        s_lo[kRefitLeaves + threadIdx.x] = lo; s_hi[kRefitLeaves + threadIdx.x] = hi; s_info[kRefitLeaves + threadIdx.x] = sub_pack(sub);
        uint32_t *list = list_out + (size_t)blockIdx.x * kRefitCap;
        for (;;) {
            if (p == 0xFFFFFFFFu) break;                                   // node is the root
            // a node whose range lies inside the block has its own index inside it (a Karras node sits at one end of its range)
            const uint32_t lp = p - blk_lo;
            bool park = p < blk_lo || p > blk_hi;
            uint32_t cl = 0, cr = 0;
            if (!park) { cl = s_cl[lp]; cr = s_cr[lp]; park = s_rlo[lp] < blk_lo || s_rhi[lp] > blk_hi; }
            if (park) {                                                    // the parent belongs to a later stage
                A.nlo[node] = lo; A.nhi[node] = hi; A.ninfo[node] = sub_pack(sub);
                const uint32_t k = atomicAdd(&s_count, 1u);
                if (k < kRefitCap) list[k] = node;
                break;
            }
            // only LDS data changes hands here: fence the local address space alone, the pair stores stay in flight
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
            const uint32_t old = __hip_atomic_fetch_add(&s_flag[lp], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (old == 0u) break;
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
            const bool mine_left = cl == node;
            const uint32_t sib = mine_left ? cr : cl;
            const uint32_t ls = sib >= (uint32_t)(n - 1) ? kRefitLeaves + (sib - (uint32_t)(n - 1) - blk_lo) : sib - blk_lo;
            const float4 slo = s_lo[ls], shi = s_hi[ls];
            const Sub ssub = sub_unpack(s_info[ls]);
            const float4 l_lo = mine_left ? lo : slo, l_hi = mine_left ? hi : shi;
            const float4 r_lo = mine_left ? slo : lo, r_hi = mine_left ? shi : hi;
            const Sub l_sub = mine_left ? sub : ssub, r_sub = mine_left ? ssub : sub;
            write_child(A.pairs + 4 * (size_t)p, n, cl, l_lo, l_hi, l_sub);
            write_child(A.pairs + 4 * (size_t)p + 2, n, cr, r_lo, r_hi, r_sub);
            lo = make_float4(fminf(l_lo.x, r_lo.x), fminf(l_lo.y, r_lo.y), fminf(l_lo.z, r_lo.z), 0.0f);
            hi = make_float4(fmaxf(l_hi.x, r_hi.x), fmaxf(l_hi.y, r_hi.y), fmaxf(l_hi.z, r_hi.z), 0.0f);
            const float area = half_area(lo, hi);
            const uint32_t lk = l_sub.kind_flags & 0xFFu, rk = r_sub.kind_flags & 0xFFu;
            sub.count = l_sub.count + r_sub.count;
            sub.first = l_sub.first;                       // the left child covers the lower part of the range
            const uint32_t kind = (lk == rk) ? lk : kMixed;
            const float as_inner = A.c_trav * area + l_sub.cost + r_sub.cost;
            const float as_leaf = A.c_isect * (float)sub.count * area;
            const bool collapse = kind != kMixed && sub.count <= A.max_leaf && as_leaf <= as_inner;
            sub.cost = collapse ? as_leaf : as_inner;
            sub.kind_flags = kind | (collapse ? kCollapsed : 0u);
            s_lo[lp] = lo; s_hi[lp] = hi; s_info[lp] = sub_pack(sub);
            node = p;
            p = s_parent[lp];
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) count_out[blockIdx.x] = min(s_count, kRefitCap);
}

// Later stages: workgroup b resumes the climbers parked by workgroups [b * kRefitFan, (b + 1) * kRefitFan) of the
// previous stage; `range` = leaves per workgroup of THIS stage.
__global__ __launch_bounds__(1024) void k_refit_resume(RefitArgs A, uint32_t n_prev, const uint32_t *list_in, const uint32_t *count_in,
                                                       uint32_t range, uint32_t *list_out, uint32_t *count_out) {
    __shared__ uint32_t s_count;
    if (threadIdx.x == 0) s_count = 0;
    __syncthreads();
    const uint32_t blk_lo = blockIdx.x * range, blk_hi = blk_lo + (range - 1u);
    // 16 threads per previous-stage range (a range parks ~2 log2(its size) climbers, so usually one pass)
    static_assert(kRefitFan * 16 == 1024, "one 16-thread group per previous-stage range");
    const uint32_t sb = blockIdx.x * kRefitFan + threadIdx.x / 16u;
    const uint32_t cnt = sb < n_prev ? count_in[sb] : 0u;
    for (uint32_t k = threadIdx.x % 16u; k < cnt; k += 16u) {
        const uint32_t node = list_in[(size_t)sb * kRefitCap + k];
        refit_climb(A, node, A.nlo[node], A.nhi[node], sub_unpack(A.ninfo[node]), blk_lo, blk_hi, &s_count,
                    list_out + (size_t)blockIdx.x * kRefitCap);
    }
    __syncthreads();
    if (threadIdx.x == 0) count_out[blockIdx.x] = min(s_count, kRefitCap);
}

// n == 1: a single leaf beside an empty leaf (count field = kind bits only, zero primitives).
__global__ void k_single_pair(const uint32_t *vals, const float4 *plo, const float4 *phi, const uint32_t *leaf_link,
                              const uint32_t *leaf_count, float4 *pairs) {
    if (threadIdx.x != 0) return;
    uint32_t g = vals[0];
    float4 lo = plo[g], hi = phi[g];
    pairs[0] = make_float4(lo.x, lo.y, lo.z, __uint_as_float(leaf_link[0]));
    pairs[1] = make_float4(hi.x, hi.y, hi.z, __uint_as_float(leaf_count[0]));
    pairs[2] = make_float4(0.0f, 0.0f, 0.0f, __uint_as_float(0u));
    pairs[3] = make_float4(0.0f, 0.0f, 0.0f, __uint_as_float(CRT_ID_KIND_TRI << 28));
}

// Config 5 animation: y = rest_y + amplitude * sin(frame + x + z) on every triangle vertex.
// sinf is OCML's accurate single-precision sine (full argument reduction), not the __sinf hardware approximation: the
// result feeds the BVH build and the tracer, and tests/test_gpu_parity.py checks it against float64 sin to 2 ulp.
// A wave moves its 64 triangles (6 KiB, contiguous) through LDS: global loads and stores are 16 bytes per lane over
// consecutive addresses, every line crosses the L1 once.  (One 96-byte record per lane straight from global memory is six
// 16-byte accesses at a 96-byte stride; with eight waves a CU their 48 KiB of lines do not survive in the 32 KiB L1
// between the first and the sixth access: 520 us for 10 M triangles.)
// RECORDS: the pass also leaves what a CRT_BVH_WIDE_DIRECT build or a crt_bvh_refit would compute from the moved triangles
// first (k_tri_bounds<false>): the 64-byte shading record, the centroid of the padded box and the centroid bounds, so the
// build that follows skips that kernel and the triangles are read once per frame instead of twice (10 M triangles:
// animation 0.34 -> 0.47 ms, build -0.35 ms).  Triangles-only scenes (first = 0).
template <bool RECORDS>
__global__ __launch_bounds__(256) void k_animate_tris(uint32_t n, const crt_tri *rest, crt_tri *out, float frame, float amplitude,
                                                      float4 *plo, float4 *tri_shade, uint32_t *scene_bounds) {
    __shared__ float4 s_rec[4][64 * 6];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float4 *sw = s_rec[wave];
    uint32_t cmn[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, cmx[3] = {0u, 0u, 0u};
    // grid-stride over 64-triangle chunks (the grid is capped when RECORDS, so that the bounds atomics stay a few thousand;
    // every wave of a block makes the same number of trips: barriers inside)
    const uint32_t n_chunks = (n + 63u) / 64u, per_round = gridDim.x * 4u;
    for (uint32_t round = 0; round * per_round < n_chunks; round++) {
        const uint32_t chunk = round * per_round + blockIdx.x * 4u + wave;
        const uint64_t base = (uint64_t)chunk * 64u;
        const uint32_t cnt = chunk < n_chunks ? (uint32_t)min((uint64_t)64, n - base) : 0u;
        const float4 *src = (const float4 *)(rest + base);
        float4 *dst = (float4 *)(out + base);
#pragma unroll
        for (uint32_t k = 0; k < 6; k++) { const uint32_t idx = k * 64 + lane; if (idx < cnt * 6) sw[idx] = src[idx]; }
        __syncthreads();
        float4 o0, o1, o2, o3;
        if (lane < cnt) {
            float4 v[3];
#pragma unroll
            for (uint32_t k = 0; k < 3; k++) {      // rows 0..2 of the record: v0, v1, v2
                float4 p = sw[lane * 6 + k];
                p.y = p.y + amplitude * sinf((frame + p.x) + p.z);
                sw[lane * 6 + k] = p;
                v[k] = p;
            }
            if (RECORDS) {      // k_tri_bounds<false>'s arithmetic on the moved vertices
                const float4 v0 = v[0], v1 = v[1], v2 = v[2], tu = sw[lane * 6 + 3], tv = sw[lane * 6 + 4], last = sw[lane * 6 + 5];
                float lo[3], hi[3];
                lo[0] = fminf(fminf(v0.x, v1.x), v2.x); hi[0] = fmaxf(fmaxf(v0.x, v1.x), v2.x);
                lo[1] = fminf(fminf(v0.y, v1.y), v2.y); hi[1] = fmaxf(fmaxf(v0.y, v1.y), v2.y);
                lo[2] = fminf(fminf(v0.z, v1.z), v2.z); hi[2] = fmaxf(fmaxf(v0.z, v1.z), v2.z);
                pad_box(lo, hi);
                plo[base + lane] = make_float4(0.5f * lo[0] + 0.5f * hi[0], 0.5f * lo[1] + 0.5f * hi[1], 0.5f * lo[2] + 0.5f * hi[2], 0.0f);
                for (int a = 0; a < 3; a++) { const uint32_t e = enc_f(0.5f * lo[a] + 0.5f * hi[a]); cmn[a] = min(cmn[a], e); cmx[a] = max(cmx[a], e); }
                float e20z;
                tri_record_head(v0, v1, v2, o0, o1, e20z);
                o2 = make_float4(e20z, last.w, tu.x, tu.y);
                o3 = make_float4(tu.z, tv.x, tv.y, tv.z);
            }
        }
        __syncthreads();
#pragma unroll
        for (uint32_t k = 0; k < 6; k++) { const uint32_t idx = k * 64 + lane; if (idx < cnt * 6) dst[idx] = sw[idx]; }
        __syncthreads();
        if (RECORDS) {
            if (lane < cnt) { sw[lane * 4] = o0; sw[lane * 4 + 1] = o1; sw[lane * 4 + 2] = o2; sw[lane * 4 + 3] = o3; }
            __syncthreads();
            float4 *rd = tri_shade + 4 * base;
#pragma unroll
            for (uint32_t k = 0; k < 4; k++) { const uint32_t idx = k * 64 + lane; if (idx < cnt * 4) rd[idx] = sw[idx]; }
            __syncthreads();
        }
    }
    if (RECORDS) reduce_centroid_bounds(cmn, cmx, scene_bounds);
}

#define HIP_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { err = std::string(#expr) + ": " + hipGetErrorString(e_); return false; } } while (0)

static bool ensure(DeviceBuffer &b, size_t bytes, std::string &err) {
    if (b.bytes >= bytes && b.ptr) return true;
    if (b.ptr) { hipError_t e = hipFree(b.ptr); (void)e; b.ptr = nullptr; b.bytes = 0; }
    size_t want = bytes < 256 ? 256 : bytes;
    HIP_TRY(hipMalloc(&b.ptr, want));
    b.bytes = want;
    return true;
}

// st (optional): a build state whose buffers can take the moved triangles' records (allocated by an earlier build of this
// scene, triangles only): the pass then writes them too and marks them fresh for the next direct build / refit.
bool animate_tris(hipStream_t stream, uint32_t n, const crt_tri *rest, crt_tri *out, float frame, float amplitude, BuildState *st, std::string &err) {
    if (n == 0) return true;
    const bool records = st && st->n_prims == n && st->tri_shade.ptr && st->tri_shade.bytes >= (size_t)n * 64 && st->plo.ptr &&
                         st->plo.bytes >= (size_t)n * 16 && st->scene_bounds.ptr;
    if (st) st->records_fresh = false;
    if (records) {
        hipLaunchKernelGGL(k_init_bounds, dim3(1), dim3(64), 0, stream, (uint32_t *)st->scene_bounds.ptr);
        hipLaunchKernelGGL(k_animate_tris<true>, dim3(std::min((n + 255u) / 256u, 4096u)), dim3(256), 0, stream, n, rest, out, frame, amplitude, (float4 *)st->plo.ptr,
                           (float4 *)st->tri_shade.ptr, (uint32_t *)st->scene_bounds.ptr);
        st->records_fresh = true;
    } else
        hipLaunchKernelGGL(k_animate_tris<false>, dim3((n + 255) / 256), dim3(256), 0, stream, n, rest, out, frame, amplitude, (float4 *)nullptr, (float4 *)nullptr,
                           (uint32_t *)nullptr);
    HIP_TRY(hipGetLastError());
    return true;
}

bool build_bvh(hipStream_t stream, const BuildInput &in, uint32_t flags, BuildState &st, std::string &err) {
    const uint32_t n = in.n_spheres + in.n_quads + in.n_tris;
    st.n_prims = n;
    st.n_pairs = n == 0 ? 0 : (n == 1 ? 1 : n - 1);
    if (st.wide_levels) st.wide_levels_hint = st.wide_levels;
    st.n_wide = 0; st.wide_levels = 0; st.levels_measured = false; st.pair_depth = 0; st.dp_fused = false;
    const bool wide = (flags & CRT_BVH_WIDE) != 0;
    // CRT_BVH_WIDE_DIRECT (bvh_direct.hip): the 8-wide tree straight from the sorted keys; no pair tree and none of its scratch
    const bool direct = (flags & CRT_BVH_WIDE_DIRECT) != 0;
    if (direct && (flags & (CRT_BVH_SAH_REFINE | CRT_BVH_LEAF_COLLAPSE))) { err = "CRT_BVH_WIDE_DIRECT builds from the Morton order alone: not with CRT_BVH_SAH_REFINE / CRT_BVH_LEAF_COLLAPSE"; return false; }
    st.pairs_valid = !direct;
    if (n == 0) return true;
    const uint32_t blocks = (n + 255) / 256;
    const size_t n4 = (size_t)n * sizeof(float4);
    const bool mixed = !(in.n_spheres == n || in.n_quads == n || in.n_tris == n);
    if (!ensure(st.plo, n4, err) || (!direct && !ensure(st.phi, n4, err))) return false;
    if (!ensure(st.keys, (size_t)n * 8, err) || !ensure(st.keys_alt, (size_t)n * 8, err)) return false;
    if (!direct || mixed) {
        if (!ensure(st.flag_a, (size_t)n * 4, err) || !ensure(st.flag_b, (size_t)n * 4, err)) return false;
        if (!ensure(st.rank_a, (size_t)n * 4, err) || !ensure(st.rank_b, (size_t)n * 4, err)) return false;
    }
    if (!direct) {
        if (!ensure(st.vals, (size_t)n * 4, err) || !ensure(st.vals_alt, (size_t)n * 4, err)) return false;
        if (!ensure(st.leaf_link, (size_t)n * 4, err) || !ensure(st.leaf_count, (size_t)n * 4, err)) return false;
        if (!ensure(st.child_l, (size_t)n * 4, err) || !ensure(st.child_r, (size_t)n * 4, err)) return false;
        if (!ensure(st.parent, (size_t)(2 * (size_t)n) * 4, err)) return false;
        if (!ensure(st.nlo, 2 * n4, err) || !ensure(st.nhi, 2 * n4, err) || !ensure(st.ninfo, 2 * n4, err)) return false;
        if (!ensure(st.pairs, (size_t)st.n_pairs * 4 * sizeof(float4), err)) return false;
    }
    if (!ensure(st.scene_bounds, 64, err)) return false;
    if (!ensure(st.tri_pack, (size_t)in.n_tris * 3 * sizeof(float4), err)) return false;
    if (!ensure(st.sphere_pack, (size_t)in.n_spheres * sizeof(float4), err)) return false;
    if (!ensure(st.quad_pack, (size_t)in.n_quads * 4 * sizeof(float4), err)) return false;
    if (!ensure(st.sphere_ids, (size_t)in.n_spheres * 4, err) || !ensure(st.quad_ids, (size_t)in.n_quads * 4, err)) return false;
    if (!ensure(st.tri_ids, (size_t)in.n_tris * 4, err)) return false;
    if (!ensure(st.tri_shade, (size_t)in.n_tris * 4 * sizeof(float4), err)) return false;

    // temp storage for rocPRIM (sized for the larger of sort and scan)
    size_t sort_bytes = 0, scan_bytes = 0;
    // The grid has cubic cells over the LARGEST extent of the centroids (k_morton), and one far-away primitive (the
    // reference scenes' ground sphere of radius 2000) must not leave a whole mesh in a single cell: 21 bits per axis
    // whenever the sort's cost does not depend on the key width (rocPRIM merge-sorts below about a million keys);
    // above that, four more than half of log2(n) (surfaces are 2-D sets), which saves a radix pass at 10 M.
    uint32_t bits = 21, half_log = 21;
    if (n > (1u << 20)) {
        half_log = 1;
        while ((1ull << (2 * (half_log - 1))) < (uint64_t)n && half_log < 21) half_log++;
        bits = half_log + 4 > 21 ? 21 : half_log + 4;
    }
    // Above the merge-sort range every radix pass is a trip of keys and values through HBM.  When the code and the index
    // fit one 64-bit word together (code << idx_bits | index) with no fewer bits per axis than half of log2(n),
    // the sort moves keys only and over the code's bits only: 5 passes of 8 bytes instead of 7 of 12 at 10 M.
    uint32_t idx_bits = 0;
    if (direct || (n > (1u << 20) && !getenv("CRT_BVH_SORT_PAIRS"))) {
        uint32_t ib = 1;
        while ((1ull << ib) < (uint64_t)n) ib++;
        const uint32_t fit = std::min(21u, (64u - ib) / 3u);
        if (direct || fit >= half_log) { idx_bits = ib; bits = std::min(bits, fit); }      // direct: n < 2^24 leaves 13 bits per axis at least
    }
    const unsigned key_bits = 3 * bits;
    // rocPRIM 4.2's radix_sort_keys(double_buffer) with begin_bit > 0 leaves 64-bit keys sorted inside 1024-key blocks only
    // wherever it merge-sorts (1025 .. 2^20 keys; tools/src/rocprim_sort_check.hip, profiles/r03_rocprim_sort_check.txt): up
    // to there the packed keys are sorted over ALL their bits, which gives the same order (the index in the low bits is what a
    // stable sort by code would have kept) at the same cost (a merge sort does not care about the key width).  Found by the
    // random soups of tests/test_gpu_fuzz.py: a direct build of a few thousand primitives got a valid but badly cut tree
    // (any partition of the primitives makes a correct BVH, which is why no parity test noticed), and now and then a cut
    // search that never ended.
#ifdef CRT_REPRO_ROCPRIM_BIT_RANGE      // (to reproduce the finding: the trees stay valid, the build reports unsorted keys when a search notices)
    const unsigned sort_lo = idx_bits;
#else
    const unsigned sort_lo = (idx_bits && n <= (1u << 20)) ? 0u : idx_bits;
#endif
    rocprim::double_buffer<uint64_t> dkeys((uint64_t *)st.keys.ptr, (uint64_t *)st.keys_alt.ptr);
    rocprim::double_buffer<uint32_t> dvals((uint32_t *)st.vals.ptr, (uint32_t *)st.vals_alt.ptr);
    if (direct && n >= (1u << 24)) { err = "wide BVH supports fewer than 2^24 primitives"; return false; }
    if (idx_bits) HIP_TRY(rocprim::radix_sort_keys(nullptr, sort_bytes, dkeys, n, sort_lo, idx_bits + key_bits, stream));
    else HIP_TRY(rocprim::radix_sort_pairs<PairSortConfig>(nullptr, sort_bytes, dkeys, dvals, n, 0, key_bits, stream));
    HIP_TRY(rocprim::exclusive_scan(nullptr, scan_bytes, (uint32_t *)st.flag_a.ptr, (uint32_t *)st.rank_a.ptr, 0u, n,
                                    rocprim::plus<uint32_t>(), stream));
    // CRT_BVH_SORT=onesweep: the keys-only sorts of more than a million keys (the per-frame rebuilds) go through the
    // hand-written sort of bvh_sort.hip instead of rocPRIM's.  Measured at 10 M keys, 39 key bits (profiles/r03_build_stats_10m_direct.md):
    // 83.7 us per digit pass + 108 us for the histograms against rocPRIM's 63.6 us per pass + 53 us of offsets + 56 us of
    // fills, i.e. 2.64 instead of 2.49 ms per config-5 frame: rocPRIM stays the default ("if it cannot
    // beat rocPRIM, keep rocPRIM and write the measured reason").  The sort is correct (tests/test_gpu_sort.py) and stays
    // in the tree as the starting point: its ranking step serialises sixteen LDS round trips per wave and tile.
    static const std::string env_sort = [] { const char *e = getenv("CRT_BVH_SORT"); return std::string(e ? e : ""); }();
    // The (code, primitive) PAIR sorts of static builds of up to a million primitives (where rocPRIM falls back to a merge
    // sort: 17 launches, 137 us for 225 k pairs) can go through it too, with 1024-key tiles: 143 us (eight digit passes of
    // 14.4 us, whose look-back chains and ranking steps are latency, plus 19 us for a histogram whose top digits all hit the
    // same LDS words), so there too the library stays the default.
    const bool own_sort = env_sort == "onesweep" && ((idx_bits != 0 && n > (1u << 20)) || (idx_bits == 0 && !direct && n <= (1u << 20) && n >= 2));
    st.sort_error_dev = nullptr;
    if (!ensure(st.temp, std::max(std::max(sort_bytes, scan_bytes), own_sort ? radix_sort_scratch_bytes(n) : (size_t)0), err)) return false;

    // the animation pass already left records, centroids and centroid bounds of exactly these triangles (animate_tris)
    const bool have_records = direct && st.records_fresh && in.n_tris == n;
    if (!have_records) hipLaunchKernelGGL(k_init_bounds, dim3(1), dim3(64), 0, stream, (uint32_t *)st.scene_bounds.ptr);
    if (!have_records) {
        const uint32_t n_other = in.n_spheres + in.n_quads;
        float4 *plo = (float4 *)st.plo.ptr, *phi = direct ? nullptr : (float4 *)st.phi.ptr;
        uint32_t *sb = (uint32_t *)st.scene_bounds.ptr;
        if (n_other) {
            const dim3 g(std::min((n_other + 255u) / 256u, 4096u));
            if (direct) hipLaunchKernelGGL(k_prim_bounds<false>, g, dim3(256), 0, stream, in, n_other, plo, phi, sb);
            else hipLaunchKernelGGL(k_prim_bounds<true>, g, dim3(256), 0, stream, in, n_other, plo, phi, sb);
        }
        if (in.n_tris) {
            const dim3 g(std::min((in.n_tris + 255u) / 256u, 4096u));
            if (direct) hipLaunchKernelGGL(k_tri_bounds<false>, g, dim3(256), 0, stream, in.tris, in.n_tris, n_other, plo, phi, (float4 *)st.tri_shade.ptr, sb);
            else hipLaunchKernelGGL(k_tri_bounds<true>, g, dim3(256), 0, stream, in.tris, in.n_tris, n_other, plo, phi, (float4 *)st.tri_shade.ptr, sb);
        }
    }
    // a direct build of a triangle scene leaves plo / tri_shade / scene_bounds in the form animate_tris writes; any other
    // build keeps boxes in plo, which a later direct build must not mistake for centroids
    st.records_fresh = direct && in.n_tris == n;
    hipLaunchKernelGGL(k_morton, dim3(blocks), dim3(256), 0, stream, n, bits, idx_bits, (const float4 *)st.plo.ptr,
                       direct ? (const float4 *)nullptr : (const float4 *)st.phi.ptr, (const uint32_t *)st.scene_bounds.ptr, (uint64_t *)st.keys.ptr, (uint32_t *)st.vals.ptr);
    HIP_TRY(hipGetLastError());
    build_trace(stream, "bounds + morton", n);
    size_t tb = st.temp.bytes;
    const uint64_t *keys = nullptr; const uint32_t *sort_error = nullptr;
    const uint32_t *own_vals = nullptr;
    if (own_sort) {
        if (!radix_sort_u64(stream, (uint64_t *)st.keys.ptr, (uint64_t *)st.keys_alt.ptr, idx_bits ? nullptr : (uint32_t *)st.vals.ptr, idx_bits ? nullptr : (uint32_t *)st.vals_alt.ptr,
                            n, idx_bits, idx_bits + key_bits, st.temp.ptr, &keys, &own_vals, &sort_error, err)) return false;
        if (!direct) st.sort_error_dev = sort_error;
    } else {
        if (idx_bits) HIP_TRY(rocprim::radix_sort_keys(st.temp.ptr, tb, dkeys, n, sort_lo, idx_bits + key_bits, stream));
        else HIP_TRY(rocprim::radix_sort_pairs<PairSortConfig>(st.temp.ptr, tb, dkeys, dvals, n, 0, key_bits, stream));
        keys = dkeys.current();
    }
    const uint32_t *vals = (own_sort && !idx_bits) ? own_vals : dvals.current();
    build_trace(stream, "sort", n);

    if (direct) {
        const uint32_t idx_mask = (uint32_t)((1ull << idx_bits) - 1ull);
        if (mixed) {
            hipLaunchKernelGGL(k_kind_flags_keys, dim3(blocks), dim3(256), 0, stream, n, keys, idx_mask, in.n_spheres, in.n_quads,
                               (uint32_t *)st.flag_a.ptr, (uint32_t *)st.flag_b.ptr);
            tb = st.temp.bytes;
            HIP_TRY(rocprim::exclusive_scan(st.temp.ptr, tb, (uint32_t *)st.flag_a.ptr, (uint32_t *)st.rank_a.ptr, 0u, n, rocprim::plus<uint32_t>(), stream));
            tb = st.temp.bytes;
            HIP_TRY(rocprim::exclusive_scan(st.temp.ptr, tb, (uint32_t *)st.flag_b.ptr, (uint32_t *)st.rank_b.ptr, 0u, n, rocprim::plus<uint32_t>(), stream));
        }
        PackOut pout{(float4 *)st.tri_pack.ptr, (float4 *)st.sphere_pack.ptr, (float4 *)st.quad_pack.ptr, (uint32_t *)st.sphere_ids.ptr,
                     (uint32_t *)st.quad_ids.ptr, (uint32_t *)st.tri_ids.ptr};
        build_trace(stream, "kind ranks", mixed ? 1u : 0u);
        st.sorted_vals = nullptr;
        return build_wide_direct(stream, in, st, keys, idx_bits, mixed ? (const uint32_t *)st.rank_a.ptr : nullptr,
                                 mixed ? (const uint32_t *)st.rank_b.ptr : nullptr, pout, sort_error, err);
    }

    // SAH refinement: the Karras hierarchy only clusters; bvh_sah.hip rebuilds the topology and reorders the primitives
    const bool sah = (flags & CRT_BVH_SAH_REFINE) != 0 && n >= 2;
    st.fuse_dp = sah && wide && !wide_collapse_greedy() && !getenv("CRT_BVH_DP_LEVELS");      // CRT_BVH_DP_LEVELS=1: the table by separate level launches (A/B)
    const uint32_t *order = vals;
    if (n >= 2) {
        if (!ensure(st.range_lo, (size_t)n * 4, err) || !ensure(st.range_hi, (size_t)n * 4, err)) return false;
        hipLaunchKernelGGL(k_hierarchy, dim3((n - 1 + 255) / 256), dim3(256), 0, stream, (int)n, keys, (uint32_t *)st.child_l.ptr,
                           (uint32_t *)st.child_r.ptr, (uint32_t *)st.parent.ptr, (uint32_t *)st.range_lo.ptr,
                           (uint32_t *)st.range_hi.ptr, idx_bits ? (uint32_t *)vals : nullptr, (uint32_t)((1ull << idx_bits) - 1));
        HIP_TRY(hipGetLastError());
    }
    if (sah) {
        if (!build_sah(stream, in, st, vals, err)) return false;
        order = (const uint32_t *)st.perm.ptr;
    }
    st.sorted_vals = order;

    // slot of a sorted position inside its kind's array = number of earlier positions of that kind: two scans, skipped
    // (with the flag kernel) when the scene holds a single kind, as triangle meshes do
    if (mixed) {
        hipLaunchKernelGGL(k_kind_flags, dim3(blocks), dim3(256), 0, stream, n, order, in.n_spheres, in.n_quads,
                           (uint32_t *)st.flag_a.ptr, (uint32_t *)st.flag_b.ptr);
        tb = st.temp.bytes;
        HIP_TRY(rocprim::exclusive_scan(st.temp.ptr, tb, (uint32_t *)st.flag_a.ptr, (uint32_t *)st.rank_a.ptr, 0u, n,
                                        rocprim::plus<uint32_t>(), stream));
        tb = st.temp.bytes;
        HIP_TRY(rocprim::exclusive_scan(st.temp.ptr, tb, (uint32_t *)st.flag_b.ptr, (uint32_t *)st.rank_b.ptr, 0u, n,
                                        rocprim::plus<uint32_t>(), stream));
    }
    PackOut pout{(float4 *)st.tri_pack.ptr, (float4 *)st.sphere_pack.ptr, (float4 *)st.quad_pack.ptr, (uint32_t *)st.sphere_ids.ptr,
                 (uint32_t *)st.quad_ids.ptr, (uint32_t *)st.tri_ids.ptr};
    const uint32_t *rank_sphere = mixed ? (const uint32_t *)st.rank_a.ptr : nullptr, *rank_quad = mixed ? (const uint32_t *)st.rank_b.ptr : nullptr;
    if (sah || n == 1) {      // the LBVH refit packs the records itself (k_refit_leaves)
        hipLaunchKernelGGL(k_pack, dim3(blocks), dim3(256), 0, stream, in, n, order, rank_sphere, rank_quad, (const float4 *)st.tri_shade.ptr, pout,
                           (uint32_t *)st.leaf_link.ptr, (uint32_t *)st.leaf_count.ptr);
        HIP_TRY(hipGetLastError());
    }

    if (n == 1) {
        hipLaunchKernelGGL(k_single_pair, dim3(1), dim3(64), 0, stream, order, (const float4 *)st.plo.ptr, (const float4 *)st.phi.ptr,
                           (const uint32_t *)st.leaf_link.ptr, (const uint32_t *)st.leaf_count.ptr, (float4 *)st.pairs.ptr);
        HIP_TRY(hipGetLastError());
        return wide ? build_wide(stream, st, err) : true;
    }
    if (sah) {
        if (!patch_sah_links(stream, st, err)) return false;
        return wide ? build_wide(stream, st, err) : true;
    }
    HIP_TRY(hipMemsetAsync(st.flag_a.ptr, 0, (size_t)n * 4, stream));
    RefitArgs ra;
    ra.n = (int)n; ra.vals = vals; ra.in = in; ra.rank_sphere = rank_sphere; ra.rank_quad = rank_quad; ra.tri_shade = (const float4 *)st.tri_shade.ptr; ra.out = pout;
    ra.child_l = (const uint32_t *)st.child_l.ptr; ra.child_r = (const uint32_t *)st.child_r.ptr; ra.parent = (const uint32_t *)st.parent.ptr;
    ra.range_lo = (const uint32_t *)st.range_lo.ptr; ra.range_hi = (const uint32_t *)st.range_hi.ptr;
    ra.nlo = (float4 *)st.nlo.ptr; ra.nhi = (float4 *)st.nhi.ptr; ra.ninfo = (float4 *)st.ninfo.ptr;
    ra.flags = (uint32_t *)st.flag_a.ptr; ra.pairs = (float4 *)st.pairs.ptr;
    ra.max_leaf = (flags & CRT_BVH_LEAF_COLLAPSE) ? (st.max_leaf < 1u ? 1u : st.max_leaf > 8u ? 8u : st.max_leaf) : 1u;
    ra.c_isect = st.c_isect; ra.c_trav = st.c_trav;
    // continuation lists of two consecutive stages (ping-pong); stage 1 has the most workgroups
    uint32_t n_wg = (n + kRefitLeaves - 1) / kRefitLeaves;
    const uint32_t n_wg2 = (n_wg + kRefitFan - 1) / kRefitFan;
    if (!ensure(st.refit_list[0], (size_t)n_wg * kRefitCap * 4, err) || !ensure(st.refit_count[0], (size_t)n_wg * 4, err)) return false;
    if (!ensure(st.refit_list[1], (size_t)n_wg2 * kRefitCap * 4, err) || !ensure(st.refit_count[1], (size_t)n_wg2 * 4, err)) return false;
    hipLaunchKernelGGL(k_refit_leaves, dim3(n_wg), dim3(kRefitLeaves), 0, stream, ra, (uint32_t *)st.refit_list[0].ptr, (uint32_t *)st.refit_count[0].ptr);
    uint64_t range = kRefitLeaves;
    for (int cur = 0; n_wg > 1; cur ^= 1) {
        const uint32_t n_prev = n_wg;
        n_wg = (n_wg + kRefitFan - 1) / kRefitFan;
        range *= kRefitFan;
        // the last stage's range covers every leaf, so nothing is parked there
        hipLaunchKernelGGL(k_refit_resume, dim3(n_wg), dim3(1024), 0, stream, ra, n_prev, (const uint32_t *)st.refit_list[cur].ptr,
                           (const uint32_t *)st.refit_count[cur].ptr, (uint32_t)std::min<uint64_t>(range, 0x80000000ull),
                           (uint32_t *)st.refit_list[cur ^ 1].ptr, (uint32_t *)st.refit_count[cur ^ 1].ptr);
    }
    HIP_TRY(hipGetLastError());
    return wide ? build_wide(stream, st, err) : true;
}

// The triangles' shading records (and, as a by-product, their centroids) from the triangles as they are now: what a build
// does first and what crt_bvh_refit needs again after the triangles moved.
void launch_tri_records(hipStream_t stream, const BuildInput &in, BuildState &st) {
    if (!in.n_tris) return;
    if (st.records_fresh && in.n_tris == st.n_prims) return;      // the animation pass wrote them (animate_tris)
    hipLaunchKernelGGL(k_init_bounds, dim3(1), dim3(64), 0, stream, (uint32_t *)st.scene_bounds.ptr);
    const dim3 g(std::min((in.n_tris + 255u) / 256u, 4096u));
    hipLaunchKernelGGL(k_tri_bounds<false>, g, dim3(256), 0, stream, in.tris, in.n_tris, in.n_spheres + in.n_quads, (float4 *)st.plo.ptr, (float4 *)nullptr,
                       (float4 *)st.tri_shade.ptr, (uint32_t *)st.scene_bounds.ptr);
}

void free_build_state(BuildState &st) {
    DeviceBuffer *all[] = {&st.plo, &st.phi, &st.keys, &st.vals, &st.keys_alt, &st.vals_alt, &st.flag_a, &st.flag_b, &st.rank_a,
                           &st.rank_b, &st.leaf_link, &st.leaf_count, &st.child_l, &st.child_r, &st.parent, &st.nlo, &st.nhi, &st.ninfo,
                           &st.scene_bounds, &st.temp, &st.range_lo, &st.range_hi, &st.sah_flag, &st.sah_ipos[0],
                           &st.sah_ipos[1], &st.sah_starts, &st.sah_desc, &st.perm, &st.pairs, &st.tri_pack, &st.sphere_pack, &st.quad_pack, &st.sphere_ids,
                           &st.quad_ids, &st.tri_ids, &st.tri_shade, &st.wide, &st.wide_queue[0], &st.wide_queue[1], &st.wide_levels_dev,
                           &st.dp_cost, &st.dp_choice, &st.dp_parent, &st.dp_depth, &st.dp_list, &st.dp_hist,
                           &st.refit_list[0], &st.refit_list[1], &st.refit_count[0], &st.refit_count[1]};
    for (DeviceBuffer *b : all) if (b->ptr) { hipError_t e = hipFree(b->ptr); (void)e; b->ptr = nullptr; b->bytes = 0; }
    if (st.host_lc) { hipError_t e = hipHostFree(st.host_lc); (void)e; st.host_lc = nullptr; }
    if (st.ev_counts) { hipError_t e = hipEventDestroy(st.ev_counts); (void)e; st.ev_counts = nullptr; }
    st.counts_pending = false; st.direct_hint_levels = 0; st.direct_hint_n = 0;
}

}  // namespace crt
