// bvh_wide.hip -- collapse of the pair BVH into 8-wide nodes with 8-bit quantised child boxes (CRT_BVH_WIDE).
//
// Why: the wavefront tracer's extend kernel is bound by the vector L1's address path, which retires about one
// 16-byte lane access to a distinct line per clock and CU (tools/ubench_nodefetch.hip), i.e. by the NUMBER of
// divergent 16-byte loads per ray, not by bytes.  A pair visit costs four such loads for two boxes; a wide visit
// costs four for eight.  The tree also shrinks from 64 B per primitive to ~25 B, most of it resident in the
// 4 MiB L2 of each XCD.
//
// Node (64 bytes on the device = what a traversal step loads, two siblings per 128-byte cache line; the links are a separate
// array of 32 bytes per node behind the nodes; crt_bvh_wide_read hands both out as one 128-byte record per node, laid out
// as listed here):
//   float4 0   {ox|ex, oy|ey, oz|ez, child_base<<8 | imask}
//              o* = node origin as an fp32 with the low 9 mantissa bits cleared (rounded DOWN), e* = biased exponent
//              of the per-axis grid step (2^(e-127)) in those low 8 bits; child_base = index of the first inner child
//              (inner children are consecutive, in slot order); imask bit s = slot s is an inner node
//   float4 1   {qlo_x[0..3], qlo_x[4..7], qlo_y[0..3], qlo_y[4..7]}      one byte per slot
//   float4 2   {qlo_z[0..3], qlo_z[4..7], qhi_x[0..3], qhi_x[4..7]}
//   float4 3   {qhi_y[0..3], qhi_y[4..7], qhi_z[0..3], qhi_z[4..7]}
//   float4 4,5 link[8]: leaf slot = kLeafTag | kind << 29 | (count - 1) << 26 | first slot in the kind's array
//              (loaded lazily, only when a leaf slot is actually hit); inner slot = its node index (diagnostic);
//              empty slot = kLeafTag | 3 << 29
//   float4 6,7 unused (read-back record only)
// Child box of slot s on axis a = [o_a + qlo * 2^e_a, o_a + qhi * 2^e_a] as REAL numbers; it contains the child's box of
// the pair BVH by construction (wide_emit rounds outwards with 2^-10 of a grid step of slack against its own single
// rounding; tests/test_gpu_parity.py::_check_wide verifies the containment exactly).  Empty slots have qlo = 255,
// qhi = 0.
//
// Slots are assigned so that slot ^ octant(ray) is a front-to-back visiting order (Ylitie, Karras, Laine 2017):
// a child whose centroid lies towards +x/+y/+z of the node centre prefers a slot with bit 0/1/2 set.
//
// Collapse: which inner nodes of the pair BVH become wide nodes is decided by dynamic programming over the whole tree
// (Ylitie, Karras, Laine 2017, section 4.1), minimising the summed surface area of the wide nodes = the expected
// number of node visits per ray: c(n, i) = cheapest way to hand the subtree of inner node n to its parent as at most i
// children.  c(n, 1) = area(n) + min_k [c(left, k) + c(right, 8 - k)] (n becomes a wide node), c(n, i) = min(c(n, i - 1),
// min_k [c(left, k) + c(right, i - k)]) (n is dissolved into its parent); a leaf costs nothing extra and takes one slot.
// Bottom-up: the pairs are bucketed by depth and one launch handles one depth, deepest first.  The greedy rule this
// replaces (keep expanding the child of largest area) left 31 % of the nodes of the 224,998-triangle scene with two
// children; see DESIGN.md for the measured difference.  Then top-down, one thread per wide node, level by level: a
// thread starts from the two children of a pair with the budgets the table prescribes and keeps dissolving inner
// children whose budget is larger than one.  Wide nodes are numbered in BFS order: index = (nodes of earlier levels) +
// position in the level's queue, so the inner children of one node are consecutive.  CRT_WIDE_GREEDY=1 (environment)
// selects the old rule for A/B runs.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <string>

#include "build_internal.h"
#include "bvh_common.h"
#include "bvh_dp.h"
#include "bvh_wide_node.h"

namespace crt {

struct WideChild { float lo[3], hi[3]; uint32_t link, count, budget, eff; };      // eff: the child's eff table (table-driven collapse)


__device__ __forceinline__ bool pad_leaf(uint32_t count) { return count != 0u && (count & 0x0FFFFFFFu) == 0u; }

__device__ __forceinline__ void read_half(const float4 *half, WideChild &c) {
    float4 a = half[0], b = half[1];
    c.lo[0] = a.x; c.lo[1] = a.y; c.lo[2] = a.z; c.link = __float_as_uint(a.w);
    c.hi[0] = b.x; c.hi[1] = b.y; c.hi[2] = b.z; c.count = __float_as_uint(b.w);
    c.budget = 1u; c.eff = 0u;
}

// ---- the table ----------------------------------------------------------------------------------------------------
// choice word of inner node n (64 bits): bits 3 (j - 2) .. +2 = k of the best split of j slots between its children
// (j = 2 .. 8: left gets k, right j - k); bits 21 .. 41 = the eff table of its LEFT child, bits 42 .. 62 that of its right
// child, where eff(i), i = 1 .. 7, three bits each, is the number of slots a node actually uses when it is offered i
// (1 = it stays a wide node of its own; 0 for a leaf).  A node's own table sits in its parent's word, so that the
// collapse knows whether a child dissolves from the word it loaded together with the child's pair: one dependent
// round trip per dissolved child instead of two.  (cost row of a node: c(1..7) and, as the eighth word, its own table,
// from where the parent's entry takes it.)
constexpr uint32_t kDpLevels = kPairLevels;      // pair trees are at most 96 deep (kMaxPairDepth), LBVH ones at most 95
struct DpLevels { uint32_t start[kDpLevels + 1]; };

// Buckets the pairs by depth: list[start[d] ..] = the pairs of depth d (any order).  Ranks inside a workgroup come from LDS
// atomics, one global atomic per workgroup and depth reserves the workgroup's run.
__global__ __launch_bounds__(256) void k_dp_scatter(uint32_t n_pairs, const uint32_t *depth, DpLevels lv, uint32_t *cursor, uint32_t *list) {
    __shared__ uint32_t s_cnt[kDpLevels], s_base[kDpLevels];
    for (uint32_t i = threadIdx.x; i < kDpLevels; i += blockDim.x) s_cnt[i] = 0u;
    __syncthreads();
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t d = p < n_pairs ? depth[p] : 0u;
    uint32_t r = 0;
    if (d) r = atomicAdd(&s_cnt[d], 1u);
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < kDpLevels; i += blockDim.x) if (s_cnt[i]) s_base[i] = atomicAdd(&cursor[i], s_cnt[i]);
    __syncthreads();
    if (d) list[lv.start[d] + s_base[d] + r] = p;
}

__global__ void k_dp_level(const float4 *pairs, const uint32_t *list, uint32_t first, uint32_t count, float *cost, unsigned long long *choice) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const uint32_t n = list[first + i];
    WideChild a, b;
    read_half(pairs + 4 * (size_t)n, a);
    read_half(pairs + 4 * (size_t)n + 2, b);
    float cl[8], cr[8];                                   // index 1 .. 7
#pragma unroll
    for (int k = 1; k < 8; k++) { cl[k] = a.count == 0u ? cost[8 * (size_t)a.link + (k - 1)] : 0.0f; cr[k] = b.count == 0u ? cost[8 * (size_t)b.link + (k - 1)] : 0.0f; }
    const uint32_t eff_l = a.count == 0u ? __float_as_uint(cost[8 * (size_t)a.link + 7]) : 0u;
    const uint32_t eff_r = b.count == 0u ? __float_as_uint(cost[8 * (size_t)b.link + 7]) : 0u;
    float c[8]; uint32_t own; unsigned long long w;
    dp_combine(a.lo, a.hi, b.lo, b.hi, cl, cr, eff_l, eff_r, c, own, w);
#pragma unroll
    for (int k = 1; k < 8; k++) cost[8 * (size_t)n + (k - 1)] = c[k];
    cost[8 * (size_t)n + 7] = __uint_as_float(own);
    choice[n] = w;
}

// ---- one wide node, eight lanes ---------------------------------------------------------------------------------------
// A node is made by a group of EIGHT lanes, one per child.  (The first version gave a node to one thread, with its eight
// children in registers behind select chains: ~5000 dependent instructions and up to seven dependent loads per node, and
// since a level of the tree cannot start before the level above it has finished, every one of the ~12 levels cost that
// chain: 0.21 of the 0.83 ms of a 225 k-triangle build.)  Now every child that the table dissolves loads its two halves in
// the same round (lane l keeps the left one, a free lane takes the right one: at most three rounds where the old loop made
// six trips), the greedy slot assignment is eight rounds of eight evaluations per lane plus a three-step reduction, and
// the quantisation is bvh_wide_node.h's (one slot per lane).
__device__ __forceinline__ uint32_t group_mask8(bool p) { return (uint32_t)((__ballot(p) >> (threadIdx.x & 56u)) & 0xFFull); }
__device__ __forceinline__ uint32_t nth_bit8(uint32_t mask, uint32_t n) {      // position of the n-th set bit (8: there is none)
    uint32_t r = 8u;
#pragma unroll
    for (int i = 7; i >= 0; i--) { const uint32_t below = (uint32_t)__builtin_popcount(mask & ((1u << i) - 1u)); r = (((mask >> i) & 1u) && below == n) ? (uint32_t)i : r; }
    return r;
}
struct Child8 { float lo[3], hi[3]; uint32_t link, count, budget, eff; bool valid; };
__device__ __forceinline__ void pull8(Child8 &c, uint32_t from, bool take) {      // every lane of the group calls this
    Child8 r;
#pragma unroll
    for (int a = 0; a < 3; a++) { r.lo[a] = __shfl(c.lo[a], (int)from, 8); r.hi[a] = __shfl(c.hi[a], (int)from, 8); }
    r.link = (uint32_t)__shfl((int)c.link, (int)from, 8); r.count = (uint32_t)__shfl((int)c.count, (int)from, 8);
    r.budget = (uint32_t)__shfl((int)c.budget, (int)from, 8); r.eff = (uint32_t)__shfl((int)c.eff, (int)from, 8);
    r.valid = __shfl((int)c.valid, (int)from, 8) != 0;
    if (take) c = r;
}
// valid children to lanes 0 .. n - 1 of the group, in order
__device__ __forceinline__ void compact8(Child8 &c, uint32_t sl) {
    const uint32_t mask = group_mask8(c.valid);
    const uint32_t from = nth_bit8(mask, sl);
    pull8(c, from & 7u, true);
    c.valid = from < 8u;
}

// Collapses the subtree of pair `src` into node `dst` of level `level`; its inner children are appended to q_out (the queue
// of level + 1) and numbered next_base + position (next_base = nodes of levels 0 .. level).  Called by whole waves;
// active, src and dst are uniform over a group of eight lanes, sl = lane & 7.
// n_pairs bounds every pair index and every node number: a link that points outside the pair array, or more nodes than
// pairs, can only come from a corrupt tree; such children are dropped and the error word set (the build fails at its next
// read-back) instead of the kernel reading or writing out of bounds.
constexpr uint32_t kWideErrWord = kWideMaxLevels + 4;
__device__ __forceinline__ void wide_emit8(const float4 *pairs, const unsigned long long *choice, uint32_t n_pairs, uint32_t level, uint32_t *level_count, uint32_t next_base,
                                           bool active, uint32_t src, uint32_t dst, uint2 *q_out, float4 *wide, uint32_t *links) {
    const uint32_t sl = threadIdx.x & 7u;
    Child8 c;
#pragma unroll
    for (int a = 0; a < 3; a++) { c.lo[a] = 0.0f; c.hi[a] = 0.0f; }
    c.link = 0u; c.count = 0u; c.budget = 1u; c.eff = 0u; c.valid = false;
    if (active && sl < 2u) {
        WideChild h;
        read_half(pairs + 4 * (size_t)src + 2 * sl, h);
#pragma unroll
        for (int a = 0; a < 3; a++) { c.lo[a] = h.lo[a]; c.hi[a] = h.hi[a]; }
        c.link = h.link; c.count = h.count;
        if (choice) {      // best split of 8 slots, and the children's own tables
            const unsigned long long w = choice[src];
            const uint32_t k8 = (uint32_t)(w >> 18) & 7u;
            c.budget = sl == 0u ? k8 : 8u - k8;
            c.eff = (uint32_t)(w >> (sl == 0u ? 21 : 42)) & kDpEffMask;
        }
        c.valid = !pad_leaf(h.count);
    }
    if (__ballot(active && sl < 2u && !c.valid) != 0ull) compact8(c, sl);
    // ---- dissolve: with the table, every inner child that the table dissolves when offered its budget (eff > 1), all of
    // them in the same round; without it (the per-frame LBVH rebuild), the inner child of largest area, one per round,
    // until there are eight
    for (;;) {
        const uint32_t n = (uint32_t)__builtin_popcount(group_mask8(c.valid));
        bool dis = false; uint32_t grant = 0u;
        if (choice) {
            if (c.valid && c.count == 0u && c.budget >= 2u && c.link < n_pairs) {
                const uint32_t bud = c.budget > 7u ? 7u : c.budget;
                const uint32_t eff = (c.eff >> (3u * (bud - 1u))) & 7u;
                if (eff > 1u) { dis = true; grant = eff; } else c.budget = 1u;      // stays a wide node of its own
            }
        } else {
            const float area = (c.valid && c.count == 0u && c.link < n_pairs) ? half_area3(c.lo, c.hi) : -1.0f;
            float best = area > -1.0f ? area : -1.0f; uint32_t who = area > -1.0f ? sl : 8u;      // (a NaN area is never chosen)
#pragma unroll
            for (int off = 1; off < 8; off <<= 1) {
                const float ob = __shfl_xor(best, off, 8); const uint32_t ow = (uint32_t)__shfl_xor((int)who, off, 8);
                const bool take = ob > best || (ob == best && ow < who);
                best = take ? ob : best; who = take ? ow : who;
            }
            dis = n < 8u && who == sl;
        }
        if (__ballot(dis) == 0ull) break;
        const uint32_t dmask = group_mask8(dis), nd = (uint32_t)__builtin_popcount(dmask);
        // lane l (dissolving) takes the left half of its pair; lane n + r takes the right half of the r-th dissolving lane's
        const uint32_t donor = nth_bit8(dmask, sl - n) & 7u;
        const uint32_t link_d = (uint32_t)__shfl((int)c.link, (int)donor, 8), grant_d = (uint32_t)__shfl((int)grant, (int)donor, 8);
        const bool taker = active && sl >= n && sl < n + nd;
        if (dis || taker) {
            const uint32_t lk = dis ? c.link : link_d, gr = dis ? grant : grant_d, side = dis ? 0u : 1u;
            WideChild h;
            read_half(pairs + 4 * (size_t)lk + 2 * side, h);
#pragma unroll
            for (int a = 0; a < 3; a++) { c.lo[a] = h.lo[a]; c.hi[a] = h.hi[a]; }
            c.link = h.link; c.count = h.count; c.budget = 1u; c.eff = 0u;
            if (choice) {      // loaded together with the pair: the split of the slots it was granted and its children's tables
                const unsigned long long w = choice[lk];
                const uint32_t kk = (uint32_t)(w >> (3u * (gr - 2u))) & 7u;
                c.budget = side ? gr - kk : kk;
                c.eff = (uint32_t)(w >> (side ? 42 : 21)) & kDpEffMask;
            }
            c.valid = !pad_leaf(h.count);
        }
        if (__ballot((dis || taker) && !c.valid) != 0ull) compact8(c, sl);
    }
    // ---- node box and the children's offsets from its centre
    float lo[3], hi[3], dv[3];
#pragma unroll
    for (int a = 0; a < 3; a++) {
        float l = c.valid ? c.lo[a] : 3.402823466e38f, h = c.valid ? c.hi[a] : -3.402823466e38f;
#pragma unroll
        for (int off = 1; off < 8; off <<= 1) { l = fminf(l, __shfl_xor(l, off, 8)); h = fmaxf(h, __shfl_xor(h, off, 8)); }
        lo[a] = l; hi[a] = h;
        dv[a] = (0.5f * c.lo[a] + 0.5f * c.hi[a]) - (0.5f * l + 0.5f * h);
    }
    // ---- slot assignment: greedy maximum of sum_a (slot bit a ? +1 : -1) * (child centroid - node centre)_a over all
    // (child, free slot) pairs; the per-frame LBVH rebuild (no table) serves the children in order of decreasing |d|_1
    // instead, each taking its best free slot (0.14 % more node visits on the headline scene, an eighth of the work when
    // this was one thread's loop; kept so that those trees stay what they were).  Ties: lowest lane, lowest slot.
    const float l1 = (fabsf(dv[0]) + fabsf(dv[1])) + fabsf(dv[2]);
    uint32_t used = 0u, my_slot = 8u;
    for (int it = 0; it < 8; it++) {
        const bool cand = c.valid && my_slot == 8u;
        float bv = -3.402823466e38f; uint32_t bs = (uint32_t)__builtin_ctz((~used & 0xFFu) | 0x100u);      // only NaN scores: the first free slot
#pragma unroll
        for (int s2 = 0; s2 < 8; s2++) {
            const float v = ((s2 & 1) ? dv[0] : -dv[0]) + ((s2 & 2) ? dv[1] : -dv[1]) + ((s2 & 4) ? dv[2] : -dv[2]);
            if (!((used >> s2) & 1u) && v > bv) { bv = v; bs = (uint32_t)s2; }
        }
        float key = choice ? bv : l1;
        key = (cand && key == key) ? key : -3.402823466e38f;
        float bk = key; uint32_t who = cand ? sl : 8u;
#pragma unroll
        for (int off = 1; off < 8; off <<= 1) {
            const float ob = __shfl_xor(bk, off, 8); const uint32_t ow = (uint32_t)__shfl_xor((int)who, off, 8);
            const bool take = ow < 8u && (who == 8u || ob > bk || (ob == bk && ow < who));
            bk = take ? ob : bk; who = take ? ow : who;
        }
        if (__ballot(who < 8u) == 0ull) break;
        const uint32_t ws = (uint32_t)__shfl((int)bs, (int)(who & 7u), 8);
        if (who < 8u) { used |= 1u << ws; if (who == sl) my_slot = ws; }
    }
    // ---- to slot order: lane s takes the child that got slot s
    uint32_t from = 8u;
#pragma unroll
    for (int j = 0; j < 8; j++) { const uint32_t sj = (uint32_t)__shfl((int)my_slot, j, 8); from = sj == sl ? (uint32_t)j : from; }
    pull8(c, from & 7u, true);
    c.valid = c.valid && from < 8u;
    if (c.valid && c.count == 0u && c.link >= n_pairs) { c.valid = false; atomicOr(&level_count[kWideErrWord], 1u); }
    uint32_t imask = group_mask8(c.valid && c.count == 0u);
    uint32_t n_inner = (uint32_t)__builtin_popcount(imask);
    uint32_t qpos = 0u;
    if (active && sl == 0u && n_inner) qpos = atomicAdd(&level_count[level + 1], n_inner);
    qpos = (uint32_t)__shfl((int)qpos, 0, 8);
    if (n_inner && (uint64_t)next_base + qpos + n_inner > (uint64_t)n_pairs) {      // more nodes than pairs: not a tree
        if (active && sl == 0u) atomicOr(&level_count[kWideErrWord], 2u);
        if (c.count == 0u) c.valid = false;
        imask = 0u; n_inner = 0u;
    }
    const uint32_t child_base = next_base + qpos;
    uint32_t lk = kWideEmptyLink;
    if (c.valid) {
        if (c.count == 0u) {      // inner children are numbered in slot order
            const uint32_t r = (uint32_t)__builtin_popcount(imask & ((1u << sl) - 1u));
            lk = child_base + r;
            if (active) q_out[qpos + r] = make_uint2(c.link, lk);
        } else {
            const uint32_t kind = c.count >> 28, cnt = c.count & 0x0FFFFFFFu;
            lk = kWideLeafTag | (kind << 29) | ((cnt - 1u) << 26) | c.link;
        }
    }
    if (active) links[8 * (size_t)dst + sl] = lk;
    if (!c.valid) {      // empty slots take no part in the node's box (wide_node_finish reduces over all eight lanes)
#pragma unroll
        for (int a = 0; a < 3; a++) { c.lo[a] = 3.402823466e38f; c.hi[a] = -3.402823466e38f; }
    }
    wide_node_finish(active, sl, dst, (child_base << 8) | imask, c.lo, c.hi, c.valid, wide, nullptr, nullptr);
}

// level_count[l] = wide nodes of level l (queue length); level l reads queue l & 1 and appends to the other one.
// Levels below *first_level were already emitted by k_wide_top.  Eight lanes per node.
constexpr uint32_t kWideLevelThreads = 256;
__global__ __launch_bounds__(kWideLevelThreads) void k_wide_level(const float4 *pairs, const unsigned long long *choice, uint32_t n_pairs, uint32_t level, uint32_t *level_count,
                                                                  const uint32_t *first_level, const uint2 *q_in, uint2 *q_out, float4 *wide, uint32_t *links) {
    if (level < *first_level) return;
    const uint32_t n_in = min(level_count[level], n_pairs);
    uint32_t next_base = 0;                         // nodes of levels 0..level = index of the first node of level + 1
    for (uint32_t l = 0; l <= level; l++) next_base += level_count[l];
    const uint32_t stride = (gridDim.x * blockDim.x) >> 3;
    for (uint32_t i = (blockIdx.x * blockDim.x + threadIdx.x) >> 3;; i += stride) {
        bool active = i < n_in;
        if (__ballot(active) == 0ull) break;
        const uint2 w = active ? q_in[i] : make_uint2(0u, 0u);
        if (w.x >= n_pairs || w.y >= n_pairs) active = false;      // (an entry nobody wrote: only after the error word was set)
        wide_emit8(pairs, choice, n_pairs, level, level_count, next_base, active, w.x, w.y, q_out, wide, links);
    }
}

// The first levels hold 1, <= 8, <= 64 nodes; a launch per level costs a kernel's start-up and drain each however few
// nodes it has.  One workgroup walks the levels that fit its threads, eight threads per node, and leaves the first
// level it did not do in *first_level (k_wide_level skips the earlier ones).  Everything it hands from one level to the
// next (queue entries, level counts) stays inside the workgroup: workgroup-scope release / acquire around the barrier.
constexpr uint32_t kWideTopThreads = 1024;
__global__ __launch_bounds__(kWideTopThreads) void k_wide_top(const float4 *pairs, const unsigned long long *choice, uint32_t n_pairs, uint32_t *level_count, uint32_t max_levels,
                                                              uint32_t *first_level, uint2 *q0, uint2 *q1, float4 *wide, uint32_t *links, const float *root_row) {
    // (the pair tree's height, which the fused table left in word 7 of the root's row, rides to the host with the level table)
    if (threadIdx.x == 0) { level_count[kWideMaxLevels + 3] = root_row ? __float_as_uint(root_row[7]) : 0u; level_count[kWideErrWord] = 0u; }
    __syncthreads();
    uint32_t next_base = 0, level = 0;
    for (; level < max_levels; level++) {
        const uint32_t n_in = __hip_atomic_load(&level_count[level], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (n_in == 0u || n_in > kWideTopThreads / 8u) break;
        next_base += n_in;
        const uint2 *q_in = (level & 1u) ? q1 : q0;
        uint2 *q_out = (level & 1u) ? q0 : q1;
        const uint32_t i = threadIdx.x >> 3;
        bool active = i < n_in;
        if (__ballot(active) != 0ull) {
            const uint2 w = active ? q_in[i] : make_uint2(0u, 0u);
            if (w.x >= n_pairs || w.y >= n_pairs) active = false;
            wide_emit8(pairs, choice, n_pairs, level, level_count, next_base, active, w.x, w.y, q_out, wide, links);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
    if (threadIdx.x == 0) *first_level = level;
}

__global__ void k_wide_init(uint32_t *level_count, uint32_t n_levels, uint2 *q0) {
    for (uint32_t i = threadIdx.x; i <= n_levels; i += blockDim.x) level_count[i] = i == 0 ? 1u : 0u;
    if (threadIdx.x == 0) q0[0] = make_uint2(0u, 0u);       // pair 0 -> wide node 0
}

#define HIP_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { err = std::string(#expr) + ": " + hipGetErrorString(e_); return false; } } while (0)

static bool ensure_w(DeviceBuffer &b, size_t bytes, std::string &err) {
    if (b.bytes >= bytes && b.ptr) return true;
    if (b.ptr) { hipError_t e = hipFree(b.ptr); (void)e; b.ptr = nullptr; b.bytes = 0; }
    size_t want = bytes < 256 ? 256 : bytes;
    HIP_TRY(hipMalloc(&b.ptr, want));
    b.bytes = want;
    return true;
}

bool wide_collapse_greedy() {
    static const bool greedy = [] { const char *e = getenv("CRT_WIDE_GREEDY"); return e && atoi(e) != 0; }();
    return greedy;
}

bool build_wide(hipStream_t stream, BuildState &st, std::string &err) {
    st.n_wide = 0; st.wide_levels = 0;
    if (st.n_pairs == 0) return true;
    if (st.n_pairs >= (1u << 24)) { err = "wide BVH supports fewer than 2^24 nodes"; return false; }
    // every wide node consumes at least one pair, so n_pairs bounds the node count and every level's queue
    if (!ensure_w(st.wide, (size_t)st.n_pairs * (kWideStrideF4 * sizeof(float4) + 32), err)) return false;      // nodes, then 32 bytes of links per node
    st.wide_cap = st.n_pairs;
    uint32_t *links = (uint32_t *)((float4 *)st.wide.ptr + (size_t)kWideStrideF4 * st.wide_cap);
    if (!ensure_w(st.wide_queue[0], (size_t)st.n_pairs * 8, err) || !ensure_w(st.wide_queue[1], (size_t)st.n_pairs * 8, err)) return false;
    if (!ensure_w(st.wide_levels_dev, (kWideMaxLevels + 5) * 4, err)) return false;
    uint32_t *lc = (uint32_t *)st.wide_levels_dev.ptr;
    // ---- the table (only when the pairs' depths were measured, i.e. after the SAH refinement: the per-frame LBVH rebuild
    // keeps the cheaper greedy rule): buckets by depth -> one launch per depth, deepest first ------------------------------
    const bool greedy = wide_collapse_greedy();
    const unsigned long long *choice = nullptr;
    if (!greedy && st.dp_fused && st.n_pairs > 1) choice = (const unsigned long long *)st.dp_choice.ptr;      // made by k_sah_group
    else if (!greedy && st.levels_measured && st.n_pairs > 1) {
        const uint32_t np = st.n_pairs;
        if (!ensure_w(st.dp_cost, (size_t)np * 8 * 4, err) || !ensure_w(st.dp_choice, (size_t)np * 8, err) || !ensure_w(st.dp_list, (size_t)np * 4, err)) return false;
        uint32_t *cursor = (uint32_t *)st.dp_hist.ptr + kDpLevels;      // zeroed by measure_pair_depths
        DpLevels lv; lv.start[0] = 0; lv.start[1] = 0;
        for (uint32_t d = 1; d < kDpLevels; d++) lv.start[d + 1] = lv.start[d] + st.level_hist[d];
        hipLaunchKernelGGL(k_dp_scatter, dim3((np + 255) / 256), dim3(256), 0, stream, np, (const uint32_t *)st.dp_depth.ptr, lv, cursor, (uint32_t *)st.dp_list.ptr);
        for (uint32_t d = st.pair_depth; d >= 1; d--)
            if (st.level_hist[d]) hipLaunchKernelGGL(k_dp_level, dim3((st.level_hist[d] + 127) / 128), dim3(128), 0, stream, (const float4 *)st.pairs.ptr, (const uint32_t *)st.dp_list.ptr,
                               lv.start[d], st.level_hist[d], (float *)st.dp_cost.ptr, (unsigned long long *)st.dp_choice.ptr);
        HIP_TRY(hipGetLastError());
        choice = (const unsigned long long *)st.dp_choice.ptr;
    }
    hipLaunchKernelGGL(k_wide_init, dim3(1), dim3(128), 0, stream, lc, kWideMaxLevels + 1, (uint2 *)st.wide_queue[0].ptr);
    uint32_t *first_level = lc + kWideMaxLevels + 2;
    hipLaunchKernelGGL(k_wide_top, dim3(1), dim3(kWideTopThreads), 0, stream, (const float4 *)st.pairs.ptr, choice, st.n_pairs, lc, kWideMaxLevels, first_level,
                       (uint2 *)st.wide_queue[0].ptr, (uint2 *)st.wide_queue[1].ptr, (float4 *)st.wide.ptr, links,
                       st.dp_fused ? (const float *)st.dp_cost.ptr : (const float *)nullptr);
    const uint32_t grid = std::min<uint32_t>(2048u, (st.n_pairs * 8u + kWideLevelThreads - 1u) / kWideLevelThreads);
    uint32_t host_lc[kWideMaxLevels + 5];
    bool height_read = false;
    uint32_t level = 3;      // k_wide_top always does levels 0, 1 and 2 (1, <= 8, <= 64 nodes; a launch that finds its level done still costs ~4.5 us)
    // launches are planned blind, a chunk of levels at a time; the first chunk is as long as the previous build's tree was
    // deep (a rebuild of the same scene then needs one chunk and no launch that finds nothing to do), else eight levels
    uint32_t chunk = st.wide_levels_hint > level ? st.wide_levels_hint - level : 8u;
    for (;; chunk = 4u) {
        const uint32_t chunk_end = std::min<uint32_t>(level + chunk, kWideMaxLevels);
        for (; level < chunk_end; level++)
            hipLaunchKernelGGL(k_wide_level, dim3(grid), dim3(kWideLevelThreads), 0, stream, (const float4 *)st.pairs.ptr, choice, st.n_pairs, level, lc, (const uint32_t *)first_level,
                               (const uint2 *)st.wide_queue[level & 1].ptr, (uint2 *)st.wide_queue[(level & 1) ^ 1].ptr,
                               (float4 *)st.wide.ptr, links);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(host_lc, lc, (kWideMaxLevels + 5) * 4, hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        if (host_lc[kWideErrWord]) { err = "the pair tree is not a tree (a link outside the pair array, or more wide nodes than pairs)"; return false; }
        if (st.dp_fused && !height_read) {      // the pair tree's height came with the table (bvh_sah.hip): the pair tracers' stacks bound it
            height_read = true;
            st.pair_depth = host_lc[kWideMaxLevels + 3] >> 21;
            if (st.pair_depth > pair_depth_limit()) {
                // Deeper than the pair tracers' stacks (a scene with very many coincident primitives).  The 8-wide tree, whose own
                // depth is checked below, is what the wavefront tracer walks: it is kept, and the pair tree is withheld from
                // the megakernel tracer and crt_bvh_read, as after a direct build or a refit.
                st.pairs_valid = false;
            }
        }
        if (host_lc[level] == 0) break;
        if (level >= kWideMaxLevels) { err = "wide BVH deeper than 64 levels"; return false; }
    }
    for (uint32_t l = 0; l <= kWideMaxLevels && host_lc[l]; l++) { st.n_wide += host_lc[l]; st.wide_levels = l + 1; }
    return true;
}

}  // namespace crt
