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

// The eight children of a node under construction live in registers: every index into c[] below comes from a fully
// unrolled loop (a dynamically indexed array would be placed in scratch memory, which made the first version of this
// kernel several times slower), and "element k" for a run-time k is a chain of selects.
__device__ __forceinline__ void take_if(WideChild &dst, const WideChild &v, bool p) {      // field-wise selects, never a conditional store
#pragma unroll
    for (int a = 0; a < 3; a++) { dst.lo[a] = p ? v.lo[a] : dst.lo[a]; dst.hi[a] = p ? v.hi[a] : dst.hi[a]; }
    dst.link = p ? v.link : dst.link; dst.count = p ? v.count : dst.count; dst.budget = p ? v.budget : dst.budget; dst.eff = p ? v.eff : dst.eff;
}
struct WideSet {
    WideChild c[8];
    __device__ __forceinline__ WideChild get(int k) const {
        WideChild r = c[0];
#pragma unroll
        for (int i = 1; i < 8; i++) take_if(r, c[i], i == k);
        return r;
    }
    __device__ __forceinline__ void set(int k, const WideChild &v) {
#pragma unroll
        for (int i = 0; i < 8; i++) take_if(c[i], v, i == k);
    }
};

// One wide node: collapses the subtree of pair `src` into node `dst` of level `level`; its inner children are appended
// to q_out (the queue of level + 1) and numbered next_base + position (next_base = nodes of levels 0 .. level).
__device__ __forceinline__ void wide_emit(const float4 *pairs, const unsigned long long *choice, uint32_t level, uint32_t *level_count,
                                          uint32_t next_base, uint32_t src, uint32_t dst, uint2 *q_out, float4 *wide, uint4 *links) {
    {
        WideSet S;
#pragma unroll
        for (int k = 0; k < 8; k++) { for (int a = 0; a < 3; a++) { S.c[k].lo[a] = 0.0f; S.c[k].hi[a] = 0.0f; } S.c[k].link = 0u; S.c[k].count = 0u; S.c[k].budget = 1u; S.c[k].eff = 0u; }
        int n = 0;
        {
            WideChild a, b;
            read_half(pairs + 4 * (size_t)src, a);
            read_half(pairs + 4 * (size_t)src + 2, b);
            if (choice) {      // best split of 8 slots, and the children's own tables
                const unsigned long long w = choice[src];
                const uint32_t k8 = (uint32_t)(w >> 18) & 7u;
                a.budget = k8; b.budget = 8u - k8; a.eff = (uint32_t)(w >> 21) & 0x1FFFFFu; b.eff = (uint32_t)(w >> 42) & 0x1FFFFFu;
            }
            if (!pad_leaf(a.count)) { S.set(n, a); n++; }
            if (!pad_leaf(b.count)) { S.set(n, b); n++; }
        }
        while (n < 8) {
            int best = -1; float best_area = -1.0f; uint32_t best_link = 0, best_budget = 1;
            if (choice) {
                // an inner child that the table dissolves when offered its budget (eff > 1); any order gives the same set
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    if (k >= n || S.c[k].count != 0u || S.c[k].budget < 2u || best >= 0) continue;
                    const uint32_t bud = S.c[k].budget > 7u ? 7u : S.c[k].budget;
                    const uint32_t eff = (S.c[k].eff >> (3 * (bud - 1u))) & 7u;
                    if (eff > 1u) { best = k; best_link = S.c[k].link; best_budget = eff; }
                    else S.c[k].budget = 1u;      // stays a wide node of its own: do not look at it again
                }
            } else {
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    if (k >= n || S.c[k].count != 0u) continue;
                    const float a = half_area3(S.c[k].lo, S.c[k].hi);
                    if (a > best_area) { best_area = a; best = k; best_link = S.c[k].link; }
                }
            }
            if (best < 0) break;
            WideChild a, b;
            read_half(pairs + 4 * (size_t)best_link, a);
            read_half(pairs + 4 * (size_t)best_link + 2, b);
            if (choice) {      // loaded together with the pair: the split of the slots it was granted and its children's tables
                const unsigned long long w = choice[best_link];
                const uint32_t kk = (uint32_t)(w >> (3 * (best_budget - 2u))) & 7u;
                a.budget = kk; b.budget = best_budget - kk; a.eff = (uint32_t)(w >> 21) & 0x1FFFFFu; b.eff = (uint32_t)(w >> 42) & 0x1FFFFFu;
            }
            const bool pa = pad_leaf(a.count), pb = pad_leaf(b.count);
            if (pa && pb) { S.set(best, S.get(n - 1)); n--; continue; }
            if (pa) { S.set(best, b); continue; }
            S.set(best, a);
            if (!pb) { S.set(n, b); n++; }
        }
        // node box, origin (rounded down to 15 mantissa bits) and per-axis grid step
        // All in single precision (the double-precision version of this block and of the quantisation below ran at a half to
        // a sixteenth of the FP32 rate and was most of this kernel's time); conservative by construction, see below.
        float lo[3], hi[3], origin[3]; uint32_t ebits[3]; float inv_scale[3];
#pragma unroll
        for (int a = 0; a < 3; a++) {
            lo[a] = S.c[0].lo[a]; hi[a] = S.c[0].hi[a];
#pragma unroll
            for (int k = 1; k < 8; k++) if (k < n) { lo[a] = fminf(lo[a], S.c[k].lo[a]); hi[a] = fmaxf(hi[a], S.c[k].hi[a]); }
            uint32_t b = __float_as_uint(lo[a]), t = b & 0xFFFFFE00u;
            if ((b & 0x80000000u) && t != b) t += 0x200u;          // negative: truncation moved it up, step one unit down
            origin[a] = __uint_as_float(t);
            // grid step 2^e with 255 * 2^e >= hi - origin.  ext carries a relative error of at most 2^-24 (one rounded
            // subtraction), the division another 2^-24, so a quotient within 2^-20 of the next power of two takes that one.
            const float ext = hi[a] - origin[a];
            int e = -100;
            if (ext > 0.0f) {
                int x; const float m = frexpf(ext / 255.0f, &x);      // ext / 255 = m * 2^x, 0.5 <= m < 1
                e = m > 0.999999f ? x + 1 : x;
            }
            if (!(ext <= 3.402823466e38f)) e = 122;                   // the subtraction overflowed (or a NaN box): 255 * 2^122 covers any finite extent
            e = e < -100 ? -100 : e > 122 ? 122 : e;
            ebits[a] = (uint32_t)(e + 127);
            inv_scale[a] = __uint_as_float((uint32_t)(127 - e) << 23);      // 2^-e, exact (e in [-100, 122])
        }
        // slot assignment: greedy maximum of sum_a (slot bit a ? +1 : -1) * (child centroid - node centre)_a.
        // slot_of: four bits per child; used / done: one bit per slot / child.
        float dv[8][3];
#pragma unroll
        for (int k = 0; k < 8; k++)
#pragma unroll
            for (int a = 0; a < 3; a++) dv[k][a] = (0.5f * S.c[k].lo[a] + 0.5f * S.c[k].hi[a]) - (0.5f * lo[a] + 0.5f * hi[a]);
        uint32_t slot_of = 0, slot_used = 0, child_done = 0;
        if (!choice) {
            // The per-frame LBVH rebuild (no table): children in order of decreasing |d|_1 (the score of a child's ideal slot, so the order in which the greedy rule
            // below serves children whose ideal slot is still free); each takes its best free slot.  An eighth of the
            // evaluations of the full greedy rule; on the headline scene it costs 0.14 % more node visits, so static (SAH +
            // table) builds keep the full rule.
            float l1[8];
#pragma unroll
            for (int k = 0; k < 8; k++) l1[k] = (fabsf(dv[k][0]) + fabsf(dv[k][1])) + fabsf(dv[k][2]);
            for (int it = 0; it < n; it++) {
                float top = -1.0f; int bk = -1;
#pragma unroll
                for (int k = 0; k < 8; k++) if (k < n && !((child_done >> k) & 1u) && l1[k] > top) { top = l1[k]; bk = k; }
                if (bk < 0) bk = __builtin_ctz(~child_done & ((1u << n) - 1u));      // only NaNs left
                float d0 = dv[0][0], d1 = dv[0][1], d2 = dv[0][2];
#pragma unroll
                for (int k = 1; k < 8; k++) { d0 = k == bk ? dv[k][0] : d0; d1 = k == bk ? dv[k][1] : d1; d2 = k == bk ? dv[k][2] : d2; }
                float best = -3.402823466e38f; int bs = -1;
#pragma unroll
                for (int sl = 0; sl < 8; sl++) {
                    const float v = ((sl & 1) ? d0 : -d0) + ((sl & 2) ? d1 : -d1) + ((sl & 4) ? d2 : -d2);
                    if (!((slot_used >> sl) & 1u) && v > best) { best = v; bs = sl; }
                }
                if (bs < 0) bs = __builtin_ctz(~slot_used & 0xFFu);
                slot_of |= (uint32_t)bs << (4 * bk); child_done |= 1u << bk; slot_used |= 1u << bs;
            }
        } else
        for (int it = 0; it < n; it++) {
            float best = -3.402823466e38f; int bk = -1, bs = -1;
#pragma unroll
            for (int k = 0; k < 8; k++) {
                if (k >= n || ((child_done >> k) & 1u)) continue;
#pragma unroll
                for (int sl = 0; sl < 8; sl++) {
                    if ((slot_used >> sl) & 1u) continue;
                    const float v = ((sl & 1) ? dv[k][0] : -dv[k][0]) + ((sl & 2) ? dv[k][1] : -dv[k][1]) + ((sl & 4) ? dv[k][2] : -dv[k][2]);
                    if (v > best) { best = v; bk = k; bs = sl; }
                }
            }
            if (bk < 0) {            // only NaN costs left: first free pair
                bk = __builtin_ctz(~child_done & ((1u << n) - 1u));
                bs = __builtin_ctz(~slot_used & 0xFFu);
            }
            slot_of |= (uint32_t)bs << (4 * bk); child_done |= 1u << bk; slot_used |= 1u << bs;
        }
        uint32_t imask = 0, n_inner = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) if (k < n && S.c[k].count == 0u) { imask |= 1u << ((slot_of >> (4 * k)) & 7u); n_inner++; }
        uint32_t qpos = 0;
        if (n_inner) qpos = atomicAdd(&level_count[level + 1], n_inner);
        const uint32_t child_base = next_base + qpos;
        // quantise: empty slots keep qlo = 255, qhi = 0
        uint32_t q[6][2];     // qlo x,y,z, qhi x,y,z; two dwords of four slots
        uint32_t link[8];
#pragma unroll
        for (int a = 0; a < 3; a++) { q[a][0] = q[a][1] = 0xFFFFFFFFu; q[3 + a][0] = q[3 + a][1] = 0u; }
#pragma unroll
        for (int sl = 0; sl < 8; sl++) link[sl] = kWideEmptyLink;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            if (k >= n) continue;
            const uint32_t sl = (slot_of >> (4 * k)) & 7u, sh = 8u * (sl & 3u);
            const bool upper = sl >= 4u;
#pragma unroll
            for (int a = 0; a < 3; a++) {
                // x = (plane - origin) / 2^e in [0, 255]: the subtraction rounds once (relative 2^-24, i.e. at most 2^-16 in x),
                // the scaling is exact; 2^-10 of slack towards the outside makes floor / ceil safe against that error, and
                // costs a grid step only for planes within 2^-10 of a grid line.
                const float xl = (S.c[k].lo[a] - origin[a]) * inv_scale[a], xh = (S.c[k].hi[a] - origin[a]) * inv_scale[a];
                float fl = floorf(xl - 0.0009765625f), fh = ceilf(xh + 0.0009765625f);
                fl = (fl >= 0.0f && fl <= 255.0f) ? fl : 0.0f;             // below the grid, or NaN / inf from an overflowed difference: the whole range
                fh = (fh >= 0.0f && fh <= 255.0f) ? fh : 255.0f;
                const uint32_t clr = ~(255u << sh), vl = (uint32_t)fl << sh, vh = (uint32_t)fh << sh;
                if (upper) { q[a][1] = (q[a][1] & clr) | vl; q[3 + a][1] |= vh; }
                else { q[a][0] = (q[a][0] & clr) | vl; q[3 + a][0] |= vh; }
            }
            uint32_t lk;
            if (S.c[k].count == 0u) {
                const uint32_t r = (uint32_t)__builtin_popcount(imask & ((1u << sl) - 1u));     // inner children are numbered in slot order
                lk = child_base + r;
                q_out[qpos + r] = make_uint2(S.c[k].link, lk);
            } else {
                const uint32_t kind = S.c[k].count >> 28, cnt = S.c[k].count & 0x0FFFFFFFu;
                lk = kWideLeafTag | (kind << 29) | ((cnt - 1u) << 26) | S.c[k].link;
            }
#pragma unroll
            for (int s2 = 0; s2 < 8; s2++) link[s2] = (uint32_t)s2 == sl ? lk : link[s2];
        }
        float4 *o = wide + (size_t)kWideStrideF4 * dst;
        o[0] = make_float4(__uint_as_float(__float_as_uint(origin[0]) | ebits[0]), __uint_as_float(__float_as_uint(origin[1]) | ebits[1]),
                           __uint_as_float(__float_as_uint(origin[2]) | ebits[2]), __uint_as_float((child_base << 8) | imask));
        o[1] = make_float4(__uint_as_float(q[0][0]), __uint_as_float(q[0][1]), __uint_as_float(q[1][0]), __uint_as_float(q[1][1]));
        o[2] = make_float4(__uint_as_float(q[2][0]), __uint_as_float(q[2][1]), __uint_as_float(q[3][0]), __uint_as_float(q[3][1]));
        o[3] = make_float4(__uint_as_float(q[4][0]), __uint_as_float(q[4][1]), __uint_as_float(q[5][0]), __uint_as_float(q[5][1]));
        links[2 * (size_t)dst] = make_uint4(link[0], link[1], link[2], link[3]);
        links[2 * (size_t)dst + 1] = make_uint4(link[4], link[5], link[6], link[7]);
    }
}

// level_count[l] = wide nodes of level l (queue length); level l reads queue l & 1 and appends to the other one.
// Levels below *first_level were already emitted by k_wide_top.
__global__ __launch_bounds__(128) void k_wide_level(const float4 *pairs, const unsigned long long *choice, uint32_t level, uint32_t *level_count,
                                                    const uint32_t *first_level, const uint2 *q_in, uint2 *q_out, float4 *wide, uint4 *links) {
    if (level < *first_level) return;
    const uint32_t n_in = level_count[level];
    uint32_t next_base = 0;                         // nodes of levels 0..level = index of the first node of level + 1
    for (uint32_t l = 0; l <= level; l++) next_base += level_count[l];
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_in; i += gridDim.x * blockDim.x) {
        const uint2 w = q_in[i];
        wide_emit(pairs, choice, level, level_count, next_base, w.x, w.y, q_out, wide, links);
    }
}

// The first levels hold 1, <= 8, <= 64, ... nodes; a launch per level costs ~25 us of dependent-load latency each however
// few nodes it has.  One workgroup walks the levels that fit its threads, one node per thread, and leaves the first
// level it did not do in *first_level (k_wide_level skips the earlier ones).  Everything it hands from one level to the
// next (queue entries, level counts) stays inside the workgroup: workgroup-scope release / acquire around the barrier.
constexpr uint32_t kWideTopThreads = 512;
__global__ __launch_bounds__(kWideTopThreads) void k_wide_top(const float4 *pairs, const unsigned long long *choice, uint32_t *level_count, uint32_t max_levels,
                                                              uint32_t *first_level, uint2 *q0, uint2 *q1, float4 *wide, uint4 *links) {
    uint32_t next_base = 0, level = 0;
    for (; level < max_levels; level++) {
        const uint32_t n_in = __hip_atomic_load(&level_count[level], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (n_in == 0u || n_in > kWideTopThreads) break;
        next_base += n_in;
        const uint2 *q_in = (level & 1u) ? q1 : q0;
        uint2 *q_out = (level & 1u) ? q0 : q1;
        if (threadIdx.x < n_in) {
            const uint2 w = q_in[threadIdx.x];
            wide_emit(pairs, choice, level, level_count, next_base, w.x, w.y, q_out, wide, links);
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
    uint4 *links = (uint4 *)((float4 *)st.wide.ptr + (size_t)kWideStrideF4 * st.wide_cap);
    if (!ensure_w(st.wide_queue[0], (size_t)st.n_pairs * 8, err) || !ensure_w(st.wide_queue[1], (size_t)st.n_pairs * 8, err)) return false;
    if (!ensure_w(st.wide_levels_dev, (kWideMaxLevels + 3) * 4, err)) return false;
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
    hipLaunchKernelGGL(k_wide_top, dim3(1), dim3(kWideTopThreads), 0, stream, (const float4 *)st.pairs.ptr, choice, lc, kWideMaxLevels, first_level,
                       (uint2 *)st.wide_queue[0].ptr, (uint2 *)st.wide_queue[1].ptr, (float4 *)st.wide.ptr, links);
    const uint32_t grid = std::min<uint32_t>(2048u, (st.n_pairs + 127u) / 128u);
    uint32_t host_lc[kWideMaxLevels + 3];
    uint32_t root_word = 0; bool height_read = false;
    uint32_t level = 1;      // k_wide_top always does level 0 (one node)
    for (;;) {
        const uint32_t chunk_end = std::min<uint32_t>(level + 8u, kWideMaxLevels);
        for (; level < chunk_end; level++)
            hipLaunchKernelGGL(k_wide_level, dim3(grid), dim3(128), 0, stream, (const float4 *)st.pairs.ptr, choice, level, lc, (const uint32_t *)first_level,
                               (const uint2 *)st.wide_queue[level & 1].ptr, (uint2 *)st.wide_queue[(level & 1) ^ 1].ptr,
                               (float4 *)st.wide.ptr, links);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(host_lc, lc, (kWideMaxLevels + 3) * 4, hipMemcpyDeviceToHost, stream));
        if (st.dp_fused && !height_read) HIP_TRY(hipMemcpyAsync(&root_word, (const float *)st.dp_cost.ptr + 7, 4, hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        if (st.dp_fused && !height_read) {      // the pair tree's height came with the table (bvh_sah.hip): the pair tracers' stacks bound it
            height_read = true;
            st.pair_depth = root_word >> 21;
            if (st.pair_depth > pair_depth_limit()) {
                err = "SAH-refined tree is " + std::to_string(st.pair_depth) + " levels deep; the pair traversal stacks hold " + std::to_string(pair_depth_limit());
                return false;
            }
        }
        if (host_lc[level] == 0) break;
        if (level >= kWideMaxLevels) { err = "wide BVH deeper than 64 levels"; return false; }
    }
    for (uint32_t l = 0; l <= kWideMaxLevels && host_lc[l]; l++) { st.n_wide += host_lc[l]; st.wide_levels = l + 1; }
    return true;
}

}  // namespace crt
