// bvh_sah.hip -- SAH refinement of the Morton LBVH (CRT_BVH_SAH_REFINE).
//
// The reference builds its tree with a full-sweep surface-area heuristic on the host
// (raytracing/src/bounding_volume_hierarchy.cpp:26-79, row B1 of SURVEY.md section 8(a)).  This is the device
// counterpart: the LBVH (Morton sort + Karras hierarchy, bvh_build.hip) supplies a spatial order and a
// clustering, and the tree topology is then rebuilt top-down with the SAH:
//
//   tiers   Level-0 items are the primitives in Morton order.  The level-(l+1) items ("groups") are the maximal
//           LBVH subtrees holding at most kGroup level-l items; a group is a contiguous run of items and, being
//           an LBVH subtree, an aligned cell of the Morton grid.  Tiers are added until one group holds everything.
//   groups  One 256-thread workgroup rebuilds one group entirely in LDS (boxes + index permutation):
//             nodes with more than 64 items: all four waves, 16-bin SAH on three axes (LDS atomic min/max on
//               order-preserving keys), stable partition by ballot + prefix counts;
//             nodes with at most 64 items: one wave per subtree, exact sweep SAH (every item's centroid is a
//               candidate plane on every axis), all nodes of a level at once as segments of lanes (sah_wave_subtree).
//           Level-0 groups may close a subtree of one kind as a multi-primitive leaf when the SAH says so.
//   layout  A subtree over k items owns k - 1 consecutive 64-byte sibling pairs, itself first, then its left
//           subtree, then its right one (depth-first order), so no allocation counter is needed and a group's
//           pairs start at a base computed from the tier sizes.  Pair 0 is the root.
// Everything is deterministic (integer atomics on encoded floats, stable partitions).
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <string>
#include <vector>

#include <rocprim/device/device_scan.hpp>

#include "build_internal.h"
#include "bvh_common.h"
#include "bvh_dp.h"

namespace crt {

#ifndef CRT_SAH_GROUP
#define CRT_SAH_GROUP 256
#endif
constexpr uint32_t kGroup = CRT_SAH_GROUP;       // items per group (A/B on MI355X: 256 builds fastest, trace speed within noise of 1024)
#ifndef CRT_SAH_THREADS
#define CRT_SAH_THREADS 256
#endif
constexpr uint32_t kSahThreads = CRT_SAH_THREADS;   // the waves of a group share its small subtrees; several groups are resident per CU
constexpr uint32_t kSahWaves = kSahThreads / 64;
constexpr uint32_t kWaveNode = 64;      // nodes up to this size are built by one wave
constexpr int kBins = 16;
constexpr uint32_t kEntRoot = 1u << 23;
constexpr float kFInf = __builtin_huge_valf();

// ---- tiers ---------------------------------------------------------------------------------------------------
// Node numbering of the LBVH: inner i in [0, n-2], leaf j -> n - 1 + j.  ipos[s] = 1-based index of the level-l
// item that holds sorted position s (null: level 0, every position is its own item).
__device__ __forceinline__ uint32_t items_under(uint32_t node, uint32_t n, const uint32_t *range_lo, const uint32_t *range_hi,
                                                const uint32_t *ipos, uint32_t &first) {
    uint32_t f, l;
    if (node < n - 1) { f = range_lo[node]; l = range_hi[node]; } else { f = l = node - (n - 1); }
    first = f;
    return ipos ? ipos[l] - ipos[f] + 1u : l - f + 1u;
}

__global__ void k_sah_flags(uint32_t n, uint32_t max_items, const uint32_t *range_lo, const uint32_t *range_hi,
                            const uint32_t *parent, const uint32_t *ipos, uint32_t *flag) {
    uint32_t x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= 2 * n - 1) return;
    uint32_t first, pf;
    if (items_under(x, n, range_lo, range_hi, ipos, first) > max_items) return;
    uint32_t p = parent[x];
    if (p != 0xFFFFFFFFu && items_under(p, n, range_lo, range_hi, ipos, pf) <= max_items) return;
    flag[first] = 1u;      // x is a maximal subtree of at most max_items items: a group starts here
}

__global__ void k_sah_starts(uint32_t n, const uint32_t *flag, const uint32_t *ipos_next, const uint32_t *ipos, uint32_t *starts) {
    uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n || !flag[s]) return;
    starts[ipos_next[s] - 1u] = ipos ? ipos[s] - 1u : s;      // first level-l item of the group
}

// ---- one group ------------------------------------------------------------------------------------------------
struct SahGroupArgs {
    const uint32_t *starts; uint32_t n_groups, n_items, pair_base, level0;
    const uint32_t *vals; const float4 *plo, *phi; uint32_t n_spheres, n_quads;   // level 0: items are primitives
    const float4 *desc_in;     // upper tiers: two float4 per item, in the layout of half a sibling pair
    float4 *desc_out;          // two float4 per group (null for the last tier)
    float4 *pairs;
    uint32_t *perm;            // level 0: final primitive order
    uint32_t max_leaf; float c_isect, c_trav;
    // fused collapse table (bvh_wide.hip; null: not wanted): eight words per pair (only a group's root row is written: the
    // next tier reads it), one choice word per pair
    float *dp_cost; unsigned long long *dp_choice; uint32_t n_pairs_total;
};

struct SahLds {
    float box[6][kGroup];            // lo.xyz, hi.xyz of the group's items
    uint16_t idx[kGroup], tmp[kGroup];
    uint8_t kind[kGroup];
    uint2 ent[2 * kGroup];           // work entries: stack of large nodes from the bottom, list of small subtrees from the top
    uint32_t bins[3 * kBins * 7];    // per axis and bin: count, lo.xyz, hi.xyz (encoded)
    uint32_t red[12];                // centroid min, centroid max, box min, box max (encoded)
    uint32_t wave_cnt[kSahWaves];
    uint32_t top, n_small, cursor, best_axis, best_plane, best_nl;
    float dp[7][kGroup];             // fused collapse table: c(1..7) of the group's pairs
    uint32_t dp_w[kGroup];           // their eff tables | subtree height << 21
    uint8_t dp_done[kGroup];
};
constexpr bool kSahCanFuseDp = kGroup <= kSahThreads;      // one thread per pair of a group

// work entry: x = a | b << 16 (item range [a, b) of the permutation), y = p | pp << 11 | side << 22 | root << 23
// (p = first pair owned by the subtree, pp/side = where its descriptor goes: half `side` of pair pp).
__device__ __forceinline__ uint2 make_ent(uint32_t a, uint32_t b, uint32_t p, uint32_t pp, uint32_t side) {
    return make_uint2(a | (b << 16), p | (pp << 11) | (side << 22));
}

__device__ __forceinline__ void put_desc(const SahGroupArgs &g, uint32_t group, uint32_t pair_base, uint32_t y,
                                         const float lo[3], const float hi[3], uint32_t link, uint32_t count) {
    float4 *dst;
    if (y & kEntRoot) { if (!g.desc_out) return; dst = g.desc_out + 2 * (size_t)group; }
    else dst = g.pairs + 4 * (size_t)(pair_base + ((y >> 11) & 0x7FFu)) + 2 * ((y >> 22) & 1u);
    dst[0] = make_float4(lo[0], lo[1], lo[2], __uint_as_float(link));
    dst[1] = make_float4(hi[0], hi[1], hi[2], __uint_as_float(count));
}

// Descriptor of the single item at permutation position a.
__device__ __forceinline__ void put_item(const SahGroupArgs &g, const SahLds &L, uint32_t group, uint32_t gs, uint32_t pair_base,
                                         uint32_t y, uint32_t a) {
    const uint32_t it = L.idx[a];
    float lo[3] = {L.box[0][it], L.box[1][it], L.box[2][it]}, hi[3] = {L.box[3][it], L.box[4][it], L.box[5][it]};
    uint32_t link, count;
    if (g.level0) { link = gs + a; count = ((uint32_t)L.kind[it] << 28) | 1u; }       // link = final position, patched to the kind's slot later
    else { link = __float_as_uint(g.desc_in[2 * (size_t)(gs + it)].w); count = __float_as_uint(g.desc_in[2 * (size_t)(gs + it) + 1].w); }
    put_desc(g, group, pair_base, y, lo, hi, link, count);
}

__device__ __forceinline__ float wave_min(float v) { for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o)); return v; }
__device__ __forceinline__ float wave_max(float v) { for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o)); return v; }

// Ascending bitonic sort of one 64-bit key per lane (keys are distinct: the lane is part of the key).
__device__ __forceinline__ unsigned long long wave_sort(unsigned long long key, uint32_t lane) {
#pragma unroll
    for (uint32_t k2 = 2; k2 <= 64; k2 <<= 1) {
#pragma unroll
        for (uint32_t j = k2 >> 1; j > 0; j >>= 1) {
            const unsigned long long other = __shfl_xor(key, (int)j);
            const bool take_min = ((lane & j) == 0) == ((lane & k2) == 0);
            key = (other < key) == take_min ? other : key;
        }
    }
    return key;
}

// One wave builds the whole subtree of a node with at most kWaveNode items, level by level.  Item j of the node lives in
// lane j for good (its box, kind and LDS index: the "home" lane).  What moves are three permutations, one per axis:
// ord[ax] of lane l = the item at position l when every open node's items are in ascending centroid order on that axis.
// The nodes of the current level are segments [s0, s1) of consecutive POSITIONS, the same in all three orders, and every
// lane carries the bookkeeping of the segment its position belongs to.  Per level and axis: fetch the boxes in that
// axis's order, segmented prefix / suffix scans of the boxes price every split of every segment at once (exact sweep
// SAH: each item's centroid is a candidate plane).  The cheapest (axis, rank) of a segment is found with a segmented
// min-scan; the items left of it are flagged at their home lanes and all three orders are partitioned stably
// (ballot + prefix count + one ds_permute each), which keeps each side sorted on every axis.  The orders are therefore
// sorted once, up front: re-sorting every level (three 21-stage bitonic networks) was two thirds of this function's
// dependent shuffles.
// Segmented inclusive scans over the wave on the VALU (DPP) instead of through the LDS crossbar: the wave-subtree loop
// spent half of its cycles waiting for ds_bpermute results (SQ_WAIT_ANY 0.49, SQ_WAIT_INST_LDS 0.24 of the wave cycles).
// Lane l combines with lanes [s0, l) of its segment: four row_shr steps inside the 16-lane rows, then row_bcast:15 into
// rows 1 and 3 and row_bcast:31 into rows 2 and 3 carry the finished part of the segment across the row borders (a lane
// takes them only when its segment starts before its row / before lane 32).  `old` = the operation's identity, which is
// what a lane keeps when DPP gives it no source.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_u(uint32_t old, uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp((int)old, (int)v, CTRL, ROW_MASK, 0xF, false); }
template <bool MAX>
__device__ __forceinline__ float seg_scan(float v, uint32_t lane, uint32_t s0) {
    const float id = MAX ? -kFInf : kFInf;
    const uint32_t idu = __float_as_uint(id);
#define CRT_SEG_STEP(CTRL, RM, COND) { float o = __uint_as_float(dpp_u<CTRL, RM>(idu, __float_as_uint(v))); o = (COND) ? o : id; v = MAX ? fmaxf(v, o) : fminf(v, o); }
    CRT_SEG_STEP(0x111, 0xF, lane >= s0 + 1u)
    CRT_SEG_STEP(0x112, 0xF, lane >= s0 + 2u)
    CRT_SEG_STEP(0x114, 0xF, lane >= s0 + 4u)
    CRT_SEG_STEP(0x118, 0xF, lane >= s0 + 8u)
    CRT_SEG_STEP(0x142, 0xA, s0 < (lane & 48u))
    CRT_SEG_STEP(0x143, 0xC, s0 < 32u)
#undef CRT_SEG_STEP
    return v;
}
// The same for the minimum of a 64-bit key held as two words.
__device__ __forceinline__ void seg_scan_min64(uint32_t &hi, uint32_t &lo, uint32_t lane, uint32_t s0) {
#define CRT_SEG_STEP(CTRL, RM, COND) { const uint32_t oh = dpp_u<CTRL, RM>(0xFFFFFFFFu, hi), ol = dpp_u<CTRL, RM>(0xFFFFFFFFu, lo); \
        const bool take = (COND) && (oh < hi || (oh == hi && ol < lo)); hi = take ? oh : hi; lo = take ? ol : lo; }
    CRT_SEG_STEP(0x111, 0xF, lane >= s0 + 1u)
    CRT_SEG_STEP(0x112, 0xF, lane >= s0 + 2u)
    CRT_SEG_STEP(0x114, 0xF, lane >= s0 + 4u)
    CRT_SEG_STEP(0x118, 0xF, lane >= s0 + 8u)
    CRT_SEG_STEP(0x142, 0xA, s0 < (lane & 48u))
    CRT_SEG_STEP(0x143, 0xC, s0 < 32u)
#undef CRT_SEG_STEP
}

__device__ __forceinline__ uint32_t lane_push(uint32_t to_lane, uint32_t v) { return (uint32_t)__builtin_amdgcn_ds_permute((int)(to_lane << 2), (int)v); }
__device__ __forceinline__ uint32_t lane_pull(uint32_t from_lane, uint32_t v) { return (uint32_t)__builtin_amdgcn_ds_bpermute((int)(from_lane << 2), (int)v); }

__device__ void sah_wave_subtree(const SahGroupArgs &g, SahLds &L, uint32_t group, uint32_t gs, uint32_t pair_base, uint2 e, uint32_t lane) {
    const uint32_t ex = (uint32_t)__builtin_amdgcn_readfirstlane((int)e.x);
    const uint32_t a = ex & 0xFFFFu, k = (ex >> 16) - a;
    const bool act = lane < k;
    // home data of item `lane`
    const uint32_t it = act ? L.idx[a + lane] : 0u;
    float lo[3], hi[3];
#pragma unroll
    for (int ax = 0; ax < 3; ax++) { lo[ax] = act ? L.box[ax][it] : kFInf; hi[ax] = act ? L.box[3 + ax][it] : -kFInf; }
    const uint32_t kind = (g.level0 && act) ? (uint32_t)L.kind[it] : 0u;
    uint32_t ord[3];
#pragma unroll
    for (int ax = 0; ax < 3; ax++) {      // lanes past k sort behind the items, in lane order: they keep ord = lane
        const float c = 0.5f * lo[ax] + 0.5f * hi[ax];
        const unsigned long long key = act ? (((unsigned long long)enc_f(c) << 8) | lane) : ((1ull << 40) | lane);
        ord[ax] = (uint32_t)wave_sort(key, lane) & 63u;
    }
    uint32_t s0 = act ? 0u : lane, s1 = act ? k : lane + 1u;         // positions past k: closed one-lane segments that stay put
    uint32_t y = (uint32_t)__builtin_amdgcn_readfirstlane((int)e.y);  // p | pp << 11 | side << 22 | root << 23 of my segment
    bool open = act;
    for (;;) {
        // the items in axis-0 order: that order is the final one (links of single items and leaves are positions)
        float blo[3], bhi[3];
#pragma unroll
        for (int d = 0; d < 3; d++) { blo[d] = __shfl(lo[d], (int)ord[0]); bhi[d] = __shfl(hi[d], (int)ord[0]); }
        const uint32_t kind_p = lane_pull(ord[0], kind), it_p = lane_pull(ord[0], it);
        if (open && s1 - s0 == 1u) {                                  // a single item: its descriptor, done
            uint32_t link, count;
            if (g.level0) { link = gs + a + lane; count = (kind_p << 28) | 1u; }       // link = final position, patched to the kind's slot later
            else { link = __float_as_uint(g.desc_in[2 * (size_t)(gs + it_p)].w); count = __float_as_uint(g.desc_in[2 * (size_t)(gs + it_p) + 1].w); }
            put_desc(g, group, pair_base, y, blo, bhi, link, count);
            open = false;
        }
        if (__ballot(open) == 0ull) break;
        const uint32_t len = s1 - s0;
        float best = kFInf, nlo[3], nhi[3];
        uint32_t best_ax = 0;
        // The right-hand boxes are prefix scans too, over the wave in reverse: lane l' of the reversed view holds position
        // 63 - l', whose segment starts (reversed) at 64 - s1.
        const uint32_t rev = 63u - lane, s0r = 64u - lane_pull(rev, s1);
#pragma unroll
        for (int ax = 0; ax < 3; ax++) {
            const uint32_t ord_rev = lane_pull(rev, ord[ax]);
            float llo[3], lhi[3], rlo[3], rhi[3];
#pragma unroll
            for (int d = 0; d < 3; d++) {
                llo[d] = ax == 0 ? blo[d] : __shfl(lo[d], (int)ord[ax]);
                lhi[d] = ax == 0 ? bhi[d] : __shfl(hi[d], (int)ord[ax]);
                rlo[d] = __shfl(lo[d], (int)ord_rev);
                rhi[d] = __shfl(hi[d], (int)ord_rev);
            }
#pragma unroll
            for (int d = 0; d < 3; d++) {      // left box up to me; (reversed) right box from me
                llo[d] = seg_scan<false>(llo[d], lane, s0); lhi[d] = seg_scan<true>(lhi[d], lane, s0);
                rlo[d] = seg_scan<false>(rlo[d], lane, s0r); rhi[d] = seg_scan<true>(rhi[d], lane, s0r);
            }
            if (ax == 0) {
#pragma unroll
                for (int d = 0; d < 3; d++) { nlo[d] = __shfl(llo[d], (int)(s1 - 1u)); nhi[d] = __shfl(lhi[d], (int)(s1 - 1u)); }      // the segment's box
            }
#pragma unroll
            for (int d = 0; d < 3; d++) { rlo[d] = __shfl(rlo[d], (int)((62u - lane) & 63u)); rhi[d] = __shfl(rhi[d], (int)((62u - lane) & 63u)); }   // right box = everything after me
            const uint32_t nl = lane - s0 + 1u, nr = s1 - lane - 1u;
            float cost = half_area3(llo, lhi) * (float)nl + half_area3(rlo, rhi) * (float)nr;
            cost = (open && lane + 1u < s1 && cost < kFInf) ? cost : kFInf;      // also turns a NaN into "invalid"
            if (cost < best) { best = cost; best_ax = (uint32_t)ax; }
        }
        // cheapest candidate of the segment: inclusive min-scan, read at the segment's last lane
        // Ties go to the most balanced cut.  (Coincident primitives price every cut of a segment alike; lowest lane first
        // peeled ONE item per level, and thousands of copies of one triangle made trees a hundred levels deep, which the
        // pair tracers' stacks and the 8-wide format's 64 levels then refused: found by the random soups.)
        const uint32_t cand_l = lane - s0 + 1u, cand_r = s1 - lane - 1u;
        const uint32_t imbalance = (cand_l > cand_r ? cand_l - cand_r : cand_r - cand_l) & 0x7Fu;
        uint32_t bk_hi = __float_as_uint(best), bk_lo = (imbalance << 16) | (best_ax << 8) | lane;
        seg_scan_min64(bk_hi, bk_lo, lane, s0);
        const unsigned long long bk = ((unsigned long long)(uint32_t)__shfl((int)bk_hi, (int)(s1 - 1u)) << 32) | (uint32_t)__shfl((int)bk_lo, (int)(s1 - 1u));
        const float split_cost = __uint_as_float((uint32_t)(bk >> 32));
        const bool valid = split_cost < kFInf;
        const uint32_t bax = valid ? ((uint32_t)bk >> 8) & 3u : 0u;            // no finite candidate: halve the axis-0 order
        const uint32_t nl = valid ? ((uint32_t)bk & 63u) - s0 + 1u : len / 2u;
        // multi-primitive leaf?  (level 0, one kind, at most max_leaf primitives, not more expensive than the split)
        bool leaf = false;
        const uint32_t kind0 = (uint32_t)__shfl((int)kind_p, (int)s0);
        const unsigned long long mism = __ballot(open && kind_p != kind0);
        // The root of the whole tree is never closed as a leaf: it has no descriptor slot (put_desc drops it), and pair 0
        // would keep its all-zero memset, i.e. two "inner" children linking back to pair 0.  It always splits into pair 0.
        const bool tree_root = (y & kEntRoot) != 0u && g.desc_out == nullptr;
        if (g.level0 && open && len <= g.max_leaf && !tree_root) {
            const unsigned long long seg = (len >= 64u ? ~0ull : (1ull << len) - 1ull) << s0;
            const float area = half_area3(nlo, nhi);
            leaf = (mism & seg) == 0ull && g.c_isect * (float)len * area <= g.c_trav * area + g.c_isect * split_cost;
        }
        const uint32_t p = y & 0x7FFu;
        if (open && lane == s0) {
            if (leaf) put_desc(g, group, pair_base, y, nlo, nhi, gs + a + s0, (kind_p << 28) | len);
            else put_desc(g, group, pair_base, y, nlo, nhi, pair_base + p, 0u);
        }
        if (leaf) open = false;
        // Which items go left: position l of the winning axis's order holds one of the segment's items; tell its home lane.
        // Closed segments (and lanes past k) flag everything "left" of a cut at their end, which leaves them in place.
        const uint32_t cut = open ? s0 + nl : s1;
        const uint32_t winner = !open ? ord[0] : bax == 0 ? ord[0] : bax == 1 ? ord[1] : ord[2];
        const uint32_t left_home = lane_push(winner, lane < cut ? 1u : 0u);
        const unsigned long long below = ((1ull << lane) - 1ull) & ~((1ull << s0) - 1ull);      // positions [s0, lane)
#pragma unroll
        for (int ax = 0; ax < 3; ax++) {      // stable partition of every order
            const bool left = lane_pull(ord[ax], left_home) != 0u;
            const uint32_t before = (uint32_t)__popcll(__ballot(left) & below);
            const uint32_t to = left ? s0 + before : cut + (lane - s0 - before);
            ord[ax] = lane_push(to, ord[ax]);
        }
        if (open) {
            if (lane < cut) { s1 = cut; y = (p + 1u) | (p << 11); }
            else { s0 = cut; y = (p + nl) | (p << 11) | (1u << 22); }
        }
    }
    const uint32_t it_final = lane_pull(ord[0], it);
    if (act) L.idx[a + lane] = (uint16_t)it_final;
}

__device__ __forceinline__ int bin_of(float c, float cmin, float scale) {
    int b = (int)((c - cmin) * scale);
    return b < 0 ? 0 : b > kBins - 1 ? kBins - 1 : b;
}

__global__ __launch_bounds__(kSahThreads) void k_sah_group(SahGroupArgs g) {
    __shared__ SahLds L;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t group = blockIdx.x;
    const uint32_t gs = g.starts[group], ge = group + 1 < g.n_groups ? g.starts[group + 1] : g.n_items, m = ge - gs;
    const uint32_t pair_base = g.pair_base + gs - group;
    for (uint32_t i = tid; i < m; i += kSahThreads) {
        float4 lo, hi;
        if (g.level0) {
            const uint32_t prim = g.vals[gs + i];
            lo = g.plo[prim]; hi = g.phi[prim];
            L.kind[i] = (uint8_t)(prim < g.n_spheres ? CRT_ID_KIND_SPHERE : prim < g.n_spheres + g.n_quads ? CRT_ID_KIND_QUAD : CRT_ID_KIND_TRI);
        } else { lo = g.desc_in[2 * (size_t)(gs + i)]; hi = g.desc_in[2 * (size_t)(gs + i) + 1]; }
        L.box[0][i] = lo.x; L.box[1][i] = lo.y; L.box[2][i] = lo.z; L.box[3][i] = hi.x; L.box[4][i] = hi.y; L.box[5][i] = hi.z;
        L.idx[i] = (uint16_t)i;
    }
    if (tid == 0) {
        const uint2 root = make_uint2(0u | (m << 16), kEntRoot);
        L.top = 0; L.n_small = 0; L.cursor = 0;
        if (m > kWaveNode) { L.ent[0] = root; L.top = 1; } else { L.ent[2 * kGroup - 1] = root; L.n_small = 1; }
    }
    // ---- nodes with more than kWaveNode items: the whole workgroup, binned SAH ----
    for (;;) {
        __syncthreads();
        const uint32_t top = L.top;
        if (top == 0) break;
        const uint2 e = L.ent[top - 1];
        const uint32_t a = e.x & 0xFFFFu, b = e.x >> 16, k = b - a, p = e.y & 0x7FFu;
        if (tid < 12) L.red[tid] = (tid % 6) < 3 ? 0xFFFFFFFFu : 0u;
        for (uint32_t t = tid; t < 3 * kBins * 7; t += kSahThreads) { const uint32_t f = t % 7; L.bins[t] = f == 0 ? 0u : f <= 3 ? 0xFFFFFFFFu : 0u; }
        __syncthreads();
        {   // bounds of the centroids and of the node
            float cmn[3] = {kFInf, kFInf, kFInf}, cmx[3] = {-kFInf, -kFInf, -kFInf}, bmn[3] = {kFInf, kFInf, kFInf}, bmx[3] = {-kFInf, -kFInf, -kFInf};
            for (uint32_t i = a + tid; i < b; i += kSahThreads) {
                const uint32_t it = L.idx[i];
#pragma unroll
                for (int ax = 0; ax < 3; ax++) {
                    const float lo = L.box[ax][it], hi = L.box[3 + ax][it], c = 0.5f * lo + 0.5f * hi;
                    cmn[ax] = fminf(cmn[ax], c); cmx[ax] = fmaxf(cmx[ax], c); bmn[ax] = fminf(bmn[ax], lo); bmx[ax] = fmaxf(bmx[ax], hi);
                }
            }
#pragma unroll
            for (int ax = 0; ax < 3; ax++) {
                cmn[ax] = wave_min(cmn[ax]); cmx[ax] = wave_max(cmx[ax]); bmn[ax] = wave_min(bmn[ax]); bmx[ax] = wave_max(bmx[ax]);
                if (lane == 0) {
                    atomicMin(&L.red[ax], enc_f(cmn[ax])); atomicMax(&L.red[3 + ax], enc_f(cmx[ax]));
                    atomicMin(&L.red[6 + ax], enc_f(bmn[ax])); atomicMax(&L.red[9 + ax], enc_f(bmx[ax]));
                }
            }
        }
        __syncthreads();
        float cmin[3], scale[3];
#pragma unroll
        for (int ax = 0; ax < 3; ax++) {
            cmin[ax] = dec_f(L.red[ax]);
            const float ext = dec_f(L.red[3 + ax]) - cmin[ax];
            scale[ax] = ext > 0.0f ? (float)kBins / ext : 0.0f;
        }
        for (uint32_t i = a + tid; i < b; i += kSahThreads) {
            const uint32_t it = L.idx[i];
            uint32_t elo[3], ehi[3]; float c[3];
#pragma unroll
            for (int ax = 0; ax < 3; ax++) {
                const float lo = L.box[ax][it], hi = L.box[3 + ax][it];
                c[ax] = 0.5f * lo + 0.5f * hi; elo[ax] = enc_f(lo); ehi[ax] = enc_f(hi);
            }
#pragma unroll
            for (int ax = 0; ax < 3; ax++) {
                uint32_t *bin = &L.bins[(ax * kBins + bin_of(c[ax], cmin[ax], scale[ax])) * 7];
                atomicAdd(&bin[0], 1u);
                atomicMin(&bin[1], elo[0]); atomicMin(&bin[2], elo[1]); atomicMin(&bin[3], elo[2]);
                atomicMax(&bin[4], ehi[0]); atomicMax(&bin[5], ehi[1]); atomicMax(&bin[6], ehi[2]);
            }
        }
        __syncthreads();
        if (wave == 0) {     // 3 axes x 15 planes, one candidate per lane
            float cost = kFInf; uint32_t nl = 0;
            const uint32_t ax = lane >> 4, plane = lane & 15u;
            if (lane < 48 && plane < kBins - 1) {
                float llo[3] = {kFInf, kFInf, kFInf}, lhi[3] = {-kFInf, -kFInf, -kFInf}, rlo[3] = {kFInf, kFInf, kFInf}, rhi[3] = {-kFInf, -kFInf, -kFInf};
                uint32_t nr = 0;
                for (uint32_t bi = 0; bi < (uint32_t)kBins; bi++) {
                    const uint32_t *bin = &L.bins[(ax * kBins + bi) * 7];
                    const uint32_t cnt = bin[0];
                    if (cnt == 0) continue;
                    if (bi <= plane) { nl += cnt; for (int d = 0; d < 3; d++) { llo[d] = fminf(llo[d], dec_f(bin[1 + d])); lhi[d] = fmaxf(lhi[d], dec_f(bin[4 + d])); } }
                    else { nr += cnt; for (int d = 0; d < 3; d++) { rlo[d] = fminf(rlo[d], dec_f(bin[1 + d])); rhi[d] = fmaxf(rhi[d], dec_f(bin[4 + d])); } }
                }
                if (nl != 0 && nr != 0) {
                    cost = half_area3(llo, lhi) * (float)nl + half_area3(rlo, rhi) * (float)nr;
                    if (!(cost < kFInf)) cost = kFInf;
                }
            }
            unsigned long long key = ((unsigned long long)__float_as_uint(cost) << 32) | lane;
            for (int o = 32; o > 0; o >>= 1) { unsigned long long other = __shfl_xor(key, o); key = other < key ? other : key; }
            const uint32_t bl = (uint32_t)key & 63u;
            const bool valid = __uint_as_float((uint32_t)(key >> 32)) < kFInf;
            if (lane == bl) { L.best_axis = valid ? ax : 3u; L.best_plane = plane; L.best_nl = valid ? nl : k / 2u; }
        }
        __syncthreads();
        const uint32_t bax = L.best_axis, bplane = L.best_plane, nl = L.best_nl;
        if (bax < 3u) {      // stable partition of idx[a, b) (no valid plane: all centroids coincide, split the order in half)
            const float cm = bax == 0 ? cmin[0] : bax == 1 ? cmin[1] : cmin[2], sc = bax == 0 ? scale[0] : bax == 1 ? scale[1] : scale[2];
            uint32_t left_before = 0;
            for (uint32_t base = a; base < b; base += kSahThreads) {
                const uint32_t i = base + tid;
                const bool valid = i < b;
                const uint32_t it = valid ? L.idx[i] : 0u;
                const float c = 0.5f * L.box[bax][it] + 0.5f * L.box[3 + bax][it];
                const bool left = valid && (uint32_t)bin_of(c, cm, sc) <= bplane;
                const unsigned long long ml = __ballot(left);
                if (lane == 0) L.wave_cnt[wave] = (uint32_t)__popcll(ml);
                __syncthreads();
                uint32_t before = left_before, chunk = 0;
                for (uint32_t w = 0; w < kSahWaves; w++) { const uint32_t cw = L.wave_cnt[w]; if (w < wave) before += cw; chunk += cw; }
                const uint32_t lrank = before + (uint32_t)__popcll(ml & ((1ull << lane) - 1ull));
                if (valid) L.tmp[left ? a + lrank : a + nl + ((i - a) - lrank)] = (uint16_t)it;
                left_before += chunk;
                __syncthreads();
            }
            for (uint32_t i = a + tid; i < b; i += kSahThreads) L.idx[i] = L.tmp[i];
        }
        __syncthreads();
        if (tid == 0) {
            float nlo[3] = {dec_f(L.red[6]), dec_f(L.red[7]), dec_f(L.red[8])}, nhi[3] = {dec_f(L.red[9]), dec_f(L.red[10]), dec_f(L.red[11])};
            put_desc(g, group, pair_base, e.y, nlo, nhi, pair_base + p, 0u);
            uint32_t t = top - 1, ns = L.n_small;
            const uint2 kids[2] = {make_ent(a, a + nl, p + 1u, p, 0u), make_ent(a + nl, b, p + nl, p, 1u)};
            for (int c = 0; c < 2; c++) {
                const uint32_t kc = (kids[c].x >> 16) - (kids[c].x & 0xFFFFu);
                if (kc > kWaveNode) L.ent[t++] = kids[c]; else { L.ent[2 * kGroup - 1 - ns] = kids[c]; ns++; }
            }
            L.top = t; L.n_small = ns;
        }
    }
    // ---- small subtrees: one wave each ----
    const uint32_t n_small = L.n_small;
    for (;;) {
        uint32_t i = 0;
        if (lane == 0) i = atomicAdd(&L.cursor, 1u);
        i = (uint32_t)__builtin_amdgcn_readfirstlane((int)i);
        if (i >= n_small) break;
        sah_wave_subtree(g, L, group, gs, pair_base, L.ent[2 * kGroup - 1 - i], lane);
    }
    __syncthreads();
    if (g.level0) for (uint32_t i = tid; i < m; i += kSahThreads) g.perm[gs + i] = g.vals[gs + L.idx[i]];
    // ---- the wide collapse's table (bvh_wide.hip: c(n, i), choice words) for the m - 1 pairs this group has just written,
    // bottom-up inside the workgroup: thread t owns pair pair_base + t and computes it in the round after its in-group
    // children are done (rounds = height of the group's subtree; two barriers each).  Children outside the group are roots
    // of the tier below: their rows are in global memory since the previous launch.  As separate level-by-level launches
    // over the finished tree this took 21 launches of ~4.5 us plus the parent / depth / bucket passes that fed them.
    // The subtree's height travels with the eff table, so the depth limit of the pair tracers needs no pass of its own.
    if (kSahCanFuseDp && g.dp_cost != nullptr && m >= 2u) {
        const uint32_t np = m - 1u;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");      // the pairs were written by other threads of this workgroup
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        const bool mine = tid < np;
        float alo[3] = {0.0f, 0.0f, 0.0f}, ahi[3] = {0.0f, 0.0f, 0.0f}, blo[3] = {0.0f, 0.0f, 0.0f}, bhi[3] = {0.0f, 0.0f, 0.0f};
        uint32_t la = 0, lb = 0, ka = 0, kb = 0;      // k: 0 = costs nothing (leaf, hole), 1 = pair `l` of this group, 2 = root `l` of a group below
        if (mine) {
            const float4 *pp = g.pairs + 4 * (size_t)(pair_base + tid);
            const float4 a0 = pp[0], a1 = pp[1], b0 = pp[2], b1 = pp[3];
            alo[0] = a0.x; alo[1] = a0.y; alo[2] = a0.z; ahi[0] = a1.x; ahi[1] = a1.y; ahi[2] = a1.z;
            blo[0] = b0.x; blo[1] = b0.y; blo[2] = b0.z; bhi[0] = b1.x; bhi[1] = b1.y; bhi[2] = b1.z;
            la = __float_as_uint(a0.w); lb = __float_as_uint(b0.w);
            const uint32_t ca = __float_as_uint(a1.w), cb = __float_as_uint(b1.w);
            if (ca == 0u && la != 0u) { if (la >= pair_base && la - pair_base < np) { ka = 1u; la -= pair_base; } else if (la < g.n_pairs_total) ka = 2u; }
            if (cb == 0u && lb != 0u) { if (lb >= pair_base && lb - pair_base < np) { kb = 1u; lb -= pair_base; } else if (lb < g.n_pairs_total) kb = 2u; }
        }
        if (tid < kGroup) L.dp_done[tid] = 0;
        __syncthreads();
        bool done = !mine;
        for (;;) {
            const bool go = !done && (ka != 1u || L.dp_done[la] != 0) && (kb != 1u || L.dp_done[lb] != 0);
            if (go) {
                float cl[8], cr[8];
#pragma unroll
                for (int k = 1; k < 8; k++) {
                    cl[k] = ka == 1u ? L.dp[k - 1][la] : ka == 2u ? g.dp_cost[8 * (size_t)la + (k - 1)] : 0.0f;
                    cr[k] = kb == 1u ? L.dp[k - 1][lb] : kb == 2u ? g.dp_cost[8 * (size_t)lb + (k - 1)] : 0.0f;
                }
                const uint32_t wa = ka == 1u ? L.dp_w[la] : ka == 2u ? __float_as_uint(g.dp_cost[8 * (size_t)la + 7]) : 0u;
                const uint32_t wb = kb == 1u ? L.dp_w[lb] : kb == 2u ? __float_as_uint(g.dp_cost[8 * (size_t)lb + 7]) : 0u;
                float c[8]; uint32_t own; unsigned long long w;
                dp_combine(alo, ahi, blo, bhi, cl, cr, wa, wb, c, own, w);
                const uint32_t h = min(1u + max(wa >> 21, wb >> 21), 1023u);
                const uint32_t word = own | (h << 21);
#pragma unroll
                for (int k = 1; k < 8; k++) L.dp[k - 1][tid] = c[k];
                L.dp_w[tid] = word;
                g.dp_choice[pair_base + tid] = w;
                if (tid == 0) {      // the group's root: the row the tier above (or the host, for the height of pair 0) reads
                    float *row = g.dp_cost + 8 * (size_t)pair_base;
#pragma unroll
                    for (int k = 1; k < 8; k++) row[k - 1] = c[k];
                    row[7] = __uint_as_float(word);
                }
            }
            __syncthreads();
            if (go) { L.dp_done[tid] = 1; done = true; }
            if (!__syncthreads_or(go ? 1 : 0)) break;
        }
    }
}

// Leaf links were written as final sorted positions; turn them into slots of the kind's packed array.
__global__ void k_sah_patch_links(uint32_t n_halves, float4 *pairs, const uint32_t *leaf_link) {
    uint32_t h = blockIdx.x * blockDim.x + threadIdx.x;
    if (h >= n_halves) return;
    const uint32_t count = __float_as_uint(pairs[2 * (size_t)h + 1].w);
    if ((count & 0x0FFFFFFFu) == 0u) return;       // inner node, hole or the empty pad leaf
    float4 lo = pairs[2 * (size_t)h];
    lo.w = __uint_as_float(leaf_link[__float_as_uint(lo.w)]);
    pairs[2 * (size_t)h] = lo;
}

#define HIP_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { err = std::string(#expr) + ": " + hipGetErrorString(e_); return false; } } while (0)

static bool ensure_buf(DeviceBuffer &b, size_t bytes, std::string &err) {
    if (b.bytes >= bytes && b.ptr) return true;
    if (b.ptr) { hipError_t e = hipFree(b.ptr); (void)e; b.ptr = nullptr; b.bytes = 0; }
    size_t want = bytes < 256 ? 256 : bytes;
    HIP_TRY(hipMalloc(&b.ptr, want));
    b.bytes = want;
    return true;
}

// Rebuilds the topology over the Morton-sorted primitives `vals` (n >= 2) and writes all n - 1 sibling pairs (leaf
// links = positions in st.perm, see k_sah_patch_links) and the final primitive order st.perm.
bool build_sah(hipStream_t stream, const BuildInput &in, BuildState &st, const uint32_t *vals, std::string &err) {
    const uint32_t n = st.n_prims;
    const size_t cap = (size_t)n + 64;          // groups over all tiers (checked below)
    if (!ensure_buf(st.sah_flag, (size_t)n * 4, err) || !ensure_buf(st.sah_ipos[0], (size_t)n * 4, err) || !ensure_buf(st.sah_ipos[1], (size_t)n * 4, err)) return false;
    if (!ensure_buf(st.sah_starts, cap * 4, err) || !ensure_buf(st.sah_desc, cap * 2 * sizeof(float4), err) || !ensure_buf(st.perm, (size_t)n * 4, err)) return false;
    const bool fuse = st.fuse_dp && kSahCanFuseDp;
    st.dp_fused = false;
    if (fuse && (!ensure_buf(st.dp_cost, (size_t)st.n_pairs * 8 * 4, err) || !ensure_buf(st.dp_choice, (size_t)st.n_pairs * 8, err))) return false;
    size_t scan_bytes = 0;
    HIP_TRY(rocprim::inclusive_scan(nullptr, scan_bytes, (uint32_t *)st.sah_flag.ptr, (uint32_t *)st.sah_ipos[0].ptr, n, rocprim::plus<uint32_t>(), stream));
    if (!ensure_buf(st.temp, scan_bytes, err)) return false;

    struct Tier { uint32_t n_items, n_groups; size_t starts_off; };
    std::vector<Tier> tiers;
    uint32_t items = n, cur = 0;
    const uint32_t *ipos = nullptr;
    size_t used = 0;
    uint32_t *starts = (uint32_t *)st.sah_starts.ptr;
    while (items > kGroup) {
        if (tiers.size() >= 24) { err = "SAH refinement: too many tiers"; return false; }
        uint32_t *next = (uint32_t *)st.sah_ipos[cur].ptr;
        HIP_TRY(hipMemsetAsync(st.sah_flag.ptr, 0, (size_t)n * 4, stream));
        hipLaunchKernelGGL(k_sah_flags, dim3((2 * n - 1 + 255) / 256), dim3(256), 0, stream, n, kGroup, (const uint32_t *)st.range_lo.ptr,
                           (const uint32_t *)st.range_hi.ptr, (const uint32_t *)st.parent.ptr, ipos, (uint32_t *)st.sah_flag.ptr);
        size_t tb = st.temp.bytes;
        HIP_TRY(rocprim::inclusive_scan(st.temp.ptr, tb, (uint32_t *)st.sah_flag.ptr, next, n, rocprim::plus<uint32_t>(), stream));
        uint32_t groups = 0;
        HIP_TRY(hipMemcpyAsync(&groups, next + (n - 1), 4, hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        if (groups == 0 || groups >= items) { err = "SAH refinement: clustering made no progress"; return false; }
        if (used + groups + 1 > cap) { err = "SAH refinement: tier table overflow"; return false; }
        hipLaunchKernelGGL(k_sah_starts, dim3((n + 255) / 256), dim3(256), 0, stream, n, (const uint32_t *)st.sah_flag.ptr, (const uint32_t *)next, ipos, starts + used);
        tiers.push_back({items, groups, used});
        used += groups;
        items = groups; ipos = next; cur ^= 1u;
    }
    HIP_TRY(hipMemsetAsync(starts + used, 0, 4, stream));      // last tier: one group that starts at item 0
    tiers.push_back({items, 1u, used});

    // pair bases: the last tier owns pairs [0, n_items - 1); tier t's groups follow the pairs of tier t + 1
    std::vector<uint32_t> base(tiers.size(), 0u);
    for (size_t t = tiers.size() - 1; t-- > 0;) base[t] = base[t + 1] + tiers[t + 1].n_items - tiers[t + 1].n_groups;
    HIP_TRY(hipMemsetAsync(st.pairs.ptr, 0, (size_t)st.n_pairs * 4 * sizeof(float4), stream));
    float4 *desc = (float4 *)st.sah_desc.ptr;
    for (size_t t = 0; t < tiers.size(); t++) {
        SahGroupArgs g;
        g.starts = starts + tiers[t].starts_off; g.n_groups = tiers[t].n_groups; g.n_items = tiers[t].n_items; g.pair_base = base[t];
        g.level0 = t == 0 ? 1u : 0u;
        g.vals = vals; g.plo = (const float4 *)st.plo.ptr; g.phi = (const float4 *)st.phi.ptr; g.n_spheres = in.n_spheres; g.n_quads = in.n_quads;
        g.desc_in = t == 0 ? nullptr : desc + 2 * tiers[t - 1].starts_off;
        g.desc_out = t + 1 == tiers.size() ? nullptr : desc + 2 * tiers[t].starts_off;
        g.pairs = (float4 *)st.pairs.ptr; g.perm = (uint32_t *)st.perm.ptr;
        g.max_leaf = st.max_leaf < 1u ? 1u : st.max_leaf > 8u ? 8u : st.max_leaf; g.c_isect = st.c_isect; g.c_trav = st.c_trav;
        g.dp_cost = fuse ? (float *)st.dp_cost.ptr : nullptr; g.dp_choice = fuse ? (unsigned long long *)st.dp_choice.ptr : nullptr; g.n_pairs_total = st.n_pairs;
        hipLaunchKernelGGL(k_sah_group, dim3(g.n_groups), dim3(kSahThreads), 0, stream, g);
    }
    HIP_TRY(hipGetLastError());
    st.sah_tiers = (uint32_t)tiers.size();
    st.dp_fused = fuse;
    return true;
}

// Depth of the rebuilt pair tree.  The Karras LBVH is at most 95 levels deep (63 Morton bits + 32 index bits), which is
// what the traversal stacks are sized for (intersect.h: kLdsStack + kSpillStack); the SAH rebuild has no such bound (a
// sweep can peel one item per level, in every tier), so it is measured and the build refuses a tree the pair tracers
// could not walk.  The same pass buckets the pairs by depth for the wide collapse's bottom-up table (bvh_wide.hip):
// parent[] of a pair = the pair that links to it (a real inner child never links to pair 0: the root is nobody's
// child, so link 0 marks the holes left by multi-primitive leaves); depth = pairs on the way up to the root; one
// histogram over depths, aggregated per workgroup in LDS (a handful of bins: per-thread global atomics on them took
// 0.5 ms for 225 k pairs).
__global__ void k_pair_parents(uint32_t n_halves, const float4 *pairs, uint32_t *parent) {
    const uint32_t h = blockIdx.x * blockDim.x + threadIdx.x;
    if (h >= n_halves) return;
    const uint32_t count = __float_as_uint(pairs[2 * (size_t)h + 1].w), link = __float_as_uint(pairs[2 * (size_t)h].w);
    if (count == 0u && link != 0u && link < n_halves / 2u) parent[link] = h >> 1;
}
__global__ __launch_bounds__(256) void k_pair_depth(uint32_t n_pairs, const uint32_t *parent, uint32_t *depth, uint32_t *hist) {
    __shared__ uint32_t s_hist[kPairLevels];
    for (uint32_t i = threadIdx.x; i < kPairLevels; i += blockDim.x) s_hist[i] = 0u;
    __syncthreads();
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < n_pairs) {
        uint32_t d = 1, q = parent[p];                      // root: 1
        if (p != 0u) {
            while (q != 0xFFFFFFFFu && q != 0u && d < kPairLevels) { d++; q = parent[q]; }
            d = q == 0u ? d + 1u : 0u;                      // met the root: count it; walked off the top: a hole, not part of the tree
        }
        if (d >= kPairLevels) d = kPairLevels - 1u;         // deeper than any tree the build accepts; keeps the table in bounds (and fails the depth check)
        depth[p] = d;
        if (d) atomicAdd(&s_hist[d], 1u);
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < kPairLevels; i += blockDim.x) if (s_hist[i]) atomicAdd(&hist[i], s_hist[i]);
}

// Launches the parent / depth / histogram pass over st.pairs and reads the histogram back (one synchronisation):
// st.level_hist[d] = pairs of depth d, st.pair_depth = the deepest level.
bool measure_pair_depths(hipStream_t stream, BuildState &st, std::string &err) {
    const uint32_t np = st.n_pairs, halves = 2 * np;
    if (!ensure_buf(st.dp_parent, (size_t)np * 4, err) || !ensure_buf(st.dp_depth, (size_t)np * 4, err) || !ensure_buf(st.dp_hist, 2 * kPairLevels * 4, err)) return false;
    HIP_TRY(hipMemsetAsync(st.dp_parent.ptr, 0xFF, (size_t)np * 4, stream));
    HIP_TRY(hipMemsetAsync(st.dp_hist.ptr, 0, 2 * kPairLevels * 4, stream));
    hipLaunchKernelGGL(k_pair_parents, dim3((halves + 255) / 256), dim3(256), 0, stream, halves, (const float4 *)st.pairs.ptr, (uint32_t *)st.dp_parent.ptr);
    hipLaunchKernelGGL(k_pair_depth, dim3((np + 255) / 256), dim3(256), 0, stream, np, (const uint32_t *)st.dp_parent.ptr, (uint32_t *)st.dp_depth.ptr, (uint32_t *)st.dp_hist.ptr);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(st.level_hist, st.dp_hist.ptr, kPairLevels * 4, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    st.pair_depth = 0;
    for (uint32_t d = 1; d < kPairLevels; d++) if (st.level_hist[d]) st.pair_depth = d;
    st.levels_measured = true;
    return true;
}

uint32_t pair_depth_limit() {
    uint32_t limit = kMaxPairDepth;
    if (const char *e = getenv("CRT_BVH_MAX_PAIR_DEPTH")) { const int v = atoi(e); if (v >= 1 && (uint32_t)v < limit) limit = (uint32_t)v; }   // test hook: can only lower it
    return limit;
}

bool patch_sah_links(hipStream_t stream, BuildState &st, std::string &err) {
    const uint32_t halves = 2 * st.n_pairs;
    hipLaunchKernelGGL(k_sah_patch_links, dim3((halves + 255) / 256), dim3(256), 0, stream, halves, (float4 *)st.pairs.ptr, (const uint32_t *)st.leaf_link.ptr);
    HIP_TRY(hipGetLastError());
    if (st.dp_fused) return true;      // the collapse table is done and carries the tree's height: build_wide checks it at its own synchronisation
    if (!measure_pair_depths(stream, st, err)) return false;
    const uint32_t limit = pair_depth_limit();
    if (st.pair_depth > limit) {
        err = "SAH-refined tree is " + std::to_string(st.pair_depth) + " levels deep; the pair traversal stacks hold " + std::to_string(limit) + " (a scene with very many coincident primitives: build it without CRT_BVH_SAH_REFINE, the Morton tree is at most 95 levels deep)";
        return false;
    }
    return true;
}

}  // namespace crt
