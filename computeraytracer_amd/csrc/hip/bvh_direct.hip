// bvh_direct.hip -- CRT_BVH_WIDE_DIRECT: the 8-wide quantised tree (node format: bvh_wide.hip / intersect_wide.h) built
// straight from the sorted Morton keys, for the per-frame rebuild of large animated scenes (SURVEY.md section 8(d),
// config 5).  No binary hierarchy, no refit of a pair tree, no collapse: the pair BVH of the other build modes
// (640 MB written and read back at 10 M triangles) never exists.
//
//   top-down  (k_dw_split, one launch per level): a node is a range of the sorted keys.  It is cut at the highest
//             differing key bit of its largest part (the LBVH's spatial-median split) until eight parts are there or
//             every part is a leaf; parts of at most `leaf_max` keys are leaves.  The slot of a part comes from the
//             sides it took in the cuts (slot bit a = high side of the first cut across axis a), which is the octant
//             rule the traversal's slot ^ ray-octant order assumes.  Inner parts are numbered in slot order and queued.
//   bottom-up (k_dw_emit, one launch per level, deepest first): a node packs the traversal records of its leaf
//             primitives (bvh_pack.h), takes their padded boxes and the exact boxes of its inner children, writes its own
//             exact box for its parent, quantises the children against it (same arithmetic as wide_emit) and writes
//             the 64-byte node and its eight links.
//
// The keys are code << idx_bits | primitive index (k_morton), unique and sorted by code; parts whose keys differ
// only in the index bits (equal codes) are cut in the middle.  A root-to-leaf path crosses at most one cut per key bit,
// so the tree has at most 64 levels (kWideMaxLevels, the traversal's stack bound).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <string>
#include <vector>

#include "build_internal.h"
#include "bvh_common.h"
#include "bvh_pack.h"
#include "bvh_wide_node.h"

namespace crt {

struct DwSplit {
    const uint64_t *keys; uint32_t idx_bits, leaf_max;
    uint32_t *level_count;
    float4 *wide; uint32_t *refs;       // refs: eight words per node in the links array (overwritten by k_dw_emit)
};

// First position in (b, e) whose key has `bit` set (keys[b] has it clear, keys[e - 1] has it set).  A group of LANES lanes
// works on one node and probes LANES positions per step: the searches over the millions of keys of a top-level node
// are chains of dependent loads, 23 deep with one probe per step, 4 deep with 64.
template <int LANES>
__device__ __forceinline__ uint32_t dw_find_cut(const uint64_t *keys, uint32_t b, uint32_t e, uint32_t bit, uint32_t gl, uint32_t gbase, uint32_t *status) {
    uint32_t lo = b, hi = e - 1u;
    // Both searches end after at most 32 steps on any input; the trip counter makes that a property of the code rather than
    // of the keys (unsorted keys or a probe group with inactive lanes would otherwise spin): status bit 2 and a cut in the
    // middle, which keeps the tree a tree; the host refuses it.
    uint32_t trips = 0;
    if (LANES == 1) {
        while (hi - lo > 1u && trips++ < 40u) { const uint32_t mid = (lo + hi) >> 1; if ((keys[mid] >> bit) & 1ull) hi = mid; else lo = mid; }
    } else {
        while (hi - lo > 1u && hi < e && trips++ < 40u) {
            const uint32_t step = (hi - lo + (uint32_t)LANES - 1u) / (uint32_t)LANES;
            const uint32_t p = min(hi, lo + (gl + 1u) * step);                        // the group's last probe is hi itself
            const bool set = ((keys[p] >> bit) & 1ull) != 0ull;
            unsigned long long m = __ballot(set) >> gbase;
            if (LANES < 64) m &= (1ull << LANES) - 1ull;
            if (m == 0ull) { trips = 99u; break; }                                    // keys[hi] does not have the bit: not a sorted range
            const uint32_t f = (uint32_t)__builtin_ctzll(m);
            const uint32_t nlo = f ? lo + f * step : lo;                              // probe f - 1 had the bit clear, so it lies below hi
            hi = min(hi, lo + (f + 1u) * step);
            lo = nlo;
        }
    }
    if (hi - lo > 1u || hi <= b || hi >= e) {
        if (gl == 0u) atomicOr(status, 4u);
        return (b + e) >> 1;
    }
    return hi;
}

// One node: range -> up to eight parts -> slots, header word, leaf references, queue entries of the inner parts.
// Every lane of the node's group computes the same values; lane gl == 0 writes.
// cap: this is the last level the host has launches for (enqueue-only rebuild, build_wide_direct): a part that would need
// another level is dropped (empty slot) and the overflow word is set; the host reads it at the next crt_sync and reports
// the tree as invalid (the hint it planned with is the depth of the previous build plus two, so this is a scene whose
// shape changed abruptly between two frames).
template <int LANES>
__device__ __forceinline__ void dw_split_node(const DwSplit &A, uint32_t level, uint32_t next_base, uint2 range, uint32_t id, uint2 *q_out, uint32_t gl, uint32_t gbase, bool cap) {
    uint32_t sb[8], se[8], sd[8];       // part k: keys [sb, se); sd: bits 0..2 side per axis, bits 3..5 "axis was cut"
#pragma unroll
    for (int k = 0; k < 8; k++) { sb[k] = 0; se[k] = 0; sd[k] = 0; }
    sb[0] = range.x; se[0] = range.y;
    int n = 1;
    const uint32_t size = range.y - range.x;
    if (size == 0u || range.y < range.x) {      // an entry nobody wrote: an empty node (every slot empty), and the build is refused
        if (gl == 0u) atomicOr(&A.level_count[kWideMaxLevels + 2], 16u);
        n = 0;
    } else if (size <= 8u * A.leaf_max) {
        // Bottom node: every part is a leaf, so where exactly the cuts fall matters little (neighbours in Morton order
        // share a leaf either way): equal runs of ceil(size / 8) keys, no searches.  The slot of a run comes from the three
        // key bits of its first key below the range's common prefix (its octant inside the range's cell).
        const uint32_t c = (size + 7u) / 8u;
        const uint64_t first = A.keys[range.x], last = A.keys[range.y - 1u];
        const uint64_t x = first ^ last;
        const uint32_t bit = x ? 63u - (uint32_t)__clzll((long long)x) : 0u;
        n = (int)((size + c - 1u) / c);
#pragma unroll
        for (int k = 0; k < 8; k++) {
            if (k >= n) continue;
            sb[k] = range.x + (uint32_t)k * c; se[k] = min(range.y, sb[k] + c);
            const uint64_t fk = A.keys[sb[k]];
            uint32_t d = 0;
#pragma unroll
            for (uint32_t j = 0; j < 3; j++) {
                if (!x || bit < A.idx_bits + j) continue;      // below the code: no spatial meaning
                const uint32_t bj = bit - j, ax = 2u - (bj - A.idx_bits) % 3u;
                d |= (8u << ax) | ((uint32_t)((fk >> bj) & 1ull) << ax);
            }
            sd[k] = d;
        }
    }
    while (n >= 1 && n < 8 && size > 8u * A.leaf_max && range.y > range.x) {
        uint32_t best = 0; int bk = -1;
#pragma unroll
        for (int k = 0; k < 8; k++) { const uint32_t sz = se[k] - sb[k]; if (k < n && sz > A.leaf_max && sz > best) { best = sz; bk = k; } }
        if (bk < 0) break;
        uint32_t b = sb[0], e = se[0], d = sd[0];
#pragma unroll
        for (int k = 1; k < 8; k++) if (k == bk) { b = sb[k]; e = se[k]; d = sd[k]; }
        const uint64_t first = A.keys[b], last = A.keys[e - 1];
        const uint32_t bit = first != last ? 63u - (uint32_t)__clzll((long long)(first ^ last)) : 0u;      // unique keys: first != last
        uint32_t m = (b + e) >> 1, dl = d, dr = d;
        if (first >= last && gl == 0u) atomicOr(&A.level_count[kWideMaxLevels + 2], 8u);                    // (not sorted, or not unique: cut in the middle)
        if (first < last && bit >= A.idx_bits) {
            m = dw_find_cut<LANES>(A.keys, b, e, bit, gl, gbase, &A.level_count[kWideMaxLevels + 2]);
            const uint32_t ax = 2u - (bit - A.idx_bits) % 3u;      // code = ... x y z x y z: bit j of the code belongs to axis 2 - j % 3
            if (!((d >> (3u + ax)) & 1u)) { dl = d | (8u << ax); dr = dl | (1u << ax); }
        }
#pragma unroll
        for (int k = 0; k < 8; k++) {
            if (k == bk) { se[k] = m; sd[k] = dl; }
            if (k == n) { sb[k] = m; se[k] = e; sd[k] = dr; }
        }
        n++;
    }
    // slots: parts in order, each takes the free slot that disagrees with its known sides on the fewest axes
    uint32_t slot_of = 0, used = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        if (k >= n) continue;
        const uint32_t side = sd[k] & 7u, known = sd[k] >> 3;
        uint32_t bs = 0, bc = 99;
#pragma unroll
        for (uint32_t sl = 0; sl < 8; sl++) {
            const uint32_t c = (uint32_t)__builtin_popcount((sl ^ side) & known) * 8u + (uint32_t)__builtin_popcount((sl ^ side) & 7u);
            if (!((used >> sl) & 1u) && c < bc) { bc = c; bs = sl; }
        }
        slot_of |= bs << (4 * k); used |= 1u << bs;
    }
    uint32_t imask = 0, n_inner = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) if (k < n && se[k] - sb[k] > A.leaf_max) { imask |= 1u << ((slot_of >> (4 * k)) & 7u); n_inner++; }
    if (cap && n_inner) {
        if (gl == 0u) atomicOr(&A.level_count[kWideMaxLevels + 2], 1u);
#pragma unroll
        for (int k = 0; k < 8; k++) if (k < n && se[k] - sb[k] > A.leaf_max) se[k] = sb[k];      // dropped: reads as an empty slot below
        imask = 0; n_inner = 0;
    }
    uint32_t qpos = 0;
    if (n_inner && gl == 0u) qpos = atomicAdd(&A.level_count[level + 1], n_inner);
    if (LANES > 1) qpos = (uint32_t)__shfl((int)qpos, (int)gbase);
    const uint32_t child_base = next_base + qpos;
    uint32_t ref[8];
#pragma unroll
    for (int sl = 0; sl < 8; sl++) ref[sl] = kWideEmptyLink;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        if (k >= n) continue;
        const uint32_t sl = (slot_of >> (4 * k)) & 7u, sz = se[k] - sb[k];
        uint32_t r;
        if (sz == 0u) continue;
        if (sz > A.leaf_max) {
            const uint32_t rank = (uint32_t)__builtin_popcount(imask & ((1u << sl) - 1u));      // inner children are numbered in slot order
            if (gl == 0u) q_out[qpos + rank] = make_uint2(sb[k], se[k]);
            r = child_base + rank;
        } else r = kWideLeafTag | ((sz - 1u) << 26) | sb[k];      // sorted position of the leaf's first primitive; kind and slot: k_dw_emit
#pragma unroll
        for (uint32_t s2 = 0; s2 < 8; s2++) ref[s2] = s2 == sl ? r : ref[s2];
    }
    if (gl != 0u) return;
    uint4 *o = (uint4 *)(A.refs + 8 * (size_t)id);
    o[0] = make_uint4(ref[0], ref[1], ref[2], ref[3]);
    o[1] = make_uint4(ref[4], ref[5], ref[6], ref[7]);
    ((uint32_t *)(A.wide + (size_t)kWideStrideF4 * id))[3] = (child_base << 8) | imask;
}

// level_count[l] = nodes of level l; level l reads queue l & 1 and appends to the other one.  Node ids are BFS: the
// nodes of level l are next_base - n_in + [0, n_in).
__global__ __launch_bounds__(256) void k_dw_split(DwSplit A, uint32_t level, const uint2 *q_in, uint2 *q_out, bool cap) {
    const uint32_t n_in = A.level_count[level];
    uint32_t next_base = 0;
    for (uint32_t l = 0; l <= level; l++) next_base += A.level_count[l];
    // A wave per node for the first levels (searches 64 probes wide: 18 instead of 43 us per level at 10 M keys), a lane
    // per node below: every lane of a group repeats the node's bookkeeping, which from a few thousand nodes on costs more
    // than the shorter searches save (8 lanes per node at 250 k nodes: 357 instead of 115 us).
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x, n_threads = gridDim.x * blockDim.x, lane = threadIdx.x & 63u;
    if (n_in <= 512u) {
        for (uint32_t i = tid >> 6; i < n_in; i += n_threads >> 6) dw_split_node<64>(A, level, next_base, q_in[i], next_base - n_in + i, q_out, lane, 0u, cap);
    } else {
        for (uint32_t i = tid; i < n_in; i += n_threads) dw_split_node<1>(A, level, next_base, q_in[i], next_base - n_in + i, q_out, 0u, 0u, cap);
    }
}

__global__ void k_dw_init(uint32_t *level_count, uint32_t n_levels, uint2 *q0, uint32_t n, const uint32_t *sort_error) {
    // [n_levels + 1] = the status word: bit 0 = tree deeper than planned (k_dw_split), bit 1 = the key sort gave up
    for (uint32_t i = threadIdx.x; i <= n_levels + 1u; i += blockDim.x) level_count[i] = i == 0 ? 1u : (i == n_levels + 1u && sort_error && *sort_error) ? 2u : 0u;
    if (threadIdx.x == 0) q0[0] = make_uint2(0u, n);
}

struct DwEmit {
    const uint64_t *keys; uint32_t idx_mask, n;
    BuildInput in; const uint32_t *rank_sphere, *rank_quad; const float4 *tri_shade; PackOut out;
    float4 *wide; uint4 *links; float4 *box;      // box: exact box of node i at [2 i], [2 i + 1]
};

// Eight lanes per node, one per slot: the leaf primitives of a node are consecutive sorted positions, so a group's
// gathers and record writes run side by side and coalesce; the node box and the packed plane bytes are reduced over the
// group with shuffles (wide_node_finish, bvh_wide_node.h).  The level's node count and first id come from the device's
// level table, and the groups are dealt grid-stride, so the host can launch this without knowing either.
__global__ __launch_bounds__(256) void k_dw_emit(DwEmit A, const uint32_t *level_count, uint32_t level) {
    const uint32_t count = level_count[level];
    uint32_t first_id = 0;
    for (uint32_t l = 0; l < level; l++) first_id += level_count[l];
    const uint32_t sl = threadIdx.x & 7u;
    for (uint32_t g0 = blockIdx.x * 32u; g0 < count; g0 += gridDim.x * 32u) {      // block-uniform trip count: whole groups are active or not
        const uint32_t grp = g0 + (threadIdx.x >> 3);
        const bool active = grp < count;
        const uint32_t id = first_id + (active ? grp : 0u);
        const uint32_t hdr = ((const uint32_t *)(A.wide + (size_t)kWideStrideF4 * id))[3];
        const uint32_t imask = hdr & 0xFFu, child_base = hdr >> 8;
        const uint32_t ref = ((const uint32_t *)A.links)[8 * (size_t)id + sl];
        float clo[3] = {3.402823466e38f, 3.402823466e38f, 3.402823466e38f}, chi[3] = {-3.402823466e38f, -3.402823466e38f, -3.402823466e38f};
        uint32_t link = ref;
        bool valid = false;
        if (active) {
            if ((imask >> sl) & 1u) {
                const uint32_t cid = child_base + (uint32_t)__builtin_popcount(imask & ((1u << sl) - 1u));
                const float4 a = A.box[2 * (size_t)cid], b = A.box[2 * (size_t)cid + 1];
                clo[0] = a.x; clo[1] = a.y; clo[2] = a.z; chi[0] = b.x; chi[1] = b.y; chi[2] = b.z;
                valid = true;
            } else if (ref != kWideEmptyLink) {
                const uint32_t s0 = ref & ((1u << 26) - 1u), cnt = ((ref >> 26) & 7u) + 1u;
                uint32_t kind0 = 0, slot0 = 0;
                for (uint32_t j = 0; j < cnt; j++) {
                    uint32_t kind, slot; float lo[3], hi[3];
                    pack_prim<true>(A.in, A.n, s0 + j, (uint32_t)A.keys[s0 + j] & A.idx_mask, A.rank_sphere, A.rank_quad, A.tri_shade, A.out, kind, slot, lo, hi);
                    if (j == 0) { kind0 = kind; slot0 = slot; }
#pragma unroll
                    for (int a = 0; a < 3; a++) { clo[a] = fminf(clo[a], lo[a]); chi[a] = fmaxf(chi[a], hi[a]); }
                }
                link = kWideLeafTag | (kind0 << 29) | ((cnt - 1u) << 26) | slot0;
                valid = true;
            }
        }
        wide_node_finish(active, sl, id, hdr, clo, chi, valid, A.wide, A.box, nullptr);
        if (active) ((uint32_t *)A.links)[8 * (size_t)id + sl] = link;
    }
}

#define HIP_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { err = std::string(#expr) + ": " + hipGetErrorString(e_); return false; } } while (0)

static bool ensure_d(DeviceBuffer &b, size_t bytes, std::string &err) {
    if (b.bytes >= bytes && b.ptr) return true;
    if (b.ptr) { hipError_t e = hipFree(b.ptr); (void)e; b.ptr = nullptr; b.bytes = 0; }
    size_t want = bytes < 256 ? 256 : bytes;
    HIP_TRY(hipMalloc(&b.ptr, want));
    b.bytes = want;
    return true;
}

bool resolve_wide_counts(BuildState &st, bool &overflow, std::string &err) {
    overflow = false;
    if (!st.counts_pending) return true;
    HIP_TRY(hipEventSynchronize(st.ev_counts));
    st.counts_pending = false;
    const uint32_t *h = st.host_lc;
    overflow = h[kWideMaxLevels + 2] != 0u;      // bit 0: deeper than planned; bit 1: the key sort gave up (as unusable, reported the same way)
    st.n_wide = 0; st.wide_levels = 0;
    for (uint32_t l = 0; l <= kWideMaxLevels && h[l]; l++) { st.n_wide += h[l]; st.wide_levels = l + 1; st.direct_hint_lc[l] = h[l]; }
    for (uint32_t l = st.wide_levels; l < kWideMaxLevels + 3; l++) st.direct_hint_lc[l] = 0;
    st.direct_hint_levels = overflow ? 0u : st.wide_levels;      // after an overflow the next build counts its levels the slow way
    return true;
}

bool build_wide_direct(hipStream_t stream, const BuildInput &in, BuildState &st, const uint64_t *keys, uint32_t idx_bits, const uint32_t *rank_sphere,
                       const uint32_t *rank_quad, const PackOut &out, const uint32_t *sort_error, std::string &err) {
    const uint32_t n = st.n_prims;
    st.n_wide = 0; st.wide_levels = 0;
    if (n == 0) return true;
    if (n >= (1u << 24)) { err = "wide BVH supports fewer than 2^24 primitives"; return false; }
    // every node has at least two children (or is the root of a scene that is one leaf), so n_pairs = max(n - 1, 1) bounds the node count
    if (!ensure_d(st.wide, (size_t)st.n_pairs * (kWideStrideF4 * sizeof(float4) + 32), err)) return false;
    st.wide_cap = st.n_pairs;
    uint4 *links = (uint4 *)((float4 *)st.wide.ptr + (size_t)kWideStrideF4 * st.wide_cap);
    if (!ensure_d(st.wide_queue[0], (size_t)st.n_pairs * 8, err) || !ensure_d(st.wide_queue[1], (size_t)st.n_pairs * 8, err)) return false;
    if (!ensure_d(st.wide_levels_dev, (kWideMaxLevels + 3) * 4, err)) return false;
    if (!ensure_d(st.nlo, (size_t)st.n_pairs * 2 * sizeof(float4), err)) return false;
    if (!st.host_lc) HIP_TRY(hipHostMalloc((void **)&st.host_lc, (kWideMaxLevels + 3) * 4, hipHostMallocDefault));
    if (!st.ev_counts) HIP_TRY(hipEventCreateWithFlags(&st.ev_counts, hipEventDisableTiming));
    uint32_t *lc = (uint32_t *)st.wide_levels_dev.ptr;
    const bool single_kind = in.n_spheres == n || in.n_quads == n || in.n_tris == n;
    static const uint32_t env_leaf = [] { const char *e = getenv("CRT_DW_LEAF"); int v = e ? atoi(e) : 2; return (uint32_t)(v < 1 ? 1 : v > 8 ? 8 : v); }();
    DwSplit S{keys, idx_bits, single_kind ? env_leaf : 1u, lc, (float4 *)st.wide.ptr, (uint32_t *)links};
    DwEmit E{keys, (uint32_t)((1ull << idx_bits) - 1ull), n, in, rank_sphere, rank_quad, (const float4 *)st.tri_shade.ptr, out,
             (float4 *)st.wide.ptr, links, (float4 *)st.nlo.ptr};
    hipLaunchKernelGGL(k_dw_init, dim3(1), dim3(128), 0, stream, lc, kWideMaxLevels + 1, (uint2 *)st.wide_queue[0].ptr, n, sort_error);
    const uint32_t grid = std::min<uint32_t>(2048u, (st.n_pairs + 255u) / 256u);
    // Enqueue-only: a rebuild of the scene the previous direct build saw (same primitive count and leaf size) replays
    // that build's launch plan two levels deeper, nothing is read back before the launches: crt_tris_animate ->
    // crt_bvh_build -> crt_dispatch is one stream of launches, the host never waits.  The level kernels take
    // their counts from the device's table; the last planned level cannot open another one (k_dw_split's cap).
    if (st.async_request && st.direct_hint_levels && st.direct_hint_n == n && st.direct_hint_leaf == S.leaf_max) {
        int slack = 2;
        if (const char *e = getenv("CRT_DW_PLAN_SLACK")) slack = atoi(e);      // test hook (read per build): a negative slack forces the overflow path
        const uint32_t L = (uint32_t)std::min<int>(std::max<int>((int)st.direct_hint_levels + slack, 1), (int)kWideMaxLevels);
        for (uint32_t level = 0; level < L; level++)
            hipLaunchKernelGGL(k_dw_split, dim3(grid), dim3(256), 0, stream, S, level, (const uint2 *)st.wide_queue[level & 1].ptr, (uint2 *)st.wide_queue[(level & 1) ^ 1].ptr, level + 1u == L);
        for (uint32_t l = L; l-- > 0;) {
            // grid from the previous build's level sizes with a quarter to spare; the kernel is grid-stride, so a level that
            // grew beyond that is still complete
            const uint32_t guess = st.direct_hint_lc[l] + st.direct_hint_lc[l] / 4u + 64u;
            hipLaunchKernelGGL(k_dw_emit, dim3(std::min<uint32_t>((guess + 31u) / 32u, 65536u)), dim3(256), 0, stream, E, (const uint32_t *)lc, l);
        }
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(st.host_lc, lc, (kWideMaxLevels + 3) * 4, hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipEventRecord(st.ev_counts, stream));
        st.counts_pending = true;
        st.n_wide = 0;
        for (uint32_t l = 0; l < kWideMaxLevels; l++) st.n_wide += st.direct_hint_lc[l];      // the previous build's count until this one's is known (only "non-zero" matters to the tracer)
        st.wide_levels = L;                                                                 // an upper bound: what the tracer sizes its stacks by
        return true;
    }
    uint32_t level = 0;
    uint32_t *host_lc = st.host_lc;
    for (;;) {
        // blind launches (an empty level costs ~5 us): the depth of a balanced tree plus two the first time, four more after a miss
        uint32_t expect = 2;
        for (uint64_t cap = 8ull * S.leaf_max; cap < n; cap *= 8ull) expect++;
        const uint32_t chunk_end = std::min<uint32_t>(level == 0 ? expect + 1u : level + 4u, kWideMaxLevels);
        for (; level < chunk_end; level++) {
            hipLaunchKernelGGL(k_dw_split, dim3(grid), dim3(256), 0, stream, S, level, (const uint2 *)st.wide_queue[level & 1].ptr, (uint2 *)st.wide_queue[(level & 1) ^ 1].ptr, false);
            build_trace(stream, "k_dw_split level", level);
        }
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(host_lc, lc, (kWideMaxLevels + 3) * 4, hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        if (host_lc[kWideMaxLevels + 2] & 2u) { err = "the key sort's look-back wait gave up"; return false; }
        if (host_lc[kWideMaxLevels + 2] & 28u) {      // 4: a cut search did not converge, 8: a range whose first key is not below its last, 16: an empty range in the queue
            err = "direct build: the sorted keys are not a strictly increasing sequence (status " + std::to_string(host_lc[kWideMaxLevels + 2]) + ")";
            std::vector<uint64_t> hk(n);      // error path only: name the first offending position
            if (hipMemcpy(hk.data(), keys, (size_t)n * 8, hipMemcpyDeviceToHost) == hipSuccess) {
                for (uint32_t i = 1; i < n; i++)
                    if (hk[i - 1] >= hk[i]) {
                        char buf[160];
                        snprintf(buf, sizeof buf, ": keys[%u] = %016llx, keys[%u] = %016llx, idx_bits %u", i - 1, (unsigned long long)hk[i - 1], i, (unsigned long long)hk[i], idx_bits);
                        err += buf;
                        break;
                    }
            }
            return false;
        }
        if (host_lc[level] == 0) break;
        if (level >= kWideMaxLevels) { err = "wide BVH deeper than 64 levels"; return false; }
    }
    for (uint32_t l = 0; l < kWideMaxLevels + 3; l++) st.direct_hint_lc[l] = 0;
    for (uint32_t l = 0; l <= kWideMaxLevels && host_lc[l]; l++) { st.n_wide += host_lc[l]; st.wide_levels = l + 1; st.direct_hint_lc[l] = host_lc[l]; }
    st.direct_hint_n = n; st.direct_hint_levels = st.wide_levels; st.direct_hint_leaf = S.leaf_max;
    st.counts_pending = false;
    for (uint32_t l = st.wide_levels; l-- > 0;) {
        hipLaunchKernelGGL(k_dw_emit, dim3(std::min<uint32_t>((host_lc[l] + 31u) / 32u, 65536u)), dim3(256), 0, stream, E, (const uint32_t *)lc, l);
        build_trace(stream, "k_dw_emit level", l);
    }
    HIP_TRY(hipGetLastError());
    return true;
}

}  // namespace crt
