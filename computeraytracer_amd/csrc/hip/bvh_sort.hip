// bvh_sort.hip -- least-significant-digit radix sort of 64-bit keys (keys only) for the per-frame rebuild of large scenes:
// the Morton keys of CRT_BVH_WIDE_DIRECT are code << idx_bits | primitive index, sorted over the code's bits.  The
// reference sorts with std::sort inside its recursive builder (raytracing/src/bounding_volume_hierarchy.cpp:26-61); the
// device build needs one global sort per frame, and at 10 M keys it was a quarter of the rebuild with rocPRIM's
// onesweep (5 passes at 2.5 TB/s plus a dozen small fills).
//
// Single-pass-per-digit design ("onesweep", Adinets & Merrill 2022), written for wave64:
//   k_rs_histogram   one read of the keys: the global histogram of EVERY digit position (LDS atomics per block, one
//                    global atomic per block, digit and bin), then k_rs_scan makes them exclusive.
//   k_rs_pass        one launch per digit.  A block takes the next 4096-key tile from a ticket counter (so tile ids
//                    follow start order and a tile only ever waits for blocks that are already running), ranks its keys
//                    stably (per wave: lanes with equal digits find each other with eight ballots; per-wave digit
//                    counters in LDS give the rank among earlier steps), turns the tile's digit counts into exclusive
//                    prefixes over all earlier tiles by decoupled look-back (one 32-bit status word per tile and digit:
//                    value | flag, written and read with agent-scope relaxed atomics, so no fence is needed: the word
//                    is its own message), reorders the tile in LDS and writes every digit's run to its place.
// 8 bytes read + 8 written per key and pass, nothing else of size n.  Stable, so passes compose.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <string>

#include "build_internal.h"

namespace crt {

constexpr uint32_t kRsItemsLarge = 16, kRsItemsSmall = 4;      // keys per thread: 4096-key tiles, 1024-key tiles up to a million keys (a
                                                                // 225 k-key sort is 55 tiles of 4096: a fifth of the chip, each ranking 16 steps in a row)
constexpr uint32_t kRsSmallMax = 1u << 20;
constexpr uint32_t kRsBins = 256;
constexpr uint32_t kRsMaxPasses = 8;
constexpr uint32_t kFlagAgg = 1u << 30, kFlagPrefix = 2u << 30, kValueMask = (1u << 30) - 1u;

// hist[pass][bin]: one block histogram in LDS per digit position, flushed with one atomic per non-zero bin
__global__ __launch_bounds__(256) void k_rs_histogram(const uint64_t *keys, uint32_t n, uint32_t bit_lo, uint32_t n_pass, uint32_t last_bits, uint32_t *hist) {
    __shared__ uint32_t s_h[kRsMaxPasses][kRsBins];
    for (uint32_t i = threadIdx.x; i < kRsMaxPasses * kRsBins; i += 256) (&s_h[0][0])[i] = 0;
    __syncthreads();
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const uint64_t k = keys[i] >> bit_lo;
        for (uint32_t p = 0; p < n_pass; p++) {
            uint32_t d = (uint32_t)(k >> (8u * p)) & 255u;
            if (p + 1 == n_pass) d &= (1u << last_bits) - 1u;
            atomicAdd(&s_h[p][d], 1u);
        }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n_pass * kRsBins; i += 256) { const uint32_t v = (&s_h[0][0])[i]; if (v) atomicAdd(&hist[i], v); }
}

// exclusive scan of every pass's 256 bins (one block per pass)
__global__ __launch_bounds__(256) void k_rs_scan(uint32_t *hist) {
    __shared__ uint32_t s[kRsBins];
    uint32_t *h = hist + blockIdx.x * kRsBins;
    const uint32_t v = h[threadIdx.x];
    s[threadIdx.x] = v;
    __syncthreads();
    for (uint32_t off = 1; off < kRsBins; off <<= 1) {
        const uint32_t a = threadIdx.x >= off ? s[threadIdx.x - off] : 0u;
        __syncthreads();
        s[threadIdx.x] += a;
        __syncthreads();
    }
    h[threadIdx.x] = s[threadIdx.x] - v;
}

// PAIRS: a 32-bit value travels with every key (vin -> vout).
template <uint32_t kRsItems, bool PAIRS>
__global__ __launch_bounds__(256) void k_rs_pass(const uint64_t *in, uint64_t *out, const uint32_t *vin, uint32_t *vout, uint32_t n, uint32_t shift, uint32_t digit_mask,
                                                 const uint32_t *gbase, uint32_t *status, uint32_t *ticket, uint32_t *error_flag) {
    constexpr uint32_t kRsTile = 256 * kRsItems;
    __shared__ uint64_t s_keys[kRsTile];              // the tile in sorted order (32 KiB at 16 keys per thread)
    __shared__ uint32_t s_vals[PAIRS ? kRsTile : 1];
    __shared__ uint32_t s_wave[4][kRsBins];           // per-wave digit counters, then per-wave exclusive offsets
    __shared__ uint32_t s_start[kRsBins];             // end (exclusive) of a digit's run inside the sorted tile
    __shared__ uint32_t s_scan[kRsBins];
    __shared__ uint32_t s_gpos[kRsBins];              // where the tile's run of a digit starts in `out`
    __shared__ uint32_t s_tile;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_tile = atomicAdd(ticket, 1u);
    for (uint32_t i = threadIdx.x; i < 4 * kRsBins; i += 256) (&s_wave[0][0])[i] = 0;
    __syncthreads();
    const uint32_t tile = s_tile;
    const uint32_t base = tile * kRsTile;
    if (base >= n) return;                            // (grid = number of tiles: does not happen)
    const uint32_t cnt = min(kRsTile, n - base);
    // ---- load + rank: wave w owns keys [w * 1024, w * 1024 + 1024) of the tile, step i covers 64 consecutive keys
    uint64_t key[kRsItems];
    uint32_t val[kRsItems];
    uint32_t rank[kRsItems];
    const uint32_t wbase = wave * 64u * kRsItems;
    const unsigned long long lt = (1ull << lane) - 1ull;
#pragma unroll
    for (uint32_t i = 0; i < kRsItems; i++) {
        const uint32_t idx = wbase + i * 64u + lane;
        key[i] = idx < cnt ? in[base + idx] : ~0ull;
        val[i] = (PAIRS && idx < cnt) ? vin[base + idx] : 0u;
    }
#pragma unroll
    for (uint32_t i = 0; i < kRsItems; i++) {
        const uint32_t idx = wbase + i * 64u + lane;
        const bool valid = idx < cnt;
        const uint32_t d = (uint32_t)(key[i] >> shift) & digit_mask;
        // lanes of this step with the same digit
        unsigned long long peers = __ballot(valid);
#pragma unroll
        for (uint32_t b = 0; b < 8; b++) {
            const unsigned long long m = __ballot((d >> b) & 1u);
            peers &= ((d >> b) & 1u) ? m : ~m;
        }
        const uint32_t leader = (uint32_t)__builtin_ctzll(peers | (1ull << 63));
        uint32_t old = 0;
        if (valid && lane == leader) { old = s_wave[wave][d]; s_wave[wave][d] = old + (uint32_t)__popcll(peers); }
        old = (uint32_t)__shfl((int)old, (int)leader);
        rank[i] = old + (uint32_t)__popcll(peers & lt);
    }
    __syncthreads();
    // ---- per digit (thread = digit): offsets of the waves inside the tile's run, tile count, look-back
    {
        const uint32_t d = threadIdx.x;
        const uint32_t c0 = s_wave[0][d], c1 = s_wave[1][d], c2 = s_wave[2][d], c3 = s_wave[3][d];
        const uint32_t total = c0 + c1 + c2 + c3;
        s_wave[0][d] = 0; s_wave[1][d] = c0; s_wave[2][d] = c0 + c1; s_wave[3][d] = c0 + c1 + c2;
        uint32_t *st = status + (size_t)tile * kRsBins + d;
        uint32_t excl = 0;
        if (tile == 0) __hip_atomic_store(st, total | kFlagPrefix, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else {
            __hip_atomic_store(st, total | kFlagAgg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // earlier tiles belong to blocks that took their ticket before this one: they are running and publish without
            // waiting for anybody later.  The spin is bounded all the same: a wedged GPU is worse than a failed build.
            uint32_t j = tile;
            uint32_t spins = 0;
            while (j > 0) {
                const uint32_t v = __hip_atomic_load(status + (size_t)(j - 1) * kRsBins + d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (v & kFlagPrefix) { excl += v & kValueMask; break; }
                if (v & kFlagAgg) { excl += v & kValueMask; j--; spins = 0; continue; }
                if (++spins > (1u << 24)) { atomicOr(error_flag, 1u); break; }
                __builtin_amdgcn_s_sleep(1);
            }
            __hip_atomic_store(st, ((excl + total) & kValueMask) | kFlagPrefix, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        s_gpos[d] = gbase[d] + excl;
        // inclusive scan of the tile's digit counts: within the wave here, across the four waves below
        uint32_t incl = total;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const uint32_t t = (uint32_t)__shfl_up((int)incl, off); if ((int)lane >= off) incl += t; }
        s_scan[d] = incl;
    }
    __syncthreads();
    {
        const uint32_t d = threadIdx.x;
        uint32_t add = 0;
        for (uint32_t w = 0; w < wave; w++) add += s_scan[w * 64u + 63u];
        s_start[d] = s_scan[d] + add;      // INCLUSIVE end of digit d's run in the sorted tile; its start is the end of digit d - 1
    }
    __syncthreads();
    // ---- reorder in LDS
#pragma unroll
    for (uint32_t i = 0; i < kRsItems; i++) {
        const uint32_t idx = wbase + i * 64u + lane;
        if (idx < cnt) {
            const uint32_t d = (uint32_t)(key[i] >> shift) & digit_mask;
            const uint32_t run_start = d ? s_start[d - 1] : 0u;
            const uint32_t pos = run_start + s_wave[wave][d] + rank[i];
            s_keys[pos] = key[i];
            if (PAIRS) s_vals[pos] = val[i];
        }
    }
    __syncthreads();
    // ---- write: position p of the sorted tile belongs to digit d(p); consecutive threads write consecutive addresses
#pragma unroll
    for (uint32_t i = 0; i < kRsItems; i++) {
        const uint32_t p = i * 256u + threadIdx.x;
        if (p < cnt) {
            const uint64_t k = s_keys[p];
            const uint32_t d = (uint32_t)(k >> shift) & digit_mask;
            const uint32_t run_start = d ? s_start[d - 1] : 0u;
            out[s_gpos[d] + (p - run_start)] = k;
            if (PAIRS) vout[s_gpos[d] + (p - run_start)] = s_vals[p];
        }
    }
}

#define HIP_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { err = std::string(#expr) + ": " + hipGetErrorString(e_); return false; } } while (0)

static uint32_t rs_tile(uint32_t n) { return 256u * (n <= kRsSmallMax ? kRsItemsSmall : kRsItemsLarge); }

size_t radix_sort_scratch_bytes(uint32_t n) {
    const size_t tiles = ((size_t)n + rs_tile(n) - 1) / rs_tile(n);
    // per pass: status words of every tile; then the histograms, the tickets and the error word
    return kRsMaxPasses * tiles * kRsBins * 4 + kRsMaxPasses * kRsBins * 4 + 256;
}

// Sorts n keys over bits [bit_lo, bit_hi); keys_a (and vals_a, if values travel along: vals_a / vals_b both non-null) hold
// the input, the result is in *sorted (keys_a or keys_b) and *sorted_vals (vals_a or vals_b).
// scratch: radix_sort_scratch_bytes(n) bytes.  Enqueue-only.  error_flag (inside scratch, last word) is set if a look-back
// spin gave up; the caller may read it at its next synchronisation point.
bool radix_sort_u64(hipStream_t stream, uint64_t *keys_a, uint64_t *keys_b, uint32_t *vals_a, uint32_t *vals_b, uint32_t n, uint32_t bit_lo, uint32_t bit_hi,
                    void *scratch, const uint64_t **sorted, const uint32_t **sorted_vals, const uint32_t **error_word, std::string &err) {
    *sorted = keys_a;
    if (sorted_vals) *sorted_vals = vals_a;
    *error_word = nullptr;
    if (n < 2 || bit_hi <= bit_lo) return true;
    const bool pairs = vals_a != nullptr && vals_b != nullptr;
    const uint32_t bits = bit_hi - bit_lo;
    const uint32_t n_pass = (bits + 7u) / 8u;
    if (n_pass > kRsMaxPasses) { err = "radix sort: more than 64 key bits"; return false; }
    if (n >= (1u << 30)) { err = "radix sort: fewer than 2^30 keys"; return false; }
    const uint32_t last_bits = bits - 8u * (n_pass - 1u);
    const uint32_t tile = rs_tile(n), tiles = (n + tile - 1u) / tile;
    uint32_t *status = (uint32_t *)scratch;
    uint32_t *hist = status + (size_t)kRsMaxPasses * tiles * kRsBins;
    uint32_t *ticket = hist + kRsMaxPasses * kRsBins;      // [0..7] tickets, [8] error flag
    // status words of the passes that run, the histograms, the tickets and the error word: one fill when they are adjacent
    if (n_pass == kRsMaxPasses) HIP_TRY(hipMemsetAsync(scratch, 0, radix_sort_scratch_bytes(n), stream));
    else {
        HIP_TRY(hipMemsetAsync(scratch, 0, (size_t)n_pass * tiles * kRsBins * 4, stream));
        HIP_TRY(hipMemsetAsync(hist, 0, kRsMaxPasses * kRsBins * 4 + 256, stream));
    }
    hipLaunchKernelGGL(k_rs_histogram, dim3(std::min<uint32_t>((n + 255u) / 256u, 2048u)), dim3(256), 0, stream, (const uint64_t *)keys_a, n, bit_lo, n_pass, last_bits, hist);
    hipLaunchKernelGGL(k_rs_scan, dim3(n_pass), dim3(256), 0, stream, hist);
    uint64_t *src = keys_a, *dst = keys_b;
    uint32_t *vsrc = vals_a, *vdst = vals_b;
    for (uint32_t p = 0; p < n_pass; p++) {
        const uint32_t mask = p + 1 == n_pass ? (1u << last_bits) - 1u : 255u;
        const uint32_t shift = bit_lo + 8u * p;
        const uint32_t *gb = hist + p * kRsBins;
        uint32_t *stt = status + (size_t)p * tiles * kRsBins;
#define CRT_RS_LAUNCH(ITEMS, PAIRS) hipLaunchKernelGGL((k_rs_pass<ITEMS, PAIRS>), dim3(tiles), dim3(256), 0, stream, (const uint64_t *)src, dst, (const uint32_t *)vsrc, vdst, n, \
                                                       shift, mask, gb, stt, ticket + p, ticket + 8)
        if (tile == 256u * kRsItemsSmall) { if (pairs) CRT_RS_LAUNCH(kRsItemsSmall, true); else CRT_RS_LAUNCH(kRsItemsSmall, false); }
        else { if (pairs) CRT_RS_LAUNCH(kRsItemsLarge, true); else CRT_RS_LAUNCH(kRsItemsLarge, false); }
#undef CRT_RS_LAUNCH
        std::swap(src, dst); std::swap(vsrc, vdst);
    }
    HIP_TRY(hipGetLastError());
    *sorted = src;
    if (sorted_vals) *sorted_vals = vsrc;
    *error_word = ticket + 8;
    return true;
}

bool radix_sort_keys_u64(hipStream_t stream, uint64_t *keys_a, uint64_t *keys_b, uint32_t n, uint32_t bit_lo, uint32_t bit_hi, void *scratch,
                         const uint64_t **sorted, const uint32_t **error_word, std::string &err) {
    return radix_sort_u64(stream, keys_a, keys_b, nullptr, nullptr, n, bit_lo, bit_hi, scratch, sorted, nullptr, error_word, err);
}

}  // namespace crt
