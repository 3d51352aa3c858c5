// trace.hip -- the path-tracing kernels that replace one vkCmdDispatch of
// shaders/compute.comp (engine/src/vk_engine.cpp:1429-1435).
//
//   k_megakernel   one lane per pixel for the whole path (the shader's own shape): main() :460-477,
//                  ray_color() :421-458.  One wave = one 8x8 pixel tile, like the shader's 8x8 workgroup.
//   wavefront      generate / extend / shade+compact / accumulate kernels over compacted ray queues
//                  (see the block comment further down).
//
// Built with -ffp-contract=off; see device_math.h for the arithmetic contract.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <mutex>

#include "build_internal.h"
#include "shade.h"
#include "intersect_wide.h"

namespace crt {

struct Frame {
    crt_scene_params p;
    uint32_t width, height, grid_w, grid_h;
    uint32_t band_rank, band_world, n_local_bands, tiles_x, n_tiles;
    uint32_t max_bounces;
    uint64_t pcg_inc;
    // wavefront batches: n_batch consecutive iterations traced together; iteration b of the batch has
    // camera.iteration = iter0 + b and camera.seed = seeds[b]
    uint32_t n_batch;
    float iter0;
    float seeds[64];
};
constexpr uint32_t kBatchMax = 64;

static DeviceScene to_device_scene(const DeviceSceneHost &h) {
    DeviceScene d;
    d.nodes = (const float4 *)h.nodes; d.n_pairs = h.n_pairs;
    d.wide = (const float4 *)h.wide; d.wide_links = h.wide_links; d.n_wide = h.n_wide;
    d.tri_pack = (const float4 *)h.tri_pack; d.sphere_pack = (const float4 *)h.sphere_pack; d.quad_pack = (const float4 *)h.quad_pack;
    d.tri_shade = (const float4 *)h.tri_shade;
    d.sphere_ids = h.sphere_ids; d.quad_ids = h.quad_ids;
    d.spheres = h.spheres; d.quads = h.quads; d.tris = h.tris; d.n_tris = h.n_tris; d.materials = h.materials;
    d.tex0 = h.tex0; d.tex0_w = h.tex0_w; d.tex0_h = h.tex0_h; d.n_textures = h.n_textures;
    d.tex_table = h.tex_table; d.ext = h.ext;
    d.dbg_boxes = (const float4 *)h.dbg_boxes; d.n_dbg_boxes = h.n_dbg_boxes;
    return d;
}

static Frame make_frame(const FrameSetup &fs, const crt_scene_params &params) {
    Frame f;
    f.p = params;
    f.width = fs.width; f.height = fs.height; f.grid_w = fs.grid_w; f.grid_h = fs.grid_h;
    f.band_rank = fs.band_rank; f.band_world = fs.band_world; f.n_local_bands = fs.n_local_bands;
    f.tiles_x = (fs.grid_w + 7) / 8;
    f.n_tiles = f.tiles_x * fs.n_local_bands;
    f.max_bounces = fs.max_bounces;
    f.pcg_inc = ((uint64_t)(uint32_t)params.camera.iteration << 1) | 1u;
    f.n_batch = 1; f.iter0 = params.camera.iteration;
    for (uint32_t b = 0; b < kBatchMax; b++) f.seeds[b] = params.camera.seed;
    return f;
}

// Blocks are dealt round-robin over the 8 XCDs (each with its own L2).  Remap so that the blocks that
// share an XCD work on one contiguous run of tiles: neighbouring tiles walk the same BVH nodes.
__device__ __forceinline__ uint32_t xcd_remap(uint32_t bid, uint32_t nblocks) {
    uint32_t q = nblocks / 8, r = nblocks % 8, xcd = bid % 8, k = bid / 8;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
}

__device__ __forceinline__ unsigned long long wave_sum(unsigned long long v) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// Pixel of (tile, lane): x, y in image space; prow = row inside this context's packed band buffer.
__device__ __forceinline__ bool tile_pixel(const Frame &f, uint32_t tile, uint32_t lane, uint32_t &x, uint32_t &y, uint32_t &prow) {
    uint32_t lb = tile / f.tiles_x, tx = tile % f.tiles_x;
    uint32_t band = lb * f.band_world + f.band_rank;
    x = tx * 8 + (lane & 7);
    y = band * 8 + (lane >> 3);
    prow = lb * 8 + (lane >> 3);
    return x < f.grid_w && y < f.grid_h;
}

template <int RNGM, bool COUNT, bool EXT = false>
__global__ __launch_bounds__(256) void k_megakernel(DeviceScene sc, Frame f, float4 *accum, uint32_t *hit_ids, DeviceStats *stats) {
    __shared__ uint32_t s_stack[kLdsStack * 256];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t tile = xcd_remap(blockIdx.x, gridDim.x) * 4 + wave;
    uint32_t x = 0, y = 0, prow = 0;
    bool active = tile < f.n_tiles && tile_pixel(f, tile, lane, x, y, prow);
    unsigned long long rays = 0, nodes = 0, prims = 0;
    if (active) {
        const size_t pix = (size_t)prow * f.width + x;
        const crt_camera_data &cam = f.p.camera;
        Rng<RNGM> rng;
        rng.init(cam.seed, x, y, y * f.width + x, cam.iteration);
        // main(), compute.comp:460-477
        float4 cum4 = accum[pix];
        float ox, oy; hash_2(rng, ox, oy);
        float u = ((float)x + ox) / (float)f.width, v = ((float)y + oy) / (float)f.height;
        V3 o, d;
        camera_get_ray(cam, u, v, rng, o, d);
        // ray_color(), :438-457
        V3 color = mk3(1.0f, 1.0f, 1.0f), emitted = mk3(0.0f, 0.0f, 0.0f), pass = mk3(0.0f, 0.0f, 0.0f);
        uint32_t primary = kIdMiss;
        const bool outline = cam.shouldRenderAABB != 0u && aabb_outline(sc, o, d);       // R15
        if (outline) pass = mk3(1.0f, 1.0f, 1.0f);
        for (uint32_t depth = 0; depth < f.max_bounces && !outline; depth++) {
            float t; uint32_t id; Counts cnt = {0, 0};
            closest_hit<COUNT, EXT>(sc, o, d, s_stack + threadIdx.x, 256, t, id, cnt);
            rays++;
            if (COUNT) { nodes += cnt.nodes; prims += cnt.prims; }
            if (depth == 0) primary = id;
            if (id == kIdMiss) { pass = shade_miss(f.p, d, color); break; }
            if (!shade_hit<RNGM, EXT>(sc, o, d, t, id, rng, color, emitted, pass)) break;
        }
        pass = pass * (1.0f / 1.0f);
        V3 cum = accumulate(mk3(cum4.x, cum4.y, cum4.z), pass, cam.iteration);
        accum[pix] = make_float4(cum.x, cum.y, cum.z, cam.iteration);
        hit_ids[pix] = primary;
    }
    rays = wave_sum(rays);
    if (COUNT) { nodes = wave_sum(nodes); prims = wave_sum(prims); }
    const unsigned long long n_active = (unsigned long long)__popcll(__ballot(active));
    if (lane == 0 && tile < f.n_tiles) {
        atomicAdd(&stats->rays, rays);
        atomicAdd(&stats->paths, n_active);
        if (COUNT) { atomicAdd(&stats->node_visits, nodes); atomicAdd(&stats->prim_tests, prims); }
    }
}

#define HIP_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { err = std::string(#expr) + ": " + hipGetErrorString(e_); return false; } } while (0)

bool launch_megakernel(hipStream_t stream, const DeviceSceneHost &sch, const FrameSetup &fs, const crt_scene_params &params,
                       float *accum, uint32_t *hit_ids, DeviceStats *stats, std::string &err) {
    DeviceScene sc = to_device_scene(sch);
    Frame f = make_frame(fs, params);
    if (f.n_tiles == 0) return true;
    dim3 grid((f.n_tiles + 3) / 4), block(256);
    float4 *acc = (float4 *)accum;
    // (rng, instrumented, extensions): the extensions' code exists only in separate kernels (intersect.h test_leaf)
    const int variant = (fs.rng_mode == CRT_RNG_REF ? 0 : 4) | (fs.instrumented ? 2 : 0) | (sc.ext ? 1 : 0);
    switch (variant) {
#define CRT_MEGA(V, R, C, A) case V: hipLaunchKernelGGL((k_megakernel<R, C, A>), grid, block, 0, stream, sc, f, acc, hit_ids, stats); break;
        CRT_MEGA(0, 0, false, false) CRT_MEGA(1, 0, false, true) CRT_MEGA(2, 0, true, false) CRT_MEGA(3, 0, true, true)
        CRT_MEGA(4, 1, false, false) CRT_MEGA(5, 1, false, true) CRT_MEGA(6, 1, true, false) CRT_MEGA(7, 1, true, true)
#undef CRT_MEGA
    }
    HIP_TRY(hipGetLastError());
    return true;
}

// ------------------------------------------------------------------------------------------------
// Wavefront tracer.  One launch sequence = init -> generate -> [extend -> shade+compact] x max_bounces -> accumulate,
// over a batch of n_batch consecutive iterations (crt_render) whose paths share the queues.
//
//   queue entry (48 B, SoA over three float4 arrays, two queues used ping-pong):
//     ray_o = {origin.xyz, rng word 0}    ray_d = {direction.xyz, batch index << 26 | pixel slot in the packed band buffer}
//     thr   = {color.rgb, rng word 1}
//     ray_color's emittedColor is not carried: a path that meets an emitter ends there (scatter() is false for a light,
//     compute.comp:416-418), so it is zero for every path still in a queue.
//   hit[i] = {t, id} of queue entry i (8 B), written by extend, read by shade.
//   pass[b][pixel] = radiance of iteration b's path through that pixel (16 B), written by shade when the path ends
//     (each pixel owns exactly one path per iteration), added to the image in iteration order by k_wf_accumulate:
//     main()'s read-modify-write (compute.comp:470-476), so a batch gives the bits n_batch dispatches would.
//
// The queue is split into sub-queues (contiguous runs of 8x8 tiles; 64 in use, kShards is the counter table's
// stride).  A path never leaves its sub-queue, so a sub-queue's capacity is its initial size and every sub-queue has
// its own size and work-cursor words: one returning atomic on ONE word saturates near 88 per microsecond on this chip
// (MI355X_MICROARCH.md, "dequeue"), which at one atomic per 64 rays capped a single-counter version of
// this design at ~5.6 Grays/s before any traversal work.
//
// extend is persistent: a fixed grid of waves pulls 64-ray batches from its home sub-queue, then steals
// from the others, until everything is drained; a late bounce with few rays costs one short launch.
// shade appends survivors to the other queue with a wave ballot + prefix count and a single atomic per wave on the
// sub-queue's own counter.
// ------------------------------------------------------------------------------------------------
constexpr uint32_t kShards = 256;
constexpr uint32_t kMaxBounces = 16;
constexpr uint32_t kCtrStride = 64;    // one counter per 256 bytes: neighbouring words share an L2 line and serialise their atomics
constexpr uint32_t kFetchBatches = 4;  // at most this many 64-ray batches per work-fetch atomic

struct WfQueues {
    float4 *ray_o[2], *ray_d[2], *thr[2];
    uint2 *hit;
    uint32_t *counts;       // [kMaxBounces + 1][kShards] live paths entering bounce b in sub-queue s
    uint32_t *cursors;      // [kMaxBounces][kShards]     extend's work-fetch cursor
    uint32_t tiles_per_shard, shard_cap;   // shard_cap = tiles_per_shard * 64 slots
    uint32_t n_shards;                     // sub-queues in use (<= kShards)
    uint32_t refill;                       // idle lanes that trigger a refill
    uint32_t leaf_lanes;                   // lanes waiting at a leaf that trigger a leaf-test phase
    uint32_t lds_depth;                    // LDS stack entries per lane (lower only with CRT_FLAG_DEBUG_SMALL_STACK)
};

constexpr uint32_t kPixBits = 26;               // ray_d.w = batch index << 26 | pixel slot (batches need < 2^26 - 1 pixel slots)
constexpr uint32_t kDeadPixel = 0xFFFFFFFFu;   // queue entry of a lane outside the rendered grid (ragged sizes only)

__global__ void k_wf_init(WfQueues q, uint32_t n_tiles, unsigned long long n_active, DeviceStats *stats) {
    for (uint32_t i = threadIdx.x; i < (kMaxBounces + 1) * kShards; i += blockDim.x) {
        uint32_t v = 0;
        if (i < q.n_shards) {
            uint32_t t0 = i * q.tiles_per_shard, t1 = min(n_tiles, t0 + q.tiles_per_shard);
            v = t1 > t0 ? (t1 - t0) * 64u : 0u;
        }
        q.counts[i * kCtrStride] = v;
    }
    for (uint32_t i = threadIdx.x; i < kMaxBounces * kShards; i += blockDim.x) q.cursors[i * kCtrStride] = 0;
    if (threadIdx.x == 0) atomicAdd(&stats->paths, n_active);
}

template <int RNGM>
__global__ __launch_bounds__(256) void k_wf_generate(DeviceScene sc, Frame f, WfQueues q, float4 *pass_buf, uint32_t pass_stride, uint32_t *hit_ids, DeviceStats *stats) {
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t vt = xcd_remap(blockIdx.x, gridDim.x) * 4 + wave;      // virtual tile: iteration-major over the batch
    if (vt >= f.n_tiles * f.n_batch) return;
    const uint32_t it = vt / f.n_tiles, tile = vt - it * f.n_tiles;
    uint32_t x = 0, y = 0, prow = 0;
    const bool active = tile_pixel(f, tile, lane, x, y, prow);
    const uint32_t shard = vt / q.tiles_per_shard;
    const uint32_t slot = shard * q.shard_cap + (vt - shard * q.tiles_per_shard) * 64 + lane;
    if (!active) {
        q.ray_o[0][slot] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        q.ray_d[0][slot] = make_float4(0.0f, 0.0f, 1.0f, u2f(kDeadPixel));
        return;
    }
    const crt_camera_data &cam = f.p.camera;
    const float seed = f.seeds[it], iteration = f.iter0 + (float)it;
    Rng<RNGM> rng;
    rng.init(seed, x, y, y * f.width + x, iteration);
    float ox, oy; hash_2(rng, ox, oy);
    float u = ((float)x + ox) / (float)f.width, v = ((float)y + oy) / (float)f.height;
    V3 o, d;
    camera_get_ray(cam, u, v, rng, o, d);
    if (cam.shouldRenderAABB != 0u) {          // R15 debug view: a sample on a box outline is white and traces nothing
        const bool outline = aabb_outline(sc, o, d);
        const unsigned long long killed = (unsigned long long)__popcll(__ballot(outline));
        if (killed && (unsigned long long)__ffsll((long long)__ballot(outline)) - 1ull == lane) atomicAdd(&stats->rays, 0ull - killed);   // extend counts every queue entry that is not a ragged-edge lane
        if (outline) {
            const uint32_t pix = prow * f.width + x;
            pass_buf[(size_t)it * pass_stride + pix] = make_float4(1.0f, 1.0f, 1.0f, 0.0f);
            if (it + 1 == f.n_batch) hit_ids[pix] = kIdMiss;
            q.ray_o[0][slot] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            q.ray_d[0][slot] = make_float4(0.0f, 0.0f, 1.0f, u2f(kDeadPixel));
            return;
        }
    }
    const uint64_t rs = rng.pack();
    q.ray_o[0][slot] = make_float4(o.x, o.y, o.z, u2f((uint32_t)rs));
    q.ray_d[0][slot] = make_float4(d.x, d.y, d.z, u2f((it << kPixBits) | (prow * f.width + x)));
    // the throughput of a primary ray is (1, 1, 1) and the hash RNG's state has no high word: k_wf_shade does not read
    // this array at bounce 0 then (16 of the 48 bytes per path this kernel would write, 16 of the 56 that one would read)
    if (RNGM != 0) q.thr[0][slot] = make_float4(1.0f, 1.0f, 1.0f, u2f((uint32_t)(rs >> 32)));
}

// Home sub-queue of a wave: blocks b and b + 8 share an XCD (and its L2), so the waves of one XCD start on
// sub-queues congruent to it modulo 8, i.e. on the same stretch of the image.
__device__ __forceinline__ uint32_t home_shard(uint32_t block, uint32_t wave, uint32_t n_shards) {
    if (n_shards < 8) return block % n_shards;
    return (block % 8) + 8 * (((block / 8) * 4 + wave) % (n_shards / 8));
}

// Persistent extend with per-lane refill.  Each wave owns a pool [pool_next, pool_end) of queue slots fetched from a
// sub-queue cursor; whenever at least kRefillLanes lanes are idle they take the next slots of the pool (ballot +
// prefix popcount), so lanes do not wait for the slowest ray of a batch.  Every wave reaches the exit: the pool can
// only be refilled from cursors that grow monotonically up to fixed sizes.
constexpr uint32_t kRefillLanes = 16;
constexpr uint32_t kLeafLanes = 24;      // lanes waiting at a leaf that trigger a leaf-test phase (swept 8..32 on four workloads)

// Per-lane traversal state behind one interface, so that k_wf_extend is written once for both node formats.  The
// overflow ("spill") part of the stack is a separate local array handed in by pointer: inside the struct its dynamic
// indexing would keep the whole state in scratch memory.  SPILL = false (wide format only) when the tree is shallow
// enough for the LDS stack.
template <bool WIDE, bool SPILL> struct LaneTrav;
template <bool SPILL> struct LaneTrav<false, SPILL> {          // sibling-pair BVH (intersect.h)
    using Entry = uint32_t;
    static constexpr int kSpill = kSpillStack;
    Traversal tr;
    CRT_DEV void begin(V3 o, V3 d) { trav_begin(tr, o, d); }
    CRT_DEV void set_tmax(float t) { tr.best_t = t; tr.tfar = t * 1.000004f; }
    CRT_DEV bool at_leaf() const { return (tr.cur & kLeafTag) != 0; }
    template <bool COUNT> CRT_DEV bool inner(const DeviceScene &sc, const uint8_t *, Entry *stack, Entry *spill, int depth, Counts &c) { return trav_inner<COUNT>(sc, tr, stack, 256, spill, depth, c); }
    template <bool COUNT, bool EXT> CRT_DEV bool leaf(const DeviceScene &sc, const uint8_t *, Entry *stack, Entry *spill, int depth, Counts &c) { return trav_leaf<COUNT, EXT>(sc, tr, stack, 256, spill, depth, c); }
};
template <bool SPILL> struct LaneTrav<true, SPILL> {           // 8-wide quantised BVH (intersect_wide.h)
    using Entry = uint2;
    static constexpr int kSpill = SPILL ? kWideSpill : 1;
    WideTrav tr;
    CRT_DEV void begin(V3 o, V3 d) { wide_begin(tr, o, d); }
    CRT_DEV void set_tmax(float t) { tr.best_t = t; }
    CRT_DEV bool at_leaf() const { return (tr.cur & kLeafTag) != 0; }
    template <bool COUNT> CRT_DEV bool inner(const DeviceScene &sc, const uint8_t *order, Entry *stack, Entry *spill, int depth, Counts &c) { return wide_inner<COUNT, SPILL>(sc.wide, sc.wide_links, order, tr, stack, 256, spill, depth, c); }
    template <bool COUNT, bool EXT> CRT_DEV bool leaf(const DeviceScene &sc, const uint8_t *order, Entry *stack, Entry *spill, int depth, Counts &c) { return wide_leaf<COUNT, SPILL, EXT>(sc, order, tr, stack, 256, spill, depth, c); }
};
static_assert(kLdsStack * 4 == kWideLds * 8, "both stack layouts use the same LDS block");

#ifndef CRT_EXTEND_WAVES
#define CRT_EXTEND_WAVES 6
#endif
// QUERY (crt_rays_query): 0 = path segments; 1 = caller's rays, closest hit in (t_near, t_max] (t_max rides in ray_o.w);
// 2 = same interval, any hit: a lane stops at the first primitive it finds.
// EXT: the variant that looks at the CRT_EXT_* flags (always with SPILL: the extensions' kernels are not the headline
// path and one stack layout keeps their number down).
template <bool COUNT, bool WIDE, bool SPILL, int QUERY = 0, bool EXT = false>
__global__ __launch_bounds__(256, CRT_EXTEND_WAVES) void k_wf_extend(DeviceScene sc, WfQueues q, uint32_t bounce, unsigned long long dead, DeviceStats *stats) {
    __shared__ uint2 s_stack[kWideLds * 256];
    __shared__ uint8_t s_order[WIDE ? kWideOrderBytes : 4];
    if (WIDE) { wide_order_table(s_order, threadIdx.x, blockDim.x); __syncthreads(); }
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t *counts = q.counts + bounce * kShards * kCtrStride;
    uint32_t *cursors = q.cursors + bounce * kShards * kCtrStride;
    const float4 *ro = q.ray_o[bounce & 1], *rd = q.ray_d[bounce & 1];
    unsigned long long nodes = 0, prims = 0;
    unsigned long long c_iter = 0, c_inner = 0, c_inner_lanes = 0, c_leaf = 0, c_leaf_lanes = 0, c_refill = 0;     // COUNT only, wave-uniform
    if (blockIdx.x == 0) {
        static_assert(kShards <= 256, "one thread of block 0 per sub-queue");
        unsigned long long n = wave_sum(threadIdx.x < kShards ? (unsigned long long)counts[threadIdx.x * kCtrStride] : 0ull);
        if (lane == 0) atomicAdd(&stats->rays, n - (wave == 0 ? dead : 0ull));
    }
    if (sc.n_pairs == 0) {      // empty scene: every ray misses
        for (uint32_t s = 0; s < q.n_shards; s++) {
            const uint32_t n = counts[s * kCtrStride];
            for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
                q.hit[s * q.shard_cap + i] = make_uint2(f2u(kInf), kIdMiss);
        }
        return;
    }
    const uint32_t ns = q.n_shards, per = ns >= 8 ? ns / 8 : 1;
    const uint32_t home = home_shard(blockIdx.x, wave, ns);
    const uint32_t waves_here = max(1u, (gridDim.x * 4u) / ns);
    // wave-uniform work state
    uint32_t k = 0;                       // position in the steal order
    uint32_t pool_next = 0, pool_end = 0, pool_base = 0;   // slots pool_base + [pool_next, pool_end)
    bool drained = false;
    // per-lane state
    bool busy = false;
    uint32_t slot = 0;
    using LT = LaneTrav<WIDE, SPILL>;
    LT lt;
    typename LT::Entry spill[LT::kSpill];
    typename LT::Entry *stack = (typename LT::Entry *)s_stack + threadIdx.x;
    const int lds_depth = (int)q.lds_depth;      // LDS stack entries per lane in use (the block holds 12 x 8 bytes = 24 x 4 bytes per lane)
    for (;;) {
        const unsigned long long idle = __ballot(!busy);
        if (COUNT) c_iter++;
        if (!drained && (uint32_t)__popcll(idle) >= q.refill) {
            if (COUNT) c_refill++;
            if (pool_next >= pool_end) {
                // fetch the next chunk: own sub-queue first, then the others of this XCD's residue class, then the rest
                while (k < ns) {
                    const uint32_t s = ns >= 8 ? (home + 8 * (k % per) + (k / per)) % ns : (home + k) % ns;
                    const uint32_t n = counts[s * kCtrStride];
                    uint32_t base = n;
                    if (n != 0) {
                        // up to kFetchBatches batches per atomic while the sub-queue is long (atomic rate), one when it is short (balance)
                        const uint32_t fetch = min(kFetchBatches, max(1u, n / (waves_here * 64u * 8u)));
                        uint32_t *cursor = cursors + s * kCtrStride;
                        // a stale (cached) cursor value can only be too small, so it never skips live work
                        if (lane == 0 && *(volatile uint32_t *)cursor < n) base = atomicAdd(cursor, 64u * fetch);
                        base = __shfl(base, 0);
                        if (base < n) { pool_base = s * q.shard_cap; pool_next = base; pool_end = min(n, base + 64u * fetch); break; }
                    }
                    k++;
                    // Skip the exhausted sub-queues that follow with ONE round of loads: lane j looks at the (k + j)-th
                    // sub-queue of the steal order (ns <= 64).  Walking them one dependent load at a time cost every
                    // wave ~100 us at the end of every launch, most of a late bounce's launch.
                    {
                        const uint32_t kk = k + lane;
                        const uint32_t sl = ns >= 8 ? (home + 8 * (kk % per) + (kk / per)) % ns : (home + kk) % ns;
                        const bool live = kk < ns && *(volatile uint32_t *)(cursors + sl * kCtrStride) < counts[sl * kCtrStride];
                        const unsigned long long m = __ballot(live);
                        k = m ? k + (uint32_t)__builtin_ctzll(m) : ns;
                    }
                }
                if (k >= ns) drained = true;
            }
            if (!drained) {
                const uint32_t rank = (uint32_t)__popcll(idle & ((1ull << lane) - 1ull));
                const uint32_t i = pool_next + rank;
                if (!busy && i < pool_end) {
                    slot = pool_base + i;
                    float4 o4 = ro[slot], d4 = rd[slot];
                    asm volatile("" :: "v"(o4.x), "v"(o4.y), "v"(o4.z), "v"(o4.w));      // both halves of the entry in one round trip (the compiler would sink the origin's load behind the dead-pixel test)
                    if (f2u(d4.w) == kDeadPixel) q.hit[slot] = make_uint2(f2u(kInf), kIdMiss);
                    else {
                        lt.begin(mk3(o4.x, o4.y, o4.z), mk3(d4.x, d4.y, d4.z));
                        if (QUERY) lt.set_tmax(o4.w);
                        busy = true;
                    }
                }
                pool_next = min(pool_end, pool_next + (uint32_t)__popcll(idle));
            }
        }
        if (__ballot(busy) == 0) {
            if (drained) break;
            continue;
        }
        // while-while: lanes whose cursor is a leaf wait until enough of them have one (or nobody can descend),
        // then the wave runs one leaf-test phase; otherwise the others do one inner step
        const bool at_leaf = busy && lt.at_leaf();
        const uint32_t n_leaf = (uint32_t)__popcll(__ballot(at_leaf));
        const bool any_inner = __ballot(busy && !at_leaf) != 0;
        Counts cnt = {0, 0};
        bool more = true;
        if (n_leaf >= q.leaf_lanes || !any_inner) {
            if (COUNT) { c_leaf++; c_leaf_lanes += n_leaf; }
            if (at_leaf) more = lt.template leaf<COUNT, EXT>(sc, s_order, stack, spill, lds_depth, cnt);
        } else {
            if (COUNT) { c_inner++; c_inner_lanes += (uint32_t)__popcll(__ballot(busy && !at_leaf)); }
            if (busy && !at_leaf) more = lt.template inner<COUNT>(sc, s_order, stack, spill, lds_depth, cnt);
        }
        if (QUERY == 2 && busy && lt.tr.best_id != kIdMiss) more = false;
        if (!more) {
            q.hit[slot] = make_uint2(f2u(lt.tr.best_t), lt.tr.best_id);
            busy = false;
        }
        if (COUNT) { nodes += cnt.nodes; prims += cnt.prims; }
    }
    if (COUNT) {
        nodes = wave_sum(nodes); prims = wave_sum(prims);
        if (lane == 0) {
            atomicAdd(&stats->node_visits, nodes); atomicAdd(&stats->prim_tests, prims);
            unsigned long long *c = stats->loop[bounce];
            atomicAdd(&c[0], c_iter); atomicAdd(&c[1], c_inner); atomicAdd(&c[2], c_inner_lanes);
            atomicAdd(&c[3], c_leaf); atomicAdd(&c[4], c_leaf_lanes); atomicAdd(&c[5], c_refill);
        }
    }
}

extern "C" __device__ uint32_t __ockl_wfred_or_u32(uint32_t);      // OR over the wave (ROCm device library)

// Bounce 0 of the 8-wide tree as PACKETS: the 64 primary rays of a queue batch are the 8 x 8 pixels of one tile, so they
// walk nearly the same nodes.  The wave keeps ONE cursor and ONE stack for all of them (wave-uniform values: scalar
// registers, scalar branches, one memory request per node instead of up to 64); a node is entered when ANY lane's ray
// enters its box, every lane tests the eight children against its own ray and its own best hit, and the leaf tests run
// for all lanes at once.  A lane thus also tests boxes and primitives that its own traversal would have skipped: tests
// that can only fail (a ray that hits a primitive enters every box around it), so hits and IDs are those of k_wf_extend.
// Children are visited in the order of the packet's first ray's octant.
// A tile that looks at distant, dense geometry (the horizon of a 10 M-triangle terrain) is not coherent: its rays end in
// hundreds of different leaves and the packet would visit the union of their paths (bounce 0 of config 5: three times
// k_wf_extend's time).  A packet that has entered more than `max_visits` nodes therefore stops being one: its lanes keep
// the hits they have and finish on their own, each with its own cursor and stack (the LaneTrav steps of k_wf_extend).
constexpr uint32_t kPrimaryWaves = 5;      // 96 VGPRs with the per-lane fallback in the kernel; launched on k_wf_extend's grid all the same (6 blocks per CU: 1.3 % faster than 5)
// FIRST = 0: rays from the bounce-0 queue (k_wf_generate), hit records for k_wf_shade.  FIRST = 1 + RNG mode: the whole
// first bounce in this kernel: the wave makes its tile's camera rays itself (k_wf_generate's code), traces them and
// shades them (k_wf_shade's code) while they are in registers, and writes the survivors straight into the bounce-1 queue.
// No bounce-0 queue, no hit records, two launches fewer: 80 bytes per path that are never written and 56 that are never
// read back (the rays, their hit records), of the ~290 the first bounce moved.  `count` = dead lanes (FIRST = 0) or paths.
#ifndef CRT_FIRST_WAVES
#define CRT_FIRST_WAVES 5
#endif
template <int FIRST>
__global__ __launch_bounds__(256, (FIRST ? CRT_FIRST_WAVES : kPrimaryWaves)) void k_wf_primary(DeviceScene sc, Frame f, WfQueues q, unsigned long long count, uint32_t max_visits,
                                                                    float4 *pass_buf, uint32_t pass_stride, uint32_t *hit_ids, DeviceStats *stats) {
    __shared__ uint2 s_lane_stack[kWideLds * 256];
    __shared__ uint4 s_stack[4][kWideSpill + 2];      // group header, pending mask, epoch
    __shared__ uint8_t s_order[kWideOrderBytes];
    wide_order_table(s_order, threadIdx.x, blockDim.x);
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t *counts = q.counts;
    uint32_t *cursors = q.cursors;
    const float4 *ro = q.ray_o[0], *rd = q.ray_d[0];
    if (blockIdx.x == 0) {
        if (FIRST) { if (threadIdx.x == 0) atomicAdd(&stats->rays, count); }
        else {
            unsigned long long n = wave_sum(threadIdx.x < kShards ? (unsigned long long)counts[threadIdx.x * kCtrStride] : 0ull);
            if (lane == 0) atomicAdd(&stats->rays, n - (wave == 0 ? count : 0ull));
        }
    }
    const uint32_t ns = q.n_shards, per = ns >= 8 ? ns / 8 : 1;
    const uint32_t home = home_shard(blockIdx.x, wave, ns);
    const uint32_t waves_here = max(1u, (gridDim.x * 4u) / ns);
    uint4 *stack = s_stack[wave];
    uint32_t k = 0, pool_next = 0, pool_end = 0, pool_base = 0, pool_shard = 0;
    for (;;) {
        if (pool_next >= pool_end) {
            while (k < ns) {
                const uint32_t s = ns >= 8 ? (home + 8 * (k % per) + (k / per)) % ns : (home + k) % ns;
                const uint32_t n = counts[s * kCtrStride];
                uint32_t base = n;
                if (n != 0) {
                    const uint32_t fetch = min(kFetchBatches, max(1u, n / (waves_here * 64u * 8u)));
                    uint32_t *cursor = cursors + s * kCtrStride;
                    if (lane == 0 && *(volatile uint32_t *)cursor < n) base = atomicAdd(cursor, 64u * fetch);
                    base = __shfl(base, 0);
                    if (base < n) { pool_base = s * q.shard_cap; pool_shard = s; pool_next = base; pool_end = min(n, base + 64u * fetch); break; }
                }
                k++;
                {
                    const uint32_t kk = k + lane;
                    const uint32_t sl = ns >= 8 ? (home + 8 * (kk % per) + (kk / per)) % ns : (home + kk) % ns;
                    const bool live = kk < ns && *(volatile uint32_t *)(cursors + sl * kCtrStride) < counts[sl * kCtrStride];
                    const unsigned long long m = __ballot(live);
                    k = m ? k + (uint32_t)__builtin_ctzll(m) : ns;
                }
            }
            if (k >= ns) break;
        }
        const uint32_t i = pool_next + lane;
        pool_next = min(pool_end, pool_next + 64u);
        const bool in_pool = i < pool_end;
        const uint32_t slot = pool_base + (in_pool ? i : pool_end - 1u);
        bool active;
        V3 o, d;
        uint32_t pix = 0, it = 0;
        Rng<(FIRST > 1 ? 1 : 0)> rng;
        if constexpr (FIRST != 0) {
            // queue slot -> virtual tile (iteration-major over the batch) -> pixel -> camera ray: k_wf_generate's code (main(), compute.comp:460-468)
            const uint32_t vt = pool_shard * q.tiles_per_shard + (i >> 6);
            it = vt / f.n_tiles;
            const uint32_t tile = vt - it * f.n_tiles;
            uint32_t x = 0, y = 0, prow = 0;
            active = in_pool && vt < f.n_tiles * f.n_batch && tile_pixel(f, tile, i & 63u, x, y, prow);
            pix = prow * f.width + x;
            const crt_camera_data &cam = f.p.camera;
            rng.init(f.seeds[it < kBatchMax ? it : 0], x, y, y * f.width + x, f.iter0 + (float)it);
            float ox, oy; hash_2(rng, ox, oy);
            const float u = ((float)x + ox) / (float)f.width, v = ((float)y + oy) / (float)f.height;
            camera_get_ray(cam, u, v, rng, o, d);
        } else {
            const float4 o4 = ro[slot], d4 = rd[slot];
            active = in_pool && f2u(d4.w) != kDeadPixel;
            o = mk3(o4.x, o4.y, o4.z); d = mk3(d4.x, d4.y, d4.z);
        }
        const unsigned long long act = __ballot(active);
        WideTrav tr;
        wide_begin(tr, o, d);
        if (act != 0ull && sc.n_pairs != 0) {
            const uint32_t oct8 = (uint32_t)__builtin_amdgcn_readlane((int)tr.oct8, (int)__builtin_ctzll(act));
            const bool one_octant = __ballot(active && tr.oct8 != oct8) == 0ull;
            const bool unx = (oct8 >> 8) & 1u, uny = (oct8 >> 9) & 1u, unz = (oct8 >> 10) & 1u;
            // epoch: leaf tests so far.  A group whose mask is older than the last leaf test is stale: some lane's best hit
            // may since have moved in front of its children.  38 % of a packet's node visits find no child to enter (3.8 of
            // 10.1 on the headline scene), and 44 % of those miss even the node's own quantisation frame (origin .. origin +
            // 255 steps, all in the header row): stale nodes get that one slab test first.
            uint32_t cur = 0, g_hdr = 0, g_pend = 0, g_epoch = 0, epoch = 0, sp = 0, visits = 0;      // wave-uniform
            bool alone = false, stale = false;
            for (;;) {
                if (!(cur & kLeafTag)) {
                    if (++visits > max_visits) { alone = true; break; }
                    const float4 *p = sc.wide + (size_t)kWideNodeF4 * cur;
                    const float4 h = p[0];
                    bool enter = true;
                    if (stale) {
                        const uint32_t hx = f2u(h.x), hy = f2u(h.y), hz = f2u(h.z);
                        const float ox = u2f(hx & 0xFFFFFE00u), oy = u2f(hy & 0xFFFFFE00u), oz = u2f(hz & 0xFFFFFE00u);
                        float tn;
                        const bool pass = active && slab(make_float4(ox, oy, oz, 0.0f), make_float4(ox + 255.0f * u2f(hx << 23), oy + 255.0f * u2f(hy << 23), oz + 255.0f * u2f(hz << 23), 0.0f),
                                                         tr.idir, tr.noid, tr.best_t * 1.000004f, tn);
                        enter = __ballot(pass) != 0ull;
                    }
                    if (enter) {
                        const float4 r1 = p[1], r2 = p[2], r3 = p[3];
                        uint32_t m = 0u;
                        if (one_octant) { if (active) m = wide_test8(h, r1, r2, r3, tr.idir, tr.noid, tr.best_t * 1.000004f, unx, uny, unz); }
                        else if (active) m = wide_test8(h, r1, r2, r3, tr.idir, tr.noid, tr.best_t * 1.000004f);
                        const uint32_t any = (uint32_t)__builtin_amdgcn_readfirstlane((int)__ockl_wfred_or_u32(m));      // DPP reduction: 8 ballots cost twice as much
                        if (any != 0u) {
                            if ((g_pend & 0xFFu) != 0u) { if (lane == 0) stack[sp] = make_uint4(g_hdr, g_pend, g_epoch, 0u); sp++; }
                            g_hdr = (uint32_t)__builtin_amdgcn_readfirstlane((int)f2u(h.w));
                            g_pend = (cur << 8) | any;
                            g_epoch = epoch;
                        }
                    }
                } else {
                    const uint32_t e = cur, kind = (e >> 29) & 3u, n = ((e >> 26) & 7u) + 1u, first = e & (kMaxLeafLink - 1u);
                    Counts cnt = {0, 0};
                    if (kind != 3u && active) test_leaf<false, false>(sc, tr.o, tr.d, first, (kind << 28) | n, tr.best_t, tr.best_id, cnt);
                    epoch++;
                }
                if ((g_pend & 0xFFu) == 0u) {
                    if (sp == 0u) break;
                    sp--;
                    const uint4 e = stack[sp];
                    g_hdr = (uint32_t)__builtin_amdgcn_readfirstlane((int)e.x); g_pend = (uint32_t)__builtin_amdgcn_readfirstlane((int)e.y);
                    g_epoch = (uint32_t)__builtin_amdgcn_readfirstlane((int)e.z);
                }
                stale = g_epoch != epoch;
                const uint32_t sl = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_order[oct8 | (g_pend & 0xFFu)]);
                const uint32_t bit = 1u << sl, imask = g_hdr & 0xFFu;
                g_pend ^= bit;
                if (imask & bit) cur = (g_hdr >> 8) + (uint32_t)__builtin_popcount(imask & (bit - 1u));
                else cur = (uint32_t)__builtin_amdgcn_readfirstlane((int)sc.wide_links[(size_t)(g_pend >> 8) * 8u + sl]);
            }
            if (alone) {      // (as a separate, non-inlined function this costs 126 instead of 96 VGPRs)
                LaneTrav<true, true> lt;
                lt.tr = tr;
                lt.tr.cur = 0; lt.tr.g_hdr = 0; lt.tr.g_pend = 0; lt.tr.sp = 0;
                uint2 spill[kWideSpill];
                Counts cnt = {0, 0};
                bool more = active;
                while (more) more = lt.at_leaf() ? lt.template leaf<false, false>(sc, s_order, s_lane_stack + threadIdx.x, spill, (int)q.lds_depth, cnt)
                                                 : lt.template inner<false>(sc, s_order, s_lane_stack + threadIdx.x, spill, (int)q.lds_depth, cnt);
                tr.best_t = lt.tr.best_t; tr.best_id = lt.tr.best_id;
            }
        }
        if constexpr (FIRST == 0) { if (in_pool) q.hit[slot] = make_uint2(f2u(tr.best_t), tr.best_id); }
        else {
            // k_wf_shade's body for bounce 0 (every lane takes part in the triangle-record fetch)
            constexpr int RNGM = FIRST > 1 ? 1 : 0;
            const uint32_t id = tr.best_id;
            if (active && it + 1 == f.n_batch) hit_ids[pix] = id;
            TriRec trec = {make_float4(0.0f, 0.0f, 0.0f, 0.0f), make_float4(0.0f, 0.0f, 0.0f, 0.0f), make_float4(0.0f, 0.0f, 0.0f, 0.0f), make_float4(0.0f, 0.0f, 0.0f, 0.0f)};
            if (sc.n_tris != 0u) trec = tri_rec_quad_fetch(sc.tri_shade, (active && (id >> 30) == CRT_ID_KIND_TRI) ? (id & 0x3FFFFFFFu) : 0u, lane);
            bool alive = false;
            V3 color = mk3(1.0f, 1.0f, 1.0f);
            if (active) {
                V3 emitted = mk3(0.0f, 0.0f, 0.0f), pass = mk3(0.0f, 0.0f, 0.0f);
                if (id == kIdMiss) pass = shade_miss(f.p, d, color);
                else alive = shade_hit<RNGM, false, true>(sc, o, d, tr.best_t, id, rng, color, emitted, pass, trec);
                if (alive && 1u >= f.max_bounces) { alive = false; pass = mk3(0.0f, 0.0f, 0.0f); }
                if (!alive) {
                    pass = pass * (1.0f / 1.0f);
                    pass_buf[(size_t)it * pass_stride + pix] = make_float4(pass.x, pass.y, pass.z, 0.0f);
                }
            }
            const unsigned long long mask = __ballot(alive);
            if (mask != 0ull) {
                uint32_t obase = 0;
                if (lane == 0) obase = atomicAdd(q.counts + (kShards + pool_shard) * kCtrStride, (uint32_t)__popcll(mask));
                obase = __shfl(obase, 0);
                if (alive) {
                    const uint32_t os = pool_shard * q.shard_cap + obase + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
                    const uint64_t rs = rng.pack();
                    q.ray_o[1][os] = make_float4(o.x, o.y, o.z, u2f((uint32_t)rs));
                    q.ray_d[1][os] = make_float4(d.x, d.y, d.z, u2f((it << kPixBits) | pix));
                    q.thr[1][os] = make_float4(color.x, color.y, color.z, u2f((uint32_t)(rs >> 32)));
                }
            }
        }
    }
}

#ifndef CRT_SHADE_WAVES
#define CRT_SHADE_WAVES 6
#endif
template <int RNGM, bool EXT = false>
__global__ __launch_bounds__(256, CRT_SHADE_WAVES) void k_wf_shade(DeviceScene sc, Frame f, WfQueues q, uint32_t bounce, float4 *pass_buf, uint32_t pass_stride, uint32_t *hit_ids) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t in = bounce & 1, out = in ^ 1;
    // The 64-path batches of ALL sub-queues, concatenated, are cut into one contiguous run per wave (sub-queues shrink
    // unevenly: sky tiles die at the first bounce; shading cost per path is near uniform, so no stealing).  Waves that
    // run together thus sit in different sub-queues in proportion to their sizes and spread the compaction atomics
    // over the counters.  Lane s holds sub-queue s's size; an inclusive wave scan of the batch counts locates the
    // sub-queue of a global batch index with one ballot.
    const uint32_t wave_id = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = (gridDim.x * blockDim.x) >> 6;
    const uint32_t my_n = lane < q.n_shards ? q.counts[(bounce * kShards + lane) * kCtrStride] : 0u;      // n_shards <= 64
    uint32_t incl = (my_n + 63u) / 64u;
    for (int off = 1; off < 64; off <<= 1) { const uint32_t v = (uint32_t)__shfl_up((int)incl, off); if ((int)lane >= off) incl += v; }
    const uint32_t n_batches = (uint32_t)__shfl((int)incl, 63);
    const bool last = bounce + 1 >= f.max_bounces;
    const uint32_t g_begin = (uint32_t)((uint64_t)wave_id * n_batches / n_waves), g_end = (uint32_t)((uint64_t)(wave_id + 1) * n_batches / n_waves);
    for (uint32_t g = g_begin; g < g_end; g++) {
        const uint32_t s = (uint32_t)__builtin_ctzll(__ballot(incl > g));
        const uint32_t n = (uint32_t)__shfl((int)my_n, (int)s);
        const uint32_t base = (g - ((uint32_t)__shfl((int)incl, (int)s) - (n + 63u) / 64u)) * 64u;
        uint32_t *out_count = q.counts + ((bounce + 1) * kShards + s) * kCtrStride;
        const uint32_t i = base + lane;
        bool alive = false;
        V3 o = mk3(0, 0, 0), d = mk3(0, 0, 0), color = mk3(0, 0, 0);
        uint32_t pix = kDeadPixel, it = 0;
        Rng<RNGM> rng;
        // every lane of the wave takes part in the triangle-record fetch (shade.h tri_rec_quad_fetch), so the hit is read
        // first and by all lanes; lanes without a live path or a triangle hit ask for record 0
        const uint32_t slot = s * q.shard_cap + min(i, n - 1u);
        const float4 d4 = q.ray_d[in][slot];
        const uint2 h = q.hit[slot];
        const bool live = i < n && f2u(d4.w) != kDeadPixel;
        TriRec trec = {make_float4(0.0f, 0.0f, 0.0f, 0.0f), make_float4(0.0f, 0.0f, 0.0f, 0.0f), make_float4(0.0f, 0.0f, 0.0f, 0.0f), make_float4(0.0f, 0.0f, 0.0f, 0.0f)};
        if (sc.n_tris != 0u) trec = tri_rec_quad_fetch(sc.tri_shade, (live && (h.y >> 30) == CRT_ID_KIND_TRI) ? (h.y & 0x3FFFFFFFu) : 0u, lane);
        if (live) {
            pix = f2u(d4.w);
            {
                const float4 o4 = q.ray_o[in][slot];
                float4 c4 = make_float4(1.0f, 1.0f, 1.0f, 0.0f);      // what k_wf_generate would have stored
                if (RNGM != 0 || bounce != 0) c4 = q.thr[in][slot];
                o = mk3(o4.x, o4.y, o4.z); d = mk3(d4.x, d4.y, d4.z);
                color = mk3(c4.x, c4.y, c4.z);
                V3 emitted = mk3(0.0f, 0.0f, 0.0f);      // see the queue-entry comment
                if (f.n_batch > 1) { it = pix >> kPixBits; pix &= (1u << kPixBits) - 1u; }      // a single iteration may use all 32 bits for the slot
                rng.unpack((uint64_t)f2u(o4.w) | ((uint64_t)f2u(c4.w) << 32), ((uint64_t)(uint32_t)(f.iter0 + (float)it) << 1) | 1u);
                const float t = u2f(h.x); const uint32_t id = h.y;
                if (bounce == 0 && it + 1 == f.n_batch) hit_ids[pix] = id;
                V3 pass = mk3(0.0f, 0.0f, 0.0f);
                if (id == kIdMiss) pass = shade_miss(f.p, d, color);
                else alive = shade_hit<RNGM, EXT, true>(sc, o, d, t, id, rng, color, emitted, pass, trec);
                if (alive && last) { alive = false; pass = mk3(0.0f, 0.0f, 0.0f); }     // "Kill ray after MAX_BOUNCES", compute.comp:457
                if (!alive) {                                                            // main(), :470-473; k_wf_accumulate adds it in
                    pass = pass * (1.0f / 1.0f);
                    pass_buf[(size_t)it * pass_stride + pix] = make_float4(pass.x, pass.y, pass.z, 0.0f);
                }
            }
        }
        const unsigned long long mask = __ballot(alive);
        if (mask == 0) continue;
        uint32_t obase = 0;
        if (lane == 0) obase = atomicAdd(out_count, (uint32_t)__popcll(mask));
        obase = __shfl(obase, 0);
        if (alive) {
            const uint32_t slot = s * q.shard_cap + obase + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
            const uint64_t rs = rng.pack();
            q.ray_o[out][slot] = make_float4(o.x, o.y, o.z, u2f((uint32_t)rs));
            q.ray_d[out][slot] = make_float4(d.x, d.y, d.z, u2f((it << kPixBits) | pix));
            q.thr[out][slot] = make_float4(color.x, color.y, color.z, u2f((uint32_t)(rs >> 32)));
        }
    }
}

// The last bounces of a SMALL batch (one sample of a 1080p frame: the reference's own per-frame dispatch, and config 5).
// From the third bounce on such a batch holds a few hundred thousand paths at most, and a k_wf_extend / k_wf_shade pair
// per bounce costs its launches plus the latency of its longest ray, ~65 us a bounce with the chip nearly empty.  This
// kernel takes the survivors of bounce `bounce0` and runs each path to its end, one lane per path: traversal (the same
// LaneTrav steps, without the wave-level phases) and shade_hit in a loop, no queue traffic in between.  Same functions,
// same order of RNG draws per path as the per-bounce kernels: the image is bit-identical.
// GEN (bounce0 == 0, no outline view): the lane makes its camera ray itself instead of reading k_wf_generate's queue entry.
template <int RNGM, bool WIDE, bool SPILL, bool EXT, bool GEN = false>
// accum != nullptr (GEN, one iteration): the path's radiance goes straight into the accumulation image (main(), compute.comp:470-476:
// every pixel has exactly one path), no radiance slot and no k_wf_accumulate launch.
__global__ __launch_bounds__(256) void k_wf_tail(DeviceScene sc, Frame f, WfQueues q, uint32_t bounce0, unsigned long long dead, float4 *pass_buf, uint32_t pass_stride, uint32_t *hit_ids, float4 *accum, DeviceStats *stats) {
    __shared__ uint2 s_stack[kWideLds * 256];
    __shared__ uint8_t s_order[WIDE ? kWideOrderBytes : 4];
    if (WIDE) { wide_order_table(s_order, threadIdx.x, blockDim.x); __syncthreads(); }
    if (bounce0 == 0 && blockIdx.x == 0) {      // bounce 0 is counted the way k_wf_extend counts it: queue entries minus ragged-edge lanes (k_wf_generate corrects for the outline view)
        unsigned long long n = wave_sum(threadIdx.x < kShards ? (unsigned long long)q.counts[threadIdx.x * kCtrStride] : 0ull);
        if ((threadIdx.x & 63u) == 0u) atomicAdd(&stats->rays, n - (threadIdx.x == 0 ? dead : 0ull));
    }
    const uint32_t in = bounce0 & 1;
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t s = t / q.shard_cap, i = t - s * q.shard_cap;
    unsigned long long rays = 0;
    if (s < q.n_shards && i < q.counts[(bounce0 * kShards + s) * kCtrStride]) {
        const uint32_t slot = s * q.shard_cap + i;
        V3 o, d, color = mk3(1.0f, 1.0f, 1.0f);
        uint32_t pix = kDeadPixel, it = 0;
        Rng<RNGM> rng;
        if constexpr (GEN) {
            // queue slot -> virtual tile -> pixel -> camera ray (k_wf_generate's code)
            const uint32_t vt = s * q.tiles_per_shard + (i >> 6);
            it = vt / f.n_tiles;
            uint32_t x = 0, y = 0, prow = 0;
            const bool active = vt < f.n_tiles * f.n_batch && tile_pixel(f, vt - it * f.n_tiles, i & 63u, x, y, prow);
            rng.init(f.seeds[it < kBatchMax ? it : 0], x, y, y * f.width + x, f.iter0 + (float)it);
            float ox, oy; hash_2(rng, ox, oy);
            const float u = ((float)x + ox) / (float)f.width, v = ((float)y + oy) / (float)f.height;
            camera_get_ray(f.p.camera, u, v, rng, o, d);
            if (active) pix = prow * f.width + x;
        } else {
            const float4 o4 = q.ray_o[in][slot], d4 = q.ray_d[in][slot];
            float4 c4 = make_float4(1.0f, 1.0f, 1.0f, 0.0f);      // what k_wf_generate would have stored (see there)
            if (RNGM != 0 || bounce0 != 0) c4 = q.thr[in][slot];
            o = mk3(o4.x, o4.y, o4.z); d = mk3(d4.x, d4.y, d4.z); color = mk3(c4.x, c4.y, c4.z);
            pix = f2u(d4.w);
            if (pix != kDeadPixel && f.n_batch > 1) { it = pix >> kPixBits; pix &= (1u << kPixBits) - 1u; }
            rng.unpack((uint64_t)f2u(o4.w) | ((uint64_t)f2u(c4.w) << 32), ((uint64_t)(uint32_t)(f.iter0 + (float)it) << 1) | 1u);
        }
        if (pix != kDeadPixel) {      // (a lane outside the rendered grid: bounce 0 only)
        using LT = LaneTrav<WIDE, SPILL>;
        typename LT::Entry spill[LT::kSpill];
        typename LT::Entry *stack = (typename LT::Entry *)s_stack + threadIdx.x;
        const int lds_depth = (int)q.lds_depth;
        V3 pass = mk3(0.0f, 0.0f, 0.0f);
        for (uint32_t b = bounce0; b < f.max_bounces; b++) {
            LT lt;
            lt.begin(o, d);
            Counts cnt = {0, 0};
            bool more = sc.n_pairs != 0;
            while (more) more = lt.at_leaf() ? lt.template leaf<false, EXT>(sc, s_order, stack, spill, lds_depth, cnt)
                                             : lt.template inner<false>(sc, s_order, stack, spill, lds_depth, cnt);
            if (b != 0) rays++;
            const uint32_t id = lt.tr.best_id;
            if (b == 0 && it + 1 == f.n_batch) hit_ids[pix] = id;
            V3 emitted = mk3(0.0f, 0.0f, 0.0f);
            bool alive = false;
            if (id == kIdMiss) pass = shade_miss(f.p, d, color);
            else alive = shade_hit<RNGM, EXT>(sc, o, d, lt.tr.best_t, id, rng, color, emitted, pass);
            if (alive && b + 1 >= f.max_bounces) { alive = false; pass = mk3(0.0f, 0.0f, 0.0f); }      // "Kill ray after MAX_BOUNCES", compute.comp:457
            if (!alive) break;
        }
        pass = pass * (1.0f / 1.0f);
        if (GEN && accum) {
            const float4 cum4 = accum[pix];
            const V3 cum = accumulate(mk3(cum4.x, cum4.y, cum4.z), pass, f.iter0);
            accum[pix] = make_float4(cum.x, cum.y, cum.z, f.iter0);
        } else pass_buf[(size_t)it * pass_stride + pix] = make_float4(pass.x, pass.y, pass.z, 0.0f);
        }
    }
    rays = wave_sum(rays);
    if ((threadIdx.x & 63u) == 0u && rays) atomicAdd(&stats->rays, rays);
}

// The path kernel with persistent waves: a lane whose path ended takes the next entry of the wave's pool as soon as
// `refill` lanes are idle (k_wf_extend's work fetch), lanes whose ray is done wait for a shade phase (`shade_lanes` of
// them, or nobody left to step), everybody else takes one traversal step per iteration.
// Faster than one lane per path where traversal is compute-bound (cirno 1080p: 1.56 instead of 1.84 ms per unbatched
// sample), slower where it waits for memory (config 5: 3.17 instead of 2.87 ms per frame): the context measures.
// GEN / accum as in k_wf_tail.
template <int RNGM, bool WIDE, bool SPILL, bool EXT, bool GEN>
__global__ __launch_bounds__(256) void k_wf_path(DeviceScene sc, Frame f, WfQueues q, uint32_t bounce0, unsigned long long dead, uint32_t shade_lanes,
                                                 float4 *pass_buf, uint32_t pass_stride, uint32_t *hit_ids, float4 *accum, DeviceStats *stats) {
    __shared__ uint2 s_stack[kWideLds * 256];
    __shared__ uint8_t s_order[WIDE ? kWideOrderBytes : 4];
    if (WIDE) { wide_order_table(s_order, threadIdx.x, blockDim.x); __syncthreads(); }
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t *counts = q.counts + bounce0 * kShards * kCtrStride;
    uint32_t *cursors = q.cursors + bounce0 * kShards * kCtrStride;
    if (bounce0 == 0 && blockIdx.x == 0) {
        unsigned long long n = wave_sum(threadIdx.x < kShards ? (unsigned long long)counts[threadIdx.x * kCtrStride] : 0ull);
        if (lane == 0) atomicAdd(&stats->rays, n - (wave == 0 ? dead : 0ull));
    }
    const uint32_t in = bounce0 & 1;
    const uint32_t ns = q.n_shards, per = ns >= 8 ? ns / 8 : 1;
    const uint32_t home = home_shard(blockIdx.x, wave, ns);
    uint32_t k = 0, pool_next = 0, pool_end = 0, pool_base = 0, pool_shard = 0;
    bool drained = false;
    bool busy = false, shading = false;      // per lane: has a path; its ray is traced and waits for the shade phase
    using LT = LaneTrav<WIDE, SPILL>;
    LT lt;
    typename LT::Entry spill[LT::kSpill];
    typename LT::Entry *stack = (typename LT::Entry *)s_stack + threadIdx.x;
    const int lds_depth = (int)q.lds_depth;
    V3 o = mk3(0, 0, 0), d = mk3(0, 0, 1), color = mk3(0, 0, 0);
    lt.begin(o, d);
    uint32_t pix = 0, it = 0, b = 0;
    Rng<RNGM> rng;
    rng.unpack(0, 1);
    unsigned long long rays = 0;
    for (;;) {
        const unsigned long long idle = __ballot(!busy);
        if (!drained && (uint32_t)__popcll(idle) >= q.refill) {
            if (pool_next >= pool_end) {
                while (k < ns) {
                    const uint32_t s = ns >= 8 ? (home + 8 * (k % per) + (k / per)) % ns : (home + k) % ns;
                    const uint32_t n = counts[s * kCtrStride];
                    uint32_t base = n;
                    if (n != 0) {
                        uint32_t *cursor = cursors + s * kCtrStride;
                        if (lane == 0 && *(volatile uint32_t *)cursor < n) base = atomicAdd(cursor, 64u);
                        base = __shfl(base, 0);
                        if (base < n) { pool_base = s * q.shard_cap; pool_shard = s; pool_next = base; pool_end = min(n, base + 64u); break; }
                    }
                    k++;
                    {
                        const uint32_t kk = k + lane;
                        const uint32_t sl = ns >= 8 ? (home + 8 * (kk % per) + (kk / per)) % ns : (home + kk) % ns;
                        const bool live = kk < ns && *(volatile uint32_t *)(cursors + sl * kCtrStride) < counts[sl * kCtrStride];
                        const unsigned long long m = __ballot(live);
                        k = m ? k + (uint32_t)__builtin_ctzll(m) : ns;
                    }
                }
                if (k >= ns) drained = true;
            }
            if (!drained) {
                const uint32_t rank = (uint32_t)__popcll(idle & ((1ull << lane) - 1ull));
                const uint32_t i = pool_next + rank;
                if (!busy && i < pool_end) {
                    bool live;
                    if constexpr (GEN) {      // queue slot -> virtual tile -> pixel -> camera ray (k_wf_generate's code)
                        const uint32_t vt = pool_shard * q.tiles_per_shard + (i >> 6);
                        it = vt / f.n_tiles;
                        uint32_t x = 0, y = 0, prow = 0;
                        live = vt < f.n_tiles * f.n_batch && tile_pixel(f, vt - it * f.n_tiles, i & 63u, x, y, prow);
                        rng.init(f.seeds[it < kBatchMax ? it : 0], x, y, y * f.width + x, f.iter0 + (float)it);
                        float ox, oy; hash_2(rng, ox, oy);
                        const float u = ((float)x + ox) / (float)f.width, v = ((float)y + oy) / (float)f.height;
                        camera_get_ray(f.p.camera, u, v, rng, o, d);
                        color = mk3(1.0f, 1.0f, 1.0f);
                        pix = prow * f.width + x;
                    } else {
                        const uint32_t slot = pool_base + i;
                        const float4 o4 = q.ray_o[in][slot], d4 = q.ray_d[in][slot];
                        live = f2u(d4.w) != kDeadPixel;
                        float4 c4 = make_float4(1.0f, 1.0f, 1.0f, 0.0f);
                        if (RNGM != 0 || bounce0 != 0) c4 = q.thr[in][slot];
                        o = mk3(o4.x, o4.y, o4.z); d = mk3(d4.x, d4.y, d4.z); color = mk3(c4.x, c4.y, c4.z);
                        pix = f2u(d4.w); it = 0;
                        if (f.n_batch > 1) { it = pix >> kPixBits; pix &= (1u << kPixBits) - 1u; }
                        rng.unpack((uint64_t)f2u(o4.w) | ((uint64_t)f2u(c4.w) << 32), ((uint64_t)(uint32_t)(f.iter0 + (float)it) << 1) | 1u);
                    }
                    if (live) {
                        b = bounce0;
                        lt.begin(o, d);
                        busy = true; shading = sc.n_pairs == 0;
                    }
                }
                pool_next = min(pool_end, pool_next + (uint32_t)__popcll(idle));
            }
        }
        if (__ballot(busy) == 0) {
            if (drained) break;
            continue;
        }
        const bool stepping = busy && !shading;
        const uint32_t n_wait = (uint32_t)__popcll(__ballot(busy && shading));
        if (n_wait >= shade_lanes || __ballot(stepping) == 0) {
            if (busy && shading) {
                if (b != 0) rays++;
                const uint32_t id = lt.tr.best_id;
                if (b == 0 && it + 1 == f.n_batch) hit_ids[pix] = id;
                V3 emitted = mk3(0.0f, 0.0f, 0.0f), pass = mk3(0.0f, 0.0f, 0.0f);
                bool alive = false;
                if (id == kIdMiss) pass = shade_miss(f.p, d, color);
                else alive = shade_hit<RNGM, EXT>(sc, o, d, lt.tr.best_t, id, rng, color, emitted, pass);
                if (alive && b + 1 >= f.max_bounces) { alive = false; pass = mk3(0.0f, 0.0f, 0.0f); }
                if (alive) { b++; lt.begin(o, d); shading = sc.n_pairs == 0; }
                else {
                    pass = pass * (1.0f / 1.0f);
                    if (GEN && accum) {
                        const float4 cum4 = accum[pix];
                        const V3 cum = accumulate(mk3(cum4.x, cum4.y, cum4.z), pass, f.iter0);
                        accum[pix] = make_float4(cum.x, cum.y, cum.z, f.iter0);
                    } else pass_buf[(size_t)it * pass_stride + pix] = make_float4(pass.x, pass.y, pass.z, 0.0f);
                    busy = false; shading = false;
                }
            }
        } else if (stepping) {
            Counts cnt = {0, 0};
            const bool more = lt.at_leaf() ? lt.template leaf<false, EXT>(sc, s_order, stack, spill, lds_depth, cnt)
                                           : lt.template inner<false>(sc, s_order, stack, spill, lds_depth, cnt);
            if (!more) shading = true;
        }
    }
    rays = wave_sum(rays);
    if (lane == 0 && rays) atomicAdd(&stats->rays, rays);
}

// main()'s accumulation (compute.comp:470-476) for the iterations of a batch, in iteration order: every active pixel
// owns exactly one path per iteration and k_wf_shade stored that path's radiance when it ended.
__global__ __launch_bounds__(256) void k_wf_accumulate(Frame f, float4 *accum, const float4 *pass_buf, uint32_t pass_stride) {
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t tile = blockIdx.x * 4 + wave;
    if (tile >= f.n_tiles) return;
    uint32_t x = 0, y = 0, prow = 0;
    if (!tile_pixel(f, tile, lane, x, y, prow)) return;
    const size_t pix = (size_t)prow * f.width + x;
    const float4 cum4 = accum[pix];
    V3 cum = mk3(cum4.x, cum4.y, cum4.z);
    float iteration = f.iter0;
    for (uint32_t b = 0; b < f.n_batch; b++) {
        const float4 p = pass_buf[(size_t)b * pass_stride + pix];
        iteration = f.iter0 + (float)b;
        cum = accumulate(cum, mk3(p.x, p.y, p.z), iteration);
    }
    accum[pix] = make_float4(cum.x, cum.y, cum.z, iteration);
}

static bool ensure_dev(DeviceBuffer &b, size_t bytes, std::string &err) {
    if (b.ptr && b.bytes >= bytes) return true;
    if (b.ptr) { hipError_t e = hipFree(b.ptr); (void)e; b.ptr = nullptr; b.bytes = 0; }
    size_t want = bytes < 256 ? 256 : bytes;
    HIP_TRY(hipMalloc(&b.ptr, want));
    b.bytes = want;
    return true;
}

// One place that picks the k_wf_extend instantiation.  wide/spill as launch_wavefront decides them; alpha (some CRT_EXT_*
// flag is set: the EXT kernels) always takes the SPILL layout; queries are never instrumented.
template <int QUERY>
static void launch_extend(bool count, bool wide, bool spill, bool alpha, dim3 grid, hipStream_t stream, const DeviceScene &sc,
                          const WfQueues &q, uint32_t bounce, unsigned long long dead, DeviceStats *stats) {
    const dim3 block(256);
    if (alpha || !wide) spill = true;
    if (QUERY != 0) count = false;
    const int variant = (count ? 8 : 0) | (wide ? 4 : 0) | (spill ? 2 : 0) | (alpha ? 1 : 0);
    switch (variant) {
#define CRT_EXT_CASE(V, C, W, S, A) case V: hipLaunchKernelGGL((k_wf_extend<C, W, S, QUERY, A>), grid, block, 0, stream, sc, q, bounce, dead, stats); break;
        CRT_EXT_CASE(2, false, false, true, false)  CRT_EXT_CASE(3, false, false, true, true)
        CRT_EXT_CASE(4, false, true, false, false)  CRT_EXT_CASE(6, false, true, true, false)  CRT_EXT_CASE(7, false, true, true, true)
        default: break;
    }
    if constexpr (QUERY == 0) {
        switch (variant) {
            CRT_EXT_CASE(10, true, false, true, false)  CRT_EXT_CASE(11, true, false, true, true)
            CRT_EXT_CASE(12, true, true, false, false)  CRT_EXT_CASE(14, true, true, true, false)  CRT_EXT_CASE(15, true, true, true, true)
            default: break;
        }
    }
#undef CRT_EXT_CASE
}

bool launch_wavefront(hipStream_t stream, const DeviceSceneHost &sch, const FrameSetup &fs, const crt_scene_params &params,
                      uint32_t n_batch, const float *seeds, WavefrontState &wf, float *accum, uint32_t accum_pixels,
                      uint32_t *hit_ids, DeviceStats *stats, std::string &err) {
    DeviceScene sc = to_device_scene(sch);
    Frame f = make_frame(fs, params);
    if (f.n_tiles == 0) return true;
    if (n_batch < 1 || n_batch > kBatchMax) { err = "wavefront batch must hold 1..64 iterations"; return false; }
    if (n_batch > 1 && accum_pixels >= (1u << kPixBits) - 1u) { err = "wavefront batches need fewer than 2^26 - 1 pixel slots"; return false; }
    f.n_batch = n_batch;
    if (seeds) for (uint32_t b = 0; b < n_batch; b++) f.seeds[b] = seeds[b];
    const uint32_t n_vt = f.n_tiles * n_batch;
    if (f.max_bounces > kMaxBounces) { err = "wavefront tracer supports at most 16 bounces"; return false; }
    if (sch.n_pairs + 1 >= kMaxLeafLink) { err = "wavefront tracer supports at most 2^26 primitives"; return false; }
    WfQueues q;
    static const uint32_t env_shards = [] { const char *e = getenv("CRT_WF_SHARDS"); int v = e ? atoi(e) : 64;
                                            return (uint32_t)((v >= 1 && v <= 64 && (v & (v - 1)) == 0) ? v : 64); }();     // k_wf_shade: one lane per sub-queue
    q.n_shards = env_shards;
    static const uint32_t env_refill = [] { const char *e = getenv("CRT_WF_REFILL"); int v = e ? atoi(e) : (int)kRefillLanes; return (uint32_t)(v < 1 ? 1 : v > 64 ? 64 : v); }();
    q.refill = env_refill;
    static const uint32_t env_leaf = [] { const char *e = getenv("CRT_WF_LEAF_LANES"); int v = e ? atoi(e) : (int)kLeafLanes; return (uint32_t)(v < 1 ? 1 : v > 64 ? 64 : v); }();
    q.leaf_lanes = env_leaf;
    q.tiles_per_shard = (n_vt + q.n_shards - 1) / q.n_shards;
    q.shard_cap = q.tiles_per_shard * 64;
    const uint32_t cap = q.shard_cap * q.n_shards;
    if (wf.capacity < cap) {
        for (int k = 0; k < 2; k++) {
            if (!ensure_dev(wf.ray_o[k], (size_t)cap * 16, err) || !ensure_dev(wf.ray_d[k], (size_t)cap * 16, err) ||
                !ensure_dev(wf.thr[k], (size_t)cap * 16, err)) return false;
        }
        if (!ensure_dev(wf.hit, (size_t)cap * 8, err) || !ensure_dev(wf.counters, (size_t)(2 * kMaxBounces + 1) * kShards * kCtrStride * 4, err)) return false;
        wf.capacity = cap;
    }
    if (!ensure_dev(wf.pass, (size_t)n_batch * accum_pixels * 16, err)) return false;
    for (int k = 0; k < 2; k++) {
        q.ray_o[k] = (float4 *)wf.ray_o[k].ptr; q.ray_d[k] = (float4 *)wf.ray_d[k].ptr;
        q.thr[k] = (float4 *)wf.thr[k].ptr;
    }
    q.hit = (uint2 *)wf.hit.ptr;
    q.counts = (uint32_t *)wf.counters.ptr;
    q.cursors = q.counts + (kMaxBounces + 1) * kShards * kCtrStride;
    // active pixels of this context: every owned band is full height except possibly the image's last band
    unsigned long long n_active = 0;
    for (uint32_t lb = 0; lb < fs.n_local_bands; lb++) {
        uint32_t y0 = (lb * fs.band_world + fs.band_rank) * 8;
        uint32_t rows = y0 < fs.grid_h ? (fs.grid_h - y0 < 8 ? fs.grid_h - y0 : 8) : 0;
        n_active += (unsigned long long)rows * fs.grid_w;
    }
    const unsigned long long dead = ((unsigned long long)f.n_tiles * 64 - n_active) * n_batch;
    n_active *= n_batch;
    float4 *acc = (float4 *)accum;
    const dim3 block(256), gen_grid((n_vt + 3) / 4), acc_grid((f.n_tiles + 3) / 4);
    // persistent grids: 256 CUs x 6 resident blocks (LDS stack: 24 KiB per block); multiple of 8 blocks and of kShards waves
    uint32_t nb = (cap + 255) / 256;
    const uint32_t resident = 256u * CRT_EXTEND_WAVES;
    nb = nb < resident ? ((nb + 15) / 16) * 16 : resident;
    const dim3 pgrid(nb);
    static const uint32_t env_sgrid = [] { const char *e = getenv("CRT_WF_SHADE_GRID"); return (uint32_t)(e ? atoi(e) : 0); }();
    const dim3 sgrid(env_sgrid ? env_sgrid : nb);
    const bool pcg = fs.rng_mode != CRT_RNG_REF;
    hipLaunchKernelGGL(k_wf_init, dim3(1), dim3(256), 0, stream, q, n_vt, n_active, stats);
    // Bounce 0 over the 8-wide tree: tiles of primary rays as packets (k_wf_primary).  The instrumented and the extension
    // kernels stay per ray.  Unless something needs the bounce-0 queue (the outline debug view, which k_wf_generate draws;
    // the path kernel from bounce 0), the packet kernel runs the whole first bounce: rays, traversal, shading.
    static const bool env_primary = [] { const char *e = getenv("CRT_WF_PRIMARY"); return !e || atoi(e) != 0; }();
    static const bool env_first = [] { const char *e = getenv("CRT_WF_FIRST"); return !e || atoi(e) != 0; }();
    static const uint32_t env_packet_visits = [] { const char *e = getenv("CRT_WF_PACKET_VISITS"); return (uint32_t)(e ? atoi(e) : 32); }();
    const bool packets = env_primary && sc.wide && !sc.ext && !fs.instrumented && sch.wide_levels <= kWideSpill && wf.tail_bounce != 0;
    const bool first_bounce = packets && env_first && params.camera.shouldRenderAABB == 0u;
    static const uint32_t env_path_shade = [] { const char *e = getenv("CRT_WF_PATH_SHADE_LANES"); int v = e ? atoi(e) : 16; return (uint32_t)(v < 1 ? 1 : v > 64 ? 64 : v); }();
    const bool persist = wf.tail_persist && sc.wide && !sc.ext;      // k_wf_path instead of k_wf_tail (8-wide tree, no extension)
    const bool gen_in_tail = wf.tail_bounce == 0 && env_first && params.camera.shouldRenderAABB == 0u;      // the path kernel makes its own camera rays
    if (!first_bounce && !gen_in_tail) {
        if (pcg) hipLaunchKernelGGL((k_wf_generate<1>), gen_grid, block, 0, stream, sc, f, q, (float4 *)wf.pass.ptr, accum_pixels, hit_ids, stats);
        else hipLaunchKernelGGL((k_wf_generate<0>), gen_grid, block, 0, stream, sc, f, q, (float4 *)wf.pass.ptr, accum_pixels, hit_ids, stats);
    }
    // small batches finish in k_wf_tail from this bounce on (0: never)
    for (uint32_t b = 0; b < f.max_bounces; b++) {
        if (wf.tail_bounce >= 0 && b == (uint32_t)wf.tail_bounce) {      // small batches: the context decides (crt_context::TailTune)
            q.lds_depth = fs.small_stack ? 1u : (sc.wide ? (uint32_t)kWideLds : (uint32_t)kLdsStack);
            const bool wide = sc.wide != nullptr, spill = sc.ext != 0 || !(wide && sch.wide_levels <= q.lds_depth + 1u);
            const dim3 tgrid((cap + 255) / 256);
            const int variant = (pcg ? 8 : 0) | (wide ? 4 : 0) | (spill ? 2 : 0) | (sc.ext ? 1 : 0);
            if (persist) {
                float4 *direct = gen_in_tail && n_batch == 1 ? acc : (float4 *)nullptr;
                const unsigned long long dead_t = b == 0 ? dead : 0ull;
                const int pv = (pcg ? 2 : 0) | (spill ? 1 : 0);
#define CRT_PATH_CASE(V, R, S) case V: if (gen_in_tail) hipLaunchKernelGGL((k_wf_path<R, true, S, false, true>), pgrid, block, 0, stream, sc, f, q, b, dead_t, env_path_shade, (float4 *)wf.pass.ptr, accum_pixels, hit_ids, direct, stats); \
                                       else hipLaunchKernelGGL((k_wf_path<R, true, S, false, false>), pgrid, block, 0, stream, sc, f, q, b, dead_t, env_path_shade, (float4 *)wf.pass.ptr, accum_pixels, hit_ids, direct, stats); break;
                switch (pv) { CRT_PATH_CASE(0, 0, false) CRT_PATH_CASE(1, 0, true) CRT_PATH_CASE(2, 1, false) CRT_PATH_CASE(3, 1, true) default: break; }
#undef CRT_PATH_CASE
                break;
            }
            if (gen_in_tail) switch (variant) {
#define CRT_TAIL_CASE(V, R, W, S, A) case V: hipLaunchKernelGGL((k_wf_tail<R, W, S, A, true>), tgrid, block, 0, stream, sc, f, q, b, dead, (float4 *)wf.pass.ptr, accum_pixels, hit_ids, n_batch == 1 ? acc : (float4 *)nullptr, stats); break;
                CRT_TAIL_CASE(2, 0, false, true, false)  CRT_TAIL_CASE(3, 0, false, true, true)
                CRT_TAIL_CASE(4, 0, true, false, false)  CRT_TAIL_CASE(6, 0, true, true, false)  CRT_TAIL_CASE(7, 0, true, true, true)
                CRT_TAIL_CASE(10, 1, false, true, false)  CRT_TAIL_CASE(11, 1, false, true, true)
                CRT_TAIL_CASE(12, 1, true, false, false)  CRT_TAIL_CASE(14, 1, true, true, false)  CRT_TAIL_CASE(15, 1, true, true, true)
#undef CRT_TAIL_CASE
                default: break;
            }
            else switch (variant) {
#define CRT_TAIL_CASE(V, R, W, S, A) case V: hipLaunchKernelGGL((k_wf_tail<R, W, S, A>), tgrid, block, 0, stream, sc, f, q, b, b == 0 ? dead : 0ull, (float4 *)wf.pass.ptr, accum_pixels, hit_ids, (float4 *)nullptr, stats); break;
                CRT_TAIL_CASE(2, 0, false, true, false)  CRT_TAIL_CASE(3, 0, false, true, true)
                CRT_TAIL_CASE(4, 0, true, false, false)  CRT_TAIL_CASE(6, 0, true, true, false)  CRT_TAIL_CASE(7, 0, true, true, true)
                CRT_TAIL_CASE(10, 1, false, true, false)  CRT_TAIL_CASE(11, 1, false, true, true)
                CRT_TAIL_CASE(12, 1, true, false, false)  CRT_TAIL_CASE(14, 1, true, true, false)  CRT_TAIL_CASE(15, 1, true, true, true)
#undef CRT_TAIL_CASE
                default: break;
            }
            break;
        }
        const unsigned long long dead_b = b == 0 ? dead : 0ull;
        if (wf.time_kernels) {      // three events per bounce: before extend, between extend and shade, after shade
            while (wf.ev_used + 3 > wf.ev.size()) { hipEvent_t a; HIP_TRY(hipEventCreate(&a)); wf.ev.push_back(a); }
            HIP_TRY(hipEventRecord(wf.ev[wf.ev_used], stream));
        }
        q.lds_depth = fs.small_stack ? 1u : (sc.wide ? (uint32_t)kWideLds : (uint32_t)kLdsStack);
        // wide trees of at most lds_depth + 1 levels never overflow the LDS stack (at most levels - 1 groups wait on it)
        const bool primary = b == 0 && packets;
        if (wf.time_kernels) { wf.ev_primary.resize(wf.ev_used / 3 + 1); wf.ev_primary[wf.ev_used / 3] = primary ? (first_bounce ? 2 : 1) : 0; }      // 2: no k_wf_shade behind it
        if (primary && first_bounce) {
            if (pcg) hipLaunchKernelGGL(k_wf_primary<2>, pgrid, block, 0, stream, sc, f, q, n_active, env_packet_visits, (float4 *)wf.pass.ptr, accum_pixels, hit_ids, stats);
            else hipLaunchKernelGGL(k_wf_primary<1>, pgrid, block, 0, stream, sc, f, q, n_active, env_packet_visits, (float4 *)wf.pass.ptr, accum_pixels, hit_ids, stats);
            if (wf.time_kernels) { HIP_TRY(hipEventRecord(wf.ev[wf.ev_used + 1], stream)); HIP_TRY(hipEventRecord(wf.ev[wf.ev_used + 2], stream)); wf.ev_used += 3; }
            continue;      // no k_wf_shade for this bounce
        }
        if (primary) hipLaunchKernelGGL(k_wf_primary<0>, pgrid, block, 0, stream, sc, f, q, dead_b, env_packet_visits, (float4 *)nullptr, 0u, (uint32_t *)nullptr, stats);
        else
            launch_extend<0>(fs.instrumented, sc.wide != nullptr, !(sc.wide && sch.wide_levels <= q.lds_depth + 1u), sc.ext != 0,
                             pgrid, stream, sc, q, b, dead_b, stats);
        if (wf.time_kernels) HIP_TRY(hipEventRecord(wf.ev[wf.ev_used + 1], stream));
        if (sc.ext) {
            if (pcg) hipLaunchKernelGGL((k_wf_shade<1, true>), sgrid, block, 0, stream, sc, f, q, b, (float4 *)wf.pass.ptr, accum_pixels, hit_ids);
            else hipLaunchKernelGGL((k_wf_shade<0, true>), sgrid, block, 0, stream, sc, f, q, b, (float4 *)wf.pass.ptr, accum_pixels, hit_ids);
        } else if (pcg) hipLaunchKernelGGL((k_wf_shade<1, false>), sgrid, block, 0, stream, sc, f, q, b, (float4 *)wf.pass.ptr, accum_pixels, hit_ids);
        else hipLaunchKernelGGL((k_wf_shade<0, false>), sgrid, block, 0, stream, sc, f, q, b, (float4 *)wf.pass.ptr, accum_pixels, hit_ids);
        if (wf.time_kernels) { HIP_TRY(hipEventRecord(wf.ev[wf.ev_used + 2], stream)); wf.ev_used += 3; }
    }
    if (!(gen_in_tail && n_batch == 1))      // (the path kernel has accumulated its single iteration itself)
        hipLaunchKernelGGL(k_wf_accumulate, acc_grid, block, 0, stream, f, acc, (const float4 *)wf.pass.ptr, accum_pixels);
    HIP_TRY(hipGetLastError());
    return true;
}

// ---- ray queries (crt_rays_query) ----------------------------------------------------------------------------
// The caller's rays become bounce-0 queue entries (slot = ray index; the sub-queues are contiguous runs of slots) and
// one k_wf_extend launch answers them into q.hit.
__global__ void k_query_init(WfQueues q, uint32_t n_rays) {
    for (uint32_t i = threadIdx.x; i < kShards; i += blockDim.x) {
        const unsigned long long first = (unsigned long long)i * q.shard_cap;
        q.counts[i * kCtrStride] = (i < q.n_shards && first < n_rays) ? (uint32_t)min((unsigned long long)q.shard_cap, n_rays - first) : 0u;
        q.cursors[i * kCtrStride] = 0;
    }
}

__global__ __launch_bounds__(256) void k_query_pack(WfQueues q, const crt_ray *rays, uint32_t n_rays) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_rays) return;
    const float4 *r = (const float4 *)(rays + i);
    const float4 a = r[0], b = r[1];
    q.ray_o[0][i] = a;                                            // origin, t_max
    q.ray_d[0][i] = make_float4(b.x, b.y, b.z, u2f(i));           // direction, any value but kDeadPixel
}

bool launch_ray_query(hipStream_t stream, const DeviceSceneHost &sch, const FrameSetup &fs, WavefrontState &wf, const crt_ray *rays,
                      uint32_t n_rays, uint32_t mode, crt_ray_hit *hits, DeviceStats *stats, std::string &err) {
    static_assert(sizeof(crt_ray) == 32 && sizeof(crt_ray_hit) == 8, "crt_ray / crt_ray_hit layout");
    if (n_rays == 0) return true;
    if (n_rays == 0xFFFFFFFFu) { err = "at most 2^32 - 2 rays per query"; return false; }
    if (sch.n_pairs + 1 >= kMaxLeafLink) { err = "wavefront tracer supports at most 2^26 primitives"; return false; }
    DeviceScene sc = to_device_scene(sch);
    WfQueues q;
    const uint32_t n_tiles = (n_rays + 63) / 64;
    q.n_shards = n_tiles >= 64 * 16 ? 64 : 1;                    // sub-queues only pay once every wave can have its own chunks
    q.refill = kRefillLanes; q.leaf_lanes = kLeafLanes;
    q.tiles_per_shard = (n_tiles + q.n_shards - 1) / q.n_shards;
    q.shard_cap = q.tiles_per_shard * 64;
    const size_t cap = (size_t)q.shard_cap * q.n_shards;
    if (cap >= 0xFFFFFFFFull) { err = "too many rays for one query"; return false; }
    if (wf.capacity < cap) {
        for (int k = 0; k < 2; k++) {
            if (!ensure_dev(wf.ray_o[k], cap * 16, err) || !ensure_dev(wf.ray_d[k], cap * 16, err) || !ensure_dev(wf.thr[k], cap * 16, err)) return false;
        }
        if (!ensure_dev(wf.hit, cap * 8, err) || !ensure_dev(wf.counters, (size_t)(2 * kMaxBounces + 1) * kShards * kCtrStride * 4, err)) return false;
        wf.capacity = (uint32_t)cap;
    }
    if (!ensure_dev(wf.query_rays, (size_t)n_rays * sizeof(crt_ray), err)) return false;
    HIP_TRY(hipMemcpyAsync(wf.query_rays.ptr, rays, (size_t)n_rays * sizeof(crt_ray), hipMemcpyDefault, stream));     // host or device memory
    for (int k = 0; k < 2; k++) { q.ray_o[k] = (float4 *)wf.ray_o[k].ptr; q.ray_d[k] = (float4 *)wf.ray_d[k].ptr; q.thr[k] = (float4 *)wf.thr[k].ptr; }
    q.hit = (uint2 *)wf.hit.ptr;
    q.counts = (uint32_t *)wf.counters.ptr;
    q.cursors = q.counts + (kMaxBounces + 1) * kShards * kCtrStride;
    q.lds_depth = fs.small_stack ? 1u : (sc.wide ? (uint32_t)kWideLds : (uint32_t)kLdsStack);
    uint32_t nb = (uint32_t)((cap + 255) / 256);
    const uint32_t resident = 256u * CRT_EXTEND_WAVES;
    nb = nb < resident ? ((nb + 15) / 16) * 16 : resident;
    const dim3 block(256), pgrid(nb);
    hipLaunchKernelGGL(k_query_init, dim3(1), dim3(256), 0, stream, q, n_rays);
    hipLaunchKernelGGL(k_query_pack, dim3((n_rays + 255) / 256), block, 0, stream, q, (const crt_ray *)wf.query_rays.ptr, n_rays);
    const bool any = mode == CRT_QUERY_ANY;
    const bool spill = !(sc.wide && sch.wide_levels <= q.lds_depth + 1u), alpha = sc.ext != 0;
    if (any) launch_extend<2>(false, sc.wide != nullptr, spill, alpha, pgrid, stream, sc, q, 0u, 0ull, stats);
    else launch_extend<1>(false, sc.wide != nullptr, spill, alpha, pgrid, stream, sc, q, 0u, 0ull, stats);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(hits, wf.hit.ptr, (size_t)n_rays * sizeof(crt_ray_hit), hipMemcpyDefault, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    return true;
}

// ---- crt_rays_query_noalloc: the same launches over a scratch buffer that lives in the code object ------------------
// For callers that cannot allocate (HIP graph capture, tight serving loops, harnesses that forbid hipMalloc).  The queue
// of crt_rays_query (ray origins + directions) and its counter tables sit in a file-scope __device__ array; every GPU
// the library runs on holds its own copy at its own address, which noalloc_init looks up once per device
// (hipGetSymbolAddress with that device current).  Host code only ever uses that looked-up pointer: the array's NAME
// is a host-side stand-in, not a device address.
constexpr uint32_t kNoAllocMaxRays = CRT_NOALLOC_MAX_RAYS;
constexpr size_t kNoAllocRayBytes = (size_t)kNoAllocMaxRays * 16;                     // one float4 array
constexpr size_t kNoAllocCtrBytes = (size_t)kShards * kCtrStride * 4;                 // one counter table (bounce 0 only)
constexpr size_t kNoAllocBytes = 2 * kNoAllocRayBytes + 2 * kNoAllocCtrBytes;         // 4 MiB + 128 KiB
static_assert(kNoAllocMaxRays % (64 * 64) == 0, "capacity = whole sub-queues of whole 64-ray tiles");
static __device__ __align__(256) unsigned char g_noalloc_scratch[kNoAllocBytes];

constexpr int kNoAllocMaxDevices = 64;
static void *g_noalloc_ptr[kNoAllocMaxDevices];      // per device: address of ITS copy of g_noalloc_scratch (null: not looked up)
static std::mutex g_noalloc_mutex;

bool noalloc_init(int device, std::string &err) {
    if (device < 0 || device >= kNoAllocMaxDevices) { err = "device index beyond the no-alloc scratch table"; return false; }
    std::lock_guard<std::mutex> lock(g_noalloc_mutex);
    if (g_noalloc_ptr[device]) return true;
    void *ptr = nullptr;
    HIP_TRY(hipGetSymbolAddress(&ptr, HIP_SYMBOL(g_noalloc_scratch)));      // the caller made `device` current
    if (!ptr) { err = "hipGetSymbolAddress returned a null scratch address"; return false; }
    g_noalloc_ptr[device] = ptr;
    return true;
}

bool launch_ray_query_noalloc(hipStream_t stream, int device, const DeviceSceneHost &sch, const FrameSetup &fs, const crt_ray *rays_dev,
                              uint32_t n_rays, uint32_t mode, crt_ray_hit *hits_dev, DeviceStats *stats, std::string &err) {
    if (n_rays == 0) return true;
    if (n_rays > kNoAllocMaxRays) { err = "at most " + std::to_string(kNoAllocMaxRays) + " rays per call (CRT_NOALLOC_MAX_RAYS); use crt_rays_query for more"; return false; }
    if (sch.n_pairs + 1 >= kMaxLeafLink) { err = "wavefront tracer supports at most 2^26 primitives"; return false; }
    unsigned char *base = nullptr;
    {
        std::lock_guard<std::mutex> lock(g_noalloc_mutex);
        if (device >= 0 && device < kNoAllocMaxDevices) base = (unsigned char *)g_noalloc_ptr[device];
    }
    if (!base) { err = "no-alloc scratch of this device was not initialised (crt_create does it)"; return false; }
    DeviceScene sc = to_device_scene(sch);
    WfQueues q{};
    const uint32_t n_tiles = (n_rays + 63) / 64;
    q.n_shards = n_tiles >= 64 * 16 ? 64 : 1;
    q.refill = kRefillLanes; q.leaf_lanes = kLeafLanes;
    q.tiles_per_shard = (n_tiles + q.n_shards - 1) / q.n_shards;
    q.shard_cap = q.tiles_per_shard * 64;
    const size_t cap = (size_t)q.shard_cap * q.n_shards;
    if (cap > kNoAllocMaxRays) { err = "internal: query capacity beyond the scratch"; return false; }
    // only queue 0 and bounce 0 are touched by a query; slot == ray index, so the kernel writes its answers straight
    // into the caller's hits array (slots past n_rays are never produced: a sub-queue's count stops at n_rays)
    q.ray_o[0] = q.ray_o[1] = (float4 *)base;
    q.ray_d[0] = q.ray_d[1] = (float4 *)(base + kNoAllocRayBytes);
    q.thr[0] = q.thr[1] = nullptr;
    q.hit = (uint2 *)hits_dev;
    q.counts = (uint32_t *)(base + 2 * kNoAllocRayBytes);
    q.cursors = (uint32_t *)(base + 2 * kNoAllocRayBytes + kNoAllocCtrBytes);
    q.lds_depth = fs.small_stack ? 1u : (sc.wide ? (uint32_t)kWideLds : (uint32_t)kLdsStack);
    uint32_t nb = (uint32_t)((cap + 255) / 256);
    const uint32_t resident = 256u * CRT_EXTEND_WAVES;
    nb = nb < resident ? ((nb + 15) / 16) * 16 : resident;
    const dim3 block(256), pgrid(nb);
    // the scratch keeps the previous call's contents: k_query_init rewrites every count and cursor word
    hipLaunchKernelGGL(k_query_init, dim3(1), dim3(256), 0, stream, q, n_rays);
    hipLaunchKernelGGL(k_query_pack, dim3((n_rays + 255) / 256), block, 0, stream, q, rays_dev, n_rays);
    const bool any = mode == CRT_QUERY_ANY;
    const bool spill = !(sc.wide && sch.wide_levels <= q.lds_depth + 1u), alpha = sc.ext != 0;
    if (any) launch_extend<2>(false, sc.wide != nullptr, spill, alpha, pgrid, stream, sc, q, 0u, 0ull, stats);
    else launch_extend<1>(false, sc.wide != nullptr, spill, alpha, pgrid, stream, sc, q, 0u, 0ull, stats);
    HIP_TRY(hipGetLastError());
    return true;
}

size_t wavefront_bytes(const WavefrontState &wf) {
    const DeviceBuffer *all[] = {&wf.ray_o[0], &wf.ray_o[1], &wf.ray_d[0], &wf.ray_d[1], &wf.thr[0], &wf.thr[1], &wf.hit, &wf.counters, &wf.pass, &wf.query_rays};
    size_t n = 0;
    for (const DeviceBuffer *b : all) n += b->bytes;
    return n;
}

void free_wavefront(WavefrontState &wf) {
    DeviceBuffer *all[] = {&wf.ray_o[0], &wf.ray_o[1], &wf.ray_d[0], &wf.ray_d[1], &wf.thr[0], &wf.thr[1],
                           &wf.hit, &wf.counters, &wf.pass, &wf.query_rays};
    for (DeviceBuffer *b : all) if (b->ptr) { hipError_t e = hipFree(b->ptr); (void)e; b->ptr = nullptr; b->bytes = 0; }
    for (hipEvent_t e : wf.ev) { hipError_t r = hipEventDestroy(e); (void)r; }
    wf.ev.clear(); wf.ev_used = 0;
    wf.capacity = 0;
}

}  // namespace crt
