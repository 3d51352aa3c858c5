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

#include "build_internal.h"
#include "shade.h"

namespace crt {

struct Frame {
    crt_scene_params p;
    uint32_t width, height, grid_w, grid_h;
    uint32_t band_rank, band_world, n_local_bands, tiles_x, n_tiles;
    uint32_t max_bounces;
    uint64_t pcg_inc;
};

static DeviceScene to_device_scene(const DeviceSceneHost &h) {
    DeviceScene d;
    d.nodes = (const float4 *)h.nodes; d.n_pairs = h.n_pairs;
    d.tri_pack = (const float4 *)h.tri_pack; d.sphere_pack = (const float4 *)h.sphere_pack; d.quad_pack = (const float4 *)h.quad_pack;
    d.sphere_ids = h.sphere_ids; d.quad_ids = h.quad_ids;
    d.spheres = h.spheres; d.quads = h.quads; d.tris = h.tris; d.materials = h.materials;
    d.tex0 = h.tex0; d.tex0_w = h.tex0_w; d.tex0_h = h.tex0_h; d.n_textures = h.n_textures;
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

template <int RNGM, bool COUNT>
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
        for (uint32_t depth = 0; depth < f.max_bounces; depth++) {
            float t; uint32_t id; Counts cnt = {0, 0};
            closest_hit<COUNT>(sc, o, d, s_stack + threadIdx.x, 256, t, id, cnt);
            rays++;
            if (COUNT) { nodes += cnt.nodes; prims += cnt.prims; }
            if (depth == 0) primary = id;
            if (id == kIdMiss) { pass = shade_miss(f.p, d, color); break; }
            if (!shade_hit(sc, o, d, t, id, rng, color, emitted, pass)) break;
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
    if (fs.rng_mode == CRT_RNG_REF) {
        if (fs.instrumented) hipLaunchKernelGGL((k_megakernel<0, true>), grid, block, 0, stream, sc, f, acc, hit_ids, stats);
        else hipLaunchKernelGGL((k_megakernel<0, false>), grid, block, 0, stream, sc, f, acc, hit_ids, stats);
    } else {
        if (fs.instrumented) hipLaunchKernelGGL((k_megakernel<1, true>), grid, block, 0, stream, sc, f, acc, hit_ids, stats);
        else hipLaunchKernelGGL((k_megakernel<1, false>), grid, block, 0, stream, sc, f, acc, hit_ids, stats);
    }
    HIP_TRY(hipGetLastError());
    return true;
}

bool launch_wavefront(hipStream_t, const DeviceSceneHost &, const FrameSetup &, const crt_scene_params &, WavefrontState &,
                      float *, uint32_t *, DeviceStats *, std::string &err) {
    err = "wavefront tracer not built yet";
    return false;
}

void free_wavefront(WavefrontState &wf) {
    DeviceBuffer *all[] = {&wf.ray_o[0], &wf.ray_o[1], &wf.ray_d[0], &wf.ray_d[1], &wf.thr[0], &wf.thr[1], &wf.emi[0], &wf.emi[1],
                           &wf.hit, &wf.counters, &wf.pass};
    for (DeviceBuffer *b : all) if (b->ptr) { hipError_t e = hipFree(b->ptr); (void)e; b->ptr = nullptr; b->bytes = 0; }
    wf.capacity = 0;
}

}  // namespace crt
