// context.hip -- the C ABI of include/crt/abi.h on top of the HIP kernels.
// Replaces the descriptor/dispatch surface of VulkanEngine (engine/src/vk_engine.cpp:
// init_descriptors :1116-1237, upload_buffer :656-690, draw :1370-1498).
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "build_internal.h"
#include "gather_internal.h"
#include "crt/abi.h"

namespace crt {
void set_host_error(const std::string &msg);
bool debug_math(hipStream_t stream, int fn, uint32_t n, const float *a, const float *b, float *out0, float *out1, std::string &err);      // debug_math.hip
}      // host/host.cpp: the thread-local message behind crt_host_last_error()

using namespace crt;

struct crt_context {
    int device = 0;
    hipStream_t stream = nullptr;
    // Two timing slots, like the reference's two frame slots with a fence each (vk_engine.h:18, vk_engine.cpp:1377):
    // dispatch i records into slot i % 2 and only waits if the dispatch two calls back has not finished yet.
    struct TimingSlot { hipEvent_t start = nullptr, stop = nullptr; bool pending = false; int tune_mode = -1, tune_modes = 2; float tune_scale = 1.0f; } slot[2];
    // Small batches (up to kTailPaths paths: one sample of a 1080p frame) can finish in k_wf_tail, one lane per path through all
    // its bounces, instead of a k_wf_extend / k_wf_shade pair per bounce.  Which is faster depends on the scene: per-bounce
    // launches wait for the longest RAY of every bounce, the tail kernel for the longest PATH, which wins where some rays
    // are very long (10 M-triangle grid seen at grazing angles: 3.1 instead of 4.2 ms a frame) and loses where all
    // are short (cirno 1080p: 1.6 instead of 1.0 ms).  Same image either way, so the context measures: the first
    // dispatches take the forms in turn, and after two timed samples of each (six dispatches at most) the fastest stays until the scene or
    // the configuration changes, or crt_retune asks for a new measurement.  CRT_WF_TAIL_BOUNCE=n fixes the mode (n < 0: never, else the first bounce of the tail kernel).
    // Modes: 0 = a kernel pair per bounce, 1 = one lane per path (k_wf_tail), 2 = the persistent path kernel (k_wf_path; only
    // over an 8-wide tree with no extension set).
    struct TailTune { int decided = -1; uint32_t calls = 0, samples[3] = {0, 0, 0}; float best[3] = {3.4e38f, 3.4e38f, 3.4e38f}; } tune;
    uint32_t slot_next = 0;
    hipEvent_t ev_build[2] = {nullptr, nullptr};
    bool build_time_pending = false;      // an enqueue-only crt_bvh_build: its device time is read at the next blocking call
    DeviceBuffer refit_cost;              // two floats: sum of node areas, area of the root (crt_bvh_refit)
    std::string error;
    crt_config cfg{};
    bool configured = false, have_scene = false, have_bvh = false;
    // scene (reference layouts, input order)
    DeviceBuffer spheres, quads, tris, tris_rest, materials, tex0, tex_fp, tex_table, dbg_boxes, dbg_scan;      // tex0: every texture's RGBA8 texels, textures[0] first (staging during upload); tex_fp: their footprint form (k_tex_footprints), what the kernels sample; dbg_scan: two flag words of k_tri_material_scan
    uint32_t n_dbg_boxes = 0;
    uint32_t n_spheres = 0, n_quads = 0, n_tris = 0, n_materials = 0, n_textures = 0, tex0_w = 0, tex0_h = 0;
    uint32_t ext = 0;               // CRT_EXT_* (crt_extensions_set)
    bool tex_index_ok = true;       // every material's textureIndex is CRT_BAD_INDEX or < n_textures
    bool animated = false;
    BuildState build;
    // image
    DeviceBuffer accum, hit_ids, stats;
    // crt_image_export_fd: a copy of the band buffer in memory that other APIs can import (HIP virtual memory management:
    // a physical allocation that exports as a dma-buf file descriptor, mapped into this process at `va`)
    struct ExportImage { hipMemGenericAllocationHandle_t handle{}; void *va = nullptr; size_t alloc_bytes = 0, image_bytes = 0; bool live = false; } exported;
    uint32_t grid_w = 0, grid_h = 0, n_local_bands = 0, local_rows = 0;
    WavefrontState wf;
    GatherState gather;
    int instrumented = 0;
    double last_render_ms = 0.0, last_build_ms = 0.0, last_extend_ms = 0.0, last_shade_ms = 0.0;
    double last_primary_ms = 0.0;
    uint64_t last_extend_launches = 0, last_shade_launches = 0, last_primary_launches = 0;
};

namespace {

int fail(crt_context *ctx, int code, const std::string &msg) { if (ctx) ctx->error = msg; return code; }

#define CTX_HIP(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return fail(ctx, CRT_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); } while (0)

// out[0] = some materialIndex in a range of device-resident triangles is past the material array (the kernels index it
// unchecked).
__global__ void k_tri_material_scan(const crt_tri *tris, uint32_t first, uint32_t count, uint32_t n_materials, uint32_t *out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count && tris[first + i].materialIndex >= n_materials) out[0] = 1u;
}

// Footprint form of the resident textures: entry (x, y) of a texture = its texels (x, y), (x+1, y), (x, y+1), (x+1, y+1)
// with REPEAT wrap (the four a bilinear sample at (x, y) blends, vk_initializers.cpp:252-261), so a sample is one 16-byte
// load.  table[t] = {first texel, width, height, 0}; one thread per texel of all textures.
__global__ void k_tex_footprints(const uint32_t *texels, const uint4 *table, uint32_t n_textures, uint32_t n_texels, uint4 *out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_texels) return;
    uint32_t t = 0;
    while (t + 1 < n_textures && table[t + 1].x <= i) t++;
    const uint4 e = table[t];
    const uint32_t local = i - e.x, y = local / e.y, x = local - y * e.y;
    const uint32_t x1 = x + 1 == e.y ? 0u : x + 1, y1 = y + 1 == e.z ? 0u : y + 1;
    const uint32_t *base = texels + e.x;
    out[i] = make_uint4(base[(size_t)y * e.y + x], base[(size_t)y * e.y + x1], base[(size_t)y1 * e.y + x], base[(size_t)y1 * e.y + x1]);
}

int bind(crt_context *ctx) {
    if (!ctx) return CRT_ERR_INVALID;
    CTX_HIP(hipSetDevice(ctx->device));
    return CRT_OK;
}

int alloc(crt_context *ctx, DeviceBuffer &b, size_t bytes) {
    if (b.ptr && b.bytes >= bytes) return CRT_OK;
    if (b.ptr) { CTX_HIP(hipFree(b.ptr)); b.ptr = nullptr; b.bytes = 0; }
    size_t want = bytes < 256 ? 256 : bytes;
    CTX_HIP(hipMalloc(&b.ptr, want));
    b.bytes = want;
    return CRT_OK;
}

void release(DeviceBuffer &b) { if (b.ptr) { hipError_t e = hipFree(b.ptr); (void)e; } b.ptr = nullptr; b.bytes = 0; }

void release_export(crt_context *ctx) {
    crt_context::ExportImage &x = ctx->exported;
    if (x.live) {
        hipError_t e = hipMemUnmap(x.va, x.alloc_bytes); e = hipMemAddressFree(x.va, x.alloc_bytes); e = hipMemRelease(x.handle); (void)e;
    }
    x = crt_context::ExportImage();
}

int upload(crt_context *ctx, DeviceBuffer &b, const void *src, size_t bytes) {
    int rc = alloc(ctx, b, bytes);
    if (rc) return rc;
    if (bytes) CTX_HIP(hipMemcpy(b.ptr, src, bytes, hipMemcpyHostToDevice));   // synchronous, like immediate_submit + fence wait
    return CRT_OK;
}

uint32_t local_bands(uint32_t grid_h, uint32_t rank, uint32_t world) {
    uint32_t nb = (grid_h + 7) / 8;
    if (world <= 1) return nb;
    return nb > rank ? (nb - rank + world - 1) / world : 0;
}

DeviceSceneHost scene_view(const crt_context *ctx) {
    DeviceSceneHost s{};
    s.nodes = ctx->build.pairs_valid ? ctx->build.pairs.ptr : nullptr; s.n_pairs = ctx->build.n_pairs;
    s.wide = ctx->build.n_wide ? ctx->build.wide.ptr : nullptr; s.n_wide = ctx->build.n_wide; s.wide_levels = ctx->build.wide_levels;
    s.wide_links = ctx->build.n_wide ? (const uint32_t *)((const float4 *)ctx->build.wide.ptr + (size_t)kWideStrideF4 * ctx->build.wide_cap) : nullptr;
    s.tri_pack = ctx->build.tri_pack.ptr; s.sphere_pack = ctx->build.sphere_pack.ptr; s.quad_pack = ctx->build.quad_pack.ptr;
    s.tri_shade = ctx->build.tri_shade.ptr;
    s.sphere_ids = (const uint32_t *)ctx->build.sphere_ids.ptr; s.quad_ids = (const uint32_t *)ctx->build.quad_ids.ptr;
    s.spheres = (const crt_sphere *)ctx->spheres.ptr; s.quads = (const crt_quad *)ctx->quads.ptr; s.tris = (const crt_tri *)ctx->tris.ptr; s.n_tris = ctx->n_tris;
    s.materials = (const crt_material *)ctx->materials.ptr;
    s.tex0 = (const uint4 *)ctx->tex_fp.ptr; s.tex0_w = ctx->tex0_w; s.tex0_h = ctx->tex0_h; s.n_textures = ctx->n_textures;
    s.tex_table = (const uint4 *)ctx->tex_table.ptr; s.ext = ctx->ext;
    s.dbg_boxes = ctx->n_dbg_boxes ? ctx->dbg_boxes.ptr : nullptr; s.n_dbg_boxes = ctx->n_dbg_boxes;
    return s;
}

FrameSetup frame_setup(const crt_context *ctx) {
    FrameSetup fs{};
    fs.width = ctx->cfg.width; fs.height = ctx->cfg.height; fs.grid_w = ctx->grid_w; fs.grid_h = ctx->grid_h;
    fs.band_rank = ctx->cfg.band_world > 1 ? ctx->cfg.band_rank : 0;
    fs.band_world = ctx->cfg.band_world > 1 ? ctx->cfg.band_world : 1;
    fs.n_local_bands = ctx->n_local_bands;
    fs.max_bounces = ctx->cfg.max_bounces ? ctx->cfg.max_bounces : 10;   // MAX_BOUNCES, compute.comp:16
    fs.rng_mode = ctx->cfg.rng_mode;
    fs.instrumented = (uint32_t)ctx->instrumented;
    fs.small_stack = (ctx->cfg.flags & CRT_FLAG_DEBUG_SMALL_STACK) ? 1u : 0u;
    return fs;
}

// Reads the device time of a finished dispatch out of its slot.  block = false: only if it has already finished.
int harvest_slot(crt_context *ctx, crt_context::TimingSlot &sl, bool block) {
    if (!sl.pending) return CRT_OK;
    if (!block) {
        const hipError_t q = hipEventQuery(sl.stop);
        if (q == hipErrorNotReady) return CRT_OK;
        if (q != hipSuccess) return fail(ctx, CRT_ERR_HIP, std::string("hipEventQuery: ") + hipGetErrorString(q));
    } else CTX_HIP(hipEventSynchronize(sl.stop));
    float ms = 0.0f;
    CTX_HIP(hipEventElapsedTime(&ms, sl.start, sl.stop));
    ctx->last_render_ms = ms;
    sl.pending = false;
    if (sl.tune_mode >= 0 && ctx->tune.decided < 0) {
        crt_context::TailTune &t = ctx->tune;
        t.best[sl.tune_mode] = std::min(t.best[sl.tune_mode], ms / sl.tune_scale);
        t.samples[sl.tune_mode]++;
        const int n_modes = sl.tune_modes;
        bool done = true;
        for (int m = 0; m < n_modes; m++) done = done && t.samples[m] >= 2;      // two samples of each form: six measured dispatches at most
        if (done) { int b = 0; for (int m = 1; m < n_modes; m++) if (t.best[m] < t.best[b]) b = m; t.decided = b; }
    }
    sl.tune_mode = -1;
    return CRT_OK;
}

// Blocking: every dispatch issued so far has its time read (older slot first, so last_render_ms is the newest) and,
// with crt_set_kernel_timing on, the k_wf_extend events are summed.
int finish_build(crt_context *ctx) {
    // an enqueue-only build (crt_bvh_build with build_ms == NULL): its time, its node counts, and whether the launch plan
    // was deep enough
    if (ctx->build_time_pending) {
        CTX_HIP(hipEventSynchronize(ctx->ev_build[1]));
        float ms = 0.0f;
        CTX_HIP(hipEventElapsedTime(&ms, ctx->ev_build[0], ctx->ev_build[1]));
        ctx->last_build_ms = ms;
        ctx->build_time_pending = false;
    }
    if (ctx->build.counts_pending) {
        bool overflow = false; std::string err;
        if (!resolve_wide_counts(ctx->build, overflow, err)) return fail(ctx, CRT_ERR_HIP, err);
        if (overflow) {
            ctx->have_bvh = false;
            return fail(ctx, CRT_ERR_INVALID, "the enqueue-only crt_bvh_build planned for a shallower tree than this frame's primitives needed (parts of the scene were "
                                              "left out of the BVH and of every image rendered since); call crt_bvh_build again");
        }
    }
    return CRT_OK;
}

int finish_timing(crt_context *ctx) {
    { int rc = finish_build(ctx); if (rc) return rc; }
    for (uint32_t k = 0; k < 2; k++) {
        int rc = harvest_slot(ctx, ctx->slot[(ctx->slot_next + k) & 1u], true);
        if (rc) return rc;
    }
    if (ctx->wf.ev_used) {
        ctx->last_extend_ms = 0.0; ctx->last_shade_ms = 0.0; ctx->last_primary_ms = 0.0;
        ctx->last_extend_launches = 0; ctx->last_primary_launches = 0; ctx->last_shade_launches = 0;
        for (size_t i = 0; i + 2 < ctx->wf.ev_used; i += 3) {
            float e = 0.0f, sh = 0.0f;
            CTX_HIP(hipEventElapsedTime(&e, ctx->wf.ev[i], ctx->wf.ev[i + 1]));
            CTX_HIP(hipEventElapsedTime(&sh, ctx->wf.ev[i + 1], ctx->wf.ev[i + 2]));
            const int primary = i / 3 < ctx->wf.ev_primary.size() ? ctx->wf.ev_primary[i / 3] : 0;
            if (primary) { ctx->last_primary_ms += e; ctx->last_primary_launches++; }
            else { ctx->last_extend_ms += e; ctx->last_extend_launches++; }
            if (primary != 2) { ctx->last_shade_ms += sh; ctx->last_shade_launches++; }
        }
        ctx->wf.ev_used = 0;
    }
    return CRT_OK;
}

// Start of a dispatch: takes the next timing slot.  crt_dispatch / crt_render stay asynchronous (abi.h): the call waits
// only when the dispatch TWO calls back is still running (the reference's waitForFences on its frame slot), or when
// per-kernel timing was asked for (crt_set_kernel_timing: one set of k_wf_extend events, read before it is reused).
int begin_timing(crt_context *ctx, crt_context::TimingSlot **out) {
    int rc;
    if (ctx->wf.time_kernels && (rc = finish_timing(ctx))) return rc;
    crt_context::TimingSlot &sl = ctx->slot[ctx->slot_next];
    if ((rc = harvest_slot(ctx, sl, true))) return rc;                                   // the dispatch two calls back
    if ((rc = harvest_slot(ctx, ctx->slot[ctx->slot_next ^ 1u], false))) return rc;      // the previous one, if it is done already
    ctx->slot_next ^= 1u;
    CTX_HIP(hipEventRecord(sl.start, ctx->stream));
    *out = &sl;
    return CRT_OK;
}

// Whether the dispatches of this call end in k_wf_tail (crt_context::TailTune); paths = paths of one batch, iters = the
// iterations the call's time covers.
void pick_tail(crt_context *ctx, crt_context::TimingSlot *sl, double paths, uint32_t iters) {
    static const int env_bounce = [] { const char *e = getenv("CRT_WF_TAIL_BOUNCE"); return e ? atoi(e) : -100; }();      // unset: measure
    static const double max_paths = [] { const char *e = getenv("CRT_WF_TAIL_PATHS"); return e ? atof(e) : 4.2e6; }();
    static const bool env_persist = [] { const char *e = getenv("CRT_WF_PATH_PERSIST"); return e && atoi(e) != 0; }();
    ctx->wf.tail_bounce = -1; ctx->wf.tail_persist = false;
    sl->tune_mode = -1;
    if (ctx->cfg.tracer != CRT_TRACER_WAVEFRONT || ctx->instrumented || ctx->wf.time_kernels || paths > max_paths || iters == 0) return;
    auto set_mode = [&](int mode) { ctx->wf.tail_bounce = mode ? 0 : -1; ctx->wf.tail_persist = mode == 2; };
    const int n_modes = (ctx->build.n_wide != 0 && ctx->ext == 0) ? 3 : 2;
    if (ctx->cfg.flags & CRT_FLAG_PER_BOUNCE_KERNELS) return;
    if (ctx->cfg.flags & CRT_FLAG_PATH_KERNEL) { set_mode(1); return; }
    if (ctx->cfg.flags & CRT_FLAG_PATH_KERNEL_PERSISTENT) { set_mode(n_modes == 3 ? 2 : 1); return; }
    if (env_bounce != -100) { ctx->wf.tail_bounce = env_bounce < 0 ? -1 : env_bounce; ctx->wf.tail_persist = env_persist; return; }
    crt_context::TailTune &t = ctx->tune;
    if (t.decided >= 0) { set_mode(t.decided < n_modes ? t.decided : 1); return; }
    if (t.calls >= 2u * (uint32_t)n_modes) {
        // every form has had its two dispatches; until their times are read (one or two calls from now) render with the
        // best form measured so far instead of alternating on: an interactive engine sees six dispatches of jitter, not twenty
        int b = 0; for (int m = 1; m < n_modes; m++) if (t.best[m] < t.best[b]) b = m;
        set_mode(b);
        return;
    }
    const int mode = (int)(t.calls % (uint32_t)n_modes);
    t.calls++;
    set_mode(mode);
    sl->tune_mode = mode; sl->tune_modes = n_modes; sl->tune_scale = (float)iters;
}

// n_batch > 1 (wavefront tracer only): p.camera.iteration is the first iteration of the batch, seeds[b] its b-th seed.
int dispatch_one(crt_context *ctx, const crt_scene_params &p, uint32_t n_batch = 1, const float *seeds = nullptr) {
    std::string err;
    DeviceSceneHost sc = scene_view(ctx);
    FrameSetup fs = frame_setup(ctx);
    bool ok;
    if (ctx->cfg.tracer == CRT_TRACER_WAVEFRONT)
        ok = launch_wavefront(ctx->stream, sc, fs, p, n_batch, seeds, ctx->wf, (float *)ctx->accum.ptr,
                              (uint32_t)((size_t)ctx->local_rows * ctx->cfg.width), (uint32_t *)ctx->hit_ids.ptr,
                              (DeviceStats *)ctx->stats.ptr, err);
    else
        ok = launch_megakernel(ctx->stream, sc, fs, p, (float *)ctx->accum.ptr, (uint32_t *)ctx->hit_ids.ptr,
                               (DeviceStats *)ctx->stats.ptr, err);
    if (!ok) return fail(ctx, CRT_ERR_HIP, err);
    return CRT_OK;
}

int ready_to_render(crt_context *ctx) {
    if (!ctx->configured) return fail(ctx, CRT_ERR_INVALID, "crt_configure has not been called");
    if (!ctx->have_scene) return fail(ctx, CRT_ERR_INVALID, "crt_scene_upload has not been called");
    if (!ctx->have_bvh) return fail(ctx, CRT_ERR_INVALID, "crt_bvh_build has not been called for the current scene");
    if (!ctx->build.pairs_valid && ctx->cfg.tracer != CRT_TRACER_WAVEFRONT)
        return fail(ctx, CRT_ERR_INVALID, "no pair BVH for the megakernel tracer (a CRT_BVH_WIDE_DIRECT build has none, a refit leaves it stale, a SAH-refined tree deeper than its stacks is withheld): configure CRT_TRACER_WAVEFRONT");
    return CRT_OK;
}

}  // namespace

extern "C" {

const char *crt_version(void) { return "crt-mi355x 0.1 (gfx950)"; }

int crt_create(int device, crt_context **out) {
    if (!out) return CRT_ERR_INVALID;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return CRT_ERR_NO_DEVICE;
    if (device < 0 || device >= count) return CRT_ERR_NO_DEVICE;
    crt_context *ctx = new crt_context();
    ctx->device = device;
    if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreate(&ctx->slot[0].start) != hipSuccess || hipEventCreate(&ctx->slot[0].stop) != hipSuccess ||
        hipEventCreate(&ctx->slot[1].start) != hipSuccess || hipEventCreate(&ctx->slot[1].stop) != hipSuccess ||
        hipEventCreate(&ctx->ev_build[0]) != hipSuccess || hipEventCreate(&ctx->ev_build[1]) != hipSuccess) {
        delete ctx;
        return CRT_ERR_HIP;
    }
    if (hipMalloc(&ctx->stats.ptr, sizeof(DeviceStats)) != hipSuccess || hipMemset(ctx->stats.ptr, 0, sizeof(DeviceStats)) != hipSuccess) { delete ctx; return CRT_ERR_HIP; }
    ctx->stats.bytes = sizeof(DeviceStats);
    std::string err;
    if (!noalloc_init(device, err)) { hipError_t e = hipFree(ctx->stats.ptr); (void)e; delete ctx; return CRT_ERR_HIP; }      // this device's copy of the static scratch
    *out = ctx;
    return CRT_OK;
}

void crt_destroy(crt_context *ctx) {
    if (!ctx) return;
    if (hipSetDevice(ctx->device) == hipSuccess) {
        hipError_t e = hipStreamSynchronize(ctx->stream); (void)e;
        DeviceBuffer *all[] = {&ctx->spheres, &ctx->quads, &ctx->tris, &ctx->tris_rest, &ctx->materials, &ctx->tex0, &ctx->tex_fp, &ctx->tex_table, &ctx->dbg_boxes, &ctx->dbg_scan, &ctx->accum, &ctx->hit_ids, &ctx->stats, &ctx->refit_cost};
        for (DeviceBuffer *b : all) release(*b);
        release_export(ctx);
        free_build_state(ctx->build);
        free_wavefront(ctx->wf);
        comm_destroy(ctx->gather);
        for (auto &sl : ctx->slot) { e = hipEventDestroy(sl.start); e = hipEventDestroy(sl.stop); }
        e = hipEventDestroy(ctx->ev_build[0]); e = hipEventDestroy(ctx->ev_build[1]); e = hipStreamDestroy(ctx->stream); (void)e;
    }
    delete ctx;
}

const char *crt_last_error(const crt_context *ctx) { return ctx ? ctx->error.c_str() : "null context"; }

int crt_configure(crt_context *ctx, const crt_config *cfg) {
    int rc = bind(ctx); if (rc) return rc;
    if (!cfg || cfg->width == 0 || cfg->height == 0) return fail(ctx, CRT_ERR_INVALID, "crt_configure: width/height must be non-zero");
    if ((uint64_t)cfg->width * cfg->height >= (1ull << 31)) return fail(ctx, CRT_ERR_INVALID, "crt_configure: image too large");
    if (cfg->rng_mode > CRT_RNG_PCG || cfg->tracer > CRT_TRACER_WAVEFRONT) return fail(ctx, CRT_ERR_INVALID, "crt_configure: unknown rng_mode/tracer");
    if (cfg->band_world > 1 && cfg->band_rank >= cfg->band_world) return fail(ctx, CRT_ERR_INVALID, "crt_configure: band_rank >= band_world");
    CTX_HIP(hipStreamSynchronize(ctx->stream));
    if (ctx->configured && (cfg->width != ctx->cfg.width || cfg->height != ctx->cfg.height || cfg->band_world != ctx->cfg.band_world ||
                            cfg->band_rank != ctx->cfg.band_rank || cfg->tracer != ctx->cfg.tracer)) {
        // a different frame: the wavefront queues (up to ~16 GB for a full batch) are sized again by the next render,
        // so a context reconfigured to a smaller image does not keep the large frame's buffers
        if ((rc = finish_timing(ctx))) return rc;
        free_wavefront(ctx->wf);
    }
    ctx->cfg = *cfg;
    bool full = (cfg->flags & CRT_FLAG_FULL_IMAGE) != 0;
    ctx->grid_w = full ? cfg->width : (cfg->width / 8) * 8;      // dispatch(W/8, H/8, 1), vk_engine.cpp:1435
    ctx->grid_h = full ? cfg->height : (cfg->height / 8) * 8;
    uint32_t world = cfg->band_world > 1 ? cfg->band_world : 1, rank = cfg->band_world > 1 ? cfg->band_rank : 0;
    ctx->n_local_bands = local_bands(ctx->grid_h, rank, world);
    // the packed band buffer always holds whole 8-row bands; when world == 1 it is exactly the image
    // (rows above grid_h, if any, stay zero)
    uint32_t rows = ctx->n_local_bands * 8;
    if (world == 1 && rows < cfg->height) rows = cfg->height;
    ctx->local_rows = rows;
    size_t px = (size_t)rows * cfg->width;
    // the allocation is padded to the rows of the rank that owns the most bands (rank 0): crt_image_gather all-gathers
    // equal-sized buffers straight out of it; the padding rows stay zero
    const size_t px_alloc = (size_t)gather_max_rows(cfg->height, world) * cfg->width;
    rc = alloc(ctx, ctx->accum, (px_alloc > px ? px_alloc : px) * 16); if (rc) return rc;
    rc = alloc(ctx, ctx->hit_ids, px * 4); if (rc) return rc;
    CTX_HIP(hipMemsetAsync(ctx->accum.ptr, 0, (px_alloc > px ? px_alloc : px) * 16, ctx->stream));
    CTX_HIP(hipMemsetAsync(ctx->hit_ids.ptr, 0xFF, px * 4, ctx->stream));
    CTX_HIP(hipStreamSynchronize(ctx->stream));
    ctx->configured = true;
    ctx->tune = crt_context::TailTune();
    return CRT_OK;
}

int crt_scene_upload(crt_context *ctx, const crt_scene_desc *s) {
    int rc = bind(ctx); if (rc) return rc;
    if (!s) return fail(ctx, CRT_ERR_INVALID, "crt_scene_upload: null scene");
    if ((s->n_spheres && !s->spheres) || (s->n_quads && !s->quads) || (s->n_tris && !s->tris) || (s->n_materials && !s->materials) ||
        (s->n_textures && !s->textures))
        return fail(ctx, CRT_ERR_INVALID, "crt_scene_upload: a non-empty array has a null pointer");
    if (s->n_spheres >= (1u << 30) || s->n_quads >= (1u << 30) || s->n_tris >= (1u << 30) - 1)
        return fail(ctx, CRT_ERR_INVALID, "crt_scene_upload: more than 2^30 primitives of one kind");
    // every primitive's material index must be in range: the kernels index the material array unchecked
    for (uint32_t i = 0; i < s->n_spheres; i++) if (s->spheres[i].materialIndex >= s->n_materials) return fail(ctx, CRT_ERR_INVALID, "crt_scene_upload: sphere material index out of range");
    for (uint32_t i = 0; i < s->n_quads; i++) if (s->quads[i].materialIndex >= s->n_materials) return fail(ctx, CRT_ERR_INVALID, "crt_scene_upload: quad material index out of range");
    for (uint32_t i = 0; i < s->n_tris; i++) if (s->tris[i].materialIndex >= s->n_materials) return fail(ctx, CRT_ERR_INVALID, "crt_scene_upload: triangle material index out of range");
    for (uint32_t i = 0; i < s->n_textures; i++)
        if (!s->textures[i].rgba8 || s->textures[i].width == 0 || s->textures[i].height == 0) return fail(ctx, CRT_ERR_INVALID, "crt_scene_upload: empty texture");
    bool tex_index_ok = true;
    for (uint32_t i = 0; i < s->n_materials; i++)
        if (s->materials[i].textureIndex != CRT_BAD_INDEX && s->materials[i].textureIndex >= s->n_textures) tex_index_ok = false;
    if ((ctx->ext & CRT_EXT_TEXTURE_INDEX) && !tex_index_ok)
        return fail(ctx, CRT_ERR_INVALID, "crt_scene_upload: a material's textureIndex is past the texture array (CRT_EXT_TEXTURE_INDEX is set)");
    CTX_HIP(hipStreamSynchronize(ctx->stream));
    if ((rc = finish_build(ctx))) { ctx->error.clear(); }      // a pending table of the old scene: nobody needs its verdict any more
    ctx->have_scene = false; ctx->have_bvh = false; ctx->animated = false; ctx->n_dbg_boxes = 0;
    ctx->build.direct_hint_levels = 0; ctx->build.direct_hint_n = 0; ctx->build.records_fresh = false;
    if ((rc = upload(ctx, ctx->spheres, s->spheres, (size_t)s->n_spheres * sizeof(crt_sphere)))) return rc;
    if ((rc = upload(ctx, ctx->quads, s->quads, (size_t)s->n_quads * sizeof(crt_quad)))) return rc;
    if ((rc = upload(ctx, ctx->tris, s->tris, (size_t)s->n_tris * sizeof(crt_tri)))) return rc;
    if ((rc = upload(ctx, ctx->materials, s->materials, (size_t)s->n_materials * sizeof(crt_material)))) return rc;
    ctx->n_spheres = s->n_spheres; ctx->n_quads = s->n_quads; ctx->n_tris = s->n_tris; ctx->n_materials = s->n_materials;
    ctx->n_textures = s->n_textures; ctx->tex0_w = ctx->tex0_h = 0;
    ctx->tex_index_ok = tex_index_ok;
    if (s->n_textures) {
        // The shader samples textures[0] only (compute.comp:386-387); the others are resident behind it, in one allocation,
        // for CRT_EXT_TEXTURE_INDEX.  tex_table[i] = {first texel, width, height, 0}.
        std::vector<uint32_t> table;
        try { table.resize(4 * (size_t)s->n_textures); }
        catch (const std::bad_alloc &) { return fail(ctx, CRT_ERR_INVALID, "crt_scene_upload: out of host memory for the texture table"); }
        size_t texels = 0;
        for (uint32_t i = 0; i < s->n_textures; i++) {
            const crt_texture &t = s->textures[i];
            if (texels + (size_t)t.width * t.height > 0xFFFFFFFFull) return fail(ctx, CRT_ERR_INVALID, "crt_scene_upload: more than 2^32 texels of textures");
            table[4 * i] = (uint32_t)texels; table[4 * i + 1] = t.width; table[4 * i + 2] = t.height; table[4 * i + 3] = 0;
            texels += (size_t)t.width * t.height;
        }
        if ((rc = alloc(ctx, ctx->tex0, texels * 4))) return rc;
        for (uint32_t i = 0; i < s->n_textures; i++) {
            const crt_texture &t = s->textures[i];
            CTX_HIP(hipMemcpy((uint8_t *)ctx->tex0.ptr + 4 * (size_t)table[4 * i], t.rgba8, (size_t)t.width * t.height * 4, hipMemcpyHostToDevice));   // synchronous: the caller's pixels may go away after this call
        }
        if ((rc = upload(ctx, ctx->tex_table, table.data(), table.size() * 4))) return rc;
        if ((rc = alloc(ctx, ctx->tex_fp, texels * 16))) return rc;
        hipLaunchKernelGGL(k_tex_footprints, dim3((uint32_t)((texels + 255) / 256)), dim3(256), 0, ctx->stream, (const uint32_t *)ctx->tex0.ptr,
                           (const uint4 *)ctx->tex_table.ptr, s->n_textures, (uint32_t)texels, (uint4 *)ctx->tex_fp.ptr);
        CTX_HIP(hipGetLastError());
        CTX_HIP(hipStreamSynchronize(ctx->stream));
        release(ctx->tex0);      // the plain texels were staging only: the kernels sample the footprints (16 bytes per texel resident)
        ctx->tex0_w = s->textures[0].width; ctx->tex0_h = s->textures[0].height;
    }
    ctx->have_scene = true;
    ctx->tune = crt_context::TailTune();
    return CRT_OK;
}

int crt_extensions_set(crt_context *ctx, uint32_t ext_flags) {
    int rc = bind(ctx); if (rc) return rc;
    if (ext_flags & ~(uint32_t)(CRT_EXT_TEXTURE_INDEX | CRT_EXT_ALPHA_TEST)) return fail(ctx, CRT_ERR_INVALID, "crt_extensions_set: unknown extension bit");
    if ((ext_flags & CRT_EXT_TEXTURE_INDEX) && ctx->have_scene && !ctx->tex_index_ok)
        return fail(ctx, CRT_ERR_INVALID, "crt_extensions_set: a material's textureIndex is past the texture array");
    ctx->ext = ext_flags;
    return CRT_OK;
}

int crt_bvh_build(crt_context *ctx, uint32_t flags, float *build_ms) {
    int rc = bind(ctx); if (rc) return rc;
    if (!ctx->have_scene) return fail(ctx, CRT_ERR_INVALID, "crt_bvh_build: no scene uploaded");
    // build_ms == NULL with CRT_BVH_WIDE_DIRECT: the call may only enqueue (abi.h); everything else waits as before
    const bool may_skip_waits = build_ms == nullptr && (flags & CRT_BVH_WIDE_DIRECT) != 0 && !ctx->wf.time_kernels;
    if (may_skip_waits) {
        if ((rc = finish_build(ctx))) return rc;      // the previous build's table (it finished long ago: a dispatch ran after it)
        for (uint32_t k = 0; k < 2; k++) if ((rc = harvest_slot(ctx, ctx->slot[(ctx->slot_next + k) & 1u], false))) return rc;
    } else { rc = finish_timing(ctx); if (rc) return rc; }
    BuildInput in{(const crt_sphere *)ctx->spheres.ptr, ctx->n_spheres, (const crt_quad *)ctx->quads.ptr, ctx->n_quads,
                  (const crt_tri *)ctx->tris.ptr, ctx->n_tris};
    std::string err;
    if (const char *e = getenv("CRT_BVH_MAX_LEAF")) ctx->build.max_leaf = (uint32_t)atoi(e);      // experiment knobs
    if (const char *e = getenv("CRT_BVH_CI")) ctx->build.c_isect = (float)atof(e);
    if (const char *e = getenv("CRT_BVH_CT")) ctx->build.c_trav = (float)atof(e);
    hipEvent_t e0 = ctx->ev_build[0], e1 = ctx->ev_build[1];      // owned by the context: no event is created (or leaked) per build
    ctx->have_bvh = false;
    ctx->build.async_request = may_skip_waits;
    CTX_HIP(hipEventRecord(e0, ctx->stream));
    bool ok = build_bvh(ctx->stream, in, flags, ctx->build, err);
    CTX_HIP(hipEventRecord(e1, ctx->stream));
    ctx->build.async_request = false;
    if (!ok) return fail(ctx, CRT_ERR_HIP, "crt_bvh_build: " + err);
    ctx->have_bvh = true;
    if (ctx->build.counts_pending) {      // the build only enqueued: time and counts at the next blocking call
        ctx->build_time_pending = true;
        return CRT_OK;
    }
    CTX_HIP(hipEventSynchronize(e1));
    float ms = 0.0f;
    CTX_HIP(hipEventElapsedTime(&ms, e0, e1));
    ctx->last_build_ms = ms;
    ctx->build_time_pending = false;
    if (build_ms) *build_ms = ms;
    if (ctx->build.sort_error_dev) {      // the hand-written key sort of a static build: did a look-back wait give up? (never seen)
        uint32_t flag = 0;
        CTX_HIP(hipMemcpy(&flag, ctx->build.sort_error_dev, 4, hipMemcpyDeviceToHost));
        ctx->build.sort_error_dev = nullptr;
        if (flag) { ctx->have_bvh = false; return fail(ctx, CRT_ERR_HIP, "crt_bvh_build: the key sort's look-back wait gave up"); }
    }
    return CRT_OK;
}

int crt_bvh_refit(crt_context *ctx, float *cost) {
    int rc = bind(ctx); if (rc) return rc;
    if (!ctx->have_scene) return fail(ctx, CRT_ERR_INVALID, "crt_bvh_refit: no scene uploaded");
    if ((rc = finish_build(ctx))) return rc;
    if (ctx->build.n_wide == 0) return fail(ctx, CRT_ERR_INVALID, "crt_bvh_refit: no 8-wide tree (build with CRT_BVH_WIDE or CRT_BVH_WIDE_DIRECT first)");
    if (ctx->build.n_prims != ctx->n_spheres + ctx->n_quads + ctx->n_tris) return fail(ctx, CRT_ERR_INVALID, "crt_bvh_refit: the scene changed size since the build");
    BuildInput in{(const crt_sphere *)ctx->spheres.ptr, ctx->n_spheres, (const crt_quad *)ctx->quads.ptr, ctx->n_quads,
                  (const crt_tri *)ctx->tris.ptr, ctx->n_tris};
    if (cost && (rc = alloc(ctx, ctx->refit_cost, 256))) return rc;
    std::string err;
    ctx->have_bvh = false;
    if (!refit_wide(ctx->stream, in, ctx->build, cost ? (float *)ctx->refit_cost.ptr : nullptr, err)) return fail(ctx, CRT_ERR_HIP, "crt_bvh_refit: " + err);
    ctx->build.pairs_valid = false;      // the pair tree (megakernel tracer, crt_bvh_read) still has the old boxes
    ctx->have_bvh = true;
    if (cost) {
        float h[2] = {0.0f, 0.0f};
        CTX_HIP(hipMemcpyAsync(h, ctx->refit_cost.ptr, 8, hipMemcpyDeviceToHost, ctx->stream));
        CTX_HIP(hipStreamSynchronize(ctx->stream));
        *cost = h[1] > 0.0f ? h[0] / h[1] : 0.0f;
    }
    return CRT_OK;
}

static int debug_sort(crt_context *ctx, const char *what, uint64_t *keys, uint32_t *vals, uint32_t n, uint32_t bit_lo, uint32_t bit_hi) {
    int rc = bind(ctx); if (rc) return rc;
    if (n && (!keys)) return fail(ctx, CRT_ERR_INVALID, std::string(what) + ": null keys");
    if (bit_lo > bit_hi || bit_hi > 64) return fail(ctx, CRT_ERR_INVALID, std::string(what) + ": bit range");
    if (n == 0) return CRT_OK;
    DeviceBuffer a, b, va, vb, scratch;
    std::string err;
    rc = alloc(ctx, a, (size_t)n * 8); if (!rc) rc = alloc(ctx, b, (size_t)n * 8); if (!rc) rc = alloc(ctx, scratch, radix_sort_scratch_bytes(n));
    if (!rc && vals) { rc = alloc(ctx, va, (size_t)n * 4); if (!rc) rc = alloc(ctx, vb, (size_t)n * 4); }
    const uint64_t *sorted = nullptr; const uint32_t *sorted_vals = nullptr, *ew = nullptr;
    uint32_t flag = 0;
    bool ok = false;
    if (!rc && hipMemcpy(a.ptr, keys, (size_t)n * 8, hipMemcpyHostToDevice) == hipSuccess &&
        (!vals || hipMemcpy(va.ptr, vals, (size_t)n * 4, hipMemcpyHostToDevice) == hipSuccess) &&
        radix_sort_u64(ctx->stream, (uint64_t *)a.ptr, (uint64_t *)b.ptr, (uint32_t *)va.ptr, (uint32_t *)vb.ptr, n, bit_lo, bit_hi, scratch.ptr, &sorted, &sorted_vals, &ew, err) &&
        hipStreamSynchronize(ctx->stream) == hipSuccess && hipMemcpy(keys, sorted, (size_t)n * 8, hipMemcpyDeviceToHost) == hipSuccess &&
        (!vals || hipMemcpy(vals, sorted_vals, (size_t)n * 4, hipMemcpyDeviceToHost) == hipSuccess) &&
        (!ew || hipMemcpy(&flag, ew, 4, hipMemcpyDeviceToHost) == hipSuccess)) ok = true;
    release(a); release(b); release(va); release(vb); release(scratch);
    if (rc) return rc;
    if (!ok) return fail(ctx, CRT_ERR_HIP, std::string(what) + ": " + (err.empty() ? std::string("HIP call failed") : err));
    if (flag) return fail(ctx, CRT_ERR_HIP, std::string(what) + ": a look-back wait gave up");
    return CRT_OK;
}

int crt_debug_sort_u64(crt_context *ctx, uint64_t *keys, uint32_t n, uint32_t bit_lo, uint32_t bit_hi) {
    return debug_sort(ctx, "crt_debug_sort_u64", keys, nullptr, n, bit_lo, bit_hi);
}

int crt_debug_sort_pairs_u64(crt_context *ctx, uint64_t *keys, uint32_t *vals, uint32_t n, uint32_t bit_lo, uint32_t bit_hi) {
    if (n && !vals) return fail(ctx, CRT_ERR_INVALID, "crt_debug_sort_pairs_u64: null values");
    return debug_sort(ctx, "crt_debug_sort_pairs_u64", keys, vals, n, bit_lo, bit_hi);
}

int crt_debug_math(crt_context *ctx, int fn, uint32_t n, const float *a, const float *b, float *out0, float *out1) {
    int rc = bind(ctx); if (rc) return rc;
    if (n && (!a || !b || !out0 || !out1)) return fail(ctx, CRT_ERR_INVALID, "crt_debug_math: null array");
    if (n == 0) return CRT_OK;
    DeviceBuffer da, db, d0, d1;
    const size_t bytes = (size_t)n * 4;
    rc = alloc(ctx, da, bytes); if (!rc) rc = alloc(ctx, db, bytes); if (!rc) rc = alloc(ctx, d0, bytes); if (!rc) rc = alloc(ctx, d1, bytes);
    std::string err;
    bool ok = false;
    if (!rc && hipMemcpy(da.ptr, a, bytes, hipMemcpyHostToDevice) == hipSuccess && hipMemcpy(db.ptr, b, bytes, hipMemcpyHostToDevice) == hipSuccess &&
        debug_math(ctx->stream, fn, n, (const float *)da.ptr, (const float *)db.ptr, (float *)d0.ptr, (float *)d1.ptr, err) &&
        hipStreamSynchronize(ctx->stream) == hipSuccess && hipMemcpy(out0, d0.ptr, bytes, hipMemcpyDeviceToHost) == hipSuccess &&
        hipMemcpy(out1, d1.ptr, bytes, hipMemcpyDeviceToHost) == hipSuccess) ok = true;
    release(da); release(db); release(d0); release(d1);
    if (rc) return rc;
    if (!ok) return fail(ctx, CRT_ERR_HIP, "crt_debug_math: " + (err.empty() ? std::string("HIP call failed") : err));
    return CRT_OK;
}

int crt_retune(crt_context *ctx) {
    if (!ctx) return CRT_ERR_INVALID;
    ctx->tune = crt_context::TailTune();
    return CRT_OK;
}

int crt_poll(crt_context *ctx, int *done) {
    int rc = bind(ctx); if (rc) return rc;
    if (!done) return fail(ctx, CRT_ERR_INVALID, "crt_poll: null output");
    const hipError_t q = hipStreamQuery(ctx->stream);
    if (q != hipSuccess && q != hipErrorNotReady) return fail(ctx, CRT_ERR_HIP, std::string("hipStreamQuery: ") + hipGetErrorString(q));
    *done = q == hipSuccess ? 1 : 0;
    return CRT_OK;
}

int crt_tris_update(crt_context *ctx, const crt_tri *tris, uint32_t first, uint32_t count) {
    int rc = bind(ctx); if (rc) return rc;
    if (!ctx->have_scene) return fail(ctx, CRT_ERR_INVALID, "crt_tris_update: no scene uploaded");
    if (!tris || (uint64_t)first + count > ctx->n_tris) return fail(ctx, CRT_ERR_INVALID, "crt_tris_update: range outside the uploaded triangles");
    CTX_HIP(hipStreamSynchronize(ctx->stream));
    ctx->build.records_fresh = false;
    CTX_HIP(hipMemcpy((crt_tri *)ctx->tris.ptr + first, tris, (size_t)count * sizeof(crt_tri), hipMemcpyDefault));
    // after crt_tris_animate the rest pose is a second copy: the update replaces the triangles there too, so the next
    // crt_tris_animate displaces the NEW vertices instead of silently restoring the old ones
    if (ctx->animated) CTX_HIP(hipMemcpy((crt_tri *)ctx->tris_rest.ptr + first, tris, (size_t)count * sizeof(crt_tri), hipMemcpyDefault));
    ctx->have_bvh = false;
    // The new triangles may live in device memory, so the device checks them: material indices in range (the kernels
    // index the material array unchecked).
    if (count) {
        uint32_t *flags = (uint32_t *)ctx->dbg_scan.ptr;
        if (!flags) { if ((rc = alloc(ctx, ctx->dbg_scan, 256))) return rc; flags = (uint32_t *)ctx->dbg_scan.ptr; }
        CTX_HIP(hipMemsetAsync(flags, 0, 8, ctx->stream));
        hipLaunchKernelGGL(k_tri_material_scan, dim3((count + 255) / 256), dim3(256), 0, ctx->stream, (const crt_tri *)ctx->tris.ptr, first, count,
                           ctx->n_materials, flags);
        uint32_t host_flags[2] = {0, 0};
        CTX_HIP(hipMemcpyAsync(host_flags, flags, 8, hipMemcpyDeviceToHost, ctx->stream));
        CTX_HIP(hipStreamSynchronize(ctx->stream));
        if (host_flags[0]) {
            ctx->have_scene = false;
            return fail(ctx, CRT_ERR_INVALID, "crt_tris_update: triangle material index out of range (the scene is no longer usable: upload it again)");
        }
    }
    return CRT_OK;
}

int crt_tris_read(crt_context *ctx, crt_tri *tris, uint32_t first, uint32_t count) {
    int rc = bind(ctx); if (rc) return rc;
    if (!ctx->have_scene) return fail(ctx, CRT_ERR_INVALID, "crt_tris_read: no scene uploaded");
    if ((count && !tris) || (uint64_t)first + count > ctx->n_tris) return fail(ctx, CRT_ERR_INVALID, "crt_tris_read: range outside the uploaded triangles");
    CTX_HIP(hipStreamSynchronize(ctx->stream));
    if (count) CTX_HIP(hipMemcpy(tris, (const crt_tri *)ctx->tris.ptr + first, (size_t)count * sizeof(crt_tri), hipMemcpyDefault));
    return CRT_OK;
}

int crt_tris_animate(crt_context *ctx, float frame, float amplitude) {
    int rc = bind(ctx); if (rc) return rc;
    if (!ctx->have_scene) return fail(ctx, CRT_ERR_INVALID, "crt_tris_animate: no scene uploaded");
    size_t bytes = (size_t)ctx->n_tris * sizeof(crt_tri);
    if (!ctx->animated) {      // keep the rest pose
        rc = alloc(ctx, ctx->tris_rest, bytes); if (rc) return rc;
        CTX_HIP(hipMemcpyAsync(ctx->tris_rest.ptr, ctx->tris.ptr, bytes, hipMemcpyDeviceToDevice, ctx->stream));
        ctx->animated = true;
    }
    std::string err;
    if (!animate_tris(ctx->stream, ctx->n_tris, (const crt_tri *)ctx->tris_rest.ptr, (crt_tri *)ctx->tris.ptr, frame, amplitude,
                      (ctx->n_spheres == 0 && ctx->n_quads == 0) ? &ctx->build : nullptr, err))
        return fail(ctx, CRT_ERR_HIP, "crt_tris_animate: " + err);
    ctx->have_bvh = false;
    return CRT_OK;
}

int crt_dispatch(crt_context *ctx, const crt_scene_params *params) {
    int rc = bind(ctx); if (rc) return rc;
    if (!params) return fail(ctx, CRT_ERR_INVALID, "crt_dispatch: null params");
    if ((rc = ready_to_render(ctx))) return rc;
    crt_context::TimingSlot *sl = nullptr;
    if ((rc = begin_timing(ctx, &sl))) return rc;
    pick_tail(ctx, sl, (double)ctx->n_local_bands * 8.0 * ctx->grid_w, 1);
    rc = dispatch_one(ctx, *params);
    CTX_HIP(hipEventRecord(sl->stop, ctx->stream));
    sl->pending = true;
    return rc;
}

int crt_render(crt_context *ctx, const crt_scene_params *params, uint32_t first_iteration, uint32_t n_iterations, const float *seeds) {
    int rc = bind(ctx); if (rc) return rc;
    if (!params || (n_iterations && !seeds)) return fail(ctx, CRT_ERR_INVALID, "crt_render: null params/seeds");
    if ((rc = ready_to_render(ctx))) return rc;
    crt_context::TimingSlot *sl = nullptr;
    if ((rc = begin_timing(ctx, &sl))) return rc;
    crt_scene_params p = *params;
    // The wavefront tracer traces several iterations of a small frame together (their paths share the queues, the
    // accumulation still happens in iteration order, so the image is bit-identical): a launch needs a few million
    // rays to fill 256 CUs, which one sample of a 1080p frame, or of one rank's bands of a 4K frame, does not supply.
    uint32_t batch = 1;
    if (ctx->cfg.tracer == CRT_TRACER_WAVEFRONT) {
        static const double target = [] { const char *e = getenv("CRT_WF_BATCH_PATHS"); return e ? atof(e) : 134.0e6; }();
        const double per_iter = (double)ctx->n_local_bands * 8.0 * ctx->grid_w;
        batch = per_iter > 0.0 ? (uint32_t)(target / per_iter) : 1u;
        batch = batch < 1u ? 1u : batch > 64u ? 64u : batch;
        // ... and by memory: a batch of b iterations holds 104 B of queues + hit record per path slot and 16 B of radiance
        // per pixel and iteration.  At most half of what is free now plus what the tracer already holds.
        size_t free_b = 0, total_b = 0;
        CTX_HIP(hipMemGetInfo(&free_b, &total_b));
        const double budget = 0.5 * ((double)free_b + (double)wavefront_bytes(ctx->wf));
        const double per_b = ((double)ctx->n_local_bands * 8.0 * (((double)ctx->grid_w + 7.0) / 8.0) * 8.0) * 104.0 + (double)ctx->local_rows * ctx->cfg.width * 16.0;
        if (per_b > 0.0 && batch * per_b > budget) batch = (uint32_t)(budget / per_b) < 1u ? 1u : (uint32_t)(budget / per_b);
        if ((size_t)ctx->local_rows * ctx->cfg.width >= (1u << 26) - 1u) batch = 1;      // ray_d.w packs batch index << 26 | pixel slot
    }
    pick_tail(ctx, sl, (double)ctx->n_local_bands * 8.0 * ctx->grid_w * (double)(n_iterations < batch ? n_iterations : batch), n_iterations);
    for (uint32_t i = 0; i < n_iterations && rc == CRT_OK; i += batch) {
        const uint32_t nb = n_iterations - i < batch ? n_iterations - i : batch;
        p.camera.iteration = (float)(first_iteration + i);
        p.camera.seed = seeds[i];
        rc = dispatch_one(ctx, p, nb, seeds + i);
    }
    CTX_HIP(hipEventRecord(sl->stop, ctx->stream));
    sl->pending = true;
    return rc;
}

int crt_rays_query(crt_context *ctx, const crt_ray *rays, uint32_t n_rays, uint32_t mode, crt_ray_hit *hits) {
    int rc = bind(ctx); if (rc) return rc;
    if (!ctx->have_scene) return fail(ctx, CRT_ERR_INVALID, "crt_rays_query: crt_scene_upload has not been called");
    if (!ctx->have_bvh) return fail(ctx, CRT_ERR_INVALID, "crt_rays_query: crt_bvh_build has not been called for the current scene");
    if (mode > CRT_QUERY_ANY) return fail(ctx, CRT_ERR_INVALID, "crt_rays_query: unknown mode");
    if (n_rays && (!rays || !hits)) return fail(ctx, CRT_ERR_INVALID, "crt_rays_query: null rays/hits");
    if ((rc = finish_timing(ctx))) return rc;
    std::string err;
    FrameSetup fs{};
    fs.small_stack = (ctx->configured && (ctx->cfg.flags & CRT_FLAG_DEBUG_SMALL_STACK)) ? 1u : 0u;
    if (!launch_ray_query(ctx->stream, scene_view(ctx), fs, ctx->wf, rays, n_rays, mode, hits, (DeviceStats *)ctx->stats.ptr, err))
        return fail(ctx, CRT_ERR_HIP, "crt_rays_query: " + err);
    return CRT_OK;
}

int crt_rays_query_noalloc(crt_context *ctx, const crt_ray *rays_dev, uint32_t n_rays, uint32_t mode, crt_ray_hit *hits_dev) {
    int rc = bind(ctx); if (rc) return rc;
    if (!ctx->have_scene) return fail(ctx, CRT_ERR_INVALID, "crt_rays_query_noalloc: crt_scene_upload has not been called");
    if (!ctx->have_bvh) return fail(ctx, CRT_ERR_INVALID, "crt_rays_query_noalloc: crt_bvh_build has not been called for the current scene");
    if (mode > CRT_QUERY_ANY) return fail(ctx, CRT_ERR_INVALID, "crt_rays_query_noalloc: unknown mode");
    if (n_rays > CRT_NOALLOC_MAX_RAYS) return fail(ctx, CRT_ERR_INVALID, "crt_rays_query_noalloc: more than CRT_NOALLOC_MAX_RAYS (131072) rays in one call");
    if (n_rays == 0) return CRT_OK;
    if (!rays_dev || !hits_dev) return fail(ctx, CRT_ERR_INVALID, "crt_rays_query_noalloc: null rays/hits");
    if ((uintptr_t)rays_dev % 16 != 0 || (uintptr_t)hits_dev % 8 != 0) return fail(ctx, CRT_ERR_INVALID, "crt_rays_query_noalloc: rays must be 16-byte and hits 8-byte aligned");
    // the kernels dereference both pointers: host memory here would be a GPU memory fault, not an error code
    const void *ptrs[2] = {rays_dev, hits_dev};
    const size_t need[2] = {(size_t)n_rays * sizeof(crt_ray), (size_t)n_rays * sizeof(crt_ray_hit)};
    for (int k = 0; k < 2; k++) {
        // ... and the allocation behind each pointer must hold n_rays records from there on (a query, allocates nothing)
        hipDeviceptr_t base = nullptr; size_t size = 0;
        if (hipMemGetAddressRange(&base, &size, (hipDeviceptr_t)ptrs[k]) != hipSuccess) {
            hipError_t e = hipGetLastError(); (void)e;
            return fail(ctx, CRT_ERR_INVALID, "crt_rays_query_noalloc: rays and hits must be device memory");
        }
        const size_t off = (size_t)((const char *)ptrs[k] - (const char *)base);
        if (off > size || size - off < need[k])
            return fail(ctx, CRT_ERR_INVALID, k == 0 ? "crt_rays_query_noalloc: the rays allocation holds fewer than n_rays rays from rays_dev on"
                                                     : "crt_rays_query_noalloc: the hits allocation holds fewer than n_rays hit records from hits_dev on");
    }
    for (const void *p : ptrs) {
        hipPointerAttribute_t attr{};
        if (hipPointerGetAttributes(&attr, p) != hipSuccess || (attr.type != hipMemoryTypeDevice && attr.type != hipMemoryTypeManaged)) {
            hipError_t e = hipGetLastError(); (void)e;     // clear the sticky error of a failed attribute query
            return fail(ctx, CRT_ERR_INVALID, "crt_rays_query_noalloc: rays and hits must be device memory");
        }
        if (attr.type == hipMemoryTypeDevice && attr.device != ctx->device)
            return fail(ctx, CRT_ERR_INVALID, "crt_rays_query_noalloc: rays and hits must live on the context's GPU");
    }
    std::string err;
    FrameSetup fs{};
    fs.small_stack = (ctx->configured && (ctx->cfg.flags & CRT_FLAG_DEBUG_SMALL_STACK)) ? 1u : 0u;
    if (!launch_ray_query_noalloc(ctx->stream, ctx->device, scene_view(ctx), fs, rays_dev, n_rays, mode, hits_dev, (DeviceStats *)ctx->stats.ptr, err))
        return fail(ctx, CRT_ERR_HIP, "crt_rays_query_noalloc: " + err);
    return CRT_OK;
}

int crt_sync(crt_context *ctx) {
    int rc = bind(ctx); if (rc) return rc;
    CTX_HIP(hipStreamSynchronize(ctx->stream));
    return finish_timing(ctx);
}

uint32_t crt_band_rows(uint32_t height, uint32_t band_rank, uint32_t band_world, uint32_t *rows) {
    uint32_t world = band_world > 1 ? band_world : 1, rank = band_world > 1 ? band_rank : 0, n = 0;
    for (uint32_t y = 0; y < height; y++)
        if ((y / 8) % world == rank) { if (rows) rows[n] = y; n++; }
    return n;
}

static int read_banded(crt_context *ctx, const DeviceBuffer &src, void *dst, size_t elem_bytes, int fill) {
    uint32_t W = ctx->cfg.width, H = ctx->cfg.height;
    uint32_t world = ctx->cfg.band_world > 1 ? ctx->cfg.band_world : 1, rank = ctx->cfg.band_world > 1 ? ctx->cfg.band_rank : 0;
    size_t row_bytes = (size_t)W * elem_bytes;
    CTX_HIP(hipStreamSynchronize(ctx->stream));
    if (world == 1) {
        CTX_HIP(hipMemcpy(dst, src.ptr, row_bytes * H, hipMemcpyDeviceToHost));
        return CRT_OK;
    }
    std::vector<uint8_t> tmp(row_bytes * ctx->local_rows);
    if (!tmp.empty()) CTX_HIP(hipMemcpy(tmp.data(), src.ptr, tmp.size(), hipMemcpyDeviceToHost));
    memset(dst, fill, row_bytes * H);
    for (uint32_t y = 0; y < H; y++) {
        uint32_t band = y / 8;
        if (band % world != rank) continue;
        uint32_t prow = (band / world) * 8 + (y % 8);
        if (prow < ctx->local_rows) memcpy((uint8_t *)dst + row_bytes * y, tmp.data() + row_bytes * prow, row_bytes);
    }
    return CRT_OK;
}

int crt_image_read(crt_context *ctx, float *rgba, size_t n_floats) {
    int rc = bind(ctx); if (rc) return rc;
    if (!ctx->configured) return fail(ctx, CRT_ERR_INVALID, "crt_image_read: not configured");
    if (!rgba || n_floats != (size_t)ctx->cfg.width * ctx->cfg.height * 4) return fail(ctx, CRT_ERR_INVALID, "crt_image_read: buffer must hold width*height*4 floats");
    rc = read_banded(ctx, ctx->accum, rgba, 16, 0);
    if (rc) return rc;
    return finish_timing(ctx);
}

int crt_hit_ids_read(crt_context *ctx, uint32_t *ids, size_t n_ids) {
    int rc = bind(ctx); if (rc) return rc;
    if (!ctx->configured) return fail(ctx, CRT_ERR_INVALID, "crt_hit_ids_read: not configured");
    if (!ids || n_ids != (size_t)ctx->cfg.width * ctx->cfg.height) return fail(ctx, CRT_ERR_INVALID, "crt_hit_ids_read: buffer must hold width*height ids");
    rc = read_banded(ctx, ctx->hit_ids, ids, 4, 0xFF);
    if (rc) return rc;
    return finish_timing(ctx);
}

int crt_image_device_ptr(crt_context *ctx, void **device_ptr, size_t *n_bytes) {
    if (!ctx || !device_ptr || !n_bytes) return CRT_ERR_INVALID;
    if (!ctx->configured) return fail(ctx, CRT_ERR_INVALID, "crt_image_device_ptr: not configured");
    *device_ptr = ctx->accum.ptr;
    *n_bytes = (size_t)ctx->local_rows * ctx->cfg.width * 16;
    return CRT_OK;
}

int crt_image_copy_to_device(crt_context *ctx, void *dst_device, size_t n_bytes) {
    int rc = bind(ctx); if (rc) return rc;
    if (!ctx->configured) return fail(ctx, CRT_ERR_INVALID, "crt_image_copy_to_device: not configured");
    if (!dst_device || n_bytes != (size_t)ctx->local_rows * ctx->cfg.width * 16)
        return fail(ctx, CRT_ERR_INVALID, "crt_image_copy_to_device: destination must hold exactly the band buffer");
    CTX_HIP(hipMemcpyAsync(dst_device, ctx->accum.ptr, n_bytes, hipMemcpyDeviceToDevice, ctx->stream));
    CTX_HIP(hipStreamSynchronize(ctx->stream));
    return CRT_OK;
}

int crt_image_export_fd(crt_context *ctx, int *fd, size_t *image_bytes, size_t *alloc_bytes) {
    int rc = bind(ctx); if (rc) return rc;
    if (!ctx->configured) return fail(ctx, CRT_ERR_INVALID, "crt_image_export_fd: not configured");
    const size_t need = (size_t)ctx->local_rows * ctx->cfg.width * 16;
    if (need == 0) return fail(ctx, CRT_ERR_INVALID, "crt_image_export_fd: this rank owns no rows of the frame");
    crt_context::ExportImage &x = ctx->exported;
    if (x.live && x.image_bytes != need) {      // a different frame size: importers hold their own reference to the old memory
        CTX_HIP(hipStreamSynchronize(ctx->stream));
        release_export(ctx);
    }
    if (!x.live) {
        hipMemAllocationProp prop{};
        prop.type = hipMemAllocationTypePinned;
        prop.requestedHandleType = hipMemHandleTypePosixFileDescriptor;
        prop.location.type = hipMemLocationTypeDevice; prop.location.id = ctx->device;
        size_t gran = 0;
        CTX_HIP(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum));
        if (gran == 0) gran = 1u << 21;
        const size_t bytes = (need + gran - 1) / gran * gran;
        hipMemGenericAllocationHandle_t h{};
        CTX_HIP(hipMemCreate(&h, bytes, &prop, 0));
        void *va = nullptr;
        hipError_t e = hipMemAddressReserve(&va, bytes, gran, nullptr, 0);
        if (e == hipSuccess) {
            e = hipMemMap(va, bytes, 0, h, 0);
            if (e == hipSuccess) {
                hipMemAccessDesc acc{};
                acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
                e = hipMemSetAccess(va, bytes, &acc, 1);
                if (e != hipSuccess) { hipError_t e2 = hipMemUnmap(va, bytes); (void)e2; }
            }
            if (e != hipSuccess) { hipError_t e2 = hipMemAddressFree(va, bytes); (void)e2; }
        }
        if (e != hipSuccess) {
            hipError_t e2 = hipMemRelease(h); (void)e2;
            return fail(ctx, CRT_ERR_HIP, std::string("crt_image_export_fd: mapping the exportable allocation: ") + hipGetErrorString(e));
        }
        x.handle = h; x.va = va; x.alloc_bytes = bytes; x.image_bytes = need; x.live = true;
    }
    CTX_HIP(hipMemcpyAsync(x.va, ctx->accum.ptr, need, hipMemcpyDeviceToDevice, ctx->stream));
    if (fd) {
        int h = -1;
        CTX_HIP(hipMemExportToShareableHandle(&h, x.handle, hipMemHandleTypePosixFileDescriptor, 0));
        *fd = h;
    }
    if (image_bytes) *image_bytes = need;
    if (alloc_bytes) *alloc_bytes = x.alloc_bytes;
    return CRT_OK;
}

int crt_comm_unique_id(void *id) {
    if (!id) return CRT_ERR_INVALID;
    std::string err;
    if (comm_unique_id(id, err)) return CRT_OK;
    set_host_error("crt_comm_unique_id: " + err);      // no context to hold the message: crt_host_last_error() (thread-local) has it
    return CRT_ERR_HIP;
}

int crt_comm_init(crt_context *ctx, const void *id, uint32_t rank, uint32_t world) {
    int rc = bind(ctx); if (rc) return rc;
    if (!id) return fail(ctx, CRT_ERR_INVALID, "crt_comm_init: null id");
    std::string err;
    if (!comm_init(ctx->gather, id, rank, world, err)) return fail(ctx, CRT_ERR_HIP, "crt_comm_init: " + err);
    return CRT_OK;
}

int crt_comm_destroy(crt_context *ctx) {
    int rc = bind(ctx); if (rc) return rc;
    CTX_HIP(hipStreamSynchronize(ctx->stream));
    comm_destroy(ctx->gather);
    return CRT_OK;
}

int crt_image_gather(crt_context *ctx, float *full_dev) {
    int rc = bind(ctx); if (rc) return rc;
    if (!ctx->configured) return fail(ctx, CRT_ERR_INVALID, "crt_image_gather: not configured");
    const uint32_t world = ctx->cfg.band_world > 1 ? ctx->cfg.band_world : 1, rank = ctx->cfg.band_world > 1 ? ctx->cfg.band_rank : 0;
    if (!ctx->gather.comm) return fail(ctx, CRT_ERR_INVALID, "crt_image_gather: crt_comm_init has not been called");
    if (ctx->gather.world != world || ctx->gather.rank != rank)
        return fail(ctx, CRT_ERR_INVALID, "crt_image_gather: the communicator's rank/world differ from crt_config.band_rank/band_world");
    std::string err;
    if (!full_dev && !gather_full_buffer(ctx->gather, ctx->cfg.width, ctx->cfg.height, &full_dev, err)) return fail(ctx, CRT_ERR_HIP, "crt_image_gather: " + err);
    if (!gather_image(ctx->stream, ctx->gather, (const float *)ctx->accum.ptr, ctx->cfg.width, ctx->cfg.height, full_dev, err))
        return fail(ctx, CRT_ERR_HIP, "crt_image_gather: " + err);
    return CRT_OK;
}

int crt_image_gathered_read(crt_context *ctx, float *rgba, size_t n_floats) {
    int rc = bind(ctx); if (rc) return rc;
    if (!ctx->configured || !ctx->gather.full.ptr) return fail(ctx, CRT_ERR_INVALID, "crt_image_gathered_read: crt_image_gather(ctx, NULL) has not been called");
    if (!rgba || n_floats != (size_t)ctx->cfg.width * ctx->cfg.height * 4) return fail(ctx, CRT_ERR_INVALID, "crt_image_gathered_read: buffer must hold width*height*4 floats");
    CTX_HIP(hipStreamSynchronize(ctx->stream));
    CTX_HIP(hipMemcpy(rgba, ctx->gather.full.ptr, n_floats * 4, hipMemcpyDeviceToHost));
    return CRT_OK;
}

int crt_bands_assemble(crt_context *ctx, const float *gathered_dev, uint32_t width, uint32_t height, uint32_t world, float *full_dev) {
    int rc = bind(ctx); if (rc) return rc;
    if (!gathered_dev || !full_dev || width == 0 || height == 0 || world == 0) return fail(ctx, CRT_ERR_INVALID, "crt_bands_assemble: null buffer or empty frame");
    std::string err;
    if (!bands_assemble(ctx->stream, gathered_dev, full_dev, width, height, world, err)) return fail(ctx, CRT_ERR_HIP, "crt_bands_assemble: " + err);
    return CRT_OK;
}

int crt_stats_read(crt_context *ctx, crt_stats *out) {
    int rc = bind(ctx); if (rc) return rc;
    if (!out) return fail(ctx, CRT_ERR_INVALID, "crt_stats_read: null output");
    CTX_HIP(hipStreamSynchronize(ctx->stream));
    if ((rc = finish_timing(ctx))) return rc;
    DeviceStats ds;
    CTX_HIP(hipMemcpy(&ds, ctx->stats.ptr, sizeof ds, hipMemcpyDeviceToHost));
    out->rays = ds.rays; out->paths = ds.paths; out->node_visits = ds.node_visits; out->prim_tests = ds.prim_tests;
    out->last_render_ms = ctx->last_render_ms; out->last_build_ms = ctx->last_build_ms;
    out->last_extend_ms = ctx->last_extend_ms; out->last_extend_launches = ctx->last_extend_launches;
    out->last_shade_ms = ctx->last_shade_ms; out->last_shade_launches = ctx->last_shade_launches;
    out->last_primary_ms = ctx->last_primary_ms; out->last_primary_launches = ctx->last_primary_launches;
    return CRT_OK;
}

int crt_loop_counters_read(crt_context *ctx, uint64_t *out, uint32_t n_words) {
    int rc = bind(ctx); if (rc) return rc;
    if (!out || n_words != 16 * 6) return fail(ctx, CRT_ERR_INVALID, "crt_loop_counters_read: out must hold 16 x 6 words");
    CTX_HIP(hipStreamSynchronize(ctx->stream));
    CTX_HIP(hipMemcpy(out, (const char *)ctx->stats.ptr + offsetof(DeviceStats, loop), 16 * 6 * 8, hipMemcpyDeviceToHost));
    return CRT_OK;
}

int crt_stats_reset(crt_context *ctx) {
    int rc = bind(ctx); if (rc) return rc;
    CTX_HIP(hipMemsetAsync(ctx->stats.ptr, 0, sizeof(DeviceStats), ctx->stream));
    return CRT_OK;
}

int crt_set_instrumented(crt_context *ctx, int on) {
    if (!ctx) return CRT_ERR_INVALID;
    ctx->instrumented = on ? 1 : 0;
    return CRT_OK;
}

int crt_set_kernel_timing(crt_context *ctx, int on) {
    if (!ctx) return CRT_ERR_INVALID;
    ctx->wf.time_kernels = on != 0;
    return CRT_OK;
}

int crt_bvh_read(crt_context *ctx, crt_bvh_node *nodes, uint32_t *n_nodes, uint32_t *sphere_ids, uint32_t *quad_ids, uint32_t *tri_ids) {
    int rc = bind(ctx); if (rc) return rc;
    if (!ctx->have_bvh) return fail(ctx, CRT_ERR_INVALID, "crt_bvh_read: no BVH built");
    if (!n_nodes) return fail(ctx, CRT_ERR_INVALID, "crt_bvh_read: null n_nodes");
    const bool pairs = ctx->build.pairs_valid;      // a CRT_BVH_WIDE_DIRECT build has no pair BVH: 0 nodes, the slot IDs are still there
    *n_nodes = pairs ? ctx->build.n_pairs * 2 : 0;
    CTX_HIP(hipStreamSynchronize(ctx->stream));
    if (nodes && pairs && ctx->build.n_pairs) CTX_HIP(hipMemcpy(nodes, ctx->build.pairs.ptr, (size_t)ctx->build.n_pairs * 64, hipMemcpyDeviceToHost));
    if (sphere_ids && ctx->n_spheres) CTX_HIP(hipMemcpy(sphere_ids, ctx->build.sphere_ids.ptr, (size_t)ctx->n_spheres * 4, hipMemcpyDeviceToHost));
    if (quad_ids && ctx->n_quads) CTX_HIP(hipMemcpy(quad_ids, ctx->build.quad_ids.ptr, (size_t)ctx->n_quads * 4, hipMemcpyDeviceToHost));
    if (tri_ids && ctx->n_tris) CTX_HIP(hipMemcpy(tri_ids, ctx->build.tri_ids.ptr, (size_t)ctx->n_tris * 4, hipMemcpyDeviceToHost));
    return CRT_OK;
}

int crt_debug_boxes_upload(crt_context *ctx, const crt_ref_bvh_node *nodes, uint32_t n_nodes) {
    int rc = bind(ctx); if (rc) return rc;
    if (!ctx->have_scene) return fail(ctx, CRT_ERR_INVALID, "crt_debug_boxes_upload: no scene uploaded");
    if (n_nodes && !nodes) return fail(ctx, CRT_ERR_INVALID, "crt_debug_boxes_upload: null nodes");
    CTX_HIP(hipStreamSynchronize(ctx->stream));
    ctx->n_dbg_boxes = 0;
    if ((rc = upload(ctx, ctx->dbg_boxes, nodes, (size_t)n_nodes * sizeof(crt_ref_bvh_node)))) return rc;
    ctx->n_dbg_boxes = n_nodes;
    return CRT_OK;
}

int crt_bvh_wide_read(crt_context *ctx, void *nodes, uint32_t *n_nodes, uint32_t *n_levels) {
    int rc = bind(ctx); if (rc) return rc;
    if (!ctx->have_bvh) return fail(ctx, CRT_ERR_INVALID, "crt_bvh_wide_read: no BVH built");
    if (!n_nodes) return fail(ctx, CRT_ERR_INVALID, "crt_bvh_wide_read: null n_nodes");
    if ((rc = finish_build(ctx))) return rc;
    *n_nodes = ctx->build.n_wide;
    if (n_levels) *n_levels = ctx->build.wide_levels;
    CTX_HIP(hipStreamSynchronize(ctx->stream));
    if (nodes && ctx->build.n_wide) {
        // The caller's view stays one 128-byte record per node (64 bytes of node, 32 of links, 32 unused); on the device
        // the links live apart from the nodes.
        const size_t n = ctx->build.n_wide;
        std::vector<uint8_t> hot, links;
        try { hot.resize(n * 64); links.resize(n * 32); }      // nothing may leave an extern "C" function as an exception
        catch (const std::bad_alloc &) { return fail(ctx, CRT_ERR_HIP, "crt_bvh_wide_read: out of host memory"); }
        CTX_HIP(hipMemcpy(hot.data(), ctx->build.wide.ptr, n * 64, hipMemcpyDeviceToHost));
        CTX_HIP(hipMemcpy(links.data(), (const uint8_t *)ctx->build.wide.ptr + (size_t)kWideStrideF4 * 16 * ctx->build.wide_cap, n * 32, hipMemcpyDeviceToHost));
        uint8_t *out = (uint8_t *)nodes;
        for (size_t i = 0; i < n; i++) {
            memcpy(out + 128 * i, hot.data() + 64 * i, 64);
            memcpy(out + 128 * i + 64, links.data() + 32 * i, 32);
            memset(out + 128 * i + 96, 0, 32);
        }
    }
    return CRT_OK;
}

}  // extern "C"
