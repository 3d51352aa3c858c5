// build_internal.h -- interfaces between context.hip, bvh_build.hip and trace.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>

#include "crt/types.h"

namespace crt {

struct DeviceBuffer { void *ptr = nullptr; size_t bytes = 0; };

struct BuildInput {            // device pointers to the uploaded arrays (reference layouts)
    const crt_sphere *spheres; uint32_t n_spheres;
    const crt_quad *quads;     uint32_t n_quads;
    const crt_tri *tris;       uint32_t n_tris;
};

constexpr uint32_t kWideMaxLevelsDecl = 64;      // = kWideMaxLevels (declared below)
struct BuildState {
    uint32_t n_prims = 0, n_pairs = 0;
    const uint32_t *sorted_vals = nullptr;
    // scratch
    DeviceBuffer plo, phi, keys, vals, keys_alt, vals_alt, flag_a, flag_b, rank_a, rank_b, leaf_link, leaf_count;
    DeviceBuffer child_l, child_r, parent, nlo, nhi, ninfo, scene_bounds, temp;
    // SAH leaf-collapse parameters (CRT_BVH_SAH_REFINE)
    uint32_t max_leaf = 4; float c_isect = 1.0f, c_trav = 1.0f;
    // SAH refinement (bvh_sah.hip): LBVH node ranges, tier tables, group descriptors, final primitive order
    DeviceBuffer range_lo, range_hi, sah_flag, sah_ipos[2], sah_starts, sah_desc, perm;
    uint32_t sah_tiers = 0;
    // results (what traversal reads)
    DeviceBuffer pairs, tri_pack, sphere_pack, quad_pack, sphere_ids, quad_ids, tri_ids;
    DeviceBuffer tri_shade;        // 64-byte shading record per triangle, input order (bvh_build.hip: k_tri_bounds)
    // 8-wide quantised nodes (bvh_wide.hip, CRT_BVH_WIDE): n_wide nodes of kWideStrideF4 float4 each, BFS order
    DeviceBuffer wide, wide_queue[2], wide_levels_dev;
    uint32_t wide_cap = 0;     // nodes the `wide` allocation was laid out for: the links start kWideStrideF4 * wide_cap float4 in
    DeviceBuffer dp_cost, dp_choice, dp_parent, dp_depth, dp_list, dp_hist;      // collapse table (bvh_wide.hip)
    DeviceBuffer refit_list[2], refit_count[2];      // parked climbers between the stages of the LBVH refit
    uint32_t n_wide = 0, wide_levels = 0;
    uint32_t wide_levels_hint = 0;                   // depth of the previous collapse (build_wide plans its launches with it)
    uint32_t pair_depth = 0;                         // levels of the SAH-refined pair tree (0: not measured, LBVH <= 95)
    bool pairs_valid = false;                        // false after a CRT_BVH_WIDE_DIRECT build: no pair tree, only the 8-wide one
    bool fuse_dp = false;                            // build_sah should make the wide collapse's table while it builds (set by build_bvh: CRT_BVH_SAH_REFINE | CRT_BVH_WIDE)
    bool dp_fused = false;                           // ... and did: dp_choice is complete, word 7 of dp_cost row 0 carries the pair tree's height << 21
    bool levels_measured = false;                    // level_hist / dp_depth describe the current pairs (bvh_sah.hip: measure_pair_depths)
    uint32_t level_hist[128] = {};                   // pairs per depth (root = 1)
    // Enqueue-only direct rebuilds (bvh_direct.hip).  A CRT_BVH_WIDE_DIRECT build that may not wait for the GPU launches its
    // level kernels for the depth of the previous such build of this scene plus two (the kernels read every count from the
    // device's level table); the table comes back through pinned memory and is looked at by resolve_wide_counts.
    bool async_request = false;                      // this build may skip every wait (crt_bvh_build with build_ms == NULL)
    uint32_t direct_hint_n = 0, direct_hint_levels = 0, direct_hint_leaf = 0;
    uint32_t direct_hint_lc[kWideMaxLevelsDecl + 3] = {};
    uint32_t *host_lc = nullptr;                     // pinned copy of the device's level table (+ overflow word)
    hipEvent_t ev_counts = nullptr;
    bool counts_pending = false;
    const uint32_t *sort_error_dev = nullptr;        // static builds through bvh_sort.hip: the sort's error word, looked at by crt_bvh_build after its wait
    bool records_fresh = false;                      // tri_shade / plo centroids / scene_bounds describe the triangles as they are now (animate_tris wrote them)
};

constexpr uint32_t kWideStrideF4 = 4;     // 64-byte nodes (what a step loads): two siblings per 128-byte line; the eight links of node i
                                          // live apart, at 32 bytes per node, behind the node array (build_wide / wide_links_of)
constexpr uint32_t kWideMaxLevels = 64;
static_assert(kWideMaxLevels == kWideMaxLevelsDecl, "BuildState::direct_hint_lc");
constexpr uint32_t kWideLeafTag = 0x80000000u;                  // link word of a leaf slot: tag | kind << 29 | (count - 1) << 26 | first slot in the kind's array
constexpr uint32_t kWideEmptyLink = kWideLeafTag | (3u << 29);  // ... of an empty slot
constexpr uint32_t kPairLevels = 128;      // bins of the depth histogram (> kMaxPairDepth)
constexpr uint32_t kMaxPairDepth = 96;    // pair trees deeper than this are refused: a walk pushes at most depth - 1 entries and the
                                          // stacks hold at least 1 (LDS, debug minimum) + 96 (scratch), intersect.h

bool build_bvh(hipStream_t stream, const BuildInput &in, uint32_t flags, BuildState &st, std::string &err);
bool build_sah(hipStream_t stream, const BuildInput &in, BuildState &st, const uint32_t *vals, std::string &err);
bool patch_sah_links(hipStream_t stream, BuildState &st, std::string &err);
bool measure_pair_depths(hipStream_t stream, BuildState &st, std::string &err);
bool build_wide(hipStream_t stream, BuildState &st, std::string &err);
bool wide_collapse_greedy();      // CRT_WIDE_GREEDY=1: no table (A/B runs)
uint32_t pair_depth_limit();      // kMaxPairDepth, or less (CRT_BVH_MAX_PAIR_DEPTH: test hook)
struct PackOut;
bool build_wide_direct(hipStream_t stream, const BuildInput &in, BuildState &st, const uint64_t *keys, uint32_t idx_bits, const uint32_t *rank_sphere,
                       const uint32_t *rank_quad, const PackOut &out, const uint32_t *sort_error, std::string &err);
// After an enqueue-only direct build: waits for its level table and takes node count, depth and the next build's hint
// from it.  overflow: the tree was deeper than the launches planned for (parts were dropped: the BVH is not usable).
bool resolve_wide_counts(BuildState &st, bool &overflow, std::string &err);
// Boxes of the existing 8-wide tree recomputed bottom-up from the primitives as they are now (crt_bvh_refit); topology,
// leaf contents and primitive order stay.  cost_dev (optional): receives sum over nodes of area(node), and area(root) in [1].
bool refit_wide(hipStream_t stream, const BuildInput &in, BuildState &st, float *cost_dev, std::string &err);
// bvh_sort.hip: LSD radix sort of 64-bit keys over bits [bit_lo, bit_hi), keys only; enqueue-only.  *sorted = the buffer that
// holds the result; *error_word = a device word that is non-zero afterwards if a look-back wait gave up (never seen).
size_t radix_sort_scratch_bytes(uint32_t n);
bool radix_sort_u64(hipStream_t stream, uint64_t *keys_a, uint64_t *keys_b, uint32_t *vals_a, uint32_t *vals_b, uint32_t n, uint32_t bit_lo, uint32_t bit_hi,
                    void *scratch, const uint64_t **sorted, const uint32_t **sorted_vals, const uint32_t **error_word, std::string &err);
bool radix_sort_keys_u64(hipStream_t stream, uint64_t *keys_a, uint64_t *keys_b, uint32_t n, uint32_t bit_lo, uint32_t bit_hi, void *scratch,
                         const uint64_t **sorted, const uint32_t **error_word, std::string &err);
bool animate_tris(hipStream_t stream, uint32_t n, const crt_tri *rest, crt_tri *out, float frame, float amplitude, BuildState *st, std::string &err);
void free_build_state(BuildState &st);

// ---- tracing (trace.hip) ----
struct DeviceSceneHost {       // host-side mirror of DeviceScene (kept POD so context.hip need not include device headers)
    const void *nodes; uint32_t n_pairs;
    const void *wide; uint32_t n_wide, wide_levels;     // 8-wide nodes (null: the wavefront tracer walks the pair BVH)
    const uint32_t *wide_links;                          // eight link words per wide node
    const void *tri_pack, *sphere_pack, *quad_pack, *tri_shade;
    const uint32_t *sphere_ids, *quad_ids;
    const crt_sphere *spheres; const crt_quad *quads; const crt_tri *tris; uint32_t n_tris;
    const crt_material *materials;
    const uint4 *tex0; uint32_t tex0_w, tex0_h, n_textures;      // bilinear footprints (intersect.h DeviceScene::tex0)
    const uint4 *tex_table; uint32_t ext;
    const void *dbg_boxes; uint32_t n_dbg_boxes;     // crt_debug_boxes_upload (R15)
};

struct FrameSetup {
    uint32_t width, height;          // imageSize()
    uint32_t grid_w, grid_h;         // pixels actually rendered
    uint32_t band_rank, band_world;  // this context owns the 8-row bands b with b % band_world == band_rank
    uint32_t n_local_bands;          // bands owned
    uint32_t max_bounces;
    uint32_t rng_mode;
    uint32_t instrumented;
    uint32_t small_stack;            // CRT_FLAG_DEBUG_SMALL_STACK
};

struct DeviceStats {
    unsigned long long rays, paths, node_visits, prim_tests;
    // instrumented k_wf_extend only, per bounce: loop iterations, inner-phase iterations, lanes stepping in them,
    // leaf-phase iterations, lanes testing in them, iterations that refilled lanes
    unsigned long long loop[16][6];
};

struct WavefrontState {
    uint32_t capacity = 0;           // paths the queues can hold
    DeviceBuffer ray_o[2], ray_d[2], thr[2], hit, counters, pass, query_rays;
    bool time_kernels = false;
    bool tail_persist = false;       // the path kernel in its persistent form (k_wf_path) where it applies
    int tail_bounce = -1;            // >= 0: this dispatch ends in k_wf_tail from that bounce on (set per dispatch by the context, trace.hip)
    std::vector<hipEvent_t> ev;      // per bounce: before k_wf_extend, between it and k_wf_shade, after k_wf_shade
    size_t ev_used = 0;
    std::vector<uint8_t> ev_primary; // per bounce recorded: 1 = the first kernel was k_wf_primary, not k_wf_extend; 2 = k_wf_primary ran the whole bounce (no k_wf_shade)
};

bool launch_megakernel(hipStream_t stream, const DeviceSceneHost &sc, const FrameSetup &fs, const crt_scene_params &params,
                       float *accum, uint32_t *hit_ids, DeviceStats *stats, std::string &err);
// n_batch (1..64) consecutive iterations traced together: iteration b has camera.iteration = params.camera.iteration + b
// and camera.seed = seeds[b] (seeds == nullptr: params.camera.seed); accum_pixels = pixels of the packed band buffer.
bool launch_wavefront(hipStream_t stream, const DeviceSceneHost &sc, const FrameSetup &fs, const crt_scene_params &params,
                      uint32_t n_batch, const float *seeds, WavefrontState &wf, float *accum, uint32_t accum_pixels,
                      uint32_t *hit_ids, DeviceStats *stats, std::string &err);
// crt_rays_query: answers n_rays caller rays (host or device memory) with one k_wf_extend launch; synchronous.
bool launch_ray_query(hipStream_t stream, const DeviceSceneHost &sc, const FrameSetup &fs, WavefrontState &wf, const crt_ray *rays,
                      uint32_t n_rays, uint32_t mode, crt_ray_hit *hits, DeviceStats *stats, std::string &err);
// crt_rays_query_noalloc: the same query over the library's static device scratch (no allocation, no copy, no wait).
// noalloc_init(device) looks the scratch's address up once per device, with that device current (crt_create).
bool noalloc_init(int device, std::string &err);
bool launch_ray_query_noalloc(hipStream_t stream, int device, const DeviceSceneHost &sc, const FrameSetup &fs, const crt_ray *rays_dev,
                              uint32_t n_rays, uint32_t mode, crt_ray_hit *hits_dev, DeviceStats *stats, std::string &err);
void free_wavefront(WavefrontState &wf);
size_t wavefront_bytes(const WavefrontState &wf);      // device memory the tracer's queues hold right now

}  // namespace crt
