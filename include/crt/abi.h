/*
 * crt/abi.h -- the drop-in boundary of the MI355X ray-tracing core.
 *
 * The reference has no plugin/FFI interface; its boundary is the Vulkan
 * descriptor + dispatch surface inside VulkanEngine (engine/src/vk_engine.cpp).
 * Each entry point below replaces one piece of that surface and says which.
 * Plain C ABI: pointers and sizes only, no C++ or torch types.
 *
 * Conventions
 *   - every function returns 0 on success, a negative crt_status otherwise;
 *     crt_last_error(ctx) returns a message owned by ctx.  Nothing aborts
 *     (the reference abort()s: vk_engine.cpp:27-34,86-95).
 *   - host buffers are copied synchronously; the caller keeps ownership and
 *     may free them as soon as the call returns (as upload_buffer's staging
 *     copy + fence wait does: vk_engine.cpp:656-690,1293-1315).
 *   - a context is bound to one GPU and is used by one thread at a time.
 *     crt_dispatch/crt_render enqueue on the context's HIP stream and return;
 *     crt_sync and every *_read call block until the work is done.
 *   - the library never falls back to the CPU: without a usable HIP device
 *     crt_create fails.
 */
#ifndef CRT_ABI_H
#define CRT_ABI_H

#include <stddef.h>
#include "crt/types.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct crt_context crt_context;

enum crt_status {
    CRT_OK = 0,
    CRT_ERR_INVALID = -1,     /* bad argument / call order */
    CRT_ERR_HIP = -2,         /* a HIP runtime call failed */
    CRT_ERR_NO_DEVICE = -3,   /* no usable GPU */
    CRT_ERR_IO = -4           /* file could not be read / parsed */
};

/* Library version string, e.g. "crt-mi355x 0.1 (gfx950)". */
const char *crt_version(void);

/* Replaces init_vulkan + allocator set-up (vk_engine.cpp:110-212): binds a
 * context to HIP device `device`, creates its stream. */
int crt_create(int device, crt_context **out);
void crt_destroy(crt_context *ctx);
const char *crt_last_error(const crt_context *ctx);

/* Replaces creation of the RGBA32F storage image (vk_engine.cpp:1157-1168) and
 * the compile-time knobs of the shader (compute.comp:15-16,21).  Allocates the
 * zero-initialised accumulation image; may be called again to resize/reset. */
int crt_configure(crt_context *ctx, const crt_config *cfg);

/* Replaces the five upload_buffer calls + texture descriptors of
 * init_descriptors (vk_engine.cpp:1177-1229).  Invalidates any BVH.  Textures (RGBA8 rows, top-down) are copied during
 * the call and kept on the device as bilinear footprints, 16 bytes per texel (four times the upload; fewer than 2^32
 * texels in all). */
int crt_scene_upload(crt_context *ctx, const crt_scene_desc *scene);

/* Limits (every one is checked and reported as an error, never a wrong image): fewer than 2^30 primitives of a kind
 * (ID tag), width x height below 2^31; the wavefront tracer and crt_rays_query take up to 2^26 - 2 primitives and
 * 16 bounces; CRT_BVH_WIDE up to 2^24 - 1 primitives and 64 levels; a wavefront batch needs fewer than 2^26 - 1
 * pixels per context (larger frames render one iteration per launch sequence).  A CRT_BVH_SAH_REFINE pair tree deeper than
 * the pair tracers' stacks (96 levels; only scenes with thousands of coincident primitives come near) fails the build when
 * it is the only tree; with CRT_BVH_WIDE the build stands and the pair tree is withheld from the megakernel tracer and
 * crt_bvh_read, as after CRT_BVH_WIDE_DIRECT or a refit. */

/* Build extensions (CRT_EXT_* of types.h).  No counterpart in the reference: its shader ignores
 * RTMaterial::GPU_t::textureIndex (compute.comp:386-387 reads textures[0]) and has no alpha test; SURVEY section 7
 * lists both as extensions of this build, off by default.  Takes effect from the next dispatch / query; may be called
 * before or after crt_scene_upload.  CRT_EXT_TEXTURE_INDEX needs every material's textureIndex to be CRT_BAD_INDEX or
 * below n_textures: the call (or a later crt_scene_upload) fails with CRT_ERR_INVALID otherwise.  Unknown bits are
 * refused. */
int crt_extensions_set(crt_context *ctx, uint32_t ext_flags);

/* Replaces BVHNode::BVHNode + gpu_serialize on the host
 * (raytracing/src/bounding_volume_hierarchy.cpp:63-130) with a device-side
 * build over the uploaded primitives.  flags: CRT_BVH_*.  build_ms (optional)
 * receives the device time of the build. */
int crt_bvh_build(crt_context *ctx, uint32_t flags, float *build_ms);
/* With build_ms == NULL and CRT_BVH_WIDE_DIRECT the call only ENQUEUES whenever it rebuilds the scene the previous direct
 * build of this context saw (same primitive count): crt_tris_animate -> crt_bvh_build -> crt_dispatch is then one
 * stream of launches and the host never waits.  Such a build plans its per-level launches from the previous build's
 * depth plus two; its device time appears in crt_stats.last_build_ms after the next crt_sync.  A tree that turns out
 * deeper than planned (the scene's shape changed abruptly between two frames) is reported by the next blocking call
 * (crt_sync, crt_stats_read, crt_image_read ...) as CRT_ERR_INVALID: the images rendered since lack parts of the
 * scene and the BVH has to be built again (that build counts its levels the slow way). */

/* Config 5, cheaper variant (SURVEY.md 8(b), row "BVH": crt_bvh_refit/rebuild): recomputes the boxes of the existing
 * 8-wide tree bottom-up from the primitives as they are now, keeping topology and primitive order; enqueue-only when
 * cost == NULL.  No counterpart in the reference (it builds once).  Images are the same bits as after a rebuild
 * (closest hits do not depend on the tree); what degrades as the geometry moves away from the pose the tree was built
 * for is traversal cost: *cost (optional; makes the call wait) receives sum over nodes of area(node) / area(root), the
 * figure to compare with the one right after a build when deciding to rebuild.  Needs a tree built with CRT_BVH_WIDE or
 * CRT_BVH_WIDE_DIRECT; afterwards the pair BVH (megakernel tracer, crt_bvh_read) is stale and refused until the next
 * build. */
int crt_bvh_refit(crt_context *ctx, float *cost);

/* Config 5 (per-frame rebuild): overwrite the triangle vertices in place from
 * device-resident or host memory, then call crt_bvh_build again.  No
 * counterpart in the reference (its BVHs are built once: vk_engine.cpp:748-934).  The
 * new triangles' material indices are checked on the device; one past the material
 * array fails with CRT_ERR_INVALID and leaves the context without a scene (upload again). */
int crt_tris_update(crt_context *ctx, const crt_tri *tris, uint32_t first, uint32_t count);
/* Reads triangles [first, first + count) back as they are on the device now (after crt_tris_animate / crt_tris_update)
 * into host or device memory.  Synchronous.  Build-only diagnostic: the parity tests hand these triangles to the oracle. */
int crt_tris_read(crt_context *ctx, crt_tri *tris, uint32_t first, uint32_t count);
/* Device-side animation used by the 10M-triangle benchmark: displaces every
 * triangle vertex's y by amplitude * sin(frame + x + z) relative to the
 * rest pose (the triangles as uploaded, or as last replaced by crt_tris_update). */
int crt_tris_animate(crt_context *ctx, float frame, float amplitude);

/* Replaces one frame of VulkanEngine::draw (vk_engine.cpp:1409-1435): the
 * 112-byte UBO write + vkCmdDispatch(W/8, H/8, 1).  One sample per pixel;
 * params->camera.iteration == 1 resets the accumulation (compute.comp:474), by the shader's own arithmetic:
 * cum + ((-cum) + pass), which equals pass up to one rounding when cum holds an earlier picture; crt_configure zeroes
 * the image exactly. */
int crt_dispatch(crt_context *ctx, const crt_scene_params *params);

/* n dispatches enqueued back to back: dispatch i uses *params with
 * camera.iteration = first_iteration + i and camera.seed = seeds[i].
 * Equivalent to n crt_dispatch calls (bit-identical image).  The wavefront
 * tracer traces up to 64 consecutive iterations as one batch (as many as keep
 * a batch within ~134 M paths; CRT_WF_BATCH_PATHS overrides: twice that is +0.8 % on the 4K headline for twice the queue memory) and adds their
 * radiance to the image in iteration order; device memory for the queues
 * grows with the batch (136 bytes per path). */
int crt_render(crt_context *ctx, const crt_scene_params *params, uint32_t first_iteration,
               uint32_t n_iterations, const float *seeds);
/* Device memory: scene + BVH (about 250 bytes per primitive including build scratch), the image (20 bytes per pixel)
 * and, for the wavefront tracer, the queues of one batch: 104 bytes per path slot + 16 bytes per pixel and iteration of
 * the batch.  A batch is capped at half of the memory that is free when crt_render is called (plus what the context's
 * queues already hold), so several contexts on one GPU shrink each other's batches instead of failing; crt_configure
 * to a different frame releases the queues.  crt_dispatch / crt_render return without waiting for the GPU: a call
 * blocks only while the dispatch two calls back is still running (two timing slots, like the reference's two frame
 * slots, vk_engine.cpp:1377) or when crt_set_kernel_timing is on. */

/* Ray queries against the current scene and BVH: closest hit or any hit ("shadow ray") for n_rays caller-supplied
 * rays, the same hit_world the path tracer runs (compute.comp:326-361: same intersection arithmetic, same interval
 * rule, same tie-break), answered by one launch of the tracer's persistent traversal kernel.  The reference has no
 * such entry point (its shader is hit_world's only caller); it serves engine-side picking, visibility and
 * light-sampling passes.  rays / hits may be host or device memory.  Synchronous.  Uses the wavefront tracer's
 * queues whatever crt_config.tracer says, so do not overlap it with a render on the same context. */
int crt_rays_query(crt_context *ctx, const crt_ray *rays, uint32_t n_rays, uint32_t mode, crt_ray_hit *hits);

/* crt_rays_query for callers that cannot allocate.  No counterpart in the reference; it exists for code that must not
 * call hipMalloc: under HIP graph capture, in tight serving loops, in harnesses that forbid allocation.  Same answers
 * as crt_rays_query (same kernels), but: rays and hits are DEVICE memory (checked), at most CRT_NOALLOC_MAX_RAYS rays
 * per call (more is an error), the queue lives in a static device array of the library (4.125 MiB per GPU, held until
 * the process ends) instead of allocated buffers, and the call only enqueues on the context's stream: no hipMalloc,
 * hipFree, copy through staging or wait; crt_sync (or any stream-ordered use of hits on that stream) waits.
 * All contexts of one GPU share that GPU's scratch: calls that overlap on one GPU (two contexts, two threads) race;
 * calls on one context are ordered by its stream. */
int crt_rays_query_noalloc(crt_context *ctx, const crt_ray *rays_dev, uint32_t n_rays, uint32_t mode, crt_ray_hit *hits_dev);

/* Replaces waitForFences(renderFence) (vk_engine.cpp:1377). */
int crt_sync(crt_context *ctx);
/* Small batches (one sample of a frame of up to ~4 M pixels) can be traced by three kernel forms whose ranking depends on
 * scene and view (types.h: CRT_FLAG_PER_BOUNCE_KERNELS / CRT_FLAG_PATH_KERNEL / CRT_FLAG_PATH_KERNEL_PERSISTENT fix one).
 * Without such a flag the context measures: its first dispatches after crt_configure / crt_scene_upload take the forms
 * in turn (two timed samples each: six dispatches at most) and the fastest stays.  crt_retune starts that measurement
 * again, e.g. after the camera moved to a view with very different ray lengths.  Images do not depend on the form.
 * No counterpart in the reference (one shader). */
int crt_retune(crt_context *ctx);
/* Non-blocking form of the same question (vkGetFenceStatus; the reference only ever blocks): *done = 1 when everything
 * enqueued on the context so far has finished, 0 while the GPU is still working. */
int crt_poll(crt_context *ctx, int *done);

/* Replaces the present pass's read of the storage image (vk_engine.cpp:1448-1458):
 * copies the raw accumulation image (RGBA32F, xyz = running sum, w = iteration,
 * row 0 = bottom of the view) into rgba (width*height*4 floats). */
int crt_image_read(crt_context *ctx, float *rgba, size_t n_floats);

/* Build-only diagnostic: primary-hit primitive ID per pixel of the most
 * recent dispatch, i.e. of the last iteration of the last crt_render
 * (CRT_ID_MISS where the camera ray hit nothing or the pixel is outside the
 * rendered grid). */
int crt_hit_ids_read(crt_context *ctx, uint32_t *ids, size_t n_ids);

/* Device pointer + size of this context's packed band buffer (the rows it
 * owns, in increasing y, each row width*4 floats) for a device-to-device
 * gather (RCCL all-gather over xGMI in the multi-GPU harness).  With
 * band_world <= 1 this is the whole image. */
int crt_image_device_ptr(crt_context *ctx, void **device_ptr, size_t *n_bytes);
/* Copies the packed band buffer into caller-owned device memory (e.g. a torch
 * tensor that RCCL will gather) and waits for the copy. */
int crt_image_copy_to_device(crt_context *ctx, void *dst_device, size_t n_bytes);
/* Device-side hand-over of the frame to another API, in place of the read-back above (the reference has no such step: its
 * storage image is written and sampled by the same Vulkan device, vk_engine.cpp:1429-1458).  Enqueues a device-to-device
 * copy of the packed band buffer (the whole image with band_world <= 1) into an allocation of the context that exports as
 * a dma-buf file descriptor (HIP virtual memory management, hipMemHandleTypePosixFileDescriptor); crt_sync / crt_poll say
 * when the copy has landed.
 *   fd           non-NULL: receives a NEW descriptor for that allocation on every call; the caller owns it (Vulkan's
 *                vkAllocateMemory with VkImportMemoryFdInfoKHR takes it over; otherwise close() it).  Engines import once
 *                and pass NULL afterwards: the allocation stays the same until the frame size changes;
 *   image_bytes  rows owned * width * 16;  alloc_bytes = the allocation's size (image_bytes rounded up to the device's
 *                allocation granularity): the size an importer has to name. */
int crt_image_export_fd(crt_context *ctx, int *fd, size_t *image_bytes, size_t *alloc_bytes);
/* ---- multi-GPU gather; the reference is single-GPU (vk_engine.cpp:156-157), so none of these has a
 * counterpart there.  One process per GPU; each renders its bands (crt_config.band_rank / band_world) and the finished
 * frame is assembled with ONE all-gather over RCCL (xGMI) + an un-interleave kernel.  RCCL is bound at run time
 * (dlopen of librccl.so.1), libcrt.so has no link-time dependency on it.
 *   crt_comm_unique_id   rank 0 makes the id (ncclGetUniqueId) and hands its CRT_COMM_ID_BYTES bytes to the other ranks
 *                        by whatever channel the engine has (a socket, MPI, a file; the bench uses torch.distributed);
 *                        it has no context, so on failure the message is crt_host_last_error()'s (thread-local).
 *                        CRT_RCCL_SONAME (environment) names the library to bind instead of librccl.so.1;
 *   crt_comm_init        collective over all ranks (ncclCommInitRank) on the context's GPU;
 *   crt_image_gather     enqueues the all-gather of the packed band buffers and the un-interleave on the context's
 *                        stream; full_dev = device buffer of width*height*4 floats that receives the whole frame on
 *                        every rank (NULL: a buffer owned by the context, read with crt_image_gathered_read);
 *                        crt_sync waits.  rank/world must be the ones given to crt_configure;
 *   crt_bands_assemble   only the un-interleave: gathered_dev = world buffers of ceil(ceil(height/8)/world)*8 rows,
 *                        rank-major (what the all-gather produces), for engines that move the bands themselves. */
int crt_comm_unique_id(void *id /* CRT_COMM_ID_BYTES */);
int crt_comm_init(crt_context *ctx, const void *id, uint32_t rank, uint32_t world);
int crt_comm_destroy(crt_context *ctx);
int crt_image_gather(crt_context *ctx, float *full_dev);
int crt_image_gathered_read(crt_context *ctx, float *rgba, size_t n_floats);
int crt_bands_assemble(crt_context *ctx, const float *gathered_dev, uint32_t width, uint32_t height, uint32_t world, float *full_dev);

/* Rows owned by (band_rank, band_world) for a given height: writes the y of
 * each owned row in increasing order; returns the count (rows may be NULL). */
uint32_t crt_band_rows(uint32_t height, uint32_t band_rank, uint32_t band_world, uint32_t *rows);

/* Counters since the last crt_stats_reset (rays = hit_world calls). */
int crt_stats_read(crt_context *ctx, crt_stats *out);
int crt_stats_reset(crt_context *ctx);
/* Makes the next dispatches count node visits and primitive tests too
 * (slower; used to measure the V and T terms of the bytes-per-ray model). */
int crt_set_instrumented(crt_context *ctx, int on);
/* Divergence profile of the instrumented k_wf_extend launches since the last crt_stats_reset: per bounce (16 rows)
 * {loop iterations, inner-phase iterations, lanes stepping in them, leaf-phase iterations, lanes testing in them,
 * iterations that refilled lanes}, summed over waves.  Build-only diagnostic (no counterpart in the reference). */
int crt_loop_counters_read(crt_context *ctx, uint64_t *out, uint32_t n_words);
/* Brackets every k_wf_extend launch (the dominant kernel) and every k_wf_shade launch with HIP events on the context's
 * stream so that crt_stats.last_extend_ms / last_shade_ms report their device time separately (bench.py's roofline
 * figures).  While it is on, crt_dispatch / crt_render wait for the previous call's events before reusing them, and
 * small batches stay on the per-bounce kernels (CRT_FLAG_PER_BOUNCE_KERNELS, types.h). */
int crt_set_kernel_timing(crt_context *ctx, int on);

/* Read back the build's BVH for cross-checks: nodes (pair layout, see
 * crt_bvh_node) and, per kind, the input ID of each sorted primitive slot.
 * Two-call protocol: pass NULL arrays to get the counts. */
int crt_bvh_read(crt_context *ctx, crt_bvh_node *nodes, uint32_t *n_nodes,
                 uint32_t *sphere_ids, uint32_t *quad_ids, uint32_t *tri_ids);

/* Read back the 8-wide nodes built with CRT_BVH_WIDE (128 bytes each, layout in
 * computeraytracer_amd/csrc/hip/bvh_wide.hip; node 0 is the root, BFS order) for cross-checks.
 * Two-call protocol: pass nodes = NULL to get the count (0 when the BVH was built without the flag).
 * Build-only diagnostic, no counterpart in the reference. */
int crt_bvh_wide_read(crt_context *ctx, void *nodes, uint32_t *n_nodes, uint32_t *n_levels);

/* Build-only diagnostic: sorts n host keys on the device over bits [bit_lo, bit_hi) with the build's own radix sort
 * (bvh_sort.hip; stable, so keys that agree on those bits keep their order) and copies them back.  For the sort's unit
 * test; no counterpart in the reference (std::sort inside its builder, bounding_volume_hierarchy.cpp:26-61). */
int crt_debug_sort_u64(crt_context *ctx, uint64_t *keys, uint32_t n, uint32_t bit_lo, uint32_t bit_hi);
/* Build-only diagnostic: the build-defined FP32 sequences of the device (sin/cos on [0, 2 pi], cube root on [0, 1], acos,
 * atan2, x^5) and its IEEE square root and division, evaluated on the GPU for n host-side arguments.  fn: 0 sincos
 * (out0 = sin, out1 = cos), 1 cbrt01, 2 acos, 3 atan2(a, b), 4 pow5, 5 sqrt, 6 a / b.  No counterpart in the reference
 * (GLSL leaves these to the driver); lets a test compare the device with the oracle function by function. */
int crt_debug_math(crt_context *ctx, int fn, uint32_t n, const float *a, const float *b, float *out0, float *out1);
/* The same with a 32-bit value travelling along with every key (the form static builds of up to a million primitives use). */
int crt_debug_sort_pairs_u64(crt_context *ctx, uint64_t *keys, uint32_t *vals, uint32_t n, uint32_t bit_lo, uint32_t bit_hi);

/* The AABB-outline debug view (camera.shouldRenderAABB, compute.comp:427-436) draws the boxes of the REFERENCE's
 * BVHNode array (set 0 binding 2, vk_engine.cpp:1203), which the device-side build does not produce.  A caller that
 * wants the view supplies that array (64-byte nodes, T1 layout; only the boxes are read); it belongs to the current
 * scene and is dropped by crt_scene_upload.  Without it the flag draws nothing. */
int crt_debug_boxes_upload(crt_context *ctx, const crt_ref_bvh_node *nodes, uint32_t n_nodes);

/* ---- host-side helpers (no GPU needed) --------------------------------- */

/* Camera::Camera + Camera::calculateProperties (raytracing/src/camera.cpp:3-12,40-62):
 * fills the 96-byte camera block; iteration = 1, seed = 0. */
void crt_camera_init(crt_camera_data *out, const float position[3], const float at[3],
                     float fov_degrees, float aspect_ratio, float aperture, float focus_distance);

/* Frame-seed schedule of vk_engine.cpp:1412: seeds[i] = float(r_i) / 1E3f with
 * r_i the i-th output of glibc rand() seeded 1 (restated, libc-independent). */
void crt_frame_seeds(uint32_t first, uint32_t count, float *seeds);

/* Quad::Quad derived fields (primitives.h:73-80): fills normal, d, w from
 * corner/u/v. */
void crt_quad_init(crt_quad *q, const float corner[3], const float u[3], const float v[3], uint32_t material);

/* Built-in scenes: the reference's four (vk_engine.cpp:825-926: "book1",
 * "quads", "corne", "cirno") and the benchmark ones ("monkey", "voxel_city",
 * "grid10m", and "lost_empire" for a checkout that has lost_empire.obj/.mesh).  mesh/texture inputs come from files under asset_dir where a
 * scene needs them.  The returned scene owns its arrays until crt_scene_free. */
typedef struct crt_host_scene crt_host_scene;
int crt_scene_builtin(const char *name, const char *asset_dir, uint32_t width, uint32_t height, crt_host_scene **out);
const crt_scene_desc *crt_host_scene_desc(const crt_host_scene *s);
const crt_scene_params *crt_host_scene_params(const crt_host_scene *s);
/* The world list handed to the reference's BVHNode ctor, as CSR over input
 * IDs, for the oracle's reference-BVH mode. */
int crt_host_scene_world(const crt_host_scene *s, uint32_t *n_objects, const uint32_t **obj_offsets,
                         const uint32_t **obj_prims, const uint8_t **obj_is_list);
void crt_scene_free(crt_host_scene *s);
const char *crt_host_last_error(void);

/* OBJ -> triangle soup with the asset baker's conventions
 * (asset-baker/asset_main.cpp:38-100: de-indexed, uv.y = 1 - uy, shapes merged).
 * *tris is malloc'ed; free with crt_free. */
int crt_load_obj(const char *path, uint32_t material_index, crt_tri **tris, uint32_t *n_tris);
void crt_free(void *p);

/* The reference's asset container (asset-lib/src/asset_loader.cpp:13-61): 16-byte
 * header {type[4], u32 version, u32 jsonLen, u32 blobLen} + JSON + LZ4 block.
 * "MESH" (mesh_asset.cpp:15-112): blob = LZ4(PNCV_F32 vertices || u32 indices).
 * "TEXI" (texture_asset.cpp:12-70): blob = LZ4(RGBA8 rows, top-down).
 * Outputs are malloc'ed; free with crt_free. */
int crt_mesh_file_read(const char *path, float **vertices_pncv, uint32_t *n_vertices, uint32_t **indices, uint32_t *n_indices);
int crt_mesh_file_write(const char *path, const float *vertices_pncv, uint32_t n_vertices, const uint32_t *indices,
                        uint32_t n_indices, const char *original_file);
int crt_texture_file_read(const char *path, uint8_t **rgba8, uint32_t *width, uint32_t *height);
int crt_texture_file_write(const char *path, const uint8_t *rgba8, uint32_t width, uint32_t height, const char *original_file);
/* OBJ -> PNCV vertex list exactly as the baker's extract_mesh_from_obj emits it
 * (asset_main.cpp:67-100): one vertex per face corner, indices 0..n-1. */
int crt_obj_to_vertices(const char *path, float **vertices_pncv, uint32_t *n_vertices);
/* In-tree LZ4 block codec (format of lz4 >= 1.9.4's LZ4_compress_default /
 * LZ4_decompress_safe).  Return the output size or a negative crt_status. */
int crt_lz4_compress_bound(int n);
int crt_lz4_compress(const uint8_t *src, int n, uint8_t *dst, int cap);
int crt_lz4_decompress(const uint8_t *src, int n, uint8_t *dst, int cap);

#ifdef __cplusplus
}
#endif
#endif /* CRT_ABI_H */
