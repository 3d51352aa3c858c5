/*
 * crt/types.h -- POD layouts at the drop-in boundary.
 *
 * Every struct here is byte-identical to the buffer element the reference
 * engine uploads to (or reads back from) its Vulkan compute pipeline, so a
 * caller that already fills the reference's SSBO/UBO staging memory can hand
 * the same bytes to crt_scene_upload()/crt_dispatch().  File:line citations
 * are relative to the upstream ComputeRaytracer checkout.
 */
#ifndef CRT_TYPES_H
#define CRT_TYPES_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CRT_BAD_INDEX 0xFFFFFFFFu

/* Hittable::Type, raytracing/include/hittable.h:13-18; shaders/compute.comp:11-13 */
enum { CRT_TYPE_SPHERE = 1, CRT_TYPE_QUAD = 2, CRT_TYPE_TRI = 4, CRT_TYPE_BVH_NODE = 8 };

/* RTMaterial::Type, raytracing/include/rt_material.h:13-18; compute.comp:6-9 */
enum { CRT_MAT_LAMBERTIAN = 1, CRT_MAT_METAL = 2, CRT_MAT_DIELECTRIC = 4, CRT_MAT_DIFFUSE_LIGHT = 8 };

/* Sphere::GPU_t, raytracing/include/primitives.h:29-34; compute.comp:34-39 (32 B) */
typedef struct crt_sphere {
    float center[3];
    float radius;            /* may be negative: flips the outward normal */
    float pad0[3];
    uint32_t materialIndex;
} crt_sphere;

/* Quad::GPU_t, primitives.h:62-70; compute.comp:41-49 (96 B) */
typedef struct crt_quad {
    float corner[3]; float d;      /* d = dot(normal, corner) */
    float u[3];      float pad0;
    float v[3];      float pad1;
    float normal[3]; float pad2;   /* normalize(cross(u, v)) */
    float w[3];      float pad3;   /* n / dot(n, n), n = cross(u, v) */
    float pad4[3];
    uint32_t materialIndex;
} crt_quad;

/* Tri::GPU_t, primitives.h:123-131; compute.comp:51-59 (96 B).  A triangle of zero area (cross(v1 - v0, v2 - v0) == 0
 * exactly: a repeated vertex) is never hit: the shader's determinant is rounding noise for it and which rays it "hits" would
 * depend on the tree. */
typedef struct crt_tri {
    float v0[3]; float pad0;
    float v1[3]; float pad1;
    float v2[3]; float pad2;
    float u[3];  float pad3;       /* texture u at v0, v1, v2 */
    float v[3];  float pad4;       /* texture v at v0, v1, v2 */
    float pad5[3];
    uint32_t materialIndex;
} crt_tri;

/* RTMaterial::GPU_t, rt_material.h:20-31; compute.comp:25-32 (48 B) */
typedef struct crt_material {
    float albedo[3];
    float fuzziness;         /* metal fuzz, or index of refraction for a dielectric */
    float pad0[3];
    uint32_t type;
    float pad1[3];
    uint32_t textureIndex;   /* CRT_BAD_INDEX = untextured */
} crt_material;

/* BVHNode::GPU_t, bounding_volume_hierarchy.h:15-24; compute.comp:61-76 (64 B).
 * Only used when a caller supplies the reference's own threaded BVH for A/B runs. */
typedef struct crt_ref_bvh_node {
    float aabb_min[3]; float pad1;
    float aabb_max[3]; float pad2;
    uint32_t objectIndex;    /* CRT_BAD_INDEX on an inner node */
    uint32_t hitIndex;
    uint32_t missIndex;
    float pad3;
    uint32_t type;
    uint32_t numChildren;    /* 0 on an inner node */
    float pad4[2];
} crt_ref_bvh_node;

/* GPUCameraData, raytracing/include/camera.h:9-22; compute.comp:78-91 (96 B) */
typedef struct crt_camera_data {
    float position[3];   uint32_t shouldRenderAABB;
    float backward[3];   float lensRadius;
    float right[3];      float focusDistance;
    float up[3];         float iteration;      /* 1-based frame counter, as float */
    float horizontal[3]; float seed;           /* float(rand()) / 1E3f, vk_engine.cpp:1412 */
    float vertical[3];   float pad;
} crt_camera_data;

/* GPUSceneData, raytracing/include/scene.h:13-17; compute.comp:93-97 (112 B) */
typedef struct crt_scene_params {
    float backgroundColor[3];   /* (-1,-1,-1) = default sky gradient */
    float pad;
    crt_camera_data camera;
} crt_scene_params;

/* One RGBA8 UNORM texture, rows top-down, 1 mip (vk_engine.cpp:597-605). */
typedef struct crt_texture {
    const uint8_t *rgba8;
    uint32_t width;
    uint32_t height;
} crt_texture;

/* What init_descriptors binds once per scene (vk_engine.cpp:1177-1229): host
 * pointers to the six arrays.  Zero-length arrays may be NULL. */
typedef struct crt_scene_desc {
    const crt_sphere   *spheres;    uint32_t n_spheres;
    const crt_quad     *quads;      uint32_t n_quads;
    const crt_tri      *tris;       uint32_t n_tris;
    const crt_material *materials;  uint32_t n_materials;
    const crt_texture  *textures;   uint32_t n_textures;
} crt_scene_desc;

/*
 * Primitive IDs.  A primitive's ID is its position in the array it was
 * uploaded in, tagged with its kind: (kind << 30) | index, kind 0 = sphere,
 * 1 = quad, 2 = tri.  CRT_ID_MISS marks "no hit".  IDs do not depend on how the
 * BVH was built.  On exactly equal hit distances the lowest ID wins.
 */
#define CRT_ID_KIND_SPHERE 0u
#define CRT_ID_KIND_QUAD   1u
#define CRT_ID_KIND_TRI    2u
#define CRT_ID_MISS        0xFFFFFFFFu
#define CRT_MAKE_ID(kind, index) (((uint32_t)(kind) << 30) | (uint32_t)(index))

/* crt_rays_query: hit_world (compute.comp:326-361) for caller-supplied rays.  Both structs are plain FP32/u32. */
typedef struct crt_ray {
    float origin[3];
    float t_max;         /* hits are searched in (0.001, t_max]; INFINITY for an unbounded ray */
    float direction[3];  /* unit length, like every ray of the shader (hit_sphere, compute.comp:229-239, relies on it) */
    uint32_t pad;
} crt_ray;               /* 32 B */
typedef struct crt_ray_hit {
    float t;             /* distance of the hit; only meaningful when id != CRT_ID_MISS */
    uint32_t id;         /* primitive ID or CRT_ID_MISS */
} crt_ray_hit;
enum {
    CRT_QUERY_CLOSEST = 0,  /* the closest hit, lowest ID on exactly equal distances (what ray_color's hit_world returns) */
    CRT_QUERY_ANY = 1       /* shadow rays: id != CRT_ID_MISS iff SOME primitive is hit in the interval; which one is unspecified */
};

/* crt_comm_unique_id / crt_comm_init: size of the opaque communicator id (= NCCL_UNIQUE_ID_BYTES). */
#define CRT_COMM_ID_BYTES 128

/* crt_rays_query_noalloc: most rays one call takes (its queue lives in a static device array of
 * 32 bytes per ray + 128 KiB of counters = 4.125 MiB per GPU). */
#define CRT_NOALLOC_MAX_RAYS 131072u

/* crt_config.rng_mode */
enum {
    CRT_RNG_REF = 0,   /* the shader's float-seeded integer hash, compute.comp:116-139 (bit-exact) */
    CRT_RNG_PCG = 1    /* pcg32 stream keyed by (pixel, iteration): well-behaved quality mode */
};

/* crt_config.tracer */
enum {
    CRT_TRACER_MEGAKERNEL = 0,  /* one lane per pixel for the whole path (the shader's shape) */
    CRT_TRACER_WAVEFRONT  = 1   /* generate / extend / shade+compact queues */
};

/* crt_config.flags */
enum {
    CRT_FLAG_FULL_IMAGE = 1u << 0, /* render all W x H pixels; default renders floor(W/8)*8 x floor(H/8)*8
                                      like dispatch(W/8, H/8, 1), vk_engine.cpp:1435 */
    CRT_FLAG_DEBUG_SMALL_STACK = 1u << 1, /* test hook: the wavefront tracer keeps one traversal-stack entry per lane in LDS,
                                      so the scratch-memory overflow path (otherwise only reached by trees deeper than the
                                      LDS stack) runs on every scene */
    /* Small wavefront batches (up to 4.2 M paths: one sample of a 1080p frame) can run as a k_wf_extend / k_wf_shade pair
     * per bounce or, each path through all its bounces in one lane, in k_wf_tail.  The image is the same bit for bit;
     * which is faster depends on the scene, so by default the context times the forms over its first dispatches and keeps
     * the fastest (context.hip, TailTune).  These flags fix the choice. */
    CRT_FLAG_PER_BOUNCE_KERNELS = 1u << 2,
    CRT_FLAG_PATH_KERNEL = 1u << 3,           /* one lane per path, a thread per pixel (k_wf_tail) */
    CRT_FLAG_PATH_KERNEL_PERSISTENT = 1u << 4 /* persistent waves that refill idle lanes (k_wf_path; 8-wide tree without extensions,
                                                 else as CRT_FLAG_PATH_KERNEL) */
};

/* crt_extensions_set: build extensions with no counterpart in the reference, all off by default.
 * With none set the renderer is the reference's shader bit for bit. */
enum {
    CRT_EXT_TEXTURE_INDEX = 1u << 0, /* a textured Lambertian samples textures[material.textureIndex]; the shader always
                                        samples textures[0] whatever the index says (compute.comp:386-387) */
    CRT_EXT_ALPHA_TEST    = 1u << 1  /* cut-out textures: a triangle hit whose material is textured counts only where the
                                        texture's alpha, fetched like the colour (bilinear, REPEAT), is >= 0.5 at the hit's
                                        (u, v); applies to paths, crt_rays_query closest and any-hit (shadow rays) alike */
};

typedef struct crt_config {
    uint32_t width, height;     /* accumulation image size (imageSize() in the shader) */
    uint32_t rng_mode;
    uint32_t tracer;
    uint32_t flags;
    uint32_t max_bounces;       /* 0 = the shader's MAX_BOUNCES (10), compute.comp:16 */
    uint32_t band_rank;         /* this context renders the 8-row bands b with b % band_world == band_rank */
    uint32_t band_world;        /* 0 or 1 = whole image */
} crt_config;

/* crt_bvh_build flags */
enum {
    CRT_BVH_LBVH       = 0,        /* Morton LBVH only (fast rebuild path) */
    CRT_BVH_SAH_REFINE = 1u << 0,  /* + SAH refinement: the LBVH clusters, the topology is rebuilt top-down with a 16-bin SAH
                                      (nodes of more than 64 items) and an exact sweep SAH (smaller nodes); may close
                                      small same-kind subtrees as multi-primitive leaves */
    CRT_BVH_LEAF_COLLAPSE = 1u << 1, /* LBVH only: keep the Morton topology but close subtrees as multi-primitive leaves
                                      where the SAH prefers it (bottom-up, during the refit) */
    CRT_BVH_WIDE = 1u << 2,         /* + collapse the pair BVH into 8-wide nodes with 8-bit quantised child boxes
                                      (64 bytes per node, loaded whole per visit, leaf links in a separate array); the wavefront tracer walks these,
                                      the megakernel and crt_bvh_read keep using the pair BVH */
    CRT_BVH_WIDE_DIRECT = 1u << 3   /* the 8-wide tree built straight from the sorted Morton keys (bvh_direct.hip): the fastest rebuild,
                                      for per-frame rebuilds of large animated scenes.  No pair BVH exists afterwards: wavefront
                                      tracer and crt_rays_query only; crt_dispatch / crt_render with the megakernel tracer fail with
                                      CRT_ERR_INVALID and crt_bvh_read reports 0 nodes (the per-kind slot IDs are still read).  Not with CRT_BVH_SAH_REFINE / CRT_BVH_LEAF_COLLAPSE. */
};

/* The build's own BVH node (32 B).  Nodes are stored as sibling pairs:
 * pair p = nodes[2p], nodes[2p+1] (64 B, one half cache line).
 *   inner: link = index of its child pair, count = 0
 *   leaf:  link = first primitive slot in the kind's sorted array,
 *          count = (kind << 28) | number of primitives (>= 1)                */
typedef struct crt_bvh_node {
    float lo[3]; uint32_t link;
    float hi[3]; uint32_t count;
} crt_bvh_node;

typedef struct crt_stats {
    uint64_t rays;            /* hit_world calls (primary + every bounce segment) since last reset */
    uint64_t paths;           /* camera samples started */
    uint64_t node_visits;     /* only counted by instrumented launches */
    uint64_t prim_tests;      /* only counted by instrumented launches */
    double   last_render_ms;  /* device time of the last crt_dispatch/crt_render */
    double   last_build_ms;   /* device time of the last crt_bvh_build */
    double   last_extend_ms;  /* summed device time of the k_wf_extend launches of the last crt_dispatch/crt_render
                                 (only with crt_set_kernel_timing on and the wavefront tracer; else 0) */
    uint64_t last_extend_launches;
    double   last_shade_ms;   /* the same for the k_wf_shade launches */
    uint64_t last_shade_launches;
    double   last_primary_ms; /* the same for the k_wf_primary launches (bounce 0 over the 8-wide tree; their time and
                                 launches are not part of last_extend_*) */
    uint64_t last_primary_launches;
} crt_stats;

#ifdef __cplusplus
}
#endif

/* Layout contract: sizes and the offsets the shader's std140/std430 blocks imply.  Checked
 * in C11 and C++ alike, so a compiler that pads differently fails the build instead of the picture. */
#include <stddef.h>
#ifdef __cplusplus
#define CRT_STATIC_ASSERT(c, m) static_assert(c, m)
#else
#define CRT_STATIC_ASSERT(c, m) _Static_assert(c, m)
#endif
CRT_STATIC_ASSERT(sizeof(crt_sphere) == 32 && offsetof(crt_sphere, radius) == 12 && offsetof(crt_sphere, materialIndex) == 28, "T3 Sphere::GPU_t");
CRT_STATIC_ASSERT(sizeof(crt_quad) == 96 && offsetof(crt_quad, d) == 12 && offsetof(crt_quad, u) == 16 && offsetof(crt_quad, v) == 32 &&
                  offsetof(crt_quad, normal) == 48 && offsetof(crt_quad, w) == 64 && offsetof(crt_quad, materialIndex) == 92, "T4 Quad::GPU_t");
CRT_STATIC_ASSERT(sizeof(crt_tri) == 96 && offsetof(crt_tri, v1) == 16 && offsetof(crt_tri, v2) == 32 && offsetof(crt_tri, u) == 48 &&
                  offsetof(crt_tri, v) == 64 && offsetof(crt_tri, materialIndex) == 92, "T2 Tri::GPU_t");
CRT_STATIC_ASSERT(sizeof(crt_material) == 48 && offsetof(crt_material, fuzziness) == 12 && offsetof(crt_material, type) == 28 &&
                  offsetof(crt_material, textureIndex) == 44, "T5 RTMaterial::GPU_t");
CRT_STATIC_ASSERT(sizeof(crt_ref_bvh_node) == 64 && offsetof(crt_ref_bvh_node, aabb_max) == 16 && offsetof(crt_ref_bvh_node, objectIndex) == 32 &&
                  offsetof(crt_ref_bvh_node, hitIndex) == 36 && offsetof(crt_ref_bvh_node, missIndex) == 40 && offsetof(crt_ref_bvh_node, type) == 48 &&
                  offsetof(crt_ref_bvh_node, numChildren) == 52, "T1 BVHNode::GPU_t");
CRT_STATIC_ASSERT(sizeof(crt_camera_data) == 96 && offsetof(crt_camera_data, shouldRenderAABB) == 12 && offsetof(crt_camera_data, lensRadius) == 28 &&
                  offsetof(crt_camera_data, focusDistance) == 44 && offsetof(crt_camera_data, iteration) == 60 && offsetof(crt_camera_data, seed) == 76 &&
                  offsetof(crt_camera_data, vertical) == 80, "T6 GPUCameraData");
CRT_STATIC_ASSERT(sizeof(crt_scene_params) == 112 && offsetof(crt_scene_params, camera) == 16, "T6 GPUSceneData");
CRT_STATIC_ASSERT(sizeof(crt_bvh_node) == 32 && sizeof(crt_ray) == 32 && sizeof(crt_ray_hit) == 8, "build-side PODs");
#endif /* CRT_TYPES_H */
