/*
 * oracle.cpp -- CPU restatement of ComputeRaytracer's hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product (computeraytracer_amd/,
 * include/) includes, links or calls this file; only tests/, bench.py's
 * cpu_baseline leg and __graft_entry__.smoke() load liboracle.so, and only as
 * the checker.  It shares no source with the HIP kernels: the arithmetic is
 * typed a second time here, straight from the reference, so that a parity test
 * compares two independent statements of it.
 *
 * PARITY UNPINNED: the reference ships no tests, golden images or known-answer
 * vectors for this path and its renderer is a GLSL compute shader that cannot
 * be built or run in this environment (no glslang / Vulkan / GPU).  What pins
 * this file instead: an independent numpy restatement of the integer hash,
 * analytic known answers for the intersectors, brute force vs BVH equality,
 * glibc rand() for the seed schedule (tests/test_oracle_*.py).
 *
 * Reference files restated (relative to the upstream checkout):
 *   shaders/compute.comp                                  R1-R15
 *   raytracing/src/bounding_volume_hierarchy.cpp          B1, B2
 *   raytracing/src/axis_aligned_bounding_box.cpp, .h      B3
 *   raytracing/include/primitives.h, hittable.h           T2-T4, leaf boxes
 *   raytracing/src/camera.cpp                             C1
 *   engine/src/vk_engine.cpp:1412,1435                    seed schedule, grid
 *
 * FP32 contract (DESIGN.md "Arithmetic"): every operation below is an IEEE-754
 * binary32 operation in the order written, round-to-nearest-even, no fused
 * multiply-add, subnormals kept.  Build with -ffp-contract=off and without
 * -ffast-math.  GLSL leaves the precision of sin/cos/pow/acos/atan
 * implementation-defined, so the build fixes them as explicit polynomial
 * sequences (dm_* below); ORC_LIBM swaps in libm to measure how far that is
 * from a "typical" implementation.
 */
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

// ---------------------------------------------------------------- layouts ---
// Restated from the reference headers (not included from include/crt/types.h on
// purpose; the sizes are asserted so the two statements cannot drift apart).
struct Sphere   { float center[3]; float radius; float pad0[3]; uint32_t materialIndex; };            // primitives.h:29-34
struct Quad     { float corner[3]; float d; float u[3]; float p0; float v[3]; float p1;
                  float normal[3]; float p2; float w[3]; float p3; float p4[3]; uint32_t materialIndex; }; // primitives.h:62-70
struct Tri      { float v0[3]; float p0; float v1[3]; float p1; float v2[3]; float p2;
                  float u[3]; float p3; float v[3]; float p4; float p5[3]; uint32_t materialIndex; };  // primitives.h:123-131
struct Material { float albedo[3]; float fuzziness; float p0[3]; uint32_t type; float p1[3]; uint32_t textureIndex; }; // rt_material.h:20-31
struct RefNode  { float mn[3]; float p1; float mx[3]; float p2; uint32_t objectIndex, hitIndex, missIndex; float p3;
                  uint32_t type, numChildren; float p4[2]; };                                           // bounding_volume_hierarchy.h:15-24
struct Camera   { float position[3]; uint32_t shouldRenderAABB; float backward[3]; float lensRadius;
                  float right[3]; float focusDistance; float up[3]; float iteration;
                  float horizontal[3]; float seed; float vertical[3]; float pad; };                     // camera.h:9-22
struct Params   { float background[3]; float pad; Camera camera; };                                    // scene.h:13-17
struct Texture  { const uint8_t *rgba8; uint32_t width, height; };
struct BuildNode { float lo[3]; uint32_t link; float hi[3]; uint32_t count; };                          // the build's own 32-byte node
static_assert(sizeof(Sphere) == 32 && sizeof(Quad) == 96 && sizeof(Tri) == 96, "primitive layout");
static_assert(sizeof(Material) == 48 && sizeof(RefNode) == 64, "material/node layout");
static_assert(sizeof(Camera) == 96 && sizeof(Params) == 112 && sizeof(BuildNode) == 32, "uniform layout");

constexpr uint32_t BAD_INDEX = 0xFFFFFFFFu;
constexpr uint32_t ID_MISS = 0xFFFFFFFFu;
enum { KIND_SPHERE = 0, KIND_QUAD = 1, KIND_TRI = 2 };
enum { TYPE_SPHERE = 1, TYPE_QUAD = 2, TYPE_TRI = 4, TYPE_BVH = 8 };
enum { MAT_LAMBERTIAN = 1, MAT_METAL = 2, MAT_DIELECTRIC = 4, MAT_DIFFUSE_LIGHT = 8 };

struct Scene {
    const Sphere *spheres; uint32_t n_spheres;
    const Quad *quads; uint32_t n_quads;
    const Tri *tris; uint32_t n_tris;
    const Material *materials; uint32_t n_materials;
    const Texture *textures; uint32_t n_textures;
};

// render flags
enum {
    ORC_TRAVERSE_BRUTE = 0,      // every primitive, lowest-ID tie-break (canonical semantics)
    ORC_TRAVERSE_FAST  = 1,      // oracle's own median-split BVH, canonical semantics (== BRUTE, faster)
    ORC_TRAVERSE_REF   = 2,      // the reference's threaded BVH, strict '<' first-tested-wins (compute.comp:326-361)
    ORC_TRAVERSE_BUILD = 3,      // the build's pair BVH downloaded from the device, canonical semantics
    ORC_TRAVERSE_MASK  = 3,
    ORC_RNG_PCG        = 1 << 4, // pcg32 instead of the shader hash
    ORC_FULL_IMAGE     = 1 << 5, // render W x H instead of floor(W/8)*8 x floor(H/8)*8
    ORC_LIBM           = 1 << 6, // libm transcendentals instead of the fixed polynomial sequences
    // bits 8-15: band_world, bits 16-23: band_rank.  band_world > 1 renders only the 8-row bands b with
    // b % band_world == band_rank (the multi-GPU partition); the other rows of accum / ids stay untouched.
    ORC_BAND_WORLD_SHIFT = 8, ORC_BAND_RANK_SHIFT = 16,
    // Build extensions (no counterpart in the reference; off by default, SURVEY section 7 / abi.h crt_extensions_set):
    ORC_EXT_TEXTURE_INDEX = 1 << 24,   // a textured Lambertian samples textures[material.textureIndex] instead of textures[0]
    ORC_EXT_ALPHA_TEST    = 1 << 25,   // a triangle hit whose material texture has alpha < 0.5 at the hit's (u, v) is no hit
};

// ------------------------------------------------------------------- vec3 ---
struct V3 { float x, y, z; };
inline V3 v3(float x, float y, float z) { return {x, y, z}; }
inline V3 v3(const float *p) { return {p[0], p[1], p[2]}; }
inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator*(V3 a, V3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
inline V3 operator*(float s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }
inline V3 operator*(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline V3 operator/(V3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
inline V3 operator-(V3 a) { return {-a.x, -a.y, -a.z}; }
// GLSL dot(): x*x' + y*y' + z*z', left to right.
inline float dot(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
// GLSL cross().
inline V3 cross(V3 a, V3 b) { return {a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y}; }
// normalize(v) = v * (1 / sqrt(dot(v, v)))   [build-defined; GLSL: v * inversesqrt(dot)]
inline V3 normalize(V3 a) { float inv = 1.0f / sqrtf(dot(a, a)); return a * inv; }

inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

// -------------------------------------------------- fixed transcendentals ---
// Build-defined FP32 sequences (DESIGN.md "Arithmetic").  Coefficients are the
// classic Cephes single-precision minimax sets.
constexpr float PI_F = 3.14159265359f;      // compute.comp:3 (rounds to 0x40490FDB)
constexpr float TWO_PI_F = 2.0f * PI_F;

bool g_libm = false;   // set per render call (ORC_LIBM); only read inside dm_*

// sin and cos of th in [0, 2*pi] (the only range the shader uses them on).
inline void dm_sincos(float th, float *s, float *c) {
    if (g_libm) { *s = sinf(th); *c = cosf(th); return; }
    float qf = floorf(th * 0.636619772f + 0.5f);        // nearest multiple of pi/2
    int q = (int)qf;
    float r = th - qf * 1.5703125f;                     // three-term Cody-Waite
    r = r - qf * 4.837512969970703125e-4f;
    r = r - qf * 7.54978995489188216e-8f;
    float z = r * r;
    float sp = ((-1.9515295891e-4f * z + 8.3321608736e-3f) * z - 1.6666654611e-1f) * z * r + r;
    float cp = ((2.443315711809948e-5f * z - 1.388731625493765e-3f) * z + 4.166664568298827e-2f) * z * z;
    cp = cp - 0.5f * z;
    cp = cp + 1.0f;
    switch (q & 3) {
        case 0: *s = sp;  *c = cp;  break;
        case 1: *s = cp;  *c = -sp; break;
        case 2: *s = -sp; *c = -cp; break;
        default: *s = -cp; *c = sp; break;
    }
}

// pow(x, 0.3333333334) for x in [0, 1]  (compute.comp:152) as a cube root:
// exponent-divide seed + two Halley steps.
inline float dm_cbrt01(float x) {
    if (g_libm) return powf(x, 0.3333333334f);
    if (x == 0.0f) return 0.0f;
    float t = u2f(f2u(x) / 3u + 709958130u);
    for (int k = 0; k < 2; k++) {
        float t3 = t * t * t;
        t = t * ((t3 + (x + x)) / ((t3 + t3) + x));
    }
    return t;
}

// pow(x, 5.0)  (compute.comp:187)
inline float dm_pow5(float x) {
    if (g_libm) return powf(x, 5.0f);
    float x2 = x * x; float x4 = x2 * x2; return x4 * x;
}

inline float dm_asin_core(float x) {   // |x| <= 0.5
    float z = x * x;
    float p = ((((4.2163199048e-2f * z + 2.4181311049e-2f) * z + 4.5470025998e-2f) * z + 7.4953002686e-2f) * z
               + 1.6666752422e-1f) * z * x + x;
    return p;
}

// acos(x), x in [-1, 1]  (compute.comp:247)
inline float dm_acos(float x) {
    if (g_libm) return acosf(x);
    if (x < -0.5f) return PI_F - 2.0f * dm_asin_core(sqrtf((1.0f + x) * 0.5f));
    if (x > 0.5f) return 2.0f * dm_asin_core(sqrtf((1.0f - x) * 0.5f));
    return 1.5707963267948966f - dm_asin_core(x);
}

inline float dm_atan_pos(float x) {   // x >= 0
    float y;
    if (x > 2.414213562373095f) { y = 1.5707963267948966f; x = -(1.0f / x); }
    else if (x > 0.4142135623730950f) { y = 0.7853981633974483f; x = (x - 1.0f) / (x + 1.0f); }
    else y = 0.0f;
    float z = x * x;
    y = y + ((((8.05374449538e-2f * z - 1.38776856032e-1f) * z + 1.99777106478e-1f) * z - 3.33329491539e-1f) * z * x + x);
    return y;
}

// atan(y, x)  (compute.comp:248)
inline float dm_atan2(float y, float x) {
    if (g_libm) return atan2f(y, x);
    if (x == 0.0f) {
        if (y > 0.0f) return 1.5707963267948966f;
        if (y < 0.0f) return -1.5707963267948966f;
        return 0.0f;
    }
    float a = dm_atan_pos(fabsf(y / x));     // first-quadrant angle
    if (x < 0.0f) a = PI_F - a;
    return (y < 0.0f) ? -a : a;
}

// --------------------------------------------------------------- RNG (R1) ---
struct Rng {
    float seed;        // hashSeed, compute.comp:116
    uint64_t state;    // pcg32 state (ORC_RNG_PCG)
    uint64_t inc;
    bool pcg;
};

// compute.comp:118-122
inline uint32_t base_hash(uint32_t sx, uint32_t sy) {
    uint32_t nx = 1103515245u * ((sx >> 1) ^ sy);
    uint32_t ny = 1103515245u * ((sy >> 1) ^ sx);
    uint32_t h32 = 1103515245u * (nx ^ (ny >> 3));
    return h32 ^ (h32 >> 16);
}

inline uint32_t pcg32_next(Rng &r) {
    uint64_t old = r.state;
    r.state = old * 6364136223846793005ull + r.inc;
    uint32_t xorshifted = (uint32_t)(((old >> 18) ^ old) >> 27);
    uint32_t rot = (uint32_t)(old >> 59);
    return (xorshifted >> rot) | (xorshifted << ((32u - rot) & 31u));
}

// vec2(seed += 0.1, seed += 0.1) evaluated left to right, then floatBitsToUint.
inline uint32_t hash_n(Rng &r) {
    r.seed = r.seed + 0.1f; float a = r.seed;
    r.seed = r.seed + 0.1f; float b = r.seed;
    return base_hash(f2u(a), f2u(b));
}

// compute.comp:124-127.  float(0xFFFFFFFFU) rounds to 2^32, so the scale is 2^-32.
inline float hash_1(Rng &r) {
    uint32_t n = r.pcg ? pcg32_next(r) : hash_n(r);
    return (float)n * (1.0f / 4294967296.0f);
}
// compute.comp:129-133.  float(0x7FFFFFFF) rounds to 2^31.
inline void hash_2(Rng &r, float out[2]) {
    uint32_t n0, n1;
    if (r.pcg) { n0 = pcg32_next(r); n1 = pcg32_next(r); }
    else { uint32_t n = hash_n(r); n0 = n; n1 = n * 48271u; }
    out[0] = (float)(n0 & 0x7FFFFFFFu) / 2147483648.0f;
    out[1] = (float)(n1 & 0x7FFFFFFFu) / 2147483648.0f;
}
// compute.comp:135-139
inline void hash_3(Rng &r, float out[3]) {
    uint32_t n0, n1, n2;
    if (r.pcg) { n0 = pcg32_next(r); n1 = pcg32_next(r); n2 = pcg32_next(r); }
    else { uint32_t n = hash_n(r); n0 = n; n1 = n * 16807u; n2 = n * 48271u; }
    out[0] = (float)(n0 & 0x7FFFFFFFu) / 2147483648.0f;
    out[1] = (float)(n1 & 0x7FFFFFFFu) / 2147483648.0f;
    out[2] = (float)(n2 & 0x7FFFFFFFu) / 2147483648.0f;
}

// compute.comp:141-144
inline void random_in_unit_disk(Rng &r, float out[2]) {
    float h[2]; hash_2(r, h);
    float hx = h[0] * 1.0f, hy = h[1] * TWO_PI_F;
    float s, c; dm_sincos(hy, &s, &c);
    out[0] = hx * c; out[1] = hx * s;
}

// compute.comp:148-155
inline V3 random_in_unit_sphere(Rng &r) {
    float h[3]; hash_3(r, h);
    float theta = h[0] * TWO_PI_F - 0.0f;
    float hy = h[1] * 2.0f - 1.0f;
    float hz = h[2] * 1.0f - 0.0f;
    float sinPhi = sqrtf(1.0f - hy * hy);
    float rad = dm_cbrt01(hz);
    float s, c; dm_sincos(theta, &s, &c);
    return rad * v3(c * sinPhi, s * sinPhi, hy);
}

// ------------------------------------------------------------------- rays ---
struct Ray { V3 o, d; };

struct Hit {
    V3 position, normal;
    float t, u, v;
    bool front;
    uint32_t materialIndex;
    uint32_t id;           // (kind << 30) | input index; build-side addition
};

constexpr float T_NEAR = 0.001f;           // compute.comp:205
constexpr float INF_F = 3.402823466e+38f;  // compute.comp:4

struct Counters { uint64_t rays, node_visits, prim_tests, paths, bounces; };

// Acceptance of a candidate distance.  Reference: strict t < record.t, so the
// first primitive tested wins a tie (compute.comp:239,260,301).  Canonical:
// order-independent, lowest ID wins a tie.
inline bool accept(bool canonical, float t, uint32_t id, const Hit &rec) {
    if (!(T_NEAR < t)) return false;
    if (t < rec.t) return true;
    return canonical && t == rec.t && id < rec.id;
}

// compute.comp:207-210
inline void set_face_normal(const Ray &ray, V3 outward, Hit &rec) {
    rec.front = dot(ray.d, outward) < 0.0f;
    rec.normal = rec.front ? outward : -outward;
}

// compute.comp:229-251
inline void hit_sphere(const Ray &ray, const Sphere &sp, uint32_t id, bool canonical, Hit &rec) {
    V3 c = v3(sp.center);
    V3 rel = ray.o - c;
    float b = dot(rel, ray.d);
    V3 qc = rel - b * ray.d;
    float disc = sp.radius * sp.radius - dot(qc, qc);
    if (disc < 0.0f) return;
    float root = -b - sqrtf(disc);
    if (!accept(canonical, root, id, rec)) return;
    rec.t = root; rec.id = id;
    rec.position = ray.o + rec.t * ray.d;
    V3 outward = (rec.position - c) / sp.radius;
    set_face_normal(ray, outward, rec);
    rec.materialIndex = sp.materialIndex;
    float theta = dm_acos(-outward.y);
    float phi = dm_atan2(-outward.z, outward.x) + PI_F;
    rec.u = phi / (2.0f * PI_F);
    rec.v = theta / PI_F;
}

// compute.comp:253-276
inline void hit_quad(const Ray &ray, const Quad &q, uint32_t id, bool canonical, Hit &rec) {
    V3 n = v3(q.normal);
    float denom = dot(n, ray.d);
    if (fabsf(denom) < 1e-8f) return;
    float t = (q.d - dot(n, ray.o)) / denom;
    if (!accept(canonical, t, id, rec)) return;
    V3 isect = ray.o + t * ray.d;
    V3 planar = isect - v3(q.corner);
    float alpha = dot(v3(q.w), cross(planar, v3(q.v)));
    float beta = dot(v3(q.w), cross(v3(q.u), planar));
    // clamp(a,0,1) != a  <=>  not (0 <= a <= 1)
    if (!(alpha >= 0.0f && alpha <= 1.0f) || !(beta >= 0.0f && beta <= 1.0f)) return;
    rec.t = t; rec.id = id;
    rec.position = isect;
    set_face_normal(ray, n, rec);
    rec.materialIndex = q.materialIndex;
    rec.u = alpha; rec.v = beta;
}

// Build extension ORC_EXT_ALPHA_TEST: the alpha channel through the same sampler as the colour fetch (T8: RGBA8 UNORM,
// bilinear, REPEAT, texel centres at +0.5, plain float weights in the order sample_texture uses).
struct ExtCtx { const Scene *sc; bool tex_index, alpha_test; };
inline float texture_alpha(const Texture &tx, float u, float v) {
    float x = u * (float)tx.width - 0.5f, y = v * (float)tx.height - 0.5f;
    float x0f = floorf(x), y0f = floorf(y);
    float fx = x - x0f, fy = y - y0f;
    auto wrap = [](float f, int n) { int i = (int)f % n; return i < 0 ? i + n : i; };
    int x0 = wrap(x0f, (int)tx.width), y0 = wrap(y0f, (int)tx.height);
    int x1 = x0 + 1 == (int)tx.width ? 0 : x0 + 1, y1 = y0 + 1 == (int)tx.height ? 0 : y0 + 1;
    auto a = [&](int xx, int yy) { return (float)tx.rgba8[4 * ((size_t)yy * tx.width + (size_t)xx) + 3] / 255.0f; };
    float top = a(x0, y0) * (1.0f - fx) + a(x1, y0) * fx;
    float bot = a(x0, y1) * (1.0f - fx) + a(x1, y1) * fx;
    return top * (1.0f - fy) + bot * fy;
}
// The texture a material samples: textures[0] in the reference (compute.comp:387), textures[textureIndex] with
// ORC_EXT_TEXTURE_INDEX; null for an untextured material (or an index past the array, which the device refuses at upload).
inline const Texture *material_texture(const Scene &sc, const Material &m, bool tex_index) {
    if (m.textureIndex == BAD_INDEX || sc.n_textures == 0) return nullptr;
    if (!tex_index) return &sc.textures[0];
    return m.textureIndex < sc.n_textures ? &sc.textures[m.textureIndex] : nullptr;
}

// compute.comp:280-315
inline void hit_tri(const Ray &ray, const Tri &tr, uint32_t id, bool canonical, Hit &rec, const ExtCtx *ext = nullptr) {
    V3 v0 = v3(tr.v0);
    V3 e10 = v3(tr.v1) - v0;
    V3 e20 = v3(tr.v2) - v0;
    if (canonical) {
        // Zero-area triangles (a repeated vertex) are never hit: their determinant is rounding noise, and which rays "hit"
        // them then depends on the tree (a hit can lie outside the triangle's box), so no order-independent answer exists.
        // The device zeroes their edges at build time (bvh_pack.h: tri_record_head); ORC_TRAVERSE_REF keeps the shader's way.
        const V3 n0 = cross(e10, e20);
        if (n0.x == 0.0f && n0.y == 0.0f && n0.z == 0.0f) return;
    }
    V3 p = cross(ray.d, e20);
    float det = dot(e10, p);
    if (fabsf(det) < 1e-8f) return;
    V3 r0 = ray.o - v0;
    V3 q = cross(r0, e10);
    float beta = dot(r0, p);
    float gamma = dot(ray.d, q);
    if (beta < 0.0f || gamma < 0.0f || (beta + gamma) > det) return;
    float invDet = 1.0f / det;
    float t = dot(e20, q) * invDet;
    if (!accept(canonical, t, id, rec)) return;
    beta = beta * invDet;
    gamma = gamma * invDet;
    float alpha = 1.0f - beta - gamma;
    const float u = (alpha * tr.u[0] + beta * tr.u[1]) + gamma * tr.u[2];
    const float v = (alpha * tr.v[0] + beta * tr.v[1]) + gamma * tr.v[2];
    if (ext && ext->alpha_test) {                 // build extension: a cut-out texel lets the ray through
        const Texture *tx = material_texture(*ext->sc, ext->sc->materials[tr.materialIndex], ext->tex_index);
        if (tx && texture_alpha(*tx, u, v) < 0.5f) return;
    }
    rec.t = t; rec.id = id;
    rec.position = ray.o + rec.t * ray.d;
    V3 outward = cross(e20, e10);                 // not normalised (sic)
    set_face_normal(ray, outward, rec);
    rec.materialIndex = tr.materialIndex;
    rec.u = u;
    rec.v = v;
}

// compute.comp:214-225.  min()/max() with a NaN operand (0 * inf) are undefined
// in GLSL; here fminf/fmaxf return the other operand, i.e. that slab does not
// constrain the interval.
inline bool hit_aabb(const Ray &ray, const float mn[3], const float mx[3], float t, V3 inv) {
    V3 tMin = (v3(mn) - ray.o) * inv;
    V3 tMax = (v3(mx) - ray.o) * inv;
    V3 t0 = {fminf(tMin.x, tMax.x), fminf(tMin.y, tMax.y), fminf(tMin.z, tMax.z)};
    V3 t1 = {fmaxf(tMin.x, tMax.x), fmaxf(tMin.y, tMax.y), fmaxf(tMin.z, tMax.z)};
    float tNearest = fmaxf(fmaxf(fmaxf(t0.x, t0.y), t0.z), T_NEAR);
    float tFurthest = fminf(fminf(fminf(t1.x, t1.y), t1.z), t);
    return tNearest < tFurthest;
}

// ----------------------------------------------------------- AABB (B3) ------
struct AABB {
    V3 mn{0, 0, 0}, mx{0, 0, 0};           // default box is the point (0,0,0): axis_aligned_bounding_box.h:10-15
    AABB() = default;
    AABB(V3 a, V3 b) {                      // axis_aligned_bounding_box.cpp:3-6
        mn = {fminf(a.x, b.x), fminf(a.y, b.y), fminf(a.z, b.z)};
        mx = {fmaxf(a.x, b.x), fmaxf(a.y, b.y), fmaxf(a.z, b.z)};
    }
    AABB(V3 a, V3 b, V3 c) : AABB(a, b) {   // :8-11
        mn = {fminf(mn.x, c.x), fminf(mn.y, c.y), fminf(mn.z, c.z)};
        mx = {fmaxf(mx.x, c.x), fmaxf(mx.y, c.y), fmaxf(mx.z, c.z)};
    }
    AABB(const AABB &a, const AABB &b) {    // :13-16
        mn = {fminf(a.mn.x, b.mn.x), fminf(a.mn.y, b.mn.y), fminf(a.mn.z, b.mn.z)};
        mx = {fmaxf(a.mx.x, b.mx.x), fmaxf(a.mx.y, b.mx.y), fmaxf(a.mx.z, b.mx.z)};
    }
    AABB pad() {                            // :18-27  (glm::epsilonEqual: |a-b| < eps)
        const float delta = 0.0001f;
        if (fabsf(mx.x - mn.x) < 0.0001f) { mn.x -= delta / 2.0f; mx.x += delta / 2.0f; }
        if (fabsf(mx.y - mn.y) < 0.0001f) { mn.y -= delta / 2.0f; mx.y += delta / 2.0f; }
        if (fabsf(mx.z - mn.z) < 0.0001f) { mn.z -= delta / 2.0f; mx.z += delta / 2.0f; }
        return AABB(mn, mx);
    }
    float area() const {                    // :29-32
        V3 l = mx - mn;
        return 2.0f * ((l.x * l.y + l.y * l.z) + l.x * l.z);
    }
    float axis_min(int a) const { return a == 0 ? mn.x : a == 1 ? mn.y : mn.z; }
};

inline AABB prim_box(const Scene &sc, uint32_t id) {
    uint32_t kind = id >> 30, i = id & 0x3FFFFFFFu;
    if (kind == KIND_SPHERE) {                                   // primitives.h:41-43
        const Sphere &s = sc.spheres[i];
        V3 c = v3(s.center), r = v3(s.radius, s.radius, s.radius);
        return AABB(c - r, c + r);
    } else if (kind == KIND_QUAD) {                              // primitives.h:82-84
        const Quad &q = sc.quads[i];
        V3 c = v3(q.corner);
        return AABB(c, (c + v3(q.u)) + v3(q.v)).pad();
    } else {                                                     // primitives.h:138-140
        const Tri &t = sc.tris[i];
        return AABB(v3(t.v0), v3(t.v1), v3(t.v2)).pad();
    }
}

// --------------------------------------- reference BVH builder (B1, B2) -----
// A top-level world object: one primitive, or a HittableList of same-kind
// primitives (hittable.h:27-59), whose box starts from the default AABB and so
// also contains the origin.
struct RefObject { bool is_list; std::vector<uint32_t> prims; AABB box; };

struct RefTree {
    // node >= 0: inner node index into inner[]; node < 0: object ~node
    struct Inner { AABB box; int left, right; };
    std::vector<Inner> inner;
    const std::vector<RefObject> *objs;
    AABB box_of(int n) const { return n >= 0 ? inner[n].box : (*objs)[~n].box; }
};

// bounding_volume_hierarchy.cpp:9-20
float sah_cost(float areaLeft, float areaRight, int numLeft, int numRight) {
    const float costTraversal = 1.0f, costIntersection = 2.15f;
    float totalArea = areaLeft + areaRight;
    float pl = areaLeft / totalArea, pr = areaRight / totalArea;
    return costTraversal + (pl * (float)numLeft * costIntersection) + (pr * (float)numRight * costIntersection);
}

struct SplitInfo { int axis = -1, mid = -1; };

// bounding_volume_hierarchy.cpp:26-61.  The reference copies the whole vector
// by value; only [start, end) is read or written, so the copy is limited to it
// (same std::sort calls on the same element order => same result).
SplitInfo get_best_split(const std::vector<RefObject> &objs, const std::vector<int> &objects, int start, int end) {
    std::vector<int> local(objects.begin() + start, objects.begin() + end);
    const int n = end - start;
    SplitInfo best; float bestCost = 3.402823466e+38f;
    for (int axis = 0; axis < 3; axis++) {
        std::sort(local.begin(), local.end(), [&](int a, int b) { return objs[a].box.axis_min(axis) < objs[b].box.axis_min(axis); });
        AABB leftBounds, rightBounds;                       // both start at the origin point
        for (int i = 1; i < n; i++) rightBounds = AABB(rightBounds, objs[local[i]].box);   // fixed right bound (sic, :40)
        for (int mid = 1; mid < n; mid++) {
            leftBounds = AABB(leftBounds, objs[local[mid - 1]].box);
            float cost = sah_cost(leftBounds.area(), rightBounds.area(), mid, n - mid);
            if (cost < bestCost) { bestCost = cost; best.axis = axis; best.mid = start + mid; }
        }
    }
    return best;
}

// bounding_volume_hierarchy.cpp:63-79; returns inner-node index
int build_ref_node(RefTree &tree, std::vector<int> &objects, int start, int end) {
    const std::vector<RefObject> &objs = *tree.objs;
    int left, right;
    if (end - start == 1) {
        left = right = ~objects[start];
    } else {
        SplitInfo split = get_best_split(objs, objects, start, end);
        std::sort(objects.begin() + start, objects.begin() + end,
                  [&](int a, int b) { return objs[a].box.axis_min(split.axis) < objs[b].box.axis_min(split.axis); });
        left = (split.mid - start == 1) ? ~objects[start] : build_ref_node(tree, objects, start, split.mid);
        right = (end - split.mid == 1) ? ~objects[split.mid] : build_ref_node(tree, objects, split.mid, end);
    }
    RefTree::Inner node;
    node.left = left; node.right = right;
    node.box = AABB(tree.box_of(left), tree.box_of(right));
    tree.inner.push_back(node);
    return (int)tree.inner.size() - 1;
}

struct RefSerialized {
    std::vector<RefNode> bvh;
    std::vector<uint32_t> sphere_ids, quad_ids, tri_ids;   // serialized slot -> input id
};

RefNode default_ref_node() {   // BVHNode::GPU_t{} (bounding_volume_hierarchy.h:15-24); empty std::any slots upload as this (vk_engine.cpp:675)
    RefNode n; memset(&n, 0, sizeof n); n.objectIndex = BAD_INDEX; return n;
}

// bounding_volume_hierarchy.cpp:89-126
void serialize_ref(const RefTree &tree, int node, uint32_t nextRight, uint32_t nodeIndex, RefSerialized &out) {
    if (node < 0) {
        const RefObject &o = (*tree.objs)[~node];
        uint32_t kind = o.prims[0] >> 30;
        std::vector<uint32_t> &buf = kind == KIND_SPHERE ? out.sphere_ids : kind == KIND_QUAD ? out.quad_ids : out.tri_ids;
        RefNode leaf = default_ref_node();
        uint32_t startIndex = (uint32_t)buf.size();
        for (uint32_t id : o.prims) buf.push_back(id);
        memcpy(leaf.mn, &o.box.mn, 12); memcpy(leaf.mx, &o.box.mx, 12);
        leaf.objectIndex = startIndex;
        leaf.type = kind == KIND_SPHERE ? TYPE_SPHERE : kind == KIND_QUAD ? TYPE_QUAD : TYPE_TRI;
        leaf.numChildren = (uint32_t)buf.size() - startIndex;
        leaf.hitIndex = nextRight; leaf.missIndex = nextRight;
        out.bvh[nodeIndex] = leaf;
    } else {
        const RefTree::Inner &in = tree.inner[node];
        uint32_t leftIndex = (uint32_t)out.bvh.size() + 1;
        uint32_t rightIndex = (uint32_t)out.bvh.size() + 2;
        out.bvh.resize(out.bvh.size() + 3, default_ref_node());
        RefNode n = default_ref_node();
        memcpy(n.mn, &in.box.mn, 12); memcpy(n.mx, &in.box.mx, 12);
        n.hitIndex = leftIndex; n.missIndex = nextRight;
        out.bvh[nodeIndex] = n;
        serialize_ref(tree, in.left, rightIndex, leftIndex, out);
        serialize_ref(tree, in.right, nextRight, rightIndex, out);
    }
}

// --------------------------------- oracle's own BVH (canonical semantics) ---
// Median split over centroids, boxes kept in double and tested in double, so a
// primitive is never culled by rounding: results equal brute force.
struct FastBVH {
    struct Node { double lo[3], hi[3]; int left, right, first, count; };
    std::vector<Node> nodes;
    std::vector<uint32_t> prims;
};

void fast_bounds(const Scene &sc, uint32_t id, double lo[3], double hi[3]) {
    AABB b = prim_box(sc, id);
    double m[3] = {b.mn.x, b.mn.y, b.mn.z}, M[3] = {b.mx.x, b.mx.y, b.mx.z};
    if ((id >> 30) == KIND_QUAD) {
        // The reference's quad box spans two opposite corners only (primitives.h:82-84: AABB(corner, corner + u + v)),
        // which leaves the other two corners of a parallelogram outside unless u and v run along the axes (they do in
        // every reference scene).  Its threaded BVH (ORC_TRAVERSE_REF) therefore clips such a quad to that box; the
        // canonical semantics are those of brute force, so this tree bounds all four corners.
        const Quad &q = sc.quads[id & 0x3FFFFFFFu];
        for (int a = 0; a < 3; a++) {
            const double c = q.corner[a], cu = c + (double)q.u[a], cv = c + (double)q.v[a], cuv = c + (double)q.u[a] + (double)q.v[a];
            m[a] = std::min(std::min(c, cu), std::min(cv, cuv)); M[a] = std::max(std::max(c, cu), std::max(cv, cuv));
        }
    }
    for (int a = 0; a < 3; a++) {
        double padv = 1e-5 * (fabs(m[a]) + fabs(M[a]) + (M[a] - m[a])) + 1e-7;
        lo[a] = m[a] - padv; hi[a] = M[a] + padv;
    }
}

int fast_build(const Scene &sc, FastBVH &bvh, std::vector<double> &blo, std::vector<double> &bhi, int first, int count) {
    FastBVH::Node node;
    for (int a = 0; a < 3; a++) { node.lo[a] = 1e300; node.hi[a] = -1e300; }
    double clo[3] = {1e300, 1e300, 1e300}, chi[3] = {-1e300, -1e300, -1e300};
    for (int i = first; i < first + count; i++) {
        uint32_t slot = bvh.prims[i];
        for (int a = 0; a < 3; a++) {
            node.lo[a] = std::min(node.lo[a], blo[3 * (size_t)slot + a]);
            node.hi[a] = std::max(node.hi[a], bhi[3 * (size_t)slot + a]);
            double c = 0.5 * (blo[3 * (size_t)slot + a] + bhi[3 * (size_t)slot + a]);
            clo[a] = std::min(clo[a], c); chi[a] = std::max(chi[a], c);
        }
    }
    node.left = node.right = -1; node.first = first; node.count = count;
    int idx = (int)bvh.nodes.size();
    bvh.nodes.push_back(node);
    if (count > 4) {
        int axis = 0;
        if (chi[1] - clo[1] > chi[axis] - clo[axis]) axis = 1;
        if (chi[2] - clo[2] > chi[axis] - clo[axis]) axis = 2;
        int mid = first + count / 2;
        std::nth_element(bvh.prims.begin() + first, bvh.prims.begin() + mid, bvh.prims.begin() + first + count,
                         [&](uint32_t a, uint32_t b) {
                             return blo[3 * (size_t)a + axis] + bhi[3 * (size_t)a + axis] < blo[3 * (size_t)b + axis] + bhi[3 * (size_t)b + axis];
                         });
        int l = fast_build(sc, bvh, blo, bhi, first, mid - first);
        int r = fast_build(sc, bvh, blo, bhi, mid, first + count - mid);
        bvh.nodes[idx].left = l; bvh.nodes[idx].right = r; bvh.nodes[idx].count = 0;
    }
    return idx;
}

// ------------------------------------------------------------ the tracer ----
struct World {
    Scene sc;
    int traverse;
    bool canonical;
    ExtCtx ext{nullptr, false, false};      // build extensions (ORC_EXT_*); ext.sc points at this->sc
    // input-id list in slot order for BRUTE/FAST
    std::vector<uint32_t> all_ids;
    FastBVH fast;
    std::vector<uint32_t> fast_ids;     // slot -> input id
    // reference mode
    RefSerialized ref;
    // build mode: the device's pair BVH and its sorted primitive id lists
    const BuildNode *build_nodes = nullptr; uint32_t n_build_nodes = 0;
    const uint32_t *build_sphere_ids = nullptr, *build_quad_ids = nullptr, *build_tri_ids = nullptr;
};

inline void test_prim(const World &w, const Ray &ray, uint32_t id, Hit &rec, Counters &cnt) {
    cnt.prim_tests++;
    uint32_t kind = id >> 30, i = id & 0x3FFFFFFFu;
    if (kind == KIND_SPHERE) hit_sphere(ray, w.sc.spheres[i], id, w.canonical, rec);
    else if (kind == KIND_QUAD) hit_quad(ray, w.sc.quads[i], id, w.canonical, rec);
    else hit_tri(ray, w.sc.tris[i], id, w.canonical, rec, &w.ext);
}

inline bool slab_double(const double lo[3], const double hi[3], const Ray &ray, const double inv[3], double tmax) {
    double tn = T_NEAR * 0.5, tf = tmax;
    const double o[3] = {ray.o.x, ray.o.y, ray.o.z};
    for (int a = 0; a < 3; a++) {
        double t0 = (lo[a] - o[a]) * inv[a], t1 = (hi[a] - o[a]) * inv[a];
        if (t0 != t0 || t1 != t1) continue;     // 0 * inf: slab does not constrain
        if (t0 > t1) std::swap(t0, t1);
        if (t0 > tn) tn = t0;
        if (t1 < tf) tf = t1;
    }
    return tn <= tf;
}

// compute.comp:326-361 (ORC_TRAVERSE_REF) and its order-independent variants.
bool hit_world(const World &w, const Ray &ray, Hit &rec, Counters &cnt, float tmax = INF_F) {
    cnt.rays++;
    rec.t = tmax; rec.id = ID_MISS;
    if (w.traverse == ORC_TRAVERSE_BRUTE) {
        for (uint32_t id : w.all_ids) test_prim(w, ray, id, rec, cnt);
    } else if (w.traverse == ORC_TRAVERSE_FAST) {
        const double inv[3] = {1.0 / (double)ray.d.x, 1.0 / (double)ray.d.y, 1.0 / (double)ray.d.z};
        int stack[128], sp = 0;
        if (!w.fast.nodes.empty()) stack[sp++] = 0;
        while (sp) {
            const FastBVH::Node &n = w.fast.nodes[stack[--sp]];
            cnt.node_visits++;
            if (!slab_double(n.lo, n.hi, ray, inv, (double)rec.t * (1.0 + 1e-6))) continue;
            if (n.count) {
                for (int i = n.first; i < n.first + n.count; i++) test_prim(w, ray, w.fast_ids[w.fast.prims[i]], rec, cnt);
            } else { stack[sp++] = n.left; stack[sp++] = n.right; }
        }
    } else if (w.traverse == ORC_TRAVERSE_REF) {
        uint32_t next = 0;
        V3 inv = {1.0f / ray.d.x, 1.0f / ray.d.y, 1.0f / ray.d.z};
        const std::vector<RefNode> &bvh = w.ref.bvh;
        while (next != BAD_INDEX) {
            const RefNode &node = bvh[next];
            cnt.node_visits++;
            if (hit_aabb(ray, node.mn, node.mx, rec.t, inv)) {
                if (node.numChildren != 0) {
                    uint32_t s = node.objectIndex, e = s + node.numChildren;
                    const std::vector<uint32_t> &ids = node.type == TYPE_SPHERE ? w.ref.sphere_ids
                                                       : node.type == TYPE_QUAD ? w.ref.quad_ids : w.ref.tri_ids;
                    if (node.type == TYPE_SPHERE || node.type == TYPE_QUAD || node.type == TYPE_TRI)
                        for (uint32_t i = s; i < e; i++) test_prim(w, ray, ids[i], rec, cnt);
                }
                next = node.hitIndex;
            } else next = node.missIndex;
        }
        // compute.comp:358 -- quads[0] is tested once more; reading the 1-byte dummy buffer of a
        // quad-less scene (vk_engine.cpp:1105) is undefined, so it is guarded.
        if (!w.ref.quad_ids.empty()) test_prim(w, ray, w.ref.quad_ids[0], rec, cnt);
    } else {   // ORC_TRAVERSE_BUILD: the device's pair BVH, canonical acceptance, exact (double) slabs
        const double inv[3] = {1.0 / (double)ray.d.x, 1.0 / (double)ray.d.y, 1.0 / (double)ray.d.z};
        uint32_t stack[128]; int sp = 0;
        if (w.n_build_nodes >= 2) stack[sp++] = 0;
        while (sp) {
            uint32_t pair = stack[--sp];
            for (int c = 0; c < 2; c++) {
                const BuildNode &n = w.build_nodes[2 * (size_t)pair + c];
                cnt.node_visits++;
                const double lo[3] = {n.lo[0], n.lo[1], n.lo[2]}, hi[3] = {n.hi[0], n.hi[1], n.hi[2]};
                if (!slab_double(lo, hi, ray, inv, (double)rec.t * (1.0 + 1e-6))) continue;
                uint32_t count = n.count & 0x0FFFFFFFu, kind = n.count >> 28;
                if (count) {
                    const uint32_t *ids = kind == KIND_SPHERE ? w.build_sphere_ids : kind == KIND_QUAD ? w.build_quad_ids : w.build_tri_ids;
                    for (uint32_t i = 0; i < count; i++) test_prim(w, ray, ids[n.link + i], rec, cnt);
                } else if (sp < 127) stack[sp++] = n.link;
            }
        }
    }
    return rec.t != INF_F;
}

// Texture fetch (T8): RGBA8 UNORM, bilinear, REPEAT, texel centres at +0.5
// (vk_initializers.cpp:252-261).  Hardware filter precision is unspecified; the
// build fixes plain float weights.
inline V3 texel(const Texture &tx, int x, int y) {
    const uint8_t *p = tx.rgba8 + 4 * ((size_t)y * tx.width + (size_t)x);
    return v3((float)p[0] / 255.0f, (float)p[1] / 255.0f, (float)p[2] / 255.0f);
}
inline int wrapi(float f, int n) {
    int i = (int)f % n;      // |f| is far below 2^31 for any sane uv
    return i < 0 ? i + n : i;
}
V3 sample_texture(const Texture &tx, float u, float v) {
    float x = u * (float)tx.width - 0.5f, y = v * (float)tx.height - 0.5f;
    float x0f = floorf(x), y0f = floorf(y);
    float fx = x - x0f, fy = y - y0f;
    int x0 = wrapi(x0f, (int)tx.width), y0 = wrapi(y0f, (int)tx.height);
    int x1 = x0 + 1 == (int)tx.width ? 0 : x0 + 1, y1 = y0 + 1 == (int)tx.height ? 0 : y0 + 1;
    V3 t00 = texel(tx, x0, y0), t10 = texel(tx, x1, y0), t01 = texel(tx, x0, y1), t11 = texel(tx, x1, y1);
    V3 top = t00 * (1.0f - fx) + t10 * fx;
    V3 bot = t01 * (1.0f - fx) + t11 * fx;
    return top * (1.0f - fy) + bot * fy;
}

// compute.comp:363-371
inline V3 emit(const World &w, const Hit &rec) {
    const Material &m = w.sc.materials[rec.materialIndex];
    return m.type == MAT_DIFFUSE_LIGHT ? v3(m.albedo) : v3(0, 0, 0);
}

// GLSL reflect(I, N) = I - 2 * dot(N, I) * N
inline V3 reflect(V3 I, V3 N) { return I - (2.0f * dot(N, I)) * N; }
// GLSL refract(I, N, eta)
inline V3 refract(V3 I, V3 N, float eta) {
    float ni = dot(N, I);
    float k = 1.0f - eta * eta * (1.0f - ni * ni);
    if (k < 0.0f) return v3(0, 0, 0);
    return eta * I - (eta * ni + sqrtf(k)) * N;
}

// compute.comp:176-191
bool should_refract(V3 incoming, V3 normal, float ior, float ratio, Rng &rng) {
    float cosTheta = fminf(dot(-incoming, normal), 1.0f);
    float sinTheta = sqrtf(1.0f - cosTheta * cosTheta);
    if (ratio * sinTheta > 0.999f) return false;
    float reflectance = (1.0f - ior) / (1.0f + ior);
    reflectance = reflectance * reflectance;
    float schlick = reflectance + (1.0f - reflectance) * dm_pow5(1.0f - cosTheta);
    if (schlick > hash_1(rng)) return false;
    return true;
}

// compute.comp:373-419
bool scatter(const World &w, Ray &ray, const Hit &rec, V3 &attenuation, Rng &rng) {
    const Material &m = w.sc.materials[rec.materialIndex];
    attenuation = v3(m.albedo);
    switch (m.type) {
        case MAT_LAMBERTIAN: {
            V3 dir = rec.normal + random_in_unit_sphere(rng);
            ray = Ray{rec.position, normalize(dir)};
            // :385-389 -- always samples textures[0], whatever textureIndex says (sic)
            // (ORC_EXT_TEXTURE_INDEX, a build extension, honours the index)
            if (const Texture *tx = material_texture(w.sc, m, w.ext.tex_index)) attenuation = sample_texture(*tx, rec.u, rec.v);
            return true;
        }
        case MAT_METAL: {
            V3 refl = reflect(ray.d, rec.normal);
            refl = refl + m.fuzziness * random_in_unit_sphere(rng);
            ray = Ray{rec.position, normalize(refl)};
            return dot(ray.d, rec.normal) > 0.0f;
        }
        case MAT_DIELECTRIC: {
            float ior = m.fuzziness;
            float ratio = rec.front ? 1.0f / ior : ior;
            V3 dir = should_refract(ray.d, rec.normal, ior, ratio, rng) ? refract(ray.d, rec.normal, ratio) : reflect(ray.d, rec.normal);
            ray = Ray{rec.position, normalize(dir)};
            attenuation = v3(1, 1, 1);
            return true;
        }
        default: return false;
    }
}

// compute.comp:163-172
Ray camera_get_ray(const Camera &cam, float u, float v, Rng &rng) {
    float disk[2]; random_in_unit_disk(rng, disk);
    float rx = cam.lensRadius * disk[0], ry = cam.lensRadius * disk[1];
    V3 offset = v3(cam.right) * rx + v3(cam.up) * ry;
    V3 llc = ((v3(cam.position) - v3(cam.horizontal) * 0.5f) - v3(cam.vertical) * 0.5f) - cam.focusDistance * v3(cam.backward);
    V3 origin = v3(cam.position) + offset;
    V3 dir = ((llc + u * v3(cam.horizontal)) + v * v3(cam.vertical)) - origin;
    return Ray{origin, normalize(dir)};
}

// compute.comp:421-458 (the AABB-outline debug mode :427-436 draws the
// reference's own BVH boxes and is only restated for ORC_TRAVERSE_REF).
V3 ray_color(const World &w, const Params &p, Ray ray, Rng &rng, int max_bounces, uint32_t *primary_id, Counters &cnt) {
    Hit rec;
    if (p.camera.shouldRenderAABB && w.traverse == ORC_TRAVERSE_REF) {
        V3 inv = {1.0f / ray.d.x, 1.0f / ray.d.y, 1.0f / ray.d.z};
        for (const RefNode &n : w.ref.bvh) {
            float mn2[3] = {n.mn[0] + 0.005f, n.mn[1] + 0.005f, n.mn[2] + 0.005f};
            float mx2[3] = {n.mx[0] - 0.005f, n.mx[1] - 0.005f, n.mx[2] - 0.005f};
            if (hit_aabb(ray, n.mn, n.mx, INF_F, inv) && !hit_aabb(ray, mn2, mx2, INF_F, inv)) return v3(1, 1, 1);
        }
    }
    V3 color = v3(1, 1, 1), emitted = v3(0, 0, 0), attenuation = v3(0, 0, 0);
    for (int depth = 0; depth < max_bounces; depth++) {
        bool hit = hit_world(w, ray, rec, cnt);
        if (depth == 0 && primary_id) *primary_id = hit ? rec.id : ID_MISS;
        if (!hit) {
            if (p.background[0] != -1.0f || p.background[1] != -1.0f || p.background[2] != -1.0f) return v3(p.background);
            float a = 0.5f * (ray.d.y + 1.0f);
            V3 sky = v3(1, 1, 1) * (1.0f - a) + v3(0.5f, 0.7f, 1.0f) * a;    // mix(x, y, a) = x*(1-a) + y*a
            return color * sky;
        }
        cnt.bounces++;
        emitted = emitted + color * emit(w, rec);
        if (!scatter(w, ray, rec, attenuation, rng)) return emitted;
        color = color * (attenuation + emitted);    // sic, :455
    }
    return v3(0, 0, 0);
}

inline void pcg_seed(Rng &r, uint32_t pixel, float seed, float iteration) {
    r.pcg = true;
    r.state = 0; r.inc = ((uint64_t)(uint32_t)iteration << 1) | 1u;
    pcg32_next(r);
    r.state += ((uint64_t)f2u(seed) << 32) | pixel;
    pcg32_next(r);
}

// compute.comp:460-477 for one pixel
void shade_pixel(const World &w, const Params &p, uint32_t W, uint32_t H, uint32_t x, uint32_t y, int flags, int max_bounces,
                 float *accum, uint32_t *ids, Counters &cnt) {
    Rng rng{};
    // :116 -- dot(vec2(x,y), vec2(y,x)) = x*y + y*x
    rng.seed = p.camera.seed * ((float)x * (float)y + (float)y * (float)x);
    if (flags & ORC_RNG_PCG) pcg_seed(rng, y * W + x, p.camera.seed, p.camera.iteration);
    float *px = accum + 4 * ((size_t)y * W + x);
    V3 cum = v3(px[0], px[1], px[2]);
    float off[2]; hash_2(rng, off);
    float u = ((float)x + off[0]) / (float)W, v = ((float)y + off[1]) / (float)H;
    cnt.paths++;
    uint32_t pid = ID_MISS;
    V3 pass = ray_color(w, p, camera_get_ray(p.camera, u, v, rng), rng, max_bounces, &pid, cnt);
    pass = pass * (1.0f / 1.0f);
    // :474 -- cum += int(iteration == 1) * -cum + pass
    float first = (float)(int)(p.camera.iteration == 1.0f);
    cum = cum + ((first * (-cum)) + pass);
    px[0] = cum.x; px[1] = cum.y; px[2] = cum.z; px[3] = p.camera.iteration;
    if (ids) ids[(size_t)y * W + x] = pid;
}

void world_init(World &w, const Scene &sc, int flags) {
    w.sc = sc;
    w.traverse = flags & ORC_TRAVERSE_MASK;
    w.canonical = w.traverse != ORC_TRAVERSE_REF;
    w.ext = ExtCtx{&w.sc, (flags & ORC_EXT_TEXTURE_INDEX) != 0, (flags & ORC_EXT_ALPHA_TEST) != 0};
    for (uint32_t i = 0; i < sc.n_spheres; i++) w.all_ids.push_back((KIND_SPHERE << 30) | i);
    for (uint32_t i = 0; i < sc.n_quads; i++) w.all_ids.push_back((KIND_QUAD << 30) | i);
    for (uint32_t i = 0; i < sc.n_tris; i++) w.all_ids.push_back(((uint32_t)KIND_TRI << 30) | i);
    if (w.traverse == ORC_TRAVERSE_FAST && !w.all_ids.empty()) {
        size_t n = w.all_ids.size();
        w.fast_ids = w.all_ids;
        std::vector<double> blo(3 * n), bhi(3 * n);
        for (size_t i = 0; i < n; i++) fast_bounds(sc, w.all_ids[i], &blo[3 * i], &bhi[3 * i]);
        w.fast.prims.resize(n);
        for (size_t i = 0; i < n; i++) w.fast.prims[i] = (uint32_t)i;
        fast_build(sc, w.fast, blo, bhi, 0, (int)n);
    }
}

void build_reference_bvh(const Scene &sc, uint32_t n_objects, const uint32_t *obj_offsets, const uint32_t *obj_prims,
                         const uint8_t *obj_is_list, RefSerialized &out) {
    std::vector<RefObject> objs(n_objects);
    for (uint32_t o = 0; o < n_objects; o++) {
        objs[o].is_list = obj_is_list && obj_is_list[o];
        objs[o].prims.assign(obj_prims + obj_offsets[o], obj_prims + obj_offsets[o + 1]);
        if (objs[o].is_list) {
            AABB box;                                               // hittable.h:36-39,58: starts at the origin point
            for (uint32_t id : objs[o].prims) box = AABB(box, prim_box(sc, id));
            objs[o].box = box;
        } else objs[o].box = prim_box(sc, objs[o].prims[0]);
    }
    RefTree tree; tree.objs = &objs;
    std::vector<int> order(n_objects);
    for (uint32_t o = 0; o < n_objects; o++) order[o] = (int)o;
    int root = build_ref_node(tree, order, 0, (int)n_objects);
    out.bvh.clear(); out.sphere_ids.clear(); out.quad_ids.clear(); out.tri_ids.clear();
    // gpu_serialize(): gpu_serialize_internal(scene, this, BAD_INDEX, 0) on an empty buffer; slot 0 appears with the first resize
    serialize_ref(tree, root, BAD_INDEX, 0, out);
}

}  // namespace

// =============================================================== C API ======
extern "C" {

struct orc_scene {
    const void *spheres; uint32_t n_spheres;
    const void *quads; uint32_t n_quads;
    const void *tris; uint32_t n_tris;
    const void *materials; uint32_t n_materials;
    const void *textures; uint32_t n_textures;   // array of {const uint8_t*, u32 w, u32 h}
};

struct orc_world_desc {      // what the reference's world generator pushes into `world` (vk_engine.cpp:825-926)
    uint32_t n_objects;
    const uint32_t *obj_offsets;   // n_objects + 1
    const uint32_t *obj_prims;     // input ids (kind << 30 | index)
    const uint8_t *obj_is_list;    // 1 = HittableList (box includes the origin), 0 = bare primitive
};

struct orc_build_bvh {       // the device's BVH, read back with crt_bvh_read()
    const void *nodes; uint32_t n_nodes;
    const uint32_t *sphere_ids, *quad_ids, *tri_ids;
};

struct orc_counters { uint64_t rays, node_visits, prim_tests, paths, bounces; };

static Scene to_scene(const orc_scene *s) {
    Scene sc;
    sc.spheres = (const Sphere *)s->spheres; sc.n_spheres = s->n_spheres;
    sc.quads = (const Quad *)s->quads; sc.n_quads = s->n_quads;
    sc.tris = (const Tri *)s->tris; sc.n_tris = s->n_tris;
    sc.materials = (const Material *)s->materials; sc.n_materials = s->n_materials;
    sc.textures = (const Texture *)s->textures; sc.n_textures = s->n_textures;
    return sc;
}

/* Render n_iter dispatches (params[i] = the 112-byte UBO of dispatch i) into
 * accum (W*H*4 floats, read-modify-write, row 0 = bottom).  ids (optional)
 * receives the primary-hit id of the LAST dispatch.  Returns 0 on success. */
int orc_render(const orc_scene *scene, const orc_world_desc *world, const orc_build_bvh *build,
               const void *params, uint32_t n_iter, uint32_t W, uint32_t H, int flags, int max_bounces, int threads,
               float *accum, uint32_t *ids, orc_counters *counters) {
    World w;
    world_init(w, to_scene(scene), flags);
    if (w.traverse == ORC_TRAVERSE_REF) {
        if (!world || world->n_objects == 0) return -1;
        build_reference_bvh(w.sc, world->n_objects, world->obj_offsets, world->obj_prims, world->obj_is_list, w.ref);
    }
    if (w.traverse == ORC_TRAVERSE_BUILD) {
        if (!build) return -2;
        w.build_nodes = (const BuildNode *)build->nodes; w.n_build_nodes = build->n_nodes;
        w.build_sphere_ids = build->sphere_ids; w.build_quad_ids = build->quad_ids; w.build_tri_ids = build->tri_ids;
    }
    g_libm = (flags & ORC_LIBM) != 0;
    if (max_bounces <= 0) max_bounces = 10;                      // MAX_BOUNCES, compute.comp:16
    uint32_t gw = (flags & ORC_FULL_IMAGE) ? W : (W / 8) * 8;    // dispatch(W/8, H/8, 1), vk_engine.cpp:1435
    uint32_t gh = (flags & ORC_FULL_IMAGE) ? H : (H / 8) * 8;
    const Params *ps = (const Params *)params;
    const uint32_t band_world = ((uint32_t)flags >> ORC_BAND_WORLD_SHIFT) & 0xFFu, band_rank = ((uint32_t)flags >> ORC_BAND_RANK_SHIFT) & 0xFFu;
    if (band_world > 1 && band_rank >= band_world) return -3;
    Counters total{};
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#endif
    for (uint32_t it = 0; it < n_iter; it++) {
        const Params &p = ps[it];
        uint64_t rays = 0, nv = 0, pt = 0, paths = 0, bounces = 0;
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : rays, nv, pt, paths, bounces)
        for (int64_t y = 0; y < (int64_t)gh; y++) {
            Counters c{};
            if (band_world > 1 && ((uint32_t)y / 8u) % band_world != band_rank) continue;
            for (uint32_t x = 0; x < gw; x++)
                shade_pixel(w, p, W, H, x, (uint32_t)y, flags, max_bounces, accum, it + 1 == n_iter ? ids : nullptr, c);
            rays += c.rays; nv += c.node_visits; pt += c.prim_tests; paths += c.paths; bounces += c.bounces;
        }
        total.rays += rays; total.node_visits += nv; total.prim_tests += pt; total.paths += paths; total.bounces += bounces;
    }
    if (counters) { counters->rays = total.rays; counters->node_visits = total.node_visits; counters->prim_tests = total.prim_tests;
                    counters->paths = total.paths; counters->bounces = total.bounces; }
    return 0;
}

/* Primary rays of ONE dispatch only (main() up to the first hit_world call, compute.comp:460-468 + :442): per pixel the
 * distance and ID of the primary hit (t = +inf, ID_MISS where the camera ray hits nothing).  Used to measure how far
 * apart the reference's first-tested-wins result and the canonical lowest-ID result are where they differ. */
int orc_primary_hits(const orc_scene *scene, const orc_world_desc *world, const void *params, uint32_t W, uint32_t H, int flags,
                     int threads, float *t_out, uint32_t *ids) {
    World w;
    world_init(w, to_scene(scene), flags);
    if (w.traverse == ORC_TRAVERSE_REF) {
        if (!world || world->n_objects == 0) return -1;
        build_reference_bvh(w.sc, world->n_objects, world->obj_offsets, world->obj_prims, world->obj_is_list, w.ref);
    }
    if (w.traverse == ORC_TRAVERSE_BUILD) return -2;
    g_libm = (flags & ORC_LIBM) != 0;
    const Params &p = *(const Params *)params;
    const uint32_t gw = (flags & ORC_FULL_IMAGE) ? W : (W / 8) * 8, gh = (flags & ORC_FULL_IMAGE) ? H : (H / 8) * 8;
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#endif
#pragma omp parallel for schedule(dynamic, 1)
    for (int64_t y = 0; y < (int64_t)gh; y++)
        for (uint32_t x = 0; x < gw; x++) {
            Rng rng{};
            rng.seed = p.camera.seed * ((float)x * (float)y + (float)y * (float)x);
            if (flags & ORC_RNG_PCG) pcg_seed(rng, (uint32_t)y * W + x, p.camera.seed, p.camera.iteration);
            float off[2]; hash_2(rng, off);
            const float u = ((float)x + off[0]) / (float)W, v = ((float)y + off[1]) / (float)H;
            Hit rec{}; Counters c{};
            const bool hit = hit_world(w, camera_get_ray(p.camera, u, v, rng), rec, c);
            t_out[(size_t)y * W + x] = hit ? rec.t : INF_F;
            ids[(size_t)y * W + x] = hit ? rec.id : ID_MISS;
        }
    return 0;
}

/* hit_world for caller-supplied rays (the checker of crt_rays_query): rays = n x {origin[3], t_max, direction[3], pad},
 * out = n x {t, id}; hits are searched in (T_NEAR, t_max].  BRUTE / FAST traversal only (canonical tie-break). */
int orc_hit_rays(const orc_scene *scene, int flags, const float *rays, uint32_t n, int threads, void *out) {
    World w;
    world_init(w, to_scene(scene), flags);
    if (w.traverse != ORC_TRAVERSE_BRUTE && w.traverse != ORC_TRAVERSE_FAST) return -1;
    struct RayHit { float t; uint32_t id; };
    RayHit *o = (RayHit *)out;
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#endif
#pragma omp parallel for schedule(dynamic, 64)
    for (int64_t i = 0; i < (int64_t)n; i++) {
        const float *r = rays + 8 * i;
        Ray ray{v3(r), v3(r + 4)};
        Hit rec{}; Counters c{};
        hit_world(w, ray, rec, c, r[3]);
        o[i].t = rec.t; o[i].id = rec.id;
    }
    return 0;
}

/* B1 + B2: build the reference's BVH for `world` and serialise it.  Two-call
 * protocol: pass NULL outputs to get the sizes. */
int orc_ref_bvh(const orc_scene *scene, const orc_world_desc *world, void *nodes_out, uint32_t *n_nodes,
                uint32_t *sphere_ids, uint32_t *n_sphere_ids, uint32_t *quad_ids, uint32_t *n_quad_ids,
                uint32_t *tri_ids, uint32_t *n_tri_ids) {
    RefSerialized ser;
    Scene sc = to_scene(scene);
    build_reference_bvh(sc, world->n_objects, world->obj_offsets, world->obj_prims, world->obj_is_list, ser);
    *n_nodes = (uint32_t)ser.bvh.size(); *n_sphere_ids = (uint32_t)ser.sphere_ids.size();
    *n_quad_ids = (uint32_t)ser.quad_ids.size(); *n_tri_ids = (uint32_t)ser.tri_ids.size();
    if (nodes_out) memcpy(nodes_out, ser.bvh.data(), ser.bvh.size() * sizeof(RefNode));
    if (sphere_ids) memcpy(sphere_ids, ser.sphere_ids.data(), ser.sphere_ids.size() * 4);
    if (quad_ids) memcpy(quad_ids, ser.quad_ids.data(), ser.quad_ids.size() * 4);
    if (tri_ids) memcpy(tri_ids, ser.tri_ids.data(), ser.tri_ids.size() * 4);
    return 0;
}

/* ---- unit hooks (tests/test_oracle_*.py) ---- */

/* kind 1/2/3 = hash_1/2/3.  seed is updated in place like the shader's inout. */
void orc_hash(float *seed, int kind, float out[3]) {
    Rng r{}; r.seed = *seed; r.pcg = false;
    out[0] = out[1] = out[2] = 0.0f;
    if (kind == 1) out[0] = hash_1(r); else if (kind == 2) hash_2(r, out); else hash_3(r, out);
    *seed = r.seed;
}
uint32_t orc_base_hash(uint32_t sx, uint32_t sy) { return base_hash(sx, sy); }
float orc_pixel_seed(float seed, uint32_t x, uint32_t y) { return seed * ((float)x * (float)y + (float)y * (float)x); }

void orc_pcg(uint32_t pixel, float seed, float iteration, uint32_t n, uint32_t *out) {
    Rng r{}; pcg_seed(r, pixel, seed, iteration);
    for (uint32_t i = 0; i < n; i++) out[i] = pcg32_next(r);
}

void orc_math(int fn, uint32_t n, const float *a, const float *b, float *out0, float *out1, int libm) {
    g_libm = libm != 0;
    for (uint32_t i = 0; i < n; i++) {
        switch (fn) {
            case 0: dm_sincos(a[i], &out0[i], &out1[i]); break;
            case 1: out0[i] = dm_cbrt01(a[i]); break;
            case 2: out0[i] = dm_acos(a[i]); break;
            case 3: out0[i] = dm_atan2(a[i], b[i]); break;
            case 4: out0[i] = dm_pow5(a[i]); break;
            case 5: out0[i] = sqrtf(a[i]); break;
            case 6: out0[i] = a[i] / b[i]; break;
        }
    }
    g_libm = false;
}

/* One ray against one primitive (kind 0/1/2).  out = {t, px,py,pz, nx,ny,nz, u, v, front}; returns 1 on a hit. */
int orc_hit_prim(int kind, const void *prim, const float origin[3], const float dir[3], float tmax, float out[10]) {
    Ray ray{v3(origin), v3(dir)};
    Hit rec{}; rec.t = tmax; rec.id = ID_MISS;
    if (kind == 0) hit_sphere(ray, *(const Sphere *)prim, 0, false, rec);
    else if (kind == 1) hit_quad(ray, *(const Quad *)prim, 0, false, rec);
    else hit_tri(ray, *(const Tri *)prim, 0, false, rec);
    if (rec.id == ID_MISS) return 0;
    out[0] = rec.t; out[1] = rec.position.x; out[2] = rec.position.y; out[3] = rec.position.z;
    out[4] = rec.normal.x; out[5] = rec.normal.y; out[6] = rec.normal.z; out[7] = rec.u; out[8] = rec.v; out[9] = rec.front ? 1.0f : 0.0f;
    return 1;
}

int orc_hit_aabb(const float mn[3], const float mx[3], const float origin[3], const float dir[3], float t) {
    Ray ray{v3(origin), v3(dir)};
    V3 inv = {1.0f / ray.d.x, 1.0f / ray.d.y, 1.0f / ray.d.z};
    return hit_aabb(ray, mn, mx, t, inv) ? 1 : 0;
}

void orc_sample_texture(const uint8_t *rgba8, uint32_t w, uint32_t h, float u, float v, float out[3]) {
    Texture tx{rgba8, w, h};
    V3 c = sample_texture(tx, u, v);
    out[0] = c.x; out[1] = c.y; out[2] = c.z;
}

/* camera_get_ray for one uv, one seed value. */
void orc_camera_ray(const void *params, float u, float v, float seed_state, float out[6]) {
    const Params *p = (const Params *)params;
    Rng r{}; r.seed = seed_state;
    Ray ray = camera_get_ray(p->camera, u, v, r);
    out[0] = ray.o.x; out[1] = ray.o.y; out[2] = ray.o.z; out[3] = ray.d.x; out[4] = ray.d.y; out[5] = ray.d.z;
}

/* C1: Camera::Camera + calculateProperties (camera.cpp:3-12,40-62), restated
 * to cross-check the host library's camera builder. */
void orc_camera(const float position[3], const float at[3], float fovDegrees, float aspect, float aperture, float focus, void *out_camera) {
    Camera c; memset(&c, 0, sizeof c);
    V3 pos = v3(position);
    V3 backward = normalize(pos - v3(at));
    float theta = fovDegrees * 0.01745329251994329576923690768489f;   // glm::radians
    float viewportHeight = 2.0f * tanf(theta / 2.0f);
    float viewportWidth = aspect * viewportHeight;
    V3 right = normalize(cross(v3(0, 1, 0), backward));
    V3 up = cross(backward, right);
    V3 horizontal = (focus * viewportWidth) * right;
    V3 vertical = (focus * viewportHeight) * up;
    memcpy(c.position, &pos, 12); memcpy(c.backward, &backward, 12); memcpy(c.right, &right, 12); memcpy(c.up, &up, 12);
    memcpy(c.horizontal, &horizontal, 12); memcpy(c.vertical, &vertical, 12);
    c.lensRadius = aperture * 0.5f; c.focusDistance = focus; c.iteration = 1.0f; c.shouldRenderAABB = 0;
    memcpy(out_camera, &c, sizeof c);
}

/* glibc rand() (TYPE_3 additive feedback, r[i] = r[i-3] + r[i-31]) seeded with
 * 1, restated so the frame-seed schedule of vk_engine.cpp:1412 does not depend
 * on the host libc.  Fills out[0..n) with the first n outputs. */
void orc_glibc_rand(uint32_t seed, uint32_t n, int32_t *out) {
    std::vector<int32_t> r(34 + 310 + n);
    r[0] = (int32_t)seed;
    for (int i = 1; i < 31; i++) {
        int64_t hi = r[i - 1] / 127773, lo = r[i - 1] % 127773;
        int64_t word = 16807 * lo - 2836 * hi;
        if (word < 0) word += 2147483647;
        r[i] = (int32_t)word;
    }
    for (int i = 31; i < 34; i++) r[i] = r[i - 31];
    for (size_t i = 34; i < r.size(); i++) r[i] = (int32_t)((uint32_t)r[i - 31] + (uint32_t)r[i - 3]);
    for (uint32_t i = 0; i < n; i++) out[i] = (int32_t)(((uint32_t)r[344 + i]) >> 1);
}

int orc_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

}  // extern "C"
