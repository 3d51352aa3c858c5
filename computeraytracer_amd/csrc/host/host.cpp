// host.cpp -- host side of the C ABI that needs no GPU: camera block, frame
// seeds, primitive constructors, the material registry, the built-in scenes
// and the OBJ loader.  Mirrors the reference's scene-definition surface
// (raytracing/include/primitives.h, rt_material.h, scene.h, camera.h and the
// world generators in engine/src/vk_engine.cpp:825-926); cited per function.
#include "crt/abi.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "host_internal.h"

namespace crt {

static thread_local std::string g_host_error;
void set_host_error(const std::string &msg) { g_host_error = msg; }

// ------------------------------------------------------------ small maths ---
struct f3 { float x, y, z; };
static inline f3 mk(float x, float y, float z) { return {x, y, z}; }
static inline f3 mk(const float *p) { return {p[0], p[1], p[2]}; }
static inline f3 operator+(f3 a, f3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
static inline f3 operator-(f3 a, f3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
static inline f3 operator*(float s, f3 a) { return {s * a.x, s * a.y, s * a.z}; }
static inline f3 operator/(f3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
static inline f3 operator-(f3 a) { return {-a.x, -a.y, -a.z}; }
static inline float dot3(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline f3 cross3(f3 a, f3 b) { return {a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y}; }
static inline f3 normalize3(f3 a) { return (1.0f / std::sqrt(dot3(a, a))) * a; }   // glm::normalize = v * inversesqrt(dot(v, v))
static inline void put(float *dst, f3 v) { dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; }

// ------------------------------------------------------------ glibc rand ----
// TYPE_3 additive-feedback generator of glibc's rand()/srand(): r[i] = r[i-3] + r[i-31],
// first 310 outputs discarded, result >> 1.  Restated so that the seed schedule
// (vk_engine.cpp:1412) and the book1 layout do not depend on the host libc.
GlibcRand::GlibcRand(uint32_t seed) {
    int32_t r[34];
    r[0] = (int32_t)(seed ? seed : 1);
    for (int i = 1; i < 31; i++) {
        int64_t hi = r[i - 1] / 127773, lo = r[i - 1] % 127773;
        int64_t word = 16807 * lo - 2836 * hi;
        if (word < 0) word += 2147483647;
        r[i] = (int32_t)word;
    }
    for (int i = 31; i < 34; i++) r[i] = r[i - 31];
    for (int i = 0; i < 34; i++) ring[i] = (uint32_t)r[i];
    pos = 34;
    for (int i = 0; i < 310; i++) step();
}
uint32_t GlibcRand::step() {
    // ring holds the last 34 words; word k lives at ring[k % 34]
    uint32_t v = ring[(pos - 31) % 34] + ring[(pos - 3) % 34];
    ring[pos % 34] = v;
    pos++;
    return v;
}
int32_t GlibcRand::next() { return (int32_t)(step() >> 1); }

}  // namespace crt

using namespace crt;

extern "C" {

const char *crt_host_last_error(void) { return g_host_error.c_str(); }

void crt_free(void *p) { free(p); }

// Camera::Camera + calculateProperties, raytracing/src/camera.cpp:3-12,40-62
void crt_camera_init(crt_camera_data *out, const float position[3], const float at[3], float fov_degrees,
                     float aspect_ratio, float aperture, float focus_distance) {
    memset(out, 0, sizeof *out);
    f3 pos = mk(position);
    f3 backward = normalize3(pos - mk(at));
    float theta = fov_degrees * 0.01745329251994329576923690768489f;   // glm::radians
    float viewportHeight = 2.0f * std::tan(theta / 2.0f);
    float viewportWidth = aspect_ratio * viewportHeight;
    f3 right = normalize3(cross3(mk(0, 1, 0), backward));
    f3 up = cross3(backward, right);
    put(out->position, pos);
    put(out->backward, backward);
    put(out->right, right);
    put(out->up, up);
    put(out->horizontal, (focus_distance * viewportWidth) * right);
    put(out->vertical, (focus_distance * viewportHeight) * up);
    out->lensRadius = aperture * 0.5f;
    out->focusDistance = focus_distance;
    out->shouldRenderAABB = 0;
    out->iteration = 1.0f;
    out->seed = 0.0f;
}

// vk_engine.cpp:1412: seed = static_cast<float>(rand()) / 1E3f
void crt_frame_seeds(uint32_t first, uint32_t count, float *seeds) {
    GlibcRand rng(1);
    for (uint32_t i = 0; i < first; i++) rng.next();
    for (uint32_t i = 0; i < count; i++) seeds[i] = static_cast<float>(rng.next()) / 1E3f;
}

// Quad::Quad, raytracing/include/primitives.h:73-80
void crt_quad_init(crt_quad *q, const float corner[3], const float u[3], const float v[3], uint32_t material) {
    memset(q, 0, sizeof *q);
    f3 c = mk(corner), uu = mk(u), vv = mk(v);
    f3 n = cross3(uu, vv);
    f3 normal = normalize3(n);
    put(q->corner, c); put(q->u, uu); put(q->v, vv);
    put(q->normal, normal);
    q->d = dot3(normal, c);
    put(q->w, n / dot3(n, n));
    q->materialIndex = material;
}

}  // extern "C"

// ------------------------------------------------------- scene building -----
namespace crt {

// Scene::register_material, raytracing/src/scene.cpp:9-29: textures get ids in
// first-seen order, materials are de-duplicated by value.
uint32_t SceneBuilder::texture_id(const std::string &name) {
    for (size_t i = 0; i < texture_names.size(); i++) if (texture_names[i] == name) return (uint32_t)i;
    texture_names.push_back(name);
    return (uint32_t)texture_names.size() - 1;
}
uint32_t SceneBuilder::register_material(uint32_t type, f3c albedo, float fuzz, const char *texture) {
    crt_material m; memset(&m, 0, sizeof m);
    m.albedo[0] = albedo.x; m.albedo[1] = albedo.y; m.albedo[2] = albedo.z;
    m.fuzziness = fuzz; m.type = type;
    m.textureIndex = (texture && *texture) ? texture_id(texture) : CRT_BAD_INDEX;
    for (size_t i = 0; i < materials.size(); i++) {
        const crt_material &o = materials[i];
        if (o.albedo[0] == m.albedo[0] && o.albedo[1] == m.albedo[1] && o.albedo[2] == m.albedo[2] &&
            o.fuzziness == m.fuzziness && o.type == m.type && o.textureIndex == m.textureIndex) return (uint32_t)i;
    }
    materials.push_back(m);
    return (uint32_t)materials.size() - 1;
}
uint32_t SceneBuilder::lambertian(f3c albedo) { return register_material(CRT_MAT_LAMBERTIAN, albedo, 0.0f, nullptr); }           // rt_material.h:66-75
uint32_t SceneBuilder::lambertian_tex(const char *tex) { return register_material(CRT_MAT_LAMBERTIAN, {0, 0, 0}, 0.0f, tex); }
uint32_t SceneBuilder::metal(f3c albedo, float fuzz) { return register_material(CRT_MAT_METAL, albedo, fuzz, nullptr); }           // :78-84
uint32_t SceneBuilder::dielectric(float ior) { return register_material(CRT_MAT_DIELECTRIC, {0, 0, 0}, ior, nullptr); }            // :87-92
uint32_t SceneBuilder::diffuse_light(f3c albedo) { return register_material(CRT_MAT_DIFFUSE_LIGHT, albedo, 0.0f, nullptr); }       // :95-99

uint32_t SceneBuilder::add_sphere(f3c c, float r, uint32_t mat) {     // primitives.h:37-39
    crt_sphere s; memset(&s, 0, sizeof s);
    s.center[0] = c.x; s.center[1] = c.y; s.center[2] = c.z; s.radius = r; s.materialIndex = mat;
    spheres.push_back(s);
    return CRT_MAKE_ID(CRT_ID_KIND_SPHERE, spheres.size() - 1);
}
uint32_t SceneBuilder::add_quad(f3c corner, f3c u, f3c v, uint32_t mat) {
    crt_quad q;
    float c[3] = {corner.x, corner.y, corner.z}, uu[3] = {u.x, u.y, u.z}, vv[3] = {v.x, v.y, v.z};
    crt_quad_init(&q, c, uu, vv, mat);
    quads.push_back(q);
    return CRT_MAKE_ID(CRT_ID_KIND_QUAD, quads.size() - 1);
}
uint32_t SceneBuilder::add_tri(f3c v0, f3c v1, f3c v2, f3c u, f3c v, uint32_t mat) {   // primitives.h:134-136
    crt_tri t; memset(&t, 0, sizeof t);
    t.v0[0] = v0.x; t.v0[1] = v0.y; t.v0[2] = v0.z;
    t.v1[0] = v1.x; t.v1[1] = v1.y; t.v1[2] = v1.z;
    t.v2[0] = v2.x; t.v2[1] = v2.y; t.v2[2] = v2.z;
    t.u[0] = u.x; t.u[1] = u.y; t.u[2] = u.z;
    t.v[0] = v.x; t.v[1] = v.y; t.v[2] = v.z;
    t.materialIndex = mat;
    tris.push_back(t);
    return CRT_MAKE_ID(CRT_ID_KIND_TRI, tris.size() - 1);
}
void SceneBuilder::push_object(uint32_t id) {              // world.push_back(std::make_shared<Prim>(...))
    obj_prims.push_back(id); obj_offsets.push_back((uint32_t)obj_prims.size()); obj_is_list.push_back(0);
}
void SceneBuilder::push_list(const std::vector<uint32_t> &ids) {   // world.push_back(std::make_shared<HittableList<T>>(list))
    for (uint32_t id : ids) obj_prims.push_back(id);
    obj_offsets.push_back((uint32_t)obj_prims.size()); obj_is_list.push_back(1);
}
// Box::Box, primitives.h:103-117 (a HittableList<Quad> of six quads)
void SceneBuilder::push_box(f3c a, f3c b, uint32_t mat) {
    f3c mn = {std::fmin(a.x, b.x), std::fmin(a.y, b.y), std::fmin(a.z, b.z)};
    f3c mx = {std::fmax(a.x, b.x), std::fmax(a.y, b.y), std::fmax(a.z, b.z)};
    f3c dx = {mx.x - mn.x, 0, 0}, dy = {0, mx.y - mn.y, 0}, dz = {0, 0, mx.z - mn.z};
    // glm's unary minus negates every component: the zeros of -dx / -dz are negative zeros (primitives.h:112-115)
    f3c ndx = {-dx.x, -0.0f, -0.0f}, ndz = {-0.0f, -0.0f, -dz.z};
    std::vector<uint32_t> ids;
    ids.push_back(add_quad({mn.x, mn.y, mx.z}, dx, dy, mat));    // front
    ids.push_back(add_quad({mx.x, mn.y, mx.z}, ndz, dy, mat));   // right
    ids.push_back(add_quad({mx.x, mn.y, mn.z}, ndx, dy, mat));   // back
    ids.push_back(add_quad({mn.x, mn.y, mn.z}, dz, dy, mat));    // left
    ids.push_back(add_quad({mn.x, mx.y, mx.z}, dx, ndz, mat));   // top
    ids.push_back(add_quad({mn.x, mn.y, mn.z}, dx, dz, mat));    // bottom
    push_list(ids);
}

void SceneBuilder::set_camera(f3c pos, f3c at, float fov, float aspect, float aperture, float focus) {
    float p[3] = {pos.x, pos.y, pos.z}, a[3] = {at.x, at.y, at.z};
    crt_camera_init(&params.camera, p, a, fov, aspect, aperture, focus);
}

void SceneBuilder::finish() {
    for (size_t i = 0; i < texture_pixels.size(); i++) {
        crt_texture t; t.rgba8 = texture_pixels[i].data(); t.width = texture_w[i]; t.height = texture_h[i];
        textures.push_back(t);
    }
    desc.spheres = spheres.data(); desc.n_spheres = (uint32_t)spheres.size();
    desc.quads = quads.data(); desc.n_quads = (uint32_t)quads.size();
    desc.tris = tris.data(); desc.n_tris = (uint32_t)tris.size();
    desc.materials = materials.data(); desc.n_materials = (uint32_t)materials.size();
    desc.textures = textures.data(); desc.n_textures = (uint32_t)textures.size();
}

}  // namespace crt
