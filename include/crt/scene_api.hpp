// crt/scene_api.hpp -- the reference's scene-definition vocabulary over the C ABI.
//
// Header-only C++17.  A world generator written against the reference's classes
// (vk_engine.cpp:825-926) reads the same here:
//
//     using namespace crt::scene;
//     World world;
//     world.push_back(std::make_shared<Quad>(Quad({2,0,0}, {0,2,0}, {0,0,2}, Lambertian({.12f,.45f,.15f}))));
//     world.push_back(std::make_shared<Box>(Box({.468f,0,.234f}, {1.063f,.595f,.829f}, Lambertian({.73f,.73f,.73f}))));
//     Scene scene; scene.serialize(world);
//     crt_scene_upload(ctx, scene.desc());
//
// What differs from the reference, on purpose: there is no host BVHNode.  The reference serialises the world by
// walking its CPU-built BVH (BVHNode::gpu_serialize, bounding_volume_hierarchy.h:60-100), which fixes primitive and
// material numbering to that tree's DFS order; here serialize() walks the world vector in the order it was filled and
// the tree is built on the GPU by crt_bvh_build.  Primitive IDs are positions in the per-type arrays
// (crt/types.h "Primitive IDs"), images do not depend on the numbering.  The world list itself is kept as CSR
// (world_offsets / world_prims / world_is_list) for callers that want the reference's own tree from the oracle.
//
// Needs libcrt.so only for crt_quad_init / crt_camera_init (host helpers, no GPU).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>

#include "crt/abi.h"

namespace crt { namespace scene {

// Just enough of glm::vec3 for brace-initialised arguments and Box's arithmetic.
struct vec3 {
    float x = 0, y = 0, z = 0;
    vec3() = default;
    vec3(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {}
    explicit vec3(float s) : x(s), y(s), z(s) {}
    vec3 operator-() const { return {-x, -y, -z}; }                  // every component, zeros become -0 like glm's
    vec3 operator+(vec3 o) const { return {x + o.x, y + o.y, z + o.z}; }
    vec3 operator-(vec3 o) const { return {x - o.x, y - o.y, z - o.z}; }
    vec3 operator*(vec3 o) const { return {x * o.x, y * o.y, z * o.z}; }
    vec3 operator*(float s) const { return {x * s, y * s, z * s}; }
};

// ---- materials: raytracing/include/rt_material.h:13-99 ---------------------
struct RTMaterial {
    enum Type : uint32_t { none = 0, lambertian = CRT_MAT_LAMBERTIAN, metal = CRT_MAT_METAL,
                           dielectric = CRT_MAT_DIELECTRIC, diffuseLight = CRT_MAT_DIFFUSE_LIGHT };
    vec3 albedo;
    float fuzziness = 0.0f;
    Type type = none;
    std::string texture;           // "" = untextured (textureIndex 0xFFFFFFFF)
protected:
    RTMaterial(vec3 a, float f, Type t, const char *tex = nullptr) : albedo(a), fuzziness(f), type(t), texture(tex ? tex : "") {}
public:
    RTMaterial() = default;
};
struct Lambertian : RTMaterial {                                      // rt_material.h:66-75
    explicit Lambertian(vec3 albedo) : RTMaterial(albedo, 0.0f, lambertian) {}
    explicit Lambertian(const char *textureName) : RTMaterial(vec3(0.0f), 0.0f, lambertian, textureName) {}
};
struct Metal : RTMaterial {                                           // rt_material.h:78-84
    Metal(vec3 albedo, float fuzziness) : RTMaterial(albedo, fuzziness, metal) {}
};
struct Dielectric : RTMaterial {                                      // rt_material.h:87-92: the IOR travels in fuzziness
    explicit Dielectric(float refractionIndex) : RTMaterial(vec3(0.0f), refractionIndex, dielectric) {}
};
struct DiffuseLight : RTMaterial {                                    // rt_material.h:95-99
    explicit DiffuseLight(vec3 albedo) : RTMaterial(albedo, 0.0f, diffuseLight) {}
};

class Hittable;
using World = std::vector<std::shared_ptr<Hittable>>;

// ---- Scene: raytracing/include/scene.h + src/scene.cpp:9-29 -----------------
class Scene {
public:
    std::vector<crt_sphere> spheres;
    std::vector<crt_quad> quads;
    std::vector<crt_tri> tris;
    std::vector<crt_material> materials;
    std::vector<std::string> textures;          // names in id order; bind pixels with set_texture()
    std::vector<uint32_t> world_offsets{0}, world_prims;
    std::vector<uint8_t> world_is_list;

    // scene.cpp:9-29: textures get ids in first-seen order, equal materials share one entry.
    uint32_t register_material(const RTMaterial &m) {
        crt_material g; std::memset(&g, 0, sizeof g);
        g.albedo[0] = m.albedo.x; g.albedo[1] = m.albedo.y; g.albedo[2] = m.albedo.z;
        g.fuzziness = m.fuzziness; g.type = m.type; g.textureIndex = CRT_BAD_INDEX;
        if (!m.texture.empty()) {
            auto it = std::find(textures.begin(), textures.end(), m.texture);
            g.textureIndex = (uint32_t)(it - textures.begin());
            if (it == textures.end()) textures.push_back(m.texture);
        }
        for (size_t i = 0; i < materials.size(); i++) {
            const crt_material &o = materials[i];
            if (o.albedo[0] == g.albedo[0] && o.albedo[1] == g.albedo[1] && o.albedo[2] == g.albedo[2] &&
                o.fuzziness == g.fuzziness && o.type == g.type && o.textureIndex == g.textureIndex) return (uint32_t)i;
        }
        materials.push_back(g);
        return (uint32_t)materials.size() - 1;
    }

    // One world entry = one leaf object of the reference's BVH (a primitive, or a HittableList / Box kept together).
    inline void serialize(const World &world);

    // Pixels for a texture name some material used (RGBA8 rows, top-down); the scene keeps a copy.
    void set_texture(const std::string &name, const uint8_t *rgba8, uint32_t width, uint32_t height) {
        auto it = std::find(textures.begin(), textures.end(), name);
        if (it == textures.end()) throw std::runtime_error("ERROR: no material uses texture '" + name + "'");
        size_t i = (size_t)(it - textures.begin());
        if (pixels_.size() < textures.size()) { pixels_.resize(textures.size()); dims_.resize(textures.size()); }
        pixels_[i].assign(rgba8, rgba8 + (size_t)width * height * 4);
        dims_[i] = {width, height};
    }

    // The six arrays init_descriptors binds (vk_engine.cpp:1177-1229).  Valid until the scene changes.
    const crt_scene_desc *desc() {
        tex_.clear();
        for (size_t i = 0; i < textures.size(); i++) {
            if (i >= pixels_.size() || pixels_[i].empty())
                throw std::runtime_error("ERROR: texture '" + textures[i] + "' has no pixels (Scene::set_texture)");
            crt_texture t; t.rgba8 = pixels_[i].data(); t.width = dims_[i].first; t.height = dims_[i].second;
            tex_.push_back(t);
        }
        desc_.spheres = spheres.data(); desc_.n_spheres = (uint32_t)spheres.size();
        desc_.quads = quads.data(); desc_.n_quads = (uint32_t)quads.size();
        desc_.tris = tris.data(); desc_.n_tris = (uint32_t)tris.size();
        desc_.materials = materials.data(); desc_.n_materials = (uint32_t)materials.size();
        desc_.textures = tex_.data(); desc_.n_textures = (uint32_t)tex_.size();
        return &desc_;
    }

private:
    std::vector<std::vector<uint8_t>> pixels_;
    std::vector<std::pair<uint32_t, uint32_t>> dims_;
    std::vector<crt_texture> tex_;
    crt_scene_desc desc_{};
};

// ---- hittables: raytracing/include/hittable.h:11-59, primitives.h ------------
class Hittable {
public:
    enum Type : uint32_t { sphere = 1, quad = 2, tri = 4, bvhNode = 8 };   // hittable.h:13-18
    virtual ~Hittable() = default;
    [[nodiscard]] virtual Type type() const = 0;
    // Appends this object's primitives to the scene arrays and their IDs to `ids`.
    virtual void gpu_serialize(Scene &scene, std::vector<uint32_t> &ids) = 0;
    [[nodiscard]] virtual bool is_list() const { return false; }
};

template <typename T>
class HittableList : public Hittable {                                // hittable.h:27-59
    static_assert(std::is_base_of<Hittable, T>::value, "HittableList<T>: T must be a Hittable");
public:
    HittableList() = default;
    explicit HittableList(const std::shared_ptr<T> &object) { add(object); }
    void add(const std::shared_ptr<T> &object) { objects.push_back(object); }
    void gpu_serialize(Scene &scene, std::vector<uint32_t> &ids) override {
        for (const auto &object : objects) object->gpu_serialize(scene, ids);
    }
    [[nodiscard]] Type type() const override {
        if (objects.empty()) throw std::runtime_error("ERROR: Cannot serialize an empty HittableList!");
        return objects.front()->type();
    }
    [[nodiscard]] bool is_list() const override { return true; }
    std::vector<std::shared_ptr<Hittable>> objects;
};

class Primitive : public Hittable {                                   // primitives.h:15-24
public:
    explicit Primitive(RTMaterial m) : material(std::move(m)) {}
    RTMaterial material;
};

class Sphere : public Primitive {                                     // primitives.h:27-58
public:
    Sphere(vec3 center_, float radius_, const RTMaterial &m) : Primitive(m), center(center_), radius(radius_) {}
    [[nodiscard]] Type type() const override { return sphere; }
    void gpu_serialize(Scene &scene, std::vector<uint32_t> &ids) override {
        crt_sphere s; std::memset(&s, 0, sizeof s);
        s.center[0] = center.x; s.center[1] = center.y; s.center[2] = center.z; s.radius = radius;
        s.materialIndex = scene.register_material(material);
        scene.spheres.push_back(s);
        ids.push_back(CRT_MAKE_ID(CRT_ID_KIND_SPHERE, scene.spheres.size() - 1));
    }
    vec3 center; float radius;
};

class Quad : public Primitive {                                       // primitives.h:61-98
public:
    Quad(vec3 corner_, vec3 u_, vec3 v_, const RTMaterial &m) : Primitive(m), corner(corner_), u(u_), v(v_) {}
    [[nodiscard]] Type type() const override { return quad; }
    void gpu_serialize(Scene &scene, std::vector<uint32_t> &ids) override {
        crt_quad q;
        const float c[3] = {corner.x, corner.y, corner.z}, uu[3] = {u.x, u.y, u.z}, vv[3] = {v.x, v.y, v.z};
        crt_quad_init(&q, c, uu, vv, scene.register_material(material));   // normal, d, w: primitives.h:73-80
        scene.quads.push_back(q);
        ids.push_back(CRT_MAKE_ID(CRT_ID_KIND_QUAD, scene.quads.size() - 1));
    }
    vec3 corner, u, v;
};

struct Box : public HittableList<Quad> {                              // primitives.h:101-118
    Box(vec3 a, vec3 b, const RTMaterial &m) {
        vec3 mn = {std::fmin(a.x, b.x), std::fmin(a.y, b.y), std::fmin(a.z, b.z)};
        vec3 mx = {std::fmax(a.x, b.x), std::fmax(a.y, b.y), std::fmax(a.z, b.z)};
        vec3 dx = {mx.x - mn.x, 0, 0}, dy = {0, mx.y - mn.y, 0}, dz = {0, 0, mx.z - mn.z};
        add(std::make_shared<Quad>(Quad({mn.x, mn.y, mx.z}, dx, dy, m)));    // front
        add(std::make_shared<Quad>(Quad({mx.x, mn.y, mx.z}, -dz, dy, m)));   // right
        add(std::make_shared<Quad>(Quad({mx.x, mn.y, mn.z}, -dx, dy, m)));   // back
        add(std::make_shared<Quad>(Quad({mn.x, mn.y, mn.z}, dz, dy, m)));    // left
        add(std::make_shared<Quad>(Quad({mn.x, mx.y, mx.z}, dx, -dz, m)));   // top
        add(std::make_shared<Quad>(Quad({mn.x, mn.y, mn.z}, dx, dz, m)));    // bottom
    }
};

class Tri : public Primitive {                                        // primitives.h:121-160
public:
    // u = (u0, u1, u2), v = (v0, v1, v2): the texture coordinates of the three corners, component-wise.
    Tri(vec3 v0_, vec3 v1_, vec3 v2_, vec3 u_, vec3 v_, const RTMaterial &m)
        : Primitive(m), v0(v0_), v1(v1_), v2(v2_), u(u_), v(v_) {}
    [[nodiscard]] Type type() const override { return tri; }
    void gpu_serialize(Scene &scene, std::vector<uint32_t> &ids) override {
        crt_tri t; std::memset(&t, 0, sizeof t);
        t.v0[0] = v0.x; t.v0[1] = v0.y; t.v0[2] = v0.z;
        t.v1[0] = v1.x; t.v1[1] = v1.y; t.v1[2] = v1.z;
        t.v2[0] = v2.x; t.v2[1] = v2.y; t.v2[2] = v2.z;
        t.u[0] = u.x; t.u[1] = u.y; t.u[2] = u.z;
        t.v[0] = v.x; t.v[1] = v.y; t.v[2] = v.z;
        t.materialIndex = scene.register_material(material);
        scene.tris.push_back(t);
        ids.push_back(CRT_MAKE_ID(CRT_ID_KIND_TRI, scene.tris.size() - 1));
    }
    vec3 v0, v1, v2, u, v;
};

inline void Scene::serialize(const World &world) {
    std::vector<uint32_t> ids;
    for (const auto &object : world) {
        ids.clear();
        object->gpu_serialize(*this, ids);
        if (ids.empty()) throw std::runtime_error("ERROR: Cannot serialize an empty HittableList!");
        world_prims.insert(world_prims.end(), ids.begin(), ids.end());
        world_offsets.push_back((uint32_t)world_prims.size());
        world_is_list.push_back(object->is_list() ? 1 : 0);
    }
}

// ---- Camera: raytracing/include/camera.h:24-30, src/camera.cpp:3-12,40-62 ----
class Camera {
public:
    Camera() { std::memset(&props, 0, sizeof props); }
    Camera(vec3 position, vec3 at, float fovDegrees_ = 70.0f, float aspectRatio_ = 1.0f, float aperture = 1.0f / 45.0f,
           float focusDistance = 10.0f) : fovDegrees(fovDegrees_), aspectRatio(aspectRatio_) {
        const float p[3] = {position.x, position.y, position.z}, a[3] = {at.x, at.y, at.z};
        crt_camera_init(&props, p, a, fovDegrees_, aspectRatio_, aperture, focusDistance);
    }
    crt_camera_data props;
    float fovDegrees = 0.0f;
    float aspectRatio = 16.0f / 9.0f;
};

}}  // namespace crt::scene
