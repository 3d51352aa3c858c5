// host_internal.h -- shared between the host-side translation units.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "crt/abi.h"

namespace crt {

void set_host_error(const std::string &msg);

struct f3c { float x, y, z; };

// assets::Vertex_f32_PNCV, asset-lib/include/mesh_asset.h:5-10 (44 bytes)
struct MeshVertex { float pos[3]; float normal[3]; float color[3]; float uv[2]; };

// glibc rand() restated (see host.cpp).
struct GlibcRand {
    explicit GlibcRand(uint32_t seed);
    int32_t next();
    // random_double(), vk_engine.cpp:42-45
    double next_double() { return next() / (2147483647 + 1.0); }
private:
    uint32_t step();
    uint32_t ring[34];
    uint64_t pos;
};

// Host mirror of Scene + the world vector the reference's generators fill.
struct SceneBuilder {
    std::vector<crt_sphere> spheres;
    std::vector<crt_quad> quads;
    std::vector<crt_tri> tris;
    std::vector<crt_material> materials;
    std::vector<std::string> texture_names;
    std::vector<std::vector<uint8_t>> texture_pixels;
    std::vector<uint32_t> texture_w, texture_h;
    std::vector<crt_texture> textures;
    // world list (CSR over input ids)
    std::vector<uint32_t> obj_offsets{0};
    std::vector<uint32_t> obj_prims;
    std::vector<uint8_t> obj_is_list;
    crt_scene_params params{};
    crt_scene_desc desc{};

    uint32_t texture_id(const std::string &name);
    uint32_t register_material(uint32_t type, f3c albedo, float fuzz, const char *texture);
    uint32_t lambertian(f3c albedo);
    uint32_t lambertian_tex(const char *texture);
    uint32_t metal(f3c albedo, float fuzz);
    uint32_t dielectric(float ior);
    uint32_t diffuse_light(f3c albedo);

    uint32_t add_sphere(f3c c, float r, uint32_t mat);
    uint32_t add_quad(f3c corner, f3c u, f3c v, uint32_t mat);
    uint32_t add_tri(f3c v0, f3c v1, f3c v2, f3c u, f3c v, uint32_t mat);
    void push_object(uint32_t id);
    void push_list(const std::vector<uint32_t> &ids);
    void push_box(f3c a, f3c b, uint32_t mat);

    void set_camera(f3c pos, f3c at, float fov, float aspect, float aperture, float focus);
    void set_background(f3c c) { params.backgroundColor[0] = c.x; params.backgroundColor[1] = c.y; params.backgroundColor[2] = c.z; }
    void finish();
};

}  // namespace crt

struct crt_host_scene { crt::SceneBuilder b; };
