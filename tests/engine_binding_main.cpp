// engine_binding_main.cpp -- a Vulkan-free stand-in for the part of VulkanEngine that INTEGRATION.md patches
// (integration/vk_engine.patch): it holds the scene the way the engine does (per-type std::vector<std::any> buffers of
// GPU_t structs, a material map, GPUSceneData with the camera block) and then runs the binding code of INTEGRATION.md
// literally: flatten<U>() over the std::any buffers, reinterpret_cast of the engine's structs to the crt_* PODs,
// crt_configure / crt_scene_upload / crt_bvh_build once, and per frame the seed of vk_engine.cpp:1412 + crt_dispatch in
// place of the UBO write + vkCmdDispatch (vk_engine.cpp:1416-1435).  tests/test_gpu_parity.py runs it on the GPU and
// checks the image against the oracle.  The engine-side structs below restate the LAYOUTS of the reference's headers
// (primitives.h:29-34,62-70,123-131, rt_material.h:20-31, camera.h:9-22, scene.h:13-17) with a local vec3: glm is not
// available here, and the layouts are the data contract.
#include <any>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <map>
#include <string>
#include <unordered_map>
#include <vector>

#include "crt/scene_api.hpp"      // only to GENERATE the worlds (the reference's generators, vk_engine.cpp:860-888)

extern "C" {
#include "crt/abi.h"
}

// ---- what the engine has ---------------------------------------------------------------------------------------
namespace engine {
struct vec3 { float x, y, z; };
struct Hittable { enum class Type : unsigned { sphere = 1, quad = 2, tri = 4, bvhNode = 8 }; };
struct Sphere { struct GPU_t { vec3 center; float radius; vec3 pad; unsigned materialIndex; }; };
struct Quad { struct GPU_t { vec3 corner; float d; vec3 u; float p0; vec3 v; float p1; vec3 normal; float p2; vec3 w; float p3; vec3 p4; unsigned materialIndex; }; };
struct Tri { struct GPU_t { vec3 v0; float p0; vec3 v1; float p1; vec3 v2; float p2; vec3 u; float p3; vec3 v; float p4; vec3 p5; unsigned materialIndex; }; };
struct RTMaterial {
    struct GPU_t {
        vec3 albedo; float fuzziness; vec3 p0; unsigned type; vec3 p1; unsigned textureIndex;
        bool operator<(const GPU_t &o) const { return memcmp(this, &o, sizeof *this) < 0; }
    };
};
struct GPUCameraData {
    vec3 position; bool shouldRenderAABB; vec3 backward; float lensRadius; vec3 right; float focusDistance;
    vec3 up; float iteration; vec3 horizontal; float seed; vec3 vertical; float pad;
};
struct GPUSceneData { vec3 backgroundColor; float pad; GPUCameraData camera; };
struct Camera { GPUCameraData props{}; };
struct Scene {
    std::unordered_map<unsigned, std::vector<std::any>> buffers;
    std::map<RTMaterial::GPU_t, unsigned> materials;          // material -> id, as Scene::register_material fills it
    std::unordered_map<std::string, unsigned> textures;       // texture name -> id
    Camera camera;
    std::vector<std::any> &get_buffer(Hittable::Type t) { return buffers[(unsigned)t]; }
};
static_assert(sizeof(Sphere::GPU_t) == 32 && sizeof(Quad::GPU_t) == 96 && sizeof(Tri::GPU_t) == 96 && sizeof(RTMaterial::GPU_t) == 48, "GPU_t layouts");
static_assert(sizeof(GPUCameraData) == 96 && sizeof(GPUSceneData) == 112, "uniform layouts");
}  // namespace engine
using namespace engine;

struct CpuTexture { std::vector<uint8_t> pixels; uint32_t w, h; };

static void put(FILE *f, const char *tag, const void *p, size_t bytes) {
    uint64_t n = bytes;
    fwrite(tag, 1, 8, f); fwrite(&n, 8, 1, f);
    if (bytes) fwrite(p, 1, bytes, f);
}

// ---- INTEGRATION.md, "Binding sketch" ------------------------------------------------------------------------------
template <typename U> static std::vector<U> flatten(std::vector<std::any> &v) {   // what upload_buffer's loop does (vk_engine.cpp:673-680)
    std::vector<U> out(v.size());
    for (size_t i = 0; i < v.size(); i++) out[i] = v[i].has_value() ? std::any_cast<U>(v[i]) : U();
    return out;
}

int main(int argc, char **argv) {
    if (argc < 6) { fprintf(stderr, "usage: engine_binding_main <corne|mixed> <width> <height> <frames> <out file>\n"); return 2; }
    const std::string name = argv[1];
    const uint32_t width = (uint32_t)atoi(argv[2]), height = (uint32_t)atoi(argv[3]);
    const int frames = atoi(argv[4]);
    const float aspect = (float)width / (float)height;

    // the world, generated as the engine's init_scene does and serialised into the engine's per-type std::any buffers
    Scene currentScene;
    std::unordered_map<std::string, CpuTexture> cpuTextures;
    {
        namespace S = crt::scene;
        S::World world; S::Camera cam; S::Scene sc;
        if (name == "corne") {
            auto white = S::Lambertian({0.73f, 0.73f, 0.73f});
            world.push_back(std::make_shared<S::Quad>(S::Quad({2, 0, 0}, {0, 2, 0}, {0, 0, 2}, S::Lambertian({0.12f, 0.45f, 0.15f}))));
            world.push_back(std::make_shared<S::Quad>(S::Quad({0, 0, 0}, {0, 2, 0}, {0, 0, 2}, S::Lambertian({0.65f, 0.05f, 0.05f}))));
            world.push_back(std::make_shared<S::Quad>(S::Quad({1.234f, 1.993f, 1.194f}, {-0.468f, 0, 0}, {0, 0, -0.378f}, S::DiffuseLight({15, 15, 15}))));
            world.push_back(std::make_shared<S::Quad>(S::Quad({0, 0, 0}, {2, 0, 0}, {0, 0, 2}, white)));
            world.push_back(std::make_shared<S::Quad>(S::Quad({2, 2, 2}, {-2, 0, 0}, {0, 0, -2}, white)));
            world.push_back(std::make_shared<S::Quad>(S::Quad({0, 0, 2}, {2, 0, 0}, {0, 2, 0}, white)));
            world.push_back(std::make_shared<S::Box>(S::Box({0.468f, 0, 0.234f}, {1.063f, 0.595f, 0.829f}, white)));
            world.push_back(std::make_shared<S::Box>(S::Box({0.955f, 0, 1.063f}, {1.550f, 1.189f, 1.658f}, white)));
            cam = S::Camera({1, 1, -2.878f}, {1, 1, 0}, 40.0f, aspect, 0.0f);
        } else {
            auto spheres = S::HittableList<S::Sphere>();
            spheres.add(std::make_shared<S::Sphere>(S::Sphere({0, -2000, 0}, 2000.0f, S::Lambertian({0.5f, 0.5f, 0.5f}))));
            spheres.add(std::make_shared<S::Sphere>(S::Sphere({-4, 1, 0}, 1.0f, S::Lambertian("earth"))));
            spheres.add(std::make_shared<S::Sphere>(S::Sphere({0, 1, 0}, 1.0f, S::Dielectric(1.5f))));
            spheres.add(std::make_shared<S::Sphere>(S::Sphere({0, 1, 0}, -0.9f, S::Dielectric(1.5f))));
            spheres.add(std::make_shared<S::Sphere>(S::Sphere({4, 1, 0}, 1.0f, S::Metal({0.7f, 0.6f, 0.5f}, 0.0f))));
            world.push_back(std::make_shared<S::Tri>(S::Tri({2, 0, 2}, {2, 2, 2}, {2, 0, 0}, {0, 1, 0}, {0, 0, 1}, S::Lambertian("earth"))));
            world.push_back(std::make_shared<S::HittableList<S::Sphere>>(spheres));
            world.push_back(std::make_shared<S::Quad>(S::Quad({-2, 0.2f, 3}, {1.5f, 0, 0}, {0, 1.5f, 0}, S::DiffuseLight({4, 4, 4}))));
            cam = S::Camera({10, 1.5f, 2}, {0, 0, -0.25f}, 30.0f, aspect);
            CpuTexture t; t.w = 4; t.h = 2;
            for (uint32_t i = 0; i < t.w * t.h; i++) { t.pixels.push_back((uint8_t)(40 * i + 15)); t.pixels.push_back((uint8_t)(255 - 30 * i)); t.pixels.push_back((uint8_t)(90 + 7 * i)); t.pixels.push_back(255); }
            cpuTextures["earth"] = t;
        }
        sc.serialize(world);
        for (const auto &tn : sc.textures) { const CpuTexture &t = cpuTextures.at(tn); sc.set_texture(tn, t.pixels.data(), t.w, t.h); }
        const crt_scene_desc *d = sc.desc();
        for (uint32_t i = 0; i < d->n_spheres; i++) { Sphere::GPU_t g; memcpy(&g, d->spheres + i, sizeof g); currentScene.get_buffer(Hittable::Type::sphere).push_back(g); }
        for (uint32_t i = 0; i < d->n_quads; i++) { Quad::GPU_t g; memcpy(&g, d->quads + i, sizeof g); currentScene.get_buffer(Hittable::Type::quad).push_back(g); }
        for (uint32_t i = 0; i < d->n_tris; i++) { Tri::GPU_t g; memcpy(&g, d->tris + i, sizeof g); currentScene.get_buffer(Hittable::Type::tri).push_back(g); }
        for (uint32_t i = 0; i < d->n_materials; i++) { RTMaterial::GPU_t g; memcpy(&g, d->materials + i, sizeof g); currentScene.materials[g] = i; }
        for (uint32_t i = 0; i < sc.textures.size(); i++) currentScene.textures[sc.textures[i]] = i;
        memcpy(&currentScene.camera.props, &cam.props, sizeof cam.props);
    }
    struct { uint32_t width, height; } windowExtent = {width, height};
    GPUSceneData sceneParameters{};
    sceneParameters.backgroundColor = name == "corne" ? vec3{0, 0, 0} : vec3{-1, -1, -1};

    // ---- the patch: init() ------------------------------------------------------------------------------------------
    crt_context *crt = nullptr;
    if (crt_create(0, &crt) != CRT_OK) { std::cerr << "ERROR: no usable HIP device" << std::endl; return 3; }

    // ---- the patch: init_descriptors(), instead of create_buffer / upload_buffer for spheres / quads / tris / materials ---
    auto spheres = flatten<Sphere::GPU_t>(currentScene.get_buffer(Hittable::Type::sphere));
    auto quads = flatten<Quad::GPU_t>(currentScene.get_buffer(Hittable::Type::quad));
    auto tris = flatten<Tri::GPU_t>(currentScene.get_buffer(Hittable::Type::tri));
    std::vector<RTMaterial::GPU_t> materials(currentScene.materials.size());
    for (auto &[material, id] : currentScene.materials) materials[id] = material;
    std::vector<crt_texture> textures(currentScene.textures.size());
    for (auto &[textureName, id] : currentScene.textures) textures[id] = {cpuTextures[textureName].pixels.data(), cpuTextures[textureName].w, cpuTextures[textureName].h};

    crt_scene_desc desc = {};
    desc.spheres = reinterpret_cast<const crt_sphere *>(spheres.data());       desc.n_spheres = (uint32_t)spheres.size();
    desc.quads = reinterpret_cast<const crt_quad *>(quads.data());             desc.n_quads = (uint32_t)quads.size();
    desc.tris = reinterpret_cast<const crt_tri *>(tris.data());                desc.n_tris = (uint32_t)tris.size();
    desc.materials = reinterpret_cast<const crt_material *>(materials.data()); desc.n_materials = (uint32_t)materials.size();
    desc.textures = textures.data();                                           desc.n_textures = (uint32_t)textures.size();
    crt_config cfg = {windowExtent.width, windowExtent.height, CRT_RNG_REF, CRT_TRACER_WAVEFRONT, 0, 0, 0, 1};
    float build_ms = 0.0f;
    if (crt_configure(crt, &cfg) != CRT_OK || crt_scene_upload(crt, &desc) != CRT_OK ||
        crt_bvh_build(crt, CRT_BVH_SAH_REFINE | CRT_BVH_WIDE, &build_ms) != CRT_OK) {
        std::cerr << "ERROR: " << crt_last_error(crt) << std::endl;
        return 4;
    }
    // the engine's vectors may be freed right away (upload copies): prove it
    std::vector<Sphere::GPU_t>().swap(spheres); std::vector<Quad::GPU_t>().swap(quads); std::vector<Tri::GPU_t>().swap(tris);

    // ---- the patch: draw(), instead of vk_engine.cpp:1416-1435; run() around it (camera.calculateProperties: iteration++) ---
    std::vector<float> seeds;
    currentScene.camera.props.iteration = 1.0f;
    for (int frame = 0; frame < frames; frame++) {
        currentScene.camera.props.seed = static_cast<float>(rand()) / 1E3f;      // NOLINT, vk_engine.cpp:1412 (glibc rand(), never seeded)
        sceneParameters.camera = currentScene.camera.props;
        static_assert(sizeof(GPUSceneData) == sizeof(crt_scene_params), "GPUSceneData is the 112-byte UBO");
        if (crt_dispatch(crt, reinterpret_cast<const crt_scene_params *>(&sceneParameters)) != CRT_OK) {
            std::cerr << "ERROR: " << crt_last_error(crt) << std::endl;
            return 5;
        }
        seeds.push_back(currentScene.camera.props.seed);
        currentScene.camera.props.iteration += 1.0f;                             // camera.cpp:42-44 (unchanged camera)
    }
    // ---- the patch: present; waitForFences + the fullscreen pass's read of the image (compute.frag:15-16) -----------------
    std::vector<float> accum((size_t)width * height * 4);
    std::vector<uint32_t> ids((size_t)width * height);
    if (crt_sync(crt) != CRT_OK || crt_image_read(crt, accum.data(), accum.size()) != CRT_OK || crt_hit_ids_read(crt, ids.data(), ids.size()) != CRT_OK) {
        std::cerr << "ERROR: " << crt_last_error(crt) << std::endl;
        return 6;
    }
    std::vector<float> presented((size_t)width * height * 3);
    for (uint32_t y = 0; y < height; y++)
        for (uint32_t x = 0; x < width; x++) {
            const float *p = &accum[4 * ((size_t)(height - 1 - y) * width + x)];            // V flip
            for (int c = 0; c < 3; c++) presented[3 * ((size_t)y * width + x) + c] = p[3] > 0.0f ? p[c] / p[3] : 0.0f;
        }
    crt_destroy(crt);

    FILE *f = fopen(argv[5], "wb");
    if (!f) return 7;
    {   // the arrays the engine handed over, for the oracle
        auto s2 = flatten<Sphere::GPU_t>(currentScene.get_buffer(Hittable::Type::sphere));
        auto q2 = flatten<Quad::GPU_t>(currentScene.get_buffer(Hittable::Type::quad));
        auto t2 = flatten<Tri::GPU_t>(currentScene.get_buffer(Hittable::Type::tri));
        put(f, "sphere  ", s2.data(), s2.size() * sizeof(Sphere::GPU_t));
        put(f, "quad    ", q2.data(), q2.size() * sizeof(Quad::GPU_t));
        put(f, "tri     ", t2.data(), t2.size() * sizeof(Tri::GPU_t));
        put(f, "material", materials.data(), materials.size() * sizeof(RTMaterial::GPU_t));
        for (auto &t : textures) { uint32_t wh[2] = {t.width, t.height}; put(f, "texdims ", wh, 8); put(f, "texture ", t.rgba8, (size_t)t.width * t.height * 4); }
    }
    put(f, "params  ", &sceneParameters, sizeof sceneParameters);
    put(f, "seeds   ", seeds.data(), seeds.size() * 4);
    put(f, "accum   ", accum.data(), accum.size() * 4);
    put(f, "ids     ", ids.data(), ids.size() * 4);
    put(f, "present ", presented.data(), presented.size() * 4);
    fclose(f);
    printf("engine_binding_main: %s %ux%u x %d frames, BVH build %.3f ms\n", name.c_str(), width, height, frames, build_ms);
    return 0;
}
