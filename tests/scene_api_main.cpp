// Expresses the reference's "quads" and "corne" worlds (vk_engine.cpp:860-888) and one mixed world through
// crt/scene_api.hpp and dumps the arrays; tests/test_scene_api.py compares them with crt_scene_builtin's.
#include <cstdio>
#include <cstring>

#include "crt/scene_api.hpp"

using namespace crt::scene;

static void put(const char *tag, const void *p, size_t bytes) {
    uint64_t n = bytes;
    fwrite(tag, 1, 8, stdout); fwrite(&n, 8, 1, stdout);
    if (bytes) fwrite(p, 1, bytes, stdout);
}

static void dump(Scene &scene, const Camera &camera) {
    const crt_scene_desc *d = scene.desc();
    put("spheres ", d->spheres, sizeof(crt_sphere) * d->n_spheres);
    put("quads   ", d->quads, sizeof(crt_quad) * d->n_quads);
    put("tris    ", d->tris, sizeof(crt_tri) * d->n_tris);
    put("material", d->materials, sizeof(crt_material) * d->n_materials);
    put("camera  ", &camera.props, sizeof camera.props);
    put("offsets ", scene.world_offsets.data(), 4 * scene.world_offsets.size());
    put("prims   ", scene.world_prims.data(), 4 * scene.world_prims.size());
    put("is_list ", scene.world_is_list.data(), scene.world_is_list.size());
    std::string names;
    for (const auto &t : scene.textures) names += t + "\n";
    put("texnames", names.data(), names.size());
}

int main(int argc, char **argv) {
    if (argc < 3) return 2;
    const std::string name = argv[1];
    const float aspectRatio = (float)atof(argv[2]);
    World world;
    Camera camera;
    Scene scene;
    if (name == "quads") {
        auto quads = HittableList<Quad>();
        quads.add(std::make_shared<Quad>(Quad({-3, -2, 5}, {0, 0, -4}, {0, 4, 0}, Lambertian({1.0f, 0.2f, 0.2f}))));
        quads.add(std::make_shared<Quad>(Quad({-2, -2, 0}, {4, 0, 0}, {0, 4, 0}, Lambertian({0.2f, 1.0f, 0.2f}))));
        quads.add(std::make_shared<Quad>(Quad({3, -2, 1}, {0, 0, 4}, {0, 4, 0}, Lambertian({0.2f, 0.2f, 1.0f}))));
        quads.add(std::make_shared<Quad>(Quad({-2, 3, 1}, {4, 0, 0}, {0, 0, 4}, Lambertian({1.0f, 0.5f, 0.0f}))));
        quads.add(std::make_shared<Quad>(Quad({-2, -3, 5}, {4, 0, 0}, {0, 0, -4}, Lambertian({0.2f, 0.8f, 0.8f}))));
        world.push_back(std::make_shared<HittableList<Quad>>(quads));
        camera = Camera({0, 0, 9}, {0, 0, 0}, 80.0f, aspectRatio, 0.0f);
    } else if (name == "corne") {
        auto white = Lambertian({0.73f, 0.73f, 0.73f});
        world.push_back(std::make_shared<Quad>(Quad({2, 0, 0}, {0, 2, 0}, {0, 0, 2}, Lambertian({0.12f, 0.45f, 0.15f}))));
        world.push_back(std::make_shared<Quad>(Quad({0, 0, 0}, {0, 2, 0}, {0, 0, 2}, Lambertian({0.65f, 0.05f, 0.05f}))));
        world.push_back(std::make_shared<Quad>(Quad({1.234f, 1.993f, 1.194f}, {-0.468f, 0, 0}, {0, 0, -0.378f}, DiffuseLight({15, 15, 15}))));
        world.push_back(std::make_shared<Quad>(Quad({0, 0, 0}, {2, 0, 0}, {0, 0, 2}, white)));
        world.push_back(std::make_shared<Quad>(Quad({2, 2, 2}, {-2, 0, 0}, {0, 0, -2}, white)));
        world.push_back(std::make_shared<Quad>(Quad({0, 0, 2}, {2, 0, 0}, {0, 2, 0}, white)));
        world.push_back(std::make_shared<Box>(Box({0.468f, 0, 0.234f}, {1.063f, 0.595f, 0.829f}, white)));
        world.push_back(std::make_shared<Box>(Box({0.955f, 0, 1.063f}, {1.550f, 1.189f, 1.658f}, white)));
        camera = Camera({1, 1, -2.878f}, {1, 1, 0}, 40.0f, aspectRatio, 0.0f);
    } else if (name == "mixed") {
        // the tail of "book1" (vk_engine.cpp:849-856) plus one textured triangle and a repeated material
        auto spheres = HittableList<Sphere>();
        spheres.add(std::make_shared<Sphere>(Sphere({0, -2000, 0}, 2000.0f, Lambertian({0.5f, 0.5f, 0.5f}))));
        spheres.add(std::make_shared<Sphere>(Sphere({-4, 1, 0}, 1.0f, Lambertian("earth"))));
        spheres.add(std::make_shared<Sphere>(Sphere({0, 1, 0}, 1.0f, Dielectric(1.5f))));
        spheres.add(std::make_shared<Sphere>(Sphere({0, 1, 0}, -0.9f, Dielectric(1.5f))));
        spheres.add(std::make_shared<Sphere>(Sphere({4, 1, 0}, 1.0f, Metal({0.7f, 0.6f, 0.5f}, 0.0f))));
        world.push_back(std::make_shared<Tri>(Tri({0, 0, 0}, {1, 0, 0}, {0, 1, 0}, {0, 1, 0}, {0, 0, 1}, Lambertian("fumo_diffuse"))));
        world.push_back(std::make_shared<HittableList<Sphere>>(spheres));
        world.push_back(std::make_shared<Tri>(Tri({0, 0, 1}, {1, 0, 1}, {0, 1, 1}, {0, 1, 0}, {0, 0, 1}, Lambertian("earth"))));
        camera = Camera({10, 1.5f, 2}, {0, 0, -0.25f}, 30.0f, aspectRatio);
        const uint8_t px[8] = {1, 2, 3, 255, 4, 5, 6, 255};
        scene.serialize(world);
        scene.set_texture("earth", px, 2, 1);
        scene.set_texture("fumo_diffuse", px, 1, 2);
        dump(scene, camera);
        return 0;
    } else if (name == "empty-list") {
        world.push_back(std::make_shared<HittableList<Quad>>());
        try { scene.serialize(world); } catch (const std::runtime_error &e) { puts(e.what()); return 0; }
        return 1;
    } else return 2;
    scene.serialize(world);
    dump(scene, camera);
    return 0;
}
