// scenes.cpp -- the built-in scenes behind crt_scene_builtin().
//
// "book1", "quads", "corne", "cirno" restate the four world generators of
// engine/src/vk_engine.cpp:825-926.  "monkey", "voxel_city" and "grid:<n>"
// ("grid10m" = "grid:100") are the benchmark scenes of BASELINE.json configs
// 2, 4 and 5.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <stdexcept>
#include <string>
#include <vector>

#include "host_internal.h"

namespace crt {

bool load_mesh_file(const std::string &path, std::vector<MeshVertex> &verts, std::string &err);        // assets.cpp
bool load_texture_file(const std::string &path, std::vector<uint8_t> &rgba, uint32_t &w, uint32_t &h, std::string &err);

static inline uint32_t mix32(uint32_t x) {   // lowbias32 integer hash
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

// Deterministic stand-in textures for assets that are not on disk.
static void procedural_texture(const std::string &name, std::vector<uint8_t> &rgba, uint32_t &w, uint32_t &h) {
    if (name == "atlas16") {         // 16x16 tiles of 16x16 texels: flat tile colour + darker mortar lines + per-texel grain
        w = h = 256;
        rgba.resize((size_t)w * h * 4);
        for (uint32_t y = 0; y < h; y++) for (uint32_t x = 0; x < w; x++) {
            uint32_t tile = (y / 16) * 16 + (x / 16);
            uint32_t hc = mix32(tile * 2654435761u + 17u);
            uint32_t grain = mix32(x * 73856093u ^ y * 19349663u) & 31u;
            bool mortar = (x % 16 == 0) || (y % 16 == 0);
            uint8_t *p = &rgba[4 * ((size_t)y * w + x)];
            for (int c = 0; c < 3; c++) {
                uint32_t base = 96u + ((hc >> (8 * c)) & 127u);
                uint32_t v = mortar ? base / 2 : base + grain;
                p[c] = (uint8_t)(v > 255u ? 255u : v);
            }
            p[3] = 255;
        }
        return;
    }
    // generic: blue/green blotches on a lat-long grid ("earth"-like), also used for "fumo_diffuse"
    w = 256; h = 128;
    rgba.resize((size_t)w * h * 4);
    for (uint32_t y = 0; y < h; y++) for (uint32_t x = 0; x < w; x++) {
        uint32_t cell = mix32((x / 16) * 9781u + (y / 16) * 6271u + (uint32_t)name.size());
        bool land = (cell & 3u) == 0u;
        uint32_t grain = mix32(x * 1664525u + y * 1013904223u) & 15u;
        uint8_t *p = &rgba[4 * ((size_t)y * w + x)];
        p[0] = (uint8_t)(land ? 60 + grain : 20 + grain);
        p[1] = (uint8_t)(land ? 140 + grain : 60 + grain);
        p[2] = (uint8_t)(land ? 50 + grain : 170 + grain);
        p[3] = 255;
    }
}

// Resolve texture id order (Scene::textures) to pixel data: <asset_dir>/<file>.tx when present, else the stand-in.
static void bind_textures(SceneBuilder &b, const std::string &asset_dir, const std::vector<std::pair<std::string, std::string>> &files) {
    for (const std::string &name : b.texture_names) {
        std::vector<uint8_t> px; uint32_t w = 0, h = 0; bool ok = false;
        for (const auto &kv : files) {
            if (ok || kv.first != name || asset_dir.empty()) continue;
            std::string err;
            ok = load_texture_file(asset_dir + "/" + kv.second, px, w, h, err);
        }
        if (!ok) procedural_texture(name, px, w, h);
        b.texture_pixels.push_back(px); b.texture_w.push_back(w); b.texture_h.push_back(h);
    }
}

// vk_engine.cpp:47-50: rand(min, max) = min + (max - min) * vec3(rd, rd, rd), x then y then z.
static f3c rand3(GlibcRand &r, float mn, float mx) {
    float x = (float)r.next_double(), y = (float)r.next_double(), z = (float)r.next_double();
    return {mn + (mx - mn) * x, mn + (mx - mn) * y, mn + (mx - mn) * z};
}

// vk_engine.cpp:825-858.  The reference leaves the order of the random_double()
// calls inside one expression to the compiler; the build fixes it left to
// right.  (center - vec3(4, 0.2, 0)).length() is glm's component count (3), so
// the "keep clear of the big sphere" test is always true (sic) and all 196
// small spheres are emitted.
static void scene_book1(SceneBuilder &b, float aspect) {
    GlibcRand rng(1);
    for (int a = -7; a < 7; a++) {
        for (int bb = -7; bb < 7; bb++) {
            double chooseMaterial = rng.next_double();
            float cx = (float)(a + 0.9 * rng.next_double());
            float cz = (float)(bb + 0.9 * rng.next_double());
            f3c center = {cx, 0.2f, cz};
            if (chooseMaterial < 0.8) {
                f3c p = rand3(rng, 0.0f, 1.0f), q = rand3(rng, 0.0f, 1.0f);
                b.push_object(b.add_sphere(center, 0.2f, b.lambertian({p.x * q.x, p.y * q.y, p.z * q.z})));
            } else if (chooseMaterial < 0.95) {
                f3c albedo = rand3(rng, 0.5f, 1.0f);
                float fuzz = rand3(rng, 0.0f, 0.5f).x;
                b.push_object(b.add_sphere(center, 0.2f, b.metal(albedo, fuzz)));
            } else {
                b.push_object(b.add_sphere(center, 0.2f, b.dielectric(1.5f)));
            }
        }
    }
    std::vector<uint32_t> list;
    list.push_back(b.add_sphere({0, -2000, 0}, 2000.0f, b.lambertian({0.5f, 0.5f, 0.5f})));
    list.push_back(b.add_sphere({-4, 1, 0}, 1.0f, b.lambertian_tex("earth")));
    list.push_back(b.add_sphere({0, 1, 0}, 1.0f, b.dielectric(1.5f)));
    list.push_back(b.add_sphere({0, 1, 0}, -0.9f, b.dielectric(1.5f)));
    list.push_back(b.add_sphere({4, 1, 0}, 1.0f, b.metal({0.7f, 0.6f, 0.5f}, 0.0f)));
    b.push_list(list);
    b.set_camera({10, 1.5f, 2}, {0, 0, -0.25f}, 30.0f, aspect, 1.0f / 45.0f, 10.0f);
    b.set_background({-1, -1, -1});
}

// vk_engine.cpp:860-872
static void scene_quads(SceneBuilder &b, float aspect) {
    std::vector<uint32_t> list;
    list.push_back(b.add_quad({-3, -2, 5}, {0, 0, -4}, {0, 4, 0}, b.lambertian({1.0f, 0.2f, 0.2f})));
    list.push_back(b.add_quad({-2, -2, 0}, {4, 0, 0}, {0, 4, 0}, b.lambertian({0.2f, 1.0f, 0.2f})));
    list.push_back(b.add_quad({3, -2, 1}, {0, 0, 4}, {0, 4, 0}, b.lambertian({0.2f, 0.2f, 1.0f})));
    list.push_back(b.add_quad({-2, 3, 1}, {4, 0, 0}, {0, 0, 4}, b.lambertian({1.0f, 0.5f, 0.0f})));
    list.push_back(b.add_quad({-2, -3, 5}, {4, 0, 0}, {0, 0, -4}, b.lambertian({0.2f, 0.8f, 0.8f})));
    b.push_list(list);
    b.set_camera({0, 0, 9}, {0, 0, 0}, 80.0f, aspect, 0.0f, 10.0f);
    b.set_background({-1, -1, -1});
}

// vk_engine.cpp:874-888
static void scene_corne(SceneBuilder &b, float aspect) {
    b.push_object(b.add_quad({2, 0, 0}, {0, 2, 0}, {0, 0, 2}, b.lambertian({0.12f, 0.45f, 0.15f})));
    b.push_object(b.add_quad({0, 0, 0}, {0, 2, 0}, {0, 0, 2}, b.lambertian({0.65f, 0.05f, 0.05f})));
    b.push_object(b.add_quad({1.234f, 1.993f, 1.194f}, {-0.468f, 0, 0}, {0, 0, -0.378f}, b.diffuse_light({15, 15, 15})));
    b.push_object(b.add_quad({0, 0, 0}, {2, 0, 0}, {0, 0, 2}, b.lambertian({0.73f, 0.73f, 0.73f})));
    b.push_object(b.add_quad({2, 2, 2}, {-2, 0, 0}, {0, 0, -2}, b.lambertian({0.73f, 0.73f, 0.73f})));
    b.push_object(b.add_quad({0, 0, 2}, {2, 0, 0}, {0, 2, 0}, b.lambertian({0.73f, 0.73f, 0.73f})));
    b.push_box({0.468f, 0, 0.234f}, {1.063f, 0.595f, 0.829f}, b.lambertian({0.73f, 0.73f, 0.73f}));
    b.push_box({0.955f, 0, 1.063f}, {1.550f, 1.189f, 1.658f}, b.lambertian({0.73f, 0.73f, 0.73f}));
    b.set_camera({1, 1, -2.878f}, {1, 1, 0}, 40.0f, aspect, 0.0f, 10.0f);
    b.set_background({0, 0, 0});
}

static bool find_mesh(const std::string &asset_dir, const char *stem, std::vector<MeshVertex> &verts, std::string &err) {
    const char *exts[] = {".mesh", ".obj"};
    for (const char *ext : exts) {
        std::string path = asset_dir + "/" + stem + ext;
        FILE *f = fopen(path.c_str(), "rb");
        if (!f) continue;
        fclose(f);
        return load_mesh_file(path, verts, err);
    }
    err = std::string("no ") + stem + ".mesh or " + stem + ".obj under '" + asset_dir + "'";
    return false;
}

// vk_engine.cpp:890-926
static bool scene_cirno(SceneBuilder &b, const std::string &asset_dir, float aspect, std::string &err) {
    std::vector<MeshVertex> verts;
    if (!find_mesh(asset_dir, "cirno_low", verts, err)) return false;
    uint32_t mat = b.lambertian_tex("fumo_diffuse");
    for (size_t i = 0; i + 2 < verts.size(); i += 3) {
        const MeshVertex &a = verts[i], &c = verts[i + 1], &d = verts[i + 2];
        b.push_object(b.add_tri({a.pos[0], a.pos[1] - 0.08f, a.pos[2]}, {c.pos[0], c.pos[1] - 0.08f, c.pos[2]},
                                {d.pos[0], d.pos[1] - 0.08f, d.pos[2]}, {a.uv[0], c.uv[0], d.uv[0]}, {a.uv[1], c.uv[1], d.uv[1]}, mat));
    }
    std::vector<uint32_t> list;
    list.push_back(b.add_sphere({0, -2000, 0}, 2000.0f, b.lambertian({0.5f, 0.5f, 0.5f})));
    list.push_back(b.add_sphere({-4, 2, 0}, 2.0f, b.metal({0.7f, 0.6f, 0.5f}, 0.05f)));
    list.push_back(b.add_sphere({4, 2, 0}, 2.0f, b.metal({0.7f, 0.6f, 0.5f}, 0.05f)));
    b.push_list(list);
    b.set_camera({0, 2, 5}, {0, 1, 0}, 80.0f, aspect, 0.0f, 10.0f);
    b.set_background({-1, -1, -1});
    return true;
}

// BASELINE config 2: monkey_smooth as Tris with one Lambertian 0.73 + ground sphere, default sky.
static bool scene_monkey(SceneBuilder &b, const std::string &asset_dir, float aspect, std::string &err) {
    std::vector<MeshVertex> verts;
    if (!find_mesh(asset_dir, "monkey_smooth", verts, err)) return false;
    uint32_t mat = b.lambertian({0.73f, 0.73f, 0.73f});
    for (size_t i = 0; i + 2 < verts.size(); i += 3) {
        const MeshVertex &a = verts[i], &c = verts[i + 1], &d = verts[i + 2];
        b.push_object(b.add_tri({a.pos[0], a.pos[1], a.pos[2]}, {c.pos[0], c.pos[1], c.pos[2]}, {d.pos[0], d.pos[1], d.pos[2]},
                                {a.uv[0], c.uv[0], d.uv[0]}, {a.uv[1], c.uv[1], d.uv[1]}, mat));
    }
    b.push_list({b.add_sphere({0, -2001, 0}, 2000.0f, b.lambertian({0.5f, 0.5f, 0.5f}))});
    b.set_camera({0, 0.5f, 4}, {0, 0, 0}, 40.0f, aspect, 0.0f, 10.0f);
    b.set_background({-1, -1, -1});
    return true;
}

// One unit face of the voxel city as two CCW (seen from outside) triangles with
// UVs into tile `tile` of the 16x16 atlas.
static void city_face(SceneBuilder &b, f3c o, f3c du, f3c dv, uint32_t tile, uint32_t mat) {
    float tu = (float)(tile % 16) / 16.0f, tv = (float)(tile / 16) / 16.0f, s = 1.0f / 16.0f;
    f3c p00 = o, p10 = {o.x + du.x, o.y + du.y, o.z + du.z}, p01 = {o.x + dv.x, o.y + dv.y, o.z + dv.z};
    f3c p11 = {p10.x + dv.x, p10.y + dv.y, p10.z + dv.z};
    b.push_object(b.add_tri(p00, p10, p11, {tu, tu + s, tu + s}, {tv, tv, tv + s}, mat));
    b.push_object(b.add_tri(p00, p11, p01, {tu, tu + s, tu}, {tv, tv + s, tv + s}, mat));
}

static int city_height(int x, int z, uint32_t seed) {
    if (x < 0 || z < 0 || x >= 192 || z >= 192) return 0;
    if (x % 12 < 2 || z % 12 < 2) return 0;                      // streets
    uint32_t lot = mix32((uint32_t)(x / 12) * 7919u + (uint32_t)(z / 12) * 104729u + seed);
    int bx = (x % 12 - 2) / (1 + (lot & 3)), bz = (z % 12 - 2) / (1 + ((lot >> 2) & 3));   // building footprints inside a lot
    uint32_t hb = mix32(lot ^ (uint32_t)(bx * 31 + bz * 131 + 7));
    int h = 1 + (int)(hb % 9u);
    if ((hb >> 8) % 11u == 0u) h += 8;                            // a few towers
    return h;
}

// BASELINE config 4 stand-in (lost_empire.obj is absent from the reference
// checkout, .MISSING_LARGE_BLOBS): axis-aligned unit cube faces, exactly
// 224,998 triangles like the original, one textured Lambertian, default sky.
static void scene_voxel_city(SceneBuilder &b, uint32_t seed, float aspect) {
    const size_t target_faces = 224998 / 2;
    uint32_t mat = b.lambertian_tex("atlas16");
    const int G = 192, C = G / 2;
    size_t faces = 0;
    // columns in rings of growing Chebyshev distance from the centre, so that stopping at the quota leaves a compact city
    for (int ring = 0; ring <= C && faces < target_faces; ring++) {
        for (int dz = -ring; dz <= ring && faces < target_faces; dz++) {
            for (int dx = -ring; dx <= ring && faces < target_faces; dx++) {
                if (std::max(std::abs(dx), std::abs(dz)) != ring) continue;
                int x = C + dx, z = C + dz;
                if (x < 0 || z < 0 || x >= G || z >= G) continue;
                int h = city_height(x, z, seed);
                uint32_t tile = mix32((uint32_t)(x * 92821 + z * 68917) + seed) & 255u;
                float fx = (float)(x - C), fz = (float)(z - C), fh = (float)h;
                // top (normal +y)
                city_face(b, {fx, fh, fz}, {0, 0, 1}, {1, 0, 0}, h == 0 ? 0u : tile, mat); faces++;
                const int nb[4][2] = {{1, 0}, {-1, 0}, {0, 1}, {0, -1}};
                for (int k = 0; k < 4 && faces < target_faces; k++) {
                    int hn = city_height(x + nb[k][0], z + nb[k][1], seed);
                    for (int y = hn; y < h && faces < target_faces; y++) {
                        float fy = (float)y;
                        uint32_t wt = (tile + 16u * (uint32_t)(y & 7)) & 255u;
                        if (k == 0) city_face(b, {fx + 1, fy, fz}, {0, 1, 0}, {0, 0, 1}, wt, mat);          // +x
                        else if (k == 1) city_face(b, {fx, fy, fz}, {0, 0, 1}, {0, 1, 0}, wt, mat);        // -x
                        else if (k == 2) city_face(b, {fx, fy, fz + 1}, {1, 0, 0}, {0, 1, 0}, wt, mat);    // +z
                        else city_face(b, {fx, fy, fz}, {0, 1, 0}, {1, 0, 0}, wt, mat);                    // -z
                        faces++;
                    }
                }
            }
        }
    }
    b.set_camera({-46.0f, 30.0f, -58.0f}, {4.0f, 2.0f, 6.0f}, 55.0f, aspect, 0.0f, 10.0f);
    b.set_background({-1, -1, -1});
}

// BASELINE config 4 with the real mesh, for a checkout that has it: <asset_dir>/lost_empire.mesh or .obj
// as Tris with one Lambertian textured by "lost_empire-RGBA" (<asset_dir>/lost_empire-RGBA.tx when present, else the
// atlas stand-in), default sky.  The reference never renders this mesh through the path tracer, so the camera is
// this harness's: from above a corner of the bounding box towards its centre.
static bool scene_lost_empire(SceneBuilder &b, const std::string &asset_dir, float aspect, std::string &err) {
    std::vector<MeshVertex> verts;
    if (!find_mesh(asset_dir, "lost_empire", verts, err)) return false;
    if (verts.size() < 3) { err = "lost_empire: empty mesh"; return false; }
    uint32_t mat = b.lambertian_tex("lost_empire-RGBA");
    float lo[3] = {verts[0].pos[0], verts[0].pos[1], verts[0].pos[2]}, hi[3] = {lo[0], lo[1], lo[2]};
    for (size_t i = 0; i + 2 < verts.size(); i += 3) {
        const MeshVertex &a = verts[i], &c = verts[i + 1], &d = verts[i + 2];
        b.push_object(b.add_tri({a.pos[0], a.pos[1], a.pos[2]}, {c.pos[0], c.pos[1], c.pos[2]}, {d.pos[0], d.pos[1], d.pos[2]},
                                {a.uv[0], c.uv[0], d.uv[0]}, {a.uv[1], c.uv[1], d.uv[1]}, mat));
        for (int k = 0; k < 3; k++) for (int ax = 0; ax < 3; ax++) {
            lo[ax] = std::fmin(lo[ax], verts[i + k].pos[ax]); hi[ax] = std::fmax(hi[ax], verts[i + k].pos[ax]);
        }
    }
    const f3c centre = {0.5f * (lo[0] + hi[0]), 0.5f * (lo[1] + hi[1]), 0.5f * (lo[2] + hi[2])};
    const float ex = hi[0] - lo[0], ey = hi[1] - lo[1], ez = hi[2] - lo[2];
    b.set_camera({centre.x - 0.55f * ex, centre.y + 0.45f * ey + 0.1f * (ex + ez), centre.z - 0.65f * ez}, centre, 55.0f, aspect, 0.0f, 10.0f);
    b.set_background({-1, -1, -1});
    return true;
}

// BASELINE config 5: n x n instances of one 1,000-triangle patch (20 x 25 cells,
// two CCW-from-above triangles per cell), flattened to world space, rest pose y = 0.
static void scene_grid(SceneBuilder &b, uint32_t n, float aspect) {
    uint32_t mat = b.lambertian({0.6f, 0.6f, 0.55f});
    b.tris.reserve((size_t)n * n * 1000);
    for (uint32_t iz = 0; iz < n; iz++) for (uint32_t ix = 0; ix < n; ix++) {
        float ox = (float)(ix * 20) - 10.0f * (float)n, oz = (float)(iz * 25) - 12.5f * (float)n;
        for (int cz = 0; cz < 25; cz++) for (int cx = 0; cx < 20; cx++) {
            float x0 = ox + (float)cx, z0 = oz + (float)cz, x1 = x0 + 1.0f, z1 = z0 + 1.0f;
            float u0 = (float)cx / 20.0f, u1 = (float)(cx + 1) / 20.0f, v0 = (float)cz / 25.0f, v1 = (float)(cz + 1) / 25.0f;
            b.add_tri({x0, 0, z0}, {x0, 0, z1}, {x1, 0, z1}, {u0, u0, u1}, {v0, v1, v1}, mat);
            b.add_tri({x0, 0, z0}, {x1, 0, z1}, {x1, 0, z0}, {u0, u1, u1}, {v0, v1, v0}, mat);
        }
    }
    // no world list: this scene is never handed to the reference's O(N^2) builder
    float ext = 10.0f * (float)n;
    b.set_camera({-0.9f * ext, 0.5f * ext + 6.0f, -1.1f * ext}, {0, 0, 0}, 50.0f, aspect, 0.0f, 10.0f);
    b.set_background({-1, -1, -1});
}

}  // namespace crt

using namespace crt;

extern "C" {

int crt_scene_builtin(const char *name, const char *asset_dir, uint32_t width, uint32_t height, crt_host_scene **out) {
    if (!name || !out || width == 0 || height == 0) { set_host_error("crt_scene_builtin: bad argument"); return CRT_ERR_INVALID; }
    crt_host_scene *s = new (std::nothrow) crt_host_scene();
    if (!s) { set_host_error("crt_scene_builtin: out of memory"); return CRT_ERR_IO; }
    try {
    SceneBuilder &b = s->b;
    // the window aspect overrides the scene's (vk_engine.cpp:731,1342)
    float aspect = (float)width / (float)height;
    std::string n(name), dir(asset_dir ? asset_dir : ""), err;
    bool ok = true;
    if (n == "book1") { scene_book1(b, aspect); bind_textures(b, dir, {{"earth", "earthmap.tx"}}); }
    else if (n == "quads") scene_quads(b, aspect);
    else if (n == "corne") scene_corne(b, aspect);
    else if (n == "cirno") { ok = scene_cirno(b, dir, aspect, err); bind_textures(b, dir, {{"fumo_diffuse", "cirno_low_u1_v1.tx"}, {"fumo_diffuse", "earthmap.tx"}}); }
    else if (n == "monkey") ok = scene_monkey(b, dir, aspect, err);
    else if (n == "voxel_city") { scene_voxel_city(b, 1, aspect); bind_textures(b, dir, {}); }
    else if (n == "lost_empire") { ok = scene_lost_empire(b, dir, aspect, err); bind_textures(b, dir, {{"lost_empire-RGBA", "lost_empire-RGBA.tx"}}); }
    else if (n == "grid10m") scene_grid(b, 100, aspect);
    else if (n.rfind("grid:", 0) == 0) {
        int k = atoi(n.c_str() + 5);
        if (k < 1 || k > 128) { ok = false; err = "grid:<n> needs 1 <= n <= 128"; } else scene_grid(b, (uint32_t)k, aspect);
    } else { ok = false; err = "unknown scene '" + n + "'"; }
    if (!ok) { set_host_error("crt_scene_builtin: " + err); delete s; return n.empty() ? CRT_ERR_INVALID : CRT_ERR_IO; }
    b.finish();
    *out = s;
    return CRT_OK;
    } catch (const std::exception &e) {      // bad_alloc from a scene or asset too large for the host: an error code, not an abort
        set_host_error(std::string("crt_scene_builtin: ") + e.what());
        delete s;
        return CRT_ERR_IO;
    }
}

const crt_scene_desc *crt_host_scene_desc(const crt_host_scene *s) { return &s->b.desc; }
const crt_scene_params *crt_host_scene_params(const crt_host_scene *s) { return &s->b.params; }

int crt_host_scene_world(const crt_host_scene *s, uint32_t *n_objects, const uint32_t **obj_offsets,
                         const uint32_t **obj_prims, const uint8_t **obj_is_list) {
    *n_objects = (uint32_t)s->b.obj_is_list.size();
    *obj_offsets = s->b.obj_offsets.data();
    *obj_prims = s->b.obj_prims.data();
    *obj_is_list = s->b.obj_is_list.data();
    return CRT_OK;
}

void crt_scene_free(crt_host_scene *s) { delete s; }

}  // extern "C"
