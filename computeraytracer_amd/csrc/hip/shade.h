// shade.h -- camera ray generation, materials, sky, texture fetch
// (R4, R10-R13 and T8 of SURVEY.md section 8(a)).
#pragma once
#include "intersect.h"

namespace crt {

// compute.comp:163-172
template <int M>
CRT_DEV void camera_get_ray(const crt_camera_data &cam, float u, float v, Rng<M> &rng, V3 &origin, V3 &dir) {
    float dx, dy; random_in_unit_disk(rng, dx, dy);
    float rx = cam.lensRadius * dx, ry = cam.lensRadius * dy;
    V3 offset = ld3(cam.right) * rx + ld3(cam.up) * ry;
    V3 pos = ld3(cam.position), H = ld3(cam.horizontal), Vv = ld3(cam.vertical);
    V3 llc = ((pos - H * 0.5f) - Vv * 0.5f) - cam.focusDistance * ld3(cam.backward);
    origin = pos + offset;
    dir = normalize3(((llc + u * H) + v * Vv) - origin);
}

// RGBA8 UNORM, bilinear, REPEAT, texel centres at +0.5 (vk_initializers.cpp:252-261); plain float weights.
CRT_DEV V3 unpack_rgb(uint32_t p) { return mk3((float)(p & 255u) / 255.0f, (float)((p >> 8) & 255u) / 255.0f, (float)((p >> 16) & 255u) / 255.0f); }
CRT_DEV V3 sample_texture(const uint4 *tex, uint32_t w, uint32_t h, float u, float v) {
    float x = u * (float)w - 0.5f, y = v * (float)h - 0.5f;
    float x0f = __builtin_floorf(x), y0f = __builtin_floorf(y);
    float fx = x - x0f, fy = y - y0f;
    int x0 = wrapi(x0f, (int)w), y0 = wrapi(y0f, (int)h);
    // One 16-byte load: the texture is resident as footprints (entry (x0, y0) = the four texels of its bilinear sample,
    // wrap included; crt_scene_upload builds them), because the address path of k_wf_shade is its busiest unit and these
    // are fully divergent accesses: two 8-byte row loads were 128 of its ~430 address cycles per 64 paths.
    const uint4 q = tex[(size_t)y0 * w + (size_t)x0];
    const V3 t00 = unpack_rgb(q.x), t10 = unpack_rgb(q.y), t01 = unpack_rgb(q.z), t11 = unpack_rgb(q.w);
    V3 top = t00 * (1.0f - fx) + t10 * fx;
    V3 bot = t01 * (1.0f - fx) + t11 * fx;
    return top * (1.0f - fy) + bot * fy;
}

// The triangle records of a wave's 64 hits, fetched by quads: in round r the four lanes of a quad read the four 16-byte
// quarters of the record that the quad's lane r wants, i.e. 64 contiguous bytes per quad and instruction, which the
// address path serves as one request where four separate 16-byte loads per lane are 64 requests per instruction (these
// loads were 45 % of k_wf_shade's address-path cycles, its busiest unit).  A 4 x 4 transpose inside the quad (two
// rounds of DPP quad_perm + selects) then hands every lane the four quarters of its own record.  Call with all 64 lanes
// active; `idx` must be a valid record for every lane (lanes without a triangle hit pass 0).
CRT_DEV float4 quad_xchg(float4 v, bool xor2) {
    float4 r;
    if (xor2) {
        r.x = u2f((uint32_t)__builtin_amdgcn_mov_dpp((int)f2u(v.x), 0x4E, 0xF, 0xF, true)); r.y = u2f((uint32_t)__builtin_amdgcn_mov_dpp((int)f2u(v.y), 0x4E, 0xF, 0xF, true));
        r.z = u2f((uint32_t)__builtin_amdgcn_mov_dpp((int)f2u(v.z), 0x4E, 0xF, 0xF, true)); r.w = u2f((uint32_t)__builtin_amdgcn_mov_dpp((int)f2u(v.w), 0x4E, 0xF, 0xF, true));      // quad_perm [2,3,0,1]
    } else {
        r.x = u2f((uint32_t)__builtin_amdgcn_mov_dpp((int)f2u(v.x), 0xB1, 0xF, 0xF, true)); r.y = u2f((uint32_t)__builtin_amdgcn_mov_dpp((int)f2u(v.y), 0xB1, 0xF, 0xF, true));
        r.z = u2f((uint32_t)__builtin_amdgcn_mov_dpp((int)f2u(v.z), 0xB1, 0xF, 0xF, true)); r.w = u2f((uint32_t)__builtin_amdgcn_mov_dpp((int)f2u(v.w), 0xB1, 0xF, 0xF, true));      // quad_perm [1,0,3,2]
    }
    return r;
}
CRT_DEV float4 sel4(bool p, float4 a, float4 b) { return make_float4(p ? a.x : b.x, p ? a.y : b.y, p ? a.z : b.z, p ? a.w : b.w); }
CRT_DEV TriRec tri_rec_quad_fetch(const float4 *tri_shade, uint32_t idx, uint32_t lane) {
    const uint32_t j = lane & 3u;
    float4 R[4];
    const uint32_t o0 = (uint32_t)__builtin_amdgcn_mov_dpp((int)idx, 0x00, 0xF, 0xF, true), o1 = (uint32_t)__builtin_amdgcn_mov_dpp((int)idx, 0x55, 0xF, 0xF, true);      // quad_perm [r,r,r,r]
    const uint32_t o2 = (uint32_t)__builtin_amdgcn_mov_dpp((int)idx, 0xAA, 0xF, 0xF, true), o3 = (uint32_t)__builtin_amdgcn_mov_dpp((int)idx, 0xFF, 0xF, 0xF, true);
    R[0] = tri_shade[4 * (size_t)o0 + j]; R[1] = tri_shade[4 * (size_t)o1 + j]; R[2] = tri_shade[4 * (size_t)o2 + j]; R[3] = tri_shade[4 * (size_t)o3 + j];
    // transpose: afterwards R[p] of lane j = quarter p of the record of lane j
    const bool b0 = (j & 1u) != 0u, b1 = (j & 2u) != 0u;
    const float4 x0 = quad_xchg(R[1], false), x1 = quad_xchg(R[0], false), x2 = quad_xchg(R[3], false), x3 = quad_xchg(R[2], false);
    const float4 T0 = sel4(b0, x0, R[0]), T1 = sel4(b0, R[1], x1), T2 = sel4(b0, x2, R[2]), T3 = sel4(b0, R[3], x3);
    const float4 y0 = quad_xchg(T2, true), y1 = quad_xchg(T3, true), y2 = quad_xchg(T0, true), y3 = quad_xchg(T1, true);
    TriRec out;
    out.a = sel4(b1, y0, T0); out.b = sel4(b1, y1, T1); out.c = sel4(b1, T2, y2); out.e = sel4(b1, T3, y3);
    return out;
}

CRT_DEV V3 reflect3(V3 I, V3 N) { return I - (2.0f * dot3(N, I)) * N; }           // GLSL reflect
CRT_DEV V3 refract3(V3 I, V3 N, float eta) {                                     // GLSL refract
    float ni = dot3(N, I);
    float k = 1.0f - eta * eta * (1.0f - ni * ni);
    if (k < 0.0f) return mk3(0.0f, 0.0f, 0.0f);
    return eta * I - (eta * ni + __builtin_sqrtf(k)) * N;
}

// compute.comp:176-191
template <int M>
CRT_DEV bool should_refract(V3 incoming, V3 normal, float ior, float ratio, Rng<M> &rng) {
    float cosTheta = __builtin_fminf(dot3(neg3(incoming), normal), 1.0f);
    float sinTheta = __builtin_sqrtf(1.0f - cosTheta * cosTheta);
    if (ratio * sinTheta > 0.999f) return false;
    float reflectance = (1.0f - ior) / (1.0f + ior);
    reflectance = reflectance * reflectance;
    float schlick = reflectance + (1.0f - reflectance) * dm_pow5(1.0f - cosTheta);
    if (schlick > hash_1(rng)) return false;
    return true;
}

// One bounce of ray_color's loop body after a hit (compute.comp:451-455): emit + scatter + throughput
// update.  Returns false when the path ends here; `result` then holds its radiance.
template <int M, bool EXT, bool PRE = false>
CRT_DEV bool shade_hit(const DeviceScene &sc, V3 &o, V3 &d, float t, uint32_t id, Rng<M> &rng,
                       V3 &color, V3 &emitted, V3 &result, const TriRec &pre = TriRec()) {
    HitRecord rec;
    resolve_hit<PRE>(sc, o, d, t, id, rec, pre);
    const float4 *mp = (const float4 *)(sc.materials + rec.materialIndex);
    float4 m0 = mp[0], m1 = mp[1], m2 = mp[2];
    V3 albedo = mk3(m0.x, m0.y, m0.z);
    float fuzz = m0.w;
    uint32_t type = f2u(m1.w), texIndex = f2u(m2.w);
    // emit(), :363-371
    V3 e = type == CRT_MAT_DIFFUSE_LIGHT ? albedo : mk3(0.0f, 0.0f, 0.0f);
    emitted = emitted + color * e;
    // scatter(), :373-419
    V3 attenuation = albedo;
    if (type == CRT_MAT_LAMBERTIAN) {
        V3 dir = rec.normal + random_in_unit_sphere(rng);
        if (texIndex != CRT_BAD_INDEX && sc.n_textures > 0) {
            resolve_uv<PRE>(sc, o, d, id, rec, pre);
            uint32_t tw, th;
            const uint4 *tex = material_texture<EXT>(sc, texIndex, tw, th);      // always textures[0] (sic, :387) unless CRT_EXT_TEXTURE_INDEX
            attenuation = sample_texture(tex, tw, th, rec.u, rec.v);
        }
        o = rec.position; d = normalize3(dir);
    } else if (type == CRT_MAT_METAL) {
        V3 refl = reflect3(d, rec.normal);
        refl = refl + fuzz * random_in_unit_sphere(rng);
        o = rec.position; d = normalize3(refl);
        if (!(dot3(d, rec.normal) > 0.0f)) { result = emitted; return false; }
    } else if (type == CRT_MAT_DIELECTRIC) {
        float ratio = rec.front ? 1.0f / fuzz : fuzz;
        V3 dir = should_refract(d, rec.normal, fuzz, ratio, rng) ? refract3(d, rec.normal, ratio) : reflect3(d, rec.normal);
        o = rec.position; d = normalize3(dir);
        attenuation = mk3(1.0f, 1.0f, 1.0f);
    } else {
        result = emitted;
        return false;
    }
    color = color * (attenuation + emitted);      // sic, :455
    return true;
}

// hit_aabb, compute.comp:214-225, with t = INFINITY (the traversal has its own fused slab test in intersect.h).
CRT_DEV bool ref_hit_aabb(V3 o, V3 inv, V3 mn, V3 mx) {
    V3 tMin = (mn - o) * inv, tMax = (mx - o) * inv;
    V3 t0 = mk3(__builtin_fminf(tMin.x, tMax.x), __builtin_fminf(tMin.y, tMax.y), __builtin_fminf(tMin.z, tMax.z));
    V3 t1 = mk3(__builtin_fmaxf(tMin.x, tMax.x), __builtin_fmaxf(tMin.y, tMax.y), __builtin_fmaxf(tMin.z, tMax.z));
    float tNearest = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(t0.x, t0.y), t0.z), kTNear);
    float tFurthest = __builtin_fminf(__builtin_fminf(__builtin_fminf(t1.x, t1.y), t1.z), __builtin_inff());
    return tNearest < tFurthest;
}

// "Render AABB Option" of ray_color (compute.comp:427-436, R15): true when the camera ray crosses the 0.005-wide
// outline of any box of the reference's BVH array (supplied by the caller, crt_debug_boxes_upload); the sample is white.
CRT_DEV bool aabb_outline(const DeviceScene &sc, V3 o, V3 d) {
    V3 inv = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    for (uint32_t i = 0; i < sc.n_dbg_boxes; i++) {
        float4 a = sc.dbg_boxes[4 * (size_t)i], b = sc.dbg_boxes[4 * (size_t)i + 1];
        V3 mn = mk3(a.x, a.y, a.z), mx = mk3(b.x, b.y, b.z);
        V3 mn2 = mk3(mn.x + 0.005f, mn.y + 0.005f, mn.z + 0.005f), mx2 = mk3(mx.x - 0.005f, mx.y - 0.005f, mx.z - 0.005f);
        if (ref_hit_aabb(o, inv, mn, mx) && !ref_hit_aabb(o, inv, mn2, mx2)) return true;
    }
    return false;
}

// compute.comp:443-449
CRT_DEV V3 shade_miss(const crt_scene_params &p, V3 d, V3 color) {
    if (p.backgroundColor[0] != -1.0f || p.backgroundColor[1] != -1.0f || p.backgroundColor[2] != -1.0f)
        return ld3(p.backgroundColor);
    float a = 0.5f * (d.y + 1.0f);
    V3 sky = mk3(1.0f, 1.0f, 1.0f) * (1.0f - a) + mk3(0.5f, 0.7f, 1.0f) * a;
    return color * sky;
}

// compute.comp:474: cum += int(iteration == 1) * -cum + pass
CRT_DEV V3 accumulate(V3 cum, V3 pass, float iteration) {
    float first = (float)(int)(iteration == 1.0f);
    return cum + ((first * neg3(cum)) + pass);
}

}  // namespace crt
