// device_math.h -- FP32 building blocks of the HIP kernels.
//
// Arithmetic contract (DESIGN.md "Arithmetic"): everything that can change a
// pixel is a plain IEEE-754 binary32 operation in a fixed order, no fused
// multiply-add (this translation unit is built with -ffp-contract=off;
// hipcc's default divide and sqrt are correctly rounded).  The only fused
// operations are in the BVH slab test, which can only cull, never change a hit.
// GLSL leaves sin/cos/pow/acos/atan precision to the implementation, so the
// build fixes them as the polynomial sequences below.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace crt {

struct V3 { float x, y, z; };

#define CRT_DEV __device__ __forceinline__

CRT_DEV V3 mk3(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
CRT_DEV V3 ld3(const float *p) { return mk3(p[0], p[1], p[2]); }
CRT_DEV V3 operator+(V3 a, V3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
CRT_DEV V3 operator-(V3 a, V3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
CRT_DEV V3 operator*(V3 a, V3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
CRT_DEV V3 operator*(float s, V3 a) { return mk3(s * a.x, s * a.y, s * a.z); }
CRT_DEV V3 operator*(V3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
CRT_DEV V3 operator/(V3 a, float s) { return mk3(a.x / s, a.y / s, a.z / s); }
CRT_DEV V3 neg3(V3 a) { return mk3(-a.x, -a.y, -a.z); }
// GLSL dot(): (x*x' + y*y') + z*z'
CRT_DEV float dot3(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
CRT_DEV V3 cross3(V3 a, V3 b) { return mk3(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y); }
// normalize(v) = v * (1 / sqrt(dot(v, v)))
CRT_DEV V3 normalize3(V3 a) { float inv = 1.0f / __builtin_sqrtf(dot3(a, a)); return a * inv; }

CRT_DEV uint32_t f2u(float f) { return __float_as_uint(f); }
CRT_DEV float u2f(uint32_t u) { return __uint_as_float(u); }

constexpr float kPi = 3.14159265359f;        // compute.comp:3
constexpr float kTwoPi = 2.0f * kPi;
constexpr float kHalfPi = 1.5707963267948966f;
constexpr float kTNear = 0.001f;             // compute.comp:205
constexpr float kInf = 3.402823466e+38f;     // compute.comp:4

// sin/cos on [0, 2*pi]: nearest-quadrant reduction (3-term Cody-Waite) + Cephes sinf/cosf kernels.
CRT_DEV void dm_sincos(float th, float &s, float &c) {
    float qf = __builtin_floorf(th * 0.636619772f + 0.5f);
    int q = (int)qf;
    float r = th - qf * 1.5703125f;
    r = r - qf * 4.837512969970703125e-4f;
    r = r - qf * 7.54978995489188216e-8f;
    float z = r * r;
    float sp = ((-1.9515295891e-4f * z + 8.3321608736e-3f) * z - 1.6666654611e-1f) * z * r + r;
    float cp = ((2.443315711809948e-5f * z - 1.388731625493765e-3f) * z + 4.166664568298827e-2f) * z * z;
    cp = cp - 0.5f * z;
    cp = cp + 1.0f;
    int k = q & 3;
    float ss = (k & 1) ? cp : sp;
    float cc = (k & 1) ? sp : cp;
    s = (k & 2) ? -ss : ss;
    c = (k == 1 || k == 2) ? -cc : cc;
}

// pow(x, 0.3333333334), x in [0, 1]: exponent-divide seed + two Halley steps.
CRT_DEV float dm_cbrt01(float x) {
    if (x == 0.0f) return 0.0f;
    float t = u2f(f2u(x) / 3u + 709958130u);
#pragma unroll
    for (int k = 0; k < 2; k++) {
        float t3 = t * t * t;
        t = t * ((t3 + (x + x)) / ((t3 + t3) + x));
    }
    return t;
}

CRT_DEV float dm_pow5(float x) { float x2 = x * x; float x4 = x2 * x2; return x4 * x; }

CRT_DEV float dm_asin_core(float x) {
    float z = x * x;
    return ((((4.2163199048e-2f * z + 2.4181311049e-2f) * z + 4.5470025998e-2f) * z + 7.4953002686e-2f) * z
            + 1.6666752422e-1f) * z * x + x;
}

CRT_DEV float dm_acos(float x) {
    if (x < -0.5f) return kPi - 2.0f * dm_asin_core(__builtin_sqrtf((1.0f + x) * 0.5f));
    if (x > 0.5f) return 2.0f * dm_asin_core(__builtin_sqrtf((1.0f - x) * 0.5f));
    return kHalfPi - dm_asin_core(x);
}

CRT_DEV float dm_atan_pos(float x) {
    float y;
    if (x > 2.414213562373095f) { y = kHalfPi; x = -(1.0f / x); }
    else if (x > 0.4142135623730950f) { y = 0.7853981633974483f; x = (x - 1.0f) / (x + 1.0f); }
    else y = 0.0f;
    float z = x * x;
    y = y + ((((8.05374449538e-2f * z - 1.38776856032e-1f) * z + 1.99777106478e-1f) * z - 3.33329491539e-1f) * z * x + x);
    return y;
}

CRT_DEV float dm_atan2(float y, float x) {
    if (x == 0.0f) {
        if (y > 0.0f) return kHalfPi;
        if (y < 0.0f) return -kHalfPi;
        return 0.0f;
    }
    float a = dm_atan_pos(__builtin_fabsf(y / x));
    if (x < 0.0f) a = kPi - a;
    return (y < 0.0f) ? -a : a;
}

// ------------------------------------------------------------------ RNG -----
// CRT_RNG_REF: the shader's float-seeded hash (compute.comp:116-139).
// CRT_RNG_PCG: pcg32, stream selected by the iteration, state keyed by (seed bits, pixel).
template <int MODE> struct Rng;

template <> struct Rng<0> {
    float seed;
    CRT_DEV void init(float cam_seed, uint32_t x, uint32_t y, uint32_t, float) {
        seed = cam_seed * ((float)x * (float)y + (float)y * (float)x);   // seed * dot(xy, yx)
    }
    CRT_DEV uint32_t hash_n() {
        seed = seed + 0.1f; uint32_t sx = f2u(seed);
        seed = seed + 0.1f; uint32_t sy = f2u(seed);
        uint32_t nx = 1103515245u * ((sx >> 1) ^ sy);
        uint32_t ny = 1103515245u * ((sy >> 1) ^ sx);
        uint32_t h32 = 1103515245u * (nx ^ (ny >> 3));
        return h32 ^ (h32 >> 16);
    }
    CRT_DEV void draw1(uint32_t &a) { a = hash_n(); }
    CRT_DEV void draw2(uint32_t &a, uint32_t &b) { uint32_t n = hash_n(); a = n; b = n * 48271u; }
    CRT_DEV void draw3(uint32_t &a, uint32_t &b, uint32_t &c) { uint32_t n = hash_n(); a = n; b = n * 16807u; c = n * 48271u; }
    CRT_DEV uint64_t pack() const { return (uint64_t)f2u(seed); }
    CRT_DEV void unpack(uint64_t v, uint64_t) { seed = u2f((uint32_t)v); }
};

template <> struct Rng<1> {
    uint64_t state, inc;
    CRT_DEV uint32_t next() {
        uint64_t old = state;
        state = old * 6364136223846793005ull + inc;
        uint32_t xorshifted = (uint32_t)(((old >> 18) ^ old) >> 27);
        uint32_t rot = (uint32_t)(old >> 59);
        return (xorshifted >> rot) | (xorshifted << ((32u - rot) & 31u));
    }
    CRT_DEV void init(float cam_seed, uint32_t, uint32_t, uint32_t pixel, float iteration) {
        state = 0; inc = ((uint64_t)(uint32_t)iteration << 1) | 1u;
        next();
        state += ((uint64_t)f2u(cam_seed) << 32) | pixel;
        next();
    }
    CRT_DEV void draw1(uint32_t &a) { a = next(); }
    CRT_DEV void draw2(uint32_t &a, uint32_t &b) { a = next(); b = next(); }
    CRT_DEV void draw3(uint32_t &a, uint32_t &b, uint32_t &c) { a = next(); b = next(); c = next(); }
    CRT_DEV uint64_t pack() const { return state; }
    CRT_DEV void unpack(uint64_t v, uint64_t stream_inc) { state = v; inc = stream_inc; }
};

template <int M> CRT_DEV float hash_1(Rng<M> &r) { uint32_t n; r.draw1(n); return (float)n * (1.0f / 4294967296.0f); }
template <int M> CRT_DEV void hash_2(Rng<M> &r, float &a, float &b) {
    uint32_t n0, n1; r.draw2(n0, n1);
    a = (float)(n0 & 0x7FFFFFFFu) / 2147483648.0f;
    b = (float)(n1 & 0x7FFFFFFFu) / 2147483648.0f;
}
template <int M> CRT_DEV void hash_3(Rng<M> &r, float &a, float &b, float &c) {
    uint32_t n0, n1, n2; r.draw3(n0, n1, n2);
    a = (float)(n0 & 0x7FFFFFFFu) / 2147483648.0f;
    b = (float)(n1 & 0x7FFFFFFFu) / 2147483648.0f;
    c = (float)(n2 & 0x7FFFFFFFu) / 2147483648.0f;
}

// compute.comp:141-144
template <int M> CRT_DEV void random_in_unit_disk(Rng<M> &r, float &ox, float &oy) {
    float hx, hy; hash_2(r, hx, hy);
    hx = hx * 1.0f; hy = hy * kTwoPi;
    float s, c; dm_sincos(hy, s, c);
    ox = hx * c; oy = hx * s;
}

// compute.comp:148-155
template <int M> CRT_DEV V3 random_in_unit_sphere(Rng<M> &r) {
    float h0, h1, h2; hash_3(r, h0, h1, h2);
    float theta = h0 * kTwoPi - 0.0f;
    float hy = h1 * 2.0f - 1.0f;
    float hz = h2 * 1.0f - 0.0f;
    float sinPhi = __builtin_sqrtf(1.0f - hy * hy);
    float rad = dm_cbrt01(hz);
    float s, c; dm_sincos(theta, s, c);
    return rad * mk3(c * sinPhi, s * sinPhi, hy);
}

}  // namespace crt
