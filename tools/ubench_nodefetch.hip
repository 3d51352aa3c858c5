// ubench_nodefetch.hip -- how should a lane fetch a 64-byte BVH pair?  A pointer chase over a table of 64-byte records
// (the record holds the index of the next one), every lane on its own chain, with about the arithmetic of two slab tests
// per step.  Variants differ only in how the 64 bytes reach the lane's registers.
//   hipcc -O3 --offload-arch=gfx950 -o ubench_nodefetch tools/ubench_nodefetch.hip && ./ubench_nodefetch
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <numeric>
#include <algorithm>
#include <random>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ float consume(float4 a, float4 b, float4 c, float4 d, float s) {
    // ~ two slab tests worth of arithmetic
    float x0 = fmaf(a.x, s, 1.0f), x1 = fmaf(b.x, s, 1.0f), y0 = fmaf(a.y, s, 2.0f), y1 = fmaf(b.y, s, 2.0f), z0 = fmaf(a.z, s, 3.0f), z1 = fmaf(b.z, s, 3.0f);
    float u0 = fmaf(c.x, s, 1.0f), u1 = fmaf(d.x, s, 1.0f), v0 = fmaf(c.y, s, 2.0f), v1 = fmaf(d.y, s, 2.0f), w0 = fmaf(c.z, s, 3.0f), w1 = fmaf(d.z, s, 3.0f);
    float tn = fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), fminf(z0, z1)), tf = fminf(fminf(fmaxf(x0, x1), fmaxf(y0, y1)), fmaxf(z0, z1));
    float sn = fmaxf(fmaxf(fminf(u0, u1), fminf(v0, v1)), fminf(w0, w1)), sf = fminf(fminf(fmaxf(u0, u1), fmaxf(v0, v1)), fmaxf(w0, w1));
    return (tn <= tf ? tn : 0.0f) + (sn <= sf ? sn : 0.0f);
}

// A: four 16-byte loads from the lane's own record, back to back (what k_wf_extend does)
__global__ __launch_bounds__(256, 6) void k_a(const float4 *t, const uint32_t *start, int steps, float *out) {
    uint32_t g = blockIdx.x * 256 + threadIdx.x, cur = start[g];
    float acc = 0.0f;
    for (int i = 0; i < steps; i++) {
        const float4 *p = t + 4 * (size_t)cur;
        float4 a = p[0], b = p[1], c = p[2], d = p[3];
        acc += consume(a, b, c, d, acc);
        cur = __float_as_uint(a.w);
    }
    out[g] = acc;
}

// C: one 16-byte load, wait for it, then the other three (they hit the line the first one brought into L1)
__global__ __launch_bounds__(256, 6) void k_c(const float4 *t, const uint32_t *start, int steps, float *out) {
    uint32_t g = blockIdx.x * 256 + threadIdx.x, cur = start[g];
    float acc = 0.0f;
    for (int i = 0; i < steps; i++) {
        const float4 *p = t + 4 * (size_t)cur;
        float4 a = p[0];
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(a.x), "+v"(a.y), "+v"(a.z), "+v"(a.w) :: "memory");
        float4 b = p[1], c = p[2], d = p[3];
        acc += consume(a, b, c, d, acc);
        cur = __float_as_uint(a.w);
    }
    out[g] = acc;
}

// D: the four lanes of a quad fetch each other's records: instruction k loads the record of quad lane k, 16 bytes per lane
// (one 64-byte segment per quad and instruction); a 4x4 transpose through LDS hands every lane its own record.
__global__ __launch_bounds__(256, 6) void k_d(const float4 *t, const uint32_t *start, int steps, float *out) {
    __shared__ float4 s_x[256 * 4];
    uint32_t g = blockIdx.x * 256 + threadIdx.x, cur = start[g];
    const uint32_t lane = threadIdx.x, q = lane & 3u, qbase = lane & ~3u;
    float acc = 0.0f;
    for (int i = 0; i < steps; i++) {
        float4 r[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            uint32_t c = __shfl(cur, (int)((lane & 63u & ~3u) + k));
            r[k] = t[4 * (size_t)c + q];
        }
        // lane (qbase + k) needs part q from this lane: s_x[(qbase + k) * 4 + q]
#pragma unroll
        for (int k = 0; k < 4; k++) s_x[(qbase + k) * 4 + q] = r[k];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        float4 a = s_x[lane * 4 + 0], b = s_x[lane * 4 + 1], c = s_x[lane * 4 + 2], d = s_x[lane * 4 + 3];
        __builtin_amdgcn_wave_barrier();
        acc += consume(a, b, c, d, acc);
        cur = __float_as_uint(a.w);
    }
    out[g] = acc;
}

// E: the two lanes of a pair fetch each other's records: instruction (k, h) loads 32 bytes [32 h, 32 h + 32) of the record of
// pair lane k, 16 bytes per lane; a lane reads the even 16-byte parts of its own record itself and gets the odd parts from
// its partner with a DPP swap.  No LDS.
__device__ __forceinline__ float dpp_swap1(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, false)); }
__device__ __forceinline__ float4 sel4(bool c, float4 a, float4 b) { return make_float4(c ? a.x : b.x, c ? a.y : b.y, c ? a.z : b.z, c ? a.w : b.w); }
__global__ __launch_bounds__(256, 6) void k_e(const float4 *t, const uint32_t *start, int steps, float *out) {
    uint32_t g = blockIdx.x * 256 + threadIdx.x, cur = start[g];
    const uint32_t pj = threadIdx.x & 1u;
    float acc = 0.0f;
    for (int i = 0; i < steps; i++) {
        const uint32_t other = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)cur, 0xB1, 0xF, 0xF, false);
        const uint32_t c0 = pj ? other : cur, c1 = pj ? cur : other;        // records of pair lane 0 and pair lane 1
        float4 r00 = t[4 * (size_t)c0 + 0 + (pj ^ 0u)], r01 = t[4 * (size_t)c0 + 2 + (pj ^ 0u)];
        float4 r10 = t[4 * (size_t)c1 + 0 + (pj ^ 1u)], r11 = t[4 * (size_t)c1 + 2 + (pj ^ 1u)];
        float4 a = sel4(pj, r10, r00), c = sel4(pj, r11, r01);               // parts 0 and 2 of my record
        float4 sb = sel4(pj, r00, r10), sd = sel4(pj, r01, r11);             // parts 1 and 3 of my partner's record
        float4 b = make_float4(dpp_swap1(sb.x), dpp_swap1(sb.y), dpp_swap1(sb.z), dpp_swap1(sb.w));
        float4 d = make_float4(dpp_swap1(sd.x), dpp_swap1(sd.y), dpp_swap1(sd.z), dpp_swap1(sd.w));
        acc += consume(a, b, c, d, acc);
        cur = __float_as_uint(a.w);
    }
    out[g] = acc;
}

// F: 32-byte records (two loads): what a quantised pair would cost to fetch
__global__ __launch_bounds__(256, 6) void k_f(const float4 *t, const uint32_t *start, int steps, float *out) {
    uint32_t g = blockIdx.x * 256 + threadIdx.x, cur = start[g];
    float acc = 0.0f;
    for (int i = 0; i < steps; i++) {
        const float4 *p = t + 4 * (size_t)cur;
        float4 a = p[0], b = p[1];
        acc += consume(a, b, b, a, acc);
        cur = __float_as_uint(a.w);
    }
    out[g] = acc;
}

// G: as A but the record is read with two 16-byte loads 32 bytes apart plus two later ones (order 0,2,1,3)
__global__ __launch_bounds__(256, 6) void k_g(const float4 *t, const uint32_t *start, int steps, float *out) {
    uint32_t g = blockIdx.x * 256 + threadIdx.x, cur = start[g];
    float acc = 0.0f;
    for (int i = 0; i < steps; i++) {
        const float4 *p = t + 4 * (size_t)cur;
        float4 a = p[0];
        typedef float f4 __attribute__((ext_vector_type(4)));
        const f4 *pv = (const f4 *)p;
        f4 bv = __builtin_nontemporal_load(pv + 1), cv = __builtin_nontemporal_load(pv + 2), dv = __builtin_nontemporal_load(pv + 3);
        float4 b = make_float4(bv.x, bv.y, bv.z, bv.w), c = make_float4(cv.x, cv.y, cv.z, cv.w), d = make_float4(dv.x, dv.y, dv.z, dv.w);
        acc += consume(a, b, c, d, acc);
        cur = __float_as_uint(a.w);
    }
    out[g] = acc;
}

// H: quad-cooperative, straight into LDS: in round r the four lanes of a quad load the four 16-byte quarters of the record of
// quad lane r with global_load_lds_dwordx4 (one contiguous 64-byte request per quad; destination = the wave's staging row r,
// lane-linear), so the record of lane L ends up contiguous in row L & 3 at slot L & ~3: four ds_read_b128, no transpose, no
// VGPR destination for the loads.  Rows are 1040 bytes apart (one 16-byte pad) so that the four lanes of a quad read
// different banks.  ACT: only lanes with (lane * 7 + step) % 5 < 3 (60 %) take a step; a round whose owner lane rests is
// skipped by its quad (what a traversal kernel with divergent lanes would do).
template <bool ACT>
__global__ __launch_bounds__(256, 6) void k_h(const float4 *t, const uint32_t *start, int steps, float *out) {
    __shared__ float4 s_x[4][4][65];
    uint32_t g = blockIdx.x * 256 + threadIdx.x, cur = start[g];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, q = lane & 3u, qbase = lane & ~3u;
    float acc = 0.0f;
    for (int i = 0; i < steps; i++) {
        const bool act = !ACT || (lane * 7u + (uint32_t)i) % 5u < 3u;
        const uint32_t want = act ? cur : 0xFFFFFFFFu;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const uint32_t c = (uint32_t)__shfl((int)want, (int)(qbase + r));
            if (c != 0xFFFFFFFFu) __builtin_amdgcn_global_load_lds((const void *)(t + 4 * (size_t)c + q), (__attribute__((address_space(3))) void *)&s_x[wave][r][0], 16, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (act) {
            const float4 *row = &s_x[wave][q][qbase];
            float4 a = row[0], b = row[1], c = row[2], d = row[3];
            acc += consume(a, b, c, d, acc);
            cur = __float_as_uint(a.w);
        }
        __builtin_amdgcn_wave_barrier();
    }
    out[g] = acc;
}

// A with the same 60 % lane activity as k_h<true>
__global__ __launch_bounds__(256, 6) void k_a_act(const float4 *t, const uint32_t *start, int steps, float *out) {
    uint32_t g = blockIdx.x * 256 + threadIdx.x, cur = start[g];
    const uint32_t lane = threadIdx.x & 63u;
    float acc = 0.0f;
    for (int i = 0; i < steps; i++) {
        if ((lane * 7u + (uint32_t)i) % 5u < 3u) {
            const float4 *p = t + 4 * (size_t)cur;
            float4 a = p[0], b = p[1], c = p[2], d = p[3];
            acc += consume(a, b, c, d, acc);
            cur = __float_as_uint(a.w);
        }
    }
    out[g] = acc;
}

int main(int argc, char **argv) {
    const uint32_t n = argc > 1 ? atoi(argv[1]) : 225000;     // records (64 B each)
    const int steps = argc > 2 ? atoi(argv[2]) : 256;
    const uint32_t blocks = 1536, lanes = blocks * 256;
    std::vector<uint32_t> perm(n);
    std::iota(perm.begin(), perm.end(), 0u);
    std::mt19937 rng(1);
    std::shuffle(perm.begin(), perm.end(), rng);
    std::vector<float> tab((size_t)n * 16);
    for (uint32_t i = 0; i < n; i++) {
        for (int k = 0; k < 16; k++) tab[(size_t)perm[i] * 16 + k] = (float)((i * 16 + k) % 97) * 0.01f;
        uint32_t next = perm[(i + 1) % n];
        memcpy(&tab[(size_t)perm[i] * 16 + 3], &next, 4);
    }
    std::vector<uint32_t> start(lanes);
    for (uint32_t i = 0; i < lanes; i++) start[i] = rng() % n;
    float4 *d_t; uint32_t *d_s; float *d_o;
    CK(hipMalloc(&d_t, tab.size() * 4)); CK(hipMalloc(&d_s, lanes * 4)); CK(hipMalloc(&d_o, lanes * 4));
    CK(hipMemcpy(d_t, tab.data(), tab.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_s, start.data(), lanes * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    struct V { const char *name; void (*k)(const float4 *, const uint32_t *, int, float *); } vs[] = {
        {"A 4 x b128 own record", k_a}, {"C b128, wait, 3 x b128", k_c}, {"D quad-cooperative + LDS transpose", k_d}, {"E pair-cooperative + DPP swap", k_e},
        {"F 2 x b128 (32-byte record)", k_f}, {"G b128 + 3 nontemporal b128", k_g}, {"H quad-cooperative, direct to LDS", k_h<false>},
        {"A at 60 % lane activity (x0.6)", k_a_act}, {"H at 60 % lane activity (x0.6)", k_h<true>}};
    for (auto &v : vs) {
        for (int rep = 0; rep < 2; rep++) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(v.k, dim3(blocks), dim3(256), 0, 0, d_t, d_s, steps, d_o);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) printf("%-40s %8.3f ms  %7.2f G fetches/s  %6.2f TB/s\n", v.name, ms, (double)lanes * steps / ms / 1e6, (double)lanes * steps * 64 / ms / 1e9);
        }
    }
    return 0;
}
