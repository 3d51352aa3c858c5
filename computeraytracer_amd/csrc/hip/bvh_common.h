// bvh_common.h -- device helpers shared by bvh_build.hip (LBVH) and bvh_sah.hip (SAH refinement).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cstdio>
#include <cstdlib>

namespace crt {

// Order-preserving float <-> uint mapping, so that integer atomic min/max order floats.
__device__ __forceinline__ uint32_t enc_f(float f) { uint32_t u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__device__ __forceinline__ float dec_f(uint32_t u) { return __uint_as_float((u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u); }

__device__ __forceinline__ float half_area(float4 lo, float4 hi) {
    float dx = hi.x - lo.x, dy = hi.y - lo.y, dz = hi.z - lo.z;
    return dx * dy + dy * dz + dx * dz;
}
__device__ __forceinline__ float half_area3(const float lo[3], const float hi[3]) {
    float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
    return dx * dy + dy * dz + dx * dz;
}

// CRT_BUILD_TRACE=1 (debugging aid, host side): the build waits after every stage it names and says so on stderr, so that a
// launch that never finishes names itself.
inline void build_trace(hipStream_t stream, const char *stage, uint32_t arg = 0) {
    static const bool on = [] { const char *e = getenv("CRT_BUILD_TRACE"); return e && atoi(e) != 0; }();
    if (!on) return;
    fprintf(stderr, "[crt build] %s %u ...", stage, arg); fflush(stderr);
    const hipError_t e = hipStreamSynchronize(stream);
    fprintf(stderr, " %s\n", e == hipSuccess ? "done" : hipGetErrorString(e)); fflush(stderr);
}

}  // namespace crt
