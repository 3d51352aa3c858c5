// debug_math.hip -- crt_debug_math: the build-defined FP32 sequences of device_math.h (and the two IEEE operations the
// arithmetic contract leans on, square root and division) evaluated on the GPU for arrays of arguments, so that a test
// can compare them bit for bit with the oracle's statement of the same sequences (oracle.cpp: orc_math, same function
// numbers) and with the committed vectors, function by function instead of only through rendered images.
#include <hip/hip_runtime.h>

#include <string>

#include "device_math.h"

namespace crt {

__global__ void k_debug_math(int fn, uint32_t n, const float *a, const float *b, float *out0, float *out1) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float r0 = 0.0f, r1 = 0.0f;
    switch (fn) {
        case 0: dm_sincos(a[i], r0, r1); break;
        case 1: r0 = dm_cbrt01(a[i]); break;
        case 2: r0 = dm_acos(a[i]); break;
        case 3: r0 = dm_atan2(a[i], b[i]); break;
        case 4: r0 = dm_pow5(a[i]); break;
        case 5: r0 = __builtin_sqrtf(a[i]); break;
        case 6: r0 = a[i] / b[i]; break;
        default: break;
    }
    out0[i] = r0; out1[i] = r1;
}

// a, b, out0, out1: device arrays of n floats.  Enqueue-only.
bool debug_math(hipStream_t stream, int fn, uint32_t n, const float *a, const float *b, float *out0, float *out1, std::string &err) {
    if (fn < 0 || fn > 6) { err = "unknown function number"; return false; }
    if (n == 0) return true;
    hipLaunchKernelGGL(k_debug_math, dim3((n + 255u) / 256u), dim3(256), 0, stream, fn, n, a, b, out0, out1);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { err = hipGetErrorString(e); return false; }
    return true;
}

}  // namespace crt
