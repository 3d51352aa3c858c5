// bvh_wide_node.h -- the second half of writing one 8-wide node, shared by the direct build's bottom-up pass
// (bvh_direct.hip: k_dw_emit) and the refit (bvh_refit.hip: k_wide_refit): eight lanes per node, one per slot, each holding
// its child's exact box; the node box is reduced over the group, every lane quantises its own slot against it (the
// arithmetic of wide_emit, bvh_wide.hip: origin rounded down to 15 mantissa bits, per-axis power-of-two grid step,
// outward rounding with 2^-10 of slack) and the plane bytes are OR-reduced over the group; lanes 0-3 write one 16-byte
// row each, lanes 0 and 1 the node's exact box for its parent.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "build_internal.h"

namespace crt {

__device__ __forceinline__ uint32_t or_quad(uint32_t v) {      // OR over the four lanes of a quad
    v |= (uint32_t)__shfl_xor((int)v, 1); v |= (uint32_t)__shfl_xor((int)v, 2);
    return v;
}

// Every lane of the wave calls this (shuffles inside); `active` is uniform over a group of eight lanes.  area (optional):
// receives the node's surface area in every lane of the group (0 for an inactive group).  box (optional): the node's exact
// box for its parent (bottom-up passes).
__device__ __forceinline__ void wide_node_finish(bool active, uint32_t sl, uint32_t id, uint32_t hdr, const float clo[3], const float chi[3], bool valid,
                                                 float4 *wide, float4 *box, float *area) {
    // node box over the group (a NaN coordinate of a primitive stays in its own slot's box: fminf / fmaxf drop it here,
    // and the slot's quantisation below maps it to the full range)
    float lo[3], hi[3];
#pragma unroll
    for (int a = 0; a < 3; a++) {
        float l = clo[a], h = chi[a];
#pragma unroll
        for (int off = 1; off < 8; off <<= 1) { l = fminf(l, __shfl_xor(l, off)); h = fmaxf(h, __shfl_xor(h, off)); }
        lo[a] = l; hi[a] = h;
    }
    const bool any = __builtin_popcount((uint32_t)((__ballot(valid) >> (threadIdx.x & 56u)) & 0xFFull)) != 0;
    if (!any) { for (int a = 0; a < 3; a++) { lo[a] = 0.0f; hi[a] = 0.0f; } }
    if (area) {
        const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        *area = (active && any) ? 2.0f * (dx * dy + dy * dz + dz * dx) : 0.0f;
    }
    // origin (rounded down to 15 mantissa bits) and per-axis grid step: wide_emit's arithmetic (bvh_wide.hip)
    float origin[3], inv_scale[3]; uint32_t ebits[3];
#pragma unroll
    for (int a = 0; a < 3; a++) {
        uint32_t b = __float_as_uint(lo[a]), tr = b & 0xFFFFFE00u;
        if ((b & 0x80000000u) && tr != b) tr += 0x200u;          // negative: truncation moved it up, step one unit down
        origin[a] = __uint_as_float(tr);
        const float ext = hi[a] - origin[a];
        int e = -100;
        if (ext > 0.0f) {
            int x; const float m = frexpf(ext / 255.0f, &x);
            e = m > 0.999999f ? x + 1 : x;
        }
        if (!(ext <= 3.402823466e38f)) e = 122;
        e = e < -100 ? -100 : e > 122 ? 122 : e;
        ebits[a] = (uint32_t)(e + 127);
        inv_scale[a] = __uint_as_float((uint32_t)(127 - e) << 23);
    }
    // this slot's six plane bytes, shifted to their place in the slot's dword; empty slots keep qlo = 255, qhi = 0
    const uint32_t sh = 8u * (sl & 3u);
    uint32_t w[6];
#pragma unroll
    for (int a = 0; a < 3; a++) {
        float fl = 255.0f, fh = 0.0f;
        if (valid) {
            const float xl = (clo[a] - origin[a]) * inv_scale[a], xh = (chi[a] - origin[a]) * inv_scale[a];
            fl = floorf(xl - 0.0009765625f); fh = ceilf(xh + 0.0009765625f);
            fl = (fl >= 0.0f && fl <= 255.0f) ? fl : 0.0f;
            fh = (fh >= 0.0f && fh <= 255.0f) ? fh : 255.0f;
        }
        w[a] = (uint32_t)fl << sh; w[3 + a] = (uint32_t)fh << sh;
    }
    uint32_t mine[6], other[6];      // the dword of this lane's quad (slots 0-3 or 4-7) and the other quad's
#pragma unroll
    for (int k = 0; k < 6; k++) { mine[k] = or_quad(w[k]); other[k] = (uint32_t)__shfl_xor((int)mine[k], 4); }
    if (!active) return;
    const bool upper = sl >= 4u;
    uint32_t q[6][2];
#pragma unroll
    for (int k = 0; k < 6; k++) { q[k][0] = upper ? other[k] : mine[k]; q[k][1] = upper ? mine[k] : other[k]; }
    float4 *o = wide + (size_t)kWideStrideF4 * id;
    if (sl == 0u) {
        o[0] = make_float4(__uint_as_float(__float_as_uint(origin[0]) | ebits[0]), __uint_as_float(__float_as_uint(origin[1]) | ebits[1]),
                           __uint_as_float(__float_as_uint(origin[2]) | ebits[2]), __uint_as_float(hdr));
        if (box) box[2 * (size_t)id] = make_float4(lo[0], lo[1], lo[2], 0.0f);
    } else if (sl == 1u) {
        o[1] = make_float4(__uint_as_float(q[0][0]), __uint_as_float(q[0][1]), __uint_as_float(q[1][0]), __uint_as_float(q[1][1]));
        if (box) box[2 * (size_t)id + 1] = make_float4(hi[0], hi[1], hi[2], 0.0f);
    } else if (sl == 2u) o[2] = make_float4(__uint_as_float(q[2][0]), __uint_as_float(q[2][1]), __uint_as_float(q[3][0]), __uint_as_float(q[3][1]));
    else if (sl == 3u) o[3] = make_float4(__uint_as_float(q[4][0]), __uint_as_float(q[4][1]), __uint_as_float(q[5][0]), __uint_as_float(q[5][1]));
}

}  // namespace crt
