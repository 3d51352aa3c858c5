// bvh_dp.h -- one node of the wide collapse's dynamic programme (bvh_wide.hip, "the table"), shared by its two drivers:
// the level-by-level pass over a finished pair tree (bvh_wide.hip: k_dp_level) and the pass fused into the SAH rebuild,
// where every workgroup finishes the table of the subtree it has just built while that subtree is still in its LDS
// (bvh_sah.hip: k_sah_group).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "bvh_common.h"

namespace crt {

constexpr uint32_t kDpEffMask = 0x1FFFFFu;      // seven 3-bit entries: eff(1) .. eff(7)

// cl[k], cr[k] (k = 1 .. 7): cost of handing the left / right child's subtree to this node as at most k children (0 for a
// leaf); eff_l / eff_r: the children's own eff tables (0 for a leaf); lo / hi: the two child boxes.
// Out: c[k] (k = 1 .. 7), this node's eff table `own`, and its choice word `w` (layout: bvh_wide.hip).
__device__ __forceinline__ void dp_combine(const float alo[3], const float ahi[3], const float blo[3], const float bhi[3], const float cl[8], const float cr[8],
                                           uint32_t eff_l, uint32_t eff_r, float c[8], uint32_t &own, unsigned long long &w) {
    float lo[3], hi[3];
#pragma unroll
    for (int ax = 0; ax < 3; ax++) { lo[ax] = fminf(alo[ax], blo[ax]); hi[ax] = fmaxf(ahi[ax], bhi[ax]); }
    const float area = half_area3(lo, hi);
    float dist[9]; uint32_t split[9];
#pragma unroll
    for (int j = 2; j <= 8; j++) {
        float best = 3.402823466e38f; uint32_t bk = 1;
#pragma unroll
        for (int k = 1; k < j; k++) {
            if (k > 7 || j - k > 7) continue;
            const float v = cl[k] + cr[j - k];
            if (v < best) { best = v; bk = (uint32_t)k; }
        }
        dist[j] = best; split[j] = bk;
    }
    uint32_t eff[8];
    c[1] = area + dist[8]; eff[1] = 1u;
    if (!(c[1] < 3.402823466e38f)) c[1] = 3.402823466e38f;      // NaN / overflow areas: still a valid (finite) table
#pragma unroll
    for (int k = 2; k < 8; k++) {
        const bool take = dist[k] < c[k - 1];
        c[k] = take ? dist[k] : c[k - 1];
        eff[k] = take ? (uint32_t)k : eff[k - 1];
    }
    w = ((unsigned long long)(eff_l & kDpEffMask) << 21) | ((unsigned long long)(eff_r & kDpEffMask) << 42);
    own = 0u;
#pragma unroll
    for (int j = 2; j <= 8; j++) w |= (unsigned long long)split[j] << (3 * (j - 2));
#pragma unroll
    for (int k = 1; k < 8; k++) own |= eff[k] << (3 * (k - 1));
}

}  // namespace crt
