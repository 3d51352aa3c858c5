// intersect_wide.h -- closest-hit search over the 8-wide quantised BVH (node format: bvh_wide.hip).
//
// One step = four 16-byte loads (header + 48 quantised plane bytes), eight slab tests, one 8-bit hit mask.
// The hit children of a node wait as ONE stack entry {child_base | imask, node | pending mask}.  The visiting order of
// a group is slot ^ ray octant ascending; "next child" of a pending mask is one byte read from a 2 KiB LDS table
// indexed by (octant, mask) (wide_order_table), cheaper than permuting the mask.  The top entry lives in registers.  Inner children are
// addressed as child_base + popcount(imask below slot); the link of a leaf slot is loaded only when that slot is
// actually taken.
//
// Conservativeness.  The planes of a child box are the real numbers P = origin + q * 2^e.  A step evaluates
// t = q * A + B with A = (1/d) * 2^e (exact) and B = fma(origin, 1/d, -(o * 1/d)); the rounding of 1/d, of o * 1/d,
// of B and of the final fma moves t by at most 2^-23 |t| + 2^-24 (|o/d| + |B|), and |t| <= |B| + 255 |A| for every
// plane of the node.  Near planes therefore use B - err, far planes B + err with
// err = 2^-21 (|o/d| + |B| + 256 |A|) (four times the bound, which also absorbs the rounding of B -+ err), and the
// far bound is the best hit so far times (1 + 2^-18): a child whose true box the ray enters before the best hit is
// never culled.
#pragma once
#include "intersect.h"

namespace crt {

#ifndef CRT_WIDE_LDS_STACK
#define CRT_WIDE_LDS_STACK 12
#endif
constexpr int kWideLds = CRT_WIDE_LDS_STACK;          // 8-byte entries per lane in LDS; deeper levels spill to scratch
constexpr int kWideSpill = 64;                        // the build refuses trees deeper than 64 levels (lds_depth may be as low as 1)
constexpr uint32_t kWideNodeF4 = 4;                   // = kWideStrideF4 (build_internal.h)
constexpr uint32_t kEmptyLeaf = kLeafTag | (3u << 29);

struct WideTrav {
    V3 o, d, idir, noid;
    float best_t;              // the far bound of the box tests is best_t * (1 + 2^-18), recomputed per visit (one v_mul: cheaper than a register)
    uint32_t best_id;
    uint32_t cur;              // inner: node index; leaf: kLeafTag (the slot's link word is in `link`)
    uint32_t link;             // leaf: tagged entry (kLeafTag | kind | count | first).  Apart from `cur` so that nothing waits for
                               // the link's load before the leaf step: a wave's next node fetches are issued while it is in flight
    uint32_t g_hdr, g_pend;    // top group: child_base << 8 | imask ; node << 8 | pending mask (bit = slot)
    uint32_t oct8;             // ray octant << 8 (octant bit a set: direction component a is negative)
    int sp;
};

CRT_DEV void wide_begin(WideTrav &tr, V3 o, V3 d) {
    tr.o = o; tr.d = d;
    tr.idir.x = __builtin_fabsf(d.x) > 1e-30f ? 1.0f / d.x : __builtin_copysignf(1e30f, d.x);
    tr.idir.y = __builtin_fabsf(d.y) > 1e-30f ? 1.0f / d.y : __builtin_copysignf(1e30f, d.y);
    tr.idir.z = __builtin_fabsf(d.z) > 1e-30f ? 1.0f / d.z : __builtin_copysignf(1e30f, d.z);
    tr.noid = mk3(-(o.x * tr.idir.x), -(o.y * tr.idir.y), -(o.z * tr.idir.z));
    tr.oct8 = ((f2u(tr.idir.x) >> 31) | ((f2u(tr.idir.y) >> 31) << 1) | ((f2u(tr.idir.z) >> 31) << 2)) << 8;
    tr.best_t = kInf; tr.best_id = kIdMiss;
    tr.cur = 0; tr.link = 0; tr.g_hdr = 0; tr.g_pend = 0; tr.sp = 0;
}

// Next child in visiting order; false when nothing is left (best_t / best_id are final).
// SPILL = false: the tree has at most kWideLds + 1 levels, so the LDS stack cannot overflow and no scratch is touched.
// order[(octant << 8) | mask] = the slot s of `mask` with the smallest s ^ octant (filled by wide_order_table).
constexpr uint32_t kWideOrderBytes = 8 * 256;
CRT_DEV void wide_order_table(uint8_t *order, uint32_t tid, uint32_t n_threads) {
    for (uint32_t e = tid; e < kWideOrderBytes; e += n_threads) {
        const uint32_t oct = e >> 8, mask = e & 0xFFu;
        uint32_t best = 0, best_pos = 8;
        for (uint32_t sl = 0; sl < 8; sl++)
            if (((mask >> sl) & 1u) && (sl ^ oct) < best_pos) { best_pos = sl ^ oct; best = sl; }
        order[e] = (uint8_t)best;
    }
}

template <bool SPILL>
CRT_DEV bool wide_advance(const uint32_t *links, const uint8_t *order, WideTrav &tr, const uint2 *stack, uint32_t stack_stride, const uint2 *spill, int lds_depth) {
    if ((tr.g_pend & 0xFFu) == 0u) {
        if (tr.sp == 0) return false;
        tr.sp--;
        uint2 e;
        if (!SPILL || tr.sp < lds_depth) e = stack[tr.sp * stack_stride];
        else e = spill[tr.sp - lds_depth];
        tr.g_hdr = e.x; tr.g_pend = e.y;
    }
    const uint32_t slot = order[tr.oct8 | (tr.g_pend & 0xFFu)];     // the mask is non-zero
    const uint32_t bit = 1u << slot, imask = tr.g_hdr & 0xFFu;
    tr.g_pend ^= bit;
    // (both fields are assigned on both paths: with one store per path the compiler merges them into a single store through a
    // selected address and keeps the pair in scratch memory)
    uint32_t cur = (tr.g_hdr >> 8) + (uint32_t)__builtin_popcount(imask & (bit - 1u)), link = tr.link;
    if (!(imask & bit)) { cur = kLeafTag; link = links[(size_t)(tr.g_pend >> 8) * 8u + slot]; }
    tr.cur = cur; tr.link = link;
    return true;
}

typedef float f32x2 __attribute__((ext_vector_type(2)));
CRT_DEV float ubyte_f(uint32_t w, int k) { return (float)((w >> (8 * k)) & 0xFFu); }     // v_cvt_f32_ubyteK
CRT_DEV f32x2 ubyte_f2(uint32_t w, int k) { f32x2 r; r.x = ubyte_f(w, k); r.y = ubyte_f(w, k + 1); return r; }
CRT_DEV f32x2 splat2(float v) { f32x2 r; r.x = v; r.y = v; return r; }
CRT_DEV f32x2 fma2(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }        // v_pk_fma_f32

// Issue cost on gfx950 (tools/ubench_valu.hip, profiles/r03_ubench_valu.txt): v_add/sub/mul/fma_f32, v_mov and v_and/or/xor
// issue every 2 cycles per SIMD once two waves are resident, everything else here (v_cvt_f32_ubyteN, v_pk_fma_f32,
// v_min/max[3]_f32, v_cmp, v_cndmask, shifts) every 4, and a v_cndmask whose mask is a VCC written long before costs ~20.
// Hence: the near/far selects take their lane masks from plain SGPR pairs (inline asm: the register allocator would hand
// one of the three masks to VCC), and the hit test is sign arithmetic instead of clamp + compare + select.

// lane ? b : a with the lane mask in an SGPR pair (never VCC)
CRT_DEV uint32_t sel_sgpr(uint32_t a, uint32_t b, unsigned long long mask) {
    uint32_t r;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(mask));
    return r;
}

// The eight slab tests of one node for one ray, planes already split into near / far words (bytes = slots, two words per
// axis and side): hit mask, bit = slot.
CRT_DEV uint32_t wide_test8_core(float4 h, const uint32_t (&qn)[3][2], const uint32_t (&qf)[3][2], V3 idir, V3 noid, float tfar) {
    const uint32_t hx = f2u(h.x), hy = f2u(h.y), hz = f2u(h.z);
    constexpr float kErr = 4.76837158203125e-07f;      // 2^-21
    // per-axis plane coefficients: t_near = q * a + bn, t_far = q * a + bf.  Bit 8 of a header word is zero, so
    // word << 23 is the grid step 2^(e - 127) as a float.
    const float ax = idir.x * u2f(hx << 23), ay = idir.y * u2f(hy << 23), az = idir.z * u2f(hz << 23);
    const float bx = __builtin_fmaf(u2f(hx & 0xFFFFFE00u), idir.x, noid.x);
    const float by = __builtin_fmaf(u2f(hy & 0xFFFFFE00u), idir.y, noid.y);
    const float bz = __builtin_fmaf(u2f(hz & 0xFFFFFE00u), idir.z, noid.z);
    // |t| <= |b| + 255 |a| for every plane of this node, so err below covers the relative part of the error too
    const float ex = __builtin_fmaf(__builtin_fabsf(ax), 256.0f, __builtin_fabsf(noid.x) + __builtin_fabsf(bx)) * kErr;
    const float ey = __builtin_fmaf(__builtin_fabsf(ay), 256.0f, __builtin_fabsf(noid.y) + __builtin_fabsf(by)) * kErr;
    const float ez = __builtin_fmaf(__builtin_fabsf(az), 256.0f, __builtin_fabsf(noid.z) + __builtin_fabsf(bz)) * kErr;
    const f32x2 ax2 = splat2(ax), ay2 = splat2(ay), az2 = splat2(az);
    const f32x2 bnx = splat2(bx - ex), bny = splat2(by - ey), bnz = splat2(bz - ez);
    const f32x2 bfx = splat2(bx + ex), bfy = splat2(by + ey), bfz = splat2(bz + ez);
    // A slot is hit iff max(N, 0) <= min(X, tfar) with N = the latest near plane, X = the earliest far plane, i.e. iff
    // X - N, X and tfar - N are all non-negative (tfar >= 0): the sign bit of the OR of the three words says "miss".
    // (X = -0 reads as a miss: the box ends at the ray's origin and nothing in it can pass t > 0.001.  A NaN from
    // inf - inf is the positive default NaN: a hit, the conservative answer.)  The sign bits are shifted into the mask
    // from slot 7 down to slot 0.
    uint32_t miss = 0;
#pragma unroll
    for (int w = 1; w >= 0; w--) {
#pragma unroll
        for (int k = 2; k >= 0; k -= 2) {
            const f32x2 tnx = fma2(ubyte_f2(qn[0][w], k), ax2, bnx), tfx = fma2(ubyte_f2(qf[0][w], k), ax2, bfx);
            const f32x2 tny = fma2(ubyte_f2(qn[1][w], k), ay2, bny), tfy = fma2(ubyte_f2(qf[1][w], k), ay2, bfy);
            const f32x2 tnz = fma2(ubyte_f2(qn[2][w], k), az2, bnz), tfz = fma2(ubyte_f2(qf[2][w], k), az2, bfz);
            const float n1 = __builtin_fmaxf(__builtin_fmaxf(tnx.y, tny.y), tnz.y), x1 = __builtin_fminf(__builtin_fminf(tfx.y, tfy.y), tfz.y);
            const float n0 = __builtin_fmaxf(__builtin_fmaxf(tnx.x, tny.x), tnz.x), x0 = __builtin_fminf(__builtin_fminf(tfx.x, tfy.x), tfz.x);
            const uint32_t s1 = f2u(x1 - n1) | f2u(x1) | f2u(tfar - n1);
            const uint32_t s0 = f2u(x0 - n0) | f2u(x0) | f2u(tfar - n0);
            miss = __builtin_amdgcn_alignbit(miss, s1, 31);      // (miss << 1) | sign(s1)
            miss = __builtin_amdgcn_alignbit(miss, s0, 31);
        }
    }
    return ~miss & 0xFFu;
}

// The eight slab tests of one node (its four 16-byte rows) for one ray: hit mask, bit = slot.
// Near planes are the low ones for a non-negative direction component, the high ones otherwise.
CRT_DEV uint32_t wide_test8(float4 h, float4 r1, float4 r2, float4 r3, V3 idir, V3 noid, float tfar) {
    const unsigned long long mx = __ballot((int32_t)f2u(idir.x) < 0), my = __ballot((int32_t)f2u(idir.y) < 0), mz = __ballot((int32_t)f2u(idir.z) < 0);
    const uint32_t lox[2] = {f2u(r1.x), f2u(r1.y)}, loy[2] = {f2u(r1.z), f2u(r1.w)}, loz[2] = {f2u(r2.x), f2u(r2.y)};
    const uint32_t hix[2] = {f2u(r2.z), f2u(r2.w)}, hiy[2] = {f2u(r3.x), f2u(r3.y)}, hiz[2] = {f2u(r3.z), f2u(r3.w)};
    const uint32_t qn[3][2] = {{sel_sgpr(lox[0], hix[0], mx), sel_sgpr(lox[1], hix[1], mx)}, {sel_sgpr(loy[0], hiy[0], my), sel_sgpr(loy[1], hiy[1], my)},
                               {sel_sgpr(loz[0], hiz[0], mz), sel_sgpr(loz[1], hiz[1], mz)}};
    const uint32_t qf[3][2] = {{sel_sgpr(hix[0], lox[0], mx), sel_sgpr(hix[1], lox[1], mx)}, {sel_sgpr(hiy[0], loy[0], my), sel_sgpr(hiy[1], loy[1], my)},
                               {sel_sgpr(hiz[0], loz[0], mz), sel_sgpr(hiz[1], loz[1], mz)}};
    return wide_test8_core(h, qn, qf, idir, noid, tfar);
}
// nx, ny, nz: the direction component is negative, wave-uniform (k_wf_primary, when the whole packet shares an octant:
// the twelve selects are scalar then)
CRT_DEV uint32_t wide_test8(float4 h, float4 r1, float4 r2, float4 r3, V3 idir, V3 noid, float tfar, bool nx, bool ny, bool nz) {
    const uint32_t lox[2] = {f2u(r1.x), f2u(r1.y)}, loy[2] = {f2u(r1.z), f2u(r1.w)}, loz[2] = {f2u(r2.x), f2u(r2.y)};
    const uint32_t hix[2] = {f2u(r2.z), f2u(r2.w)}, hiy[2] = {f2u(r3.x), f2u(r3.y)}, hiz[2] = {f2u(r3.z), f2u(r3.w)};
    const uint32_t qn[3][2] = {{nx ? hix[0] : lox[0], nx ? hix[1] : lox[1]}, {ny ? hiy[0] : loy[0], ny ? hiy[1] : loy[1]}, {nz ? hiz[0] : loz[0], nz ? hiz[1] : loz[1]}};
    const uint32_t qf[3][2] = {{nx ? lox[0] : hix[0], nx ? lox[1] : hix[1]}, {ny ? loy[0] : hiy[0], ny ? loy[1] : hiy[1]}, {nz ? loz[0] : hiz[0], nz ? loz[1] : hiz[1]}};
    return wide_test8_core(h, qn, qf, idir, noid, tfar);
}

// Node visit without the final wide_advance: tests the eight children of node tr.cur and makes them the top group.
template <bool COUNT, bool SPILL>
CRT_DEV void wide_visit(const float4 *nodes, WideTrav &tr, uint2 *stack, uint32_t stack_stride, uint2 *spill, int lds_depth, Counts &cnt) {
    const uint32_t node = tr.cur;
    const float4 *p = nodes + (size_t)kWideNodeF4 * node;
    const float4 h = p[0], r1 = p[1], r2 = p[2], r3 = p[3];
    if (COUNT) cnt.nodes += 8;
    const uint32_t m = wide_test8(h, r1, r2, r3, tr.idir, tr.noid, tr.best_t * 1.000004f);
    if (m != 0u) {
        if ((tr.g_pend & 0xFFu) != 0u) {
            const uint2 e = make_uint2(tr.g_hdr, tr.g_pend);
            if (!SPILL || tr.sp < lds_depth) stack[tr.sp * stack_stride] = e;
            else if (tr.sp < lds_depth + kWideSpill) spill[tr.sp - lds_depth] = e;
            tr.sp++;
        }
        tr.g_hdr = f2u(h.w);
        tr.g_pend = (node << 8) | m;
    }
}

// Inner step: tr.cur is a node index.
template <bool COUNT, bool SPILL>
CRT_DEV bool wide_inner(const float4 *nodes, const uint32_t *links, const uint8_t *order, WideTrav &tr, uint2 *stack, uint32_t stack_stride, uint2 *spill, int lds_depth, Counts &cnt) {
    wide_visit<COUNT, SPILL>(nodes, tr, stack, stack_stride, spill, lds_depth, cnt);
    return wide_advance<SPILL>(links, order, tr, stack, stack_stride, spill, lds_depth);
}

// Leaf step: tr.cur is a tagged leaf entry.
template <bool COUNT, bool SPILL, bool EXT = false>
CRT_DEV bool wide_leaf(const DeviceScene &sc, const uint8_t *order, WideTrav &tr, const uint2 *stack, uint32_t stack_stride, const uint2 *spill, int lds_depth, Counts &cnt) {
    const uint32_t e = tr.link;
    const uint32_t kind = (e >> 29) & 3u, n = ((e >> 26) & 7u) + 1u, first = e & (kMaxLeafLink - 1u);
    if (kind != 3u) {                                   // 3 = empty slot (a box test can only reach it through rounding)
        test_leaf<COUNT, EXT>(sc, tr.o, tr.d, first, (kind << 28) | n, tr.best_t, tr.best_id, cnt);
    }
    return wide_advance<SPILL>(sc.wide_links, order, tr, stack, stack_stride, spill, lds_depth);
}

}  // namespace crt
