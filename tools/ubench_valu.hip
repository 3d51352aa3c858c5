// ubench_valu.hip -- what is the VALU issue ceiling of one gfx950 SIMD?  ("defend or correct the VALU roof")
//
// bench.py prices k_wf_extend / k_wf_primary against 1024 SIMDs x one wave64 VALU instruction per 4 cycles (16 lanes per
// clock and SIMD = 39.3 T lane-instructions/s at 2.4 GHz).  MI355X_MICROARCH.md says the CDNA4 SIMD is 32 lanes wide and
// v_fma_f32 issues every 2 cycles once more than one wave is resident.  This program measures it: W waves per SIMD
// (W = 1, 2, 4, 6, 8), every wave running a long unrolled run of independent VALU instructions, one instruction class per
// kernel:
//   fma      v_fma_f32            (8 independent accumulators)
//   pkfma    v_pk_fma_f32         (8 independent accumulator pairs; two FMAs per lane and instruction)
//   cvt      v_cvt_f32_ubyte0..3  (the byte-to-float conversions of a quantised-node visit)
//   minmax   v_max3_f32 / v_min3_f32
//   mix      the instruction mix of one wide_test8 (intersect_wide.h): 48 cvt : 24 pk_fma : 16 max/min : 16 max3/min3 :
//            8 v_cmp : 8 v_cndmask, in that interleaving
// Reported: wave-instructions per clock and SIMD, from the kernel's own s_memrealtime-free measure: HIP-event time x the
// shader clock the runtime reports (hipDeviceAttributeClockRate), and lanes per clock and SIMD = 64 x that.
//   hipcc -O3 --offload-arch=gfx950 -o tools/bin/ubench_valu tools/ubench_valu.hip && tools/bin/ubench_valu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int kUnroll = 16;      // repetitions of the 8-instruction group per loop trip

template <int KIND>
__global__ __launch_bounds__(256) void k_valu(int trips, float seed, float *out) {
    float a0 = seed + threadIdx.x, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
    f32x2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a0}, p5 = {a3, a2}, p6 = {a5, a4}, p7 = {a7, a6};
    const float m = 0.999f, c = 0.001f;
    const f32x2 m2 = {m, m}, c2 = {c, c};
    uint32_t w = __float_as_uint(seed) + threadIdx.x;
    for (int t = 0; t < trips; t++) {
#pragma unroll
        for (int u = 0; u < kUnroll; u++) {
            if (KIND == 0) {
                asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                             "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
            } else if (KIND == 1) {
                asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
                             "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(m2), "v"(c2));
            } else if (KIND == 2) {
                asm volatile("v_cvt_f32_ubyte0 %0, %8\n v_cvt_f32_ubyte1 %1, %8\n v_cvt_f32_ubyte2 %2, %8\n v_cvt_f32_ubyte3 %3, %8\n"
                             "v_cvt_f32_ubyte0 %4, %8\n v_cvt_f32_ubyte1 %5, %8\n v_cvt_f32_ubyte2 %6, %8\n v_cvt_f32_ubyte3 %7, %8\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(w));
            } else if (KIND == 3) {
                asm volatile("v_max3_f32 %0, %0, %1, %8\n v_min3_f32 %1, %1, %2, %8\n v_max3_f32 %2, %2, %3, %8\n v_min3_f32 %3, %3, %4, %8\n"
                             "v_max3_f32 %4, %4, %5, %8\n v_min3_f32 %5, %5, %6, %8\n v_max3_f32 %6, %6, %7, %8\n v_min3_f32 %7, %7, %0, %8\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));
            } else {
                // one slot pair of wide_test8, 15 instructions: 6 cvt, 3 pk_fma (near) ... kept to the real 48:24:16:16:8:8 ratio
                // over a group of 120 = 8 of these blocks
                asm volatile(
                    "v_cvt_f32_ubyte0 %0, %[w]\n v_cvt_f32_ubyte1 %1, %[w]\n v_cvt_f32_ubyte2 %2, %[w]\n v_cvt_f32_ubyte3 %3, %[w]\n"
                    "v_cvt_f32_ubyte0 %4, %[w]\n v_cvt_f32_ubyte1 %5, %[w]\n"
                    "v_pk_fma_f32 %[q0], %[q0], %[m2], %[c2]\n v_pk_fma_f32 %[q1], %[q1], %[m2], %[c2]\n v_pk_fma_f32 %[q2], %[q2], %[m2], %[c2]\n"
                    "v_max_f32 %6, 0, %0\n v_max3_f32 %6, %1, %2, %6\n v_min_f32 %7, %3, %[m]\n v_min3_f32 %7, %4, %5, %7\n"
                    "v_cmp_le_f32 vcc, %6, %7\n v_cndmask_b32 %0, 0, %1, vcc\n"
                    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), [q0] "+v"(p0), [q1] "+v"(p1), [q2] "+v"(p2)
                    : [w] "v"(w), [m] "v"(m), [m2] "v"(m2), [c2] "v"(c2) : "vcc");
            }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y;
}

// ---- survey of instruction classes: which VALU instructions issue at the 2-cycle rate, which at the 4-cycle rate ----
// Each kernel repeats ONE instruction over 8 independent destination registers.
#define SURVEY_KERNEL(NAME, ASM8)                                                                                        \
    __global__ __launch_bounds__(256) void k_sv_##NAME(int trips, float seed, float *out) {                              \
        float a0 = seed + threadIdx.x, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f; \
        float m = 0.999f + seed * 1e-9f, c = 0.001f + seed * 1e-9f;                                                      \
        uint32_t w = __float_as_uint(seed) + threadIdx.x;                                                                \
        float den = __uint_as_float(threadIdx.x & 255u);      /* a denormal operand */                                      \
        for (int t = 0; t < trips; t++) {                                                                                \
            _Pragma("unroll") for (int u = 0; u < kUnroll; u++) {                                                        \
                asm volatile(ASM8 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)      \
                             : [m] "v"(m), [c] "v"(c), [w] "v"(w), [d] "v"(den) : "vcc", "s10", "s11");                                              \
            }                                                                                                            \
        }                                                                                                                \
        out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                                     \
    }
#define R8(I) I(0) I(1) I(2) I(3) I(4) I(5) I(6) I(7)
#define I_ADD(k) "v_add_f32 %" #k ", %" #k ", %[m]\n"
#define I_MUL(k) "v_mul_f32 %" #k ", %" #k ", %[m]\n"
#define I_SUB(k) "v_sub_f32 %" #k ", %" #k ", %[m]\n"
#define I_MAX(k) "v_max_f32 %" #k ", %" #k ", %[m]\n"
#define I_MIN(k) "v_min_f32 %" #k ", %" #k ", %[m]\n"
#define I_MAX3(k) "v_max3_f32 %" #k ", %" #k ", %[m], %[c]\n"
#define I_MED3(k) "v_med3_f32 %" #k ", %" #k ", %[m], %[c]\n"
#define I_MOV(k) "v_mov_b32 %" #k ", %[m]\n"
#define I_AND(k) "v_and_b32 %" #k ", %" #k ", %[w]\n"
#define I_OR(k) "v_or_b32 %" #k ", %" #k ", %[w]\n"
#define I_LSHL(k) "v_lshlrev_b32 %" #k ", 3, %" #k "\n"
#define I_ADDU(k) "v_add_u32 %" #k ", %" #k ", %[w]\n"
#define I_PERM(k) "v_perm_b32 %" #k ", %" #k ", %[w], %[c]\n"
#define I_ANDOR(k) "v_and_or_b32 %" #k ", %" #k ", %[w], %[c]\n"
#define I_OR3(k) "v_or3_b32 %" #k ", %" #k ", %[w], %[c]\n"
#define I_BFE(k) "v_bfe_u32 %" #k ", %" #k ", 8, 8\n"
#define I_LSHLOR(k) "v_lshl_or_b32 %" #k ", %" #k ", 8, %[w]\n"
#define I_CNDMASK(k) "v_cndmask_b32 %" #k ", %" #k ", %[m], vcc\n"
#define I_CNDMASK64(k) "v_cndmask_b32_e64 %" #k ", %" #k ", %[m], s[10:11]\n"
#define I_CNDMASKC(k) "v_cndmask_b32 %" #k ", %[c], %[m], vcc\n"
#define I_ADDC(k) "v_addc_co_u32 %" #k ", vcc, %" #k ", %" #k ", vcc\n"
#define I_CMPSEL32(k) "v_cmp_le_f32 vcc, %" #k ", %[m]\n v_cndmask_b32 %" #k ", %[c], %[m], vcc\n"
#define I_CMPSEL64(k) "v_cmp_le_f32_e64 s[10:11], %" #k ", %[m]\n v_cndmask_b32_e64 %" #k ", %[c], %[m], s[10:11]\n"
#define I_CMPSELN(k) "v_cmp_le_f32 vcc, %" #k ", %[m]\n s_nop 1\n v_cndmask_b32 %" #k ", %[c], %[m], vcc\n"
#define I_CMP(k) "v_cmp_le_f32 vcc, %" #k ", %[m]\n"
#define I_CVTU32(k) "v_cvt_f32_u32 %" #k ", %" #k "\n"
#define I_CVTUB0(k) "v_cvt_f32_ubyte0 %" #k ", %[w]\n"
#define I_CVTSDWA(k) "v_cvt_f32_u32_sdwa %" #k ", %[w] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1\n"
#define I_FMA(k) "v_fma_f32 %" #k ", %" #k ", %[m], %[c]\n"
#define I_FMAC(k) "v_fmac_f32 %" #k ", %[m], %[c]\n"
#define I_MAD(k) "v_mad_u32_u24 %" #k ", %" #k ", %[w], %[w]\n"
#define I_RCP(k) "v_rcp_f32 %" #k ", %" #k "\n"
#define I_MULLEG(k) "v_mul_legacy_f32 %" #k ", %" #k ", %[m]\n"
#define I_CVTPK(k) "v_cvt_pkrtz_f16_f32 %" #k ", %" #k ", %[m]\n"
#define I_ALIGNBIT(k) "v_alignbit_b32 %" #k ", %" #k ", %[w], 31\n"
#define I_BCNT(k) "v_bcnt_u32_b32 %" #k ", %" #k ", %[w]\n"
#define I_XOR(k) "v_xor_b32 %" #k ", %" #k ", %[w]\n"
#define I_SUBREV(k) "v_subrev_f32 %" #k ", %[m], %" #k "\n"
#define I_ADD3(k) "v_add3_u32 %" #k ", %" #k ", %[w], %[w]\n"
#define I_FMAS(k) "v_fma_f32 %" #k ", s10, %" #k ", %[c]\n"
#define I_FMAS2(k) "v_fma_f32 %" #k ", s10, %[m], %[c]\n"
#define I_MULS(k) "v_mul_f32 %" #k ", s10, %" #k "\n"
#define I_MULLIT(k) "v_mul_f32 %" #k ", 0x3f7fbe77, %" #k "\n"
#define I_MULINL(k) "v_mul_f32 %" #k ", 1.0, %" #k "\n"
#define I_ANDLIT(k) "v_and_b32 %" #k ", 0xfffffe00, %" #k "\n"
#define I_ANDINL(k) "v_and_b32 %" #k ", 63, %" #k "\n"
#define I_ANDS(k) "v_and_b32 %" #k ", s10, %" #k "\n"
#define I_FMAINL(k) "v_fma_f32 %" #k ", %" #k ", %[m], 1.0\n"
#define I_FMAABS(k) "v_fma_f32 %" #k ", |%" #k "|, %[m], -%[c]\n"
#define I_ADDE64(k) "v_add_f32_e64 %" #k ", %" #k ", %[m]\n"
#define I_MOVS(k) "v_mov_b32 %" #k ", s10\n"
#define I_MULSDWA(k) "v_mul_f32_sdwa %" #k ", %[m], %[w] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n"
#define I_ADDSDWA(k) "v_add_f32_sdwa %" #k ", %" #k ", %[w] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2\n"
#define I_ORSDWA(k) "v_or_b32_sdwa %" #k ", %[c], %[w] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n"
#define I_FMAMIX(k) "v_fma_mix_f32 %" #k ", %[w], %[m], %" #k " op_sel_hi:[1,0,0]\n"
#define I_MULDEN(k) "v_mul_f32 %" #k ", %[m], %[d]\n"
#define I_MOVLIT(k) "v_mov_b32 %" #k ", 0x3f7fbe77\n"
SURVEY_KERNEL(add, R8(I_ADD)) SURVEY_KERNEL(mul, R8(I_MUL)) SURVEY_KERNEL(sub, R8(I_SUB)) SURVEY_KERNEL(max, R8(I_MAX)) SURVEY_KERNEL(min, R8(I_MIN))
SURVEY_KERNEL(max3, R8(I_MAX3)) SURVEY_KERNEL(med3, R8(I_MED3)) SURVEY_KERNEL(mov, R8(I_MOV)) SURVEY_KERNEL(and_, R8(I_AND)) SURVEY_KERNEL(or_, R8(I_OR))
SURVEY_KERNEL(lshl, R8(I_LSHL)) SURVEY_KERNEL(addu, R8(I_ADDU)) SURVEY_KERNEL(perm, R8(I_PERM)) SURVEY_KERNEL(andor, R8(I_ANDOR)) SURVEY_KERNEL(or3, R8(I_OR3))
SURVEY_KERNEL(bfe, R8(I_BFE)) SURVEY_KERNEL(lshlor, R8(I_LSHLOR)) SURVEY_KERNEL(cndmask, R8(I_CNDMASK)) SURVEY_KERNEL(cmp, R8(I_CMP)) SURVEY_KERNEL(cvtu32, R8(I_CVTU32))
SURVEY_KERNEL(cvtub0, R8(I_CVTUB0)) SURVEY_KERNEL(cvtsdwa, R8(I_CVTSDWA)) SURVEY_KERNEL(fma, R8(I_FMA)) SURVEY_KERNEL(fmac, R8(I_FMAC)) SURVEY_KERNEL(mad24, R8(I_MAD))
SURVEY_KERNEL(rcp, R8(I_RCP)) SURVEY_KERNEL(mulleg, R8(I_MULLEG)) SURVEY_KERNEL(cvtpk, R8(I_CVTPK)) SURVEY_KERNEL(alignbit, R8(I_ALIGNBIT)) SURVEY_KERNEL(bcnt, R8(I_BCNT))
SURVEY_KERNEL(cndmask64, "s_mov_b64 s[10:11], exec\n" R8(I_CNDMASK64)) SURVEY_KERNEL(cndmaskc, R8(I_CNDMASKC)) SURVEY_KERNEL(addc, R8(I_ADDC))
SURVEY_KERNEL(cmpsel32, R8(I_CMPSEL32)) SURVEY_KERNEL(cmpsel64, R8(I_CMPSEL64)) SURVEY_KERNEL(cmpseln, R8(I_CMPSELN))
SURVEY_KERNEL(cmp1sel8v, "v_cmp_le_f32 vcc, %0, %[m]\n" R8(I_CNDMASKC)) SURVEY_KERNEL(cmp1sel8s, "v_cmp_le_f32_e64 s[10:11], %0, %[m]\n" R8(I_CNDMASK64))
SURVEY_KERNEL(xor_, R8(I_XOR)) SURVEY_KERNEL(subrev, R8(I_SUBREV)) SURVEY_KERNEL(add3, R8(I_ADD3))
SURVEY_KERNEL(mullit, R8(I_MULLIT)) SURVEY_KERNEL(mulinl, R8(I_MULINL)) SURVEY_KERNEL(andlit, R8(I_ANDLIT)) SURVEY_KERNEL(andinl, R8(I_ANDINL))
SURVEY_KERNEL(ands, "s_mov_b32 s10, 0x3f7fbe77\n" R8(I_ANDS)) SURVEY_KERNEL(fmainl, R8(I_FMAINL)) SURVEY_KERNEL(fmaabs, R8(I_FMAABS))
SURVEY_KERNEL(adde64, R8(I_ADDE64)) SURVEY_KERNEL(movs, "s_mov_b32 s10, 0x3f7fbe77\n" R8(I_MOVS)) SURVEY_KERNEL(movlit, R8(I_MOVLIT))
SURVEY_KERNEL(mulsdwa, R8(I_MULSDWA)) SURVEY_KERNEL(addsdwa, R8(I_ADDSDWA)) SURVEY_KERNEL(orsdwa, R8(I_ORSDWA)) SURVEY_KERNEL(fmamix, R8(I_FMAMIX)) SURVEY_KERNEL(mulden, R8(I_MULDEN))
SURVEY_KERNEL(fmas, "s_mov_b32 s10, 0x3f7fbe77\n" R8(I_FMAS)) SURVEY_KERNEL(fmas2, "s_mov_b32 s10, 0x3f7fbe77\n" R8(I_FMAS2)) SURVEY_KERNEL(muls, "s_mov_b32 s10, 0x3f7fbe77\n" R8(I_MULS))

typedef void (*survey_fn)(int, float, float *);
static double run_survey(survey_fn fn, int blocks_per_cu, int n_cu, int trips, float *out, double clock_hz) {
    const int grid = blocks_per_cu * n_cu;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(fn, dim3(grid), dim3(256), 0, 0, trips / 8, 1.0f, out);
    CK(hipDeviceSynchronize());
    double best = 1e30;
    for (int rep = 0; rep < 3; rep++) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(fn, dim3(grid), dim3(256), 0, 0, trips, 1.0f, out);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        best = std::min(best, (double)ms);
    }
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    return (double)blocks_per_cu * trips * kUnroll * 8 / (best * 1e-3 * clock_hz);
}

template <int KIND>
static double run(int blocks_per_cu, int n_cu, int trips, float *out, double clock_hz, int instr_per_group) {
    const int grid = blocks_per_cu * n_cu;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_valu<KIND>, dim3(grid), dim3(256), 0, 0, trips / 8, 1.0f, out);      // warm
    CK(hipDeviceSynchronize());
    double best = 1e30;
    for (int rep = 0; rep < 5; rep++) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k_valu<KIND>, dim3(grid), dim3(256), 0, 0, trips, 1.0f, out);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        best = std::min(best, (double)ms);
    }
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    // wave-instructions issued per SIMD: every block is 4 waves (one per SIMD), blocks_per_cu blocks per CU
    const double per_simd = (double)blocks_per_cu * trips * kUnroll * instr_per_group;
    return per_simd / (best * 1e-3 * clock_hz);      // wave-instructions per clock and SIMD
}

int main() {
    int dev = 0; CK(hipSetDevice(dev));
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, dev));
    int khz = 0; CK(hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, dev));
    const int n_cu = prop.multiProcessorCount;
    const double clock_hz = khz * 1e3;
    printf("device %s, %d CUs, shader clock %.0f MHz (hipDeviceAttributeClockRate)\n", prop.name, n_cu, clock_hz / 1e6);
    float *out; CK(hipMalloc(&out, (size_t)8 * n_cu * 256 * 4));
    const char *names[] = {"v_fma_f32", "v_pk_fma_f32", "v_cvt_f32_ubyteN", "v_max3/min3_f32", "wide_test8 mix"};
    printf("%-18s %6s %14s %14s\n", "instruction", "waves", "instr/clk/SIMD", "lanes/clk/SIMD");
    const int ws[] = {1, 2, 4, 6, 8};
    for (int kind = 0; kind < 5; kind++)
        for (int w : ws) {
            const int trips = 4096;
            double r = 0;
            switch (kind) {
                case 0: r = run<0>(w, n_cu, trips, out, clock_hz, 8); break;
                case 1: r = run<1>(w, n_cu, trips, out, clock_hz, 8); break;
                case 2: r = run<2>(w, n_cu, trips, out, clock_hz, 8); break;
                case 3: r = run<3>(w, n_cu, trips, out, clock_hz, 8); break;
                default: r = run<4>(w, n_cu, trips, out, clock_hz, 15); break;
            }
            printf("%-18s %6d %14.4f %14.2f\n", names[kind], w, r, r * 64.0);
        }
    struct { const char *name; survey_fn fn; } sv[] = {
        {"v_mul_f32_sdwa byte (denormal src)", k_sv_mulsdwa}, {"v_add_f32_sdwa byte", k_sv_addsdwa}, {"v_or_b32_sdwa byte", k_sv_orsdwa}, {"v_fma_mix_f32 f16 src0", k_sv_fmamix}, {"v_mul_f32 denormal src", k_sv_mulden},
        {"v_mul_f32 literal", k_sv_mullit}, {"v_mul_f32 inline 1.0", k_sv_mulinl}, {"v_and_b32 literal", k_sv_andlit}, {"v_and_b32 inline 63", k_sv_andinl}, {"v_and_b32 sgpr (x8/9)", k_sv_ands},
        {"v_fma_f32 inline 1.0 src2", k_sv_fmainl}, {"v_fma_f32 abs/neg modifiers", k_sv_fmaabs}, {"v_add_f32_e64 (VOP3 encoding)", k_sv_adde64},
        {"v_mov_b32 sgpr (x8/9)", k_sv_movs}, {"v_mov_b32 literal", k_sv_movlit},
        {"v_fma_f32 sgpr src0 (x8/9)", k_sv_fmas}, {"v_fma_f32 sgpr src0, vgpr consts (x8/9)", k_sv_fmas2}, {"v_mul_f32 sgpr src0 (x8/9)", k_sv_muls}, {"v_add_f32", k_sv_add}, {"v_mul_f32", k_sv_mul}, {"v_sub_f32", k_sv_sub}, {"v_subrev_f32", k_sv_subrev}, {"v_fma_f32", k_sv_fma}, {"v_fmac_f32", k_sv_fmac},
        {"v_mul_legacy_f32", k_sv_mulleg}, {"v_max_f32", k_sv_max}, {"v_min_f32", k_sv_min}, {"v_max3_f32", k_sv_max3}, {"v_med3_f32", k_sv_med3},
        {"v_mov_b32", k_sv_mov}, {"v_and_b32", k_sv_and_}, {"v_or_b32", k_sv_or_}, {"v_xor_b32", k_sv_xor_}, {"v_lshlrev_b32", k_sv_lshl}, {"v_add_u32", k_sv_addu}, {"v_add3_u32", k_sv_add3},
        {"v_perm_b32", k_sv_perm}, {"v_and_or_b32", k_sv_andor}, {"v_or3_b32", k_sv_or3}, {"v_bfe_u32", k_sv_bfe}, {"v_lshl_or_b32", k_sv_lshlor}, {"v_alignbit_b32", k_sv_alignbit},
        {"v_bcnt_u32_b32", k_sv_bcnt}, {"v_mad_u32_u24", k_sv_mad24}, {"v_cndmask_b32", k_sv_cndmask}, {"v_cndmask_b32_e64 sgpr", k_sv_cndmask64}, {"v_cndmask_b32 const srcs", k_sv_cndmaskc}, {"v_addc_co_u32", k_sv_addc}, {"1 cmp vcc + 8 cndmask vcc (x8/9)", k_sv_cmp1sel8v}, {"1 cmp sgpr + 8 cndmask sgpr (x8/9)", k_sv_cmp1sel8s}, {"(cmp vcc + cndmask vcc)/2", k_sv_cmpsel32}, {"(cmp sgpr + cndmask sgpr)/2", k_sv_cmpsel64}, {"(cmp vcc, nop, cndmask)/2", k_sv_cmpseln}, {"v_cmp_le_f32", k_sv_cmp}, {"v_cvt_f32_u32", k_sv_cvtu32},
        {"v_cvt_f32_ubyte0", k_sv_cvtub0}, {"v_cvt_f32_u32 sdwa", k_sv_cvtsdwa}, {"v_cvt_pkrtz_f16_f32", k_sv_cvtpk}, {"v_rcp_f32", k_sv_rcp},
    };
    printf("\nsurvey: wave-instructions per clock and SIMD at 1 / 6 waves per SIMD\n");
    for (auto &e : sv) {
        const double r1 = run_survey(e.fn, 1, n_cu, 2048, out, clock_hz), r6 = run_survey(e.fn, 6, n_cu, 2048, out, clock_hz);
        printf("%-22s %8.4f %8.4f   -> %s\n", e.name, r1, r6, r6 > 0.33 ? "2-cycle class" : r6 > 0.17 ? "4-cycle class" : "slower");
    }
    CK(hipFree(out));
    return 0;
}
