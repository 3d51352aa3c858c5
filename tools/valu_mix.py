#!/usr/bin/env python3
"""Static VALU instruction mix of the tracer's kernels, by issue class (runs in the build container: hipcc -S, no GPU).

tools/ubench_valu.hip measured two classes on gfx950 (profiles/r03_ubench_valu.txt): v_add/sub/mul/fma/fmac_f32, v_mov_b32,
v_and/or/xor_b32 and v_add_u32 issue every ~2 cycles per SIMD with several waves resident (0.40-0.44 wave-instructions per
clock at 6-8 waves), every other VALU instruction every ~4 (0.24-0.245).  The issue ceiling of a kernel therefore depends
on its mix: ceiling = 1 / (f_fast / r_fast + f_slow / r_slow) wave-instructions per clock and SIMD.  This tool counts the
mix over each kernel's body (static counts: every instruction once, whatever its trip count -- an estimate, said so in
the output) and writes profiles/rNN_valu_mix.json, which bench.py reads for the roofline's `peak`.

usage: tools/valu_mix.py > profiles/r03_valu_mix.json
"""
import json, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
FAST = {"v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_fma_f32", "v_fmac_f32", "v_mul_legacy_f32", "v_mov_b32", "v_and_b32", "v_or_b32",
        "v_xor_b32", "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_not_b32"}
R_FAST, R_SLOW = 0.42, 0.2447      # wave-instructions per clock and SIMD at 6 waves (profiles/r03_ubench_valu.txt: v_fma_f32 0.4197, wide_test8 mix 0.2447)
KERNELS = {"k_wf_extend": "_ZN3crt11k_wf_extendILb0ELb1ELb0ELi0ELb0EEE", "k_wf_primary": "_ZN3crt12k_wf_primaryILi1EEE", "k_wf_shade": "_ZN3crt10k_wf_shadeILi0ELb0EEE"}


def main():
    from computeraytracer_amd import abi
    csrc = os.path.join(ROOT, "computeraytracer_amd", "csrc")
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "trace.s")
        subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-fno-slp-vectorize", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math",
                        "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(csrc, "host"), "-I" + os.path.join(csrc, "hip"), "-I" + os.path.join(csrc, "dist"),
                        "--cuda-device-only", "-S", "-o", out, os.path.join(csrc, "hip", "trace.hip")], check=True, stderr=subprocess.DEVNULL)
        text = open(out).read()
    res = {"code_rev": abi.code_rev(), "r_fast": R_FAST, "r_slow": R_SLOW, "note": "static instruction counts over the kernel body (an estimate of the dynamic mix)", "kernels": {}}
    for name, sym in KERNELS.items():
        m = re.search(r"^" + re.escape(sym) + r"[^\n]*\n(.*?)s_endpgm", text, re.S | re.M)
        if not m:
            continue
        ops = re.findall(r"^\s+(v_[a-z0-9_]+)", m.group(1), re.M)
        ops = [re.sub(r"_(e32|e64|sdwa|dpp)$", "", o) for o in ops]
        fast = sum(1 for o in ops if o in FAST)
        n = len(ops)
        f_fast = fast / n
        ceiling = 1.0 / (f_fast / R_FAST + (1.0 - f_fast) / R_SLOW)
        res["kernels"][name] = {"valu_static": n, "fast_class": fast, "f_fast": round(f_fast, 4), "issue_ceiling_instr_per_clk_simd": round(ceiling, 4),
                                "lanes_per_clk_simd": round(64.0 * ceiling, 2)}
    json.dump(res, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
