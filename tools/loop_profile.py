#!/usr/bin/env python3
"""Divergence profile of k_wf_extend per bounce (instrumented launch): python tools/loop_profile.py [workload] [bvh]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from computeraytracer_amd import abi
import bench

wl = sys.argv[1] if len(sys.argv) > 1 else "lost_empire_4k"
bvh = {"lbvh": abi.BVH_LBVH, "sah": abi.BVH_SAH_REFINE, "sah-wide": abi.BVH_SAH_REFINE | abi.BVH_WIDE, "lbvh-wide": abi.BVH_WIDE}[sys.argv[2] if len(sys.argv) > 2 else "sah-wide"]
scene_name, W, H, _ = bench.WORKLOADS[wl]
scene = abi.builtin_scene(scene_name, W, H, os.path.join(ROOT, "tests", "golden"))
ctx = abi.Context(0)
ctx.configure(W, H, tracer=abi.TRACER_WAVEFRONT)
ctx.upload(scene); ctx.build(bvh)
seeds = abi.frame_seeds(0, 2)
ctx.render(scene.params, 1, seeds[:1]); ctx.sync()
ctx.stats_reset(); ctx.set_instrumented(True)
ctx.render(scene.params, 2, seeds[1:]); ctx.sync()
st = ctx.stats(); c = ctx.loop_counters().astype(np.float64)
print(f"rays {st.rays} V {st.node_visits / st.rays:.2f} T {st.prim_tests / st.rays:.2f}")
print("bounce   iters(M)  inner%  lanes/inner  leaf%  lanes/leaf  refill%  idle%")
for b in range(16):
    it, inn, inl, lf, lfl, rf = c[b]
    if it == 0: continue
    print(f"{b:6d} {it / 1e6:10.3f} {100 * inn / it:7.1f} {inl / max(inn, 1):12.1f} {100 * lf / it:6.1f} {lfl / max(lf, 1):11.1f} {100 * rf / it:8.1f} {100 * (it - inn - lf) / it:6.1f}")
t = c.sum(axis=0)
print(f"   all {t[0] / 1e6:10.3f} {100 * t[1] / t[0]:7.1f} {t[2] / t[1]:12.1f} {100 * t[3] / t[0]:6.1f} {t[4] / t[3]:11.1f} {100 * t[5] / t[0]:8.1f} {100 * (t[0] - t[1] - t[3]) / t[0]:6.1f}")
ctx.close()
