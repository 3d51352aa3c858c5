#!/usr/bin/env python3
"""Time of unbatched single-sample dispatches (the engine's per-frame pattern): python tools/dispatch_time.py [workload] [n]
Honours CRT_WF_TAIL_BOUNCE etc.; prints ms per dispatch after a warm-up that lets the context settle its kernel choice."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from computeraytracer_amd import abi
import bench

wl = sys.argv[1] if len(sys.argv) > 1 else "cirno_1080p"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
scene_name, W, H, _ = bench.WORKLOADS[wl]
scene = abi.builtin_scene(scene_name, W, H, os.path.join(ROOT, "tests", "golden"))
ctx = abi.Context(0)
ctx.configure(W, H, tracer=abi.TRACER_WAVEFRONT, flags=int(os.environ.get("CRT_CFG_FLAGS", "0")))
ctx.upload(scene)
ctx.build(abi.BVH_SAH_REFINE | abi.BVH_WIDE)
seeds = abi.frame_seeds(0, 20 + n)
for i in range(20):
    ctx.dispatch(scene.params_for(i + 1, seeds[i]))
ctx.sync()
t0 = time.perf_counter()
for i in range(20, 20 + n):
    ctx.dispatch(scene.params_for(i + 1, seeds[i]))
ctx.sync()
print(f"{wl}: {(time.perf_counter() - t0) / n * 1e3:.3f} ms per dispatch")
ctx.close()
