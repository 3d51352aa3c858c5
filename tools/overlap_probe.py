#!/usr/bin/env python3
"""Does the GPU overlap the launch sequences of two contexts (shade of one under extend of the other)?
python tools/overlap_probe.py [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from computeraytracer_amd import abi
import bench

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 32
scene_name, W, H, _ = bench.WORKLOADS["lost_empire_4k"]
scene = abi.builtin_scene(scene_name, W, H, os.path.join(ROOT, "tests", "golden"))
bvh = abi.BVH_SAH_REFINE | abi.BVH_WIDE
seeds = abi.frame_seeds(0, 4 * steps + 1)
ctxs = []
for k in range(2):
    c = abi.Context(0)
    c.configure(W, H, tracer=abi.TRACER_WAVEFRONT)
    c.upload(scene); c.build(bvh)
    c.render(scene.params, 1, seeds[:16]); c.sync()
    ctxs.append(c)

def timed(fn):
    for c in ctxs: c.sync()
    t0 = time.perf_counter(); fn()
    for c in ctxs: c.sync()
    return time.perf_counter() - t0

def one():
    ctxs[0].render(scene.params, 1, seeds[:2 * steps])
def two():
    ctxs[0].render(scene.params, 1, seeds[:steps]); ctxs[1].render(scene.params, 1, seeds[:steps])
for rep in range(3):
    t1 = timed(one); t2 = timed(two)
    print(f"one context {2 * steps} steps: {t1 * 1e3 / (2 * steps):.3f} ms/step   two contexts {steps} steps each: {t2 * 1e3 / (2 * steps):.3f} ms/step", flush=True)
for c in ctxs: c.close()
