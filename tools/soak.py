#!/usr/bin/env python3
"""The engine's frame loop for a while: N crt_dispatch calls on one context (camera.iteration counting up), every 500th frame a
crt_image_read; device memory, descriptors and the time per dispatch at the start and at the end.  python tools/soak.py [N]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from computeraytracer_amd import abi
from hip_util import Hip

n = int(sys.argv[1]) if len(sys.argv) > 1 else 6000
hip = Hip()
scene = abi.builtin_scene("cirno", 640, 360, os.path.join(ROOT, "tests", "golden"))
ctx = abi.Context(0)
ctx.configure(640, 360, tracer=abi.TRACER_WAVEFRONT)
ctx.upload(scene)
ctx.build(abi.BVH_SAH_REFINE | abi.BVH_WIDE)
seeds = abi.frame_seeds(0, n + 1)
for i in range(50):
    ctx.dispatch(scene.params_for(i + 1, seeds[i]))
ctx.sync()
free0, fds0 = hip.mem_free(), len(os.listdir("/proc/self/fd"))
times = []
t_last = time.perf_counter()
for i in range(50, n):
    ctx.dispatch(scene.params_for(i + 1, seeds[i]))
    if i % 500 == 499:
        ctx.sync()
        img = ctx.image()
        now = time.perf_counter(); times.append((now - t_last) / 500 * 1e3); t_last = now
        assert img[0, 0, 3] == float(i + 1), (img[0, 0, 3], i + 1)
        print(f"frame {i + 1}: {times[-1]:.3f} ms per dispatch, free {hip.mem_free() >> 20} MiB, fds {len(os.listdir('/proc/self/fd'))}", flush=True)
ctx.sync()
free1, fds1 = hip.mem_free(), len(os.listdir("/proc/self/fd"))
print(f"device memory: {free0 >> 20} -> {free1 >> 20} MiB free; descriptors {fds0} -> {fds1}; ms per dispatch first {times[0]:.3f} last {times[-1]:.3f}")
ctx.close()
sys.exit(0 if (free0 - free1 < (32 << 20) and fds1 - fds0 <= 1 and times[-1] < 1.3 * times[0]) else 1)
