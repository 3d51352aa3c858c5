#!/usr/bin/env python3
"""BVH builds only (for rocprofv3 --kernel-trace): python tools/build_only.py [workload] [bvh] [reps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from computeraytracer_amd import abi
import bench

wl = sys.argv[1] if len(sys.argv) > 1 else "lost_empire_4k"
bvh = {"lbvh": abi.BVH_LBVH, "sah": abi.BVH_SAH_REFINE, "sah-wide": abi.BVH_SAH_REFINE | abi.BVH_WIDE, "lbvh-wide": abi.BVH_WIDE, "direct": abi.BVH_WIDE_DIRECT}[sys.argv[2] if len(sys.argv) > 2 else "sah-wide"]
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
scene_name, W, H, _ = bench.WORKLOADS[wl]
scene = abi.builtin_scene(scene_name, W, H, os.path.join(ROOT, "tests", "golden"))
ctx = abi.Context(0)
ctx.configure(64, 64)
ctx.upload(scene)
ms = [ctx.build(bvh) for _ in range(reps)]
print("build ms:", " ".join(f"{m:.3f}" for m in ms))
ctx.close()
