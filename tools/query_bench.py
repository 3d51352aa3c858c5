#!/usr/bin/env python3
"""Throughput of crt_rays_query (closest / any hit) on the headline scene with device-resident rays and results."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch                                  # before libcrt.so ("Sharing a process with PyTorch-ROCm")
import ctypes as C
from computeraytracer_amd import abi

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 24
scene = abi.builtin_scene("voxel_city", 64, 64, os.path.join(ROOT, "tests", "golden"))
rs = np.random.RandomState(1)
centres = ((scene.tris["v0"] + scene.tris["v1"] + scene.tris["v2"]) / 3.0).astype(np.float32)
a = centres[rs.randint(0, len(centres), n)] + rs.standard_normal((n, 3)).astype(np.float32) * 6 + np.float32([0, 8, 0])
b = centres[rs.randint(0, len(centres), n)]
rays = np.zeros(n, abi.ray_dtype)
d = b - a; dist = np.linalg.norm(d, axis=1).astype(np.float32)
rays["origin"], rays["direction"], rays["t_max"] = a, d / dist[:, None], dist * np.float32(0.999)      # shadow rays towards surface points
dev = torch.device("cuda", 0)
rays_d = torch.from_numpy(rays.view(np.uint8).reshape(n, 32)).to(dev)
hits_d = torch.empty((n, 2), dtype=torch.int32, device=dev)
with abi.Context(0) as c:
    c.configure(64, 64); c.upload(scene); c.build(abi.BVH_SAH_REFINE | abi.BVH_WIDE)
    for mode, label in ((abi.QUERY_CLOSEST, "closest"), (abi.QUERY_ANY, "any")):
        for rep in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            rc = abi.lib.crt_rays_query(c._h, C.c_void_p(rays_d.data_ptr()), n, mode, C.c_void_p(hits_d.data_ptr()))
            dt = time.perf_counter() - t0
            assert rc == 0
        ids = hits_d[:, 1].cpu().numpy().view(np.uint32)
        print(f"{label:8s}: {n / dt / 1e6:8.1f} Mrays/s ({dt * 1e3:.2f} ms for {n} rays, {100.0 * (ids != abi.ID_MISS).mean():.1f} % hit)", flush=True)
