#!/usr/bin/env python3
"""Every tree format at the sizes where the builders change gear (workgroup, tile, tier and sort boundaries), closest hits
against the oracle: python tools/size_sweep.py [big]   (big: also 2^20 - 1 .. 2^20 + 1 and 1.2 M, where the key sort changes)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from computeraytracer_amd import abi
from fuzz_scenes import random_scene, soup_rays
import oracle_lib as orc
import test_gpu_fuzz as t

def soup_of(n_t, seed, mixed):
    rs = np.random.RandomState(seed)
    n_c = 12
    centres = rs.uniform(-10, 10, (n_c, 3)).astype(np.float32)
    k = rs.randint(0, n_c, n_t)
    c = (centres[k] + rs.standard_normal((n_t, 3)).astype(np.float32) * np.float32(1.5)).astype(np.float32)
    e = (rs.uniform(-1, 1, (2, n_t, 3)) * 0.08).astype(np.float32)
    tris = np.zeros(n_t, abi.tri_dtype)
    tris["v0"], tris["v1"], tris["v2"] = c, c + e[0], c + e[1]
    base = random_scene(1, 64, 64)
    spheres = base.spheres[:3] if mixed else base.spheres[:0]
    quads = base.quads[:2] if mixed else base.quads[:0]
    return abi.HostScene(spheres=spheres, quads=quads, tris=tris, materials=base.materials, textures=base.textures, params=base.params)

sizes = [1, 2, 3, 4, 5, 7, 8, 9, 15, 16, 17, 31, 32, 33, 63, 64, 65, 127, 128, 129, 255, 256, 257, 511, 512, 513, 1023, 1024, 1025, 2047, 2048, 2049,
         4095, 4096, 4097, 8191, 8192, 8193, 16383, 16384, 16385, 65535, 65536, 65537, 131071, 131072, 131073]
if len(sys.argv) > 1 and sys.argv[1] == "big":
    sizes = [(1 << 20) - 1, 1 << 20, (1 << 20) + 1, 1200000]
ctx = abi.Context(0)
ctx.configure(64, 64, tracer=abi.TRACER_WAVEFRONT)
failures = 0
for n_t in sizes:
    for mixed in (False, True):
        scene = soup_of(n_t, n_t, mixed)
        rays = soup_rays(scene, n_t, 4096)
        want = orc.hit_rays(scene, rays, flags=orc.TRAVERSE_FAST)
        hit = want["id"] != abi.ID_MISS
        ctx.upload(scene)
        line = f"n_tris {n_t:8d} mixed {int(mixed)} hits {int(hit.sum()):5d}:"
        for name, flags, _ in t.TREES:
            try:
                ms = ctx.build(flags)
                got = ctx.rays_query(rays, abi.QUERY_CLOSEST)
                bad = int(((got["id"] != want["id"]) | (hit & (got["t"].view(np.uint32) != want["t"].view(np.uint32)))).sum())
                line += f" {name} {'ok' if bad == 0 else 'MISMATCH ' + str(bad)}"
                failures += bad != 0
            except abi.CrtError as e:
                line += f" {name} ERROR {str(e)[:80]}"
                failures += 1
        print(line, flush=True)
print("failures", failures)
ctx.close()
sys.exit(1 if failures else 0)
