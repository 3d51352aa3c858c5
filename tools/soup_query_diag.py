#!/usr/bin/env python3
"""Diagnose tests/test_gpu_fuzz.py::test_random_soups_of_random_size_on_every_tree for given seeds: which rays differ from the
oracle on which tree, and how the ray meets the triangles involved."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from computeraytracer_amd import abi
from fuzz_scenes import random_soup, soup_rays
import oracle_lib as orc
import test_gpu_fuzz as t

def describe(scene, rays, i, hid):
    if hid == abi.ID_MISS: return "miss"
    k, idx = int(hid) >> 30, int(hid) & 0x3FFFFFFF
    if k != 2: return f"kind {k} index {idx}"
    tr = scene.tris[idx]; e1 = (tr["v1"] - tr["v0"]).astype(np.float64); e2 = (tr["v2"] - tr["v0"]).astype(np.float64); c = np.cross(e1, e2)
    dd = rays["direction"][i].astype(np.float64)
    return (f"tri {idx} sin {np.linalg.norm(c) / (np.linalg.norm(e1) * np.linalg.norm(e2)):.3g} edges {np.linalg.norm(e1):.3g} {np.linalg.norm(e2):.3g} "
            f"dist/size {np.linalg.norm(tr['v0'] - rays['origin'][i]) / max(np.linalg.norm(e1), np.linalg.norm(e2)):.3g} cos(ray, normal) {abs(np.dot(dd, c)) / max(np.linalg.norm(c), 1e-300):.3g}")

ctx = abi.Context(0)
ctx.configure(64, 64, tracer=abi.TRACER_WAVEFRONT)
for seed in map(int, sys.argv[1:]):
    scene = random_soup(seed)
    rays = soup_rays(scene, seed, 8192)
    wf = orc.hit_rays(scene, rays, flags=orc.TRAVERSE_FAST); wb = orc.hit_rays(scene, rays)
    print("seed", seed, "tris", len(scene.tris), "spheres", len(scene.spheres), "quads", len(scene.quads), "oracle fast != brute", int((wf["id"] != wb["id"]).sum()), flush=True)
    ctx.upload(scene)
    for name, flags, _ in t.TREES:
        ctx.build(flags)
        got = ctx.rays_query(rays, abi.QUERY_CLOSEST)
        bad = np.nonzero((got["id"] != wb["id"]) | ((wb["id"] != abi.ID_MISS) & (got["t"].view(np.uint32) != wb["t"].view(np.uint32))))[0]
        print("  ", name, "mismatches vs brute", len(bad))
        for i in bad[:2]:
            print("      ray", i, "gpu", got[i], "->", describe(scene, rays, i, got["id"][i]))
            print("      ray", i, "oracle", wb[i], "->", describe(scene, rays, i, wb["id"][i]))
ctx.close()
