#!/usr/bin/env python3
"""Diagnose one seed of tests/test_gpu_fuzz.py's ray-query check: python tools/fuzz_diag.py SEED"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from computeraytracer_amd import abi
from fuzz_scenes import random_scene
import oracle_lib as orc
import test_gpu_fuzz as t

seed = int(sys.argv[1])
scene = random_scene(seed, 72, 48)
print("scene", len(scene.spheres), len(scene.quads), len(scene.tris))
rs = np.random.RandomState(seed)
rays = np.zeros(4096, abi.ray_dtype)
rays["origin"] = scene.params["camera"]["position"][0]
d = rs.standard_normal((4096, 3)).astype(np.float32)
rays["direction"] = d / np.linalg.norm(d, axis=1, keepdims=True).astype(np.float32)
rays["t_max"] = np.inf
want = orc.hit_rays(scene, rays)
wantf = orc.hit_rays(scene, rays, flags=orc.TRAVERSE_FAST)
print("oracle brute vs fast differ:", int((want["id"] != wantf["id"]).sum()))
ctx = abi.Context(0)
ctx.configure(72, 48, tracer=abi.TRACER_WAVEFRONT)
ctx.upload(scene)
for name, flags, _ in t.TREES:
    ctx.build(flags)
    got = ctx.rays_query(rays, abi.QUERY_CLOSEST)
    bad = np.nonzero((got["id"] != want["id"]) | (got["t"].view(np.uint32) != want["t"].view(np.uint32)))[0]
    print(name, "mismatches", len(bad))
    for i in bad[:4]:
        print("   ray", i, rays[i], "gpu", got[i], "oracle", want[i])
        k, idx = int(want["id"][i]) >> 30, int(want["id"][i]) & 0x3FFFFFFF
        if want["id"][i] != 0xFFFFFFFF:
            print("   oracle prim", k, idx, (scene.spheres, scene.quads, scene.tris)[k][idx])
        k, idx = int(got["id"][i]) >> 30, int(got["id"][i]) & 0x3FFFFFFF
        if got["id"][i] != 0xFFFFFFFF:
            print("   gpu prim", k, idx, (scene.spheres, scene.quads, scene.tris)[k][idx])
ctx.close()
