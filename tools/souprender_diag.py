#!/usr/bin/env python3
"""Diagnose tests/test_gpu_fuzz.py::test_random_soups_rendered_animated_and_refitted for one seed."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from computeraytracer_amd import abi
from fuzz_scenes import random_soup
import oracle_lib as orc
seed = int(sys.argv[1])
rs = np.random.RandomState(4000 + seed)
scene = random_soup(seed)
if len(scene.tris) > 30000:
    scene = abi.HostScene(spheres=scene.spheres, quads=scene.quads, tris=scene.tris[:30000], materials=scene.materials, textures=scene.textures, params=scene.params)
tris = scene.tris.copy()
tris["materialIndex"] = rs.randint(0, len(scene.materials), len(tris))
tris["u"] = rs.uniform(-1, 2, (len(tris), 3)).astype(np.float32); tris["v"] = rs.uniform(-1, 2, (len(tris), 3)).astype(np.float32)
w, h = 48, 32
k = int(rs.randint(0, len(tris)))
at = tris["v0"][k]
size = float(max(np.linalg.norm(tris["v1"][k] - tris["v0"][k]), 1e-3))
d = rs.standard_normal(3); d /= np.linalg.norm(d)
pos = at + d * size * float(np.exp(rs.uniform(np.log(5.0), np.log(200.0))))
params = scene.params.copy()
params["camera"] = abi.camera(tuple(pos), tuple(at), float(rs.uniform(20, 90)), w / h, 0.0, 10.0)
scene = abi.HostScene(spheres=scene.spheres, quads=scene.quads, tris=tris, materials=scene.materials, textures=scene.textures, params=params)
seeds = abi.frame_seeds(0, 2)
ps = [scene.params_for(i + 1, seeds[i]) for i in range(2)]
ctx = abi.Context(0)
ctx.configure(w, h, tracer=abi.TRACER_WAVEFRONT)
ctx.upload(scene)
ctx.build(abi.BVH_SAH_REFINE | abi.BVH_WIDE)
ctx.animate(float(seed) + 0.5, 0.25 * size)
moved_tris = ctx.tris_read()
moved = abi.HostScene(spheres=scene.spheres, quads=scene.quads, tris=moved_tris, materials=scene.materials, textures=scene.textures, params=params)
print("tris", len(tris), "size", size, "max |dy|", float(np.abs(moved_tris["v0"][:, 1] - tris["v0"][:, 1]).max()))
for depth_flag, nm in ((orc.TRAVERSE_FAST, "fast"), (orc.TRAVERSE_BRUTE, "brute")):
    if nm == "brute" and len(tris) > 3000: continue
    ref2, ids2, _ = orc.render(moved, ps, w, h, flags=depth_flag)
    for how in ("refit", "rebuild-direct", "rebuild-sahwide", "rebuild-lbvh"):
        if how == "refit": ctx.refit()
        elif how == "rebuild-direct": ctx.build(abi.BVH_WIDE_DIRECT)
        elif how == "rebuild-sahwide": ctx.build(abi.BVH_SAH_REFINE | abi.BVH_WIDE)
        else: ctx.build(abi.BVH_LBVH)
        ctx.render(ps[0], 1, seeds); ctx.sync()
        img, ids = ctx.image(), ctx.hit_ids()
        bad = (img.view(np.uint32) != ref2.view(np.uint32)).any(axis=2)
        print(nm, how, "id mismatches", int((ids != ids2).sum()), "pixel mismatches", int(bad.sum()))
        if bad.any():
            y, x = np.argwhere(bad)[0]
            print("   first", y, x, "gpu", img[y, x], "oracle", ref2[y, x], "id", ids[y, x], ids2[y, x])
ctx.close()

# which kernel form disagrees?  (same tree, static scene this time)
ref, rids, _ = orc.render(scene, ps, w, h, flags=orc.TRAVERSE_FAST)
for nm, tracer, fl in (("megakernel", abi.TRACER_MEGAKERNEL, 0), ("per-bounce", abi.TRACER_WAVEFRONT, abi.FLAG_PER_BOUNCE_KERNELS), ("path", abi.TRACER_WAVEFRONT, abi.FLAG_PATH_KERNEL),
                       ("persistent", abi.TRACER_WAVEFRONT, abi.FLAG_PATH_KERNEL_PERSISTENT)):
    c2 = abi.Context(0)
    c2.configure(w, h, tracer=tracer, flags=fl)
    c2.upload(scene)
    c2.build(abi.BVH_SAH_REFINE | abi.BVH_WIDE)
    for how in ("render2", "dispatch+dispatch"):
        if how == "render2":
            c2.render(ps[0], 1, seeds)
        else:
            c2.dispatch(ps[0]); c2.dispatch(ps[1])
        c2.sync()
        img, ids = c2.image(), c2.hit_ids()
        bad = (img.view(np.uint32) != ref.view(np.uint32)).any(axis=2)
        print(nm, how, "id mismatches", int((ids != rids).sum()), "pixel mismatches", int(bad.sum()), "of which sky", int((bad & (rids == abi.ID_MISS)).sum()))
    c2.close()
