"""How far the exhaustive test and a tree can disagree on near-degenerate triangles (CPU oracle, brute force against its own
tree; rays aimed along the slivers): python tools/sliver_exp.py.  See DESIGN.md, section 2 (closest-hit semantics)."""
import sys
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, ROOT)
import numpy as np
from computeraytracer_amd import abi
from fuzz_scenes import random_scene
import oracle_lib as orc
base = random_scene(3, 64, 64)
for eps in (0.0, 1e-7, 1e-6, 1e-5, 1e-4, 1e-3):
    tot_bad = 0; tot_hits = 0
    for rep in range(6):
        rs = np.random.RandomState(100 + rep)
        n = 64
        tris = np.zeros(n, abi.tri_dtype)
        a = rs.uniform(-2, 2, (n, 3)).astype(np.float32)
        b = (a + rs.uniform(-1.5, 1.5, a.shape)).astype(np.float32)
        c = (b + (rs.standard_normal(a.shape) * eps).astype(np.float32)).astype(np.float32)
        tris["v0"], tris["v1"], tris["v2"] = a, b, c
        scene = abi.HostScene(spheres=base.spheres[:0], quads=base.quads[:0], tris=tris, materials=base.materials, textures=base.textures, params=base.params)
        k = np.repeat(np.arange(n), 1024)
        p0, p1 = a[k], b[k]
        s = rs.uniform(-1.5, 2.5, (len(k), 1)).astype(np.float32)
        target = (p0 + (p1 - p0) * s + rs.standard_normal((len(k), 3)).astype(np.float32) * np.float32(max(eps, 1e-6))).astype(np.float32)
        origin = (target + rs.standard_normal((len(k), 3)).astype(np.float32) * np.float32(3.0)).astype(np.float32)
        along = rs.randint(0, 2, len(k)).astype(bool)
        origin[along] = (p0[along] - (p1[along] - p0[along]) * np.float32(2.0) + rs.standard_normal((int(along.sum()), 3)).astype(np.float32) * np.float32(max(10 * eps, 1e-5))).astype(np.float32)
        d = target - origin
        rays = np.zeros(len(k), abi.ray_dtype)
        rays["origin"] = origin
        rays["direction"] = d / np.maximum(np.linalg.norm(d, axis=1, keepdims=True), 1e-20).astype(np.float32)
        rays["t_max"] = np.inf
        wb = orc.hit_rays(scene, rays)
        wf = orc.hit_rays(scene, rays, flags=orc.TRAVERSE_FAST)
        tot_bad += int((wb["id"] != wf["id"]).sum()); tot_hits += int((wb["id"] != abi.ID_MISS).sum())
    print(f"eps {eps:g}: brute hits {tot_hits}, brute != tree on {tot_bad} of {6*64*1024} rays", flush=True)
