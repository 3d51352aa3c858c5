#!/usr/bin/env python3
"""SAH cost of the device's pair BVH against the restated reference builder's tree (full-sweep SAH, one primitive per
leaf) on the same scene: python tools/sah_cost.py [scene]   (needs the GPU for the device tree)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from computeraytracer_amd import abi
import oracle_lib as orc

name = sys.argv[1] if len(sys.argv) > 1 else "voxel_city"
scene = abi.builtin_scene(name, 256, 144, os.path.join(ROOT, "tests", "golden"))

def area(lo, hi):
    d = np.maximum(hi.astype(np.float64) - lo.astype(np.float64), 0.0)
    return d[:, 0] * d[:, 1] + d[:, 1] * d[:, 2] + d[:, 0] * d[:, 2]

ref, *_ = orc.ref_bvh(scene)
ra = area(ref["min"], ref["max"])
inner = ref["numChildren"] == 0
root = ra[0]
print(f"{name}: {scene.n_prims} primitives")
print(f"reference builder : inner {inner.sum():8d}  sum(A_inner)/A_root {ra[inner].sum() / root:10.2f}  sum(A_leaf * n)/A_root {(ra[~inner] * ref['numChildren'][~inner]).sum() / root:9.2f}")
with abi.Context(0) as c:
    c.configure(64, 64)
    c.upload(scene)
    for label, flags in (("LBVH", abi.BVH_LBVH), ("LBVH + SAH refine", abi.BVH_SAH_REFINE)):
        c.build(flags)
        nodes, *_ = c.bvh()
        a = area(nodes["lo"], nodes["hi"])
        cnt = nodes["count"] & 0x0FFFFFFF
        is_inner = nodes["count"] == 0
        pad = (nodes["count"] != 0) & (cnt == 0)
        # the root box is not stored (pair 0 holds its two children): union of the two
        rlo = np.minimum(nodes["lo"][0], nodes["lo"][1]); rhi = np.maximum(nodes["hi"][0], nodes["hi"][1])
        r = area(rlo[None], rhi[None])[0]
        print(f"{label:18s}: inner {is_inner.sum() + 1:8d}  sum(A_inner)/A_root {(a[is_inner].sum() + r) / r:10.2f}  sum(A_leaf * n)/A_root {(a[~is_inner & ~pad] * cnt[~is_inner & ~pad]).sum() / r:9.2f}")
