"""Helper of tests/test_gpu_sort.py::test_static_build_through_the_own_sort_gives_the_same_tree: builds the 8-wide SAH tree of a
scene in a fresh interpreter (CRT_BVH_SORT is read once per process) and prints a digest of the tree with the nodes of a
level taken as a set (node numbers inside a level follow an atomic counter)."""
import hashlib
import os
import sys

import numpy as np

from computeraytracer_amd import abi


def main():
    name = sys.argv[1]
    golden = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    scene = abi.builtin_scene(name, 64, 48, golden)
    with abi.Context(0) as ctx:
        ctx.configure(64, 48, tracer=abi.TRACER_WAVEFRONT)
        ctx.upload(scene)
        ctx.build(abi.BVH_SAH_REFINE | abi.BVH_WIDE)
        wide, levels = ctx.bvh_wide()
        a = np.asarray(wide)[:, :16].copy()
        a[:, 3] &= 0xFF
        a = a[np.lexsort(a.T[::-1])]
        ctx.render(scene.params, 1, abi.frame_seeds(0, 1)); ctx.sync()
        img = ctx.image()
    print("tree", len(wide), levels, hashlib.sha256(a.tobytes()).hexdigest(), hashlib.sha256(img.tobytes()).hexdigest())


if __name__ == "__main__":
    main()
