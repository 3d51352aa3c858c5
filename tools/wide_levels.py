import os, sys
sys.path.insert(0, os.getcwd())
from computeraytracer_amd import abi
for name, flags in [("voxel_city", abi.BVH_SAH_REFINE | abi.BVH_WIDE), ("voxel_city", abi.BVH_WIDE), ("cirno", abi.BVH_SAH_REFINE | abi.BVH_WIDE), ("monkey", abi.BVH_SAH_REFINE | abi.BVH_WIDE), ("book1", abi.BVH_SAH_REFINE | abi.BVH_WIDE), ("grid:10", abi.BVH_WIDE)]:
    sc = abi.builtin_scene(name, 64, 64, os.path.join(os.getcwd(), "tests", "golden"))
    with abi.Context(0) as c:
        c.configure(64, 64); c.upload(sc); c.build(flags)
        w, lv = c.bvh_wide()
        print(name, flags, "nodes", len(w), "levels", lv)
