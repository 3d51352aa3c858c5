import sys, os
sys.path.insert(0,'/root/repo/tests'); sys.path.insert(0,'/root/repo')
from computeraytracer_amd import abi
from fuzz_scenes import random_soup
c = abi.Context(0)
c.configure(64, 64, tracer=abi.TRACER_WAVEFRONT)
for seed in (1, 2, 7, 10, 14):
    scene = random_soup(seed)
    c.upload(scene)
    cost = {}
    for name, flags in (("lbvh-wide", abi.BVH_LBVH | abi.BVH_WIDE), ("direct", abi.BVH_WIDE_DIRECT)):
        try:
            c.build(flags); cost[name] = round(c.refit(want_cost=True), 2)
        except Exception as e:
            cost[name] = str(e)[:60]
    print(os.environ.get("CRT_LIB_PATH", "libcrt.so")[-18:], seed, len(scene.tris), cost, flush=True)
c.close()
