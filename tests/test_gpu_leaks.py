"""Device memory and file descriptors over many context lifetimes: what a context allocates it gives back."""
import os

import numpy as np
import pytest

from computeraytracer_amd import abi
from fuzz_scenes import random_scene, random_soup
from hip_util import Hip

pytestmark = pytest.mark.gpu


def _fds():
    return len(os.listdir("/proc/self/fd"))


def test_contexts_give_back_their_device_memory_and_descriptors():
    hip = Hip()
    TREES = [abi.BVH_LBVH, abi.BVH_SAH_REFINE, abi.BVH_LEAF_COLLAPSE, abi.BVH_WIDE_DIRECT, abi.BVH_LBVH | abi.BVH_WIDE, abi.BVH_SAH_REFINE | abi.BVH_WIDE]      # (the last one is what the refit below needs)

    def cycle(k):
        rs = np.random.RandomState(k)
        with abi.Context(0) as ctx:
            for rep in range(3):
                w, h = int(rs.randint(16, 200)), int(rs.randint(16, 120))
                scene = random_soup(400 + 3 * k + rep) if rep % 2 else random_scene(7000 + 3 * k + rep, w, h)
                tracer = abi.TRACER_WAVEFRONT if rep != 1 else abi.TRACER_MEGAKERNEL
                ctx.configure(w, h, tracer=tracer)
                ctx.upload(scene)
                for flags in TREES:
                    if tracer == abi.TRACER_MEGAKERNEL and flags == abi.BVH_WIDE_DIRECT:
                        continue
                    ctx.build(flags)
                    seeds = abi.frame_seeds(0, 2)
                    ctx.render(scene.params_for(1, seeds[0]), 1, seeds)
                ctx.sync()
                if tracer == abi.TRACER_WAVEFRONT:
                    rays = np.zeros(1000, abi.ray_dtype); rays["direction"][:, 2] = 1.0; rays["t_max"] = np.inf
                    ctx.rays_query(rays, abi.QUERY_CLOSEST)
                    if len(scene.tris):
                        ctx.animate(1.0, 0.1); ctx.refit(); ctx.sync()
                fd, _, _ = ctx.image_export_fd(want_fd=True)
                os.close(fd)

    cycle(0)                                   # first use: the runtime's own pools, code objects, the no-alloc scratch
    hip.lib.hipDeviceSynchronize()
    free0, fds0 = hip.mem_free(), _fds()
    for k in range(1, 13):
        cycle(k)
    hip.lib.hipDeviceSynchronize()
    free1, fds1 = hip.mem_free(), _fds()
    assert free0 - free1 < 64 << 20, (free0, free1)        # (the runtime may keep a pool or two; a leaked context would be hundreds of MiB)
    assert fds1 - fds0 <= 2, (fds0, fds1)
