"""Random scenes: every primitive kind, every material kind, a texture, lens blur, either background mode; the GPU (both
tracers, every tree format) against the oracle, bit for bit.  The built-in scenes exercise the paths their authors thought
of; these exercise combinations nobody chose (spheres of negative radius next to slivers, lights behind glass, quads seen
edge-on, a camera inside a sphere)."""
import numpy as np
import pytest

from computeraytracer_amd import abi
from fuzz_scenes import random_scene
import oracle_lib as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = abi.Context(0)
    yield c
    c.close()


TREES = [("lbvh", abi.BVH_LBVH, True), ("sah", abi.BVH_SAH_REFINE, True), ("collapse", abi.BVH_LEAF_COLLAPSE, True), ("lbvh-wide", abi.BVH_LBVH | abi.BVH_WIDE, True),
         ("sah-wide", abi.BVH_SAH_REFINE | abi.BVH_WIDE, True), ("direct", abi.BVH_WIDE_DIRECT, False)]


@pytest.mark.parametrize("seed", range(8))
def test_random_scenes_bit_exact_on_every_tree_and_tracer(ctx, seed):
    w, h, n = 72, 48, 3
    scene = random_scene(seed, w, h)
    seeds = abi.frame_seeds(0, n)
    ps = [scene.params_for(i + 1, seeds[i]) for i in range(n)]
    ref_img, ref_ids, _ = orc.render(scene, ps, w, h, flags=orc.TRAVERSE_BRUTE)
    for name, flags, megakernel_too in TREES:
        for tracer in ([abi.TRACER_WAVEFRONT, abi.TRACER_MEGAKERNEL] if megakernel_too else [abi.TRACER_WAVEFRONT]):
            ctx.configure(w, h, tracer=tracer)
            ctx.upload(scene)
            ctx.build(flags)
            ctx.render(ps[0], 1, seeds); ctx.sync()
            img, ids = ctx.image(), ctx.hit_ids()
            assert np.array_equal(ids, ref_ids), (name, tracer, int((ids != ref_ids).sum()))
            assert np.array_equal(img.view(np.uint32), ref_img.view(np.uint32)), (name, tracer, int((img.view(np.uint32) != ref_img.view(np.uint32)).any(axis=2).sum()))
    # closest-hit queries on the last tree: rays from the camera position in random directions
    rs = np.random.RandomState(seed)
    rays = np.zeros(4096, abi.ray_dtype)
    rays["origin"] = scene.params["camera"]["position"][0]
    d = rs.standard_normal((4096, 3)).astype(np.float32)
    rays["direction"] = d / np.linalg.norm(d, axis=1, keepdims=True).astype(np.float32)
    rays["t_max"] = np.inf
    got = ctx.rays_query(rays, abi.QUERY_CLOSEST)
    want = orc.hit_rays(scene, rays)
    hit = want["id"] != abi.ID_MISS                              # crt_ray_hit.t is only meaningful for a hit (crt/types.h)
    assert np.array_equal(got["id"], want["id"]) and np.array_equal(got["t"][hit].view(np.uint32), want["t"][hit].view(np.uint32))
