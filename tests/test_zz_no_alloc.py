"""crt_rays_query_noalloc (the entry point that allocates nothing) against crt_rays_query, through the C ABI.

Collected last on purpose (file name): the round's end runs pytest -x."""
import numpy as np
import pytest

from computeraytracer_amd import abi

pytestmark = pytest.mark.gpu


from hip_util import Hip as _Hip


def _rays(scene, n, seed):
    rs = np.random.RandomState(seed)
    centres = np.concatenate([scene.spheres["center"], (scene.tris["v0"] + scene.tris["v1"] + scene.tris["v2"]) / 3.0]).astype(np.float32)
    a = centres[rs.randint(0, len(centres), n)] + rs.standard_normal((n, 3)).astype(np.float32) * np.float32(0.5)
    b = centres[rs.randint(0, len(centres), n)] + rs.standard_normal((n, 3)).astype(np.float32) * np.float32(0.5)
    rays = np.zeros(n, abi.ray_dtype)
    rays["origin"] = a
    d = (b - a).astype(np.float64)
    rays["direction"] = (d / np.maximum(np.linalg.norm(d, axis=1), 1e-30)[:, None]).astype(np.float32)
    rays["t_max"] = np.where(rs.uniform(size=n) < 0.5, np.inf, np.linalg.norm(b - a, axis=1)).astype(np.float32)
    return rays


@pytest.mark.parametrize("bvh", [abi.BVH_SAH_REFINE | abi.BVH_WIDE, abi.BVH_LBVH], ids=["sah-wide", "lbvh-pairs"])
def test_noalloc_query_equals_the_allocating_one(golden_dir, bvh):
    hip = _Hip()
    n_max = 131072
    d_rays, d_hits = hip.malloc(n_max * 32), hip.malloc(n_max * 8)
    try:
        scene = abi.builtin_scene("cirno", 64, 64, golden_dir)
        with abi.Context(0) as ctx:
            ctx.configure(64, 64)
            ctx.upload(scene)
            ctx.build(bvh)
            # two calls in a row, the second smaller (the scratch keeps the first call's contents), each in both modes;
            # then the largest size the entry point takes and a tiny one
            for n, seed in [(100_000, 3), (4_097, 4), (n_max, 5), (1, 6)]:
                rays = _rays(scene, n, seed)
                for mode in (abi.QUERY_CLOSEST, abi.QUERY_ANY):
                    want = ctx.rays_query(rays, mode)
                    hip.h2d(d_rays, rays)
                    hip.memset(d_hits, 0xA5, n_max * 8)
                    ctx.rays_query_noalloc(d_rays, n, mode, d_hits)
                    ctx.sync()
                    got = np.zeros(n_max, abi.ray_hit_dtype)
                    hip.d2h(got, d_hits)
                    assert (got["id"][n:] == 0xA5A5A5A5).all()                     # nothing written past n_rays
                    got = got[:n]
                    if mode == abi.QUERY_CLOSEST:
                        assert np.array_equal(got["id"], want["id"])
                        hit = want["id"] != abi.ID_MISS
                        assert 0.05 < hit.mean() < 0.99 or n == 1
                        assert np.array_equal(got["t"][hit].view(np.uint32), want["t"][hit].view(np.uint32))
                    else:
                        assert np.array_equal(got["id"] != abi.ID_MISS, want["id"] != abi.ID_MISS)
            # the headline property: a series of calls leaves the device's free memory where it was
            rays = _rays(scene, 50_000, 7)
            hip.h2d(d_rays, rays)
            ctx.sync()
            free0 = hip.mem_free()
            for k in range(6):
                ctx.rays_query_noalloc(d_rays, 50_000 - 1000 * k, abi.QUERY_CLOSEST if k % 2 else abi.QUERY_ANY, d_hits)
            assert hip.mem_free() == free0                    # while the calls are still in flight ...
            ctx.sync()
            assert hip.mem_free() == free0                    # ... and after
            # buffers shorter than n_rays records are refused instead of faulting on the GPU
            small_r, small_h = hip.malloc(1000 * 32), hip.malloc(1000 * 8)
            try:
                with pytest.raises(abi.CrtError, match="fewer than n_rays rays"):
                    ctx.rays_query_noalloc(small_r, 1001, abi.QUERY_CLOSEST, d_hits)
                with pytest.raises(abi.CrtError, match="fewer than n_rays hit"):
                    ctx.rays_query_noalloc(d_rays, 1001, abi.QUERY_CLOSEST, small_h)
                with pytest.raises(abi.CrtError, match="fewer than n_rays rays"):
                    ctx.rays_query_noalloc(small_r + 32 * 10, 991, abi.QUERY_CLOSEST, d_hits)       # an interior pointer: counted from there
                hip.h2d(small_r, rays[:1000])
                ctx.rays_query_noalloc(small_r, 1000, abi.QUERY_CLOSEST, small_h); ctx.sync()        # exactly full is fine
            finally:
                hip.free(small_r); hip.free(small_h)
            # errors: too many rays, host memory, null, bad mode; zero rays is a no-op
            with pytest.raises(abi.CrtError, match="131072"):
                ctx.rays_query_noalloc(d_rays, n_max + 1, abi.QUERY_CLOSEST, d_hits)
            host = np.zeros(16, abi.ray_dtype)
            with pytest.raises(abi.CrtError, match="device memory"):
                ctx.rays_query_noalloc(host.ctypes.data, 16, abi.QUERY_CLOSEST, d_hits)
            with pytest.raises(abi.CrtError):
                ctx.rays_query_noalloc(0, 16, abi.QUERY_CLOSEST, d_hits)
            with pytest.raises(abi.CrtError, match="unknown mode"):
                ctx.rays_query_noalloc(d_rays, 16, 9, d_hits)
            ctx.rays_query_noalloc(0, 0, abi.QUERY_CLOSEST, 0)
            # the context still renders and queries normally afterwards
            assert len(ctx.rays_query(_rays(scene, 100, 9))) == 100
    finally:
        hip.free(d_rays); hip.free(d_hits)
