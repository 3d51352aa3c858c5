"""GPU parity of the two build extensions (crt_extensions_set): per-material textureIndex and alpha-tested triangle hits.

Neither exists in the reference (its shader reads textures[0] whatever textureIndex says, compute.comp:386-387, and has
no alpha test), so the semantics are this build's and the oracle (oracle.cpp ORC_EXT_*) is their only statement:
"parity unpinned" beyond that.  With no extension set the renderer must stay the reference's, which the first test
checks on the same scene.  Bar as everywhere: primary-hit IDs identical, accumulation image bit-exact.
"""
import numpy as np
import pytest

from computeraytracer_amd import abi
import oracle_lib as orc

pytestmark = pytest.mark.gpu

BVHS = [abi.BVH_LBVH, abi.BVH_SAH_REFINE, abi.BVH_SAH_REFINE | abi.BVH_WIDE]
BVH_IDS = ["lbvh", "sah", "sah-wide"]


@pytest.fixture(scope="module")
def ctx():
    c = abi.Context(0)
    yield c
    c.extensions_set(0)
    c.close()


def _cutout_scene(w, h, n_tris=700, seed=3, tex_sizes=((16, 16), (8, 32), (5, 7))):
    """Layers of random triangles over book1's ground sphere.  Three textured Lambertians (textures of different sizes,
    alpha in {0, 255} blocks, smooth ramps and values next to the 0.5 cut), one untextured Lambertian, one metal, a
    light.  Texture coordinates run past [0, 1] on purpose (REPEAT)."""
    base = abi.builtin_scene("book1", w, h)
    rs = np.random.RandomState(seed)
    mats = np.zeros(6, abi.material_dtype)
    mats["albedo"] = rs.uniform(0.2, 0.9, (6, 3)).astype(np.float32)
    mats["type"] = [abi.MAT_LAMBERTIAN] * 4 + [abi.MAT_METAL, abi.MAT_DIFFUSE_LIGHT]
    mats["fuzziness"] = [0, 0, 0, 0, 0.3, 0]
    mats["textureIndex"] = [0, 1, 2, abi.BAD_INDEX, abi.BAD_INDEX, abi.BAD_INDEX]
    textures = []
    for (th, tw) in tex_sizes:
        t = rs.randint(0, 256, (th, tw, 4)).astype(np.uint8)
        a = np.where(rs.uniform(size=(th, tw)) < 0.45, 0, 255)
        a[::3, ::2] = rs.randint(0, 256, a[::3, ::2].shape)          # anything
        a[1::4, 1::3] = rs.choice([127, 128], a[1::4, 1::3].shape)   # around the cut
        t[..., 3] = a
        textures.append(t)
    tris = np.zeros(n_tris, abi.tri_dtype)
    c = rs.uniform(-5, 5, (n_tris, 3)).astype(np.float32) * np.float32([1, 0.35, 1]) + np.float32([0, 1.2, 0])
    e = rs.uniform(-1.4, 1.4, (2, n_tris, 3)).astype(np.float32)
    tris["v0"], tris["v1"], tris["v2"] = c, c + e[0], c + e[1]
    tris["u"] = rs.uniform(-1.5, 2.5, (n_tris, 3)).astype(np.float32)
    tris["v"] = rs.uniform(-1.5, 2.5, (n_tris, 3)).astype(np.float32)
    tris["materialIndex"] = rs.randint(0, 6, n_tris)
    spheres = np.zeros(2, abi.sphere_dtype)
    spheres["center"], spheres["radius"], spheres["materialIndex"] = [(0, -2000, 0), (0, 1, 0)], [2000, 1.0], [3, 0]
    return abi.HostScene(spheres=spheres, tris=tris, materials=mats, textures=textures, params=base.params.copy(), name="cutout")


def _render(ctx, scene, w, h, n_iter, bvh, tracer, ext):
    ctx.configure(w, h, tracer=tracer)
    ctx.extensions_set(ext)
    ctx.upload(scene)
    ctx.build(bvh)
    seeds = abi.frame_seeds(0, n_iter)
    ps = [scene.params_for(i + 1, seeds[i]) for i in range(n_iter)]
    ctx.render(ps[0], 1, seeds)
    ctx.sync()
    return ctx.image(), ctx.hit_ids(), ps


EXT_CASES = [(0, 0), (abi.EXT_TEXTURE_INDEX, orc.EXT_TEXTURE_INDEX), (abi.EXT_ALPHA_TEST, orc.EXT_ALPHA_TEST),
             (abi.EXT_TEXTURE_INDEX | abi.EXT_ALPHA_TEST, orc.EXT_TEXTURE_INDEX | orc.EXT_ALPHA_TEST)]


@pytest.mark.parametrize("bvh", BVHS, ids=BVH_IDS)
@pytest.mark.parametrize("ext,orc_ext", EXT_CASES, ids=["none", "texidx", "alpha", "texidx+alpha"])
def test_extensions_bit_exact(ctx, ext, orc_ext, bvh):
    w, h, n_iter = 96, 64, 3
    scene = _cutout_scene(w, h)
    ref = orc.render(scene, [scene.params_for(i + 1, s) for i, s in enumerate(abi.frame_seeds(0, n_iter))], w, h,
                     flags=orc.TRAVERSE_BRUTE | orc_ext)
    for tracer in (abi.TRACER_MEGAKERNEL, abi.TRACER_WAVEFRONT):
        img, ids, _ = _render(ctx, scene, w, h, n_iter, bvh, tracer, ext)
        assert np.array_equal(ids, ref[1]), f"{(ids != ref[1]).sum()} primary-hit ID mismatches"
        assert np.array_equal(img.view(np.uint32), ref[0].view(np.uint32))
    assert (ref[1] != abi.ID_MISS).mean() > 0.5


def test_extensions_change_the_picture_and_off_is_the_reference(ctx):
    """The four settings give four different images on this scene (so the test above cannot pass by ignoring a flag),
    and alpha testing uncovers primitives behind cut-out triangles."""
    w, h = 96, 64
    scene = _cutout_scene(w, h)
    out = [_render(ctx, scene, w, h, 2, abi.BVH_SAH_REFINE | abi.BVH_WIDE, abi.TRACER_WAVEFRONT, e) for e, _ in EXT_CASES]
    for i in range(4):
        for j in range(i + 1, 4):
            assert not np.array_equal(out[i][0], out[j][0])
    assert np.array_equal(out[0][1], out[1][1])            # textureIndex does not move a primary hit
    assert (out[0][1] != out[2][1]).mean() > 0.02          # the alpha test does


@pytest.mark.parametrize("tracer", [abi.TRACER_MEGAKERNEL, abi.TRACER_WAVEFRONT], ids=["megakernel", "wavefront"])
def test_degenerate_texture_sizes(ctx, tracer):
    """1 x 1, 1 x 3 and 2 x 1 textures: every bilinear footprint wraps onto itself or its only neighbour."""
    w, h, n_iter = 64, 48, 2
    scene = _cutout_scene(w, h, n_tris=300, tex_sizes=((1, 1), (1, 3), (2, 1)))
    ref = orc.render(scene, [scene.params_for(i + 1, s) for i, s in enumerate(abi.frame_seeds(0, n_iter))], w, h,
                     flags=orc.TRAVERSE_BRUTE | orc.EXT_TEXTURE_INDEX | orc.EXT_ALPHA_TEST)
    img, ids, _ = _render(ctx, scene, w, h, n_iter, abi.BVH_SAH_REFINE | abi.BVH_WIDE, tracer, abi.EXT_TEXTURE_INDEX | abi.EXT_ALPHA_TEST)
    assert np.array_equal(ids, ref[1])
    assert np.array_equal(img.view(np.uint32), ref[0].view(np.uint32))
    ref0 = orc.render(scene, [scene.params_for(i + 1, s) for i, s in enumerate(abi.frame_seeds(0, n_iter))], w, h, flags=orc.TRAVERSE_BRUTE)
    img0, ids0, _ = _render(ctx, scene, w, h, n_iter, abi.BVH_SAH_REFINE | abi.BVH_WIDE, tracer, 0)
    assert np.array_equal(ids0, ref0[1]) and np.array_equal(img0.view(np.uint32), ref0[0].view(np.uint32))


@pytest.mark.parametrize("bvh", BVHS, ids=BVH_IDS)
def test_alpha_tested_ray_queries(ctx, bvh):
    """crt_rays_query with CRT_EXT_ALPHA_TEST: closest hits bit-exact, any-hit (shadow rays) = 'something opaque in the
    interval'; both through the allocating and the no-alloc entry point's shared kernels."""
    scene = _cutout_scene(64, 64)
    ctx.configure(64, 64)
    ctx.extensions_set(abi.EXT_ALPHA_TEST | abi.EXT_TEXTURE_INDEX)
    ctx.upload(scene)
    ctx.build(bvh)
    rs = np.random.RandomState(5)
    n = 5000
    rays = np.zeros(n, abi.ray_dtype)
    rays["origin"] = rs.uniform(-6, 6, (n, 3)).astype(np.float32) * np.float32([1, 0.2, 1]) + np.float32([0, 3, 0])
    d = rs.standard_normal((n, 3))
    d[:, 1] = -np.abs(d[:, 1])
    rays["direction"] = (d / np.linalg.norm(d, axis=1)[:, None]).astype(np.float32)
    rays["t_max"] = np.where(rs.uniform(size=n) < 0.5, np.inf, rs.uniform(0.5, 6.0, n)).astype(np.float32)
    ref = orc.hit_rays(scene, rays, flags=orc.TRAVERSE_BRUTE | orc.EXT_ALPHA_TEST | orc.EXT_TEXTURE_INDEX)
    plain = orc.hit_rays(scene, rays, flags=orc.TRAVERSE_BRUTE)
    assert (ref["id"] != plain["id"]).mean() > 0.05        # the cut-outs matter for these rays
    got = ctx.rays_query(rays, abi.QUERY_CLOSEST)
    assert np.array_equal(got["id"], ref["id"]), f"{(got['id'] != ref['id']).sum()} closest-hit IDs differ"
    hit = ref["id"] != abi.ID_MISS
    assert np.array_equal(got["t"][hit].view(np.uint32), ref["t"][hit].view(np.uint32))
    anyh = ctx.rays_query(rays, abi.QUERY_ANY)
    assert np.array_equal(anyh["id"] != abi.ID_MISS, hit)
    ctx.extensions_set(0)
    got = ctx.rays_query(rays, abi.QUERY_CLOSEST)
    assert np.array_equal(got["id"], plain["id"])


def test_extension_argument_errors(ctx):
    scene = _cutout_scene(32, 32, n_tris=20)
    bad = abi.HostScene(spheres=scene.spheres, tris=scene.tris, materials=scene.materials.copy(), textures=scene.textures,
                        params=scene.params, name="bad-index")
    bad.materials["textureIndex"][1] = 7          # past the three textures: fine for the reference's behaviour, not for the extension
    ctx.configure(32, 32)
    ctx.extensions_set(0)
    with pytest.raises(abi.CrtError):
        ctx.extensions_set(1 << 9)                # unknown bit
    ctx.upload(bad)
    with pytest.raises(abi.CrtError):
        ctx.extensions_set(abi.EXT_TEXTURE_INDEX)
    ctx.extensions_set(abi.EXT_ALPHA_TEST)        # textures[0] only: any index is fine
    ctx.upload(scene)
    ctx.extensions_set(abi.EXT_TEXTURE_INDEX)
    with pytest.raises(abi.CrtError):
        ctx.upload(bad)
    ctx.extensions_set(0)
