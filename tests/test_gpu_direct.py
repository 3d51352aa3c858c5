"""CRT_BVH_WIDE_DIRECT (bvh_direct.hip): the 8-wide tree built straight from the sorted Morton keys, no pair BVH.

Same bar as every other tree: primary-hit IDs identical to the oracle's, accumulation image bit-exact, ray queries
bit-exact; plus the structure of the tree itself (every primitive in exactly one leaf slot, every quantised box
contains what is beneath it) and the error behaviour of what the mode cannot serve.
"""
import numpy as np
import pytest

from computeraytracer_amd import abi
import oracle_lib as orc
from test_gpu_parity import _check_wide, _gpu_render, _query_rays, _tiny_scene

pytestmark = pytest.mark.gpu
DIRECT = abi.BVH_WIDE_DIRECT


@pytest.fixture(scope="module")
def ctx():
    c = abi.Context(0)
    yield c
    c.close()


CASES = [("quads", 64, 64, 3), ("corne", 96, 96, 3), ("book1", 200, 112, 3), ("voxel_city", 160, 96, 2), ("grid:2", 96, 64, 2),
         ("cirno", 96, 60, 2), ("monkey", 96, 60, 2)]


@pytest.mark.parametrize("name,w,h,n_iter", CASES)
def test_direct_wide_bit_exact_and_structure(ctx, golden_dir, name, w, h, n_iter):
    scene = abi.builtin_scene(name, w, h, golden_dir)
    img, ids, ps = _gpu_render(ctx, scene, w, h, n_iter, bvh=DIRECT, tracer=abi.TRACER_WAVEFRONT)
    ref_img, ref_ids, _ = orc.render(scene, ps, w, h, flags=orc.TRAVERSE_FAST)
    assert np.array_equal(ids, ref_ids), f"{(ids != ref_ids).sum()} primary-hit ID mismatches"
    assert np.array_equal(img.view(np.uint32), ref_img.view(np.uint32))
    nodes, sid, qid, tid = ctx.bvh()
    assert len(nodes) == 0                      # no pair BVH in this mode
    wide, levels = ctx.bvh_wide()
    assert len(wide) > 0 and 1 <= levels <= 64
    _check_wide(wide, levels, scene, sid, qid, tid)


def test_direct_leaves_hold_up_to_two_primitives(ctx, golden_dir):
    """Single-kind scenes close parts of two keys as one leaf slot (the default of the CRT_DW_LEAF experiment knob)."""
    scene = abi.builtin_scene("voxel_city", 128, 72, golden_dir)
    ctx.configure(128, 72, tracer=abi.TRACER_WAVEFRONT)
    ctx.upload(scene)
    ctx.build(DIRECT)
    wide, _ = ctx.bvh_wide()
    links = wide[:, 16:24]
    leaf_links = links[(links & 0x80000000 != 0) & (links != 0xE0000000)]
    cnt = ((leaf_links >> 26) & 7) + 1
    assert cnt.max() == 2 and cnt.sum() == len(scene.tris)
    mixed = abi.builtin_scene("book1", 64, 64)       # spheres only as well; cirno/monkey scenes mix a ground sphere in
    m = abi.builtin_scene("monkey", 64, 64, golden_dir)
    ctx.upload(m)
    ctx.build(DIRECT)
    wide, _ = ctx.bvh_wide()
    links = wide[:, 16:24]
    leaf_links = links[(links & 0x80000000 != 0) & (links != 0xE0000000)]
    if len(m.spheres) and len(m.tris):
        assert (((leaf_links >> 26) & 7) + 1).max() == 1      # mixed kinds: one primitive per leaf slot
    del mixed


@pytest.mark.parametrize("kind", ["tris2", "tris3", "tris4", "spheres2"])
def test_direct_tiny_scenes(ctx, kind):
    w, h = 48, 48
    scene = _tiny_scene(kind, w, h)
    img, ids, ps = _gpu_render(ctx, scene, w, h, 2, bvh=DIRECT, tracer=abi.TRACER_WAVEFRONT)
    ref_img, ref_ids, _ = orc.render(scene, ps, w, h, flags=orc.TRAVERSE_BRUTE)
    assert np.array_equal(ids, ref_ids) and np.array_equal(img.view(np.uint32), ref_img.view(np.uint32))
    assert (ref_ids != abi.ID_MISS).any()


def test_direct_single_and_empty(ctx):
    q = abi.builtin_scene("quads", 32, 32)
    single = abi.HostScene(quads=q.quads[1:2], materials=q.materials, params=q.params)
    img, ids, ps = _gpu_render(ctx, single, 32, 32, 2, bvh=DIRECT, tracer=abi.TRACER_WAVEFRONT)
    ref_img, ref_ids, _ = orc.render(single, ps, 32, 32, flags=orc.TRAVERSE_BRUTE)
    assert np.array_equal(img.view(np.uint32), ref_img.view(np.uint32)) and np.array_equal(ids, ref_ids)
    empty = abi.HostScene(materials=q.materials, params=q.params)
    img, ids, ps = _gpu_render(ctx, empty, 32, 32, 1, bvh=DIRECT, tracer=abi.TRACER_WAVEFRONT)
    ref_img, ref_ids, _ = orc.render(empty, ps, 32, 32, flags=orc.TRAVERSE_BRUTE)
    assert np.array_equal(img.view(np.uint32), ref_img.view(np.uint32)) and (ids == abi.ID_MISS).all()


@pytest.mark.parametrize("name", ["book1", "cirno", "voxel_city"])
def test_direct_ray_queries(ctx, golden_dir, name):
    scene = abi.builtin_scene(name, 64, 64, golden_dir)
    ctx.configure(64, 64)
    ctx.upload(scene)
    ctx.build(DIRECT)
    rays = _query_rays(scene, 4000, 11)
    ref = orc.hit_rays(scene, rays, flags=orc.TRAVERSE_BRUTE)
    got = ctx.rays_query(rays, abi.QUERY_CLOSEST)
    assert np.array_equal(got["id"], ref["id"])
    hit = ref["id"] != abi.ID_MISS
    assert np.array_equal(got["t"][hit].view(np.uint32), ref["t"][hit].view(np.uint32))
    anyh = ctx.rays_query(rays, abi.QUERY_ANY)
    assert np.array_equal(anyh["id"] != abi.ID_MISS, hit)


def test_direct_million_triangles_agrees_with_the_collapsed_lbvh(ctx):
    """grid:34 (1.16 M triangles, above the packed-key threshold of the other modes as well): same raw image and IDs as
    the pair-BVH route; the tree is read back and checked level by level for primitive coverage."""
    w, h = 640, 360
    scene = abi.builtin_scene("grid:34", w, h)
    a_img, a_ids, _ = _gpu_render(ctx, scene, w, h, 2, bvh=abi.BVH_LBVH | abi.BVH_WIDE, tracer=abi.TRACER_WAVEFRONT)
    b_img, b_ids, _ = _gpu_render(ctx, scene, w, h, 2, bvh=DIRECT, tracer=abi.TRACER_WAVEFRONT)
    assert np.array_equal(a_ids, b_ids) and np.array_equal(a_img.view(np.uint32), b_img.view(np.uint32))
    assert (a_ids != abi.ID_MISS).mean() > 0.2
    wide, levels = ctx.bvh_wide()
    _, _, _, tid = ctx.bvh()
    links = wide[:, 16:24]
    leaf = (links & 0x80000000 != 0) & (links != 0xE0000000)
    first, cnt = links[leaf] & 0x3FFFFFF, ((links[leaf] >> 26) & 7) + 1
    covered = np.zeros(len(scene.tris) + 1, np.int64)
    np.add.at(covered, first, 1)
    np.add.at(covered, first + cnt, -1)
    assert (np.cumsum(covered)[:-1] == 1).all()          # every sorted slot in exactly one leaf
    assert np.array_equal(np.sort(tid & np.uint32(0x3FFFFFFF)), np.arange(len(scene.tris), dtype=np.uint32))      # full IDs: kind << 30 | index
    imask = wide[:, 3] & 0xFF
    n_inner = np.array([bin(int(m)).count("1") for m in imask[:4096]]).sum()
    assert n_inner > 0 and levels <= 12


def test_direct_animate_and_rebuild(ctx):
    """Config 5's loop in this mode: animate, rebuild from the Morton order, render one sample; against the oracle on
    the triangles read back from the device (same protocol as test_config5_animate_rebuild_and_update)."""
    w, h = 128, 80
    scene = abi.builtin_scene("grid:10", w, h)
    ctx.configure(w, h, tracer=abi.TRACER_WAVEFRONT)
    ctx.upload(scene)
    seeds = abi.frame_seeds(0, 4)
    accum = np.zeros((h, w, 4), np.float32)
    for frame in (1, 2, 3):
        ctx.animate(float(frame), 0.35)
        ctx.build(DIRECT)
        tris = ctx.tris_read()
        p = scene.params_for(1, seeds[frame - 1])
        ctx.render(p, 1, np.float32([seeds[frame - 1]])); ctx.sync()
        moved = abi.HostScene(tris=tris, materials=scene.materials, textures=scene.textures, params=scene.params, name="grid-animated")
        ref_img, ref_ids, _ = orc.render(moved, [p], w, h, flags=orc.TRAVERSE_FAST, accum=accum.copy())
        assert np.array_equal(ctx.hit_ids(), ref_ids)
        assert np.array_equal(ctx.image().view(np.uint32), ref_img.view(np.uint32)), f"frame {frame}"
        assert (ref_ids != abi.ID_MISS).mean() > 0.2
        accum = ref_img


def test_direct_mode_errors(ctx):
    scene = abi.builtin_scene("corne", 32, 32)
    ctx.configure(32, 32, tracer=abi.TRACER_MEGAKERNEL)
    ctx.upload(scene)
    with pytest.raises(abi.CrtError):
        ctx.build(DIRECT | abi.BVH_SAH_REFINE)
    with pytest.raises(abi.CrtError):
        ctx.build(DIRECT | abi.BVH_LEAF_COLLAPSE)
    ctx.build(DIRECT)
    seeds = abi.frame_seeds(0, 1)
    with pytest.raises(abi.CrtError):          # the megakernel walks the pair BVH, which this build does not have
        ctx.render(scene.params_for(1, seeds[0]), 1, seeds)
    ctx.build(abi.BVH_LBVH)                    # ... and a pair build afterwards serves it again
    ctx.render(scene.params_for(1, seeds[0]), 1, seeds)
    ctx.sync()
    assert len(ctx.bvh()[0]) > 0


# ---- small batches: a kernel pair per bounce or one lane per path (k_wf_tail); CRT_FLAG_PER_BOUNCE_KERNELS / CRT_FLAG_PATH_KERNEL ----
MODES = [abi.FLAG_PER_BOUNCE_KERNELS, abi.FLAG_PATH_KERNEL, abi.FLAG_PATH_KERNEL_PERSISTENT]


@pytest.mark.parametrize("name,w,h,n_iter,bvh,rng", [("book1", 200, 112, 3, abi.BVH_SAH_REFINE | abi.BVH_WIDE, abi.RNG_REF),
                                                    ("book1", 128, 72, 3, abi.BVH_LBVH, abi.RNG_PCG),
                                                    ("corne", 100, 50, 2, abi.BVH_SAH_REFINE, abi.RNG_REF),       # ragged: lanes outside the rendered grid
                                                    ("cirno", 96, 60, 2, DIRECT, abi.RNG_REF),
                                                    ("voxel_city", 640, 360, 2, abi.BVH_SAH_REFINE | abi.BVH_WIDE, abi.RNG_PCG)],
                         ids=["book1-wide", "book1-pairs-pcg", "corne-ragged", "cirno-direct", "voxel-640-pcg"])
def test_path_kernel_and_per_bounce_kernels_match_the_oracle(golden_dir, name, w, h, n_iter, bvh, rng):
    scene = abi.builtin_scene(name, w, h, golden_dir)
    ref, rays = None, []
    for mode in MODES:
        c = abi.Context(0)
        try:
            img, ids, ps = _gpu_render(c, scene, w, h, n_iter, bvh=bvh, tracer=abi.TRACER_WAVEFRONT, rng_mode=rng, flags=mode)
            if ref is None:
                ref = orc.render(scene, ps, w, h, flags=orc.TRAVERSE_FAST | (orc.RNG_PCG if rng == abi.RNG_PCG else 0))
            assert np.array_equal(ids, ref[1]), f"mode {mode}: {(ids != ref[1]).sum()} primary-hit ID mismatches"
            assert np.array_equal(img.view(np.uint32), ref[0].view(np.uint32)), f"mode {mode}"
            rays.append(c.stats().rays)
        finally:
            c.close()
    assert rays[0] == rays[1] == rays[2] == ref[2].rays          # all count hit_world calls


def test_path_kernel_debug_view_and_extensions(golden_dir):
    """R15's outline view (samples killed before any traversal) and the build extensions through k_wf_tail."""
    from test_gpu_extensions import _cutout_scene
    w, h = 96, 64
    scene = _cutout_scene(w, h)
    ext, orc_ext = abi.EXT_TEXTURE_INDEX | abi.EXT_ALPHA_TEST, orc.EXT_TEXTURE_INDEX | orc.EXT_ALPHA_TEST
    ps = [scene.params_for(i + 1, s) for i, s in enumerate(abi.frame_seeds(0, 2))]
    ref = orc.render(scene, ps, w, h, flags=orc.TRAVERSE_BRUTE | orc_ext)
    for mode in MODES:
        c = abi.Context(0)
        try:
            c.configure(w, h, tracer=abi.TRACER_WAVEFRONT, flags=mode)
            c.extensions_set(ext)
            c.upload(scene)
            c.build(abi.BVH_SAH_REFINE | abi.BVH_WIDE)
            c.render(ps[0], 1, abi.frame_seeds(0, 2))
            c.sync()
            assert np.array_equal(c.hit_ids(), ref[1]) and np.array_equal(c.image().view(np.uint32), ref[0].view(np.uint32)), f"mode {mode}"
        finally:
            c.close()


def test_mode_is_measured_and_then_kept(golden_dir):
    """Default (no flag): the first dispatches alternate between the two forms, then one is kept; every frame is the
    oracle's whichever form rendered it."""
    w, h = 128, 72
    scene = abi.builtin_scene("book1", w, h)
    c = abi.Context(0)
    try:
        c.configure(w, h, tracer=abi.TRACER_WAVEFRONT)
        c.upload(scene)
        c.build(abi.BVH_SAH_REFINE | abi.BVH_WIDE)
        seeds = abi.frame_seeds(0, 16)
        ps = [scene.params_for(i + 1, seeds[i]) for i in range(16)]
        for i in range(16):
            c.dispatch(ps[i])
        c.sync()
        ref_img, ref_ids, cnt = orc.render(scene, ps, w, h, flags=orc.TRAVERSE_FAST)
        assert np.array_equal(c.hit_ids(), ref_ids) and np.array_equal(c.image().view(np.uint32), ref_img.view(np.uint32))
        assert c.stats().rays == cnt.rays
    finally:
        c.close()
