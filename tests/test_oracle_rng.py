"""R1 (compute.comp:116-139) and the seed schedule (vk_engine.cpp:1412): the oracle against independent restatements."""
import ctypes as C

import numpy as np

from computeraytracer_amd import abi
import oracle_lib as orc


def np_base_hash(sx, sy):
    """numpy uint32 restatement of base_hash, written from the GLSL text, not from oracle.cpp."""
    sx, sy = np.uint32(sx), np.uint32(sy)
    k = np.uint32(1103515245)
    with np.errstate(over="ignore"):
        nx = k * ((sx >> np.uint32(1)) ^ sy)
        ny = k * ((sy >> np.uint32(1)) ^ sx)
        h = k * (nx ^ (ny >> np.uint32(3)))
    return h ^ (h >> np.uint32(16))


def np_hash(seed, kind):
    """hash_1/2/3 with numpy float32 arithmetic; returns (values, new seed)."""
    seed = np.float32(seed)
    a = np.float32(seed + np.float32(0.1))
    b = np.float32(a + np.float32(0.1))
    n = np_base_hash(a.view(np.uint32), b.view(np.uint32))
    with np.errstate(over="ignore"):
        if kind == 1:
            vals = [np.float32(n) * np.float32(1.0 / 4294967296.0)]
        elif kind == 2:
            vals = [np.float32(x & np.uint32(0x7FFFFFFF)) / np.float32(2147483648.0) for x in (n, n * np.uint32(48271))]
        else:
            vals = [np.float32(x & np.uint32(0x7FFFFFFF)) / np.float32(2147483648.0)
                    for x in (n, n * np.uint32(16807), n * np.uint32(48271))]
    return vals, b


def test_base_hash_matches_numpy_restatement():
    rng = np.random.default_rng(1)
    a = rng.integers(0, 2 ** 32, 20000, dtype=np.uint64).astype(np.uint32)
    b = rng.integers(0, 2 ** 32, 20000, dtype=np.uint64).astype(np.uint32)
    want = np_base_hash(a, b)
    got = np.array([orc.lib.orc_base_hash(int(x), int(y)) for x, y in zip(a[:4000], b[:4000])], dtype=np.uint32)
    assert np.array_equal(got, want[:4000])


def test_hash_123_match_numpy_on_seed_pixel_triples():
    rng = np.random.default_rng(2)
    seeds = abi.frame_seeds(0, 32)
    n = 0
    for _ in range(4000):
        seed = seeds[rng.integers(0, 32)] if rng.random() < 0.8 else np.float32(rng.random() * 100)
        x, y = int(rng.integers(0, 3840)), int(rng.integers(0, 2160))
        s = np.float32(seed) * np.float32(np.float32(x) * np.float32(y) + np.float32(y) * np.float32(x))
        assert orc.lib.orc_pixel_seed(float(seed), x, y) == s
        cs = C.c_float(float(s))
        out = (C.c_float * 3)()
        for kind in (2, 3, 1):
            want, s = np_hash(s, kind)
            orc.lib.orc_hash(C.byref(cs), kind, out)
            assert [np.float32(out[i]) for i in range(len(want))] == want
            assert np.float32(cs.value) == s
            n += 1
    assert n == 12000


def test_reference_rng_quirks():
    # with seed ~1e6 and x*y >= 1, |hashSeed| >= 2^21 so "seed += 0.1" is a no-op in f32 and every
    # draw of that pixel in that frame returns the same n; row 0 / column 0 have hashSeed == 0 for every frame.
    seed = abi.frame_seeds(0, 1)[0]
    s = C.c_float(orc.lib.orc_pixel_seed(float(seed), 17, 5))
    before = s.value
    o1, o2 = (C.c_float * 3)(), (C.c_float * 3)()
    orc.lib.orc_hash(C.byref(s), 2, o1)
    orc.lib.orc_hash(C.byref(s), 2, o2)
    assert s.value == before and list(o1) == list(o2)
    assert orc.lib.orc_pixel_seed(float(seed), 0, 99) == 0.0 and orc.lib.orc_pixel_seed(float(seed), 99, 0) == 0.0
    assert orc.lib.orc_pixel_seed(float(seed), 3, 8) == orc.lib.orc_pixel_seed(float(seed), 6, 4)   # equal x*y share a stream


def test_glibc_rand_restatement_matches_libc_and_published_values():
    want = [1804289383, 846930886, 1681692777, 1714636915, 1957747793]   # SURVEY 8(d), glibc rand() seeded 1
    got = orc.glibc_rand(2000)
    assert list(got[:5]) == want
    libc = C.CDLL("libc.so.6")
    libc.srand(1)
    assert [libc.rand() for _ in range(2000)] == list(got)
    # the host library's own restatement (crt_frame_seeds) agrees: seed_i = float(r_i) / 1E3f
    seeds = abi.frame_seeds(0, 2000)
    assert np.array_equal(seeds, (got.astype(np.float32) / np.float32(1e3)).astype(np.float32))
    assert np.array_equal(abi.frame_seeds(100, 50), seeds[100:150])


def test_pcg32_known_answer():
    # pcg32 reference demo: pcg32_srandom(42, 54) -> 0xa15c02b7 0x7b47f409 0xba1d3330 0x83d2f293 0xbfa4784b 0xcbed606e
    out = np.zeros(6, dtype=np.uint32)
    orc.lib.orc_pcg(42, 0.0, 54.0, 6, out.ctypes.data)
    assert [hex(v) for v in out] == ["0xa15c02b7", "0x7b47f409", "0xba1d3330", "0x83d2f293", "0xbfa4784b", "0xcbed606e"]
