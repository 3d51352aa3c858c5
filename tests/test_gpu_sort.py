"""The build's own radix sort (bvh_sort.hip, crt_debug_sort_u64) against numpy on Morton-shaped keys: code << idx_bits |
index with long runs of equal codes, sorted over the code's bits only; stable, so keys with equal codes keep their order
(= a stable numpy sort by the masked key).  Sizes cross the tile sizes (1024 keys up to a million keys, 4096 above), the
single-tile case and several thousand tiles; with and without a value per key."""
import os
import subprocess
import sys

import numpy as np
import pytest

from computeraytracer_amd import abi

pytestmark = pytest.mark.gpu


def _keys(n, idx_bits, code_bits, seed, runs):
    rs = np.random.RandomState(seed)
    code = rs.randint(0, 1 << min(code_bits, 62), size=n, dtype=np.int64).astype(np.uint64)
    if runs:        # long runs of equal codes (flat terrain: many centroids in one cell)
        code = code[rs.randint(0, max(1, n // runs), size=n)]
    if code_bits < 64 - idx_bits:
        code &= np.uint64((1 << code_bits) - 1)
    return (code << np.uint64(idx_bits)) | np.arange(n, dtype=np.uint64)[rs.permutation(n)]


@pytest.mark.parametrize("n,idx_bits,code_bits,runs", [(1, 1, 39, 0), (2, 1, 8, 0), (4096, 12, 39, 0), (4097, 13, 39, 50), (100_000, 17, 30, 0),
                                                       (1_000_003, 20, 39, 1000), (3_000_000, 22, 13, 0), (10_000_000, 24, 39, 7)])
def test_radix_sort_matches_a_stable_numpy_sort(n, idx_bits, code_bits, runs):
    keys = _keys(n, idx_bits, code_bits, n % 97, runs)
    with abi.Context(0) as ctx:
        got = ctx.debug_sort_u64(keys, idx_bits, idx_bits + code_bits)
    order = np.argsort(keys >> np.uint64(idx_bits), kind="stable")
    assert np.array_equal(got, keys[order])


def test_radix_sort_full_width_and_empty_range():
    rs = np.random.RandomState(5)
    keys = rs.randint(0, 1 << 62, size=70_000, dtype=np.int64).astype(np.uint64) * np.uint64(3)
    with abi.Context(0) as ctx:
        assert np.array_equal(ctx.debug_sort_u64(keys, 0, 64), np.sort(keys))
        assert np.array_equal(ctx.debug_sort_u64(keys, 5, 5), keys)          # no bits: nothing moves
        assert np.array_equal(ctx.debug_sort_u64(keys, 60, 64), keys[np.argsort(keys >> np.uint64(60), kind="stable")])


@pytest.mark.parametrize("n,code_bits,runs", [(1, 63, 0), (3, 63, 0), (1024, 63, 0), (1025, 63, 9), (224_998, 63, 0), (224_998, 63, 300), (1 << 20, 30, 0), ((1 << 20) + 1, 63, 5)])
def test_radix_sort_of_pairs_matches_a_stable_numpy_sort(n, code_bits, runs):
    # the static builds' form: 63-bit Morton codes, the primitive index as the value
    keys = _keys(n, 0, code_bits, n % 89, runs) >> np.uint64(0)
    rs = np.random.RandomState(n % 83)
    keys = (keys & ~np.uint64((1 << 20) - 1)) | (keys & np.uint64((1 << 20) - 1)) >> np.uint64(rs.randint(0, 12))      # equal keys happen
    vals = rs.permutation(n).astype(np.uint32)
    with abi.Context(0) as ctx:
        got_k, got_v = ctx.debug_sort_pairs_u64(keys, vals, 0, code_bits)
    mask = np.uint64((1 << code_bits) - 1)
    order = np.argsort(keys & mask, kind="stable")
    assert np.array_equal(got_k, keys[order]) and np.array_equal(got_v, vals[order])


@pytest.mark.parametrize("name", ["voxel_city", "cirno"])
def test_static_build_through_the_own_sort_gives_the_same_tree(name):
    # CRT_BVH_SORT=onesweep sends the (Morton code, primitive) pairs of a static build through bvh_sort.hip instead of rocPRIM;
    # both sorts are stable, so the order, the tree and the image are the same.  The variable is read once per process.
    here = os.path.dirname(os.path.abspath(__file__))
    out = []
    for sort in ("rocprim", "onesweep"):
        env = dict(os.environ, CRT_BVH_SORT=sort, PYTHONPATH=os.pathsep.join([os.path.dirname(here), here, os.environ.get("PYTHONPATH", "")]))
        r = subprocess.run([sys.executable, os.path.join(here, "sort_build_check.py"), name], env=env, capture_output=True, text=True, timeout=240)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
        out.append([l for l in r.stdout.splitlines() if l.startswith("tree ")][-1])
    assert out[0] == out[1] and int(out[0].split()[1]) > 0
