"""The device's build-defined FP32 sequences (device_math.h) and its IEEE square root and division, function by function
(crt_debug_math), against the oracle's statement of the same sequences (oracle.cpp: orc_math) and against the committed
vectors: bit for bit.  Rendered images exercise these functions only where the scenes happen to (acos / atan2 only on
spheres, the cube root only on diffuse bounces); this pins each of them over its whole domain, denormal and boundary
arguments included."""
import os

import numpy as np
import pytest

from computeraytracer_amd import abi
import oracle_lib as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = abi.Context(0)
    yield c
    c.close()


def _same(a, b):
    return np.array_equal(np.asarray(a).view(np.uint32), np.asarray(b).view(np.uint32))


def test_sincos_bit_equal(ctx):
    th = np.concatenate([np.linspace(0, 2 * np.pi, 400001), [0.0, np.pi / 2, np.pi, 1.5 * np.pi, 2 * np.pi, 1e-30, 6.2831855]]).astype(np.float32)
    s, c = ctx.debug_math(0, th)
    rs, rc = orc.math_fn(0, th)
    assert _same(s, rs) and _same(c, rc)


def test_cbrt_acos_atan2_pow5_bit_equal(ctx):
    rng = np.random.default_rng(11)
    x = np.concatenate([np.float32([0.0, 2.0 ** -31, 2.0 ** -126, 1e-40, 1e-6, 0.5, 1.0]), rng.random(400000).astype(np.float32),
                        (rng.integers(1, 2 ** 31, 50000) / 2.0 ** 31).astype(np.float32)])
    assert _same(ctx.debug_math(1, x)[0], orc.math_fn(1, x)[0])
    u = np.concatenate([np.linspace(-1, 1, 300001), [-1.0, 1.0, 0.0, -0.0, 0.99999994, -0.99999994]]).astype(np.float32)
    assert _same(ctx.debug_math(2, u)[0], orc.math_fn(2, u)[0])
    y = np.concatenate([rng.standard_normal(300000), [0.0, 1.0, -1.0, 0.0, 0.0, -0.0, 1e-30, -1e-30]]).astype(np.float32)
    xx = np.concatenate([rng.standard_normal(300000), [-1.0, 0.0, 0.0, 0.0, 1.0, -1.0, -1e-30, 1e30]]).astype(np.float32)
    assert _same(ctx.debug_math(3, y, xx)[0], orc.math_fn(3, y, xx)[0])
    p = rng.random(100000).astype(np.float32)
    assert _same(ctx.debug_math(4, p)[0], orc.math_fn(4, p)[0])


def test_sqrt_and_division_are_correctly_rounded(ctx):
    # the arithmetic contract relies on IEEE sqrt and division on the device (no approximate reciprocal, no flush of
    # denormal results): bit-equal to the host's correctly rounded operations, denormals and huge quotients included
    rng = np.random.default_rng(12)
    a = np.concatenate([np.abs(rng.standard_normal(300000)) * 10.0 ** rng.uniform(-38, 38, 300000), [0.0, 1e-45, 1.1754944e-38, 3.4028235e38, 1.0, 2.0]]).astype(np.float32)
    assert _same(ctx.debug_math(5, a)[0], np.sqrt(a))
    assert _same(ctx.debug_math(5, a)[0], orc.math_fn(5, a)[0])
    num = np.concatenate([rng.standard_normal(300000) * 10.0 ** rng.uniform(-30, 30, 300000), [1.0, 1.0, 1e-38, 3e38, -1.0, 0.0]]).astype(np.float32)
    den = np.concatenate([rng.standard_normal(300000) * 10.0 ** rng.uniform(-30, 30, 300000), [3.0, 1e-30, 1e10, 1e-10, 1e30, 5.0]]).astype(np.float32)
    with np.errstate(all="ignore"):
        want = (num / den).astype(np.float32)
    assert _same(ctx.debug_math(6, num, den)[0], want)
    assert _same(ctx.debug_math(6, num, den)[0], orc.math_fn(6, num, den)[0])


def test_device_equals_the_committed_vectors(ctx, golden_dir):
    g = np.load(os.path.join(golden_dir, "oracle_vectors.npz"))
    s, c = ctx.debug_math(0, g["sincos_in"])
    assert _same(s, g["sin_out"]) and _same(c, g["cos_out"])
    assert _same(ctx.debug_math(1, g["cbrt_in"])[0], g["cbrt_out"])
    assert _same(ctx.debug_math(2, g["acos_in"])[0], g["acos_out"])
    assert _same(ctx.debug_math(3, g["atan2_y"], g["atan2_x"])[0], g["atan2_out"])


def test_debug_math_rejects_unknown_functions(ctx):
    with pytest.raises(abi.CrtError, match="unknown function"):
        ctx.debug_math(9, np.float32([1.0]))
