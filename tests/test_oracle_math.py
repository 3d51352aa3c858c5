"""The build-defined FP32 transcendental sequences (DESIGN.md "Arithmetic") against float64 truth and libm."""
import numpy as np

import oracle_lib as orc


def ulp_err(got, want64):
    got = got.astype(np.float64)
    ulp = np.spacing(np.abs(want64).astype(np.float32)).astype(np.float64)
    return np.abs(got - want64) / ulp


def test_sincos_on_0_2pi():
    th = np.linspace(0, 2 * np.pi, 200001).astype(np.float32)
    s, c = orc.math_fn(0, th)
    t64 = th.astype(np.float64)
    assert np.abs(s - np.sin(t64)).max() < 1.5e-7 and np.abs(c - np.cos(t64)).max() < 1.5e-7
    ls, lc = orc.math_fn(0, th, libm=True)
    assert np.abs(s - ls).max() < 2.5e-7 and np.abs(c - lc).max() < 2.5e-7


def test_cbrt01():
    rng = np.random.default_rng(0)
    x = np.concatenate([np.float32([2.0 ** -31, 1e-6, 0.5, 1.0]), rng.random(200000).astype(np.float32),
                        (rng.integers(1, 2 ** 31, 20000) / 2.0 ** 31).astype(np.float32)])
    got = orc.math_fn(1, x)[0]
    assert ulp_err(got, np.cbrt(x.astype(np.float64))).max() <= 4.0   # measured 3.2 ulp; GLSL pow() is specified far looser
    assert orc.math_fn(1, np.float32([0.0]))[0][0] == 0.0
    # pow(x, 0.3333333334) in libm differs from the cube root by well under the tolerance above
    # libm's pow(x, 0.33333334f): the exponent is not exactly 1/3, worth ~3 ulp at x = 2^-31
    assert ulp_err(orc.math_fn(1, x, libm=True)[0], np.cbrt(x.astype(np.float64))).max() <= 4.0


def test_acos_atan2_pow5():
    x = np.linspace(-1, 1, 100001).astype(np.float32)
    assert np.abs(orc.math_fn(2, x)[0] - np.arccos(x.astype(np.float64))).max() < 5e-7
    rng = np.random.default_rng(3)
    y, xx = rng.standard_normal(100000).astype(np.float32), rng.standard_normal(100000).astype(np.float32)
    assert np.abs(orc.math_fn(3, y, xx)[0] - np.arctan2(y.astype(np.float64), xx.astype(np.float64))).max() < 5e-7
    p = rng.random(1000).astype(np.float32)
    assert ulp_err(orc.math_fn(4, p)[0], p.astype(np.float64) ** 5).max() <= 3.0   # three roundings
    for yy, xv, want in [(0.0, -1.0, np.pi), (1.0, 0.0, np.pi / 2), (-1.0, 0.0, -np.pi / 2), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0)]:
        assert abs(orc.math_fn(3, np.float32([yy]), np.float32([xv]))[0][0] - want) < 3e-7


def test_committed_vectors(golden_dir):
    import os
    g = np.load(os.path.join(golden_dir, "oracle_vectors.npz"))
    s, c = orc.math_fn(0, g["sincos_in"])
    assert np.array_equal(s, g["sin_out"]) and np.array_equal(c, g["cos_out"])
    assert np.array_equal(orc.math_fn(1, g["cbrt_in"])[0], g["cbrt_out"])
    assert np.array_equal(orc.math_fn(2, g["acos_in"])[0], g["acos_out"])
    assert np.array_equal(orc.math_fn(3, g["atan2_y"], g["atan2_x"])[0], g["atan2_out"])
