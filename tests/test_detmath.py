"""Deterministic elementary functions (csrc/pt_detmath.h, mirrored in oracle/orc_detmath.h):
accuracy against the platform libm (numpy -> glibc) and the sync of the two copies."""
import os
import re

import numpy as np

from common import ulp_diff
from conftest import ROOT


def _vec(orc, which, a, b=None):
    b = np.zeros_like(a) if b is None else b
    return np.array([orc.detmath(which, float(x), float(y)) for x, y in zip(a, b)])


def _libm(f, a, b=None):
    """glibc through Python's math module (numpy's SIMD sin/cos are NOT 1-ulp near zeros)."""
    if b is None:
        return np.array([f(float(x)) for x in a])
    return np.array([f(float(x), float(y)) for x, y in zip(a, b)])


def test_accuracy_against_libm(orc):
    import math

    rng = np.random.default_rng(11)
    n = 20000
    x = rng.uniform(-7, 7, n)
    assert ulp_diff(_vec(orc, 3, x), _libm(math.sin, x)).max() <= 1
    assert ulp_diff(_vec(orc, 4, x), _libm(math.cos, x)).max() <= 1
    big = rng.uniform(-1e5, 1e5, n)
    assert ulp_diff(_vec(orc, 3, big), _libm(math.sin, big)).max() <= 1
    c = rng.uniform(-1, 1, n)
    assert ulp_diff(_vec(orc, 5, c), _libm(math.acos, c)).max() <= 1
    y, xx = rng.uniform(-1, 1, n), rng.uniform(-1, 1, n)
    assert ulp_diff(_vec(orc, 6, y, xx), _libm(math.atan2, y, xx)).max() <= 1
    # the one-division forms (round 2): around the switch points of acos (|x| = 0.5, 1) and of atan2 (|y/x| = tan(pi/8), 1),
    # where atan2's quotient of two rounded sums can cost a second ulp
    edge = np.concatenate([rng.uniform(0.49, 0.51, 4000), -rng.uniform(0.49, 0.51, 4000), 1 - np.exp(rng.uniform(-40, -1, 4000)),
                           np.exp(rng.uniform(-40, -1, 4000)) - 1, np.exp(rng.uniform(-40, -1, 2000))])
    assert ulp_diff(_vec(orc, 5, edge), _libm(math.acos, edge)).max() <= 1
    xs = rng.uniform(-1, 1, 8000)
    for ratio in (rng.uniform(0.41, 0.42, 8000), rng.uniform(0.99, 1.01, 8000)):
        assert ulp_diff(_vec(orc, 6, ratio * xs, xs), _libm(math.atan2, ratio * xs, xs)).max() <= 2
        assert ulp_diff(_vec(orc, 6, xs, ratio * xs), _libm(math.atan2, xs, ratio * xs)).max() <= 2
    wy, wx = (np.exp(rng.uniform(-300, 300, 8000)) * rng.choice([-1, 1], 8000) for _ in range(2))
    assert ulp_diff(_vec(orc, 6, wy, wx), _libm(math.atan2, wy, wx)).max() <= 1
    p = np.exp(rng.uniform(-30, 30, n))
    assert ulp_diff(_vec(orc, 10, p), _libm(math.log, p)).max() <= 1
    assert ulp_diff(_vec(orc, 8, p), _libm(math.log2, p)).max() <= 2
    e = rng.uniform(-50, 50, n)
    assert ulp_diff(_vec(orc, 11, e), _libm(math.exp, e)).max() <= 1
    # the renderer's only pow: GTR1 sampling, pow(0.25^2, 1 - e1)  (sampling.rs:132)
    t = rng.uniform(0, 1, n)
    assert ulp_diff(_vec(orc, 7, np.full(n, 0.0625), t), _libm(math.pow, np.full(n, 0.0625), t)).max() <= 4


def test_sin_cos_almost_always_equal_glibc(orc):
    """What keeps the render within north_star's tolerance of a libm-based renderer is not the ulp bound but HOW OFTEN
    the last bit differs (every difference can flip a checker cell or an offset sign down the path): the round-2 kernels
    (double-double leading terms, one final rounding) differ from glibc in ~0.5 % (sin) / ~0.3 % (cos) of the calls over
    the renderer's argument range [0, 2 pi); the fdlibm kernels of round 1 did in 3.4 %. mpmath confirms the split: glibc is
    correctly rounded in 99.87 % of the calls, these kernels in 99.6-99.8 %."""
    import math

    rng = np.random.default_rng(5)
    x = rng.uniform(0.0, 2.0 * math.pi, 60000)
    ms = np.mean(_vec(orc, 3, x) != _libm(math.sin, x))
    mc = np.mean(_vec(orc, 4, x) != _libm(math.cos, x))
    assert ms < 0.008 and mc < 0.006, (ms, mc)
    tiny = np.concatenate([rng.uniform(-1e-6, 1e-6, 2000), [0.0, -0.0, 1e-300, -1e-300, 2.0 ** -27, -(2.0 ** -26)]])
    s, c = _vec(orc, 3, tiny), _vec(orc, 4, tiny)
    assert ulp_diff(s, _libm(math.sin, tiny)).max() <= 1 and ulp_diff(c, _libm(math.cos, tiny)).max() <= 1
    assert np.signbit(orc.detmath(3, -0.0, 0.0)) and not np.signbit(orc.detmath(3, 0.0, 0.0))


def test_special_values(orc):
    d = orc.detmath
    assert d(5, 1.0) == 0.0 and d(5, -1.0) == np.pi and np.isnan(d(5, 1.0000001))
    assert d(6, 0.0, -1.0) == np.pi and d(6, 0.0, 1.0) == 0.0 and d(6, 1.0, 0.0) == np.pi / 2 and d(6, -1.0, 0.0) == -np.pi / 2
    import math
    inf = math.inf
    for y, x in [(0.0, 0.0), (-0.0, 0.0), (0.0, -0.0), (-0.0, -0.0), (1.0, 1.0), (-1.0, 1.0), (1.0, -1.0), (-1.0, -1.0), (inf, inf), (inf, -inf),
                 (-inf, inf), (1.0, inf), (1.0, -inf), (-1.0, -inf), (inf, 1.0), (-inf, -1.0), (1e-310, 1.0), (1e-310, -1.0), (1.0, 1e-310), (-0.0, 1.0), (-0.0, -1.0)]:
        got, want = d(6, y, x), math.atan2(y, x)
        assert got == want and math.copysign(1.0, got) == math.copysign(1.0, want), (y, x, got, want)
    assert np.isnan(d(6, np.nan, 1.0)) and np.isnan(d(6, 1.0, np.nan)) and np.isnan(d(5, np.nan)) and np.isnan(d(5, -1.0000001))
    assert d(5, 0.5) == math.acos(0.5) and d(5, -0.5) == math.acos(-0.5) and d(5, 0.0) == math.pi / 2 and d(5, 1e-20) == math.pi / 2
    assert d(3, 0.0) == 0.0 and d(4, 0.0) == 1.0 and np.isnan(d(3, np.inf)) and np.isnan(d(4, np.nan))
    assert d(10, 1.0) == 0.0 and d(10, 0.0) == -np.inf and np.isnan(d(10, -1.0)) and d(10, np.inf) == np.inf
    assert d(11, 0.0) == 1.0 and d(11, 1000.0) == np.inf and d(11, -1000.0) == 0.0
    assert d(8, 8.0) == 3.0 and d(8, 0.5) == -1.0


def test_oracle_copy_is_in_sync_with_product_header():
    a = open(os.path.join(ROOT, "thu-acg-f2024-path-tracer_amd", "csrc", "pt_detmath.h")).read()
    b = open(os.path.join(ROOT, "oracle", "orc_detmath.h")).read()
    body = lambda s: s[s.index("#pragma once"):]
    assert body(a) == body(b), "oracle/orc_detmath.h must be a verbatim copy of csrc/pt_detmath.h (bit-exact parity depends on it)"
    # no libm call may sneak in; the only fused operation is the explicit, exactly specified __builtin_fma (dm_fma)
    code = re.sub(r"//.*", "", body(a))
    assert "std::" not in code and "#include <cmath>" not in code and "math.h" not in code
    assert set(re.findall(r"\b\w*fma\w*\b", code)) == {"dm_fma", "__builtin_fma"}
