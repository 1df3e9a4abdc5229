"""Coefficients of the sin/cos kernels in csrc/pt_detmath.h (and its verbatim copy oracle/orc_detmath.h).
    sin(x) = x - x^3/6 + x^5 P(z),   cos(x) = 1 - z/2 + z^2/24 + z^3 Q(z),   z = x^2, |x| <= pi/4 (+ slack)
P and Q are near-minimax (Chebyshev fits computed at 120 digits, mpmath.chebyfit) of degree 5 in z; the leading terms
are evaluated in double-double inside the kernels, so the approximation error of P and Q — printed below relative to
the function value — is what bounds the distance of the result from the correctly rounded one.
    python tools/make_detmath_coeffs.py
"""
import mpmath as mp

mp.mp.dps = 120
ZMAX = (mp.pi / 4 + mp.mpf("1e-5")) ** 2


def fit(g, name, deg=5):
    coeffs, err = mp.chebyfit(g, [mp.mpf(0), ZMAX], deg + 1, error=True)     # highest power first
    coeffs = coeffs[::-1]
    print(f"// {name}: max |fit - g| on [0, (pi/4)^2] = {mp.nstr(err, 3)}")
    for i, c in enumerate(coeffs):
        print(f"    {name}{i} = {float(c)!r},   // {float(c).hex()}")
    return [float(c) for c in coeffs]


def gs(z):
    if z == 0:
        return mp.mpf(1) / 120
    x = mp.sqrt(z)
    return (mp.sin(x) - x + x ** 3 / 6) / x ** 5


def gc(z):
    if z == 0:
        return -mp.mpf(1) / 720
    x = mp.sqrt(z)
    return (mp.cos(x) - 1 + z / 2 - z * z / 24) / z ** 3


P = fit(gs, "P")
Q = fit(gc, "Q")
for name, v in (("S1 = -1/6", -mp.mpf(1) / 6), ("C2 = 1/24", mp.mpf(1) / 24)):
    hi = float(v)
    lo = float(v - mp.mpf(hi))
    print(f"// {name}: hi = {hi!r} ({hi.hex()}), lo = {lo!r} ({lo.hex()})")
# relative approximation error of the rounded polynomials over the interval
worst_s = worst_c = 0
for k in range(0, 2001):
    x = (mp.pi / 4) * k / 2000
    if x == 0:
        continue
    z = x * x
    ps = sum(mp.mpf(c) * z ** i for i, c in enumerate(P))
    pc = sum(mp.mpf(c) * z ** i for i, c in enumerate(Q))
    s = x - x ** 3 / 6 + x ** 5 * ps
    c = 1 - z / 2 + z * z / 24 + z ** 3 * pc
    worst_s = max(worst_s, abs(s - mp.sin(x)) / mp.sin(x))
    worst_c = max(worst_c, abs(c - mp.cos(x)) / mp.cos(x))
print("// relative approximation error with the coefficients rounded to f64: sin", mp.nstr(worst_s, 3), "= 2^", mp.nstr(mp.log(worst_s, 2), 4),
      " cos", mp.nstr(worst_c, 3), "= 2^", mp.nstr(mp.log(worst_c, 2), 4))


# ---- exp2 kernel of pow(2^k, y): e^r = 1 + r + r^2/2 + r^3 E(r), |r| <= ln2/2 (+ slack) ----------------------------
def ge(r):
    if r == 0:
        return mp.mpf(1) / 6
    return (mp.exp(r) - 1 - r - r * r / 2) / r ** 3


RMAX = mp.log(2) / 2 + mp.mpf("1e-6")
for deg in (10, 11, 12):
    coeffs, err = mp.chebyfit(ge, [-RMAX, RMAX], deg + 1, error=True)
    print(f"// E degree {deg}: max |fit - g| = {mp.nstr(err, 3)} (relative to e^r: x r^3 <= {mp.nstr(err * RMAX ** 3 / mp.exp(-RMAX), 3)})")
coeffs = mp.chebyfit(ge, [-RMAX, RMAX], 12)[::-1]
for i, c in enumerate(coeffs):
    print(f"    E{i} = {float(c)!r},   // {float(c).hex()}")
ln2 = mp.log(2)
hi = float(ln2)
print(f"// ln2: hi = {hi!r} ({hi.hex()}), lo = {float(ln2 - mp.mpf(hi))!r} ({float(ln2 - mp.mpf(hi)).hex()})")
