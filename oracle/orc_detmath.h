// Deterministic f64 elementary functions: sin/cos, acos, atan2, log, log2, exp, pow.
//
// Why: the integrator has discontinuities that turn a 1-ulp difference between two libm
// implementations into a visibly different sample (the 3-D checker on the y = 0 ground plane
// floors a coordinate that is 0 +- 1e-16, texture.rs:44-48; the light-plane offset sign,
// camera.rs:217). With these functions the GPU kernels and the CPU oracle (in its "det" math
// mode) execute the SAME sequence of IEEE-754 operations (+ - * / sqrt and fma — each exactly
// specified, so a hardware fma on the GPU and on the CPU give the same bits; no tables beyond
// the listed constants), so their results agree bit for bit and parity tests can demand
// equality instead of a tolerance.
//
// Algorithms: fdlibm's (Sun Microsystems, freely redistributable) 3-part Cody-Waite reduction
// (valid for |x| < 2^19*pi/2; larger or non-finite arguments, which the renderer never
// produces, fall back to an fmod-style reduction), e_acos, s_atan/e_atan2, e_log, e_exp with
// their polynomials evaluated by fma; log2(x) = log(x)/ln2 and pow(x,y) = exp(y*log(x)) (x > 0)
// are composed from those (<= 4 ulp for |y log x| <= 4; the renderer only calls pow(0.0625, y),
// y in (0,1]).
// sin / cos have their own kernels (round 2): the two leading terms of each series are carried
// in double-double (error-free products by fma, Fast2Sum), the rest is a degree-5 polynomial in
// x^2 fitted at 120 digits (tools/make_detmath_coeffs.py; approximation error 2^-64), and the
// result is rounded ONCE — it is the correctly rounded value in all but ~0.3 % of the calls.
// Why that matters: glibc's sin/cos (what the Rust reference calls) are correctly rounded in
// 99.87 % of calls, fdlibm's kernels in 96.5 %; every last-bit difference of a sampled
// direction can flip a checker cell or an offset sign further down the path, and with the
// fdlibm kernels the image RMSE against the libm-mode oracle at 4000 spp was 1.6e-4 on
// scene 6 — above north_star's 1e-4 (tests/test_gpu_parity.py::test_north_star_tolerance).
// Accuracy of each function is pinned against glibc in tests/test_detmath.py.
//
// This file is compiled for host and device; oracle/orc_detmath.h is a verbatim copy (the
// oracle may not include product headers and vice versa; a test keeps the two in sync).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define PT_DM __host__ __device__ inline
#else
#define PT_DM inline
#endif

namespace detmath {

PT_DM uint64_t dm_bits(double x) {
    uint64_t u;
    __builtin_memcpy(&u, &x, 8);
    return u;
}
PT_DM double dm_from_bits(uint64_t u) {
    double x;
    __builtin_memcpy(&x, &u, 8);
    return x;
}
PT_DM int32_t dm_hi(double x) { return (int32_t)(dm_bits(x) >> 32); }
PT_DM uint32_t dm_lo(double x) { return (uint32_t)dm_bits(x); }
PT_DM double dm_words(int32_t hi, uint32_t lo) { return dm_from_bits(((uint64_t)(uint32_t)hi << 32) | lo); }
PT_DM double dm_abs(double x) { return dm_from_bits(dm_bits(x) & 0x7FFFFFFFFFFFFFFFull); }
PT_DM double dm_nan() { return dm_from_bits(0x7FF8000000000000ull); }
PT_DM double dm_sqrt(double x) { return __builtin_sqrt(x); }   // IEEE correctly rounded on both sides

// ---- sin / cos ------------------------------------------------------------------------
PT_DM double dm_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
// sin(x + y), |x| <= pi/4 (+ slack), y = tail of the reduced argument.
//   sin x = x - x^3/6 + x^5 P(z), z = x^2. x^2, x^3 and x^3 * (-1/6) are double-double; the sum is rounded once.
PT_DM double dm_ksin(double x, double y) {
    const double S1h = -0x1.5555555555555p-3, S1l = -0x1.5555555555555p-57;   // -1/6
    const double P0 = 0x1.1111111111111p-7, P1 = -0x1.a01a01a019ed6p-13, P2 = 0x1.71de3a550cb3ap-19, P3 = -0x1.ae6455341c2b5p-26,
                 P4 = 0x1.61225b335745dp-33, P5 = -0x1.ab93f5e6a14a5p-41;
    if ((dm_hi(x) & 0x7fffffff) < 0x3e500000) return x;   // |x| < 2^-26: x^3/6 is below half an ulp (keeps -0)
    const double zh = x * x;
    const double zl = dm_fma(x, x, -zh);                      // x^2 = zh + zl exactly
    double p = dm_fma(P5, zh, P4);
    p = dm_fma(p, zh, P3);
    p = dm_fma(p, zh, P2);
    p = dm_fma(p, zh, P1);
    p = dm_fma(p, zh, P0);
    const double wh = x * zh;
    const double wl = dm_fma(x, zh, -wh) + x * zl;            // x^3 = wh + wl
    const double ch = wh * S1h;
    const double cl = dm_fma(wh, S1h, -ch) + (wh * S1l + wl * S1h);   // -x^3/6 = ch + cl
    const double q = (wh * zh) * p;                           // x^5 P(z)
    const double t = dm_fma(-0.5 * zh, y, y);                 // sin(x+y) - sin(x) ~ y cos x ~ y (1 - z/2)
    const double sh = x + ch;
    const double sl = (x - sh) + ch;                          // Fast2Sum, |x| >= |ch|
    return sh + (sl + (cl + (q + t)));
}
// cos(x + y): cos x = 1 - z/2 + z^2/24 + z^3 Q(z); z/2, z^2 and z^2/24 are double-double.
PT_DM double dm_kcos(double x, double y) {
    const double C2h = 0x1.5555555555555p-5, C2l = 0x1.5555555555555p-59;     // 1/24
    const double Q0 = -0x1.6c16c16c16c17p-10, Q1 = 0x1.a01a01a019f8ap-16, Q2 = -0x1.27e4fb775f620p-22, Q3 = 0x1.1eed8e6c45572p-29,
                 Q4 = -0x1.93957ddf1b130p-37, Q5 = 0x1.abe6c47d7a407p-45;
    const double zh = x * x;
    const double zl = dm_fma(x, x, -zh);
    double q = dm_fma(Q5, zh, Q4);
    q = dm_fma(q, zh, Q3);
    q = dm_fma(q, zh, Q2);
    q = dm_fma(q, zh, Q1);
    q = dm_fma(q, zh, Q0);
    const double wh = zh * zh;
    const double wl = dm_fma(zh, zh, -wh) + (2.0 * zh) * zl;  // z^2 = wh + wl
    const double dh = wh * C2h;
    const double dl = dm_fma(wh, C2h, -dh) + (wh * C2l + wl * C2h);   // z^2/24 = dh + dl
    const double r = (wh * zh) * q;                           // z^3 Q(z)
    const double h = 0.5 * zh;
    const double ah = 1.0 - h;
    const double al = (1.0 - ah) - h;                         // Fast2Sum, 1 >= h
    const double bh = ah + dh;
    const double bl = (ah - bh) + dh;                         // Fast2Sum, ah >= 0.69 > dh
    return bh + (bl + (al + ((dl - 0.5 * zl) + (r - x * y))));   // cos(x+y) - cos(x) ~ -y sin x ~ -x y
}
// argument reduction: x = n*(pi/2) + (y0 + y1), returns n mod 4 ... (n as int)
PT_DM int dm_rem_pio2(double x, double& y0, double& y1) {
    const double invpio2 = 6.36619772367581382433e-01, pio2_1 = 1.57079632673412561417e+00, pio2_1t = 6.07710050650619224932e-11,
                 pio2_2 = 6.07710050630396597660e-11, pio2_2t = 2.02226624879595063154e-21, pio2_3 = 2.02226624871116645580e-21,
                 pio2_3t = 8.47842766036889956997e-32;
    int32_t hx = dm_hi(x);
    int32_t ix = hx & 0x7fffffff;
    if (ix <= 0x3fe921fb) {   // |x| <= pi/4
        y0 = x;
        y1 = 0.0;
        return 0;
    }
    if (ix >= 0x7ff00000) {   // inf / nan
        y0 = y1 = x - x;
        return 0;
    }
    double ax = dm_abs(x);
    if (ix > 0x413921fb) {
        // |x| > 2^19*pi/2: outside the renderer's domain. Deterministic, reduced-accuracy path.
        const double two_pi = 6.28318530717958647692528676655900577;
        double q = ax / two_pi;
        double fq = (double)(int64_t)q;   // q < 2^63 assumed; beyond that precision is gone anyway
        if (!(q < 9.0e18)) {
            y0 = y1 = dm_nan();
            return 0;
        }
        ax = ax - fq * two_pi;
        if (ax < 0.0) ax = 0.0;
    }
    // medium size: Cody-Waite with up to three 33-bit pieces of pi/2
    int32_t n = (int32_t)(ax * invpio2 + 0.5);
    double fn = (double)n;
    double r = dm_fma(-fn, pio2_1, ax);   // fn * pio2_1 is exact (33-bit constant), fused only to save an instruction
    double w = fn * pio2_1t;
    int32_t j = (dm_hi(ax) & 0x7fffffff) >> 20;
    y0 = r - w;
    int32_t i = j - ((dm_hi(y0) >> 20) & 0x7ff);
    if (i > 16) {   // 2nd iteration needed, good to 118 bits
        double t = r;
        w = fn * pio2_2;
        r = t - w;
        w = fn * pio2_2t - ((t - r) - w);
        y0 = r - w;
        i = j - ((dm_hi(y0) >> 20) & 0x7ff);
        if (i > 49) {   // 3rd iteration, 151 bits
            t = r;
            w = fn * pio2_3;
            r = t - w;
            w = fn * pio2_3t - ((t - r) - w);
            y0 = r - w;
        }
    }
    y1 = (r - y0) - w;
    if (hx < 0) {
        y0 = -y0;
        y1 = -y1;
        return -n;
    }
    return n;
}
PT_DM void sincos(double x, double& s, double& c) {
    double y0, y1;
    const int n = dm_rem_pio2(x, y0, y1);   // |x| <= pi/4: n = 0, y0 = x, y1 = 0 — ONE copy of the kernels serves every lane of a wave
    const double ks = dm_ksin(y0, y1), kc = dm_kcos(y0, y1);
    const double a = (n & 1) ? kc : ks, b = (n & 1) ? ks : kc;   // quadrant: 0 (s, c)  1 (c, -s)  2 (-s, -c)  3 (-c, s)
    s = (n & 2) ? -a : a;
    c = ((n + 1) & 2) ? -b : b;
}
PT_DM double sin(double x) { double s, c; sincos(x, s, c); return s; }
PT_DM double cos(double x) { double s, c; sincos(x, s, c); return c; }

// ---- acos -----------------------------------------------------------------------------
// fdlibm's e_acos.c rational approximation R(z) = p(z)/q(z) of (asin(s) - s)/s, z = s^2, with its three argument ranges
// folded into ONE straight-line form (a wave of rays spans all three, and every branch a wave takes costs every lane):
//   |x| <  0.5 : z = x^2,          acos = pi/2 - (x + x R)
//   |x| >= 0.5 : z = (1 - |x|)/2,  s = sqrt(z),  acos = 2 asin(s) (x > 0)  or  pi - 2 asin(s) (x < 0),
//                asin(s_true) - s = s R + e/(2s) with e = z - s^2 exactly (fma), the two quotients over one denominator:
//                (2 z p + e q) / (2 s q)
// so every lane evaluates the two polynomials once and divides once. <= 1 ulp from glibc on [-1, 1] (tests/test_detmath.py).
PT_DM double acos(double x) {
    const double pi = 3.14159265358979311600e+00, pio2_hi = 1.57079632679489655800e+00, pio2_lo = 6.12323399573676603587e-17,
                 pS0 = 1.66666666666666657415e-01, pS1 = -3.25565818622400915405e-01, pS2 = 2.01212532134862925881e-01,
                 pS3 = -4.00555345006794114027e-02, pS4 = 7.91534994289814532176e-04, pS5 = 3.47933107596021167570e-05,
                 qS1 = -2.40339491173441421878e+00, qS2 = 2.02094576023350569471e+00, qS3 = -6.88283971605453293030e-01,
                 qS4 = 7.70381505559019352791e-02;
    const double ax = dm_abs(x);
    if (!(ax <= 1.0)) return dm_nan();   // |x| > 1 or NaN
    const bool big = ax >= 0.5;
    const double z = big ? (1.0 - ax) * 0.5 : x * x;
    const double p = z * dm_fma(z, dm_fma(z, dm_fma(z, dm_fma(z, dm_fma(z, pS5, pS4), pS3), pS2), pS1), pS0);
    const double q = dm_fma(z, dm_fma(z, dm_fma(z, dm_fma(z, qS4, qS3), qS2), qS1), 1.0);
    const double s = big ? dm_sqrt(z) : 0.0;
    const double e = dm_fma(-s, s, z);
    const double num = big ? dm_fma(2.0 * z, p, e * q) : p;
    double den = big ? 2.0 * s * q : q;
    if (den == 0.0) den = 1.0;           // x = +-1: s = 0 and num = 0
    const double r = num / den;
    if (!big) return pio2_hi - (x - dm_fma(-x, r, pio2_lo));
    return x > 0.0 ? 2.0 * (s + r) : pi - 2.0 * (s + (r - pio2_lo));
}

// ---- atan / atan2 ---------------------------------------------------------------------
// atan2 in one straight-line form with ONE division (fdlibm's e_atan2.c + s_atan.c take a quotient y/x, then one of four
// reductions — three with a second quotient — and a wave of directions takes them all). With a = min(|x|,|y|),
// b = max(|x|,|y|) the angle of (b, a) is in [0, pi/4]; past tan(pi/8) it is taken relative to pi/4:
//   atan(a/b) = pi/4 + atan((a - b)/(a + b)),
// so t = num/den has |t| <= tan(pi/8) < 0.4375, the range of s_atan.c's odd polynomial aT[]. Then the octant is undone:
// K + sigma r with a two-term K = 0, pi/2 or pi, then the sign of y. Signed zeros follow IEEE
// (atan2(+-0, -0) = +-pi); NaN in, NaN out. <= 1 ulp from glibc, 2 ulp in narrow bands around |y/x| = tan(pi/8) (tests/test_detmath.py).
PT_DM double atan2(double y, double x) {
    const double pio4_hi = 7.85398163397448278999e-01, pio4_lo = 3.06161699786838301793e-17, pio2_hi = 1.57079632679489655800e+00,
                 pio2_lo = 6.12323399573676603587e-17, pi_hi = 3.1415926535897931160E+00, pi_lo = 1.2246467991473531772E-16,
                 tan_pio8 = 4.14213562373095034188e-01;
    const double aT0 = 3.33333333333329318027e-01, aT1 = -1.99999999998764832476e-01, aT2 = 1.42857142725034663711e-01,
                 aT3 = -1.11111104054623557880e-01, aT4 = 9.09088713343650656196e-02, aT5 = -7.69187620504482999495e-02,
                 aT6 = 6.66107313738753120669e-02, aT7 = -5.83357013379057348645e-02, aT8 = 4.97687799461593236017e-02,
                 aT9 = -3.65315727442169155270e-02, aT10 = 1.62858201153657823623e-02;
    if (x != x || y != y) return x + y;   // NaN
    const double ax = dm_abs(x), ay = dm_abs(y);
    const bool sw = ay > ax;
    const double a = sw ? ax : ay, b = sw ? ay : ax;
    bool far = a > tan_pio8 * b;
    double num = far ? a - b : a;
    double den = far ? b + a : b;
    if (a == b) {                        // the diagonal (also inf/inf) and the origin (0/0): exact
        far = b != 0.0;
        num = 0.0;
        den = 1.0;
    }
    const double t = num / den;
    const double z = t * t;
    const double w = z * z;
    const double s1 = z * dm_fma(w, dm_fma(w, dm_fma(w, dm_fma(w, dm_fma(w, aT10, aT8), aT6), aT4), aT2), aT0);
    const double s2 = w * dm_fma(w, dm_fma(w, dm_fma(w, dm_fma(w, aT9, aT7), aT5), aT3), aT1);
    double r = far ? pio4_hi - (dm_fma(t, s1 + s2, -pio4_lo) - t) : dm_fma(-t, s1 + s2, t);
    // octant: angle = K + sigma r with K = 0, pi/2, pi/2, pi and sigma = +, -, +, - for (x >= 0, |y| <= |x|), (x >= 0, |y| > |x|),
    // (x < 0, |y| > |x|), (x < 0, |y| <= |x|): one two-term constant, one final rounding
    const bool xneg = (int64_t)dm_bits(x) < 0;
    const double k_hi = sw ? pio2_hi : xneg ? pi_hi : 0.0, k_lo = sw ? pio2_lo : xneg ? pi_lo : 0.0;
    r = k_hi + ((sw != xneg ? -r : r) + k_lo);
    return (int64_t)dm_bits(y) < 0 ? -r : r;
}

// ---- log / log2 -----------------------------------------------------------------------
PT_DM double log(double x) {
    const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10, two54 = 1.80143985094819840000e+16,
                 Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01, Lg3 = 2.857142874366239149e-01,
                 Lg4 = 2.222219843214978396e-01, Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
                 Lg7 = 1.479819860511658591e-01;
    int32_t hx = dm_hi(x);
    uint32_t lx = dm_lo(x);
    int32_t k = 0;
    if (hx < 0x00100000) {   // x < 2^-1022
        if ((((uint32_t)hx & 0x7fffffffu) | lx) == 0) return -two54 / 0.0;   // log(+-0) = -inf
        if (hx < 0) return dm_nan();                                        // log(-#) = NaN
        k -= 54;
        x *= two54;
        hx = dm_hi(x);
    }
    if (hx >= 0x7ff00000) return x + x;
    k += (hx >> 20) - 1023;
    hx &= 0x000fffff;
    int32_t i = (hx + 0x95f64) & 0x100000;
    x = dm_words(hx | (i ^ 0x3ff00000), dm_lo(x));   // normalise x or x/2
    k += (i >> 20);
    double f = x - 1.0;
    double dk = (double)k;
    if ((0x000fffff & (2 + hx)) < 3) {   // |f| < 2^-20
        if (f == 0.0) {
            if (k == 0) return 0.0;
            return dk * ln2_hi + dk * ln2_lo;
        }
        double R = f * f * (0.5 - 0.33333333333333333 * f);
        if (k == 0) return f - R;
        return dk * ln2_hi - ((R - dk * ln2_lo) - f);
    }
    double s = f / (2.0 + f);
    double z = s * s;
    i = hx - 0x6147a;
    double w = z * z;
    int32_t j = 0x6b851 - hx;
    double t1 = w * dm_fma(w, dm_fma(w, Lg6, Lg4), Lg2);
    double t2 = z * dm_fma(w, dm_fma(w, dm_fma(w, Lg7, Lg5), Lg3), Lg1);
    i |= j;
    double R = t2 + t1;
    if (i > 0) {
        double hfsq = 0.5 * f * f;
        if (k == 0) return f - (hfsq - s * (hfsq + R));
        return dk * ln2_hi - ((hfsq - (s * (hfsq + R) + dk * ln2_lo)) - f);
    }
    if (k == 0) return f - s * (f - R);
    return dk * ln2_hi - ((s * (f - R) - dk * ln2_lo) - f);
}
PT_DM double log2(double x) { return log(x) / 6.93147180559945286227e-01; }

// ---- exp / pow ------------------------------------------------------------------------
PT_DM double exp(double x) {
    const double o_threshold = 7.09782712893383973096e+02, u_threshold = -7.45133219101941108420e+02,
                 ln2HI = 6.93147180369123816490e-01, ln2LO = 1.90821492927058770002e-10, invln2 = 1.44269504088896338700e+00,
                 P1 = 1.66666666666666019037e-01, P2 = -2.77777777770155933842e-03, P3 = 6.61375632143793436117e-05,
                 P4 = -1.65339022054652515390e-06, P5 = 4.13813679705723846039e-08, twom1000 = 9.33263618503218878990e-302;
    uint32_t hx = (uint32_t)dm_hi(x);
    int xsb = (int)((hx >> 31) & 1u);
    hx &= 0x7fffffffu;
    double hi = 0.0, lo = 0.0;
    int32_t k = 0;
    if (hx >= 0x40862E42u) {   // |x| >= 709.78
        if (hx >= 0x7ff00000u) {
            if (((hx & 0xfffffu) | dm_lo(x)) != 0) return x + x;   // NaN
            return xsb == 0 ? x : 0.0;
        }
        if (x > o_threshold) return 1.0e300 * 1.0e300;
        if (x < u_threshold) return twom1000 * twom1000;
    }
    if (hx > 0x3fd62e42u) {   // |x| > 0.5 ln2
        if (hx < 0x3FF0A2B2u) {
            hi = x - (xsb ? -ln2HI : ln2HI);
            lo = xsb ? -ln2LO : ln2LO;
            k = 1 - xsb - xsb;
        } else {
            k = (int32_t)(invln2 * x + (xsb ? -0.5 : 0.5));
            double t = (double)k;
            hi = x - t * ln2HI;
            lo = t * ln2LO;
        }
        x = hi - lo;
    } else if (hx < 0x3e300000u) {   // |x| < 2^-28
        return 1.0 + x;
    }
    double t = x * x;
    double c = dm_fma(-t, dm_fma(t, dm_fma(t, dm_fma(t, dm_fma(t, P5, P4), P3), P2), P1), x);
    if (k == 0) return 1.0 - ((x * c) / (c - 2.0) - x);
    double y = 1.0 - ((lo - (x * c) / (2.0 - c)) - hi);
    if (k >= -1021) return dm_words(dm_hi(y) + (int32_t)((uint32_t)k << 20), dm_lo(y));
    y = dm_words(dm_hi(y) + (int32_t)((uint32_t)(k + 1000) << 20), dm_lo(y));
    return y * twom1000;
}
// 2^(th + tl) for |th| < 1000, rounded once: e^r = 1 + r + r^2/2 + r^3 E(r) with r = (th - n) ln2 in double-double
// (E: degree-11 Chebyshev fit at 120 digits, tools/make_detmath_coeffs.py).
PT_DM double dm_exp2_dd(double th, double tl) {
    const double LN2H = 0x1.62e42fefa39efp-1, LN2L = 0x1.abc9e3b39803fp-56;
    const double E0 = 0x1.5555555555555p-3, E1 = 0x1.5555555555555p-5, E2 = 0x1.1111111111111p-7, E3 = 0x1.6c16c16c16c17p-10,
                 E4 = 0x1.a01a01a0196aep-13, E5 = 0x1.a01a01a019b64p-16, E6 = 0x1.71de3a5aa6f3ap-19, E7 = 0x1.27e4fb7a271ecp-22,
                 E8 = 0x1.ae642c871071fp-26, E9 = 0x1.1eed7a04130cdp-29, E10 = 0x1.61bfa26897079p-33, E11 = 0x1.94328811e7eb4p-37;
    const double fn = (double)(int64_t)(th + (th < 0.0 ? -0.5 : 0.5));
    const double f = th - fn;                                 // exact, |f| <= 1/2
    const double rh = f * LN2H;
    const double rl = dm_fma(f, LN2H, -rh) + (f * LN2L + tl * LN2H);
    double e = dm_fma(E11, rh, E10);
    e = dm_fma(e, rh, E9);
    e = dm_fma(e, rh, E8);
    e = dm_fma(e, rh, E7);
    e = dm_fma(e, rh, E6);
    e = dm_fma(e, rh, E5);
    e = dm_fma(e, rh, E4);
    e = dm_fma(e, rh, E3);
    e = dm_fma(e, rh, E2);
    e = dm_fma(e, rh, E1);
    e = dm_fma(e, rh, E0);
    const double ah = rh * rh;
    const double al = dm_fma(rh, rh, -ah) + (2.0 * rh) * rl;  // r^2 = ah + al
    const double s3 = (ah * rh) * e;
    const double uh = 1.0 + rh;
    const double ul = (1.0 - uh) + rh;                        // Fast2Sum, 1 >= |rh|
    const double s2h = 0.5 * ah;
    const double vh = uh + s2h;
    const double vl = (uh - vh) + s2h;                        // Fast2Sum, uh >= 0.65 > s2h
    const double m = vh + (vl + (ul + (rl + (0.5 * al + s3))));
    return dm_words(dm_hi(m) + (int32_t)((uint32_t)(int32_t)fn << 20), dm_lo(m));   // m in [0.70, 1.42]: the result stays normal
}
// x > 0 only (the renderer's single use: GTR1 sampling, pow(0.25^2, 1 - e1), sampling.rs:132). A base that is an exact
// power of two — the renderer's always is — goes through the once-rounded exp2 above (the direction it yields is sampled,
// so every last-bit difference from libm can flip a checker cell at the next hit); other bases through exp(y log x).
PT_DM double pow(double x, double y) {
    if (!(x > 0.0)) return dm_nan();
    const uint64_t bx = dm_bits(x);
    const int32_t ex = (int32_t)((bx >> 52) & 0x7ff);
    if ((bx & 0x000FFFFFFFFFFFFFull) == 0 && ex != 0 && ex != 0x7ff) {
        const double k = (double)(ex - 1023);
        const double th = k * y;
        if (dm_abs(th) < 1000.0) return dm_exp2_dd(th, dm_fma(k, y, -th));
    }
    return exp(y * log(x));
}

}  // namespace detmath
