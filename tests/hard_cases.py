"""Directed hard cases for the fast Euler step's arithmetic (cv_device.h: div_with_recip, sqrt_and_rsqrt) -- TEST
INFRASTRUCTURE.  Everything here is exact integer / rational arithmetic; no floating-point result is trusted.

Division.  For doubles n, d the quotient n/d is never exactly half-way between two doubles, but it can come within one
quantum of it.  Take D odd in [2^52, 2^53) and an odd j; M = -j D^-1 mod 2^54 is odd, and when it falls in [2^53, 2^54)
N = (D M + j) / 2^54 is an integer below 2^53, hence a double, with

    N / D = M / 2^54 + j / (2^54 D):

M / 2^54 is the midpoint of the neighbouring doubles (M - 1) / 2^54 and (M + 1) / 2^54 of [1/2, 1), and the quotient
misses it by j / (2^54 D), i.e. by |j| 2^52 / N quanta of 2^-106 relative.  RN(N / D) = (M + sign(j)) / 2^54.

Square root.  For an odd M in [2^53, 2^54) and an integer X of 53 bits with X 2^E = M^2 + j (E = 54 or 55, so j = 7 mod 8),
sqrt(X 2^(E-2)) = M/2 + j / (4 M) + ...: M/2 is the midpoint of the integers (M - 1)/2 and (M + 1)/2 (doubles with ulp 1),
missed by j / (4 M) ulp (relative j / (2 M^2) >= 2^-109 |j|).  The four square roots of -j modulo 2^E come from
Hensel lifting.

div_with_recip_model is the exact-arithmetic model of the three operations of div_with_recip (every fma = ONE rounding
of the exact value)."""
from fractions import Fraction

import numpy as np


def _rn(fr):
    """round-to-nearest-even of an exact rational to a double (int / int true division is correctly rounded)"""
    return fr.numerator / fr.denominator


def division_hard_cases(rng, count, j_values=(1, -1, 3, -3, 5, -5, 7, -7, 9, -9, 11, -11, 13, -13, 15, -15)):
    """returns arrays (n, d, j, expect, quanta): n / d misses a rounding boundary by quanta = |j| 2^52 / N units of
    2^-106 relative (on the side of sign(j)), expect = RN(n / d) by construction; mantissas random, exponents scattered
    over +-60 binades"""
    n, d, jj, ex, qu = [], [], [], [], []
    while len(n) < count:
        D = int(rng.integers(1 << 52, 1 << 53)) | 1
        j = int(j_values[len(n) % len(j_values)])
        M = (-j * pow(D, -1, 1 << 54)) % (1 << 54)
        if M < (1 << 53):
            continue
        N, rem = divmod(D * M + j, 1 << 54)
        assert rem == 0 and 0 < N < (1 << 53)
        en, ed = int(rng.integers(-60, 61)), int(rng.integers(-60, 61))
        n.append(float(np.ldexp(float(N), en)))
        d.append(float(np.ldexp(float(D), ed)))
        jj.append(j)
        ex.append(float(np.ldexp(float(M + (1 if j > 0 else -1)), en - ed - 54)))
        qu.append(abs(j) * float(1 << 52) / float(N))
    return np.array(n), np.array(d), np.array(jj, dtype=np.int64), np.array(ex), np.array(qu)


def _sqrts_mod_pow2(c, E):
    """the four x with x^2 = c (mod 2^E), c = 1 (mod 8), E >= 3"""
    assert c % 8 == 1
    x = 1
    for k in range(3, E):          # invariant: x^2 = c (mod 2^k); a solution mod 2^k is fixed up to +-, +2^(k-1)
        if (x * x - c) % (1 << (k + 1)):
            x += 1 << (k - 1)
    mod = 1 << E
    return sorted({x % mod, (-x) % mod, (x + (mod >> 1)) % mod, (-x + (mod >> 1)) % mod})


def sqrt_hard_cases(j_max=400):
    """returns arrays (x, j, expect): sqrt(x) within |j| / (4 M) ulp of a rounding boundary, expect = RN(sqrt(x));
    every j = 7 (mod 8) with |j| <= j_max, both exponent parities"""
    xs, js, ex = [], [], []
    for j in range(-j_max, j_max + 1):
        if j % 8 != 7:
            continue
        for E in (54, 55):
            for M in _sqrts_mod_pow2((-j) % (1 << E), E):
                if not ((1 << 53) <= M < (1 << 54)):
                    continue
                X, rem = divmod(M * M + j, 1 << E)
                if rem or not ((1 << 52) <= X < (1 << 53)):
                    continue
                # sqrt(X 2^(E-2)) = M/2 + ...; scaled by an even power of two to stay in a comfortable range
                xs.append(float(np.ldexp(float(X), E - 2 - 106)))
                js.append(j)
                ex.append(float(np.ldexp(float(M + (1 if j > 0 else -1)), -1 - 53)))
    return np.array(xs), np.array(js, dtype=np.int64), np.array(ex)


def step_ulps(y, u):
    """y moved by u units in the last place (same sign, finite)"""
    return (np.ascontiguousarray(y, dtype=np.float64).view(np.int64) + np.asarray(u, dtype=np.int64)).view(np.float64)


def div_with_recip_model(n, d, y):
    """exact model of cv_device.h div_with_recip: q0 = RN(n y); rem = RN(n - d q0); q = RN(q0 + rem y)"""
    out = np.empty(len(n))
    for i, (a, b, c) in enumerate(zip(n.tolist(), d.tolist(), y.tolist())):
        q0 = a * c
        rem = _rn(Fraction(a) - Fraction(b) * Fraction(q0))
        out[i] = _rn(Fraction(q0) + Fraction(rem) * Fraction(c))
    return out


def recip_error_units(d, y):
    """kappa = (d y - 1) / 2^-53, exactly (as a float of the exact rational): how far y is from 1/d"""
    return np.array([float((Fraction(b) * Fraction(c) - 1) * (1 << 53)) for b, c in zip(d.tolist(), y.tolist())])


def recip_error_units_fast(d, y):
    """the same in extended precision (x87 long double: 64-bit significand, good to 2^-11 of a unit), vectorised"""
    return np.asarray((d.astype(np.longdouble) * y.astype(np.longdouble) - 1) * np.longdouble(2.0 ** 53), dtype=np.float64)


def ulp_distance(a, b):
    return np.abs(np.ascontiguousarray(a).view(np.int64) - np.ascontiguousarray(b).view(np.int64))


def ulp_of(q):
    q = np.abs(q)
    return np.ldexp(1.0, np.frexp(q)[1] - 53)


def two_prod(a, b):
    """a b = p + e exactly (Veltkamp / Dekker; no overflow or underflow in the ranges used here)"""
    def split(x):
        c = 134217729.0 * x
        hi = c - (c - x)
        return hi, x - hi
    p = a * b
    ah, al = split(a)
    bh, bl = split(b)
    return p, ((ah * bh - p) + ah * bl + al * bh) + al * bl


def gap_over_ulp(n, d, y, q_ieee, rem_recorded, eps_recorded):
    """|q0 + rem y - n/d| / ulp(n/d) for q0 = RN(n y), rem = RN(n - d q0): the share of the boundary spacing inside which
    div_with_recip's last rounding can go the wrong way.  d (q0 + rem y - n/d) = -rem (1 - d y) - (exact remainder - rem)."""
    q0 = n * y
    p, e = two_prod(d, q0)
    t = n - p                                   # exact (n and p agree to a few ulp)
    s = t - e
    bb = s - t
    err = (t - (s - bb)) + (-e - bb)            # exact remainder = s + err, s = RN(...) = the recorded remainder
    with np.errstate(invalid="ignore", divide="ignore"):
        g = (-s * eps_recorded - err) / d       # signed: (value before the last rounding) - n/d
        out = np.abs(g) / ulp_of(q_ieee)
    return np.where(np.isfinite(out) & (q_ieee != 0.0), out, 0.0), float((s != rem_recorded).sum()), g
