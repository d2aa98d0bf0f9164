"""Accuracy of curvis_amd/csrc/cv_math.h (the deterministic elementary functions shared by the
kernels and the oracle's CVO_CV flavour) against mpmath: every function < 1 ulp (atan2 < 1.5).
Runs on the host twin; tests/test_gpu_parity.py checks device == host bit for bit."""
import math
import random

import numpy as np
import pytest

import common

mpmath = pytest.importorskip("mpmath")
from mpmath import mp, mpf  # noqa: E402

mp.prec = 300
N = 4000
OPS = {"sin": 0, "cos": 1, "atan": 2, "acos": 3, "log": 4}
REF = {"sin": mpmath.sin, "cos": mpmath.cos, "atan": mpmath.atan, "acos": mpmath.acos, "log": mpmath.log}


def max_ulp_error(name, xs):
    got = common.twin_math(OPS[name], np.array(xs, dtype=np.float64))
    worst = 0.0
    for x, g in zip(xs, got):
        ex = REF[name](mpf(x))
        u = math.ulp(abs(float(ex))) if ex != 0 else 5e-324
        worst = max(worst, float(abs(mpf(float(g)) - ex) / u))
    return worst


def near_multiples(rng, kmax, dist_exp):
    out = []
    for _ in range(N // 4):
        k = rng.randint(1, kmax)
        base = float(mpf(k) * mpmath.pi / 2)
        d = math.ldexp(rng.uniform(1, 2), rng.randint(*dist_exp)) * rng.choice([-1, 1])
        out.append(base + d)
        out.append(base)
        out.append(math.nextafter(base, math.inf))
        out.append(math.nextafter(base, -math.inf))
    return out


@pytest.mark.parametrize("name", ["sin", "cos"])
def test_sincos_accuracy(name):
    rng = random.Random(5)
    sets = [
        [rng.uniform(-0.79, 0.79) for _ in range(N)],
        [rng.uniform(-4, 7) for _ in range(N)],
        [rng.uniform(-6.25, 6.25) for _ in range(N)],                      # main path (two 44-bit pieces of pi/64)
        [rng.choice([-1, 1]) * (6.25 + rng.uniform(-1e-9, 1e-9)) for _ in range(N // 8)],  # main <-> other path at 6.25
        [rng.uniform(-1024, 1024) for _ in range(N)],                      # fdlibm-style pi/2 reduction first
        [rng.uniform(1024, 1.1e6) * rng.choice([-1, 1]) for _ in range(N)],  # Cody-Waite 33-bit pieces
        [rng.choice([-1, 1]) * math.ldexp(rng.uniform(1, 2), rng.randint(20, 1023)) for _ in range(N)],  # Payne-Hanek
        near_multiples(rng, 3, (-45, -10)),                                # main <-> other path switch at 2^-20
        near_multiples(rng, 650, (-45, -10)),
        near_multiples(rng, 10 ** 15, (-30, 5)),
        [10.0 ** rng.uniform(-320, -1) for _ in range(N // 4)],
    ]
    for xs in sets:
        assert max_ulp_error(name, xs) < 0.85


@pytest.mark.parametrize("name,gen", [
    ("atan", lambda r: r.uniform(-700, 700)), ("atan", lambda r: r.uniform(-3, 3)),
    ("atan", lambda r: r.choice([-1, 1]) * 10 ** r.uniform(-12, 25)),
    ("acos", lambda r: r.uniform(-1, 1)), ("acos", lambda r: 1 - 10 ** r.uniform(-16, -1)),
    ("acos", lambda r: -1 + 10 ** r.uniform(-16, -1)),
    ("log", lambda r: r.uniform(1, 1e6)), ("log", lambda r: r.uniform(0.5, 2)),
    ("log", lambda r: 10 ** r.uniform(-320, 300)),
])
def test_other_functions_accuracy(name, gen):
    rng = random.Random(9)
    assert max_ulp_error(name, [gen(rng) for _ in range(N)]) < 0.95


def test_atan2_accuracy_and_quadrants():
    rng = random.Random(2)
    ys = np.array([rng.uniform(-5, 5) for _ in range(N)])
    xs = np.array([rng.uniform(-5, 5) for _ in range(N)])
    got = common.twin_math(5, ys, xs)
    worst = 0.0
    for y, x, g in zip(ys, xs, got):
        ex = mpmath.atan2(mpf(y), mpf(x))
        worst = max(worst, float(abs(mpf(float(g)) - ex) / math.ulp(abs(float(ex)))))
    assert worst < 0.7   # fdlibm-style atan2 measured 1.45; with the remainder correction and one final rounding 0.65
    sp = common.twin_math(5, np.array([0.0, -0.0, 1.0, -1.0, 0.0, 1.0]), np.array([-1.0, -1.0, 0.0, 0.0, 1.0, 1.0]))
    assert list(sp) == [math.pi, -math.pi, math.pi / 2, -math.pi / 2, 0.0, math.pi / 4]


def test_equatorial_theta_shortcut():
    """theta == fl(pi/2) returns at once (the equatorial rays hold it for ever): the values are the correctly
    rounded ones, which is also what the general reduction gives for this argument (the golden frames, whose
    middle pixel row evaluates it ~2000 times per ray, did not change when the shortcut went in)."""
    x = np.array([math.pi / 2])
    s, c = common.twin_math(0, x)[0], common.twin_math(1, x)[0]
    assert s == 1.0
    assert c == float(mpmath.cos(mpf(math.pi / 2))) == 6.123233995736766e-17
    for name, op in (("sin", 0), ("cos", 1)):   # neighbours take the general route and stay accurate
        assert max_ulp_error(name, [math.nextafter(math.pi / 2, 0.0), math.nextafter(math.pi / 2, 4.0)]) < 0.85


def test_special_values_match_ieee_semantics():
    t = common.twin_math
    assert np.isnan(t(0, np.array([np.inf, np.nan]))).all() and np.isnan(t(1, np.array([-np.inf]))).all()
    assert list(t(0, np.array([0.0, -0.0]))) == [0.0, -0.0] and np.signbit(t(0, np.array([-0.0])))[0]
    assert list(t(1, np.array([0.0, math.pi / 2, math.pi]))) == [1.0, 6.123233995736766e-17, -1.0]
    assert list(t(3, np.array([1.0, -1.0, 0.0]))) == [0.0, math.pi, math.pi / 2]
    assert np.isnan(t(3, np.array([1.0000000000000002, -1.5, np.nan]))).all()
    assert list(t(4, np.array([1.0, 0.0, np.inf]))) == [0.0, -np.inf, np.inf] and np.isnan(t(4, np.array([-1.0])))[0]
    assert list(t(2, np.array([np.inf, -np.inf, 0.0]))) == [math.pi / 2, -math.pi / 2, 0.0]


def test_table_row_boundaries():
    """atan / log on and next to every row boundary of their tables (and the branch switches)."""
    at, lg = common.table_edge_inputs()
    assert max_ulp_error("atan", [float(v) for v in at]) < 0.8
    assert max_ulp_error("log", [float(v) for v in lg]) < 0.62


def loop_arguments(metric_name, n_rays=96, iterations=2200):
    """the arguments the Euler loop really hands to the elementary functions: theta of every step of a fan of rays
    from the default camera (and, for the Interstellar metric, x = 2(|l| - a)/(pi m) and 1 + x^2 of r(l))"""
    import oracle_lib as O
    om = O.ellis(1.0) if metric_name == "ellis" else O.interstellar(0.1, 1e-4, 1.0)
    oc = O.camera(res=(16, 6))
    th, ls = [], []
    for px in range(16):
        for py in range(6):
            d = np.zeros(3)
            O.lib().cvo_camera_outward_world(O.C.byref(oc), px, py, O._dp(d))
            tr = O.photon_trajectory(O.CV, om, (0.0, 5.0, math.pi / 2, 0.0), tuple(d), iterations, 0.05)
            keep = np.abs(tr[:, 1]) <= 100.0
            th.append(tr[keep, 2])
            ls.append(tr[keep, 1])
    th, ls = np.concatenate(th), np.concatenate(ls)
    if metric_name == "ellis":
        return {"sin": th, "cos": th}
    x = 2.0 * (np.abs(ls) - 1e-4) / (math.pi * 0.1)
    x = x[np.abs(ls) > 1e-4]
    return {"sin": th, "cos": th, "atan": x, "log": 1.0 + x * x}


@pytest.mark.parametrize("metric_name", ["ellis", "interstellar"])
def test_loop_arguments_against_glibc_and_binary128(metric_name):
    """The bit-exact GPU-vs-oracle tests share cv_math.h between both sides, so they cannot see an error IN it.
    This covers that blind spot with two independent references on the arguments the loop produces (not random
    ones): glibc (what a Linux build of the reference calls) and binary128 (libquadmath).  cv_math.h must stay
    below 0.8 ulp (sin, cos), 0.65 (atan, log) of the truth, and within one ulp of glibc."""
    import oracle_lib as O
    bounds = {"sin": 0.8, "cos": 0.8, "atan": 0.65, "log": 0.65}
    for name, xs in loop_arguments(metric_name).items():
        op = OPS[name]
        cv = common.twin_math(op, xs)                     # the product header, compiled for x86
        assert np.array_equal(cv.view(np.uint64), O.math_array(O.CV, op, xs).view(np.uint64))
        gl = O.math_array(O.LIBM, op, xs)
        e_cv, e_gl = O.quad_ulp_errors(op, xs, cv), O.quad_ulp_errors(op, xs, gl)
        assert e_cv.max() < bounds[name], (name, e_cv.max())
        assert e_gl.max() < 0.56                          # glibc 2.35 claims < 0.55 ulp for these
        same = cv == gl
        off = np.abs(cv[~same] - gl[~same]) / np.spacing(np.abs(gl[~same])) if (~same).any() else np.zeros(1)
        assert off.max() <= 1.0                           # never more than one ulp apart
        assert same.mean() > 0.95, (name, same.mean())
        print("%s/%s: %d arguments, identical to glibc %.4f %%, max error cv %.3f ulp, glibc %.3f ulp" % (
            metric_name, name, xs.size, 100 * same.mean(), e_cv.max(), e_gl.max()))
