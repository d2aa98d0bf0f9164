"""The fast Euler step's guarantee, measured instead of estimated (cv_device.h: div_with_recip, sqrt_and_rsqrt, the
shared reciprocals of ray_step_fast; reference operations: src/metrics.rs:223-297).

div_with_recip(n, d, y) = RN(q0 + rem y), q0 = RN(n y), rem = n - d q0, returns RN(n/d) unless a rounding boundary lies
within (n/d)(|kappa| + 1)^2 2^-106 of n/d, y = (1 + kappa 2^-53)/d (tools/gpu_fast_step_rounding.py has the derivation and the
rates measured on real rays: profiles/round5_fast_step_rounding.txt).  These tests go to the boundary on purpose:
operands whose exact quotient / root sits within a few quanta of 2^-106 of a rounding boundary (tests/hard_cases.py,
exact integer constructions), fed with reciprocals that are off by the amounts the step's own reciprocals are off by."""
import numpy as np
import pytest

import common
import hard_cases as H
import oracle_lib as O
import curvis_amd

pytestmark = pytest.mark.gpu


def test_div_with_recip_on_directed_hard_cases(gpu_ctx):
    """(i) a mis-rounding IS reachable: with y one ulp or more from RN(1/d) and n/d within <= 15 quanta ABOVE a boundary the
    quotient comes out one ulp low; (ii) it is confined: never more than ONE ulp, never on the other side of the boundary,
    and only as far out as eps (eps + eta) reaches; (iii) with y = RN(1/d) (what interstellar_x uses for 1/(pi m)) it cannot
    happen (Markstein); (iv) the device executes exactly the three roundings of the exact-arithmetic model."""
    rng = np.random.default_rng(11)
    n, d, j, expect, quanta = H.division_hard_cases(rng, 20000)
    assert np.array_equal(n / d, expect)                                   # the construction itself (x86 IEEE division)
    assert np.array_equal(gpu_ctx.selftest_math(6, n, d), expect)          # the device's IEEE division (strict step)
    y0 = 1.0 / d
    reached = {}
    for u in range(-4, 5):
        y = H.step_ulps(y0, u)
        got = gpu_ctx.selftest_math3(0, n, d, y)
        assert np.array_equal(got, common.twin_math3(0, n, d, y))          # device == x86 twin of the same source
        assert np.array_equal(got[:1500], H.div_with_recip_model(n[:1500], d[:1500], y[:1500]))
        bad = got != expect
        reached[u] = int(bad.sum())
        assert H.ulp_distance(got, expect).max() <= 1
        if u == 0:
            assert not bad.any(), "a correctly rounded reciprocal must give the correctly rounded quotient"
            continue
        kap = np.abs(H.recip_error_units_fast(d, y))                       # y = (1 +- kap 2^-53) / d
        assert not (bad & (j < 0)).any(), "the value before the last rounding is never above n/d for |eps| >= |eta|"
        assert (np.abs(got[bad]) < np.abs(expect[bad])).all()
        # a boundary `quanta` units of 2^-106 away is crossed only if the gap reaches it: eps (eps + eta) plus the rounding of
        # a remainder that no longer fits a double, together <= (kap + 1)^2 quanta
        assert ((kap[bad] + 1.0) ** 2 >= quanta[bad]).all()
    print("mis-rounded hard cases of 20000 by reciprocal offset u (ulp):", reached)
    assert reached[0] == 0 and all(reached[u] > 0 for u in (-4, -3, -2, -1, 1, 2, 3, 4))
    # ordinary operands: a reciprocal off by up to 8 ulp never shows
    a = rng.uniform(1.0, 2.0, 2_000_000) * 2.0 ** rng.integers(-40, 40, 2_000_000)
    b = rng.uniform(1.0, 2.0, 2_000_000) * 2.0 ** rng.integers(-40, 40, 2_000_000)
    yb = H.step_ulps(1.0 / b, rng.integers(-8, 9, b.size))
    assert np.array_equal(gpu_ctx.selftest_math3(0, a, b, yb), a / b)


def test_sqrt_and_rsqrt_on_directed_hard_cases(gpu_ctx):
    """square roots within |j| / (4 M) ulp of a boundary (exact constructions, every feasible |j| <= 2000): the device's IEEE
    sqrt gets them all; sqrt_and_rsqrt's single residual step is one ulp off only on arguments with j = -1 -- significands
    1 + 2^-51 (sqrt = 1 + 2^-52 - 2^-105...) and all-ones -- i.e. three significands out of 2^53 per pair of binades."""
    x, j, expect = H.sqrt_hard_cases(2000)
    assert len(x) > 600 and np.array_equal(np.sqrt(x), expect)
    assert np.array_equal(gpu_ctx.selftest_math(7, x), expect)
    root = gpu_ctx.selftest_math3(1, x)
    bad = root != expect
    print("sqrt_and_rsqrt: %d of %d hard cases one ulp off, j of those: %s" % (int(bad.sum()), len(x), sorted(set(j[bad].tolist()))))
    assert H.ulp_distance(root, expect).max() <= 1
    assert set(j[bad].tolist()) <= {-1}, "only the three j = -1 significands are within reach of a reciprocal root good to an ulp"
    # the reciprocal root it hands to the step: within 1.75 units of 2^-53 of 1/sqrt(x)
    y = gpu_ctx.selftest_math3(2, x)
    kap = np.array([float((H.Fraction(a) * H.Fraction(b) ** 2 - 1) * (1 << 52)) for a, b in zip(x.tolist(), y.tolist())])
    assert np.abs(kap).max() <= 1.75, np.abs(kap).max()     # measured: 1.50
    xr = np.random.default_rng(5).uniform(1.0, 4.0, 4_000_000)
    assert np.array_equal(gpu_ctx.selftest_math3(1, xr), np.sqrt(xr))


def test_reciprocal_through_refined_seed(gpu_ctx):
    """recip_refined and the -1/x of atan (cv_div_nr: seed, one third-order step, one Newton step = Markstein's
    correctly rounded reciprocal): equal to IEEE on random arguments and on all-ones significands (the theorem's one
    exception); recip_newton of a reciprocal that is an ulp off restores RN(1/d)."""
    rng = np.random.default_rng(6)
    ones = np.ldexp(float((1 << 53) - 1), np.arange(-104, -40))
    xs = np.concatenate([ones[ones >= 2.0], rng.uniform(2.0, 2000.0, 2_000_000)])
    got = gpu_ctx.selftest_math3(5, np.full_like(xs, -1.0), xs)
    off = got != -1.0 / xs
    print("cv_div_nr(-1, x): %d of %d differ from IEEE (all-ones significands among them: %d)" % (int(off.sum()), xs.size, int(off[:int((ones >= 2.0).sum())].sum())))
    assert H.ulp_distance(got, -1.0 / xs).max() <= 1 and not off[int((ones >= 2.0).sum()):].any()
    d = rng.uniform(1.0, 2.0, 1_000_000) * 2.0 ** rng.integers(-40, 40, 1_000_000)
    r = gpu_ctx.selftest_math3(4, d)
    assert H.ulp_distance(r, 1.0 / d).max() <= 1
    for u in (-1, 1):
        assert np.array_equal(gpu_ctx.selftest_math3(6, d, H.step_ulps(1.0 / d, u)), 1.0 / d)


@pytest.mark.parametrize("metric,l_cam", [("ellis", 5.0), ("ellis", 3.0), ("interstellar", 5.0), ("interstellar", 0.008)])
def test_shared_reciprocals_of_real_steps(gpu_ctx, metric, l_cam):
    """the reciprocals the step REALLY forms (recorded through the probe hook of ray_step_fast) on every Euler step of
    256 rays of the config's camera: how far each is from 1/d (kappa, in units of 2^-53) stays inside the envelope the
    documented rate is computed from, every quotient equals the IEEE quotient, the fast step's new state equals the strict
    step's bit for bit."""
    om, oc, pm, pc = common.scene(metric, res=(1920, 1080), pos=(0.0, l_cam, common.HALF_PI, 0.0))
    rng = np.random.default_rng(3)
    nr = 256
    px, py = rng.integers(0, 1920, nr), rng.integers(0, 1080, nr)
    dirs = np.zeros((nr, 3))
    for i in range(nr):
        O.lib().cvo_camera_outward_world(O.C.byref(oc), int(px[i]), int(py[i]), O._dp(dirs[i]))
    tr = gpu_ctx.compute_photon_trajectory(pm, np.tile([0.0, l_cam, common.HALF_PI, 0.0], (nr, 1)), dirs, 2300, 0.05)
    keep = np.cumprod(np.abs(tr[:, :, 1]) <= 100.0, axis=1).astype(bool)
    states = np.ascontiguousarray(tr[keep][:, [1, 2, 5, 6, 7]])
    quot, fast, strict, took = gpu_ctx.selftest_fast_step(pm, states)
    assert took.mean() > 0.98
    assert np.array_equal(fast.view(np.uint64), strict.view(np.uint64))
    # |kappa| per quotient, measured maxima (profiles/round5_fast_step_rounding.txt: Ellis 1.95 / 3.47 / 3.47 / 7.42 / 8.67 / 7.16,
    # Interstellar - / 5.93 / 6.04 / 9.00 / 14.10 / 12.02 -- its 1/r and 1/sin are themselves products of one seed) + margin
    envelope = {0: 2.5, 1: 4.5, 2: 4.5, 3: 9.0, 4: 10.5, 5: 9.0} if metric == "ellis" else {1: 7.5, 2: 7.5, 3: 11.0, 4: 17.0, 5: 15.0}
    seen, per_step = {}, 0.0
    for k in range(6):
        nn, d, y, qf, qi, rem, eps = (quot[:, k, i] for i in range(7))
        ok = took & (d == d)
        if not ok.any():
            continue
        kap = np.abs(eps[ok]) * 2.0 ** 53
        seen[curvis_amd.Context.FAST_STEP_QUOTIENTS[k]] = (round(float(kap.max()), 2), round(float(np.sqrt((kap ** 2).mean())), 2))
        assert kap.max() <= envelope[k], (k, kap.max())
        assert np.array_equal(qf[ok].view(np.uint64), qi[ok].view(np.uint64)), k
        p, mismatch, _ = H.gap_over_ulp(nn[ok], d[ok], y[ok], qi[ok], rem[ok], eps[ok])
        assert mismatch == 0
        if k != 5:                                       # the production kernels do not integrate phi
            per_step += p.sum() / len(states)
    print("%s l = %g: %d steps; |kappa| (max, rms) per quotient: %s; expected mis-rounded quotients per step %.3g" % (
        metric, l_cam, len(states), seen, per_step))
    assert (metric == "ellis") == ("r' = l/r" in seen)
    # THE documented rate: < 2^-49 per step Ellis, < 2^-48 Interstellar, i.e. ~4e-6 per 1080p frame and
    # ~1.5e-2 per full configs[4] render -- the exact gap of every recorded quotient over the spacing of the boundaries
    assert per_step < (2.0 ** -49 if metric == "ellis" else 2.0 ** -48), per_step


def _same(a, b):
    """bit-equal where both are numbers (signed zeros included), NaN where either is"""
    a, b = np.asarray(a), np.asarray(b)
    nan = np.isnan(a) | np.isnan(b)
    return bool(np.array_equal(np.isnan(a), np.isnan(b)) and np.array_equal(a.view(np.uint64)[~nan], b.view(np.uint64)[~nan]))


def test_shared_quotients_of_the_pixel_kernel(gpu_ctx):
    """The efficient renderer's per-pixel kernel forms nine of its twelve quotients from shared reciprocals (cv_device.h
    recip_chain / unit3 / div_const; reference operations: src/cameras.rs:150-172, src/systems.rs:498-506, src/images.rs:115-121).
    The shared y is the first half of the device's own division, so every quotient must BE the IEEE quotient: on operands
    whose exact quotient sits within a few quanta of 2^-106 of a rounding boundary, on ordinary ones, and -- through the guards --
    on zeros of either sign, subnormals, the exponent limits, infinities and NaN."""
    rng = np.random.default_rng(611)
    n, d, j, expect, quanta = H.division_hard_cases(rng, 20000)
    n, d = np.abs(n), np.abs(d)
    assert np.array_equal(gpu_ctx.selftest_math3(10, n, d), n / d)          # hard cases: numerator, denominator > 0
    a = rng.uniform(1.0, 2.0, 2_000_000) * 2.0 ** rng.integers(-299, 299, 2_000_000)
    b = rng.uniform(1.0, 2.0, 2_000_000) * 2.0 ** rng.integers(-299, 299, 2_000_000)
    assert np.array_equal(gpu_ctx.selftest_math3(10, a, b), a / b)          # the whole guarded range (and quotients far outside it)
    # what the kernel really divides: pixel indices by resolutions (no guard: an index is +0 or 1 .. 2^32) ...
    for res in (1.0, 2.0, 3.0, 144.0, 256.0, 1080.0, 1920.0, 2160.0, 3840.0, 4097.0, 65535.0, 4294967295.0):
        px = np.concatenate([np.arange(0.0, min(res, 8192.0)), np.floor(rng.uniform(0.0, res, 4096))])
        assert _same(gpu_ctx.selftest_math3(10, px, np.full_like(px, res)), px / res)
    # ... and angles by pi and 2 pi; zeros, tiny values, NaN and whatever else is no angle go to the `/` operator: results unchanged
    special = np.array([0.0, -0.0, 5e-324, -5e-324, 2.2250738585072014e-308, 1e-310, 2.0 ** -301, 2.0 ** -300, 2.0 ** 299, 2.0 ** 300,
                        1.7976931348623157e308, -1.0, -3.5, np.inf, -np.inf, np.nan, 1.0, 3.0, 8.0, np.nextafter(8.0, 0.0)])
    ang = np.concatenate([rng.uniform(0.0, 2.0 * np.pi, 1_000_000), np.arccos(1.0 - 2.0 ** -np.arange(1.0, 54.0)), [np.pi, 2.0 * np.pi],
                          2.0 ** -rng.uniform(0.0, 320.0, 100000), special])
    with np.errstate(all="ignore"):
        for c in (np.pi, 2.0 * np.pi):
            assert _same(gpu_ctx.selftest_math3(11, ang, np.full_like(ang, c)), ang / c)
    # v / |v|: ordinary vectors over the exponent range, components that are zeros of either sign or tiny, norms that leave the range
    m = 1_000_000
    v = [rng.uniform(-2.0, 2.0, m) * 2.0 ** rng.integers(-8, 8, m) * 2.0 ** rng.integers(-140, 141) for _ in range(3)]
    vs = [np.concatenate([v[k], rng.choice(special, 40000), rng.uniform(-1, 1, 40000) * 2.0 ** rng.integers(-1074, 1024, 40000).astype(float)])
          for k in range(3)]
    for k in range(3):                                                      # a third of the first block: one exact zero (the centre row / column of a frame)
        vs[k][k * 100000:(k + 1) * 100000:3] = 0.0
        vs[k][k * 100000 + 1:(k + 1) * 100000:3] = -0.0
    with np.errstate(all="ignore"):
        norm = np.sqrt((vs[0] * vs[0] + vs[1] * vs[1]) + vs[2] * vs[2])
        for k in range(3):
            assert _same(gpu_ctx.selftest_math3(7 + k, vs[0], vs[1], vs[2]), vs[k] / norm), k


def test_square_root_without_the_range_wrappers(gpu_ctx):
    """cv_device.h sqrt_plain (the efficient pixel kernel's three norms): the compiler's own chain without its scaling and class
    wrappers inside [2^-700, 2^700), the operator outside -- the IEEE root everywhere: on arguments whose root lies within 2^-106 of
    a rounding boundary, over the whole guarded range and beyond it, on zeros, negatives, infinities and NaN."""
    rng = np.random.default_rng(612)
    x, _, expect = H.sqrt_hard_cases(2000)
    assert np.array_equal(np.sqrt(x), expect)
    assert np.array_equal(gpu_ctx.selftest_math3(12, x), expect)
    a = rng.uniform(1.0, 4.0, 2_000_000) * 4.0 ** rng.integers(-500, 500, 2_000_000)
    assert np.array_equal(gpu_ctx.selftest_math3(12, a), np.sqrt(a))
    special = np.array([0.0, -0.0, 5e-324, 2.2250738585072014e-308, 1e-310, 2.0 ** -701, 2.0 ** -700, np.nextafter(2.0 ** -700, 0.0), 2.0 ** 699,
                        2.0 ** 700, np.nextafter(2.0 ** 700, 0.0), 1.7976931348623157e308, -1.0, -1e-320, np.inf, -np.inf, np.nan, 1.0, 2.0, 3.0, 4.0])
    with np.errstate(all="ignore"):
        assert _same(gpu_ctx.selftest_math3(12, special), np.sqrt(special))


def test_saturating_conversion_to_u32(gpu_ctx):
    """cv_device.h rust_as_u32 on the device is one v_cvt_u32_f64: it must be Rust's `as u32` (src/images.rs:115-121: truncation
    toward zero, NaN -> 0, negatives -> 0, 2^32 and above -> u32::MAX) on every class of value"""
    rng = np.random.default_rng(613)
    v = np.concatenate([rng.uniform(-10.0, 5000.0, 200000), rng.uniform(0.0, 2.0 ** 33, 200000), 2.0 ** rng.uniform(-1074, 1023, 100000),
                        -(2.0 ** rng.uniform(-1074, 1023, 100000)), np.arange(0.0, 4097.0), np.nextafter(np.arange(1.0, 4097.0), 0.0),
                        [0.0, -0.0, 0.5, 0.9999999999999999, 1.0, 4294967294.5, 4294967295.0, 4294967295.5, 4294967296.0, 1e300, -1e300,
                         np.inf, -np.inf, np.nan, 5e-324, -5e-324]])
    want = np.where(np.isnan(v), 0.0, np.clip(np.trunc(np.nan_to_num(v, nan=0.0, posinf=1e308, neginf=-1e308)), 0.0, 4294967295.0))
    assert np.array_equal(gpu_ctx.selftest_math3(13, v), want)
