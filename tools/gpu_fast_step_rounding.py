"""How often can the fast Euler step's shared-reciprocal quotients mis-round?  MEASURED on the device, written to
profiles/round5_fast_step_rounding.txt (python tools/gpu_fast_step_rounding.py > gpurun_out/fast_step_rounding.txt).

cv_device.h forms each of the step's quotients as  q0 = RN(n y); rem = n - d q0 (exact fma); q = RN(q0 + rem y)  with y a
PRODUCT of one refined 1/r and one refined 1/sin(theta).  Writing y = (1 + eps)/d and q0 = (n/d)(1 + eps)(1 + eta)
(|eta| <= 2^-53 the rounding of q0), the value that is rounded last is

    q0 + rem y = (n/d) (1 - eps (eps + eta)) + rho y,

rho = the rounding error of the remainder fma (zero while q0 is within an ulp of n/d, up to |eps + eta| quanta of 2^-106
relative otherwise: the remainder of a q0 that is k ulp off needs log2 k more bits than a double has).  So q = RN(n/d)
unless a rounding boundary lies within the gap g = |rem (y - 1/d) + rho/d| <= (n/d) (|kappa| + 1)^2 2^-106 of n/d
(kappa = eps / 2^-53).  For a quotient whose distance from the boundaries is uniform -- the directed hard cases of
tests/test_gpu_fast_step.py show the boundaries are reachable and the error is then exactly one ulp -- the probability of
that is the gap divided by the spacing of the boundaries:

    P(mis-rounded) = g / ulp(q)          per quotient, an EXACT function of (n, d, y), evaluated below in exact arithmetic
                                         (error-free product and sum transformations) for every recorded quotient.

Part 1 evaluates that expression over the quotients the kernels really form -- every quotient of every Euler step of
rays of BASELINE's configurations, recorded by curvis_selftest_fast_step -- and sums it: the expected number of
mis-rounded quotients per step, per 1080p frame, per video.  Part 2 checks the model where it can be checked: at directed
hard cases, where mis-rounding happens exactly when the model says so.  Part 3 does the same for the square root."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import common  # noqa: E402
import hard_cases as H  # noqa: E402
import oracle_lib as O  # noqa: E402
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import refpaths  # noqa: E402  (the reference's own camera paths: tests/golden/paths)
import curvis_amd  # noqa: E402

NAMES = curvis_amd.Context.FAST_STEP_QUOTIENTS


ulp_of, two_prod, gap_over_ulp = H.ulp_of, H.two_prod, H.gap_over_ulp


def states_of(ctx, metric, pm, pose, nrays, iterations, rng, res=(1920, 1080)):
    """(l, theta, p_l, p_theta, p_phi) before every Euler step of `nrays` random pixels of the camera, inside |l| <= R"""
    oc = O.camera(pose[0], pose[1], pose[2], 15.0, 43.0, res)
    px, py = rng.integers(0, res[0], nrays), rng.integers(0, res[1], nrays)
    dirs = np.zeros((nrays, 3))
    for i in range(nrays):
        O.lib().cvo_camera_outward_world(O.C.byref(oc), int(px[i]), int(py[i]), O._dp(dirs[i]))
    pos = np.tile(np.array(pose[0], dtype=np.float64), (nrays, 1))
    tr = ctx.compute_photon_trajectory(pm, pos, dirs, iterations, 0.05)
    keep = np.abs(tr[:, :, 1]) <= 100.0
    keep &= np.cumprod(keep, axis=1).astype(bool)          # up to the escape only
    st = tr[keep]
    return np.ascontiguousarray(st[:, [1, 2, 5, 6, 7]])


def measure(ctx, name, pm, states, steps_per_frame, frames_per_video=None):
    quot, fast, strict, took = ctx.selftest_fast_step(pm, states)
    n = len(states)
    print("## %s: %d recorded Euler steps, fast path taken by %.4f %% of them" % (name, n, 100.0 * took.mean()))
    same_state = (fast.view(np.uint64) == strict.view(np.uint64)).all(axis=1)
    print("new state of the fast step bit-identical to the strict step's: %d of %d" % (int(same_state.sum()), n))
    print("| quotient | formed | max abs(kappa) | rms kappa | mean P(mis-rounded) per quotient | quotient != IEEE quotient |")
    print("|---|---|---|---|---|---|")
    total, total_phi = 0.0, 0.0
    for k in range(6):
        nn, d, y, qf, qi, rem, eps = (quot[:, k, i] for i in range(7))
        ok = took & np.isfinite(qi) & (d == d)
        if not ok.any():
            continue
        kappa = -eps[ok] / 2.0 ** -53                     # y = (1 + kappa 2^-53)/d
        p, rem_mismatch, _ = gap_over_ulp(nn[ok], d[ok], y[ok], qi[ok], rem[ok], eps[ok])
        assert rem_mismatch == 0, "host remainder != the device's fma"
        if k == 5:
            total_phi = p.sum() / n           # only the kernels that integrate phi form it (debug dump, escape angles)
        else:
            total += p.sum() / n
        print("| %s | %d | %.2f | %.2f | %.3g (= 2^%.1f) | %d |" % (
            NAMES[k], int(ok.sum()), np.abs(kappa).max(), np.sqrt((kappa ** 2).mean()), p.mean(), np.log2(max(p.mean(), 1e-300)),
            int((qf[ok].view(np.uint64) != qi[ok].view(np.uint64)).sum())))
    # the Ellis step's square root: its last residual step adds |sqrt x - g| kappa_y 2^-53 <= 1/2 ulp x 1.5 x 2^-53 to a value
    # that is otherwise exact to 2^-107, against boundaries one ulp apart: <= 0.75 x 2^-53 (directed cases below: reachable
    # at the three j = -1 significands only)
    root = 0.75 * 2.0 ** -53 if "r' = l/r" in [NAMES[k] for k in range(6) if (took & (quot[:, k, 1] == quot[:, k, 1])).any()] else 0.0
    total += root
    print("expected mis-rounded operations per Euler step of the production kernels (the quotients above without 1/(r^2 sin^2)%s): "
          "%.3g = 2^%.1f   [with phi integrated: %.3g]" % (", + <= %.2g for the square root" % root if root else "", total, np.log2(total), total + total_phi))
    print("  -> per frame of %.4g steps: %.3g" % (steps_per_frame, total * steps_per_frame))
    if frames_per_video:
        print("  -> per render of %d such frames: %.3g  (probability that ONE quotient of the whole render is one ulp off)" % (
            frames_per_video, total * steps_per_frame * frames_per_video))
    print()
    return total


def directed_division(ctx, rng):
    print("## directed hard cases, division: n/d within |j| <= 15 quanta (2^-106 relative) of a rounding boundary, y = RN(1/d) moved by u ulp")
    n, d, j, ex, quanta = H.division_hard_cases(rng, 20000)
    assert np.array_equal(ctx.selftest_math(6, n, d), ex)
    print("| u | kappa of y | mis-rounded (of %d) | largest error | among j > 0 | among j < 0 | device == exact-arithmetic model (first 2000) | "
          "mis-rounded <=> gap g reaches the boundary (first 2000) |" % len(n))
    print("|---|---|---|---|---|---|---|---|")
    sub = slice(0, 2000)
    y0 = 1.0 / d
    for u in range(-4, 5):
        y = H.step_ulps(y0, u)
        got = ctx.selftest_math3(0, n, d, y)
        bad = got != ex
        kap = H.recip_error_units(d[:2000], y[:2000])
        mdl = H.div_with_recip_model(n[:2000], d[:2000], y[:2000])
        q0 = n[sub] * y[sub]
        rem = np.array([H._rn(H.Fraction(a) - H.Fraction(b) * H.Fraction(c)) for a, b, c in zip(n[sub].tolist(), d[sub].tolist(), q0.tolist())])
        eps = np.array([H._rn(1 - H.Fraction(b) * H.Fraction(c)) for b, c in zip(d[sub].tolist(), y[sub].tolist())])
        _, _, g = gap_over_ulp(n[sub], d[sub], y[sub], ex[sub], rem, eps)
        rel = g / (n[sub] / d[sub]) * 2.0 ** 106          # in quanta, signed towards larger |n/d|
        pred = ((j[sub] > 0) & (rel < -quanta[sub])) | ((j[sub] < 0) & (rel > quanta[sub]))
        print("| %+d | %.2f .. %.2f | %d | %d ulp | %d | %d | %s | %d of 2000 agree |" % (
            u, kap.min(), kap.max(), int(bad.sum()), int(H.ulp_distance(got, ex).max()), int((bad & (j > 0)).sum()),
            int((bad & (j < 0)).sum()), bool(np.array_equal(mdl, got[:2000])), int((pred == bad[sub]).sum())))
    a = rng.uniform(1.0, 2.0, 2_000_000) * 2.0 ** rng.integers(-40, 40, 2_000_000)
    b = rng.uniform(1.0, 2.0, 2_000_000) * 2.0 ** rng.integers(-40, 40, 2_000_000)
    yb = H.step_ulps(1.0 / b, rng.integers(-8, 9, b.size))
    print("random operands, y off by up to 8 ulp: %d of %d quotients differ from IEEE" % (
        int((ctx.selftest_math3(0, a, b, yb) != a / b).sum()), a.size))
    print()


def directed_sqrt(ctx):
    print("## directed hard cases, square root: sqrt(x) within |j| / (4 M) ulp of a rounding boundary (j = 7 mod 8, |j| <= 2000)")
    x, j, ex = H.sqrt_hard_cases(2000)
    assert np.array_equal(ctx.selftest_math(7, x), ex), "the device's IEEE sqrt"
    root = ctx.selftest_math3(1, x)
    y = ctx.selftest_math3(2, x)
    kap = np.array([float((H.Fraction(a) * H.Fraction(b) ** 2 - 1) * (1 << 52)) for a, b in zip(x.tolist(), y.tolist())])  # y sqrt(x) - 1 ~ (x y^2 - 1)/2
    bad = root != ex
    print("%d cases; sqrt_and_rsqrt mis-rounds %d of them (largest error %d ulp), all with |j| <= %s; its reciprocal root y is within %.2f units of "
          "2^-53 of 1/sqrt(x)" % (len(x), int(bad.sum()), int(H.ulp_distance(root, ex).max()), np.abs(j[bad]).max() if bad.any() else "-",
                                  np.abs(kap).max()))
    for jj in sorted(set(j[bad].tolist())):
        print("  j = %d: %d of %d cases mis-rounded" % (jj, int((bad & (j == jj)).sum()), int((j == jj).sum())))
    ys = 1.0 / np.sqrt(x)
    for u in (-2, -1, 0, 1, 2):
        r = ctx.selftest_math3(3, x, ex, H.step_ulps(ys, u))
        b = r != ex
        print("  last residual step from g = RN(sqrt x) with y = RN(1/sqrt x) %+d ulp: %d mis-rounded, |j| <= %s" % (
            u, int(b.sum()), np.abs(j[b]).max() if b.any() else "-"))
    xr = np.random.default_rng(5).uniform(1.0, 4.0, 4_000_000)
    print("random x in [1, 4): %d of %d roots differ from IEEE" % (int((ctx.selftest_math3(1, xr) != np.sqrt(xr)).sum()), xr.size))
    # the -1/x of atan (cv_div_nr): a reciprocal through one refined seed -- Markstein's exception is an all-ones significand
    ones = np.ldexp(float((1 << 53) - 1), np.arange(-52, 12).astype(np.int64) - 52)
    xs = np.concatenate([ones[ones >= 2.0], np.random.default_rng(6).uniform(2.0, 2000.0, 2_000_000)])
    print("cv_div_nr(-1, x), x >= 2 incl. all-ones significands: %d of %d differ from IEEE" % (
        int((ctx.selftest_math3(5, np.full_like(xs, -1.0), xs) != -1.0 / xs).sum()), xs.size))
    print()


def main():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from curvis_amd import paths, rendering
    ctx = curvis_amd.Context(0)
    rng = np.random.default_rng(2026)
    print("# fast Euler step: expected mis-rounded quotients, measured on %s" % ctx.device_info()["name"])
    print()
    default = ((0.0, 5.0, np.pi / 2, 0.0), (-1.0, 0.0, 0.0), (0.0, 0.0, 1.0))
    it = rendering.Interpolator.from_file(refpaths.reference_path_file("path_through.csv"))
    times = rendering.times_of_frames(it.min_time(), it.max_time(), 24.0)
    poses = [(tuple(it.camera_position(t)), tuple(it.camera_forward(t)), tuple(it.camera_up(t))) for t in times]
    ell, inter = curvis_amd.EllisMetric(1.0), curvis_amd.InterstellarMetric(0.1, 1e-4, 1.0)
    measure(ctx, "configs[1] / configs[3]'s metric: Ellis, default pose (l = 5), 1920x1080", ell,
            states_of(ctx, O.ellis(1.0), ell, default, 1024, 2300, rng), 4.05e9, 240)
    orbit = ((0.0, 3.0, np.pi / 2, 0.0), (-1.0, 0.0, 0.0), (0.0, 0.0, 1.0))
    measure(ctx, "configs[3]: Ellis, orbit pose (l = 3), 1920x1080", ell,
            states_of(ctx, O.ellis(1.0), ell, orbit, 1024, 2300, rng), 3.85e9, 240)
    measure(ctx, "configs[2]: Interstellar, default pose (l = 5), 3840x2160", inter,
            states_of(ctx, O.interstellar(), inter, default, 1024, 2300, rng, (3840, 2160)), 1.64e10, 480)
    measure(ctx, "configs[4]: Interstellar, fly-through frame 240 (l = 0.008, inside the throat), 3840x2160", inter,
            states_of(ctx, O.interstellar(), inter, poses[240], 1024, 2300, rng, (3840, 2160)), 1.61e10, 480)
    directed_division(ctx, rng)
    directed_sqrt(ctx)
    ctx.close()


if __name__ == "__main__":
    main()
