"""tests/hard_cases.py on CPU: the exact constructions behind tests/test_gpu_fast_step.py, and the fast step's pure-fma
primitives (div_with_recip, the square root's residual step, recip_newton) on the x86 twin of cv_device.h -- the same
source the kernels compile, so what holds here bit for bit is what the device is asserted against under -m gpu."""
import numpy as np

import common
import hard_cases as H


def test_division_cases_sit_where_the_construction_says():
    rng = np.random.default_rng(1)
    n, d, j, expect, quanta = H.division_hard_cases(rng, 3000)
    assert np.array_equal(n / d, expect)                       # RN(n/d) by construction == the FPU's
    for a, b, jj, e, q in list(zip(n.tolist(), d.tolist(), j.tolist(), expect.tolist(), quanta.tolist()))[:200]:
        exact = H.Fraction(a) / H.Fraction(b)
        mid = H.Fraction(e) - (H.Fraction(e) - H.Fraction(float(np.nextafter(e, 0.0 if jj > 0 else np.inf)))) / 2
        rel = (exact - mid) / exact * (1 << 106)
        assert (rel > 0) == (jj > 0) and abs(float(abs(rel)) - q) < 1e-6 * q   # `quanta` units of 2^-106 from the boundary


def test_div_with_recip_model_and_gap_predict_every_case_on_the_twin():
    rng = np.random.default_rng(2)
    n, d, j, expect, quanta = H.division_hard_cases(rng, 4000)
    reached = {}
    for u in (-3, -1, 0, 1, 3):
        y = H.step_ulps(1.0 / d, u)
        got = common.twin_math3(0, n, d, y)
        assert np.array_equal(got[:800], H.div_with_recip_model(n[:800], d[:800], y[:800]))
        bad = got != expect
        reached[u] = int(bad.sum())
        assert H.ulp_distance(got, expect).max() <= 1 and not (bad & (j < 0)).any()
        q0 = n * y
        rem = np.array([H._rn(H.Fraction(a) - H.Fraction(b) * H.Fraction(c)) for a, b, c in zip(n[:800].tolist(), d[:800].tolist(), q0[:800].tolist())])
        eps = np.array([H._rn(1 - H.Fraction(b) * H.Fraction(c)) for b, c in zip(d[:800].tolist(), y[:800].tolist())])
        _, mismatch, g = H.gap_over_ulp(n[:800], d[:800], y[:800], expect[:800], rem, eps)
        rel = g / (n[:800] / d[:800]) * 2.0 ** 106
        pred = ((j[:800] > 0) & (rel < -quanta[:800])) | ((j[:800] < 0) & (rel > quanta[:800]))
        assert mismatch == 0 and np.array_equal(pred, bad[:800])           # mis-rounded <=> the exact gap reaches the boundary
        kap = np.abs(H.recip_error_units_fast(d, y))
        assert ((kap[bad] + 1.0) ** 2 >= quanta[bad]).all()
    assert reached[0] == 0 and min(reached[u] for u in (-3, -1, 1, 3)) > 0


def test_sqrt_cases_and_the_residual_step_on_the_twin():
    x, j, expect = H.sqrt_hard_cases(800)
    assert len(x) > 250 and np.array_equal(np.sqrt(x), expect) and set((j % 8).tolist()) == {7}
    for a, jj, e in list(zip(x.tolist(), j.tolist(), expect.tolist()))[:100]:     # within |j| / (4 M) ulp of the boundary
        mid = (H.Fraction(e) + H.Fraction(float(np.nextafter(e, 0.0 if jj > 0 else np.inf)))) / 2
        assert (H.Fraction(a) > mid * mid) == (jj > 0)
        assert abs(H.Fraction(a) - mid * mid) / H.Fraction(a) < H.Fraction(abs(jj) + 1, 1 << 106)
    r = common.twin_math3(3, x, expect, 1.0 / np.sqrt(x))
    bad = r != expect
    assert H.ulp_distance(r, expect).max() <= 1 and set(j[bad].tolist()) <= {-1}
    d = np.random.default_rng(3).uniform(1.0, 2.0, 100000)
    for u in (-1, 1):
        assert np.array_equal(common.twin_math3(6, d, H.step_ulps(1.0 / d, u)), 1.0 / d)   # recip_newton restores RN(1/d)
