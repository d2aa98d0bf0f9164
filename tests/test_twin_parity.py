"""CPU check that the kernel source (cv_device.h compiled for x86 as tests/host_twin) is bit-identical
to the oracle's CVO_CV flavour -- full final state of every ray, not just pixels.  The GPU tests then
only have to establish that gfx950 executes the same IEEE operations."""
import numpy as np
import pytest

import common
import oracle_lib as O


@pytest.mark.parametrize("fast", [0, 1])
@pytest.mark.parametrize("metric,pos,fwd,cap", [
    ("ellis", (0.0, 5.0, common.HALF_PI, 0.0), (-1.0, 0.0, 0.0), 4096),
    ("interstellar", (0.0, 5.0, common.HALF_PI, 0.0), (-1.0, 0.0, 0.0), 4096),
    ("ellis", (0.0, 3.0, common.HALF_PI, 1.0), (-1.0, 0.1, 0.05), 2500),
    ("interstellar", (0.0, -2.0, 1.2, 4.0), (1.0, 0.2, -0.1), 3000),
    ("flat", (0.0, 5.0, 1.0, 0.5), (1.0, 0.3, 0.2), 4096),
])
def test_twin_equals_oracle_cv(metric, pos, fwd, cap, fast):
    sp, sn = common.make_skies(128, 64, "check")
    om, oc, pm, pc = common.scene(metric, res=(24, 14), pos=pos, fwd=fwd)
    want_rgb, want_dbg, st = O.render_image(O.CV, om, oc, O.sky(sp), O.sky(sn), cap, 100.0, 0.05, debug=True)
    got_rgb, got_dbg = common.twin_render(pm, pc, sp, sn, cap, 100.0, 0.05, fast=fast)
    common.assert_debug_equal(got_dbg, want_dbg, check_t=False)
    assert np.array_equal(got_rgb, want_rgb)
    assert st.n_pos + st.n_neg + st.n_none == 24 * 14


def test_flavours_agree_to_rounding():
    """libm vs cv_math flavour of the oracle: same escape classification and step counts away from the
    ill-conditioned pole rows, final state to ~1e-10 (error growth over ~2000 Euler steps)."""
    sp, sn = common.make_skies(128, 64, "smooth")
    om, oc, _, _ = common.scene("ellis", res=(32, 18))
    a_rgb, a, _ = O.render_image(O.LIBM, om, oc, O.sky(sp), O.sky(sn), 4096, 100.0, 0.05, debug=True)
    b_rgb, b, _ = O.render_image(O.CV, om, oc, O.sky(sp), O.sky(sn), 4096, 100.0, 0.05, debug=True)
    same = (a["steps"] == b["steps"]) & (a["code"] == b["code"])
    assert same.mean() > 0.97
    d = np.abs(a["x"][same][:, 1:3] - b["x"][same][:, 1:3])
    assert np.median(d) < 1e-11
    assert np.abs(a_rgb.astype(int) - b_rgb.astype(int)).max() <= 1 or (a_rgb != b_rgb).any(axis=2).mean() < 0.03


def test_fuzz_twin_vs_oracle():
    """30 random scenes (metric parameters, pose, step, radius, cap): the kernel source on x86, strict and
    fast step, against the oracle -- full ray state."""
    rng = np.random.default_rng(2024)
    sp, sn = common.make_skies(64, 32, "check")
    for trial in range(30):
        om, oc, pm, pc, delta, cap, R = common.random_scene(rng, res=(12, 8))
        with np.errstate(all="ignore"):
            want_rgb, want_dbg, _ = O.render_image(O.CV, om, oc, O.sky(sp), O.sky(sn), cap, R, delta, debug=True)
        for fast in (0, 1):
            got_rgb, got_dbg = common.twin_render(pm, pc, sp, sn, cap, R, delta, fast=fast)
            common.assert_debug_equal(got_dbg, want_dbg, check_t=False)
            assert np.array_equal(got_rgb, want_rgb), (trial, fast)
