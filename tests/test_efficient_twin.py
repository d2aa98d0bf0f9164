"""Efficient renderer (the CLI's variant): host sampler (cv_sampler.h) + per-ray / per-pixel functions
(cv_efficient.h) compiled for x86 against the oracle's restatement of src/systems.rs:333-527,
src/sampling.rs and interp 1.0.3 -- sample tables and pixels bit for bit.  CPU only."""
import numpy as np
import pytest

import common
import oracle_lib as O

CASES = [
    ("ellis", (0.0, 5.0, common.HALF_PI, 0.0), (-1.0, 0.0, 0.0), 4096, 100),   # KAT-5 configuration
    ("ellis", (0.0, 3.0, common.HALF_PI, 0.9), (-1.0, 0.0, 0.0), 4096, 100),   # path_orbit pose
    ("interstellar", (0.0, 5.0, common.HALF_PI, 0.0), (-1.0, 0.0, 0.0), 4096, 100),
    ("ellis", (0.0, -2.5, 1.1, 2.0), (1.0, 0.3, 0.1), 3000, 60),               # -l side, tilted camera
]


@pytest.mark.parametrize("fast", [0, 1])
@pytest.mark.parametrize("metric,pos,fwd,cap,n0", CASES)
def test_twin_efficient_equals_oracle(metric, pos, fwd, cap, n0, fast):
    sp, sn = common.make_skies(256, 128, "check")
    om, oc, pm, pc = common.scene(metric, res=(48, 27), pos=pos, fwd=fwd)
    # the reference wires max_iterations_sampling to sampling_initial_nums (src/main.rs:47,107)
    want_rgb, want, st = O.render_image_efficient(O.CV, om, oc, O.sky(sp), O.sky(sn), cap, 100.0, 0.05, n0, n0, 1e-5, 1e-5)
    got_rgb, got = common.twin_render_efficient(pm, pc, sp, sn, cap, 100.0, 0.05, n0, n0, 1e-5, 1e-5, fast=fast)
    assert got["calls"] == want["calls"] and got["steps"] == want["steps"]
    for k in ("a", "e", "s"):
        assert np.array_equal(common.bits(got[k]), common.bits(want[k])), k
    assert np.array_equal(got_rgb, want_rgb)


def test_kat5_counts_through_the_batched_sampler():
    sp, sn = common.make_skies(64, 32, "check")
    om, oc, pm, pc = common.scene("ellis", res=(8, 6))
    _, got = common.twin_render_efficient(pm, pc, sp, sn, 4096, 100.0, 0.05, 100, 100, 1e-5, 1e-5, fast=1)
    assert len(got["a"]) == 678 and got["calls"] == 712 and got["steps"] == 1496307


def test_nan_direction_maps_to_texel_zero():
    """A NaN rotation (axis == 0 -> 0/0, src/systems.rs:504) reaches the sky lookup as a NaN direction and
    lands on texel (0, 0) through `NaN as u32 == 0` (src/images.rs:118-119): checked on the lookup itself."""
    import ctypes as C
    img = np.arange(64 * 32 * 4, dtype=np.uint32).astype(np.uint8).reshape(32, 64, 4)
    s = O.sky(np.ascontiguousarray(img))
    px = (C.c_uint8 * 4)()
    O.lib().cvo_sky_pixel(O.CV, C.byref(s), O._dp(O.vec(np.nan, np.nan, np.nan)), px)
    assert list(px) == list(img[0, 0])
