"""The C oracle (libm flavour) against tests/ref_python.py, an independent pure-Python restatement of the
reference's per-pixel path: every ray's final position and momentum (bit patterns), step count, escape code and
texel indices must agree.  Two restatements written separately from the reference text agreeing to the last bit
is the strongest pin available for the oracle here (the reference itself cannot be built: no Rust toolchain)."""
import math

import numpy as np
import pytest

import common
import oracle_lib as O
import ref_python as R

HALF_PI = math.pi / 2

SCENES = [
    ("ellis", (12, 7), (0.0, 5.0, HALF_PI, 0.0), (-1.0, 0.0, 0.0), (0.0, 0.0, 1.0), 2600),
    ("interstellar", (10, 6), (0.0, 5.0, HALF_PI, 0.0), (-1.0, 0.0, 0.0), (0.0, 0.0, 1.0), 2600),
    ("ellis", (9, 5), (0.0, 3.0, 1.1, 2.0), (-1.0, 0.2, 0.1), (0.1, 0.0, 1.0), 1500),      # cap binds for some rays
    ("interstellar", (8, 5), (0.0, -2.0, 1.2, 4.0), (1.0, 0.2, -0.1), (0.0, 0.3, 1.0), 3000),  # camera in the -l space
    ("flat", (6, 4), (0.0, 5.0, 1.0, 0.5), (1.0, 0.3, 0.2), (0.0, 0.0, 1.0), 2500),
]


def _metric(name):
    return {"ellis": R.Ellis(1.0), "interstellar": R.Interstellar(0.1, 1e-4, 1.0), "flat": R.Flat()}[name]


@pytest.mark.parametrize("name,res,pos,fwd,up,cap", SCENES)
def test_oracle_equals_independent_python_restatement(name, res, pos, fwd, up, cap):
    sp, sn = common.make_skies(64, 32, "check")
    om = {"ellis": lambda: O.ellis(1.0), "interstellar": lambda: O.interstellar(0.1, 1e-4, 1.0), "flat": O.flat}[name]()
    oc = O.camera(pos, fwd, up, 15.0, 43.0, res)
    rgb, dbg, st = O.render_image(O.LIBM, om, oc, O.sky(sp), O.sky(sn), cap, 100.0, 0.05, debug=True)
    metric = _metric(name)
    cam = R.Camera(pos, fwd, up, 15.0, 43.0, res[0], res[1])
    total = 0
    for py in range(res[1]):
        for px in range(res[0]):
            x, p, steps, code, d = R.render_pixel(metric, cam, px, py, cap, 100.0, 0.05)
            rec = dbg[py, px]
            assert steps == int(rec["steps"]) and code == int(rec["code"]), (px, py)
            # t and p_t are dead lanes of the path; the oracle reconstructs them for its dump the same way
            for k in range(4):
                assert np.float64(x[k]).view(np.uint64) == rec["x"][k].view(np.uint64) or (x[k] != x[k] and rec["x"][k] != rec["x"][k]), ("x", k, px, py)
                assert np.float64(p[k]).view(np.uint64) == rec["p"][k].view(np.uint64) or (p[k] != p[k] and rec["p"][k] != rec["p"][k]), ("p", k, px, py)
            if code != 0:
                sky = sp if code > 0 else sn
                tx, ty = R.sky_indices(d, sky.shape[1], sky.shape[0])
                assert (tx, ty) == (int(rec["tx"]), int(rec["ty"])), (px, py)
                cx, cy = min(tx, sky.shape[1] - 1), min(ty, sky.shape[0] - 1)
                assert tuple(rgb[py, px]) == tuple(sky[cy, cx, :3])
            else:
                assert tuple(rgb[py, px]) == (0, 0, 0)
            total += steps
    assert total == st.steps


def test_camera_matrix_equals_the_oracle():
    rng = np.random.default_rng(11)
    for _ in range(50):
        fwd, up = rng.uniform(-1, 1, 3), rng.uniform(-1, 1, 3)
        rot, inv = R.orientation(list(fwd), list(up))
        oc = O.camera((0.0, 5.0, 1.0, 0.0), tuple(fwd), tuple(up), 15.0, 43.0, (16, 9))
        got = np.array(oc.rot, dtype=np.float64).reshape(3, 3)
        assert np.array_equal(np.array(rot).view(np.uint64), got.view(np.uint64))


@pytest.mark.parametrize("name,pos,n0", [("ellis", (0.0, 5.0, HALF_PI, 0.0), 24), ("interstellar", (0.0, 4.0, 1.3, 0.7), 20)])
def test_efficient_renderer_equals_independent_python_restatement(name, pos, n0):
    """render_image_efficient (what `curvis image|video` run): sample table, call/step counts and pixels"""
    sp, sn = common.make_skies(64, 32, "check")
    om = {"ellis": lambda: O.ellis(1.0), "interstellar": lambda: O.interstellar(0.1, 1e-4, 1.0)}[name]()
    res, fwd, up = (10, 6), (-1.0, 0.1, 0.05), (0.0, 0.0, 1.0)
    oc = O.camera(pos, fwd, up, 15.0, 43.0, res)
    want_rgb, want, _ = O.render_image_efficient(O.LIBM, om, oc, O.sky(sp), O.sky(sn), 3000, 100.0, 0.05, n0, 12, 1e-3, 1e-3)
    cam = R.Camera(pos, fwd, up, 15.0, 43.0, res[0], res[1])
    rgb, (sa, se, ss), steps, calls = R.render_image_efficient(_metric(name), cam, sp, sn, 3000, 100.0, 0.05, n0, 12, 1e-3, 1e-3)
    assert np.array_equal(np.array(sa).view(np.uint64), np.asarray(want["a"]).view(np.uint64))
    assert np.array_equal(np.array(se).view(np.uint64), np.asarray(want["e"]).view(np.uint64))
    assert np.array_equal(np.array(ss), np.asarray(want["s"]))
    assert (steps, calls) == (want["steps"], want["calls"])
    assert np.array_equal(np.array(rgb, dtype=np.uint8), want_rgb)
