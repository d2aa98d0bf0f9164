#!/usr/bin/env python3
"""Parity fuzz of the "direct" mode (curvis_render_direct): N random scenes (tests/common.random_scene) at 40x24, pixels
and counters against the oracle's counterpart (cvo_render_image_direct, cv_math flavour).  Scenes on which the oracle
panics (camera outside the escape radius, undefined tangent rotation for the camera direction) must be errors here."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import common, oracle_lib as O, curvis_amd
N = int(sys.argv[1]) if len(sys.argv) > 1 else 400
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 20260929)
ctx = curvis_amd.Context(0)
sp, sn = common.make_skies(512, 256, "smooth")
ctx.set_sky(0, curvis_amd.SphericalImage(sp)); ctx.set_sky(1, curvis_amd.SphericalImage(sn))
bad = 0; panics = 0; steps = 0; kinds = {}
t0 = time.time()
for trial in range(N):
    om, oc, pm, pc, delta, cap, R = common.random_scene(rng, res=(40, 24))
    kinds[om.kind] = kinds.get(om.kind, 0) + 1
    want = None
    with np.errstate(all="ignore"):
        try:
            want, st = O.render_image_direct(O.CV, om, oc, O.sky(sp), O.sky(sn), cap, R, delta)
        except RuntimeError:
            panics += 1
    try:
        got, s = ctx.render_direct(pm, pc, cap, R, delta)
    except curvis_amd.CurvisError:
        got = None
    if want is None or got is None:
        ok = want is None and got is None
    else:
        ok = np.array_equal(got, want) and (s.rays, s.steps, s.n_pos, s.n_neg, s.n_none, s.n_oob) == (st.rays, st.steps, st.n_pos, st.n_neg, st.n_none, st.n_oob)
        steps += int(st.steps)
    if not ok:
        bad += 1
        print("MISMATCH trial", trial, "kind", om.kind, "oracle panic" if want is None else "", "gpu error" if got is None else "", flush=True)
print("direct mode: scenes %d (by kind %s, %d of them errors on both sides), Euler steps %d, mismatching scenes %d, %.0f s" % (
    N, kinds, panics, steps, bad, time.time() - t0))
sys.exit(1 if bad else 0)
