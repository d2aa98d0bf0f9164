#!/usr/bin/env python3
"""Deep parity fuzz (one-off, beyond tests/): N random scenes (tests/common.random_scene) at 40x24, full ray state
(l, theta, p_l, p_theta, p_phi, steps, code, texel) of the default GPU kernels against the oracle (cv_math
flavour).  Interstellar scenes are over-sampled: they exercise the table-driven atan / log."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import common, oracle_lib as O, curvis_amd
N = int(sys.argv[1]) if len(sys.argv) > 1 else 400
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 20260928)
ctx = curvis_amd.Context(0)
sp, sn = common.make_skies(128, 64, "check")
ctx.set_sky(0, curvis_amd.SphericalImage(sp)); ctx.set_sky(1, curvis_amd.SphericalImage(sn))
bad = 0; rays = 0; steps = 0; kinds = {}
t0 = time.time()
for trial in range(N):
    while True:
        om, oc, pm, pc, delta, cap, R = common.random_scene(rng, res=(40, 24))
        if om.kind == O.INTERSTELLAR or rng.random() < 0.3:
            break
    kinds[om.kind] = kinds.get(om.kind, 0) + 1
    with np.errstate(all="ignore"):
        want_rgb, want_dbg, st = O.render_image(O.CV, om, oc, O.sky(sp), O.sky(sn), cap, R, delta, debug=True)
    rgb, s, dbg = ctx.render_brute(pm, pc, cap, R, delta, debug=True)
    rgb2, s2 = ctx.render_brute(pm, pc, cap, R, delta)          # fused default path
    ok = np.array_equal(rgb, want_rgb) and np.array_equal(rgb2, want_rgb) and s2.steps == st.steps
    for f in ("steps", "code", "tx", "ty"):
        ok = ok and np.array_equal(dbg[f], want_dbg[f])
    for f in ("x", "p"):
        a, b = dbg[f].view(np.uint64), want_dbg[f].view(np.uint64)
        nan = np.isnan(dbg[f]) & np.isnan(want_dbg[f])
        ok = ok and bool(np.all((a == b) | nan))
    rays += 40 * 24; steps += int(st.steps)
    if not ok:
        bad += 1
        print("MISMATCH trial", trial, "kind", om.kind, flush=True)
print("scenes %d (by kind %s), rays %d, Euler steps %d, mismatching scenes %d, %.0f s" % (N, kinds, rays, steps, bad, time.time() - t0))
sys.exit(1 if bad else 0)
