#!/usr/bin/env python3
"""Fast step (shared reciprocals, third-order seed refinement, Markstein quotients) against the strict step (the
compiler's IEEE division and sqrt) on the DEVICE, full 1080p frames, random cameras: final state of every ray
(l, theta, phi, p_l, p_theta, p_phi, step count, escape code, texel) must be identical.  One frame = 2 M rays,
~4e9 Euler steps, ~2.5e10 divisions and square roots.

    python tools/gpu_fast_vs_strict.py [frames] [seed]
"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import curvis_amd
from curvis_amd import skies
N = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = curvis_amd.Context(0)
ctx.set_sky(0, curvis_amd.SphericalImage(skies.smooth(512, 256, 0))); ctx.set_sky(1, curvis_amd.SphericalImage(skies.smooth(512, 256, 1)))
bad = 0; steps = 0; kinds = {}
t0 = time.time()
for it in range(N):
    kind = ["ellis", "interstellar", "ellis", "interstellar", "flat"][it % 5]
    if kind == "ellis":
        m = curvis_amd.EllisMetric(float(10.0 ** rng.uniform(-0.5, 0.7)))
    elif kind == "interstellar":
        m = curvis_amd.InterstellarMetric(float(10.0 ** rng.uniform(-1.5, 0.3)), float(10.0 ** rng.uniform(-4, 0.3)), float(10.0 ** rng.uniform(-0.3, 0.5)))
    else:
        m = curvis_amd.FlatSphericalMetric()
    l = float(rng.uniform(0.5, 12.0) * rng.choice([-1.0, 1.0]))
    th = float(rng.uniform(0.25, np.pi - 0.25)) if rng.random() < 0.7 else float(np.pi / 2)
    fwd = rng.normal(size=3); fwd[0] -= np.sign(l) * 1.5
    if it % 3 == 0:   # axis-aligned camera in the equatorial plane: middle pixel column with theta == fl(pi/2), middle row with p_phi == 0
        th = float(np.pi / 2); fwd = np.array([-np.sign(l), 0.0, 0.0])
    cam = curvis_amd.Camera((0.0, l, th, float(rng.uniform(0, 6.28))), tuple(float(v) for v in fwd), (0.0, 0.0, 1.0),
                            float(rng.uniform(10, 40)), 43.0, 1920, 1080)
    cap = int(rng.choice([2500, 4096]))
    res = []
    for fast in (1, 0):
        ctx.set_option("fast_math", fast)
        rgb, st, dbg = ctx.render_brute(m, cam, cap, 100.0, 0.05, debug=True)
        res.append((rgb, st, dbg))
    (r1, s1, d1), (r0, s0, d0) = res
    ok = np.array_equal(r1, r0) and s1.steps == s0.steps
    for f in ("steps", "code", "tx", "ty"):
        ok = ok and np.array_equal(d1[f], d0[f])
    for f in ("x", "p"):
        a, b = d1[f].view(np.uint64), d0[f].view(np.uint64)
        ok = ok and bool(np.all((a == b) | (np.isnan(d1[f]) & np.isnan(d0[f]))))
    steps += int(s1.steps); kinds[kind] = kinds.get(kind, 0) + 1
    if not ok:
        bad += 1
        print("MISMATCH frame", it, kind, "l", l, "theta", th, flush=True)
ctx.set_option("fast_math", 1)
print("frames %d (%s), Euler steps %.3e, frames with a differing ray: %d, %.0f s" % (N, kinds, steps, bad, time.time() - t0))
sys.exit(1 if bad else 0)
