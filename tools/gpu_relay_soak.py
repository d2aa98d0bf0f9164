#!/usr/bin/env python3
"""Relay kernel soak: random frame sizes (down to a few workgroups per CU), 1-8 frames per launch, random cameras
and metrics, automatic segment and forced short segments; every launch must reproduce the static kernel's frames,
its statistics and its PER-FRAME counters.  python tools/gpu_relay_soak.py [launches] [seed]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import curvis_amd
from curvis_amd import skies
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
ctx = curvis_amd.Context(0)
ctx.set_sky(0, curvis_amd.SphericalImage(skies.checker(1024, 512, 1))); ctx.set_sky(1, curvis_amd.SphericalImage(skies.checker(1024, 512, 2)))
bad = 0; relay_launches = 0; parks = 0; t0 = time.time()
for it in range(N):
    w = int(rng.integers(300, 2000)); h = int(rng.integers(200, 1200)); nf = int(rng.integers(1, 9))
    if w * h * nf > 6_000_000: nf = max(1, 6_000_000 // (w * h))
    kind = rng.integers(0, 3)
    m = [curvis_amd.EllisMetric(float(rng.uniform(0.5, 3))), curvis_amd.InterstellarMetric(0.1, float(10 ** rng.uniform(-4, 0)), 1.0),
         curvis_amd.FlatSphericalMetric()][kind]
    cams = [curvis_amd.Camera((0.0, float(rng.uniform(1, 9) * rng.choice([-1, 1])), float(rng.uniform(0.4, 2.7)), float(rng.uniform(0, 6))),
                              (float(rng.normal()) - 1.0, float(rng.normal()), float(rng.normal())), (0.0, 0.0, 1.0), float(rng.uniform(10, 40)), 43.0, w, h)
            for _ in range(nf)]
    cap = int(rng.choice([600, 2500, 4096]))
    ctx.set_option("variant", 1)
    want, sw = ctx.render_brute(m, cams, cap, 100.0, 0.05)
    fw = [(f.rays, f.steps, f.n_pos, f.n_neg, f.n_none, f.n_oob) for f in ctx.frame_stats()]   # per-frame counters
    ctx.set_option("variant", -1)
    ctx.set_option("relay_min_blocks", 0 if it % 3 == 0 else -1)
    ctx.set_option("relay_segment", int(rng.choice([0, 0, 64, 300])))
    got, sg = ctx.render_brute(m, cams, cap, 100.0, 0.05)
    fg = [(f.rays, f.steps, f.n_pos, f.n_neg, f.n_none, f.n_oob) for f in ctx.frame_stats()]
    relay_launches += ctx.get_option("last_relay_launches"); parks += ctx.get_option("last_relay_parks")
    if not (np.array_equal(got, want) and fg == fw and len(fg) == nf and (sg.rays, sg.steps, sg.n_pos, sg.n_neg, sg.n_none) == (sw.rays, sw.steps, sw.n_pos, sw.n_neg, sw.n_none)):
        bad += 1
        print("MISMATCH", it, w, h, nf, kind, cap, flush=True)
print("launches %d, of which relay %d (hand-overs %d), mismatches %d, %.0f s" % (N, relay_launches, parks, bad, time.time() - t0))
sys.exit(1 if bad else 0)
