#!/usr/bin/env python3
"""Relay kernel ("variant" 2) vs static kernel on single-frame launches: parity of the frame and statistics,
kernel time, number of launches, segment sweep."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import curvis_amd
from curvis_amd import skies
ctx = curvis_amd.Context(0)
ctx.set_sky(0, curvis_amd.SphericalImage(skies.checker(2048, 1024, 1))); ctx.set_sky(1, curvis_amd.SphericalImage(skies.checker(2048, 1024, 2)))
def cam(w, h, l=5.0):
    return curvis_amd.Camera((0.0, l, np.pi / 2, 0.0), (-1.0, 0.0, 0.0), (0.0, 0.0, 1.0), 15.0, 43.0, w, h)
cases = [("ellis", curvis_amd.EllisMetric(1.0), 1920, 1080, 4096), ("interstellar", curvis_amd.InterstellarMetric(0.1, 1e-4, 1.0), 1920, 1080, 4096),
         ("ellis", curvis_amd.EllisMetric(1.0), 256, 144, 40000), ("ellis", curvis_amd.EllisMetric(1.0), 333, 211, 3000),
         ("ellis", curvis_amd.EllisMetric(1.0), 1280, 720, 4096), ("ellis", curvis_amd.EllisMetric(1.0), 3840, 2160, 4096),
         ("interstellar", curvis_amd.InterstellarMetric(0.1, 1e-4, 1.0), 3840, 2160, 8192)]
segs = [int(v) for v in os.environ.get("SEGS", "0").split(",")]
for name, m, w, h, cap in cases:
    c = cam(w, h)
    ctx.set_option("variant", 1)
    ref, st1 = ctx.render_brute(m, c, cap, 100.0, 0.05)
    t1 = []
    for _ in range(4):
        _, s = ctx.render_brute(m, c, cap, 100.0, 0.05, download=False); t1.append(s.integrate_ms)
    for seg in segs:
        ctx.set_option("variant", 2); ctx.set_option("relay_segment", seg); ctx.set_option("relay_min_blocks", int(os.environ.get("MINB", "-1")))
        rgb, st2 = ctx.render_brute(m, c, cap, 100.0, 0.05)
        same = np.array_equal(rgb, ref) and (st1.steps, st1.rays, st1.n_pos, st1.n_neg, st1.n_none) == (st2.steps, st2.rays, st2.n_pos, st2.n_neg, st2.n_none)
        t2 = []
        for _ in range(4):
            _, s = ctx.render_brute(m, c, cap, 100.0, 0.05, download=False); t2.append(s.integrate_ms)
        print("%s %dx%d cap %d seg %d: identical %s; static %.3f ms (min %.3f), relay %.3f ms (min %.3f), launches %d" % (
            name, w, h, cap, seg, same, sum(t1) / 4, min(t1), sum(t2) / 4, min(t2), ctx.get_option("last_relay_launches")),
              "parks", ctx.get_option("last_relay_parks"), "relay waves that asked", ctx.get_option("last_relay_waiters"), flush=True)
ctx.set_option("variant", 1)
