#!/usr/bin/env python3
"""Efficient renderer, single 1080p image: (first-launch depth, later depth) settings that reach fewer launches."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import curvis_amd
from curvis_amd import skies
ctx = curvis_amd.Context(0)
ctx.set_sky(0, curvis_amd.SphericalImage(skies.smooth(2048, 1024, 0))); ctx.set_sky(1, curvis_amd.SphericalImage(skies.smooth(2048, 1024, 1)))
ARGS = (40000, 100.0, 0.05, 100, 100, 1e-5, 1e-5)
for name, metric in (("ellis", curvis_amd.EllisMetric(1.0)), ("interstellar", curvis_amd.InterstellarMetric(0.1, 1e-4, 1.0))):
    for l in (5.0, 3.0, 8.0):
        cam = curvis_amd.Camera((0.0, l, np.pi / 2, 0.0), (-1.0, 0.0, 0.0), (0.0, 0.0, 1.0), 15.0, 43.0, 1920, 1080)
        for first, spec in ((-1, -1), (7, 8), (8, 8), (8, 9), (9, 9), (8, 10), (9, 10), (10, 10)):
            ctx.set_option("sampling_speculation_first", first); ctx.set_option("sampling_speculation", spec)
            ts = []
            for _ in range(5):
                t0 = time.perf_counter()
                _, st = ctx.render_efficient(metric, cam, *ARGS, download=False)
                ts.append((time.perf_counter() - t0) * 1e3)
            print("%s l=%.0f first %2d later %2d: wall %.3f ms (kernels %.3f), launches %d, points %d" % (
                name, l, first, spec, float(np.median(ts[1:])), st.integrate_ms, ctx.get_option("last_sampling_launches"), ctx.get_option("last_sampling_evaluated")), flush=True)
