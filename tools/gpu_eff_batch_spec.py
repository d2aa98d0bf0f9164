#!/usr/bin/env python3
"""Efficient renderer, batches of 1080p frames: (first-launch depth, later depth) settings."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import curvis_amd
from curvis_amd import skies
ctx = curvis_amd.Context(0)
ctx.set_sky(0, curvis_amd.SphericalImage(skies.smooth(2048, 1024, 0))); ctx.set_sky(1, curvis_amd.SphericalImage(skies.smooth(2048, 1024, 1)))
ARGS = (40000, 100.0, 0.05, 100, 100, 1e-5, 1e-5)
def cam(l, phi=0.0):
    return curvis_amd.Camera((0.0, l, np.pi / 2, phi), (-1.0, 0.0, 0.0), (0.0, 0.0, 1.0), 15.0, 43.0, 1920, 1080)
for name, metric in (("ellis", curvis_amd.EllisMetric(1.0)), ("interstellar", curvis_amd.InterstellarMetric(0.1, 1e-4, 1.0))):
    for k in (4, 8, 16, 30):
        cams = [cam(5.0 - 0.1 * i, 0.05 * i) for i in range(k)]
        for first, spec in ((-1, -1), (3, 4), (4, 5), (4, 6), (5, 6), (5, 7), (6, 7), (6, 8)):
            ctx.set_option("sampling_speculation_first", first); ctx.set_option("sampling_speculation", spec)
            ts = []
            for _ in range(4):
                t0 = time.perf_counter()
                _, st = ctx.render_efficient(metric, cams, *ARGS, download=False)
                ts.append((time.perf_counter() - t0) * 1e3 / k)
            print("%s %2d frames first %2d later %2d: %.3f ms/frame wall (kernels %.3f), launches %d, points %d" % (
                name, k, first, spec, float(np.median(ts[1:])), st.integrate_ms / k, ctx.get_option("last_sampling_launches"), ctx.get_option("last_sampling_evaluated")), flush=True)
