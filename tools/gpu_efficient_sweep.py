#!/usr/bin/env python3
"""Efficient renderer (what `curvis image` runs): time per 1080p frame and number of sampling launches as a
function of the speculation depth; single frame and a batch of 8 orbit frames."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import curvis_amd
from curvis_amd import skies
ctx = curvis_amd.Context(0)
ctx.set_sky(0, curvis_amd.SphericalImage(skies.smooth(2048, 1024, 0))); ctx.set_sky(1, curvis_amd.SphericalImage(skies.smooth(2048, 1024, 1)))
def cam(l, phi=0.0):
    return curvis_amd.Camera((0.0, l, np.pi / 2, phi), (-1.0, 0.0, 0.0), (0.0, 0.0, 1.0), 15.0, 43.0, 1920, 1080)
# reference defaults (src/settings.rs): propagation cap 40000, R 100, delta 0.05, 100 initial samples, 100 (=initial) rounds, thr 1e-5
SPECS = [int(v) for v in os.environ.get('SPECS', '0,2,4,6,8').split(',')]
BATCHES = [int(v) for v in os.environ.get('BATCHES', '8').split(',')]
ARGS = (40000, 100.0, 0.05, 100, 100, 1e-5, 1e-5)
for name, metric in (("ellis", curvis_amd.EllisMetric(1.0)), ("interstellar", curvis_amd.InterstellarMetric(0.1, 1e-4, 1.0))):
    for spec in SPECS:
        ctx.set_option("sampling_speculation", spec)
        for label, cams in [("1 frame", cam(5.0))] + [("%d frames" % k, [cam(5.0 - 0.1 * i, 0.05 * i) for i in range(k)]) for k in BATCHES]:
            n = 1 if not isinstance(cams, list) else len(cams)
            ctx.render_efficient(metric, cams, *ARGS, download=False)
            t0 = time.perf_counter()
            for _ in range(3):
                out = ctx.render_efficient(metric, cams, *ARGS, download=False)
            dt = (time.perf_counter() - t0) / 3 / n * 1e3
            print("%s spec %d %s: %.3f ms/frame wall, launches %d, evaluated %d" % (
                name, spec, label, dt, ctx.get_option("last_sampling_launches"), ctx.get_option("last_sampling_evaluated")), flush=True)
