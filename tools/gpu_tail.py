#!/usr/bin/env python3
"""How much of a single-frame launch is tail/ramp?  Same camera rendered as 1, 2, 4, 8 frames per launch."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import curvis_amd
from curvis_amd import skies

ctx = curvis_amd.Context(0)
sp, sn = skies.smooth(2048, 1024, 0), skies.smooth(2048, 1024, 1)
ctx.set_sky(0, curvis_amd.SphericalImage(sp)); ctx.set_sky(1, curvis_amd.SphericalImage(sn))
cam = curvis_amd.Camera((0.0, 5.0, np.pi / 2, 0.0), (-1.0, 0.0, 0.0), (0.0, 0.0, 1.0), 15.0, 43.0, 1920, 1080)
for name, metric in (("ellis", curvis_amd.EllisMetric(1.0)), ("interstellar", curvis_amd.InterstellarMetric(0.1, 1e-4, 1.0))):
    for n in (1, 2, 4, 8):
        cams = [cam] * n
        ctx.render_brute(metric, cams, 4096, 100.0, 0.05, download=False)
        ms = []
        for _ in range(3):
            _, st = ctx.render_brute(metric, cams, 4096, 100.0, 0.05, download=False)
            ms.append(st.integrate_ms / n)
        print("%s frames/launch %d: %.3f ms/frame (min of 3: %.3f)" % (name, n, sum(ms) / 3, min(ms)), flush=True)
