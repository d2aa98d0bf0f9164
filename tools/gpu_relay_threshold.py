#!/usr/bin/env python3
"""Where the relay kernel starts to pay: static vs relay on small single frames (relay_min_blocks forced to 0)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import curvis_amd
from curvis_amd import skies
ctx = curvis_amd.Context(0)
ctx.set_sky(0, curvis_amd.SphericalImage(skies.smooth(512, 256, 0))); ctx.set_sky(1, curvis_amd.SphericalImage(skies.smooth(512, 256, 1)))
ctx.set_option("relay_min_blocks", 0)
for name, m in (("ellis", curvis_amd.EllisMetric(1.0)), ("interstellar", curvis_amd.InterstellarMetric(0.1, 1e-4, 1.0))):
    for w, h in ((512, 288), (640, 360), (720, 405), (800, 450), (880, 495), (960, 540)):
        c = curvis_amd.Camera((0.0, 5.0, np.pi / 2, 0.0), (-1.0, 0.0, 0.0), (0.0, 0.0, 1.0), 15.0, 43.0, w, h)
        t = {1: [], 2: []}
        for it in range(12):
            for v in (1, 2):
                ctx.set_option("variant", v)
                _, s = ctx.render_brute(m, c, 4096, 100.0, 0.05, download=False)
                if it >= 2: t[v].append(s.integrate_ms)
        a, b = np.median(t[1]), np.median(t[2])
        wg = ((w + 7) // 8) * ((h + 7) // 8) / 4.0
        print("%s %dx%d: %.0f workgroups = %.1f per CU: static %.3f relay %.3f ms  ratio %.3f" % (name, w, h, wg, wg / 256.0, a, b, b / a), flush=True)
