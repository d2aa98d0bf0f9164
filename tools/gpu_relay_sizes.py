#!/usr/bin/env python3
"""Static vs relay kernel over frame sizes (single-frame launches, Ellis + Interstellar, interleaved runs)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import curvis_amd
from curvis_amd import skies
ctx = curvis_amd.Context(0)
ctx.set_sky(0, curvis_amd.SphericalImage(skies.smooth(2048, 1024, 0))); ctx.set_sky(1, curvis_amd.SphericalImage(skies.smooth(2048, 1024, 1)))
ctx.set_option("relay_min_blocks", 0)
N = int(os.environ.get("N", "8"))
for name, m in (("ellis", curvis_amd.EllisMetric(1.0)), ("interstellar", curvis_amd.InterstellarMetric(0.1, 1e-4, 1.0))):
    for w, h in ((640, 360), (960, 540), (1280, 720), (1600, 900), (1920, 1080), (2560, 1440)):
        c = curvis_amd.Camera((0.0, 5.0, np.pi / 2, 0.0), (-1.0, 0.0, 0.0), (0.0, 0.0, 1.0), 15.0, 43.0, w, h)
        t = {1: [], 2: []}
        for v in (1, 2):
            ctx.set_option("variant", v); ctx.render_brute(m, c, 4096, 100.0, 0.05, download=False)
        for it in range(N):
            for v in (1, 2):
                ctx.set_option("variant", v)
                _, s = ctx.render_brute(m, c, 4096, 100.0, 0.05, download=False)
                t[v].append(s.integrate_ms)
        a, b = np.array(t[1]), np.array(t[2])
        print("%s %dx%d (%d fresh workgroups): static median %.3f min %.3f | relay median %.3f min %.3f | ratio of medians %.3f" % (
            name, w, h, ((w + 7) // 8) * ((h + 7) // 8) // 4, np.median(a), a.min(), np.median(b), b.min(), np.median(b) / np.median(a)), flush=True)
