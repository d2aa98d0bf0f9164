#!/usr/bin/env python3
"""Host-side phase timing of the efficient renderer (CURVIS_DEBUG_TIMING): 1, 8 and 30 frames."""
import os, sys, time
import numpy as np
os.environ["CURVIS_DEBUG_TIMING"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import curvis_amd
from curvis_amd import skies
ctx = curvis_amd.Context(0)
ctx.set_sky(0, curvis_amd.SphericalImage(skies.smooth(2048, 1024, 0))); ctx.set_sky(1, curvis_amd.SphericalImage(skies.smooth(2048, 1024, 1)))
def cam(l, phi=0.0):
    return curvis_amd.Camera((0.0, l, np.pi / 2, phi), (-1.0, 0.0, 0.0), (0.0, 0.0, 1.0), 15.0, 43.0, 1920, 1080)
ARGS = (40000, 100.0, 0.05, 100, 100, 1e-5, 1e-5)
m = curvis_amd.EllisMetric(1.0)
for n in (1, 8, 30):
    cams = cam(5.0) if n == 1 else [cam(5.0 - 0.1 * i, 0.05 * i) for i in range(n)]
    ctx.render_efficient(m, cams, *ARGS, download=False)
    t0 = time.perf_counter()
    _, st = ctx.render_efficient(m, cams, *ARGS, download=False)
    dt = (time.perf_counter() - t0) * 1e3
    print("frames %d: wall %.3f ms (%.3f per frame); stats total %.3f integrate %.3f pixel %.3f" % (n, dt, dt / n, st.total_ms, st.integrate_ms, st.shade_ms), flush=True)
