#!/usr/bin/env python3
"""efficient_pixel_kernel timing: batch of 16 1080p frames, with big (8192x4096) and small skies."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import curvis_amd
from curvis_amd import skies
ctx = curvis_amd.Context(0)
def cam(l, phi=0.0):
    return curvis_amd.Camera((0.0, l, np.pi / 2, phi), (-1.0, 0.0, 0.0), (0.0, 0.0, 1.0), 15.0, 43.0, 1920, 1080)
ARGS = (40000, 100.0, 0.05, 100, 100, 1e-5, 1e-5)
cams = [cam(5.0 - 0.1 * i, 0.05 * i) for i in range(16)]
for sw, sh in ((8192, 4096), (1024, 512)):
    ctx.set_sky(0, curvis_amd.SphericalImage(skies.smooth(sw, sh, 0))); ctx.set_sky(1, curvis_amd.SphericalImage(skies.smooth(sw, sh, 1)))
    ctx.render_efficient(curvis_amd.EllisMetric(1.0), cams, *ARGS, download=False)
    _, st = ctx.render_efficient(curvis_amd.EllisMetric(1.0), cams, *ARGS, download=False)
    print("sky %dx%d: pixel kernel %.3f ms/frame, sampling %.3f ms/batch, total %.3f" % (sw, sh, st.shade_ms / 16, st.integrate_ms, st.total_ms), flush=True)
