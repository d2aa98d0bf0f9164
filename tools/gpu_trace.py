#!/usr/bin/env python3
"""Collect the per-wave trace of one config-2 frame (CURVIS_TRACE_FILE diagnostics of the static kernel)."""
import os, sys
import numpy as np
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
out = os.path.join(root, "gpurun_out")
import curvis_amd
from curvis_amd import skies
ctx = curvis_amd.Context(0)
ctx.set_sky(0, curvis_amd.SphericalImage(skies.smooth(512, 256, 0))); ctx.set_sky(1, curvis_amd.SphericalImage(skies.smooth(512, 256, 1)))
cam = curvis_amd.Camera((0.0, 5.0, np.pi / 2, 0.0), (-1.0, 0.0, 0.0), (0.0, 0.0, 1.0), 15.0, 43.0, 1920, 1080)
m = curvis_amd.EllisMetric(1.0)
for _ in range(2):
    ctx.render_brute(m, cam, 4096, 100.0, 0.05, download=False)
os.environ["CURVIS_TRACE_FILE"] = os.path.join(out, "trace_config2.bin")
_, st = ctx.render_brute(m, cam, 4096, 100.0, 0.05, download=False)
print("kernel ms", st.integrate_ms)
