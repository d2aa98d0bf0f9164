#!/usr/bin/env python3
"""Workload for profiling efficient_pixel_kernel alone: a few 32-frame 1080p render_efficient calls on one context (frames left in
HBM).  Run under rocprofv3 (tools/gpu_eff_pixel_profile.sh) or plain: prints the HIP-event time of the per-pixel kernel per frame."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import curvis_amd  # noqa: E402
from curvis_amd import skies  # noqa: E402

ctx = curvis_amd.Context(0)
ctx.set_sky(0, curvis_amd.SphericalImage(skies.smooth(8192, 4096, 128)))
ctx.set_sky(1, curvis_amd.SphericalImage(skies.smooth(8192, 4096, 32)))
cams = [curvis_amd.Camera((0.0, 3.0, np.pi / 2, 2 * np.pi * k / 240), (-1.0, 0.0, 0.0), (0.0, 0.0, 1.0), 15.0, 43.0, 1920, 1080) for k in range(32)]
ts = []
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    _, st = ctx.render_efficient(curvis_amd.EllisMetric(1.0), cams, 4096, 100.0, 0.05, 100, 100, 1e-5, 1e-5, download=False)
    ts.append(st.shade_ms / 32)
print("efficient_pixel_kernel: %.4f ms per 1080p frame (32-frame launches: %s)" % (float(np.median(ts)), " ".join("%.4f" % t for t in ts)))
