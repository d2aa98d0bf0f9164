#!/usr/bin/env python3
"""Wave trace of the relay kernel on config 2 for a given segment length (env SEG)."""
import os, sys
import numpy as np
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import curvis_amd
from curvis_amd import skies
ctx = curvis_amd.Context(0)
ctx.set_sky(0, curvis_amd.SphericalImage(skies.smooth(512, 256, 0))); ctx.set_sky(1, curvis_amd.SphericalImage(skies.smooth(512, 256, 1)))
cam = curvis_amd.Camera((0.0, 5.0, np.pi / 2, 0.0), (-1.0, 0.0, 0.0), (0.0, 0.0, 1.0), 15.0, 43.0, 1920, 1080)
m = curvis_amd.EllisMetric(1.0)
ctx.set_option("variant", 2)
for seg in [int(v) for v in os.environ.get("SEGS", "256,1024").split(",")]:
    ctx.set_option("relay_segment", seg)
    for _ in range(2):
        ctx.render_brute(m, cam, 4096, 100.0, 0.05, download=False)
    os.environ["CURVIS_TRACE_FILE"] = os.path.join(root, "gpurun_out", "relay_trace_%d.bin" % seg)
    _, st = ctx.render_brute(m, cam, 4096, 100.0, 0.05, download=False)
    os.environ.pop("CURVIS_TRACE_FILE")
    print("seg", seg, "kernel ms", st.integrate_ms, "parks", ctx.get_option("last_relay_parks"))
