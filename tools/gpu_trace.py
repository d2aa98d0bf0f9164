#!/usr/bin/env python3
"""Collect the per-wave traces of one config-2 frame (CURVIS_TRACE_FILE diagnostics): static kernel
(trace_config2.bin) and relay kernel (trace_config2_relay.bin)."""
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
for variant, name in ((1, "trace_config2.bin"), (2, "trace_config2_relay.bin")):
    ctx.set_option("variant", variant)
    for _ in range(2):
        ctx.render_brute(m, cam, 4096, 100.0, 0.05, download=False)
    os.environ["CURVIS_TRACE_FILE"] = os.path.join(out, name)
    _, st = ctx.render_brute(m, cam, 4096, 100.0, 0.05, download=False)
    os.environ.pop("CURVIS_TRACE_FILE")
    print("variant", variant, "kernel ms", st.integrate_ms)
