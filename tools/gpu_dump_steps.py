#!/usr/bin/env python3
"""Dump per-ray step counts of config 2 (Ellis 1080p cap 4096) as u16 into gpurun_out/steps_config2.npy."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import curvis_amd
from curvis_amd import skies
ctx = curvis_amd.Context(0)
ctx.set_sky(0, curvis_amd.SphericalImage(skies.smooth(512, 256, 0))); ctx.set_sky(1, curvis_amd.SphericalImage(skies.smooth(512, 256, 1)))
cam = curvis_amd.Camera((0.0, 5.0, np.pi / 2, 0.0), (-1.0, 0.0, 0.0), (0.0, 0.0, 1.0), 15.0, 43.0, 1920, 1080)
_, st, dbg = ctx.render_brute(curvis_amd.EllisMetric(1.0), cam, 4096, 100.0, 0.05, download=False, debug=True)
steps = dbg["steps"].astype(np.uint16)
out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "steps_config2.npy")
np.save(out, steps)
print(steps.shape, steps.min(), steps.max(), steps.mean())
