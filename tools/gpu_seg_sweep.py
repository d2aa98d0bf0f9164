"""Relay kernel: segment length sweep on one frame (env SEGS, RES); variant 1 = static kernel for comparison."""
import os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import curvis_amd
from curvis_amd import skies
ctx = curvis_amd.Context(0)
ctx.set_sky(0, curvis_amd.SphericalImage(skies.smooth(512, 256, 0))); ctx.set_sky(1, curvis_amd.SphericalImage(skies.smooth(512, 256, 1)))
cam = curvis_amd.Camera((0.0, 5.0, np.pi / 2, 0.0), (-1.0, 0.0, 0.0), (0.0, 0.0, 1.0), 15.0, 43.0, 1920, 1080)
m = curvis_amd.EllisMetric(1.0)
SEGS = [int(v) for v in os.environ.get("SEGS", "400,600,800,1200,1500,2000").split(",")]
W, H = [int(v) for v in os.environ.get("RES", "1920x1080").split("x")]
cam = curvis_amd.Camera((0.0, 5.0, np.pi / 2, 0.0), (-1.0, 0.0, 0.0), (0.0, 0.0, 1.0), 15.0, 43.0, W, H)
for rnd in range(3):
    for variant, seg in [(1, 0), (2, 0)] + [(2, v) for v in SEGS]:
        ctx.set_option("variant", variant); ctx.set_option("relay_segment", seg)
        ts = []
        for _ in range(10):
            _, st = ctx.render_brute(m, cam, 4096, 100.0, 0.05, download=False)
            ts.append(st.integrate_ms)
        print("variant %d seg %4d: median %.3f min %.3f parks %d" % (variant, seg, float(np.median(ts[2:])), min(ts[2:]), ctx.get_option("last_relay_parks")), flush=True)
