"""kernel time of the three renderers on one 1080p frame (Ellis and Interstellar, default camera, cap 4096)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import curvis_amd
from curvis_amd import skies
ctx = curvis_amd.Context(0)
ctx.set_sky(0, curvis_amd.SphericalImage(skies.smooth(4096, 2048, 128)))
ctx.set_sky(1, curvis_amd.SphericalImage(skies.smooth(4096, 2048, 32)))
cam = curvis_amd.Camera((0.0, 5.0, np.pi / 2, 0.0), (-1.0, 0.0, 0.0), (0.0, 0.0, 1.0), 15.0, 43.0, 1920, 1080)
for name, m in (("ellis", curvis_amd.EllisMetric(1.0)), ("interstellar", curvis_amd.InterstellarMetric(0.1, 1e-4, 1.0))):
    rows = {}
    for mode in ("brute", "direct", "efficient"):
        ts, steps = [], 0
        for _ in range(6):
            if mode == "brute":
                _, st = ctx.render_brute(m, cam, 4096, 100.0, 0.05, download=False)
            elif mode == "direct":
                _, st = ctx.render_direct(m, cam, 4096, 100.0, 0.05, download=False)
            else:
                _, st = ctx.render_efficient(m, cam, 4096, 100.0, 0.05, 100, 100, 1e-5, 1e-5, download=False)
            ts.append(st.kernel_ms); steps = st.steps
        rows[mode] = (float(np.median(ts[2:])), steps)
    d, _ = ctx.render_direct(m, cam, 4096, 100.0, 0.05)
    e, _ = ctx.render_efficient(m, cam, 4096, 100.0, 0.05, 100, 100, 1e-5, 1e-5)
    diff = np.abs(d.astype(int) - e.astype(int)).max(axis=2)
    print("%s 1920x1080: " % name + ", ".join("%s %.2f ms (%.3g steps, %.0f G steps/s)" % (k, v[0], v[1], v[1] / v[0] / 1e6) for k, v in rows.items()) +
          "; direct vs efficient pixels: identical %.4f, <= 1 LSB %.4f, max %d" % ((diff == 0).mean(), (diff <= 1).mean(), diff.max()), flush=True)
ctx.close()
