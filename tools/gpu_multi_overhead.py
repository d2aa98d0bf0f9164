"""host-side time of multi-frame relay launches beyond their kernels (wall - HIP events), per call"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import curvis_amd
from curvis_amd import skies
ctx = curvis_amd.Context(0)
ctx.set_sky(0, curvis_amd.SphericalImage(skies.smooth(8192, 4096, 128)))
ctx.set_sky(1, curvis_amd.SphericalImage(skies.smooth(8192, 4096, 32)))
cam = curvis_amd.Camera((0.0, 5.0, np.pi / 2, 0.0), (-1.0, 0.0, 0.0), (0.0, 0.0, 1.0), 15.0, 43.0, 1920, 1080)
m = curvis_amd.EllisMetric(1.0)
for auto in (1, 0):
    for variant in (-1, 1):
        ctx.set_option("relay_auto_verify", auto)
        ctx.set_option("variant", variant)
        for nf in (1, 6):
            rows = []
            for k in range(7):
                t0 = time.perf_counter()
                _, st = ctx.render_brute(m, cam if nf == 1 else [cam] * nf, 4096, 100.0, 0.05, download=False)
                dt = (time.perf_counter() - t0) * 1e3
                rows.append("%.2f/%.2f/%d" % (dt, st.integrate_ms, ctx.get_option("last_relay_launches")))
            print("auto_verify %d variant %2d frames %d: wall/kernel/relay-launches per call: %s" % (auto, variant, nf, "  ".join(rows)), flush=True)
ctx.close()
