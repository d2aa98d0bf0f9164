"""accuracy of the v_rcp_f64 / v_rsq_f64 hardware seeds, and static-vs-persistent on a skewed workload"""
import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import curvis_amd
from curvis_amd import skies
ctx = curvis_amd.Context(0)
rng = np.random.default_rng(0)
x = np.concatenate([rng.uniform(0.5, 2.0, 1 << 20), 10.0 ** rng.uniform(-30, 30, 1 << 20)])
r = ctx.selftest_math(9, x)
rel = np.abs(r * x - 1.0)
print("v_rcp_f64: max rel err 2^%.2f, mean 2^%.2f" % (np.log2(rel.max()), np.log2(rel.mean())))
q = ctx.selftest_math(10, x)
rel = np.abs(q * q * x - 1.0) / 2
print("v_rsq_f64: max rel err 2^%.2f, mean 2^%.2f" % (np.log2(rel.max()), np.log2(rel.mean())))
sp, sn = skies.smooth(4096, 2048, 128), skies.smooth(4096, 2048, 32)
ctx.set_sky(0, curvis_amd.SphericalImage(sp)); ctx.set_sky(1, curvis_amd.SphericalImage(sn))
m = curvis_amd.EllisMetric(1.0)
for l, cap in ((5.0, 4096), (1.5, 40000), (0.3, 40000), (3.0, 40000)):
    cam = curvis_amd.Camera((0, l, np.pi / 2, 0), (-1, 0, 0), (0, 0, 1), 15, 43, 1920, 1080)
    for v in (1, 0):
        ctx.set_option("variant", v)
        ctx.render_brute(m, cam, cap, 100.0, 0.05, download=False)
        t0 = time.perf_counter()
        for _ in range(3):
            _, st = ctx.render_brute(m, cam, cap, 100.0, 0.05, download=False)
        dt = (time.perf_counter() - t0) / 3
        print("l=%.1f cap=%d variant=%d: %.2f ms/frame, %.1f G steps/s, steps/ray %.0f, capped %d" % (l, cap, v, dt * 1e3, st.steps / dt / 1e9, st.steps / st.rays, st.n_none))
