import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import curvis_amd
from curvis_amd import skies
ctx = curvis_amd.Context(0)
ctx.set_sky(0, curvis_amd.SphericalImage(skies.smooth(2048, 1024, 0))); ctx.set_sky(1, curvis_amd.SphericalImage(skies.smooth(2048, 1024, 1)))
cam = curvis_amd.Camera((0.0, 5.0, np.pi / 2, 0.0), (-1.0, 0.0, 0.0), (0.0, 0.0, 1.0), 15.0, 43.0, 1920, 1080)
m = curvis_amd.EllisMetric(1.0)
ARGS = (40000, 100.0, 0.05, 100, 100, 1e-5, 1e-5)
for rep in range(8):
    t0 = time.perf_counter()
    _, st = ctx.render_efficient(m, cam, *ARGS, download=False)
    print("call %d: wall %.3f ms, stats total %.3f integrate %.3f" % (rep, (time.perf_counter() - t0) * 1e3, st.total_ms, st.integrate_ms), flush=True)
