import os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import curvis_amd._abi as A
A.LIB_PATH = sys.argv[1]
import curvis_amd
from curvis_amd import skies
ctx = curvis_amd.Context(0)
ctx.set_sky(0, curvis_amd.SphericalImage(skies.smooth(512, 256, 0))); ctx.set_sky(1, curvis_amd.SphericalImage(skies.smooth(512, 256, 1)))
for (w, h, cap) in ((256, 144, 40000), (640, 360, 4096), (1920, 1080, 4096)):
    cam = curvis_amd.Camera((0.0, 5.0, np.pi / 2, 0.0), (-1.0, 0.0, 0.0), (0.0, 0.0, 1.0), 15.0, 43.0, w, h)
    ts = []
    for _ in range(12):
        _, st = ctx.render_brute(curvis_amd.EllisMetric(1.0), cam, cap, 100.0, 0.05, download=False)
        ts.append(st.integrate_ms)
    print(os.path.basename(sys.argv[1]), w, h, "median %.3f ms" % float(np.median(ts[2:])), flush=True)
