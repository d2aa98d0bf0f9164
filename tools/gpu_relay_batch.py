import os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import curvis_amd
from curvis_amd import skies
ctx = curvis_amd.Context(0)
ctx.set_sky(0, curvis_amd.SphericalImage(skies.smooth(512, 256, 0))); ctx.set_sky(1, curvis_amd.SphericalImage(skies.smooth(512, 256, 1)))
def cam(i, w=1920, h=1080):
    return curvis_amd.Camera((0.0, 5.0 - 0.05 * i, np.pi / 2, 0.03 * i), (-1.0, 0.0, 0.0), (0.0, 0.0, 1.0), 15.0, 43.0, w, h)
for name, m, res in (("ellis", curvis_amd.EllisMetric(1.0), (1920, 1080)), ("interstellar4k", curvis_amd.InterstellarMetric(0.1, 1e-4, 1.0), (3840, 2160))):
    for nf in (2, 3, 4, 6):
        if res[0] > 2000 and nf > 3: continue
        cams = [cam(i, *res) for i in range(nf)]
        for rnd in range(2):
            for mx in (2, 8):
                ctx.set_option("relay_max_frames", mx)
                ts = []
                for _ in range(6):
                    _, st = ctx.render_brute(m, cams, 4096 if res[0] < 2000 else 8192, 100.0, 0.05, download=False)
                    ts.append(st.integrate_ms / nf)
                print("%s %d frames relay_max_frames %d: %.3f ms/frame (relay launches %d)" % (name, nf, mx, float(np.median(ts[1:])), ctx.get_option("last_relay_launches")), flush=True)
