#!/usr/bin/env python3
"""A/B two builds of libcurvis_hip.so on the same box, interleaved: build/ab/old.so vs build/ab/new.so.
Each measurement runs in a child process (the library path is fixed at import)."""
import os, subprocess, sys
import numpy as np
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys
import numpy as np
sys.path.insert(0, %r)
import curvis_amd._abi as A
A.LIB_PATH = sys.argv[1]
import ctypes
_L = ctypes.CDLL(A.LIB_PATH)   # an older build may lack the newest entry points: bind what it has
for _n in list(A.SYMBOLS):
    if not hasattr(_L, _n): A.SYMBOLS.pop(_n)
import curvis_amd
from curvis_amd import skies
ctx = curvis_amd.Context(0)
ctx.set_sky(0, curvis_amd.SphericalImage(skies.smooth(512, 256, 0))); ctx.set_sky(1, curvis_amd.SphericalImage(skies.smooth(512, 256, 1)))
cam = curvis_amd.Camera((0.0, 5.0, np.pi / 2, 0.0), (-1.0, 0.0, 0.0), (0.0, 0.0, 1.0), 15.0, 43.0, 1920, 1080)
out = []
for m, nf in ((curvis_amd.EllisMetric(1.0), 1), (curvis_amd.EllisMetric(1.0), 6), (curvis_amd.InterstellarMetric(0.1, 1e-4, 1.0), 1)):
    ts = []
    for _ in range(12):
        _, st = ctx.render_brute(m, cam if nf == 1 else [cam] * nf, 4096, 100.0, 0.05, download=False)
        ts.append(st.integrate_ms / nf)
    out.append(float(np.median(ts[2:])))
ts = []
for _ in range(8):
    _, st = ctx.render_efficient(curvis_amd.EllisMetric(1.0), cam, 40000, 100.0, 0.05, 100, 100, 1e-5, 1e-5, download=False)
    ts.append(st.integrate_ms)
out.append(float(np.median(ts[2:])))   # efficient renderer, one image: sampling kernels (lone waves)
cams = [curvis_amd.Camera((0.0, 3.0 + 0.05 * k, np.pi / 2, 0.1 * k), (-1.0, 0.0, 0.0), (0.0, 0.0, 1.0), 15.0, 43.0, 1920, 1080) for k in range(32)]
ts = []
for _ in range(8):
    _, st = ctx.render_efficient(curvis_amd.EllisMetric(1.0), cams, 4096, 100.0, 0.05, 100, 100, 1e-5, 1e-5, download=False)
    ts.append(st.shade_ms / 32)
out.append(float(np.median(ts[2:])))   # efficient renderer, 32 frames per call: the per-pixel kernel, per frame
print(" ".join("%%.4f" %% v for v in out))
''' % root
res = {"old": [], "new": []}
NAMES = os.environ.get("LIBS", "old,new").split(",")
res = {n: [] for n in NAMES}
for rnd in range(int(os.environ.get("ROUNDS", "4"))):
    for name in NAMES:
        r = subprocess.run([sys.executable, "-c", CHILD, os.path.join(root, "build", "ab", name + ".so")], capture_output=True, text=True)
        vals = [float(v) for v in r.stdout.split()] if r.returncode == 0 else None
        print(name, r.stdout.strip() if vals else r.stderr[-400:], flush=True)
        if vals: res[name].append(vals)
for name in res:
    print(name, "median over rounds [ellis x1, ellis x6 per frame, interstellar x1, efficient-image sampling kernels, efficient pixel kernel per frame of a 32-frame call] ms:", np.median(np.array(res[name]), axis=0))
