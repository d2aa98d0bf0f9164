#!/usr/bin/env python3
"""How many rays of the equatorial pixel column end with theta != fl(pi/2) (i.e. left the immediate-return argument
of sincos at some point), config 2 and the Interstellar frame."""
import os, sys, math
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import curvis_amd
from curvis_amd import skies
ctx = curvis_amd.Context(0)
ctx.set_sky(0, curvis_amd.SphericalImage(skies.smooth(512, 256, 0))); ctx.set_sky(1, curvis_amd.SphericalImage(skies.smooth(512, 256, 1)))
cam = curvis_amd.Camera((0.0, 5.0, np.pi / 2, 0.0), (-1.0, 0.0, 0.0), (0.0, 0.0, 1.0), 15.0, 43.0, 1920, 1080)
for name, m in (("ellis", curvis_amd.EllisMetric(1.0)), ("interstellar", curvis_amd.InterstellarMetric(0.1, 1e-4, 1.0))):
    rgb, st, dbg = ctx.render_brute(m, cam, 4096, 100.0, 0.05, debug=True)
    th = dbg["x"][..., 2]
    rows = np.nonzero(np.any(th == math.pi / 2, axis=1))[0]
    cols = np.nonzero(np.any(th == math.pi / 2, axis=0))[0]
    print(name, "rays ending exactly at fl(pi/2): %d, in %d rows and columns %s" % (int((th == math.pi / 2).sum()), rows.size, cols.tolist()[:8]))
    for r in (539, 540, 541):
        d = th[r] - math.pi / 2
        print("  row %d: %d of %d rays end exactly at fl(pi/2); |theta - pi/2| max %.3e, rays within 2^-20 but not equal: %d" % (
            r, int((d == 0).sum()), d.size, float(np.abs(d).max()), int(((np.abs(d) < 2.0 ** -20) & (d != 0)).sum())))
    near = (np.abs(th - math.pi / 2) < 2.0 ** -20) & (th != math.pi / 2)
    print("  whole frame: rays ending within 2^-20 of pi/2 but not at it:", int(near.sum()))
