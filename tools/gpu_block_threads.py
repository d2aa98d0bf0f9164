#!/usr/bin/env python3
"""Workgroup size of the static / relay kernels ("block_threads" = 64 / 128 / 256) on one config-2 frame, a
six-frame launch and a 960x540 frame: kernel ms (median of 7) and equality of the frames."""
import os, sys
import numpy as np
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import curvis_amd
from curvis_amd import skies
ctx = curvis_amd.Context(0)
ctx.set_sky(0, curvis_amd.SphericalImage(skies.smooth(512, 256, 0))); ctx.set_sky(1, curvis_amd.SphericalImage(skies.smooth(512, 256, 1)))
def cam(w, h):
    return curvis_amd.Camera((0.0, 5.0, np.pi / 2, 0.0), (-1.0, 0.0, 0.0), (0.0, 0.0, 1.0), 15.0, 43.0, w, h)
for name, m, cap in (("ellis", curvis_amd.EllisMetric(1.0), 4096), ("interstellar", curvis_amd.InterstellarMetric(0.1, 1e-4, 1.0), 4096)):
    for (w, h, nf) in ((1920, 1080, 1), (960, 540, 1), (1920, 1080, 6)):
        ref = None
        for variant in (1, 2):
            if nf > 1 and variant == 2:
                continue
            for bt in (256, 128, 64):
                ctx.set_option("variant", variant); ctx.set_option("block_threads", bt)
                ts = []
                for _ in range(8):
                    if nf == 1:
                        img, st = ctx.render_brute(m, cam(w, h), cap, 100.0, 0.05, download=False)
                    else:
                        img, st = ctx.render_brute(m, [cam(w, h)] * nf, cap, 100.0, 0.05, download=False)
                    ts.append(st.integrate_ms / nf)
                if nf == 1:
                    img, st = ctx.render_brute(m, cam(w, h), cap, 100.0, 0.05)
                    if ref is None: ref = img
                    same = bool(np.array_equal(ref, img))
                else:
                    same = None
                print("%s %dx%d x%d variant %d block %3d: median %.3f min %.3f ms/frame  same=%s" % (
                    name, w, h, nf, variant, bt, float(np.median(ts[1:])), min(ts[1:]), same), flush=True)
