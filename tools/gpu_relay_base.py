#!/usr/bin/env python3
"""Relay kernel with hand-overs disabled (segment >= cap) against the static kernel, interleaved: the cost of the
relay kernel's structure alone."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import curvis_amd
from curvis_amd import skies
ctx = curvis_amd.Context(0)
ctx.set_sky(0, curvis_amd.SphericalImage(skies.smooth(2048, 1024, 0))); ctx.set_sky(1, curvis_amd.SphericalImage(skies.smooth(2048, 1024, 1)))
ctx.set_option("relay_min_blocks", 0)
m = curvis_amd.EllisMetric(1.0)
c = curvis_amd.Camera((0.0, 5.0, np.pi / 2, 0.0), (-1.0, 0.0, 0.0), (0.0, 0.0, 1.0), 15.0, 43.0, 1920, 1080)
t = {}
confs = [("static", 1, 0), ("relay seg 4096", 2, 4096), ("relay seg 1024", 2, 1024), ("relay seg 512", 2, 512)]
for name, v, seg in confs:
    ctx.set_option("variant", v); ctx.set_option("relay_segment", seg); ctx.render_brute(m, c, 4096, 100.0, 0.05, download=False); t[name] = []
for it in range(10):
    for name, v, seg in confs:
        ctx.set_option("variant", v); ctx.set_option("relay_segment", seg)
        _, s = ctx.render_brute(m, c, 4096, 100.0, 0.05, download=False)
        t[name].append(s.integrate_ms)
for name in t:
    a = np.array(t[name]); print("%-16s median %.3f min %.3f max %.3f" % (name, np.median(a), a.min(), a.max()), flush=True)
