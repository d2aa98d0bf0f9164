#!/usr/bin/env python3
"""Relay kernel: a budget of hand-overs per launch (option relay_max_parks) against launch time, configs[1] and Interstellar
1080p / 4K, interleaved rounds.  Output -> profiles/round4_relay_parks.txt"""
import os, sys
import numpy as np
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import curvis_amd
from curvis_amd import skies
ctx = curvis_amd.Context(0)
ctx.set_sky(0, curvis_amd.SphericalImage(skies.smooth(8192, 4096, 128))); ctx.set_sky(1, curvis_amd.SphericalImage(skies.smooth(8192, 4096, 32)))
ctx.set_option("relay_auto_verify", 0)
BUDGETS = (0, 4000, 3000, 2000, 1000, 500)
for name, metric, res, cap in (("ellis 1080p", curvis_amd.EllisMetric(1.0), (1920, 1080), 4096), ("interstellar 1080p", curvis_amd.InterstellarMetric(0.1, 1e-4, 1.0), (1920, 1080), 4096),
                               ("interstellar 4K", curvis_amd.InterstellarMetric(0.1, 1e-4, 1.0), (3840, 2160), 8192)):
    cam = curvis_amd.Camera((0.0, 5.0, np.pi / 2, 0.0), (-1.0, 0.0, 0.0), (0.0, 0.0, 1.0), 15.0, 43.0, res[0], res[1])
    ctx.set_option("variant", 1)
    ts = [ctx.render_brute(metric, cam, cap, 100.0, 0.05, download=False)[1].integrate_ms for _ in range(8)]
    print("%s static: %.4f ms" % (name, np.median(ts[2:])), flush=True)
    ctx.set_option("variant", 2)
    rows = {}
    for rnd in range(5):
        for b in BUDGETS:
            ctx.set_option("relay_max_parks", b)
            ts, parks = [], []
            for _ in range(8 if res[0] == 1920 else 4):
                _, st = ctx.render_brute(metric, cam, cap, 100.0, 0.05, download=False)
                ts.append(st.integrate_ms); parks.append(ctx.get_option("last_relay_parks"))
            rows.setdefault(b, []).append((float(np.median(ts[1:])), float(np.median(parks))))
    for b in BUDGETS:
        v = np.array(rows[b])
        print("%s max_parks=%d: %.4f ms (rounds %s), parks %.0f -> hand-over traffic %.1f MB" % (
            name, b, np.median(v[:, 0]), " ".join("%.3f" % x for x in v[:, 0]), np.median(v[:, 1]), np.median(v[:, 1]) * 64 * 40 * 2 / 1e6), flush=True)
