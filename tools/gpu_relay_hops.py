#!/usr/bin/env python3
"""Relay kernel: hand-overs per tile (option relay_max_hops) and segment length against launch time and number of parks,
configs[1] and Interstellar 1080p, one process, interleaved.  Output: profiles/round4_relay_hops.txt"""
import os, sys
import numpy as np
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import curvis_amd
from curvis_amd import skies
ctx = curvis_amd.Context(0)
ctx.set_sky(0, curvis_amd.SphericalImage(skies.smooth(8192, 4096, 128))); ctx.set_sky(1, curvis_amd.SphericalImage(skies.smooth(8192, 4096, 32)))
cam = curvis_amd.Camera((0.0, 5.0, np.pi / 2, 0.0), (-1.0, 0.0, 0.0), (0.0, 0.0, 1.0), 15.0, 43.0, 1920, 1080)
ctx.set_option("relay_auto_verify", 0)
rows = {}
CASES = [(h, s) for h in (0, 1, 2, 3) for s in (0,)] + [(1, 800), (1, 1600), (0, 1600), (2, 1600)]
for name, metric in (("ellis", curvis_amd.EllisMetric(1.0)), ("interstellar", curvis_amd.InterstellarMetric(0.1, 1e-4, 1.0))):
    ctx.set_option("variant", 1)
    ts = [ctx.render_brute(metric, cam, 4096, 100.0, 0.05, download=False)[1].integrate_ms for _ in range(10)]
    print("%s static: %.4f ms" % (name, np.median(ts[2:])), flush=True)
    ctx.set_option("variant", 2)
    for rnd in range(4):
        for hops, seg in CASES:
            ctx.set_option("relay_max_hops", hops); ctx.set_option("relay_segment", seg)
            ts, parks = [], []
            for _ in range(10):
                _, st = ctx.render_brute(metric, cam, 4096, 100.0, 0.05, download=False)
                ts.append(st.integrate_ms); parks.append(ctx.get_option("last_relay_parks"))
            rows.setdefault((name, hops, seg), []).append((float(np.median(ts[2:])), float(np.median(parks))))
    for hops, seg in CASES:
        v = np.array(rows[(name, hops, seg)])
        print("%s max_hops=%d segment=%d: %.4f ms (rounds %s), parks %.0f -> hand-over traffic %.1f MB" % (
            name, hops, seg, np.median(v[:, 0]), " ".join("%.3f" % x for x in v[:, 0]), np.median(v[:, 1]), np.median(v[:, 1]) * 64 * 40 * 2 / 1e6), flush=True)
# correctness of the chosen policy: every launch checked against the static kernel
ctx.set_option("relay_verify", 1)
for hops in (0, 1, 2):
    ctx.set_option("relay_max_hops", hops); ctx.set_option("relay_segment", 0)
    for metric in (curvis_amd.EllisMetric(1.0), curvis_amd.InterstellarMetric(0.1, 1e-4, 1.0)):
        for _ in range(5):
            ctx.render_brute(metric, cam, 4096, 100.0, 0.05, download=False)
print("relay_verify over 30 launches (hops 0/1/2, both metrics): no mismatch; relay_mismatches =", ctx.get_option("relay_mismatches"))
