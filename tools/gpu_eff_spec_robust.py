#!/usr/bin/env python3
"""Single-image efficient render: speculation settings (first-launch depth, later depth) over several cameras."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import curvis_amd
from curvis_amd import skies
ctx = curvis_amd.Context(0)
ctx.set_sky(0, curvis_amd.SphericalImage(skies.smooth(2048, 1024, 0))); ctx.set_sky(1, curvis_amd.SphericalImage(skies.smooth(2048, 1024, 1)))
ARGS = (40000, 100.0, 0.05, 100, 100, 1e-5, 1e-5)
combos = [(3, 6), (7, 8), (8, 8), (6, 8), (7, 6), (5, 7)]
tot = {c: [] for c in combos}
for name, metric in (("ellis", curvis_amd.EllisMetric(1.0)), ("ellis rho 2", curvis_amd.EllisMetric(2.0)), ("interstellar", curvis_amd.InterstellarMetric(0.1, 1e-4, 1.0)), ("interstellar a=0.5", curvis_amd.InterstellarMetric(0.2, 0.5, 1.0))):
    for l in (5.0, 2.5, 12.0, -4.0, 30.0):
        cam = curvis_amd.Camera((0.0, l, np.pi / 2, 0.3), (-1.0 if l > 0 else 1.0, 0.05, 0.02), (0.0, 0.0, 1.0), 15.0, 43.0, 1920, 1080)
        row = []
        for first, spec in combos:
            ctx.set_option("sampling_speculation_first", first); ctx.set_option("sampling_speculation", spec)
            try:
                ctx.render_efficient(metric, cam, *ARGS, download=False)
                t0 = time.perf_counter()
                for _ in range(2):
                    ctx.render_efficient(metric, cam, *ARGS, download=False)
                dt = (time.perf_counter() - t0) / 2 * 1e3
                row.append("%d/%d: %.2f ms L%d" % (first, spec, dt, ctx.get_option("last_sampling_launches")))
                tot[(first, spec)].append(dt)
            except curvis_amd.CurvisError as e:
                row.append("%d/%d: error" % (first, spec))
        print(name, "l=%g" % l, " | ".join(row), flush=True)
print("mean ms per image:", {("%d/%d" % c): round(float(np.mean(v)), 3) for c, v in tot.items()})
