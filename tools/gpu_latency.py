#!/usr/bin/env python3
"""Per-step latency of a lone wave (escape_angle_kernel): n identical rays, wall time of the call / steps."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import curvis_amd
ctx = curvis_amd.Context(0)
for name, m in (("ellis", curvis_amd.EllisMetric(1.0)), ("interstellar", curvis_amd.InterstellarMetric(0.1, 1e-4, 1.0))):
    for n in (1, 64, 1024, 8192, 65536, 262144, 393216):
        al = np.full(n, 0.7)
        ctx.compute_escape_angles_range(m, 5.0, al, 0.05, 40000, 100.0)
        best = 1e9
        for _ in range(5):
            t0 = time.perf_counter()
            ang, sp, st = ctx.compute_escape_angles_range(m, 5.0, al, 0.05, 40000, 100.0)
            best = min(best, time.perf_counter() - t0)
        print("%s n=%d: steps %d, call %.3f ms, %.1f ns per step of the slowest wave, %.1f G ray-steps/s" % (
            name, n, st[0], best * 1e3, best / st[0] * 1e9, n * float(st[0]) / best / 1e9), flush=True)
