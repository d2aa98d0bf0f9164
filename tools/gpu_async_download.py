#!/usr/bin/env python3
"""Where the time of a render call with a host frame goes: frame left in HBM / synchronous download / overlapped download
(option "async_download"), 1920x1080 Ellis (BASELINE configs[1]) single-frame launches, interleaved rounds on one box.

Per mode: wall time per call, the kernel's HIP-event time, the host time of the call beyond the kernel, and for the overlapped
mode the time curvis_ctx_download_wait takes right after a call (= what is left of the DMA when the kernels are done).

    python tools/gpu_async_download.py [rounds] > gpurun_out/async_download.txt   -> profiles/round5_async_download.txt"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import curvis_amd  # noqa: E402
from curvis_amd import skies  # noqa: E402

W, H, CAP, R, DELTA = 1920, 1080, 4096, 100.0, 0.05
N = int(os.environ.get("CALLS", "40"))


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    ctx = curvis_amd.Context(0)
    ctx.set_sky(0, curvis_amd.SphericalImage(skies.smooth(8192, 4096, 128)))
    ctx.set_sky(1, curvis_amd.SphericalImage(skies.smooth(8192, 4096, 32)))
    metric = curvis_amd.EllisMetric(1.0)
    cam = curvis_amd.Camera((0.0, 5.0, np.pi / 2, 0.0), (-1.0, 0.0, 0.0), (0.0, 0.0, 1.0), 15.0, 43.0, W, H)
    bufs = [curvis_amd.HostBuffer(W * H * 3) for _ in range(2)]
    print("# %dx%d Ellis cap %d, %d single-frame calls per cell, %d interleaved rounds; HSA_ENABLE_SDMA=%s" % (
        W, H, CAP, N, rounds, os.environ.get("HSA_ENABLE_SDMA", "(unset)")))
    for _ in range(3):
        ctx.render_brute(metric, cam, CAP, R, DELTA, download=False)
    res = {}

    def run(mode):
        kern, tail = [], []
        if mode == "overlapped":
            ctx.set_option("async_download", 1)
        ctx.render_brute(metric, cam, CAP, R, DELTA, download=mode != "hbm", out=None if mode == "hbm" else bufs[0].array)
        ctx.download_wait()
        t0 = time.perf_counter()
        for k in range(N):
            _, st = ctx.render_brute(metric, cam, CAP, R, DELTA, download=mode != "hbm", out=None if mode == "hbm" else bufs[k & 1].array)
            kern.append(st.integrate_ms)
        ctx.download_wait()
        wall = (time.perf_counter() - t0) / N * 1e3
        if mode == "overlapped":  # how long the copy still takes when the call returns (separate calls, outside the timing above)
            for k in range(8):
                ctx.render_brute(metric, cam, CAP, R, DELTA, out=bufs[k & 1].array)
                t1 = time.perf_counter()
                ctx.download_wait()
                tail.append((time.perf_counter() - t1) * 1e3)
            ctx.set_option("async_download", 0)
        return wall, float(np.mean(kern)), (float(np.median(tail)) if tail else None)

    for rnd in range(rounds):
        for mode in ("hbm", "synchronous", "overlapped"):
            wall, kern, tail = run(mode)
            res.setdefault(mode, []).append((wall, kern))
            print("round %d  %-12s wall %7.3f ms per call | kernel %7.3f ms | call beyond the kernel %6.3f ms%s" % (
                rnd, mode, wall, kern, wall - kern, "" if tail is None else " | download_wait right after a call %.3f ms" % tail), flush=True)
    print()
    base = np.median([w for w, _ in res["hbm"]])
    for mode in ("hbm", "synchronous", "overlapped"):
        w = np.median([x for x, _ in res[mode]])
        k = np.median([x for _, x in res[mode]])
        print("%-12s median wall %7.3f ms per call (%+.3f ms, %+.2f %% over the frame left in HBM), kernel %7.3f ms" % (mode, w, w - base, (w / base - 1) * 100, k))
    for b in bufs:
        b.close()
    ctx.close()


if __name__ == "__main__":
    main()
