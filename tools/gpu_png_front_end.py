#!/usr/bin/env python3
"""Device PNG front end (curvis_ctx_deflate_frames): kernel time and bytes per frame for the frames `curvis video` saves --
efficient-mode and brute-mode 1080p, brute-mode 4K -- in batches of 1 / 8 / 32 frames, next to the host writer's cost per
frame (`curvis_image_save_rgb8_level(-1)` into /dev/shm).  Algorithmic bytes = frame read once + stream written.
Output -> profiles/round4_png_front_end.txt"""
import os, sys, time, zlib
import numpy as np
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import curvis_amd
from curvis_amd import skies, _abi
ctx = curvis_amd.Context(0)
only = sys.argv[1] if len(sys.argv) > 1 else ""
def scene(kind, res, sky):
    sp = skies.smooth(8192, 4096, 128) if sky == "smooth" else skies.checker(8192, 4096, seed=0xC0FFEE)
    sn = skies.smooth(8192, 4096, 32) if sky == "smooth" else skies.checker(8192, 4096, seed=0xBADC0DE)
    ctx.set_sky(0, curvis_amd.SphericalImage(sp)); ctx.set_sky(1, curvis_amd.SphericalImage(sn))
    m = curvis_amd.EllisMetric(1.0) if kind != "interstellar" else curvis_amd.InterstellarMetric(0.1, 1e-4, 1.0)
    return m
CASES = [("efficient", "ellis", (1920, 1080), "smooth", (1, 8, 32)), ("efficient", "ellis", (1920, 1080), "check", (8,)),
         ("brute", "ellis", (1920, 1080), "smooth", (1, 8)), ("brute", "interstellar", (3840, 2160), "smooth", (1, 4)),
         # how far the kernels go when a launch has enough bytes to fill the chip: 64 4K frames = 1.6 GB of pixels per call
         ("efficient", "ellis", (3840, 2160), "check", (64,))]
if only == "profile":   # a short fixed workload for rocprofv3
    CASES = [("efficient", "ellis", (1920, 1080), "smooth", (8,))]
for mode, kind, res, sky, batches in CASES:
    m = scene(kind, res, sky)
    W, H = res
    for nf in batches:
        cams = [curvis_amd.Camera((0.0, 3.0 + 0.1 * k, np.pi / 2, 0.2 * k), (-1.0, 0.0, 0.0), (0.0, 0.0, 1.0), 15.0, 43.0, W, H) for k in range(nf)]
        if mode == "efficient":
            rgb, st = ctx.render_efficient(m, cams, 4096, 100.0, 0.05, 100, 100, 1e-5, 1e-5)
        else:
            rgb, st = ctx.render_brute(m, cams, 4096 if W == 1920 else 8192, 100.0, 0.05)
        buf = curvis_amd.HostBuffer(nf * W * H * 3)
        ts, walls = [], []
        for rep in range(6):
            t0 = time.perf_counter()
            streams, ms = ctx.deflate_frames(W, H, nf, out=buf.array)
            walls.append((time.perf_counter() - t0) * 1e3)
            ts.append(ms)
        ms, wall = float(np.median(ts[1:])), float(np.median(walls[1:]))
        zbytes = sum(len(z) for z in streams)
        for k in (0, nf - 1):   # spot check: decodes to the frame
            raw = np.frombuffer(zlib.decompress(streams[k]), np.uint8).reshape(H, W * 3 + 1)
            assert np.array_equal(np.cumsum(raw[:, 1:].astype(np.uint32), axis=0).astype(np.uint8).reshape(H, W, 3), rgb[k])
        # host writer on the same frames
        t0 = time.perf_counter()
        hb = 0
        for k in range(min(nf, 4)):
            p = "/dev/shm/curvis_png_probe.png"
            _abi.check(_abi.lib().curvis_image_save_rgb8_level(p.encode(), rgb[k].ctypes.data, W, H, -1))
            hb += os.path.getsize(p)
        host_ms = (time.perf_counter() - t0) * 1e3 / min(nf, 4)
        raw_b = nf * W * H * 3
        print("%s %s %dx%d %s sky, %2d frames/call: kernels %.3f ms/frame (call %.3f ms/frame incl. the six launches, one synchronisation, D2H of the streams) | "
              "stream %.3f MB/frame (%.1fx; host writer's file %.3f MB) | algorithmic %.2f MB/frame -> %.0f GB/s of 8000 | host fast writer %.2f ms/frame" % (
                  mode, kind, W, H, sky, nf, ms / nf, wall / nf, zbytes / nf / 1e6, raw_b / zbytes, hb / min(nf, 4) / 1e6,
                  (raw_b + zbytes) / nf / 1e6, (raw_b + zbytes) / (ms * 1e-3) / 1e9, host_ms), flush=True)
        buf.close()
