#!/usr/bin/env python3
"""Random interleavings of the calls that touch a context's frame buffers with the overlapped download on (option
"async_download"): render calls of the three modes at changing sizes and batch lengths, with and without a host buffer,
uploads, explicit downloads, the device PNG front end, switching the option off and on.  A second context does the same
synchronously; every host buffer is compared when the contract says it is complete (the next call with a host buffer has
returned, or curvis_ctx_download_wait), and everything that reads "the last render" is compared at once.

    python tools/gpu_async_fuzz.py [steps] [seed] > gpurun_out/async_fuzz.txt      -> profiles/round5_async_fuzz.txt"""
import os
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import curvis_amd  # noqa: E402
from curvis_amd import skies  # noqa: E402

CAP, R, DELTA = 2048, 100.0, 0.05


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    A, B = curvis_amd.Context(0), curvis_amd.Context(0)
    for c in (A, B):
        c.set_sky(0, curvis_amd.SphericalImage(skies.checker(512, 256, seed=3)))
        c.set_sky(1, curvis_amd.SphericalImage(skies.checker(512, 256, seed=4)))
    A.set_option("async_download", 1)
    bufs = [curvis_amd.HostBuffer(3 * 1920 * 1080 * 3) for _ in range(3)]
    pending = None          # (buffer index, expected array) of the download possibly still in flight
    counts = {}
    checked = 0
    last_shape = None       # (n, H, W) of the last render / upload on A

    def settle():
        nonlocal pending, checked
        if pending is not None:
            i, want = pending
            got = bufs[i].array[:want.size].reshape(want.shape)
            assert np.array_equal(got, want), "buffer %d differs after its download was due" % i
            checked += 1
            pending = None

    def camera(w, h):
        l = float(rng.choice([-1.0, 1.0]) * rng.uniform(1.2, 6.0))
        pos = (0.0, l, float(rng.uniform(0.6, 2.5)), float(rng.uniform(0, 6.2)))
        return curvis_amd.Camera(pos, tuple(rng.uniform(-1, 1, 3)), (0.0, 0.0, 1.0), 15.0, 43.0, w, h)

    for step in range(steps):
        op = rng.choice(["brute", "brute", "brute_nobuf", "efficient", "direct", "upload", "download", "deflate", "wait", "toggle"])
        counts[op] = counts.get(op, 0) + 1
        metric = curvis_amd.EllisMetric(1.0) if rng.random() < 0.6 else curvis_amd.InterstellarMetric(0.1, 1e-4, 1.0)
        w, h = int(rng.choice([64, 96, 160, 320])), int(rng.choice([36, 54, 100, 200]))
        if rng.random() < 0.15:
            w, h = 1920, 1080                              # a frame whose copy takes long enough to be overtaken by the next call
        if op in ("brute", "brute_nobuf"):
            n = int(rng.integers(1, 4))
            cams = [camera(w, h) for _ in range(n)]
            want, st_b = B.render_brute(metric, cams, CAP, R, DELTA)
            if op == "brute":
                i = int(rng.integers(0, 3))
                if pending is not None and pending[0] == i:
                    A.download_wait()
                    settle()
                _, st_a = A.render_brute(metric, cams, CAP, R, DELTA, out=bufs[i].array)
                settle()                                   # the previous download: complete now that this call has returned
                pending = (i, want.copy())
            else:
                _, st_a = A.render_brute(metric, cams, CAP, R, DELTA, download=False)
            assert (st_a.rays, st_a.steps, st_a.n_pos, st_a.n_neg) == (st_b.rays, st_b.steps, st_b.n_pos, st_b.n_neg)
            last_shape = (n, h, w)
        elif op in ("efficient", "direct"):
            cam = camera(w, h)
            try:
                if op == "efficient":
                    want, _ = B.render_efficient(metric, cam, CAP, R, DELTA, 60, 40, 1e-4, 1e-4)
                else:
                    want, _ = B.render_direct(metric, cam, CAP, R, DELTA)
            except curvis_amd.CurvisError:
                continue                                   # a scene on which the reference panics (reported as an error): nothing rendered
            i = int(rng.integers(0, 3))
            if pending is not None and pending[0] == i:
                A.download_wait()
                settle()
            if op == "efficient":
                A.render_efficient(metric, cam, CAP, R, DELTA, 60, 40, 1e-4, 1e-4, out=bufs[i].array)
            else:
                A.render_direct(metric, cam, CAP, R, DELTA, out=bufs[i].array)
            settle()
            pending = (i, want.copy())
            last_shape = (1, h, w)
        elif op == "upload":
            frames = rng.integers(0, 256, size=(int(rng.integers(1, 3)), h, w, 3), dtype=np.uint8)
            A.upload_frames(frames)
            B.upload_frames(frames)
            last_shape = frames.shape[:3]
        elif op == "download" and last_shape:
            n, hh, ww = last_shape
            assert np.array_equal(A.download_frames(ww, hh, n), B.download_frames(ww, hh, n)), "explicit download differs at step %d" % step
            checked += 1
        elif op == "deflate" and last_shape:
            n, hh, ww = last_shape
            sa, _ = A.deflate_frames(ww, hh, n)
            sb, _ = B.deflate_frames(ww, hh, n)
            assert sa == sb and all(len(zlib.decompress(z)) == hh * (ww * 3 + 1) for z in sa), "streams differ at step %d" % step
            checked += 1
        elif op == "wait":
            A.download_wait()
            settle()
        elif op == "toggle":
            A.set_option("async_download", 0)              # waits for the download in flight
            settle()
            if rng.random() < 0.8:
                A.set_option("async_download", 1)
    A.download_wait()
    settle()
    print("seed %d, %d steps: %s; %d comparisons, downloads overlapped %d, no mismatch" % (
        seed, steps, ", ".join("%s %d" % kv for kv in sorted(counts.items())), checked, A.get_option("downloads_overlapped")))
    for b in bufs:
        b.close()
    A.close()
    B.close()


if __name__ == "__main__":
    main()
