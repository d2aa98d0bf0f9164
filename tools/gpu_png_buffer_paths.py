#!/usr/bin/env python3
"""The three ways a batch of frames leaves `curvis video` with the device PNG front end, on real contents (1920x1080, 64 frames):
streams that fit the stream-sized batch buffer (smooth sky), streams that need a larger one (the pool grows, the streams are made
again), frames that do not compress (noise: host encoder through pageable memory) -- each against `--gpu-png off`, pixel for pixel.

    python tools/gpu_png_buffer_paths.py > gpurun_out/png_buffer_paths.txt      -> profiles/round5_png_buffer_paths.txt"""
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import refpaths  # noqa: E402  (the reference's own camera paths: tests/golden/paths)
from curvis_amd import paths, pngio, skies  # noqa: E402

BIN = os.path.join(ROOT, "curvis_amd", "bin", "curvis")


def main():
    rng = np.random.default_rng(7)
    with tempfile.TemporaryDirectory(dir="/dev/shm" if os.path.isdir("/dev/shm") else None) as d:
        W, H = 2048, 1024
        smooth = skies.smooth(W, H, 128)
        grain = smooth.copy()
        grain[..., :3] = np.clip(grain[..., :3].astype(np.int16) + rng.integers(-24, 25, size=(H, W, 3)), 0, 255).astype(np.uint8)
        noise = rng.integers(0, 256, size=(H, W, 4), dtype=np.uint8)
        noise[..., 3] = 255
        cam, sim, vid = (os.path.join(d, n) for n in ("cam.toml", "sim.toml", "vid.toml"))
        open(cam, "w").write("resolution_x = 1920\nresolution_y = 1080\ndiagonal = 43.0\nfocal_length = 15.0\n")
        open(sim, "w").write("ray_integration_step = 0.05\nescape_radius = 100.0\nray_integration_max_itarations = 4096\nsampling_initial_nums = 100\n"
                             "sampling_max_iterations = 50\nsampling_convergence_threshold_1 = 1e-5\nsampling_convergence_threshold_2 = 1e-5\n")
        open(vid, "w").write('video_name = "v"\nframe_rate = 1.07\nfilepath_to_camera_path = "%s"\n' % refpaths.reference_path_file("path_orbit.csv"))
        for name, sky in (("smooth sky", smooth), ("smooth sky + grain (+-24)", grain), ("noise sky", noise)):
            a, b = os.path.join(d, "a.png"), os.path.join(d, "b.png")
            pngio.write_png(a, sky, 1)
            pngio.write_png(b, sky[::-1].copy(), 1)
            res = {}
            for mode in ("on", "off"):
                out = os.path.join(d, "o_" + mode)
                subprocess.run(["rm", "-rf", out])
                os.mkdir(out)
                t0 = time.perf_counter()
                r = subprocess.run([BIN, "video", a, b, out, "-v", vid, "-c", cam, "-s", sim, "--gpu-png", mode, "--stats", os.path.join(out, "st.jsonl")],
                                   capture_output=True, text=True)
                dt = time.perf_counter() - t0
                assert r.returncode in (0, 101), r.stderr[-500:]
                res[mode] = (dt, json.load(open(os.path.join(out, "st.jsonl.summary.json"))))
            s = res["on"][1]
            frames = s["frames"]
            dv = s["devices"]
            same = all(np.array_equal(pngio.read_png(os.path.join(d, "o_on", "tmp", "frame_%d.png" % k)),
                                      pngio.read_png(os.path.join(d, "o_off", "tmp", "frame_%d.png" % k))) for k in range(frames))
            size = sum(os.path.getsize(os.path.join(d, "o_on", "tmp", "frame_%d.png" % k)) for k in range(frames)) / frames / 1e6
            print("%-28s %3d frames, %.2f MB per PNG (raw 6.22): device front end %3d frames, host fall-back %3d, buffers regrown %d x; "
                  "wall %.2f s (--gpu-png off: %.2f s); pixels equal to --gpu-png off: %s" % (
                      name, frames, size, sum(x["gpu_png_frames"] for x in dv), sum(x["gpu_png_fallback_frames"] for x in dv),
                      sum(x["gpu_png_buffer_regrown"] for x in dv), res["on"][0], res["off"][0], same), flush=True)
            assert same


if __name__ == "__main__":
    main()
