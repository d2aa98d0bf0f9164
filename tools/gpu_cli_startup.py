#!/usr/bin/env python3
"""Wall time of `curvis image` and of a short `curvis video` as a user runs them -- process start to files on disk -- with
8192x4096 PNG backgrounds, and where it goes (CURVIS_DEBUG_TIMING=1: the binary's own phase clock on stderr).

The render kernels of the reference's default mode take ~4 ms per 1080p image; a run is decode (two 128 MiB textures),
runtime start-up, upload, render, PNG encode -- so for one image or a short video (BASELINE configs[3] is 240 frames) the
host side decides what the user waits for.

    python tools/gpu_cli_startup.py [runs] > gpurun_out/cli_startup.txt       -> profiles/round5_cli_startup.txt"""
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


def once(args, env_extra):
    env = dict(os.environ, CURVIS_DEBUG_TIMING="1", **env_extra)
    t0 = time.perf_counter()
    r = subprocess.run([BIN] + args, capture_output=True, text=True, env=env)
    dt = time.perf_counter() - t0
    if r.returncode not in (0, 101):
        raise SystemExit("curvis failed: rc %d\n%s" % (r.returncode, r.stderr[-800:]))
    phases = [ln for ln in r.stderr.splitlines() if ln.startswith("[curvis timing]") and ("worker" not in ln or "worker 0" in ln or "device workers" in ln)]
    return dt, phases


def main():
    runs = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    base = "/dev/shm" if os.path.isdir("/dev/shm") else None
    with tempfile.TemporaryDirectory(dir=base) as d:
        sky = [os.path.join(d, "sky_pos.png"), os.path.join(d, "sky_neg.png")]
        t0 = time.perf_counter()
        pngio.write_png(sky[0], skies.smooth(8192, 4096, 128))
        pngio.write_png(sky[1], skies.smooth(8192, 4096, 32))
        print("# two 8192x4096 RGBA PNG backgrounds (%.1f / %.1f MB on disk), written in %.1f s; files in %s" % (
            os.path.getsize(sky[0]) / 1e6, os.path.getsize(sky[1]) / 1e6, time.perf_counter() - t0, d))
        cam, sim, img, vid = (os.path.join(d, n) for n in ("cam.toml", "sim.toml", "img.toml", "vid.toml"))
        open(cam, "w").write("resolution_x = 1920\nresolution_y = 1080\ndiagonal = 43.0\nfocal_length = 15.0\n")
        open(sim, "w").write("ray_integration_step = 0.05\nescape_radius = 100.0\nray_integration_max_itarations = 4096\nsampling_initial_nums = 100\n"
                             "sampling_max_iterations = 50\nsampling_convergence_threshold_1 = 1e-5\nsampling_convergence_threshold_2 = 1e-5\n")
        open(img, "w").write('image_name = "img"\nt = 0.0\nl = 5.0\ntheta = 1.5707963267948966\nphi = 0.0\nforward_x = -1.0\nforward_y = 0.0\nforward_z = 0.0\n'
                             'up_x = 0.0\nup_y = 0.0\nup_z = 1.0\n')
        open(vid, "w").write('video_name = "v"\nframe_rate = 4.0\nfilepath_to_camera_path = "%s"\n' % refpaths.reference_path_file("path_orbit.csv"))
        cases = [("curvis image (default mode: efficient), 1920x1080", ["image", sky[0], sky[1], os.path.join(d, "o_img"), "-i", img, "-c", cam, "-s", sim]),
                 ("curvis image --mode brute, 1920x1080 cap 4096 (configs[1])", ["image", sky[0], sky[1], os.path.join(d, "o_img"), "-i", img, "-c", cam, "-s", sim, "--mode", "brute"]),
                 ("curvis video (default mode), path_orbit.csv at 4 fps = 240 frames of 1920x1080 (configs[3])", ["video", sky[0], sky[1], os.path.join(d, "o_vid"), "-v", vid, "-c", cam, "-s", sim])]
        try:  # the reference's README suggests JPEG star maps: the same run with 8192x4096 JPEG backgrounds (Pillow writes them)
            from PIL import Image
            rng = np.random.default_rng(3)
            jp = [os.path.join(d, "sky_pos.jpg"), os.path.join(d, "sky_neg.jpg")]
            for path, blue in zip(jp, (128, 32)):
                a = skies.smooth(8192, 4096, blue)[..., :3].astype(np.int16)
                stars = rng.random((4096, 8192)) < 0.004                                    # a few stars on a smooth sky: mostly flat blocks, as in a star map
                a[stars] = np.minimum(255, a[stars] + rng.integers(60, 200, size=(int(stars.sum()), 1)))
                Image.fromarray(np.clip(a, 0, 255).astype(np.uint8)).save(path, quality=92, subsampling=2)
            print("# two 8192x4096 JPEG backgrounds (4:2:0, quality 92; %.1f / %.1f MB)" % (os.path.getsize(jp[0]) / 1e6, os.path.getsize(jp[1]) / 1e6))
            cases.append(("curvis image (default mode), JPEG backgrounds", ["image", jp[0], jp[1], os.path.join(d, "o_img"), "-i", img, "-c", cam, "-s", sim]))
        except ImportError:
            pass
        # a star-map-like PNG pair (Pillow's adaptive filters, ~35 MB each): what a real background costs to load
        try:
            from PIL import Image
            rng = np.random.default_rng(5)
            sp = [os.path.join(d, "stars_pos.png"), os.path.join(d, "stars_neg.png")]
            yy = np.linspace(-1, 1, 4096)[:, None]
            xx = np.linspace(0, 1, 8192)[None, :]
            band = np.exp(-(yy * 3 + 0.3 * np.sin(xx * 6.28)) ** 2) * 60
            for path, tint in zip(sp, ((1.0, 0.9, 0.8), (0.8, 0.9, 1.0))):
                a = np.zeros((4096, 8192, 3), np.float32) + band[..., None] * np.array(tint, np.float32)
                n = 200000
                ys, xs, br = rng.integers(0, 4096, n), rng.integers(0, 8192, n), rng.pareto(2.0, n) * 40
                for c in range(3):
                    np.add.at(a[..., c], (ys, xs), br * rng.uniform(0.7, 1.0, n))
                a += rng.normal(0, 1.5, a.shape)
                Image.fromarray(np.clip(a, 0, 255).astype(np.uint8)).save(path)
            print("# two 8192x4096 star-map-like RGB PNG backgrounds (adaptive filters; %.1f / %.1f MB)" % (os.path.getsize(sp[0]) / 1e6, os.path.getsize(sp[1]) / 1e6))
            cases.append(("curvis image (default mode), star-map PNG backgrounds", ["image", sp[0], sp[1], os.path.join(d, "o_img"), "-i", img, "-c", cam, "-s", sim]))
        except ImportError:
            pass
        for title, args in cases:
            print("\n## " + title)
            walls, last = [], None
            for r in range(runs):
                for o in ("o_img", "o_vid"):
                    subprocess.run(["rm", "-rf", os.path.join(d, o)])
                    os.mkdir(os.path.join(d, o))
                dt, last = once(args, {})
                walls.append(dt)
            v = np.array(walls) * 1e3
            print("wall (process start to exit) %s ms, median %.0f" % (" ".join("%.0f" % x for x in v), np.median(v)))
            for ln in last:
                print("    " + ln)


if __name__ == "__main__":
    main()
