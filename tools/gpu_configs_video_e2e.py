#!/usr/bin/env python3
"""BASELINE configs[3] and configs[4] IN FULL through the binary on one GPU, in the reference's default renderer (`--mode efficient`,
what its own CLI runs: src/rendering.rs:299-307) and in the per-pixel mode the headline metric is about (`--mode brute`): files in ->
PNG frames out, the binary's own choices for contexts, batch and writers.
  configs[3]: Ellis, path_orbit.csv, 240 frames (4 fps), 1920x1080, cap 4096
  configs[4]: Interstellar (m 0.1, a 1e-4, rho 1), path_through.csv, 480 frames (24 fps), 3840x2160, cap 8192
python tools/gpu_configs_video_e2e.py > gpurun_out/configs_video_e2e.txt   -> profiles/<round>_configs_video_e2e.txt"""
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import refpaths  # noqa: E402
from curvis_amd import pngio, skies  # noqa: E402

BIN = os.path.join(ROOT, "curvis_amd", "bin", "curvis")
CONFIGS = [("configs[3]", "ellis", "path_orbit.csv", 4.0, 240, (1920, 1080), 4096),
           ("configs[4]", "interstellar", "path_through.csv", 24.0, 480, (3840, 2160), 8192)]
SIM = ("escape_radius = 100.0\nray_integration_max_itarations = %d\nray_integration_step = 0.05\nsampling_initial_nums = 100\n"
       "sampling_max_iterations = 50\nsampling_convergence_threshold_1 = 1e-5\nsampling_convergence_threshold_2 = 1e-5\n")


def main():
    base = "/dev/shm" if os.path.isdir("/dev/shm") else None
    d = tempfile.mkdtemp(prefix="curvis_cfg_", dir=base)
    try:
        pngio.write_png(os.path.join(d, "pos.png"), skies.smooth(8192, 4096, 128)[..., :3], level=1)
        pngio.write_png(os.path.join(d, "neg.png"), skies.smooth(8192, 4096, 32)[..., :3], level=1)
        print("# %s | skies 8192x4096 (smooth), output in %s" % (time.strftime("%Y-%m-%d"), d))
        for name, metric, csv, fps, n_frames, res, cap in CONFIGS:
            open(os.path.join(d, "sim.toml"), "w").write(SIM % cap)
            open(os.path.join(d, "cam.toml"), "w").write("resolution_x = %d\nresolution_y = %d\ndiagonal = 43.0\nfocal_length = 15.0\n" % res)
            open(os.path.join(d, "vid.toml"), "w").write('video_name = "v"\nframe_rate = %r\nfilepath_to_camera_path = "%s"\n' % (fps, refpaths.reference_path_file(csv)))
            extra = []
            if metric == "interstellar":
                open(os.path.join(d, "met.toml"), "w").write("m = 0.1\na = 0.0001\nrho = 1.0\n")
                extra = ["-m", os.path.join(d, "met.toml")]
            for mode in os.environ.get("CFG_MODES", "efficient,brute").split(","):
                for rep in range(2):
                    out = os.path.join(d, "out")
                    shutil.rmtree(out, ignore_errors=True)
                    os.mkdir(out)
                    st = os.path.join(d, "st.jsonl")
                    t0 = time.perf_counter()
                    r = subprocess.run([BIN, "video", os.path.join(d, "pos.png"), os.path.join(d, "neg.png"), out, "-v", os.path.join(d, "vid.toml"),
                                        "-s", os.path.join(d, "sim.toml"), "-c", os.path.join(d, "cam.toml"), "--mode", mode, "--stats", st] + extra + os.environ.get("CFG_EXTRA", "").split(),
                                       capture_output=True, text=True, timeout=1200)
                    wall = time.perf_counter() - t0
                    if r.returncode not in (0, 101):
                        print("%s %s: exit %d %s" % (name, mode, r.returncode, r.stderr[-300:]))
                        break
                    s = json.load(open(st + ".summary.json"))
                    n = len([f for f in os.listdir(os.path.join(out, "tmp")) if f.endswith(".png")])
                    steps = sum(json.loads(ln)["steps"] for ln in open(st))
                    dv = s["devices"]
                    print("%s %-9s run %d: %d frames of %dx%d on disk in %.2f s of process wall (%.2f s from the first context to the last file: %.1f frames/s) | "
                          "%d worker(s), kernels %.3f ms/frame, %.3g Euler steps%s" % (
                              name, mode, rep, n, res[0], res[1], wall, s["wall_s"], s["frames_per_s"], len(dv),
                              sum(x["kernel_ms_per_frame"] * x["frames"] for x in dv) / max(1, s["frames"]), steps,
                              " = %.0f G ray-steps/s end to end" % (steps / s["wall_s"] / 1e9) if mode == "brute" else " (the sampler's)"), flush=True)
    finally:
        shutil.rmtree(d, ignore_errors=True)


if __name__ == "__main__":
    main()
