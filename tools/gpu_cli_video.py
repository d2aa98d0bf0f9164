"""end-to-end timing of the curvis binary on config 4 (path_orbit.csv @ 4 fps = 240 frames, 1920x1080):
sky PNGs in, 240 frame PNGs out, for the efficient (reference CLI) and brute (per-pixel) renderers."""
import os, subprocess, sys, time, tempfile, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from curvis_amd import paths, pngio, skies
BIN = os.path.join(ROOT, "curvis_amd", "bin", "curvis")
d = tempfile.mkdtemp(prefix="curvis_video_")
t0 = time.perf_counter()
pngio.write_png(os.path.join(d, "pos.png"), skies.smooth(4096, 2048, 128)[..., :3], level=1)
pngio.write_png(os.path.join(d, "neg.png"), skies.smooth(4096, 2048, 32)[..., :3], level=1)
print("wrote skies in %.1f s" % (time.perf_counter() - t0))
open(os.path.join(d, "vid.toml"), "w").write('video_name = "v"\nframe_rate = 4.0\nfilepath_to_camera_path = "%s"\n' % paths.path_file("path_orbit.csv"))
open(os.path.join(d, "cam.toml"), "w").write("resolution_x = 1920\nresolution_y = 1080\ndiagonal = 43.0\nfocal_length = 15.0\n")
open(os.path.join(d, "sim.toml"), "w").write("escape_radius = 100.0\nray_integration_max_itarations = 4096\nray_integration_step = 0.05\nsampling_initial_nums = 100\nsampling_max_iterations = 50\nsampling_convergence_threshold_1 = 1e-5\nsampling_convergence_threshold_2 = 1e-5\n")
for mode, extra in (("efficient", ["--batch", "16", "--writers", "32"]), ("brute", ["--batch", "8", "--writers", "32"]), ("efficient", ["--batch", "16", "--writers", "1"])):
    out = os.path.join(d, "out_" + mode + extra[-1])
    os.mkdir(out)
    t0 = time.perf_counter()
    r = subprocess.run([BIN, "video", os.path.join(d, "pos.png"), os.path.join(d, "neg.png"), out, "-v", os.path.join(d, "vid.toml"), "-c", os.path.join(d, "cam.toml"), "-s", os.path.join(d, "sim.toml"), "--mode", mode, "--stats", os.path.join(out, "st.jsonl")] + extra, capture_output=True, text=True)
    dt = time.perf_counter() - t0
    n = len([f for f in os.listdir(os.path.join(out, "tmp")) if f.endswith(".png")])
    lines = [json.loads(l) for l in open(os.path.join(out, "st.jsonl"))]
    batches = {}
    for l in lines:
        batches[(l["device"], l["frame"] // 1)] = l
    gpu_ms = sum(l["batch_kernel_ms"] / l["batch_frames"] for l in lines)
    print("mode %-9s %s: rc %d, %d frames in %.2f s wall (%.1f ms/frame end to end), GPU kernels %.1f ms/frame" % (mode, " ".join(extra), r.returncode, n, dt, dt * 1e3 / max(n, 1), gpu_ms / max(n, 1)))
    if r.returncode != 0:
        print(r.stderr[-500:])
