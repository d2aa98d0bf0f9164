"""Contexts per GPU x frames per launch of `curvis video --mode efficient` (the reference's default renderer), from a
measurement that can tell: >= 10 s per cell on ONE box in ONE session, interleaved rounds.

The round-4 sweep ran ~1 s per cell (2 398 frames, 0.3 s of it start-up) and showed +-10 % noise with no monotone trend.  Here
path_orbit.csv is sampled at 500 fps = 29 9xx frames of 1920x1080 per cell (the path's 60 s; the reference's own off-by-one
stops a run a few frames before the end, exit status 101 -- the frames before it are written), output into /dev/shm so that
the box's overlay file system is not what is measured, device PNG front end, 16 writer threads.

python tools/gpu_eff_contexts_sweep.py [rounds] > gpurun_out/eff_contexts_sweep.txt   -> profiles/round5_eff_contexts_sweep.txt"""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import refpaths  # noqa: E402  (the reference's own camera paths: tests/golden/paths)
from curvis_amd import paths, pngio, skies  # noqa: E402
import gpu_cli_video as V  # noqa: E402

CS = [int(v) for v in os.environ.get("SWEEP_C", "1,2,3").split(",")]
BS = [int(v) for v in os.environ.get("SWEEP_B", "8,16,32").split(",")]
CELLS = [(c, b) for c in CS for b in BS]


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    fps = float(os.environ.get("SWEEP_FPS", "500"))
    mode = os.environ.get("SWEEP_MODE", "efficient")   # "brute": the per-pixel integrator (configs[3]); use SWEEP_FPS=8 (480 frames)
    base = None  # a RAM file system when it has room for a cell's frames (~0.1 MB each), else the default temporary directory
    try:
        st = os.statvfs("/dev/shm")
        if st.f_bavail * st.f_frsize > 8 << 30:
            base = "/dev/shm"
    except OSError:
        pass
    d = tempfile.mkdtemp(prefix="curvis_sweep_", dir=base)
    sky = (os.path.join(d, "pos.png"), os.path.join(d, "neg.png"))
    pngio.write_png(sky[0], skies.smooth(4096, 2048, 128)[..., :3], level=1)
    pngio.write_png(sky[1], skies.smooth(4096, 2048, 32)[..., :3], level=1)
    sim, vid, cam = os.path.join(d, "sim.toml"), os.path.join(d, "vid.toml"), os.path.join(d, "cam.toml")
    open(sim, "w").write("escape_radius = 100.0\nray_integration_max_itarations = 4096\nray_integration_step = 0.05\nsampling_initial_nums = 100\n"
                         "sampling_max_iterations = 50\nsampling_convergence_threshold_1 = 1e-5\nsampling_convergence_threshold_2 = 1e-5\n")
    open(cam, "w").write("resolution_x = 1920\nresolution_y = 1080\ndiagonal = 43.0\nfocal_length = 15.0\n")
    # SWEEP_PATH=through: configs[4]'s poses (path_through.csv, Interstellar metric, cap 8192: every frame its own camera radius, i.e. one
    # sampler job per frame) instead of configs[3]'s orbit (every frame l = 3: the frames of a call share one job); 20 s of path
    through = os.environ.get("SWEEP_PATH", "orbit") == "through"
    extra = []
    if through:
        if "SWEEP_FPS" not in os.environ:
            fps = 1500.0
        met = os.path.join(d, "met.toml")
        open(met, "w").write("m = 0.1\na = 0.0001\nrho = 1.0\n")
        text = open(sim).read().replace("ray_integration_max_itarations = 4096", "ray_integration_max_itarations = 8192")
        open(sim, "w").write(text)
        extra = ["-m", met]
    open(vid, "w").write('video_name = "v"\nframe_rate = %r\nfilepath_to_camera_path = "%s"\n' % (
        fps, refpaths.reference_path_file("path_through.csv" if through else "path_orbit.csv")))
    print("# curvis video --mode %s, 1920x1080, %s at %g fps, ONE MI355X, device PNG front end, 16 writer threads, output in %s" % (mode, "path_through.csv (Interstellar, cap 8192)" if through else "path_orbit.csv", fps, d))
    print("# host: %d logical CPUs visible, cgroup CPU quota %s; %d interleaved rounds over the %d cells" % (os.cpu_count(), V.cpu_quota(), rounds, len(CELLS)))
    res = {cell: [] for cell in CELLS}
    for rnd in range(rounds):
        for (c, b) in CELLS:
            s = V.run(d, "c%d_b%d_r%d" % (c, b, rnd), sky, vid, cam, sim,
                      ["--gpu-png", "on", "--writers", os.environ.get("SWEEP_WRITERS", "16"), "--contexts-per-device", str(c), "--batch", str(b)] + extra, None, mode=mode)
            if not s:
                continue
            dv = s["devices"]
            res[(c, b)].append(s["frames_per_s"])
            print("round %d  C=%d batch=%-3d %8.1f frames/s (wall %6.2f s, %d frames) | render kernels %.3f ms/frame + PNG kernels %.3f | render(+deflate) call %.3f | "
                  "buffer waits %.2f s, writer drain %.2f s | samplers prefetched %d, found ready %d" % (
                      rnd, c, b, s["frames_per_s"], s["wall_s"], s["frames"], np.mean([x["kernel_ms_per_frame"] for x in dv]),
                      np.mean([x.get("gpu_png_kernel_ms_per_frame", 0.0) for x in dv]), np.mean([x["render_call_ms_per_frame"] for x in dv]),
                      sum(x["buffer_wait_s"] for x in dv), s["writer_drain_s"], sum(x.get("sampler_prefetches", 0) for x in dv),
                      sum(x.get("sampler_prefetch_hits", 0) for x in dv)), flush=True)
    print("\n| contexts per GPU | " + " | ".join("batch %d" % b for b in BS) + " |")
    print("|---|" + "---|" * len(BS))
    for c in CS:
        print("| %d | " % c + " | ".join(("%.0f (%s)" % (np.median(res[(c, b)]), ", ".join("%.0f" % v for v in res[(c, b)]))) if res[(c, b)] else "-" for b in BS) + " |")
    best = {c: max((np.median(res[(c, b)]), b) for b in BS if res[(c, b)]) for c in CS if any(res[(c, b)] for b in BS)}
    for c, (v, b) in best.items():
        print("best of C = %d: %.0f frames/s at --batch %d" % (c, v, b))
    if 1 in best and 2 in best:
        gain = best[2][0] / best[1][0] - 1.0
        print("C = 2 over C = 1 (each at its best batch): %+.1f %%  ->  %s" % (
            100 * gain, "keep two contexts per GPU in this mode" if gain > 0.05 else "drop the extra context (rule: > 5 % or fewer moving parts)"))
    subprocess.run(["rm", "-rf", d])


if __name__ == "__main__":
    main()
