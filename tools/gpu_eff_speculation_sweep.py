"""Speculation depth of the efficient renderer's sampler at the NEW operating point of `curvis video --mode efficient`
(4 contexts per GPU, 32 frames per launch: profiles/round5_eff_contexts_sweep.txt).  The sampler evaluates a speculative
subtree below every refined interval so that it needs few launches (efficient_host.h: depth 4 / first launch 3 for batches of
more than five frames) -- chosen when ONE context paid every launch's latency alone.  With four contexts overlapping each
other, speculative points cost GPU time the others could use: is a shallower tree faster now?  29 970 frames per cell as in the
contexts sweep.  python tools/gpu_eff_speculation_sweep.py [rounds] > gpurun_out/eff_speculation_sweep.txt"""
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

CELLS = [(-1, -1), (4, 3), (3, 3), (2, 2), (1, 1), (0, 0), (3, 2), (4, 2), (6, 3), (6, 4), (8, 4)]


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    contexts = os.environ.get("SWEEP_C", "4")
    base = None
    try:
        st = os.statvfs("/dev/shm")
        if st.f_bavail * st.f_frsize > 8 << 30:
            base = "/dev/shm"
    except OSError:
        pass
    d = tempfile.mkdtemp(prefix="curvis_spec_", dir=base)
    sky = (os.path.join(d, "pos.png"), os.path.join(d, "neg.png"))
    pngio.write_png(sky[0], skies.smooth(4096, 2048, 128)[..., :3], level=1)
    pngio.write_png(sky[1], skies.smooth(4096, 2048, 32)[..., :3], level=1)
    sim, vid, cam = os.path.join(d, "sim.toml"), os.path.join(d, "vid.toml"), os.path.join(d, "cam.toml")
    open(sim, "w").write("escape_radius = 100.0\nray_integration_max_itarations = 4096\nray_integration_step = 0.05\nsampling_initial_nums = 100\n"
                         "sampling_max_iterations = 50\nsampling_convergence_threshold_1 = 1e-5\nsampling_convergence_threshold_2 = 1e-5\n")
    open(cam, "w").write("resolution_x = 1920\nresolution_y = 1080\ndiagonal = 43.0\nfocal_length = 15.0\n")
    open(vid, "w").write('video_name = "v"\nframe_rate = 500.0\nfilepath_to_camera_path = "%s"\n' % refpaths.reference_path_file("path_orbit.csv"))
    print("# curvis video --mode efficient --contexts-per-device %s --batch 32, 1920x1080, 29 970 frames per cell, ONE MI355X; sampler speculation depth "
          "(sampling_speculation, sampling_speculation_first; -1 = the library's automatic choice)" % contexts)
    res = {c: [] for c in CELLS}
    for rnd in range(rounds):
        for (sp, fi) in CELLS:
            os.environ["CURVIS_CTX_OPTIONS"] = "sampling_speculation=%d,sampling_speculation_first=%d" % (sp, fi)
            s = V.run(d, "s%d_f%d_r%d" % (sp + 1, fi + 1, rnd), sky, vid, cam, sim,
                      ["--gpu-png", "on", "--writers", "16", "--contexts-per-device", contexts, "--batch", "32"], None, mode="efficient")
            if not s:
                continue
            dv = s["devices"]
            res[(sp, fi)].append(s["frames_per_s"])
            print("round %d  speculation %2d first %2d %8.1f frames/s | render kernels %.3f ms/frame | render(+deflate) call %.3f ms/frame" % (
                rnd, sp, fi, s["frames_per_s"], np.mean([x["kernel_ms_per_frame"] for x in dv]), np.mean([x["render_call_ms_per_frame"] for x in dv])), flush=True)
    os.environ.pop("CURVIS_CTX_OPTIONS", None)
    print("\n| speculation depth | below the first grid | frames/s (median; runs) |\n|---|---|---|")
    for c in CELLS:
        if res[c]:
            print("| %s | %s | %.0f (%s) |" % ("automatic" if c[0] < 0 else c[0], "automatic" if c[1] < 0 else c[1], np.median(res[c]), ", ".join("%.0f" % v for v in res[c])))
    subprocess.run(["rm", "-rf", d])


if __name__ == "__main__":
    main()
