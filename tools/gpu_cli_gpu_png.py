"""`curvis video` with the PNG front end on the device against the host writer, on ONE GPU box (run there:
python tools/gpu_cli_gpu_png.py > gpurun_out/cli_gpu_png.txt; summary -> profiles/round4_cli_gpu_png.txt).

The reference's default renderer (`--mode efficient`) costs ~0.3 ms of GPU per 1080p frame: there the host's PNG encoding
(5-7 ms of a writer thread per frame) was the limit of the whole program (profiles/round3_cli_video.txt: 584 frames/s with 16
writer threads).  Long runs (path_orbit.csv at 40 fps = 2400 frames) so that start-up (decoding the two sky files, context
creation) does not dominate."""
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import refpaths  # noqa: E402  (the reference's own camera paths: tests/golden/paths)
from curvis_amd import paths, pngio, skies  # noqa: E402
import gpu_cli_video as V  # noqa: E402


def line(tag, s):
    dv, en = s["devices"][0], s["encode"]
    print(("%-44s %7.1f frames/s (wall %.2f s, %d frames) | GPU render kernels %.3f ms/frame + PNG kernels %.3f | render(+deflate) call %.3f ms/frame | "
           "writer thread per frame: filter %.2f + deflate %.2f + checksum %.2f + write %.2f = %.2f ms | %.2f -> %.3f MB | waits: buffers %.2f s, drain %.2f s") % (
        tag, s["frames_per_s"], s["wall_s"], s["frames"], dv["kernel_ms_per_frame"], dv.get("gpu_png_kernel_ms_per_frame", 0.0), dv["render_call_ms_per_frame"],
        en["filter_ms"], en["deflate_ms"], en["checksum_ms"], en["write_ms"], en["thread_ms_per_frame"], en["raw_mb_per_frame"], en["file_mb_per_frame"],
        dv["buffer_wait_s"], s["writer_drain_s"]), flush=True)


def main():
    d = tempfile.mkdtemp(prefix="curvis_gpupng_")
    print("# `curvis video`: PNG front end on the device (--gpu-png on, the default with the fast writer) vs host writer (--gpu-png off)")
    print("host: %d logical CPUs visible, cgroup CPU quota %s; output directory %s (%s)" % (
        os.cpu_count(), V.cpu_quota(), d, subprocess.run(["df", "-T", d], capture_output=True, text=True).stdout.splitlines()[-1].split()[1]))
    sky = (os.path.join(d, "pos.png"), os.path.join(d, "neg.png"))
    pngio.write_png(sky[0], skies.smooth(4096, 2048, 128)[..., :3], level=1)
    pngio.write_png(sky[1], skies.smooth(4096, 2048, 32)[..., :3], level=1)
    cks = (os.path.join(d, "cpos.png"), os.path.join(d, "cneg.png"))
    pngio.write_png(cks[0], skies.checker(4096, 2048, seed=0xC0FFEE)[..., :3], level=1)
    pngio.write_png(cks[1], skies.checker(4096, 2048, seed=0xBADC0DE)[..., :3], level=1)
    sim, vid, cam = os.path.join(d, "sim.toml"), os.path.join(d, "vid.toml"), os.path.join(d, "cam.toml")
    open(sim, "w").write("escape_radius = 100.0\nray_integration_max_itarations = 4096\nray_integration_step = 0.05\nsampling_initial_nums = 100\n"
                         "sampling_max_iterations = 50\nsampling_convergence_threshold_1 = 1e-5\nsampling_convergence_threshold_2 = 1e-5\n")
    open(cam, "w").write("resolution_x = 1920\nresolution_y = 1080\ndiagonal = 43.0\nfocal_length = 15.0\n")
    open(vid, "w").write('video_name = "v"\nframe_rate = 40.0\nfilepath_to_camera_path = "%s"\n' % refpaths.reference_path_file("path_orbit.csv"))
    print("\n## --mode efficient (the reference's renderer), ONE context on the GPU (--contexts-per-device 1), path_orbit.csv @ 40 fps = 2398 frames before the reference's own off-by-one panic, 1920x1080, Ellis")
    for tag, s_, extra in (("host writer, 16 threads", sky, ["--gpu-png", "off", "--writers", "16"]),
                           ("device front end, 16 threads", sky, ["--gpu-png", "on", "--writers", "16"]),
                           ("device front end, 4 threads", sky, ["--gpu-png", "on", "--writers", "4"]),
                           ("device front end, 16 thr, --batch 32", sky, ["--gpu-png", "on", "--writers", "16", "--batch", "32"]),
                           ("host writer, 16 thr, checker sky", cks, ["--gpu-png", "off", "--writers", "16"]),
                           ("device front end, 16 thr, checker sky", cks, ["--gpu-png", "on", "--writers", "16"])):
        b = ([] if "--batch" in extra else ["--batch", "16"]) + ["--contexts-per-device", "1"]
        s = V.run(d, "eff_%d" % abs(hash(tag)), s_, vid, cam, sim, b + extra, None, mode="efficient")
        if s:
            line(tag, s)
    print("\n### contexts per device (`--contexts-per-device C`: C host threads with a context each on the ONE GPU; the adaptive sampler's host "
          "side of one overlaps the kernels of another)")
    for cpd in (2, 3, 4):  # 2 is the default in this mode
        for tag, extra in (("device front end, C = %d" % cpd, ["--gpu-png", "on"]), ("host writer, C = %d" % cpd, ["--gpu-png", "off"])):
            s = V.run(d, "cpd_%d_%d" % (cpd, abs(hash(tag))), sky, vid, cam, sim, ["--batch", "16", "--writers", "16", "--contexts-per-device", str(cpd)] + extra, None, mode="efficient")
            if s:
                line(tag, s)
    print("\n### capacity of the host's 16 writer threads in that mode (one GPU feeding them, every frame saved 1 + K times, `--encode-bench K`)")
    for tag, extra in (("host writer, K = 3", ["--gpu-png", "off", "--encode-bench", "3"]), ("device front end, K = 3", ["--gpu-png", "on", "--encode-bench", "3"]),
                       ("device front end, K = 15", ["--gpu-png", "on", "--encode-bench", "15"])):
        s = V.run(d, "cap_%d" % abs(hash(tag)), sky, vid, cam, sim, ["--batch", "16", "--writers", "16", "--contexts-per-device", "1"] + extra, None, mode="efficient")
        if s:
            line(tag, s)
            eb, en = s["encode_bench"], s["encode"]
            print("    -> the pool saved %d frames in %.2f s = %.0f frames/s (%.2f ms of a writer thread per extra save)" % (
                eb["frames"] + en["frames"], s["wall_s"], (eb["frames"] + en["frames"]) / s["wall_s"], eb["thread_ms_per_frame"]), flush=True)
    open(vid, "w").write('video_name = "v"\nframe_rate = 4.0\nfilepath_to_camera_path = "%s"\n' % refpaths.reference_path_file("path_orbit.csv"))
    print("\n## --mode brute (BASELINE configs[3]: 240 frames): the GPU is the limit either way; the host's share shrinks")
    for tag, extra in (("host writer, 16 threads", ["--gpu-png", "off"]), ("device front end, 16 threads", ["--gpu-png", "on"])):
        s = V.run(d, "br_%d" % abs(hash(tag)), sky, vid, cam, sim, ["--batch", "8", "--writers", "16"] + extra, 240)
        if s:
            line(tag, s)
    subprocess.run(["rm", "-rf", d])


if __name__ == "__main__":
    main()
