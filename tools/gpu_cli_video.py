"""Host side of `curvis video --mode brute` at multi-GPU frame rates, measured on ONE GPU box (run on the GPU box:
python tools/gpu_cli_video.py > gpurun_out/cli_video.txt; the summary goes to profiles/<round>_cli_video.txt).

The reference saves one PNG per frame (src/rendering.rs:291-316).  Eight MI355X render ~800 frames/s of BASELINE
configs[3] (1080p, 8 / 10 ms) and ~148 frames/s of configs[4] (4K, 8 / 54 ms); the host has to encode and write that.
With one GPU the host's capacity is measured by `--encode-bench K`: every rendered frame is encoded K more times by
the same writer pool into a scratch file, so the pool sees (K + 1) x the frame rate of one GPU.

Per stage, from `--stats FILE` (FILE.summary.json): render call (GPU kernels + D2H), hand-over to the writer pool,
per-frame writer-thread time split into filter / deflate / checksums / file write."""
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import refpaths  # noqa: E402  (the reference's own camera paths: tests/golden/paths)
from curvis_amd import paths, pngio, skies  # noqa: E402

BIN = os.path.join(ROOT, "curvis_amd", "bin", "curvis")


def cpu_quota():
    try:
        q = open("/sys/fs/cgroup/cpu.max").read().split()
        return None if q[0] == "max" else float(q[0]) / float(q[1])
    except (OSError, ValueError, IndexError):
        return None


def run(d, tag, sky, vid, cam, sim, extra, expect_frames, mode="brute"):
    out = os.path.join(d, "out_" + tag)
    os.mkdir(out)
    st = os.path.join(out, "st.jsonl")
    t0 = time.perf_counter()
    r = subprocess.run([BIN, "video", sky[0], sky[1], out, "-v", vid, "-c", cam, "-s", sim, "--mode", mode, "--stats", st] + extra,
                       capture_output=True, text=True)
    dt = time.perf_counter() - t0
    if r.returncode not in (0, 101):  # 101 = the reference's own panic in the last segment of a path (frames before it are written)
        print("%s: rc %d %s" % (tag, r.returncode, r.stderr[-400:]))
        return None
    n = len([f for f in os.listdir(os.path.join(out, "tmp")) if f.endswith(".png")])
    assert expect_frames is None or n == expect_frames, (n, expect_frames)
    s = json.load(open(st + ".summary.json"))
    s["process_wall_s"] = dt
    # per batch: what the render call costs beyond its kernels (uploads, counters, D2H of the frames); the first launch
    # of a launch shape also carries the relay kernel's one-off check against the static kernel
    seen = {}
    for ln in open(st):
        r = json.loads(ln)
        seen[(r["device"], r["batch_kernel_ms"], r["batch_call_ms"])] = ((r["batch_call_ms"] - r["batch_kernel_ms"]) / r["batch_frames"], r["frame"])
    over = sorted(v[0] for v in seen.values())
    s["call_overhead_ms_per_frame_median"] = over[len(over) // 2]
    s["call_overhead_ms_per_frame_max"] = over[-1]
    subprocess.run(["rm", "-rf", out])
    return s


def line(tag, s):
    dv, en = s["devices"][0], s["encode"]
    txt = ("%-34s %6.1f frames/s | GPU kernel %6.2f ms/frame, render call %6.2f (beyond the kernels: median %.2f ms/frame, first launch %.2f), buffer waits %.2f s | writer thread per frame: "
           "filter %.2f + deflate %.2f + checksum %.2f + write %.2f = %.2f ms (%.0f MB/s), %.2f -> %.2f MB | drain %.2f s") % (
        tag, s["frames_per_s"], dv["kernel_ms_per_frame"], dv["render_call_ms_per_frame"], s["call_overhead_ms_per_frame_median"],
        s["call_overhead_ms_per_frame_max"], dv["buffer_wait_s"],
        en["filter_ms"], en["deflate_ms"], en["checksum_ms"], en["write_ms"], en["thread_ms_per_frame"], en["mb_per_s_per_thread"],
        en["raw_mb_per_frame"], en["file_mb_per_frame"], s["writer_drain_s"])
    if "encode_bench" in s:
        eb = s["encode_bench"]
        total = (eb["frames"] + en["frames"]) / s["wall_s"]
        txt += " | pool encoded %d frames in all = %.0f frames/s = %.2f GB/s raw" % (eb["frames"] + en["frames"], total, total * en["raw_mb_per_frame"] / 1e3)
    print(txt, flush=True)


def main():
    d = tempfile.mkdtemp(prefix="curvis_video_")
    print("# `curvis video --mode brute` on one MI355X: host stages and writer-pool scaling")
    print("host: %d logical CPUs visible, cgroup CPU quota %s; output directory %s (%s)" % (
        os.cpu_count(), cpu_quota(), d, subprocess.run(["df", "-T", d], capture_output=True, text=True).stdout.splitlines()[-1].split()[1]))
    sky = (os.path.join(d, "pos.png"), os.path.join(d, "neg.png"))
    pngio.write_png(sky[0], skies.smooth(4096, 2048, 128)[..., :3], level=1)
    pngio.write_png(sky[1], skies.smooth(4096, 2048, 32)[..., :3], level=1)
    cks = (os.path.join(d, "cpos.png"), os.path.join(d, "cneg.png"))   # a sky with structure: hash-coloured 64-texel cells
    pngio.write_png(cks[0], skies.checker(4096, 2048, seed=0xC0FFEE)[..., :3], level=1)
    pngio.write_png(cks[1], skies.checker(4096, 2048, seed=0xBADC0DE)[..., :3], level=1)
    sim = os.path.join(d, "sim.toml")
    SIM = ("escape_radius = 100.0\nray_integration_max_itarations = %d\nray_integration_step = 0.05\nsampling_initial_nums = 100\n"
           "sampling_max_iterations = 50\nsampling_convergence_threshold_1 = 1e-5\nsampling_convergence_threshold_2 = 1e-5\n")
    # ---- configs[3]: orbit, 240 frames, 1080p, Ellis
    vid = os.path.join(d, "vid.toml")
    open(vid, "w").write('video_name = "v"\nframe_rate = 4.0\nfilepath_to_camera_path = "%s"\n' % refpaths.reference_path_file("path_orbit.csv"))
    cam = os.path.join(d, "cam.toml")
    open(cam, "w").write("resolution_x = 1920\nresolution_y = 1080\ndiagonal = 43.0\nfocal_length = 15.0\n")
    open(sim, "w").write(SIM % 4096)
    print("\n## configs[3]: path_orbit.csv @ 4 fps = 240 frames, 1920x1080, Ellis, cap 4096, --batch 8, smooth sky (6.22 MB raw per frame)")
    for w in (1, 2, 4, 8, 16, 32):
        s = run(d, "c3_w%d" % w, sky, vid, cam, sim, ["--batch", "8", "--writers", str(w)], 240)
        if s:
            line("fast writer, %2d writer threads" % w, s)
    for lvl in (1, 6):
        s = run(d, "c3_l%d" % lvl, sky, vid, cam, sim, ["--batch", "8", "--writers", "16", "--png-level", str(lvl)], 240)
        if s:
            line("zlib level %d, 16 writer threads" % lvl, s)
    s = run(d, "c3_ck", cks, vid, cam, sim, ["--batch", "8", "--writers", "16"], 240)
    if s:
        line("fast writer, 16 thr, checker sky", s)
    print("\n### the reference's default renderer (`--mode efficient`: ~0.4 ms of GPU per frame) -- here the HOST is the limit")
    for w in (4, 16, 32):
        s = run(d, "c3_eff_w%d" % w, sky, vid, cam, sim, ["--batch", "16", "--writers", str(w)], 240, mode="efficient")
        if s:
            line("efficient mode, %2d writer threads" % w, s)
    print("\n### encode capacity of this host (one GPU feeding it, every frame encoded 1 + K times)")
    for w, k in ((8, 63), (16, 95), (32, 95), (64, 95)):
        s = run(d, "c3_eb_w%d" % w, sky, vid, cam, sim, ["--batch", "8", "--writers", str(w), "--encode-bench", str(k)], 240)
        if s:
            line("fast writer, %2d threads, K = %d" % (w, k), s)
    s = run(d, "c3_eb_ck", cks, vid, cam, sim, ["--batch", "8", "--writers", "16", "--encode-bench", "95"], 240)
    if s:
        line("fast, 16 thr, K = 95, checker sky", s)
    s = run(d, "c3_eb_z1", sky, vid, cam, sim, ["--batch", "8", "--writers", "16", "--encode-bench", "31", "--png-level", "1"], 240)
    if s:
        line("zlib level 1, 16 thr, K = 31", s)
    # ---- configs[4]: a 24-frame shard of the 4K Interstellar video
    print("\n## configs[4], 24-frame shard: path_through.csv @ 24 fps (frames 0..23 of 480), 3840x2160, Interstellar, cap 8192 (24.9 MB raw per frame)")
    import numpy as np
    full = open(refpaths.reference_path_file("path_through.csv")).read().splitlines()
    # the first 50 rows of the path cover t = 0 .. 0.98 s: exactly 24 frames at 24 fps
    shard = os.path.join(d, "through_shard.csv")
    open(shard, "w").write("\n".join(full[:51]) + "\n")
    t_last = float(full[50].split(",")[0])
    n4 = len(np.arange(0.0, t_last, 1.0 / 24.0))
    open(vid, "w").write('video_name = "v"\nframe_rate = 24.0\nfilepath_to_camera_path = "%s"\n' % shard)
    open(cam, "w").write("resolution_x = 3840\nresolution_y = 2160\ndiagonal = 43.0\nfocal_length = 15.0\n")
    open(sim, "w").write(SIM % 8192)
    met = os.path.join(d, "met.toml")
    open(met, "w").write("m = 0.1\na = 0.0001\nrho = 1.0\n")
    # frames counted the way times_of_frames does (float accumulation)
    t, cnt = float(full[1].split(",")[0]), 0
    while t < t_last:
        cnt += 1
        t += 1.0 / 24.0
    for w, k in ((16, 0), (16, 63), (32, 63)):
        s = run(d, "c4_w%d_k%d" % (w, k), sky, vid, cam, sim, ["-m", met, "--batch", "4", "--writers", str(w)] + (["--encode-bench", str(k)] if k else []), None)
        if s:
            line("fast writer, %2d threads, K = %d" % (w, k), s)
    subprocess.run(["rm", "-rf", d])


if __name__ == "__main__":
    main()
