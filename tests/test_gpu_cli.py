"""End-to-end: the `curvis` binary on a real GPU against the oracle -- file-level drop-in check
(PNG skies in, PNG frames out, reference naming, reference quirks)."""
import ctypes as C
import json
import os
import subprocess

import numpy as np
import pytest

import common
import oracle_lib as O
import refpaths
from curvis_amd import paths, pngio, rendering

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "curvis_amd", "bin", "curvis")

SIM = ("escape_radius = 100.0\nray_integration_max_itarations = 4096\nray_integration_step = 0.05\n"
       "sampling_initial_nums = 100\nsampling_max_iterations = 50\n"
       "sampling_convergence_threshold_1 = 1e-5\nsampling_convergence_threshold_2 = 2e-5\n")


def run(*args, **kw):
    return subprocess.run([BIN] + [str(a) for a in args], capture_output=True, text=True, timeout=600, **kw)


@pytest.fixture(scope="module")
def scene_files(tmp_path_factory):
    d = tmp_path_factory.mktemp("cli")
    sp, sn = common.make_skies(512, 256, "check")
    pngio.write_png(d / "pos.png", sp[..., :3])   # RGB8 -> alpha 255, as DynamicImage::get_pixel
    pngio.write_png(d / "neg.png", sn)            # RGBA8
    (d / "sim.toml").write_text(SIM)
    (d / "cam.toml").write_text("resolution_x = 96\nresolution_y = 54\ndiagonal = 43.0\nfocal_length = 15.0\n")
    (d / "met.toml").write_text("m = 0.1\na = 0.0001\nrho = 1.0\n")
    return d, sp, sn


def test_image_default_pose_efficient_and_brute(scene_files):
    d, sp, sn = scene_files
    out = d / "out_img"
    out.mkdir()
    r = run("image", d / "pos.png", d / "neg.png", out, "-s", d / "sim.toml", "-c", d / "cam.toml", "--stats", out / "st.json")
    assert r.returncode == 0, r.stderr
    assert "Image rendering" in r.stdout
    got = pngio.read_png(out / "output_image.png")
    om, oc, _, _ = common.scene("ellis", res=(96, 54))
    # the reference wires max_iterations_sampling to sampling_initial_nums (100), thresholds 1e-5 / 2e-5
    want, smp, _ = O.render_image_efficient(O.CV, om, oc, O.sky(sp), O.sky(sn), 4096, 100.0, 0.05, 100, 100, 1e-5, 2e-5)
    assert np.array_equal(got, want)
    st = json.loads((out / "st.json").read_text())
    assert st["steps"] == smp["steps"] and st["mode"] == "efficient"
    # per-pixel integrator through the same binary, Interstellar metric file
    r = run("image", d / "pos.png", d / "neg.png", out, "-s", d / "sim.toml", "-c", d / "cam.toml", "-m", d / "met.toml",
            "--mode", "brute")
    assert r.returncode == 0, r.stderr
    got = pngio.read_png(out / "output_image.png")
    om, oc, _, _ = common.scene("interstellar", res=(96, 54))
    want, _, _ = O.render_image(O.CV, om, oc, O.sky(sp), O.sky(sn), 4096, 100.0, 0.05)
    assert np.array_equal(got, want)
    # "direct" mode (an extension): compute_escape_angle for every pixel, no sampling / interpolation
    r = run("image", d / "pos.png", d / "neg.png", out, "-s", d / "sim.toml", "-c", d / "cam.toml", "-m", d / "met.toml",
            "--mode", "direct", "--stats", out / "st_direct.json")
    assert r.returncode == 0, r.stderr
    want, st_d = O.render_image_direct(O.CV, om, oc, O.sky(sp), O.sky(sn), 4096, 100.0, 0.05)
    assert np.array_equal(pngio.read_png(out / "output_image.png"), want)
    st = json.loads((out / "st_direct.json").read_text())
    assert st["steps"] == st_d.steps and st["mode"] == "direct"


def test_backgrounds_written_with_every_png_filter_give_the_same_image(scene_files, tmp_path):
    """the backgrounds' decoder (own inflater, rows reconstructed in place by a stage that trails it, a third writing the RGBA
    image) inside the whole run: skies whose rows cycle through Sub / Up / Average / Paeth / None, large enough for many
    hand-overs between the stages (2048 x 1024: 6-8 MiB of scanlines), must give the file the filter-None skies give"""
    from test_cli_host import _write_png_filtered
    d, _, _ = scene_files
    sp, sn = common.make_skies(2048, 1024, "check")
    rng = np.random.default_rng(12)
    sp = sp.copy()
    sp[..., :3] = np.clip(sp[..., :3].astype(np.int16) + rng.integers(-9, 10, size=sp[..., :3].shape), 0, 255).astype(np.uint8)
    outs = []
    for tag, filters in (("none", [0]), ("mixed", [4, 1, 2, 3, 4, 4, 0, 3]), ("paeth", [4])):
        a, b = tmp_path / ("pos_%s.png" % tag), tmp_path / ("neg_%s.png" % tag)
        _write_png_filtered(a, sp[..., :3], filters)   # RGB
        _write_png_filtered(b, sn, filters)            # RGBA
        out = tmp_path / ("out_" + tag)
        out.mkdir()
        r = run("image", a, b, out, "-s", d / "sim.toml", "-c", d / "cam.toml")
        assert r.returncode == 0, r.stderr
        outs.append(pngio.read_png(out / "output_image.png"))
    assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2])
    om, oc, _, _ = common.scene("ellis", res=(96, 54))
    want, _, _ = O.render_image_efficient(O.CV, om, oc, O.sky(sp), O.sky(sn), 4096, 100.0, 0.05, 100, 100, 1e-5, 2e-5)
    assert np.array_equal(outs[0], want)


def test_image_rows_split_over_devices(scene_files):
    """image --mode brute --devices 3: three contexts render row bands of the one frame (one band per GPU when the box
    has three; all on GPU 0 through the CURVIS_TEST_SHARE_DEVICE hook only when it has fewer); file and statistics equal
    the single-device run."""
    d, sp, sn = scene_files
    outs = []
    for n in (1, 3):
        out = d / ("out_rows%d" % n)
        out.mkdir()
        r = run("image", d / "pos.png", d / "neg.png", out, "-s", d / "sim.toml", "-c", d / "cam.toml", "--mode", "brute",
                "--devices", n, "--stats", out / "st.json", env=common.share_env(n))
        assert r.returncode == 0, r.stderr
        st = json.loads((out / "st.json").read_text())
        outs.append((pngio.read_png(out / "output_image.png"), st["rays"], st["steps"], st["n_pos"], st["n_neg"], st["n_none"]))
    assert np.array_equal(outs[0][0], outs[1][0]) and outs[0][1:] == outs[1][1:]
    om, oc, _, _ = common.scene("ellis", res=(96, 54))
    want, _, _ = O.render_image(O.CV, om, oc, O.sky(sp), O.sky(sn), 4096, 100.0, 0.05)
    assert np.array_equal(outs[1][0], want)


def test_video_orbit_frames_and_quirks(scene_files):
    d, sp, sn = scene_files
    out = d / "out_vid"
    out.mkdir()
    (out / "tmp").mkdir()
    (out / "tmp" / "stale.png").write_bytes(b"x")  # the tmp folder is deleted and recreated (src/rendering.rs:276-287)
    orbit = refpaths.reference_path_file("path_orbit.csv")
    (d / "vid.toml").write_text('video_name = "v"\nframe_rate = 0.25\nfilepath_to_camera_path = "%s"\n' % orbit)
    r = run("video", d / "pos.png", d / "neg.png", out, "-v", d / "vid.toml", "-s", d / "sim.toml", "-c", d / "cam.toml",
            "--batch", "4", "--contexts-per-device", "4", "--stats", out / "st.jsonl")
    assert r.returncode == 0, r.stderr
    assert not (out / "tmp" / "stale.png").exists()
    it = rendering.Interpolator.from_file(orbit)
    times = rendering.times_of_frames(it.min_time(), it.max_time(), 0.25)
    assert len(times) == 15 and "Rendering 15 frames..." in r.stdout
    frames = sorted(os.listdir(out / "tmp"), key=lambda s: int(s[6:-4]))
    assert frames == ["frame_%d.png" % k for k in range(15)]
    om = O.ellis(1.0)
    for k in (0, 7, 14):
        oc = O.camera(tuple(it.camera_position(times[k])), tuple(it.camera_forward(times[k])), tuple(it.camera_up(times[k])),
                      15.0, 43.0, (96, 54))
        # video path: threshold_1 is passed twice (src/rendering.rs:305-306)
        want, _, _ = O.render_image_efficient(O.CV, om, oc, O.sky(sp), O.sky(sn), 4096, 100.0, 0.05, 100, 100, 1e-5, 1e-5)
        assert np.array_equal(pngio.read_png(out / "tmp" / ("frame_%d.png" % k)), want), k
    lines = [json.loads(ln) for ln in (out / "st.jsonl").read_text().splitlines()]
    assert sorted(ln["frame"] for ln in lines) == list(range(15))
    assert all(ln["batch_call_ms"] >= ln["batch_kernel_ms"] > 0 for ln in lines)
    # --stats also writes the host profile: per-device table (PCI identity, frames, kernel / render-call time) and the
    # writer threads' stage times
    summ = json.loads((out / "st.jsonl.summary.json").read_text())
    # --mode efficient runs several contexts (host threads) per GPU -- up to six for long videos: more than half of its render call is host work and
    # its kernels do not fill the GPU (profiles/round5_eff_contexts_sweep.txt).  Asked for explicitly here: the automatic choice
    # gives a video this short ONE context (each costs a start-up of its own; checked at the end of this test)
    assert summ["frames"] == 15 and summ["png_level"] == -1 and len(summ["devices"]) == 4
    devs = summ["devices"]
    assert {dv["device"] for dv in devs} == {0} and len({dv["pci_bus_id"] for dv in devs}) == 1 and len(devs[0]["pci_bus_id"].split(":")) == 3
    assert sum(dv["frames"] for dv in devs) == 15 and all(dv["kernel_ms_per_frame"] > 0 for dv in devs)
    # the default writer: PNG front end on the device (filter, Huffman coding, Adler-32 in HIP kernels; the frames checked above
    # against the oracle were decoded from ITS streams), a writer thread only wraps the stream and adds the CRC
    assert summ["gpu_png"] is True and sum(dv["gpu_png_frames"] for dv in devs) == 15 and all(dv["gpu_png_fallback_frames"] == 0 for dv in devs)
    assert all(dv["gpu_png_kernel_ms_per_frame"] > 0 for dv in devs) and summ["encode"]["filter_ms"] == 0 and summ["encode"]["deflate_ms"] == 0
    enc = summ["encode"]
    assert enc["frames"] == 15 and enc["thread_ms_per_frame"] > 0 and enc["file_mb_per_frame"] < enc["raw_mb_per_frame"]
    assert "pci_bus_id" in r.stdout and "writer thread" in r.stdout
    # the frames are written by the fast PNG writer; zlib (--png-level 6) and a single slow writer thread that is kept
    # busy long after the last render (--encode-bench) give the same pixels
    out2 = d / "out_vid_z"
    out2.mkdir()
    r2 = run("video", d / "pos.png", d / "neg.png", out2, "-v", d / "vid.toml", "-s", d / "sim.toml", "-c", d / "cam.toml",
             "--batch", "4", "--png-level", "6", "--writers", "1", "--encode-bench", "3")
    assert r2.returncode == 0, r2.stderr
    for k in range(15):
        a = pngio.read_png(out / "tmp" / ("frame_%d.png" % k))
        assert np.array_equal(a, pngio.read_png(out2 / "tmp" / ("frame_%d.png" % k))), k
    r3 = run("video", d / "pos.png", d / "neg.png", out2, "-v", d / "vid.toml", "-s", d / "sim.toml", "-c", d / "cam.toml",
             "--batch", "2", "--writers", "1", "--encode-bench", "5", "--stats", out2 / "st3.jsonl")   # HOST fast writer, pool starved of writers: back-pressure, no crash
    assert r3.returncode == 0, r3.stderr
    s3 = json.loads((out2 / "st3.jsonl.summary.json").read_text())
    assert s3["gpu_png"] is False and s3["encode"]["deflate_ms"] > 0          # --encode-bench measures the host encoder
    for k in range(15):   # host fast writer == device front end == zlib, pixel for pixel
        assert np.array_equal(pngio.read_png(out2 / "tmp" / ("frame_%d.png" % k)), pngio.read_png(out / "tmp" / ("frame_%d.png" % k))), k
    # frames whose streams do not fit the batch buffer (here: a test hook shrinks it) are fetched as pixels and go to the host encoder
    r5 = run("video", d / "pos.png", d / "neg.png", out2, "-v", d / "vid.toml", "-s", d / "sim.toml", "-c", d / "cam.toml", "--stats", out2 / "st5.jsonl",
             env=dict(os.environ, CURVIS_TEST_SMALL_PNG_BUFFER="1"))
    assert r5.returncode == 0, r5.stderr
    s5 = json.loads((out2 / "st5.jsonl.summary.json").read_text())
    assert sum(dv["gpu_png_fallback_frames"] for dv in s5["devices"]) == 15 and sum(dv["gpu_png_frames"] for dv in s5["devices"]) == 0
    assert s5["encode"]["deflate_ms"] > 0 and len(s5["devices"]) == 1        # automatic: a video this short gets ONE context
    for k in (0, 7, 14):
        assert np.array_equal(pngio.read_png(out2 / "tmp" / ("frame_%d.png" % k)), pngio.read_png(out / "tmp" / ("frame_%d.png" % k))), k
    # the batch buffers are sized for zlib streams, not pixels (an eighth of the raw size; pinning memory costs time at both
    # ends of a run); streams that need more make the pool hand out larger buffers and are produced once more -- here from a
    # start of 300 bytes (test hook): every frame still comes from the device front end, none falls back to the host encoder
    r6 = run("video", d / "pos.png", d / "neg.png", out2, "-v", d / "vid.toml", "-s", d / "sim.toml", "-c", d / "cam.toml", "--batch", "4",
             "--stats", out2 / "st6.jsonl", env=dict(os.environ, CURVIS_TEST_STREAM_POOL_BYTES="300"))
    assert r6.returncode == 0, r6.stderr
    s6 = json.loads((out2 / "st6.jsonl.summary.json").read_text())
    assert sum(dv["gpu_png_frames"] for dv in s6["devices"]) == 15 and sum(dv["gpu_png_fallback_frames"] for dv in s6["devices"]) == 0
    assert sum(dv["gpu_png_buffer_regrown"] for dv in s6["devices"]) >= 1
    for k in range(15):
        assert np.array_equal(pngio.read_png(out2 / "tmp" / ("frame_%d.png" % k)), pngio.read_png(out / "tmp" / ("frame_%d.png" % k))), k
    r4 = run("video", d / "pos.png", d / "neg.png", out2, "-v", d / "vid.toml", "-s", d / "sim.toml", "-c", d / "cam.toml", "--gpu-png", "off")
    assert r4.returncode == 0, r4.stderr
    assert np.array_equal(pngio.read_png(out2 / "tmp" / "frame_14.png"), pngio.read_png(out / "tmp" / "frame_14.png"))


def test_video_off_by_one_panics_like_the_reference(scene_files):
    """frame_rate 30 on path_orbit.csv: the reference panics at frame index 1799 (README.md:107); with a
    coarser resolution of the same tail (frame times crafted to end inside the last CSV segment) the binary
    renders the frames before it and exits with status 101."""
    d, sp, sn = scene_files
    out = d / "out_vid2"
    out.mkdir()
    pos, fwd, up = paths.load_path(refpaths.reference_path_file("path_orbit.csv"))
    short = d / "short.csv"
    rows = [paths.HEADER] + [",".join(repr(float(x)) for x in list(pos[i]) + list(fwd[i]) + list(up[i])) for i in range(4)]
    short.write_text("\n".join(rows))
    # times: 0, 0.05, 0.10, 0.15 < max_time (0.18018); 0.15 lies in the last segment -> index out of bounds
    (d / "vid2.toml").write_text('video_name = "v"\nframe_rate = 20.0\nfilepath_to_camera_path = "%s"\n' % short)
    r = run("video", d / "pos.png", d / "neg.png", out, "-v", d / "vid2.toml", "-s", d / "sim.toml", "-c", d / "cam.toml")
    assert r.returncode == 101, (r.returncode, r.stderr)
    assert "index out of bounds" in r.stderr
    assert sorted(os.listdir(out / "tmp")) == ["frame_0.png", "frame_1.png", "frame_2.png"]


def test_video_rccl_sky_broadcast_path(scene_files):
    """--sky-broadcast rccl (the multi-GPU default) forced on the single GPU: rank 0 uploads, the textures
    go through ncclBroadcast (curvis_ctx_bcast_skies); frames must equal those of the upload path."""
    d, sp, sn = scene_files
    orbit = refpaths.reference_path_file("path_orbit.csv")
    (d / "vid3.toml").write_text('video_name = "v"\nframe_rate = 0.1\nfilepath_to_camera_path = "%s"\n' % orbit)
    outs = []
    for tag, env_extra, flag in (("rccl", {"CURVIS_FORCE_RCCL": "1"}, "rccl"), ("upload", {}, "upload")):
        out = d / ("out_vid_" + tag)
        out.mkdir()
        r = run("video", d / "pos.png", d / "neg.png", out, "-v", d / "vid3.toml", "-s", d / "sim.toml", "-c", d / "cam.toml",
                "--sky-broadcast", flag, env=dict(os.environ, **env_extra))
        assert r.returncode == 0, r.stderr
        outs.append(out)
    names = sorted(os.listdir(outs[0] / "tmp"))
    assert names == sorted(os.listdir(outs[1] / "tmp")) and len(names) == 6
    # every rank checks the broadcast textures against the decoded files; a flipped bit stops the run
    bad = d / "out_vid_bad"
    bad.mkdir()
    r = run("video", d / "pos.png", d / "neg.png", bad, "-v", d / "vid3.toml", "-s", d / "sim.toml", "-c", d / "cam.toml",
            "--sky-broadcast", "rccl", env=dict(os.environ, CURVIS_FORCE_RCCL="1", CURVIS_TEST_CORRUPT_BCAST="1"))
    assert r.returncode != 0 and "arrived corrupted" in r.stderr, (r.returncode, r.stderr)
    for n in names:
        assert np.array_equal(pngio.read_png(outs[0] / "tmp" / n), pngio.read_png(outs[1] / "tmp" / n))


def test_image_with_jpeg_skies(scene_files):
    """`.jpg` backgrounds (what the reference's README suggests): the binary decodes them with its own decoder
    (jpeg_io.h); the frame equals the oracle's render over the very texels that decoder produced -- the render path
    is format-agnostic, and JPEG decoding itself is outside the pixel-parity claims (zune-jpeg cannot be run here)."""
    PIL = pytest.importorskip("PIL.Image")
    import struct
    d, sp, sn = scene_files
    out = d / "out_jpg"
    out.mkdir()
    PIL.fromarray(sp[..., :3]).save(d / "pos.jpg", quality=92, subsampling=2)
    PIL.fromarray(sn[..., :3]).save(d / "neg.jpg", quality=92, progressive=True)
    skies = []
    for name in ("pos.jpg", "neg.jpg"):
        r = run("selftest-png", d / name, d / "dump.rgba")
        assert r.returncode == 0, r.stderr
        raw = (d / "dump.rgba").read_bytes()
        w, h = struct.unpack("<II", raw[:8])
        skies.append(np.frombuffer(raw[8:], np.uint8).reshape(h, w, 4).copy())
    r = run("image", d / "pos.jpg", d / "neg.jpg", out, "-s", d / "sim.toml", "-c", d / "cam.toml", "--mode", "brute")
    assert r.returncode == 0, r.stderr
    om, oc, _, _ = common.scene("ellis", res=(96, 54))
    want, _, _ = O.render_image(O.CV, om, oc, O.sky(skies[0]), O.sky(skies[1]), 4096, 100.0, 0.05)
    assert np.array_equal(pngio.read_png(out / "output_image.png"), want)


def test_python_image_rendering_system_equals_the_binary(scene_files):
    """curvis_amd.rendering.ImageRenderingSystem (mirror of src/rendering.rs:16-117) over the same files as
    `curvis image`: same PNG, name extension replaced like PathBuf::with_extension"""
    import curvis_amd
    d, sp, sn = scene_files
    out_bin, out_py = d / "out_irs_bin", d / "out_irs_py"
    out_bin.mkdir()
    (d / "img.toml").write_text('image_name = "shot.final"\nt = 0.0\nl = 3.0\ntheta = 1.3\nphi = 0.7\nforward_x = -1.0\nforward_y = 0.2\n'
                                'forward_z = 0.1\nup_x = 0.0\nup_y = 0.0\nup_z = 1.0\n')
    r = run("image", d / "pos.png", d / "neg.png", out_bin, "-i", d / "img.toml", "-s", d / "sim.toml", "-c", d / "cam.toml")
    assert r.returncode == 0, r.stderr
    assert sorted(os.listdir(out_bin)) == ["shot.png"]            # with_extension("png") replaces ".final"
    st = rendering.ImageRenderingSettings(d / "pos.png", d / "neg.png", out_py, "shot.final", (0.0, 3.0, 1.3, 0.7), (-1.0, 0.2, 0.1),
                                          (0.0, 0.0, 1.0), 15.0, 43.0, 96, 54, 100.0, 4096, 0.05, 100, 100, 1e-5, 2e-5)
    path = rendering.ImageRenderingSystem(curvis_amd.EllisMetric(1.0), st).render()   # creates the folder like the reference
    assert os.path.basename(path) == "shot.png"
    assert np.array_equal(pngio.read_png(path), pngio.read_png(out_bin / "shot.png"))
    om = O.ellis(1.0)
    oc = O.camera((0.0, 3.0, 1.3, 0.7), (-1.0, 0.2, 0.1), (0.0, 0.0, 1.0), 15.0, 43.0, (96, 54))
    want, _, _ = O.render_image_efficient(O.CV, om, oc, O.sky(sp), O.sky(sn), 4096, 100.0, 0.05, 100, 100, 1e-5, 2e-5)
    assert np.array_equal(pngio.read_png(path), want)


@pytest.mark.parametrize("mode", ["efficient", "brute"])
def test_fast_exit_drops_nothing_when_stdout_and_stats_are_pipes(scene_files, mode):
    """main() leaves through std::_Exit after flushing (host/curvis_cli.cpp: skips the HIP runtime's exit handlers).  _Exit runs
    no atexit handler and flushes no stdio buffer, so everything must already be out: stdout is a PIPE here (fully buffered, the
    case in which an unflushed tail would vanish), --stats is a FIFO read by this test, and the PNG frames are re-read.  Every
    byte that the ordinary exit (CURVIS_SLOW_EXIT=1, through the exit handlers) delivers must arrive."""
    import threading
    d, sp, sn = scene_files
    orbit = refpaths.reference_path_file("path_orbit.csv")
    got = {}
    for how in ("fast", "slow"):
        out = d / ("out_exit_%s_%s" % (mode, how))
        out.mkdir()
        (d / "vid_exit.toml").write_text('video_name = "v"\nframe_rate = 1.0\nfilepath_to_camera_path = "%s"\n' % orbit)
        fifo = out / "st.fifo"
        os.mkfifo(fifo)
        records = []

        def reader():
            with open(fifo) as f:       # blocks until the binary opens its end; reads to EOF = the binary's fclose
                records.extend(f.read().splitlines())
        t = threading.Thread(target=reader)
        t.start()
        env = dict(os.environ)
        env.pop("CURVIS_SLOW_EXIT", None)
        if how == "slow":
            env["CURVIS_SLOW_EXIT"] = "1"
        r = run("video", d / "pos.png", d / "neg.png", out, "-v", d / "vid_exit.toml", "-s", d / "sim.toml", "-c", d / "cam.toml",
                "--mode", mode, "--stats", fifo, env=env)
        t.join(timeout=60)
        assert not t.is_alive() and r.returncode == 0, r.stderr[-2000:]
        recs = sorted((json.loads(ln) for ln in records), key=lambda x: x["frame"])
        assert [x["frame"] for x in recs] == list(range(60)), (how, len(recs))
        summ = json.loads((out / "st.fifo.summary.json").read_text())
        assert summ["frames"] == 60
        lines = r.stdout.splitlines()
        assert "Rendering 60 frames..." in r.stdout and lines[-1].startswith("video: 60 frames in "), lines[-3:]
        frames = [pngio.read_png(out / "tmp" / ("frame_%d.png" % k)) for k in range(60)]     # every file complete (CRC + Adler checked)
        got[how] = (len(lines), [(x["frame"], x["rays"], x["steps"], x["n_pos"], x["n_neg"], x["n_none"]) for x in recs], frames)
    assert got["fast"][0] == got["slow"][0] and got["fast"][1] == got["slow"][1]
    assert all(np.array_equal(a, b) for a, b in zip(got["fast"][2], got["slow"][2]))


@pytest.mark.parametrize("video", ["through", "orbit"])
def test_video_efficient_with_the_device_sampler_prefetched(scene_files, video):
    """`curvis video` in the reference's default mode at the operating point of a long video: 64 frames per call -> the device-resident
    sampler (sampler_kernel), the next call's sampler prefetched under the current call (curvis_ctx_prefetch_efficient), one and two
    worker contexts.  The fly-through (Interstellar: every frame its own camera radius, one sampler job per frame) and the orbit (all
    frames l = 3: the batches have EQUAL keys, the oldest matching prefetch must be the one taken).  Every 8th frame against the
    oracle's render_image_efficient; the binary's summary says that every call after a worker's first found its sampler ready."""
    d, sp, sn = scene_files
    csv = refpaths.reference_path_file("path_%s.csv" % video)
    it = rendering.Interpolator.from_file(csv)
    fps = 288 / (it.max_time() - it.min_time())
    times = rendering.times_of_frames(it.min_time(), it.max_time(), fps)
    n = len(times)
    assert 286 <= n <= 289
    om = O.interstellar(0.1, 1e-4, 1.0) if video == "through" else O.ellis(1.0)
    cap = 8192 if video == "through" else 4096
    (d / "vid_dev.toml").write_text('video_name = "v"\nframe_rate = %r\nfilepath_to_camera_path = "%s"\n' % (fps, csv))
    (d / "sim_dev.toml").write_text(SIM.replace("ray_integration_max_itarations = 4096", "ray_integration_max_itarations = %d" % cap))
    want = {}
    for contexts in (1, 2):
        out = d / ("out_dev_%s_%d" % (video, contexts))
        out.mkdir()
        args = ["video", d / "pos.png", d / "neg.png", out, "-v", d / "vid_dev.toml", "-s", d / "sim_dev.toml", "-c", d / "cam.toml",
                "--batch", "64", "--contexts-per-device", contexts, "--stats", out / "st.jsonl"]
        if video == "through":
            args += ["-m", d / "met.toml"]
        # (device_sampler = 1: the last, short batch of a worker would otherwise fall under the library's 48-frame threshold)
        r = run(*args, env=dict(os.environ, CURVIS_CTX_OPTIONS="device_sampler=1"))
        assert r.returncode in (0, 101), r.stderr[-2000:]       # 101: the reference's own off-by-one at the end of a path
        files = [f for f in os.listdir(out / "tmp") if f.endswith(".png")]
        assert len(files) >= n - 3
        for k in range(0, len(files), 8):
            if k not in want:
                oc = O.camera(tuple(it.camera_position(times[k])), tuple(it.camera_forward(times[k])), tuple(it.camera_up(times[k])), 15.0, 43.0, (96, 54))
                want[k] = O.render_image_efficient(O.CV, om, oc, O.sky(sp), O.sky(sn), cap, 100.0, 0.05, 100, 100, 1e-5, 1e-5)[0]
            assert np.array_equal(pngio.read_png(out / "tmp" / ("frame_%d.png" % k)), want[k]), (video, contexts, k)
        summ = json.loads((out / "st.jsonl.summary.json").read_text())
        devs = summ["devices"]
        assert len(devs) == contexts
        for dv in devs:       # a worker with b batches prefetches its first and every following one (b), and every call finds one ready
            assert dv["batches"] >= 2 and dv["sampler_prefetches"] == dv["batches"] and dv["sampler_prefetch_hits"] == dv["batches"], dv
