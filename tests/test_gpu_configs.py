"""BASELINE.json configs[3] and configs[4] under -m gpu: the two VIDEO configurations rendered with the per-pixel
integrator (RelativisticSystem::render_image, src/systems.rs:307-330) frame by frame as
VideoRenderingSystem::render does (src/rendering.rs:291-316), camera poses from the Interpolator
(src/interpolation.rs:63-91) on the reference's two camera paths:

  configs[3]  Ellis, path_orbit.csv   @ 4 fps  = 240 frames, 1920x1080, cap 4096
  configs[4]  Interstellar, path_through.csv @ 24 fps = 480 frames, 3840x2160, cap 8192, per-frame
              early-termination statistics

Every frame of both videos is rendered at a reduced resolution (what the oracle can follow in seconds) through
curvis_render_brute_batch and compared with the oracle frame by frame: pixels, and the PER-FRAME counters
(rays, executed Euler steps, escaped +l / -l, capped) that curvis_ctx_frame_stats returns for each frame of a
multi-frame launch.  Selected frames -- first / quarter / half / last of the orbit; the first frame and the two
frames nearest the throat (l = 0) of the fly-through -- are rendered at the FULL size of the config and compared
the same way.  The same two videos then go through the `curvis video --mode brute --stats` binary."""
import json
import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

import common
import oracle_lib as O
import refpaths
import curvis_amd
from curvis_amd import paths, pngio, rendering

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "curvis_amd", "bin", "curvis")
THREADS = common.host_threads(128)

VIDEOS = {
    # name: (metric, csv, fps, frames, small res, full res, cap)
    "orbit": ("ellis", "path_orbit.csv", 4.0, 240, (96, 54), (1920, 1080), 4096),
    "through": ("interstellar", "path_through.csv", 24.0, 480, (64, 36), (3840, 2160), 8192),
}


def video_poses(csv, fps):
    it = rendering.Interpolator.from_file(refpaths.reference_path_file(csv))
    times = rendering.times_of_frames(it.min_time(), it.max_time(), fps)
    return times, [(tuple(it.camera_position(t)), tuple(it.camera_forward(t)), tuple(it.camera_up(t))) for t in times]


def metrics_of(name):
    if name == "ellis":
        return O.ellis(1.0), curvis_amd.EllisMetric(1.0)
    return O.interstellar(0.1, 1e-4, 1.0), curvis_amd.InterstellarMetric(0.1, 1e-4, 1.0)


def oracle_frames(om, poses, res, sp, sn, cap):
    """oracle (cv flavour) render of every pose, frames spread over the host threads: [(rgb, Stats)]; memoised for the session --
    the small-frame test and the test through the binary need the same 240 / 480 frames"""
    import hashlib
    key = ("oracle_frames", om.kind, om.rho, om.m, om.a, tuple(poses), tuple(res), hashlib.sha1(sp.tobytes() + sn.tobytes()).hexdigest(), cap)
    return common.oracle_memo(key, lambda: _oracle_frames(om, poses, res, sp, sn, cap))


def _oracle_frames(om, poses, res, sp, sn, cap):
    osp, osn = O.sky(sp), O.sky(sn)

    def work(p):
        oc = O.camera(p[0], p[1], p[2], 15.0, 43.0, res)
        rgb, _, st = O.render_image(O.CV, om, oc, osp, osn, cap, 100.0, 0.05)
        return rgb, (st.rays, st.steps, st.n_pos, st.n_neg, st.n_none, st.n_oob)
    with ThreadPoolExecutor(THREADS) as ex:
        return list(ex.map(work, poses))


def stats_tuple(s):
    return (int(s.rays), int(s.steps), int(s.n_pos), int(s.n_neg), int(s.n_none), int(s.n_oob))


@pytest.mark.parametrize("video", ["orbit", "through"])
def test_video_config_every_frame_small(gpu_ctx, video):
    """all 240 / 480 frames at reduced resolution: pixels and per-frame counters of multi-frame launches"""
    metric, csv, fps, n_frames, res, _, cap = VIDEOS[video]
    times, poses = video_poses(csv, fps)
    assert len(times) == n_frames
    om, pm = metrics_of(metric)
    sp, sn = common.make_skies(512, 256, "check")
    want = oracle_frames(om, poses, res, sp, sn, cap)
    gpu_ctx.set_sky(0, curvis_amd.SphericalImage(sp))
    gpu_ctx.set_sky(1, curvis_amd.SphericalImage(sn))
    cams = [curvis_amd.Camera(p[0], p[1], p[2], 15.0, 43.0, res[0], res[1]) for p in poses]
    # launch shapes: 7 frames (relay kernel territory, 64 counter replicas), 48 (static kernel, 8 replicas), the rest
    k = 0
    for batch in (7, 48, 10 ** 6):
        part = cams[k:k + batch]
        if not part:
            break
        rgb, st = gpu_ctx.render_brute(pm, part, cap, 100.0, 0.05)
        per = gpu_ctx.frame_stats()
        assert len(per) == len(part)
        for j in range(len(part)):
            assert np.array_equal(rgb[j], want[k + j][0]), "frame %d pixels" % (k + j)
            assert stats_tuple(per[j]) == want[k + j][1], "frame %d counters %s vs %s" % (
                k + j, stats_tuple(per[j]), want[k + j][1])
        assert stats_tuple(st)[:5] == tuple(sum(w[1][i] for w in want[k:k + len(part)]) for i in range(5))
        k += len(part)
    assert k == n_frames
    # the persistent kernel (staged path: shade_kernel keeps the counters) on a few frames that straddle waves
    gpu_ctx.set_option("variant", 0)
    try:
        sel = [0, n_frames // 2, n_frames - 1]
        rgb, _ = gpu_ctx.render_brute(pm, [cams[i] for i in sel], cap, 100.0, 0.05)
        per = gpu_ctx.frame_stats()
        for j, i in enumerate(sel):
            assert np.array_equal(rgb[j], want[i][0]) and stats_tuple(per[j]) == want[i][1]
    finally:
        gpu_ctx.set_option("variant", -1)
    # the fly-through's frames are not alike: the per-frame numbers really differ from frame to frame.  (The orbit's
    # are all the SAME frame: render_image never rotates into the world frame, src/systems.rs:540-561, and the
    # metric is spherically symmetric, so phi of the camera does not enter -- 240 identical sets of counters.)
    if video == "through":
        assert len({w[1][1] for w in want}) > n_frames // 4
    else:
        assert len({w[1] for w in want}) == 1


def full_size_frames(video, n_frames, poses):
    if video == "orbit":
        return [0, 60, 120, 239]
    ls = np.array([abs(p[0][1]) for p in poses])
    near = sorted(np.argsort(ls)[:2].tolist())
    return [0] + near


def full_size_oracle_frame(video, i, threads=None):
    """oracle (cv flavour) render of frame i of a video config at FULL size on the 2048x1024 checker skies: (rgb, counters);
    memoised -- the background prefetcher (ORACLE_PREFETCH below, started by conftest) computes the fly-through's frames while
    the first test files run"""
    metric, csv, fps, n_frames, _, res, cap = VIDEOS[video]

    def make():
        _, poses = video_poses(csv, fps)
        om, _ = metrics_of(metric)
        sp, sn = common.make_skies(2048, 1024, "check")
        oc = O.camera(poses[i][0], poses[i][1], poses[i][2], 15.0, 43.0, res)
        return common.oracle_full_frame_stats(O.CV, om, oc, sp, sn, cap, threads=threads or THREADS)
    return common.oracle_memo(("full_size_oracle_frame", video, i), make)


def ORACLE_PREFETCH(selected):
    """jobs for common.oracle_prefetch: the three full-size fly-through frames (~60 s of 16 CPUs) when their test is selected"""
    if not any("test_video_config_full_size_frames[through]" in n for n in selected):
        return []
    _, poses = video_poses(VIDEOS["through"][1], VIDEOS["through"][2])
    T = common.prefetch_threads()
    return [lambda i=i: full_size_oracle_frame("through", i, threads=T) for i in full_size_frames("through", VIDEOS["through"][3], poses)]


@pytest.mark.parametrize("video", ["orbit", "through"])
def test_video_config_full_size_frames(gpu_ctx, video):
    """frames {0, 60, 120, 239} of the orbit at 1920x1080 cap 4096; frame 0 (l = -4) and the two frames nearest
    l = 0 of the fly-through at 3840x2160 cap 8192 (Interstellar): ONE launch per video, per-frame counters and
    every pixel against the oracle."""
    metric, csv, fps, n_frames, _, res, cap = VIDEOS[video]
    times, poses = video_poses(csv, fps)
    sel = full_size_frames(video, n_frames, poses)
    if video == "through":
        assert abs(poses[sel[1]][0][1]) < 0.02 and abs(poses[sel[2]][0][1]) < 0.02 and poses[0][0][1] == -4.0
    om, pm = metrics_of(metric)
    sp, sn = common.make_skies(2048, 1024, "check")
    gpu_ctx.set_sky(0, curvis_amd.SphericalImage(sp))
    gpu_ctx.set_sky(1, curvis_amd.SphericalImage(sn))
    cams = [curvis_amd.Camera(poses[i][0], poses[i][1], poses[i][2], 15.0, 43.0, res[0], res[1]) for i in sel]
    rgb, st = gpu_ctx.render_brute(pm, cams, cap, 100.0, 0.05)
    per = gpu_ctx.frame_stats()
    assert len(per) == len(sel)
    common.oracle_budget("%s: %d full-size frames, cv flavour" % (video, len(sel)), metric, O.CV, float(st.steps), threads=THREADS)
    for j, i in enumerate(sel):
        want_rgb, want = full_size_oracle_frame(video, i)
        assert want[0] == res[0] * res[1]
        assert stats_tuple(per[j]) == want, "frame %d: %s vs %s" % (i, stats_tuple(per[j]), want)
        assert np.array_equal(rgb[j], want_rgb), "frame %d pixels" % i
        print("%s frame %d (l = %.4f): %d steps, +l %d, -l %d, capped %d" % ((video, i, poses[i][0][1]) + want[1:5]))


def ray_classes(dbg, cap):
    """the ray classes whose parity is ill-conditioned: rays that hit the step cap, rays whose theta
    has left [0, pi] (they crossed a pole: 1/sin^2 exploded on the way), and rays that spent much longer than their
    frame's median near the throat"""
    steps = dbg["steps"].astype(np.int64)
    th = dbg["x"][..., 2]
    return {"capped": dbg["code"] == 0, "pole-crossing (final theta outside [0, pi])": (th < 0.0) | (th > np.pi),
            "throat-whirling (steps > 1.25 x the frame's median)": steps > 1.25 * np.median(steps)}


def compare_with_glibc(got_rgb, got_dbg, want_rgb, want_dbg, label):
    """GPU frame + per-ray dump against ONE glibc flavour of the oracle: returns the measured fractions and, when
    anything differs, which ray classes the differing rays belong to (printed -- the assertion is the caller's)"""
    d = np.abs(got_rgb.astype(int) - want_rgb.astype(int)).max(axis=-1)
    same = {f: got_dbg[f] == want_dbg[f] for f in ("steps", "code", "tx", "ty")}
    texel = same["tx"] & same["ty"]
    out = dict(pixels=float((d == 0).mean()), le1=float((d <= 1).mean()), texel=float(texel.mean()),
               steps=float(same["steps"].mean()), code=float(same["code"].mean()))
    bad = ~(texel & same["steps"] & same["code"]) | (d != 0)
    if bad.any():
        cls = ray_classes(got_dbg, 0)
        print("%s: %d differing rays: %s" % (label, int(bad.sum()), ", ".join(
            "%s %d" % (k, int((v & bad).sum())) for k, v in cls.items())))
    return out


@pytest.mark.parametrize("video", ["orbit", "through"])
def test_video_config_every_frame_against_glibc(gpu_ctx, video):
    """The arithmetic-independent parity check (it shares no elementary function with the product: the oracle's glibc
    flavours call libm's sin/cos/sincos/atan/log/acos/atan2, src/metrics.rs:68,257,262,461-485) on the camera poses of
    the two VIDEO configs -- every one of the 240 orbit poses at l = 3 and of the 480 fly-through poses from l = -4
    through the Interstellar throat (|l| < 0.02) to l = +4 --, ALL THREE glibc flavours, reduced resolution, checkerboard
    sky (a pixel is right only if the exact texel is): pixels, raw texel indices, step counts and escape codes of every
    ray.  Measured: identical for every ray of every frame (CPU pre-run of round 5 and this test on the device); that is
    what is asserted."""
    metric, csv, fps, n_frames, res, _, cap = VIDEOS[video]
    times, poses = video_poses(csv, fps)
    om, pm = metrics_of(metric)
    sp, sn = common.make_skies(512, 256, "check")
    osp, osn = O.sky(sp), O.sky(sn)
    gpu_ctx.set_sky(0, curvis_amd.SphericalImage(sp))
    gpu_ctx.set_sky(1, curvis_amd.SphericalImage(sn))

    def work(p):
        oc = O.camera(p[0], p[1], p[2], 15.0, 43.0, res)
        return [O.render_image(fl, om, oc, osp, osn, cap, 100.0, 0.05, debug=True)[:2] for fl in O.GLIBC_FLAVOURS]
    common.oracle_budget("%s: %d poses at %dx%d, three glibc flavours" % (video, n_frames, res[0], res[1]), metric, O.LIBM,
                         3.0 * n_frames * res[0] * res[1] * 2000, threads=THREADS)
    with ThreadPoolExecutor(THREADS) as ex:
        want = list(ex.map(work, poses))
    worst = {fl: dict(pixels=1.0, le1=1.0, texel=1.0, steps=1.0, code=1.0) for fl in O.GLIBC_FLAVOURS}
    state_differs = 0
    for k, p in enumerate(poses):
        cam = curvis_amd.Camera(p[0], p[1], p[2], 15.0, 43.0, res[0], res[1])
        rgb, _, dbg = gpu_ctx.render_brute(pm, cam, cap, 100.0, 0.05, debug=True)
        for fl, (w_rgb, w_dbg) in zip(O.GLIBC_FLAVOURS, want[k]):
            m = compare_with_glibc(rgb, dbg, w_rgb, w_dbg, "%s frame %d (l = %.4f) vs %s" % (video, k, p[0][1], O.FLAVOUR_NAMES[fl]))
            for key, v in m.items():
                worst[fl][key] = min(worst[fl][key], v)
            state_differs += int((dbg["x"][..., 1:3].view(np.uint64) != w_dbg["x"][..., 1:3].view(np.uint64)).any())
    for fl in O.GLIBC_FLAVOURS:
        print("%s, %d frames at %dx%d vs %s: worst frame -- pixels identical %.6f, <= 1 LSB %.6f, texel indices %.6f, step "
              "counts %.6f, escape codes %.6f" % ((video, n_frames) + res + (O.FLAVOUR_NAMES[fl],) + tuple(
                  worst[fl][k] for k in ("pixels", "le1", "texel", "steps", "code"))))
        assert all(v == 1.0 for v in worst[fl].values()), (O.FLAVOUR_NAMES[fl], worst[fl])
    # ... while the trajectories themselves do differ from glibc's in their last bits (else this would prove nothing)
    assert state_differs > n_frames


@pytest.mark.parametrize("video", ["orbit", "through"])
def test_video_config_full_size_frames_against_glibc(gpu_ctx, video):
    """FULL-size frames of the two video configs at the hard poses -- orbit l = 3 (pole-crossing rays, the survey's
    min-211-step rays) at 1920x1080; the fly-through's frame 0 (l = -4) and the two frames nearest l = 0 (camera inside
    the Interstellar throat) at 3840x2160 cap 8192 -- against the glibc flavour LLVM's own lowering points at
    (CVO_LIBM_SINCOS_INL, profiles/round3_llvm_sincos_probe.txt), every 8th row (sized for the 16-CPU quota of a GPU box;
    all three flavours over all rows: tools/gpu_libm_parity.py -> profiles/round5_libm_parity_poses.txt), smooth AND
    checkerboard sky semantics through the raw texel indices.  Measured and asserted: every pixel, texel index, step count
    and escape code identical; the populations of the ill-conditioned ray classes are printed."""
    metric, csv, fps, n_frames, _, res, cap = VIDEOS[video]
    times, poses = video_poses(csv, fps)
    sel = [0, 60] if video == "orbit" else full_size_frames(video, n_frames, poses)
    om, pm = metrics_of(metric)
    sp, sn = common.make_skies(8192, 4096, "smooth")
    osp, osn = O.sky(sp), O.sky(sn)
    gpu_ctx.set_sky(0, curvis_amd.SphericalImage(sp))
    gpu_ctx.set_sky(1, curvis_amd.SphericalImage(sn))
    W, H = res
    T = common.host_threads(64)
    common.oracle_budget("%s: every 8th row of %d full-size frames, glibc" % (video, len(sel)), metric, O.LIBM_SINCOS_INL,
                         len(sel) * W * (H // 8) * 2000.0, threads=T)
    for i in sel:
        p = poses[i]
        cam = curvis_amd.Camera(p[0], p[1], p[2], 15.0, 43.0, W, H)
        rgb, _, dbg = gpu_ctx.render_brute(pm, cam, cap, 100.0, 0.05, debug=True)
        oc = O.camera(p[0], p[1], p[2], 15.0, 43.0, res)
        w_rgb = np.zeros((H, W, 3), np.uint8)
        w_dbg = np.zeros((H, W), O.RAY_DEBUG)

        def work(t):
            r, d, st = O.render_image(O.LIBM_SINCOS_INL, om, oc, osp, osn, cap, 100.0, 0.05, row_begin=8 * t, row_step=8 * T,
                                      debug=True)
            w_rgb[8 * t::8 * T] = r[8 * t::8 * T]
            w_dbg[8 * t::8 * T] = d[8 * t::8 * T]
            return st.rays
        with ThreadPoolExecutor(T) as ex:
            assert sum(ex.map(work, range(T))) == W * len(range(0, H, 8))
        g_rgb, g_dbg = rgb.reshape(H, W, 3)[::8], dbg.reshape(H, W)[::8]
        label = "%s frame %d (l = %.4f), %dx%d every 8th row vs %s" % (video, i, p[0][1], W, H, O.FLAVOUR_NAMES[O.LIBM_SINCOS_INL])
        m = compare_with_glibc(g_rgb, g_dbg, w_rgb[::8], w_dbg[::8], label)
        cls = ray_classes(g_dbg, cap)
        print("%s: pixels identical %.6f, <= 1 LSB %.6f, texel indices %.6f, step counts %.6f, escape codes %.6f; rays %d (%s); "
              "final state bit-identical for %.4f of them" % ((label,) + tuple(m[k] for k in ("pixels", "le1", "texel", "steps", "code")) + (
                  g_dbg.size, ", ".join("%s %d" % (k, int(v.sum())) for k, v in cls.items()),
                  float((g_dbg["x"][..., 1:3].view(np.uint64) == w_dbg[::8]["x"][..., 1:3].view(np.uint64)).all(axis=-1).mean()))))
        assert all(v == 1.0 for v in m.values()), (label, m)
        assert (g_dbg["x"][..., 1:3].view(np.uint64) != w_dbg[::8]["x"][..., 1:3].view(np.uint64)).any()


@pytest.mark.parametrize("video", ["orbit", "through"])
def test_video_config_every_frame_efficient_against_glibc(gpu_ctx, video):
    """The reference's DEFAULT renderer (render_image_efficient, src/systems.rs:333-527 -- what `curvis video` runs) at every
    pose of the two video configs against all three glibc flavours of the oracle: the adaptive sampler's tables (every alpha,
    every escape space) and EVERY pixel, checkerboard sky -- the pixel ON the optical axis (W/2, H/2) included.  There the rotation
    axis of step 5 is cam_bg x out_bg with out_bg = cam_bg in exact arithmetic (src/systems.rs:415-416, :498-506): the reference
    normalises either an exact zero (NaN -> `as u32` = texel (0, 0), the survey's edge fixture) or a vector of rounding noise, and
    which of the two depends on the last bit of the two per-frame HOST values cam_bg / rot_bg.  Rounds 1-5 took those with
    cv_math.h and got the exact zero in 238 of the 240 orbit frames where glibc gets it in all 240 (frames 29 and 168 were an
    expected-list here).  Round 6 takes them over the platform libm like the reference does (csrc/cv_frame_host.h): no pixel of
    any frame differs from any flavour, asserted without exceptions."""
    metric, csv, fps, n_frames, _, _, cap = VIDEOS[video]
    res = (192, 108) if video == "orbit" else (128, 72)
    times, poses = video_poses(csv, fps)
    om, pm = metrics_of(metric)
    sp, sn = common.make_skies(2048, 1024, "check")
    osp, osn = O.sky(sp), O.sky(sn)
    gpu_ctx.set_sky(0, curvis_amd.SphericalImage(sp))
    gpu_ctx.set_sky(1, curvis_amd.SphericalImage(sn))

    def work(p):
        oc = O.camera(p[0], p[1], p[2], 15.0, 43.0, res)
        return [O.render_image_efficient(fl, om, oc, osp, osn, cap, 100.0, 0.05, 100, 100, 1e-5, 1e-5)[:2] for fl in O.GLIBC_FLAVOURS]
    with ThreadPoolExecutor(THREADS) as ex:
        want = list(ex.map(work, poses))
    axis_pixel = (res[1] // 2, res[0] // 2)
    axis_texel_00 = 0       # frames whose optical-axis pixel is the 0/0 case (texel (0, 0) of the + sky)
    # the fly-through (every frame its own camera radius) through the DEVICE-RESIDENT sampler, the orbit through the host-paced one
    gpu_ctx.set_option("device_sampler", 1 if video == "through" else 0)
    for k0 in range(0, n_frames, 32):
        part = poses[k0:k0 + 32]
        cams = [curvis_amd.Camera(p[0], p[1], p[2], 15.0, 43.0, res[0], res[1]) for p in part]
        rgb, _ = gpu_ctx.render_efficient(pm, cams, cap, 100.0, 0.05, 100, 100, 1e-5, 1e-5)
        assert gpu_ctx.get_option("last_sampler_path") == (1 if video == "through" else 0)
        for j in range(len(part)):
            a, e, sp_ = gpu_ctx.samples(j)
            for fl, (w_rgb, w_smp) in zip(O.GLIBC_FLAVOURS, want[k0 + j]):
                assert np.array_equal(a, w_smp["a"]) and np.array_equal(sp_, w_smp["s"]), ("sample table", k0 + j, O.FLAVOUR_NAMES[fl])
                assert np.nanmax(np.abs(e - w_smp["e"])) < 1e-7      # measured: 7e-13 (orbit), 3e-9 (fly-through, throat poses)
                d = (rgb[j] != w_rgb).any(axis=2)
                assert not d.any(), ("frame %d vs %s: pixels %s" % (k0 + j, O.FLAVOUR_NAMES[fl], np.argwhere(d)[:5].tolist()))
            axis_texel_00 += int(np.array_equal(rgb[j][axis_pixel], sp[0, 0, :3]))
    gpu_ctx.set_option("device_sampler", -1)
    print("%s, efficient renderer, %d frames at %dx%d vs %s: sample tables and every pixel identical in every frame (optical-axis pixel "
          "included; it is texel (0, 0) of the + sky in %d frames)" % (
              video, n_frames, res[0], res[1], ", ".join(O.FLAVOUR_NAMES[fl] for fl in O.GLIBC_FLAVOURS), axis_texel_00))


@pytest.mark.parametrize("video", ["orbit", "through"])
def test_video_config_full_size_efficient_against_glibc(gpu_ctx, video):
    """the same at the configs' FULL size (1920x1080 / 3840x2160, 8192x4096 checkerboard skies) at the hard poses: orbit frames 0
    and 29 (one of the two whose optical-axis pixel differed while cam_bg / rot_bg were cv_math.h values), the fly-through's frame 0
    and the two frames inside the throat -- all three glibc flavours, sample tables and every pixel, no exceptions"""
    metric, csv, fps, n_frames, _, res, cap = VIDEOS[video]
    times, poses = video_poses(csv, fps)
    sel = [0, 29] if video == "orbit" else full_size_frames(video, n_frames, poses)
    om, pm = metrics_of(metric)
    sp, sn = common.make_skies(8192, 4096, "check")
    osp, osn = O.sky(sp), O.sky(sn)
    gpu_ctx.set_sky(0, curvis_amd.SphericalImage(sp))
    gpu_ctx.set_sky(1, curvis_amd.SphericalImage(sn))
    cams = [curvis_amd.Camera(poses[i][0], poses[i][1], poses[i][2], 15.0, 43.0, res[0], res[1]) for i in sel]
    rgb, _ = gpu_ctx.render_efficient(pm, cams, cap, 100.0, 0.05, 100, 100, 1e-5, 1e-5)
    tables = [gpu_ctx.samples(j) for j in range(len(sel))]

    def work(job):
        j, fl = job
        p = poses[sel[j]]
        oc = O.camera(p[0], p[1], p[2], 15.0, 43.0, res)
        return j, fl, O.render_image_efficient(fl, om, oc, osp, osn, cap, 100.0, 0.05, 100, 100, 1e-5, 1e-5)[:2]
    with ThreadPoolExecutor(min(THREADS, 12)) as ex:
        for j, fl, (w_rgb, w_smp) in ex.map(work, [(j, fl) for j in range(len(sel)) for fl in O.GLIBC_FLAVOURS]):
            a, e, s_ = tables[j]
            assert np.array_equal(a, w_smp["a"]) and np.array_equal(s_, w_smp["s"]) and np.nanmax(np.abs(e - w_smp["e"])) < 1e-7
            d = (rgb[j] != w_rgb).any(axis=2)
            assert not d.any(), (video, sel[j], O.FLAVOUR_NAMES[fl], np.argwhere(d)[:5].tolist())
    print("%s frames %s at %dx%d, efficient renderer vs three glibc flavours: every pixel identical" % (video, sel, res[0], res[1]))


SIM = ("escape_radius = 100.0\nray_integration_max_itarations = %d\nray_integration_step = 0.05\n"
       "sampling_initial_nums = 100\nsampling_max_iterations = 50\n"
       "sampling_convergence_threshold_1 = 1e-5\nsampling_convergence_threshold_2 = 2e-5\n")


@pytest.mark.parametrize("video", ["orbit", "through"])
def test_video_config_through_the_binary(tmp_path, video):
    """`curvis video --mode brute --stats`: all frames of the config's video at reduced resolution, PNG frames and the
    per-frame JSON statistics against the oracle; two device workers (two GPUs when the box has them, else both on the one GPU), one of whose render
    calls is made to fail once (the batch is re-queued on the other worker), then --resume after deleting frames."""
    metric, csv, fps, n_frames, res, _, cap = VIDEOS[video]
    times, poses = video_poses(csv, fps)
    om, _ = metrics_of(metric)
    sp, sn = common.make_skies(512, 256, "check")
    want = oracle_frames(om, poses, res, sp, sn, cap)
    d = tmp_path
    (d / "out").mkdir()
    pngio.write_png(d / "pos.png", sp)
    pngio.write_png(d / "neg.png", sn)
    (d / "sim.toml").write_text(SIM % cap)
    (d / "cam.toml").write_text("resolution_x = %d\nresolution_y = %d\ndiagonal = 43.0\nfocal_length = 15.0\n" % res)
    (d / "vid.toml").write_text('video_name = "v"\nframe_rate = %r\nfilepath_to_camera_path = "%s"\n' % (fps, refpaths.reference_path_file(csv)))
    args = [BIN, "video", d / "pos.png", d / "neg.png", d / "out", "-v", d / "vid.toml", "-s", d / "sim.toml", "-c", d / "cam.toml",
            "--mode", "brute", "--batch", "16", "--devices", "2", "--sky-broadcast", "upload", "--stats", d / "st.jsonl"]
    if metric == "interstellar":
        (d / "met.toml").write_text("m = 0.1\na = 0.0001\nrho = 1.0\n")
        args += ["-m", d / "met.toml"]
    env = common.share_env(2, CURVIS_TEST_FAIL_BATCH="1:2")   # two real GPUs when the box has them
    r = subprocess.run([str(a) for a in args], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "injected test fault" in r.stderr and "re-queued" in r.stderr
    assert "Rendering %d frames..." % n_frames in r.stdout
    lines = {}
    for ln in (d / "st.jsonl").read_text().splitlines():
        rec = json.loads(ln)
        lines[rec["frame"]] = rec
    assert sorted(lines) == list(range(n_frames))
    for k in range(n_frames):
        rec = lines[k]
        got = (rec["rays"], rec["steps"], rec["n_pos"], rec["n_neg"], rec["n_none"], rec["n_oob"])
        assert got == want[k][1], "frame %d statistics %s vs %s" % (k, got, want[k][1])
        assert rec["time"] == times[k] and rec["mode"] == "brute"
    for k in range(0, n_frames, 7):
        assert np.array_equal(pngio.read_png(d / "out" / "tmp" / ("frame_%d.png" % k)), want[k][0]), k
    # --resume: frames already on disk are kept (and not rendered again), missing ones are rendered
    gone = [3, n_frames // 2, n_frames - 1]
    for k in gone:
        os.remove(d / "out" / "tmp" / ("frame_%d.png" % k))
    keep = d / "out" / "tmp" / "frame_5.png"
    stamp = os.stat(keep).st_mtime_ns
    r = subprocess.run([str(a) for a in args] + ["--resume"], capture_output=True, text=True, timeout=900,
                       env=common.share_env(2))
    assert r.returncode == 0, r.stderr[-2000:]
    assert "Resuming: %d of %d frames already present" % (n_frames - len(gone), n_frames) in r.stdout
    assert os.stat(keep).st_mtime_ns == stamp
    resumed = sorted(json.loads(ln)["frame"] for ln in (d / "st.jsonl").read_text().splitlines())
    assert resumed == sorted(gone)
    for k in gone:
        assert np.array_equal(pngio.read_png(d / "out" / "tmp" / ("frame_%d.png" % k)), want[k][0]), k
    assert len(os.listdir(d / "out" / "tmp")) == n_frames


@pytest.mark.parametrize("mode", ["efficient", "brute"])
def test_python_video_driver_modes(gpu_ctx, mode):
    """curvis_amd.rendering.VideoRenderingSystem (mirror of src/rendering.rs:178-327): the default mode renders what
    the reference's video loop renders -- render_image_efficient with sampling_initial_nums for BOTH alphas_num and
    max_iterations_sampling and threshold_1 for BOTH thresholds (src/main.rs:91-110, src/rendering.rs:299-307) --,
    mode="brute" the per-pixel integrator; frames and per-frame statistics against the oracle."""
    sp, sn = common.make_skies(512, 256, "check")
    gpu_ctx.set_sky(0, curvis_amd.SphericalImage(sp))
    gpu_ctx.set_sky(1, curvis_amd.SphericalImage(sn))
    it = rendering.Interpolator.from_file(refpaths.reference_path_file("path_through.csv"))
    res, cap = (40, 24), 4096
    v = rendering.VideoRenderingSystem(curvis_amd.InterstellarMetric(0.1, 1e-4, 1.0), gpu_ctx, it, 1.5, res, 43.0, 15.0, 100.0,
                                       cap, 0.05, batch=7, mode=mode, sampling_initial_nums=60,
                                       sampling_convergence_threshold_1=2e-5)
    frames = {}
    stats = v.render(on_frame=lambda k, rgb, d: frames.__setitem__(k, rgb))
    times = v.times_of_frames()
    assert [d["frame"] for d in stats] == list(range(len(times))) == sorted(frames) and len(times) == 30
    om = O.interstellar(0.1, 1e-4, 1.0)
    for d in stats:
        k = d["frame"]
        t = times[k]
        oc = O.camera(tuple(it.camera_position(t)), tuple(it.camera_forward(t)), tuple(it.camera_up(t)), 15.0, 43.0, res)
        if mode == "efficient":
            want, smp, st = O.render_image_efficient(O.CV, om, oc, O.sky(sp), O.sky(sn), cap, 100.0, 0.05, 60, 60, 2e-5, 2e-5)
            assert d["steps"] == smp["steps"] and d["rays"] == res[0] * res[1], k
            assert (d["n_pos"], d["n_neg"], d["n_none"]) == (st.n_pos, st.n_neg, st.n_none), k
        else:
            want, _, st = O.render_image(O.CV, om, oc, O.sky(sp), O.sky(sn), cap, 100.0, 0.05)
            assert (d["rays"], d["steps"], d["n_pos"], d["n_neg"], d["n_none"]) == (st.rays, st.steps, st.n_pos, st.n_neg, st.n_none), k
        assert np.array_equal(frames[k], want), k
        assert d["mode"] == mode and d["batch_frames"] in (7, 2)
