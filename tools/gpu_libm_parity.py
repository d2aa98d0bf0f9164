"""GPU (cv_math.h) against the oracle's glibc flavours -- the arithmetic a Linux build of the reference performs
(Rust f64::sin/cos/atan/ln -> llvm intrinsics -> glibc libm, src/metrics.rs:68,257,262) -- at the FULL size of
BASELINE configs[0..2], ray by ray, and the elementary functions themselves on the arguments the Euler loop
produces, against glibc and against binary128 (libquadmath).  ALL THREE glibc arithmetics are measured: sin and cos
as separate libcalls (CVO_LIBM), merged into one sincos() per reference function (CVO_LIBM_SINCOS), and merged with
update_relativistic_object fully inlined (CVO_LIBM_SINCOS_INL) -- which of them rustc/LLVM emits for the reference
cannot be determined here (no Rust toolchain).  Writes profiles/round3_libm_parity.txt (run on the
GPU box: python tools/gpu_libm_parity.py > gpurun_out/libm_parity.txt).  The numbers this prints are the ones the
parity tests assert (tests/test_gpu_parity.py, tests/test_golden.py)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import common  # noqa: E402
import oracle_lib as O  # noqa: E402
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import refpaths  # noqa: E402  (the reference's own camera paths: tests/golden/paths)
import curvis_amd  # noqa: E402
from curvis_amd import skies  # noqa: E402

THREADS = common.host_threads(128)   # the cgroup's CPU quota, not the machine's core count


def ulp_diff(a, b):
    """|a - b| in ulps of b, for finite same-sign doubles (bit distance)"""
    ia = np.ascontiguousarray(a).view(np.int64)
    ib = np.ascontiguousarray(b).view(np.int64)
    return np.abs(ia - ib)


def compare_frame(ctx, name, metric, res, cap, sky_res, pose=None):
    if pose is None:
        om, oc, pm, pc = common.scene(metric, res=res)
    else:
        om, oc, pm, pc = common.scene(metric, res=res, pos=pose[0], fwd=pose[1], up=pose[2])
    sp, sn = skies.smooth(sky_res[0], sky_res[1], 128), skies.smooth(sky_res[0], sky_res[1], 32)
    cp, cn = skies.checker(sky_res[0], sky_res[1], seed=0xC0FFEE), skies.checker(sky_res[0], sky_res[1], seed=0xBADC0DE)
    sys_ = curvis_amd.RelativisticSystem(pm, curvis_amd.SphericalImage(sp), curvis_amd.SphericalImage(sn), pc, context=ctx)
    got_rgb, got = sys_.render_image_debug(cap, 100.0, 0.05)
    n = got.size
    print("## %s: %s %dx%d cap %d, skies %dx%d, %d rays" % (name, metric, res[0], res[1], cap, sky_res[0], sky_res[1], n))
    if pose is not None:
        # the ray classes whose parity is ill-conditioned
        steps = got["steps"].astype(np.int64)
        th = got["x"][..., 2]
        print("camera (t, l, theta, phi) = %s; ray classes: capped %d, pole-crossing (final theta outside [0, pi]) %d, throat-whirling "
              "(steps > 1.25 x median %d) %d; step counts %d .. %d" % (
                  tuple(round(float(v), 6) for v in pose[0]), int((got["code"] == 0).sum()), int(((th < 0) | (th > np.pi)).sum()),
                  int(np.median(steps)), int((steps > 1.25 * np.median(steps)).sum()), int(steps.min()), int(steps.max())))
    res_all = {}
    for fl in O.GLIBC_FLAVOURS:
        res_all[fl] = compare_with_flavour(fl, got_rgb, got, om, oc, sp, sn, cp, cn, cap)
    print()
    return res_all


def compare_with_flavour(fl, got_rgb, got, om, oc, sp, sn, cp, cn, cap):
    n = got.size
    t0 = time.time()
    want_rgb, want, _ = common.oracle_full_frame(fl, om, oc, sp, sn, cap, threads=THREADS)
    t_or = time.time() - t0
    d = np.abs(got_rgb.astype(int) - want_rgb.astype(int)).max(axis=2)
    same_steps = got["steps"] == want["steps"]
    same_code = got["code"] == want["code"]
    same_texel = (got["tx"] == want["tx"]) & (got["ty"] == want["ty"])
    # checkerboard sky: colours from the raw indices of both sides (nearest texel, src/images.rs:115-121)
    def lookup(dbg):
        H, W = cp.shape[:2]
        ty, tx = np.minimum(dbg["ty"], H - 1), np.minimum(dbg["tx"], W - 1)
        out = np.where((dbg["code"] == 1)[..., None], cp[ty, tx, :3], cn[ty, tx, :3])
        out[dbg["code"] == 0] = 0
        return out
    same_check = (lookup(got) == lookup(want)).all(axis=2)
    cls = same_steps & same_code
    state_same = np.ones(got.shape, bool)
    worst = 0
    for f, idx in (("x", (1, 2)), ("p", (1, 2))):  # l, theta, p_l, p_theta (phi is not read by render_image)
        for i in idx:
            g, w = got[f][..., i], want[f][..., i]
            eq = g.view(np.uint64) == w.view(np.uint64)
            state_same &= eq
            sel = cls & ~eq & np.isfinite(g) & np.isfinite(w) & (np.sign(g) == np.sign(w))
            if sel.any():
                worst = max(worst, int(ulp_diff(g[sel], w[sel]).max()))
    bad_rows = sorted(set(np.nonzero(~(d == 0))[0].tolist()))
    print("### vs %s (oracle %.1f s on %d threads)" % (O.FLAVOUR_NAMES[fl], t_or, THREADS))
    print("pixels, smooth sky: identical %d of %d (%.6f), <= 1 LSB per channel %.6f, max difference %d" % (
        int((d == 0).sum()), n, (d == 0).mean(), (d <= 1).mean(), int(d.max())))
    print("pixels, checkerboard sky (exact texel needed): identical %d of %d (%.6f)" % (int(same_check.sum()), n, same_check.mean()))
    print("raw texel indices (tx, ty) identical: %d of %d (%.6f)" % (int(same_texel.sum()), n, same_texel.mean()))
    print("step counts identical: %d (%.6f); escape codes identical: %d (%.6f)" % (
        int(same_steps.sum()), same_steps.mean(), int(same_code.sum()), same_code.mean()))
    print("final (l, theta, p_l, p_theta) bit-identical: %d of %d (%.4f); largest difference among rays with the same "
          "step count and code: %d ulp" % (int(state_same.sum()), n, state_same.mean(), worst))
    if bad_rows:
        print("rows with differing pixels: %s%s" % (bad_rows[:40], " ..." if len(bad_rows) > 40 else ""))
    return dict(exact=(d == 0).mean(), le1=(d <= 1).mean(), texel=same_texel.mean(), steps=same_steps.mean(),
                code=same_code.mean())


def compare_efficient(ctx):
    om, oc, pm, pc = common.scene("ellis", res=(960, 540))
    sp, sn = common.make_skies(2048, 1024, "smooth")
    sys_ = curvis_amd.RelativisticSystem(pm, curvis_amd.SphericalImage(sp), curvis_amd.SphericalImage(sn), pc, context=ctx)
    got = sys_.render_image_efficient(40000, 100.0, 0.05, 100, 100, 1e-5, 1e-5)
    a, e, s = ctx.samples(0)
    print("## render_image_efficient, default image 960x540 (cap 40000, n0 = 100, thr 1e-5), smooth 2048x1024 skies")
    for fl in O.GLIBC_FLAVOURS:
        want, smp, _ = O.render_image_efficient(fl, om, oc, O.sky(sp), O.sky(sn), 40000, 100.0, 0.05, 100, 100, 1e-5, 1e-5)
        d = np.abs(got.astype(int) - want.astype(int)).max(axis=2)
        print("### vs %s" % O.FLAVOUR_NAMES[fl])
        print("sample tables: %d vs %d points, alphas identical %s, escape spaces identical %s, max |escape angle difference| %.3g" % (
            len(a), len(smp["a"]), bool(np.array_equal(a, smp["a"])), bool(np.array_equal(s, smp["s"])),
            float(np.nanmax(np.abs(e - smp["e"]))) if len(a) == len(smp["a"]) else float("nan")))
        print("pixels: identical %d of %d (%.6f), <= 1 LSB %.6f, max difference %d" % (
            int((d == 0).sum()), d.size, (d == 0).mean(), (d <= 1).mean(), int(d.max())))
    print()


def function_sweep(ctx):
    """cv_math.h ON THE DEVICE against glibc and binary128 on the arguments of the loop: theta of every step of
    2048 rays of the config-2 camera (Ellis) and of 1024 rays of the Interstellar camera (+ x and 1 + x^2 of r(l)),
    the acos / atan2 arguments of the sky lookup of a whole 960x540 frame."""
    print("## elementary functions on the arguments the loop produces: device cv_math.h vs glibc vs binary128")
    print("| metric | function | arguments | device == host twin | identical to glibc | max |cv - glibc| (ulp) | max error cv (ulp) | max error glibc (ulp) |")
    print("|---|---|---|---|---|---|---|---|")
    rng = np.random.default_rng(7)
    for metric, nrays in (("ellis", 2048), ("interstellar", 1024)):
        om, oc, pm, pc = common.scene(metric, res=(1920, 1080))
        px = rng.integers(0, 1920, nrays)
        py = rng.integers(0, 1080, nrays)
        dirs = np.zeros((nrays, 3))
        for i in range(nrays):
            O.lib().cvo_camera_outward_world(O.C.byref(oc), int(px[i]), int(py[i]), O._dp(dirs[i]))
        pos = np.tile(np.array([0.0, 5.0, np.pi / 2, 0.0]), (nrays, 1))
        tr = ctx.compute_photon_trajectory(pm, pos, dirs, 2200, 0.05)
        keep = np.abs(tr[:, :, 1]) <= 100.0
        th, ls = tr[:, :, 2][keep], tr[:, :, 1][keep]
        args = {"sin": (0, th), "cos": (1, th)}
        if metric == "interstellar":
            x = 2.0 * (np.abs(ls) - 1e-4) / (np.pi * 0.1)
            x = x[np.abs(ls) > 1e-4]
            args.update({"atan": (2, x), "log": (4, 1.0 + x * x)})
        for name, (op, xs) in args.items():
            dev = ctx.selftest_math(op, xs)
            host = common.twin_math(op, xs)
            gl = O.math_array(O.LIBM, op, xs)
            same = dev == gl
            off = ulp_diff(dev[~same], gl[~same]).max() if (~same).any() else 0
            print("| %s | %s | %d | %s | %.4f %% | %d | %.3f | %.3f |" % (
                metric, name, xs.size, bool(np.array_equal(dev.view(np.uint64), host.view(np.uint64))), 100 * same.mean(),
                off, O.quad_ulp_errors(op, xs, dev).max(), O.quad_ulp_errors(op, xs, gl).max()))
    # sky lookup: acos(z / |v|) and atan2(y, x) of the final directions of a frame
    om, oc, pm, pc = common.scene("ellis", res=(960, 540))
    sp, sn = common.make_skies(512, 256, "smooth")
    sys_ = curvis_amd.RelativisticSystem(pm, curvis_amd.SphericalImage(sp), curvis_amd.SphericalImage(sn), pc, context=ctx)
    _, dbg = sys_.render_image_debug(4096, 100.0, 0.05)
    esc = dbg["code"] != 0
    x, p = dbg["x"][esc], dbg["p"][esc]
    r2 = 1.0 + x[:, 1] * x[:, 1]
    r = np.sqrt(r2)
    s = np.sin(x[:, 2])
    d0, d1, d2 = p[:, 1], p[:, 2] * (1.0 / r2) * r, p[:, 3] * (1.0 / (r2 * (s * s))) * r
    rn = np.sqrt(d0 * d0 + d1 * d1 + d2 * d2)
    for name, op, a, b in (("acos", 3, d2 / rn, None), ("atan2", 5, d1, d0)):
        dev = ctx.selftest_math(op, a, b)
        host = common.twin_math(op, a, b)
        gl = O.math_array(O.LIBM, op, a, b)
        same = dev == gl
        off = ulp_diff(dev[~same], gl[~same]).max() if (~same).any() else 0
        print("| ellis (sky lookup) | %s | %d | %s | %.4f %% | %d | %.3f | %.3f |" % (
            name, a.size, bool(np.array_equal(dev.view(np.uint64), host.view(np.uint64))), 100 * same.mean(), off,
            O.quad_ulp_errors(op, a, dev, b).max(), O.quad_ulp_errors(op, a, gl, b).max()))
    print()


def poses_main():
    """the same comparison at the camera poses of the two VIDEO configs -- orbit l = 3 (pole-crossing
    rays), the fly-through's frame 0 (l = -4) and its two frames nearest l = 0 (camera inside the Interstellar throat) -- at the
    configs' FULL sizes, all three glibc flavours, every ray.  python tools/gpu_libm_parity.py poses > profiles/round5_libm_parity_poses.txt"""
    from curvis_amd import paths, rendering
    ctx = curvis_amd.Context(0)
    print("# GPU (fast step, cv_math.h) vs the oracle's three glibc arithmetics (%s) at the camera poses of configs[3] and configs[4], full size" % (
        os.confstr("CS_GNU_LIBC_VERSION")))
    print("device: %s; host threads used by the oracle: %d" % (ctx.device_info()["name"], THREADS))
    print()

    def video_poses(csv, fps):
        it = rendering.Interpolator.from_file(refpaths.reference_path_file(csv))
        times = rendering.times_of_frames(it.min_time(), it.max_time(), fps)
        return [(tuple(it.camera_position(t)), tuple(it.camera_forward(t)), tuple(it.camera_up(t))) for t in times]
    orbit = video_poses("path_orbit.csv", 4.0)
    for k in (0, 60):
        compare_frame(ctx, "configs[3] frame %d of 240" % k, "ellis", (1920, 1080), 4096, (8192, 4096), pose=orbit[k])
    through = video_poses("path_through.csv", 24.0)
    ls = np.array([abs(p[0][1]) for p in through])
    for k in [0] + sorted(np.argsort(ls)[:2].tolist()):
        compare_frame(ctx, "configs[4] frame %d of 480" % k, "interstellar", (3840, 2160), 8192, (8192, 4096), pose=through[k])
    ctx.close()


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "poses":
        return poses_main()
    ctx = curvis_amd.Context(0)
    print("# GPU (fast step, cv_math.h) vs the oracle's three glibc arithmetics (%s): sin/cos separate, one sincos() per "
          "reference function, sincos() with update inlined -- full-size BASELINE configurations" % (
        os.confstr("CS_GNU_LIBC_VERSION")))
    print("device: %s; host threads used by the oracle: %d" % (ctx.device_info()["name"], THREADS))
    print()
    compare_frame(ctx, "configs[0]", "ellis", (256, 144), 40000, (512, 256))
    compare_frame(ctx, "configs[1]", "ellis", (1920, 1080), 4096, (8192, 4096))
    compare_frame(ctx, "configs[2]", "interstellar", (3840, 2160), 8192, (8192, 4096))
    compare_efficient(ctx)
    function_sweep(ctx)
    ctx.close()


if __name__ == "__main__":
    main()
