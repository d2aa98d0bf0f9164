"""GPU parity tests proper: the HIP path (through the C ABI) against the CPU oracle on the same
seeded inputs.  Bit-exact against the oracle's CVO_CV flavour (same IEEE operation sequence).
Against the glibc-libm flavour (the arithmetic of a Linux build of the reference) the tolerance BASELINE.json's
north_star states is "<= 1 ULP per channel"; what is MEASURED (profiles/round2_libm_parity.txt: configs[0..2] at
full size, the default efficient image) is every pixel, texel index, step count and escape code IDENTICAL, and the
tests assert exactly that -- a regression that changes a single pixel fails."""
import numpy as np
import pytest

import common
import oracle_lib as O
import curvis_amd

pytestmark = pytest.mark.gpu


def test_device_math_bit_identical_to_host(gpu_ctx):
    """cv_math.h on gfx950 == cv_math.h on x86, IEEE division and sqrt correctly rounded on device."""
    rng = np.random.default_rng(1234)
    n = 1 << 20
    u = rng.uniform
    cases = {
        0: np.concatenate([u(-4, 7, n), u(-1e6, 1e6, n // 4), np.ldexp(u(1, 2, n // 4), rng.integers(20, 1023, n // 4)),
                           np.array([0.0, -0.0, np.pi / 2, np.pi, np.inf, np.nan, 1e-310, 5e-324]),
                           np.concatenate([sg * (6.25 + u(-1e-9, 1e-9, 4096)) for sg in (1.0, -1.0)]),   # main <-> other path
                           np.arange(-400, 401) * (np.pi / 64) + np.ldexp(u(-2, 2, 801), -20)]),     # ... and at 2^-20
        2: np.concatenate([u(-700, 700, n), 10.0 ** u(-12, 25, n // 4), np.array([0.0, np.inf, -np.inf, np.nan])]),
        3: np.concatenate([u(-1, 1, n), 1 - 10.0 ** u(-16, -1, n // 4), np.array([1.0, -1.0, 1.0000000000000002, np.nan])]),
        4: np.concatenate([u(1, 1e6, n), 10.0 ** u(-300, 300, n // 4), np.array([0.0, -1.0, 1.0, np.inf, 5e-324])]),
    }
    cases[1] = cases[0]
    at_edges, lg_edges = common.table_edge_inputs()   # row boundaries of the atan / log tables
    cases[2] = np.concatenate([cases[2], at_edges])
    cases[4] = np.concatenate([cases[4], lg_edges])
    for op, a in cases.items():
        got = gpu_ctx.selftest_math(op, a)
        want = common.twin_math(op, a)
        assert np.array_equal(got.view(np.uint64), want.view(np.uint64)), "cv_math op %d differs on device" % op
    a, b = u(-5, 5, n), u(-5, 5, n)
    assert np.array_equal(gpu_ctx.selftest_math(5, a, b).view(np.uint64), common.twin_math(5, a, b).view(np.uint64))
    # IEEE-754 division / sqrt / fma: numpy on the host is the reference
    a = np.concatenate([u(-1e3, 1e3, n), 10.0 ** u(-300, 300, n)])
    b = np.concatenate([u(-1e3, 1e3, n), 10.0 ** u(-300, 300, n)])
    with np.errstate(all="ignore"):
        assert np.array_equal(gpu_ctx.selftest_math(6, a, b).view(np.uint64), (a / b).view(np.uint64))
        assert np.array_equal(gpu_ctx.selftest_math(7, np.abs(a)).view(np.uint64), np.sqrt(np.abs(a)).view(np.uint64))
    assert np.array_equal(gpu_ctx.selftest_math(8, a, b).view(np.uint64), common.twin_math(8, a, b).view(np.uint64))


CASES = [
    ("ellis", (64, 36), (0.0, 5.0, common.HALF_PI, 0.0), (-1.0, 0.0, 0.0), 4096),
    ("interstellar", (64, 36), (0.0, 5.0, common.HALF_PI, 0.0), (-1.0, 0.0, 0.0), 4096),
    ("ellis", (61, 35), (0.0, 3.0, common.HALF_PI, 1.0), (-1.0, 0.1, 0.05), 2500),     # ragged tiles, cap binds
    ("interstellar", (40, 24), (0.0, -2.0, 1.2, 4.0), (1.0, 0.2, -0.1), 3000),          # camera in the -l space
    ("flat", (32, 18), (0.0, 5.0, 1.0, 0.5), (1.0, 0.3, 0.2), 4096),
]


@pytest.mark.parametrize("fast", [1, 0])
@pytest.mark.parametrize("variant", [0, 1])
@pytest.mark.parametrize("metric,res,pos,fwd,cap", CASES)
def test_full_ray_state_bit_exact_vs_oracle(gpu_ctx, variant, fast, metric, res, pos, fwd, cap):
    sp, sn = common.make_skies(256, 128, "check")
    om, oc, pm, pc = common.scene(metric, res=res, pos=pos, fwd=fwd)
    want_rgb, want_dbg, st = O.render_image(O.CV, om, oc, O.sky(sp), O.sky(sn), cap, 100.0, 0.05, debug=True)
    gpu_ctx.set_option("variant", variant)
    gpu_ctx.set_option("fast_math", fast)
    sys_ = curvis_amd.RelativisticSystem(pm, curvis_amd.SphericalImage(sp), curvis_amd.SphericalImage(sn), pc,
                                         context=gpu_ctx)
    got_rgb, got_dbg = sys_.render_image_debug(cap, 100.0, 0.05)
    common.assert_debug_equal(got_dbg, want_dbg, check_t=True)
    assert np.array_equal(got_rgb, want_rgb)
    s = sys_.last_stats
    assert (s.rays, s.steps, s.n_pos, s.n_neg, s.n_none) == (st.rays, st.steps, st.n_pos, st.n_neg, st.n_none)
    # the non-debug kernel (phi not integrated) must give the same pixels
    assert np.array_equal(sys_.render_image(cap, 100.0, 0.05), want_rgb)
    gpu_ctx.set_option("variant", -1)
    gpu_ctx.set_option("fast_math", 1)


@pytest.mark.parametrize("metric,res,pos,fwd,cap", CASES)
def test_relay_kernel_bit_exact(gpu_ctx, metric, res, pos, fwd, cap):
    """"variant" 2 (end-game hand-over of tiles between waves through HBM and a ticket ring): forced onto small
    frames with short segments so that tiles are parked and relayed many times (and the host has to launch extra
    relay workgroups); pixels and statistics must equal the oracle's."""
    sp, sn = common.make_skies(256, 128, "check")
    om, oc, pm, pc = common.scene(metric, res=res, pos=pos, fwd=fwd)
    want_rgb, _, st = O.render_image(O.CV, om, oc, O.sky(sp), O.sky(sn), cap, 100.0, 0.05, debug=True)
    gpu_ctx.set_sky(0, curvis_amd.SphericalImage(sp))
    gpu_ctx.set_sky(1, curvis_amd.SphericalImage(sn))
    try:
        gpu_ctx.set_option("variant", 2)
        gpu_ctx.set_option("relay_min_blocks", 0)
        parks = 0
        for seg in (16, 100, 0):
            gpu_ctx.set_option("relay_segment", seg)
            for _ in range(2):
                rgb, s = gpu_ctx.render_brute(pm, pc, cap, 100.0, 0.05)
                assert np.array_equal(rgb, want_rgb), seg
                assert (s.rays, s.steps, s.n_pos, s.n_neg, s.n_none) == (st.rays, st.steps, st.n_pos, st.n_neg, st.n_none)
                assert gpu_ctx.get_option("last_relay_launches") >= 1
                parks += gpu_ctx.get_option("last_relay_parks")
        if cap >= 1000 and metric != "flat":
            assert parks > 0   # the hand-over path was really exercised
        # several frames in one relay launch (up to "relay_max_frames" = 8)
        gpu_ctx.set_option("relay_segment", 50)
        for nf in (2, 5):
            rgbn, sn_ = gpu_ctx.render_brute(pm, [pc] * nf, cap, 100.0, 0.05)
            assert all(np.array_equal(rgbn[i], want_rgb) for i in range(nf))
            assert (sn_.rays, sn_.steps) == (nf * st.rays, nf * st.steps) and gpu_ctx.get_option("last_relay_launches") >= 1
        rgb9, _ = gpu_ctx.render_brute(pm, [pc] * 9, cap, 100.0, 0.05)       # beyond it: static kernel
        assert np.array_equal(rgb9[8], want_rgb) and gpu_ctx.get_option("last_relay_launches") == 0
        # other workgroup sizes ("block_threads"): one and two waves per workgroup, relay and static kernels
        for bt in (64, 128):
            gpu_ctx.set_option("block_threads", bt)
            for variant in (2, 1):
                gpu_ctx.set_option("variant", variant)
                rgb, s = gpu_ctx.render_brute(pm, pc, cap, 100.0, 0.05)
                assert np.array_equal(rgb, want_rgb), (bt, variant)
                assert (s.rays, s.steps, s.n_pos, s.n_neg, s.n_none) == (st.rays, st.steps, st.n_pos, st.n_neg, st.n_none)
    finally:
        gpu_ctx.set_option("block_threads", 0)
        gpu_ctx.set_option("variant", -1)
        gpu_ctx.set_option("relay_min_blocks", -1)
        gpu_ctx.set_option("relay_segment", 0)


@pytest.mark.parametrize("metric,res,pos,fwd,cap", CASES[:3])
def test_row_bands_equal_the_full_frame(gpu_ctx, metric, res, pos, fwd, cap):
    """curvis_render_brute_rows: any split of the rows reproduces the full frame and its statistics (all kernel
    variants; ragged bands that cut 8x8 tiles)."""
    sp, sn = common.make_skies(256, 128, "check")
    om, oc, pm, pc = common.scene(metric, res=res, pos=pos, fwd=fwd)
    gpu_ctx.set_sky(0, curvis_amd.SphericalImage(sp))
    gpu_ctx.set_sky(1, curvis_amd.SphericalImage(sn))
    full, st = gpu_ctx.render_brute(pm, pc, cap, 100.0, 0.05)
    H = res[1]
    try:
        for variant in (1, 0, 2):
            gpu_ctx.set_option("variant", variant)
            gpu_ctx.set_option("relay_min_blocks", 0)
            for cuts in ([0, H], [0, 1, H], [0, 5, 13, H - 1, H], [0, H // 2, H]):
                parts, steps, rays = [], 0, 0
                for b, e in zip(cuts, cuts[1:]):
                    rgb, s = gpu_ctx.render_brute_rows(pm, pc, b, e - b, cap, 100.0, 0.05)
                    parts.append(rgb); steps += s.steps; rays += s.rays
                assert np.array_equal(np.concatenate(parts, axis=0), full), (variant, cuts)
                assert (steps, rays) == (st.steps, st.rays)
        with pytest.raises(curvis_amd.CurvisError):
            gpu_ctx.render_brute_rows(pm, pc, H - 1, 2, cap, 100.0, 0.05)
    finally:
        gpu_ctx.set_option("variant", -1)
        gpu_ctx.set_option("relay_min_blocks", -1)


oracle_full_frame = common.oracle_full_frame


@pytest.mark.parametrize("metric,res,cap", [("ellis", (1920, 1080), 4096), ("interstellar", (960, 540), 8192)])
def test_full_size_frame_bit_exact(gpu_ctx, metric, res, cap):
    """BASELINE configs[1] at full size (and configs[2]'s metric/cap at quarter size): every ray of the
    frame -- final state, step count, texel index, pixel -- bit-exact against the oracle (cv flavour), for
    the fast (shared-reciprocal) and the strict (compiler IEEE) kernels, persistent and static."""
    sp, sn = common.make_skies(2048, 1024, "check")
    om, oc, pm, pc = common.scene(metric, res=res)
    common.oracle_budget("%s %dx%d: cv flavour + three glibc flavours" % ((metric,) + res), metric, O.LIBM, 4.0 * res[0] * res[1] * 2000)
    want_rgb, want_dbg, steps = oracle_full_frame(O.CV, om, oc, sp, sn, cap)
    sys_ = curvis_amd.RelativisticSystem(pm, curvis_amd.SphericalImage(sp), curvis_amd.SphericalImage(sn), pc,
                                         context=gpu_ctx)
    # -1 = the library's automatic choice: at these sizes the relay kernel for the production render (checked below
    # through last_relay_launches), the static kernel for the debug dump
    for variant, fast in [(0, 1), (1, 1), (0, 0), (-1, 1), (2, 0)]:
        gpu_ctx.set_option("variant", variant)
        gpu_ctx.set_option("fast_math", fast)
        got_rgb, got_dbg = sys_.render_image_debug(cap, 100.0, 0.05)
        common.assert_debug_equal(got_dbg, want_dbg, check_t=True)
        assert np.array_equal(got_rgb, want_rgb)
        assert sys_.last_stats.steps == steps
        # the production (non-debug) kernels: fused epilogue shading and the staged two-kernel form
        for fuse in (1, 0):
            gpu_ctx.set_option("fuse_shade", fuse)
            assert np.array_equal(sys_.render_image(cap, 100.0, 0.05), want_rgb)
            s = sys_.last_stats
            assert (s.steps, s.n_pos + s.n_neg + s.n_none) == (steps, res[0] * res[1])
            if fuse and variant in (-1, 2):
                assert gpu_ctx.get_option("last_relay_launches") >= 1 and gpu_ctx.get_option("last_relay_parks") > 0
        gpu_ctx.set_option("fuse_shade", 1)
    gpu_ctx.set_option("variant", -1)
    gpu_ctx.set_option("fast_math", 1)
    # the same frame in the reference's own arithmetic -- each of the three glibc flavours (sin/cos as separate libcalls,
    # one sincos() per function, sincos() with update inlined; which one rustc emits is unknown): texel indices, step
    # counts, escape codes and therefore pixels of EVERY ray identical (measured at full size,
    # profiles/round3_libm_parity.txt)
    for fl in O.GLIBC_FLAVOURS:
        libm_rgb, libm_dbg, libm_steps = oracle_full_frame(fl, om, oc, sp, sn, cap)
        assert libm_steps == steps and np.array_equal(libm_rgb, want_rgb), O.FLAVOUR_NAMES[fl]
        for f in ("steps", "code", "tx", "ty"):
            assert np.array_equal(libm_dbg[f], want_dbg[f]), (O.FLAVOUR_NAMES[fl], f)


def test_config1_256x144_pixels(gpu_ctx):
    """BASELINE config 1 (256x144, all defaults, cap 40000): bit-exact vs oracle(cv) AND pixel-identical to
    each of the oracle's three glibc arithmetics of the reference (measured: 36 864 of 36 864, profiles/round3_libm_parity.txt)."""
    sp, sn = common.make_skies(512, 256, "smooth")
    om, oc, pm, pc = common.scene("ellis", res=(256, 144))
    sys_ = curvis_amd.RelativisticSystem(pm, curvis_amd.SphericalImage(sp), curvis_amd.SphericalImage(sn), pc,
                                         context=gpu_ctx)
    got = sys_.render_image(40000, 100.0, 0.05)
    want_cv, _, st = O.render_image(O.CV, om, oc, O.sky(sp), O.sky(sn), 40000, 100.0, 0.05)
    assert np.array_equal(got, want_cv)
    assert sys_.last_stats.steps == st.steps
    _, dbg = sys_.render_image_debug(40000, 100.0, 0.05)
    for fl in O.GLIBC_FLAVOURS:  # all three glibc arithmetics (oracle/curvis_oracle.h)
        want_libm, dbg_libm, _ = O.render_image(fl, om, oc, O.sky(sp), O.sky(sn), 40000, 100.0, 0.05, debug=True)
        diff = np.abs(got.astype(int) - want_libm.astype(int)).max(axis=2)
        frac_exact = float((diff == 0).mean())
        frac_le1 = float((diff <= 1).mean())
        print("config1 vs %s: exact %.5f, <=1 LSB %.5f, max %d" % (O.FLAVOUR_NAMES[fl], frac_exact, frac_le1, diff.max()))
        assert frac_exact == 1.0 and diff.max() == 0, "%s: rows with differing pixels: %s" % (
            O.FLAVOUR_NAMES[fl], sorted(set(np.nonzero(diff)[0].tolist())))
        # raw texel indices, step counts and escape codes of every ray as well (checkerboard-sky exactness)
        for f in ("steps", "code", "tx", "ty"):
            assert np.array_equal(dbg[f], dbg_libm[f]), (O.FLAVOUR_NAMES[fl], f)


def test_batch_equals_single_frames(gpu_ctx):
    sp, sn = common.make_skies(256, 128, "check")
    cams = []
    for k in range(3):
        _, _, pm, pc = common.scene("ellis", res=(48, 27), pos=(0.0, 3.0 + k, common.HALF_PI, 0.3 * k))
        cams.append(pc)
    gpu_ctx.set_sky(0, curvis_amd.SphericalImage(sp))
    gpu_ctx.set_sky(1, curvis_amd.SphericalImage(sn))
    batch, st = gpu_ctx.render_brute(pm, cams, 3000, 100.0, 0.05)
    total = 0
    for k, c in enumerate(cams):
        one, s1 = gpu_ctx.render_brute(pm, c, 3000, 100.0, 0.05)
        assert np.array_equal(batch[k], one)
        total += s1.steps
    assert st.steps == total and st.rays == 3 * 48 * 27


def test_camera_outside_is_an_error(gpu_ctx):
    sp, sn = common.make_skies(64, 32, "check")
    _, _, pm, pc = common.scene("ellis", res=(8, 8), pos=(0.0, 150.0, common.HALF_PI, 0.0))
    sys_ = curvis_amd.RelativisticSystem(pm, curvis_amd.SphericalImage(sp), curvis_amd.SphericalImage(sn), pc,
                                         context=gpu_ctx)
    with pytest.raises(curvis_amd.CurvisError) as e:
        sys_.render_image(100, 100.0, 0.05)
    assert e.value.code == -4


def test_zero_iterations_is_black(gpu_ctx):
    sp, sn = common.make_skies(64, 32, "check")
    _, _, pm, pc = common.scene("ellis", res=(16, 9))
    sys_ = curvis_amd.RelativisticSystem(pm, curvis_amd.SphericalImage(sp), curvis_amd.SphericalImage(sn), pc,
                                         context=gpu_ctx)
    img = sys_.render_image(0, 100.0, 0.05)
    assert img.shape == (9, 16, 3) and not img.any() and sys_.last_stats.n_none == 16 * 9


EFF_CASES = [
    ("ellis", (96, 54), (0.0, 5.0, common.HALF_PI, 0.0), (-1.0, 0.0, 0.0), 4096, 100),
    ("ellis", (64, 36), (0.0, 3.0, common.HALF_PI, 0.9), (-1.0, 0.0, 0.0), 4096, 100),
    ("interstellar", (64, 36), (0.0, 5.0, common.HALF_PI, 0.0), (-1.0, 0.0, 0.0), 4096, 100),
    ("ellis", (40, 30), (0.0, -2.5, 1.1, 2.0), (1.0, 0.3, 0.1), 3000, 60),
]


@pytest.mark.parametrize("fast", [1, 0])
@pytest.mark.parametrize("metric,res,pos,fwd,cap,n0", EFF_CASES)
def test_efficient_mode_bit_exact_vs_oracle(gpu_ctx, fast, metric, res, pos, fwd, cap, n0):
    """render_image_efficient (what `curvis image` / `curvis video` call): the adaptive sample table
    (alphas, escape angles, escape spaces), the sampler bookkeeping and every pixel, bit for bit."""
    sp, sn = common.make_skies(512, 256, "check")
    om, oc, pm, pc = common.scene(metric, res=res, pos=pos, fwd=fwd)
    want_rgb, want, st = O.render_image_efficient(O.CV, om, oc, O.sky(sp), O.sky(sn), cap, 100.0, 0.05, n0, n0, 1e-5,
                                                  1e-5)
    gpu_ctx.set_option("fast_math", fast)
    sys_ = curvis_amd.RelativisticSystem(pm, curvis_amd.SphericalImage(sp), curvis_amd.SphericalImage(sn), pc,
                                         context=gpu_ctx)
    got_rgb = sys_.render_image_efficient(cap, 100.0, 0.05, n0, n0, 1e-5, 1e-5)
    a, e, s = gpu_ctx.samples(0)
    info = gpu_ctx.sampling_info(0)
    gpu_ctx.set_option("fast_math", 1)
    assert (info.n_samples, info.calls, info.steps) == (len(want["a"]), want["calls"], want["steps"])
    assert np.array_equal(common.bits(a), common.bits(want["a"]))
    assert np.array_equal(common.bits(e), common.bits(want["e"]))
    assert np.array_equal(common.bits(s), common.bits(want["s"]))
    assert np.array_equal(got_rgb, want_rgb)
    assert sys_.last_stats.steps == want["steps"]


def test_efficient_batch_equals_single_frames(gpu_ctx):
    sp, sn = common.make_skies(256, 128, "check")
    gpu_ctx.set_sky(0, curvis_amd.SphericalImage(sp))
    gpu_ctx.set_sky(1, curvis_amd.SphericalImage(sn))
    cams = []
    for k in range(4):
        _, _, pm, pc = common.scene("ellis", res=(40, 24), pos=(0.0, 3.0, common.HALF_PI, 0.4 * k))
        cams.append(pc)
    batch, st = gpu_ctx.render_efficient(pm, cams, 4096, 100.0, 0.05, 100, 100, 1e-5, 1e-5)
    tables = [gpu_ctx.samples(k) for k in range(4)]
    for k, c in enumerate(cams):
        one, _ = gpu_ctx.render_efficient(pm, c, 4096, 100.0, 0.05, 100, 100, 1e-5, 1e-5)
        assert np.array_equal(batch[k], one)
        a, e, s = gpu_ctx.samples(0)
        assert np.array_equal(common.bits(a), common.bits(tables[k][0]))
        assert np.array_equal(common.bits(e), common.bits(tables[k][1]))


def test_efficient_default_960x540_vs_oracle(gpu_ctx):
    """the reference's default image (settings/defaults: 960x540, cap 40000, n0 = 100, thr 1e-5) through
    the CLI's renderer."""
    sp, sn = common.make_skies(2048, 1024, "smooth")
    om, oc, pm, pc = common.scene("ellis", res=(960, 540))
    want_rgb, want, st = O.render_image_efficient(O.CV, om, oc, O.sky(sp), O.sky(sn), 40000, 100.0, 0.05, 100, 100,
                                                  1e-5, 1e-5)
    sys_ = curvis_amd.RelativisticSystem(pm, curvis_amd.SphericalImage(sp), curvis_amd.SphericalImage(sn), pc,
                                         context=gpu_ctx)
    got = sys_.render_image_efficient(40000, 100.0, 0.05, 100, 100, 1e-5, 1e-5)
    assert np.array_equal(got, want_rgb)
    # against the three glibc flavours (what a Linux build of the reference computes, with or without sincos merging)
    for fl in O.GLIBC_FLAVOURS:
        libm_rgb, _, _ = O.render_image_efficient(fl, om, oc, O.sky(sp), O.sky(sn), 40000, 100.0, 0.05, 100, 100, 1e-5,
                                                  1e-5)
        d = np.abs(got.astype(int) - libm_rgb.astype(int)).max(axis=2)
        print("efficient 960x540 vs %s: exact %.5f, <=1 LSB %.5f, max %d" % (O.FLAVOUR_NAMES[fl], (d == 0).mean(), (d <= 1).mean(), d.max()))
        assert d.max() == 0, O.FLAVOUR_NAMES[fl]  # measured: 518 400 of 518 400 pixels identical (profiles/round3_libm_parity.txt)


def config3_oracle_frame(threads=None):
    """oracle (cv flavour) render of BASELINE configs[2] at full size with the per-ray dump; memoised -- the background
    prefetcher (ORACLE_PREFETCH, started by conftest) computes it while the first test files run"""
    def make():
        sp, sn = common.make_skies(2048, 1024, "check")
        om, oc, _, _ = common.scene("interstellar", res=(3840, 2160))
        return oracle_full_frame(O.CV, om, oc, sp, sn, 8192, threads=threads or common.host_threads(128))
    return common.oracle_memo(("config3_oracle_frame",), make)


def ORACLE_PREFETCH(selected):
    if not any("test_config3_full_size_4k_interstellar" in n for n in selected):
        return []
    return [lambda: config3_oracle_frame(threads=common.prefetch_threads())]


def test_config3_full_size_4k_interstellar(gpu_ctx):
    """BASELINE configs[2] at full size: Interstellar (m=0.1, a=1e-4, rho=1), 3840x2160, cap 8192 --
    8 294 400 rays, 1.6e10 Euler steps -- pixels, step total and escape counts bit-exact against the oracle
    (rows striped over the host cores)."""
    import os
    # never skipped: a silently skipped BASELINE config must not read as green.  On a small host the oracle run is
    # just slower (1.6e10 steps at ~25 M steps/s per core); the GPU boxes have 128+ cores.
    sp, sn = common.make_skies(2048, 1024, "check")
    om, oc, pm, pc = common.scene("interstellar", res=(3840, 2160))
    common.oracle_budget("configs[2], 3840x2160 Interstellar: cv flavour in full + every 8th row in glibc", "interstellar", O.CV,
                         (1.0 + 1.0 / 8) * 3840 * 2160 * 2000, threads=common.host_threads(128))
    want_rgb, want_dbg, steps = config3_oracle_frame()
    sys_ = curvis_amd.RelativisticSystem(pm, curvis_amd.SphericalImage(sp), curvis_amd.SphericalImage(sn), pc,
                                         context=gpu_ctx)
    got = sys_.render_image(8192, 100.0, 0.05)
    s = sys_.last_stats
    assert np.array_equal(got, want_rgb)
    assert s.steps == steps
    assert (s.n_pos, s.n_neg, s.n_none) == (int((want_dbg["code"] == 1).sum()), int((want_dbg["code"] == -1).sum()),
                                           int((want_dbg["code"] == 0).sum()))
    # The same frame in the reference's own (glibc) arithmetic -- the one check that shares no code with the product's
    # cv_math.h: every 8th row (1 036 800 rays, 2.0e9 Euler steps: sized for the 16-CPU quota of a GPU box) in the
    # flavour LLVM's own lowering points at (one sincos() per update with g33's sine taken from it,
    # profiles/round3_llvm_sincos_probe.txt): pixel, raw texel index, step count and escape code of every ray identical.
    # All three flavours over the full frame: tools/gpu_libm_parity.py -> profiles/round3_libm_parity.txt.
    from concurrent.futures import ThreadPoolExecutor
    _, got_dbg = sys_.render_image_debug(8192, 100.0, 0.05)
    assert np.array_equal(got_dbg["steps"], want_dbg["steps"]) and np.array_equal(got_dbg["code"], want_dbg["code"])
    T = common.host_threads(64)
    osp, osn = O.sky(sp), O.sky(sn)
    H = 2160
    libm_rgb = np.zeros_like(want_rgb)
    libm_dbg = np.zeros((H, 3840), O.RAY_DEBUG)

    def work(i):
        r, d, st = O.render_image(O.LIBM_SINCOS_INL, om, oc, osp, osn, 8192, 100.0, 0.05, row_begin=8 * i, row_step=8 * T, debug=True)
        libm_rgb[8 * i::8 * T] = r[8 * i::8 * T]
        libm_dbg[8 * i::8 * T] = d[8 * i::8 * T]
        return st.rays, st.steps
    with ThreadPoolExecutor(T) as ex:
        parts = list(ex.map(work, range(T)))
    assert sum(p[0] for p in parts) == 3840 * (H // 8) and sum(p[1] for p in parts) == int(got_dbg["steps"][::8].sum())
    assert np.array_equal(got[::8], libm_rgb[::8])
    for f in ("steps", "code", "tx", "ty"):
        assert np.array_equal(got_dbg[f][::8], libm_dbg[f][::8]), f
    # ... while the trajectories themselves DO differ in their last bits on some rays (else this would prove nothing)
    assert (got_dbg["x"][::8] != libm_dbg["x"][::8]).any()


def test_rotated_skies_and_random_cameras(gpu_ctx):
    """SphericalImage orientation (src/images.rs:71-90, :132-142) and arbitrary camera poses: pixels and texel
    indices bit-exact against the oracle."""
    import ctypes as C
    rng = np.random.default_rng(42)
    sp, sn = common.make_skies(256, 128, "check")
    for trial in range(4):
        fwd_s, up_s = rng.uniform(-1, 1, 3), rng.uniform(-1, 1, 3)
        rot, inv, upo = np.zeros(9), np.zeros(9), np.zeros(3)
        assert O.lib().cvo_orientation_new(O._dp(fwd_s), O._dp(up_s), O._dp(rot), O._dp(inv), O._dp(upo)) == 0
        pos = (0.0, float(rng.uniform(-6, 6)), float(rng.uniform(0.4, 2.7)), float(rng.uniform(0, 6.2)))
        fwd, up = tuple(rng.uniform(-1, 1, 3)), tuple(rng.uniform(-1, 1, 3))
        metric = ["ellis", "interstellar", "flat", "ellis"][trial]
        if metric == "flat":
            pos = (0.0, abs(pos[1]) + 1.0, pos[2], pos[3])
        om, oc, pm, pc = common.scene(metric, res=(37, 21), pos=pos, fwd=fwd, up=up, focal=float(rng.uniform(8, 40)))
        want_rgb, want_dbg, st = O.render_image(O.CV, om, oc, O.sky(sp, inv), O.sky(sn, inv), 3000, 100.0, 0.05, debug=True)
        sys_ = curvis_amd.RelativisticSystem(pm, curvis_amd.SphericalImage(sp, fwd_s, up_s),
                                             curvis_amd.SphericalImage(sn, fwd_s, up_s), pc, context=gpu_ctx)
        got_rgb, got_dbg = sys_.render_image_debug(3000, 100.0, 0.05)
        common.assert_debug_equal(got_dbg, want_dbg)
        assert np.array_equal(got_rgb, want_rgb)
        assert np.array_equal(sys_.render_image(3000, 100.0, 0.05), want_rgb)


def _assert_debug_equal_nan_tolerant(got, want):
    """bit-exact except that NaN == NaN regardless of payload (x86 and gfx950 propagate payloads differently)."""
    for f in ("steps", "code", "tx", "ty"):
        assert np.array_equal(got[f], want[f]), f
    for f in ("x", "p"):
        g, w = got[f][..., 1:], want[f][..., 1:]
        both_nan = np.isnan(g) & np.isnan(w)
        same = (common.bits(g) == common.bits(w)) | both_nan
        bad = np.argwhere(~same)
        assert bad.size == 0, "%s differs at %s: %r vs %r" % (f, bad[0], g[tuple(bad[0])], w[tuple(bad[0])])


ADVERSARIAL = [
    # name, metric ctor args, camera pos, fwd, up, delta, cap, R, res
    ("near_pole_camera", ("ellis", 1.0), (0.0, 4.0, 0.02, 0.3), (-1.0, 0.05, 0.02), (0.0, 0.0, 1.0), 0.05, 3000, 100.0, (24, 16)),
    ("pole_crossing_rows", ("ellis", 1.0), (0.0, 3.0, common.HALF_PI, 0.0), (-1.0, 0.0, 0.0), (0.0, 0.0, 1.0), 0.05, 4096, 100.0, (64, 9)),
    ("camera_in_throat", ("ellis", 1.0), (0.0, 0.0, common.HALF_PI, 1.0), (1.0, 0.2, 0.1), (0.0, 0.0, 1.0), 0.05, 3000, 100.0, (24, 16)),
    ("negative_zero_l", ("ellis", 1.0), (0.0, -0.0, 1.0, 1.0), (-1.0, 0.2, 0.1), (0.0, 0.0, 1.0), 0.05, 3000, 100.0, (16, 12)),
    ("huge_delta", ("ellis", 1.0), (0.0, 5.0, common.HALF_PI, 0.0), (-1.0, 0.0, 0.0), (0.0, 0.0, 1.0), 7.5, 400, 100.0, (24, 16)),
    ("cap_bound", ("ellis", 1.0), (0.0, 5.0, common.HALF_PI, 0.0), (-1.0, 0.0, 0.0), (0.0, 0.0, 1.0), 1e-3, 300, 100.0, (24, 16)),
    ("tiny_rho", ("ellis", 1e-3), (0.0, 0.5, 1.2, 0.0), (-1.0, 0.01, 0.0), (0.0, 0.0, 1.0), 0.01, 6000, 20.0, (24, 16)),
    ("huge_rho", ("ellis", 1e3), (0.0, 500.0, 1.2, 0.0), (-1.0, 0.3, 0.1), (0.0, 0.0, 1.0), 10.0, 5000, 1e4, (24, 16)),
    ("interstellar_long_throat", ("interstellar", 0.5, 2.0, 1.0), (0.0, 4.0, common.HALF_PI, 0.0), (-1.0, 0.02, 0.01), (0.0, 0.0, 1.0), 0.05, 6000, 60.0, (24, 16)),
    ("interstellar_inside_throat", ("interstellar", 0.1, 1.0, 1.0), (0.0, 0.5, 1.0, 0.0), (1.0, 0.3, 0.2), (0.0, 0.0, 1.0), 0.05, 4000, 60.0, (24, 16)),
    ("strict_fallback_radius", ("ellis", 1.0), (0.0, 5.0, common.HALF_PI, 0.0), (-1.0, 0.0, 0.0), (0.0, 0.0, 1.0), 1e93, 50, 1e95, (16, 12)),
    ("flat_space", ("flat",), (0.0, 5.0, 1.0, 0.5), (-1.0, 0.2, 0.1), (0.0, 0.0, 1.0), 0.05, 4096, 100.0, (24, 16)),
    ("flat_through_origin", ("flat",), (0.0, 2.0, common.HALF_PI, 0.0), (-1.0, 0.0, 0.0), (0.0, 0.0, 1.0), 0.25, 2000, 50.0, (17, 11)),
]


@pytest.mark.parametrize("case", ADVERSARIAL, ids=[c[0] for c in ADVERSARIAL])
def test_adversarial_fast_equals_strict_equals_oracle(gpu_ctx, case):
    """Guards of the shared-reciprocal step: poles, throat, zero / negative-zero l, huge and tiny parameters,
    cap-bound rays, ranges where the host-side guard disables the shortcut.  Fast == strict == oracle."""
    name, margs, pos, fwd, up, delta, cap, R, res = case
    sp, sn = common.make_skies(128, 64, "check")
    if margs[0] == "ellis":
        om, pm = O.ellis(margs[1]), curvis_amd.EllisMetric(margs[1])
    elif margs[0] == "interstellar":
        om, pm = O.interstellar(*margs[1:]), curvis_amd.InterstellarMetric(*margs[1:])
    else:
        om, pm = O.flat(), curvis_amd.FlatSphericalMetric()
    oc = O.camera(pos, fwd, up, 15.0, 43.0, res)
    pc = curvis_amd.Camera(pos, fwd, up, 15.0, 43.0, res[0], res[1])
    with np.errstate(all="ignore"):
        want_rgb, want_dbg, st = O.render_image(O.CV, om, oc, O.sky(sp), O.sky(sn), cap, R, delta, debug=True)
    sys_ = curvis_amd.RelativisticSystem(pm, curvis_amd.SphericalImage(sp), curvis_amd.SphericalImage(sn), pc,
                                         context=gpu_ctx)
    try:
        for variant in (1, 0):
            for fast in (1, 0):
                gpu_ctx.set_option("variant", variant)
                gpu_ctx.set_option("fast_math", fast)
                got_rgb, got_dbg = sys_.render_image_debug(cap, R, delta)
                _assert_debug_equal_nan_tolerant(got_dbg, want_dbg)
                assert np.array_equal(got_rgb, want_rgb), (variant, fast)
                assert np.array_equal(sys_.render_image(cap, R, delta), want_rgb), (variant, fast)
    finally:
        gpu_ctx.set_option("variant", -1)
        gpu_ctx.set_option("fast_math", 1)


def test_rccl_sky_broadcast_entry_point(gpu_ctx):
    """curvis_ctx_bcast_skies on a single-rank RCCL communicator (the 1-GPU box cannot host more ranks):
    shapes and both textures go through ncclBroadcast and the context renders the same frame afterwards."""
    import ctypes as C
    from curvis_amd import _abi
    try:
        rccl = C.CDLL("librccl.so")
    except OSError:
        rccl = C.CDLL("/opt/rocm/lib/librccl.so")

    class UniqueId(C.Structure):
        _fields_ = [("internal", C.c_char * 128)]

    uid = UniqueId()
    assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
    comm = C.c_void_p()
    rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
    assert rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0) == 0
    try:
        sp, sn = common.make_skies(256, 128, "check")
        om, oc, pm, pc = common.scene("ellis", res=(32, 18))
        gpu_ctx.set_sky(0, curvis_amd.SphericalImage(sp))
        gpu_ctx.set_sky(1, curvis_amd.SphericalImage(sn))
        before, _ = gpu_ctx.render_brute(pm, pc, 4096, 100.0, 0.05)
        _abi.check(_abi.lib().curvis_ctx_bcast_skies(gpu_ctx._h, comm, 0), gpu_ctx._h)
        # what sits in HBM after the broadcast is the texture that went in (curvis_ctx_read_sky, used by the binary to
        # verify a broadcast on every GPU)
        assert np.array_equal(gpu_ctx.read_sky(0, 0, sp.size), sp.ravel()) and np.array_equal(gpu_ctx.read_sky(1, 1024, 4096), sn.ravel()[1024:5120])
        with pytest.raises(curvis_amd.CurvisError):
            gpu_ctx.read_sky(1, sn.size - 10, 11)
        after, _ = gpu_ctx.render_brute(pm, pc, 4096, 100.0, 0.05)
        want, _, _ = O.render_image(O.CV, om, oc, O.sky(sp), O.sky(sn), 4096, 100.0, 0.05)
        assert np.array_equal(before, want) and np.array_equal(after, want)
        # a root WITHOUT skies: the "no skies" flag travels with the shapes, so every rank (here: the one) comes
        # back with CURVIS_E_NO_SKY after the header broadcast instead of the root returning early and its peers
        # waiting inside ncclBroadcast for ever
        bare = curvis_amd.Context(0)
        try:
            rc = _abi.lib().curvis_ctx_bcast_skies(bare._h, comm, 0)
            assert rc == _abi.E_NO_SKY
            assert b"root rank" in _abi.lib().curvis_last_error(bare._h)
        finally:
            bare.close()
    finally:
        rccl.ncclCommDestroy.argtypes = [C.c_void_p]
        rccl.ncclCommDestroy(comm)


def test_library_api_escape_angles_and_trajectories(gpu_ctx):
    """src/lib.rs:37 re-exports compute_escape_angle and compute_photon_trajectory: both through the ABI,
    bit-exact against the oracle (trajectories including the t and p_t lanes)."""
    import ctypes as C
    alphas = np.concatenate([np.linspace(-0.3, 3.45, 41), [0.0, np.pi / 2, 2.9, 3.0, np.pi]])
    for name, om, pm in (("ellis", O.ellis(), curvis_amd.EllisMetric(1.0)),
                         ("interstellar", O.interstellar(), curvis_amd.InterstellarMetric(0.1, 1e-4, 1.0))):
        ang, spc, st = gpu_ctx.compute_escape_angles_range(pm, 5.0, alphas, 0.05, 4096, 100.0)
        for i, a in enumerate(alphas):
            code, want, steps = O.compute_escape_angle(O.CV, om, 5.0, float(a), 0.05, 4096, 100.0)
            assert spc[i] == code and st[i] == steps, (name, a)
            if code != 0:
                assert np.float64(ang[i]).view(np.uint64) == np.float64(want).view(np.uint64), (name, a)
            else:
                assert np.isnan(ang[i])
        pos = np.array([[0.0, 5.0, np.pi / 2, 0.0], [2.5, -3.0, 1.0, 4.0], [0.0, 0.5, 2.0, 1.0]])
        dirs = np.array([[np.cos(np.pi / 4), 0.0, np.sin(np.pi / 4)], [-1.0, 0.2, 0.3], [0.1, -0.9, 0.2]])
        traj = gpu_ctx.compute_photon_trajectory(pm, pos, dirs, 300, 0.01)
        for i in range(3):
            x, p = np.zeros(4), np.zeros(4)
            O.lib().cvo_new_photon(O.CV, C.byref(om), O._dp(pos[i].copy()), O._dp(dirs[i].copy()), O._dp(x), O._dp(p))
            for k in range(300):
                want = np.concatenate([x, p])
                assert np.array_equal(traj[i, k].view(np.uint64), want.view(np.uint64)), (name, i, k)
                O.lib().cvo_update(O.CV, C.byref(om), O._dp(x), O._dp(p), 0.01)


def test_fuzz_gpu_vs_oracle(gpu_ctx):
    """60 random scenes through the ABI (brute, all four kernel flavours on a subset) and 12 through the
    efficient renderer, against the oracle."""
    rng = np.random.default_rng(777)
    sp, sn = common.make_skies(64, 32, "check")
    gpu_ctx.set_sky(0, curvis_amd.SphericalImage(sp))
    gpu_ctx.set_sky(1, curvis_amd.SphericalImage(sn))
    try:
        for trial in range(60):
            om, oc, pm, pc, delta, cap, R = common.random_scene(rng, res=(16, 9))
            with np.errstate(all="ignore"):
                want_rgb, want_dbg, st = O.render_image(O.CV, om, oc, O.sky(sp), O.sky(sn), cap, R, delta, debug=True)
            flavours = [(1, 1), (1, 0), (0, 1), (0, 0)] if trial % 6 == 0 else [(1, 1)]
            for variant, fast in flavours:
                gpu_ctx.set_option("variant", variant)
                gpu_ctx.set_option("fast_math", fast)
                rgb, s, dbg = gpu_ctx.render_brute(pm, pc, cap, R, delta, debug=True)
                _assert_debug_equal_nan_tolerant(dbg, want_dbg)
                assert np.array_equal(rgb, want_rgb), (trial, variant, fast)
                rgb2, s2 = gpu_ctx.render_brute(pm, pc, cap, R, delta)
                assert np.array_equal(rgb2, want_rgb) and s2.steps == st.steps
    finally:
        gpu_ctx.set_option("variant", -1)
        gpu_ctx.set_option("fast_math", 1)
    done = 0
    for trial in range(40):
        om, oc, pm, pc, delta, cap, R = common.random_scene(rng, res=(16, 9))
        if om.kind == O.FLAT:
            continue
        try:
            with np.errstate(all="ignore"):
                want_rgb, want, _ = O.render_image_efficient(O.CV, om, oc, O.sky(sp), O.sky(sn), cap, R, delta, 40, 40,
                                                             1e-4, 1e-4)
        except RuntimeError:
            # the reference panics (fewer than 3 finite samples / undefined rotation): the ABI must report it too
            with pytest.raises(curvis_amd.CurvisError):
                gpu_ctx.render_efficient(pm, pc, cap, R, delta, 40, 40, 1e-4, 1e-4)
            continue
        rgb, _ = gpu_ctx.render_efficient(pm, pc, cap, R, delta, 40, 40, 1e-4, 1e-4)
        a, e, s = gpu_ctx.samples(0)
        assert np.array_equal(common.bits(a), common.bits(want["a"])) and np.array_equal(common.bits(e), common.bits(want["e"]))
        assert np.array_equal(rgb, want_rgb), trial
        done += 1
        if done >= 12:
            break
    assert done >= 6


def test_efficient_speculation_is_transparent(gpu_ctx):
    """speculative subtree evaluation changes the number of launches, never the result or the bookkeeping."""
    sp, sn = common.make_skies(256, 128, "check")
    om, oc, pm, pc = common.scene("ellis", res=(64, 36))
    gpu_ctx.set_sky(0, curvis_amd.SphericalImage(sp))
    gpu_ctx.set_sky(1, curvis_amd.SphericalImage(sn))
    out = {}
    try:
        for depth in (0, 2, 4, 6):
            gpu_ctx.set_option("sampling_speculation", depth)
            rgb, st = gpu_ctx.render_efficient(pm, pc, 4096, 100.0, 0.05, 100, 100, 1e-5, 1e-5)
            info = gpu_ctx.sampling_info(0)
            out[depth] = (rgb, gpu_ctx.samples(0), (info.n_samples, info.calls, info.steps, info.rounds),
                          gpu_ctx.get_option("last_sampling_launches"), gpu_ctx.get_option("last_sampling_evaluated"))
    finally:
        gpu_ctx.set_option("sampling_speculation", -1)
    base = out[0]
    assert base[2][:3] == (678, 712, 1496307) and base[3] == base[2][3] + 1 and base[4] == 712
    for depth in (2, 4, 6):
        o = out[depth]
        assert np.array_equal(o[0], base[0]) and o[2] == base[2]
        for a, b in zip(o[1], base[1]):
            assert np.array_equal(common.bits(a), common.bits(b))
        assert o[3] < base[3] and o[4] > base[4]
    assert out[6][3] <= 5


@pytest.mark.parametrize("res", [(1, 1), (1, 9), (9, 1), (7, 9), (257, 3), (3, 257)])
def test_degenerate_frame_shapes(gpu_ctx, res):
    """one pixel, one row, one column, frames smaller than an 8x8 tile or a wave, widths that are no multiple of the
    tile: every kernel variant against the oracle, alone and as a three-frame batch (per-frame counters included)"""
    sp, sn = common.make_skies(128, 64, "check")
    om, oc, pm, pc = common.scene("ellis", res=res, pos=(0.0, 3.0, 1.3, 0.4), fwd=(-1.0, 0.2, 0.1))
    want, _, st = O.render_image(O.CV, om, oc, O.sky(sp), O.sky(sn), 3000, 100.0, 0.05)
    gpu_ctx.set_sky(0, curvis_amd.SphericalImage(sp))
    gpu_ctx.set_sky(1, curvis_amd.SphericalImage(sn))
    ref = (st.rays, st.steps, st.n_pos, st.n_neg, st.n_none, st.n_oob)
    try:
        for variant in (-1, 0, 1, 2):
            gpu_ctx.set_option("variant", variant)
            gpu_ctx.set_option("relay_min_blocks", 0)
            rgb, s = gpu_ctx.render_brute(pm, pc, 3000, 100.0, 0.05)
            assert rgb.shape == (res[1], res[0], 3) and np.array_equal(rgb, want), variant
            assert (s.rays, s.steps, s.n_pos, s.n_neg, s.n_none, s.n_oob) == ref, variant
            rgb3, s3 = gpu_ctx.render_brute(pm, [pc, pc, pc], 3000, 100.0, 0.05)
            per = gpu_ctx.frame_stats()
            assert all(np.array_equal(rgb3[k], want) for k in range(3)), variant
            assert [(f.rays, f.steps, f.n_pos, f.n_neg, f.n_none, f.n_oob) for f in per] == [ref] * 3, variant
        eff, _ = gpu_ctx.render_efficient(pm, pc, 3000, 100.0, 0.05, 40, 40, 1e-4, 1e-4)
        weff, _, _ = O.render_image_efficient(O.CV, om, oc, O.sky(sp), O.sky(sn), 3000, 100.0, 0.05, 40, 40, 1e-4, 1e-4)
        assert np.array_equal(eff, weff)
    finally:
        gpu_ctx.set_option("variant", -1)
        gpu_ctx.set_option("relay_min_blocks", -1)


def test_largest_frame_8k(gpu_ctx):
    """a frame of 7680x4320 (33.2 M rays, four times BASELINE's largest): every 64th row against the oracle, and the
    frame's counters against themselves across kernels (relay staging area 1.9 GB, framebuffer 100 MB)"""
    import os
    sp, sn = common.make_skies(2048, 1024, "check")
    om, oc, pm, pc = common.scene("ellis", res=(7680, 4320))
    gpu_ctx.set_sky(0, curvis_amd.SphericalImage(sp))
    gpu_ctx.set_sky(1, curvis_amd.SphericalImage(sn))
    rgb, s = gpu_ctx.render_brute(pm, pc, 4096, 100.0, 0.05)
    assert s.rays == 7680 * 4320 and s.n_pos + s.n_neg + s.n_none == s.rays
    from concurrent.futures import ThreadPoolExecutor
    T = common.host_threads(64)
    rows = list(range(5, 4320, 64))
    osp, osn = O.sky(sp), O.sky(sn)

    def work(i):   # rows i, i + T*64, ... of the 64-row comb
        mine = rows[i::T]
        out = {}
        for r in mine:
            img, _, _ = O.render_image(O.CV, om, oc, osp, osn, 4096, 100.0, 0.05, row_begin=r, row_step=1 << 30)
            out[r] = img[r].copy()
        return out
    with ThreadPoolExecutor(T) as ex:
        for part in ex.map(work, range(T)):
            for r, line in part.items():
                assert np.array_equal(rgb[r], line), r
    gpu_ctx.set_option("variant", 1)
    try:
        rgb1, s1 = gpu_ctx.render_brute(pm, pc, 4096, 100.0, 0.05)
    finally:
        gpu_ctx.set_option("variant", -1)
    assert np.array_equal(rgb1, rgb) and (s1.steps, s1.n_pos, s1.n_neg, s1.n_none) == (s.steps, s.n_pos, s.n_neg, s.n_none)


def test_relay_safety_net_and_verify_option(gpu_ctx):
    """the relay kernel leans on in-order workgroup dispatch (not promised by HIP): a launch that reports waves that
    gave up waiting is rendered again by the static kernel and the context stops using the relay kernel -- neither a
    hang nor a wrong frame (fault injected through "relay_test_fault"); "relay_verify" repeats every relay render with
    the static kernel and compares frames and per-frame counters."""
    sp, sn = common.make_skies(256, 128, "check")
    om, oc, pm, pc = common.scene("interstellar", res=(61, 35), pos=(0.0, 3.0, 1.2, 1.0), fwd=(-1.0, 0.1, 0.05))
    want, _, st = O.render_image(O.CV, om, oc, O.sky(sp), O.sky(sn), 3000, 100.0, 0.05)
    gpu_ctx.set_sky(0, curvis_amd.SphericalImage(sp))
    gpu_ctx.set_sky(1, curvis_amd.SphericalImage(sn))
    try:
        gpu_ctx.set_option("variant", 2)
        gpu_ctx.set_option("relay_min_blocks", 0)
        gpu_ctx.set_option("relay_segment", 64)
        gpu_ctx.set_option("relay_verify", 1)
        rgb, s = gpu_ctx.render_brute(pm, [pc, pc], 3000, 100.0, 0.05)
        assert np.array_equal(rgb[0], want) and np.array_equal(rgb[1], want) and s.steps == 2 * st.steps
        assert gpu_ctx.get_option("last_relay_launches") >= 1 and gpu_ctx.get_option("relay_fallbacks") == 0
        gpu_ctx.set_option("relay_verify", 0)
        before = gpu_ctx.get_option("relay_fallbacks")
        gpu_ctx.set_option("relay_test_fault", 1)
        rgb, s = gpu_ctx.render_brute(pm, pc, 3000, 100.0, 0.05)
        assert np.array_equal(rgb, want) and (s.rays, s.steps, s.n_pos, s.n_neg, s.n_none) == (st.rays, st.steps, st.n_pos, st.n_neg, st.n_none)
        assert gpu_ctx.get_option("relay_disabled") == 1 and gpu_ctx.get_option("relay_fallbacks") == before + 1
        rgb, _ = gpu_ctx.render_brute(pm, pc, 3000, 100.0, 0.05)          # stays on the static kernel
        assert np.array_equal(rgb, want) and gpu_ctx.get_option("last_relay_launches") == 0
    finally:
        for k, v in (("relay_disabled", 0), ("relay_verify", 0), ("variant", -1), ("relay_min_blocks", -1), ("relay_segment", 0)):
            gpu_ctx.set_option(k, v)


@pytest.mark.parametrize("metric,res,pos", [("ellis", (160, 90), None), ("interstellar", (96, 54), (0.0, -2.0, 1.1, 0.7)),
                                            ("flat", (61, 35), (0.0, 4.0, 1.3, 0.2))])
def test_direct_mode_bit_exact_vs_oracle(gpu_ctx, metric, res, pos):
    """curvis_render_direct ("direct" mode, not a reference function): compute_escape_angle for the alpha of every pixel,
    then step 5 -- pixels and counters bit-exact against the oracle's counterpart; fast and strict step; ragged
    frame sizes (tiles padded)."""
    sp, sn = common.make_skies(1024, 512, "smooth")
    kw = {} if pos is None else {"pos": pos}
    om, oc, pm, pc = common.scene(metric, res=res, **kw)
    want, st = O.render_image_direct(O.CV, om, oc, O.sky(sp), O.sky(sn), 4096, 100.0, 0.05)
    sys_ = curvis_amd.RelativisticSystem(pm, curvis_amd.SphericalImage(sp), curvis_amd.SphericalImage(sn), pc, context=gpu_ctx)
    try:
        for fast in (1, 0):
            gpu_ctx.set_option("fast_math", fast)
            got = sys_.render_image_direct(4096, 100.0, 0.05)
            s = sys_.last_stats
            assert np.array_equal(got, want), (metric, fast, int((got != want).any(axis=2).sum()))
            assert (s.rays, s.steps, s.n_pos, s.n_neg, s.n_none, s.n_oob) == (st.rays, st.steps, st.n_pos, st.n_neg, st.n_none, st.n_oob)
    finally:
        gpu_ctx.set_option("fast_math", 1)


def test_direct_mode_1080p_against_efficient(gpu_ctx):
    """a 1080p frame: the direct image and the efficient image (the reference's CLI renderer) differ only where the
    interpolation error moves a texel; the cap binds nowhere, so every ray escapes"""
    sp, sn = common.make_skies(2048, 1024, "smooth")
    _, _, pm, pc = common.scene("ellis", res=(1920, 1080))
    sys_ = curvis_amd.RelativisticSystem(pm, curvis_amd.SphericalImage(sp), curvis_amd.SphericalImage(sn), pc, context=gpu_ctx)
    d = sys_.render_image_direct(40000, 100.0, 0.05)
    sd = sys_.last_stats
    e = sys_.render_image_efficient(40000, 100.0, 0.05, 100, 100, 1e-5, 1e-5)
    diff = np.abs(d.astype(int) - e.astype(int)).max(axis=2)
    print("direct vs efficient 1080p: identical %.5f, <= 1 LSB %.5f, max %d; direct kernel %.2f ms, %d steps" % (
        (diff == 0).mean(), (diff <= 1).mean(), diff.max(), sd.kernel_ms, sd.steps))
    assert sd.rays == 1920 * 1080 and sd.n_none == 0 and (diff <= 1).mean() > 0.99


def test_render_into_page_locked_host_buffer(gpu_ctx):
    """curvis_host_alloc: frames rendered straight into page-locked host memory (what `curvis video` hands to its PNG
    writers) equal the frames returned through ordinary pageable memory"""
    sp, sn = common.make_skies(256, 128, "check")
    _, _, pm, pc = common.scene("ellis", res=(160, 90))
    gpu_ctx.set_sky(0, curvis_amd.SphericalImage(sp))
    gpu_ctx.set_sky(1, curvis_amd.SphericalImage(sn))
    want, s0 = gpu_ctx.render_brute(pm, [pc, pc, pc], 4096, 100.0, 0.05)
    buf = curvis_amd.HostBuffer(3 * 160 * 90 * 3 + 64)
    try:
        buf.array[:] = 7
        got, s1 = gpu_ctx.render_brute(pm, [pc, pc, pc], 4096, 100.0, 0.05, out=buf.array)
        assert np.array_equal(got, want) and s1.steps == s0.steps
        assert got.ctypes.data == buf.array.ctypes.data and (buf.array[-64:] == 7).all()   # in place, nothing beyond the frames
        with pytest.raises(ValueError):
            gpu_ctx.render_brute(pm, [pc, pc, pc], 4096, 100.0, 0.05, out=buf.array[:100])
    finally:
        del got
        buf.close()


def test_relay_seat_belt_checks_the_first_launch_of_every_shape(capfd):
    """The relay kernel's hand-over is argued from gfx950 facts, not from the HIP memory model, so the first
    relay launch of every launch shape of a context -- and every 1024th after it -- is repeated by the static kernel and
    compared.  Clean launches: one check per shape, the next one 1024 launches later.  A hand-over that delivers a wrong state (hook "relay_test_corrupt": every relay wave
    of that launch perturbs the tile it reloads) is caught: the frame returned is the static kernel's,
    the mismatch is counted and reported, and the context stays on the static kernel."""
    sp, sn = common.make_skies(2048, 1024, "smooth")   # smooth: every change of direction shows (a 64-texel checker cell would hide it)
    om, oc, pm, pc = common.scene("ellis", res=(480, 270))
    want, _, st = O.render_image(O.CV, om, oc, O.sky(sp), O.sky(sn), 4096, 100.0, 0.05)
    ctx = curvis_amd.Context(0)
    try:
        ctx.set_sky(0, curvis_amd.SphericalImage(sp))
        ctx.set_sky(1, curvis_amd.SphericalImage(sn))
        ctx.set_option("variant", 2)
        ctx.set_option("relay_min_blocks", 0)
        ctx.set_option("relay_segment", 64)   # a hand-over point every 64 steps: every launch below passes tiles on
        assert ctx.get_option("relay_auto_verify") == 1 and ctx.get_option("relay_verified_shapes") == 0
        rgb, s = ctx.render_brute(pm, pc, 4096, 100.0, 0.05)
        assert np.array_equal(rgb, want) and s.steps == st.steps
        assert ctx.get_option("last_relay_launches") >= 1 and ctx.get_option("last_relay_parks") > 0   # statistics are the relay launch's
        assert ctx.get_option("relay_verified_shapes") == 1 and ctx.get_option("relay_mismatches") == 0
        rgb, _ = ctx.render_brute(pm, pc, 4096, 100.0, 0.05)
        assert np.array_equal(rgb, want) and ctx.get_option("relay_verified_shapes") == 1               # checked once per shape
        rgb2, s2 = ctx.render_brute(pm, [pc, pc], 4096, 100.0, 0.05)                                     # another shape: checked again
        assert np.array_equal(rgb2[0], want) and np.array_equal(rgb2[1], want) and s2.steps == 2 * st.steps
        assert ctx.get_option("relay_verified_shapes") == 2 and ctx.get_option("relay_disabled") == 0
        # the check comes back every `relay_recheck_every`-th relay launch of a shape (default 1024: ~0.1 % overhead), so a
        # hand-over that starts to fail LATER in a context's life is caught too: with 3, launches 0, 3, 6 of a shape are checked
        assert ctx.get_option("relay_recheck_every") == 1024
        ctx.set_option("relay_recheck_every", 3)
        base = ctx.get_option("relay_checks")
        assert base == 2                                                                                 # the two first launches above
        for k in range(5):                                                                               # launches 2..6 of the one-frame shape
            rgb, _ = ctx.render_brute(pm, pc, 4096, 100.0, 0.05)
            assert np.array_equal(rgb, want)
        assert ctx.get_option("relay_checks") == base + 2 and ctx.get_option("relay_mismatches") == 0   # launches 3 and 6
        # a band of the frame or another step cap is another shape (they change the hand-over pattern)
        ctx.render_brute_rows(pm, pc, 0, 136, 4096, 100.0, 0.05)
        ctx.render_brute(pm, pc, 3000, 100.0, 0.05)
        assert ctx.get_option("relay_verified_shapes") == 4
        ctx.set_option("relay_recheck_every", 1024)
        # an unchecked launch with the hook shows that the hook does corrupt a frame ...
        ctx.set_option("relay_auto_verify", 0)
        ctx.set_option("relay_test_corrupt", 1)
        bad, _ = ctx.render_brute(pm, pc, 4096, 100.0, 0.05)
        assert ctx.get_option("last_relay_parks") > 0
        assert not np.array_equal(bad, want)
        # ... and with the seat belt on the same fault never reaches the caller
        ctx.set_option("relay_auto_verify", 1)                                                          # forgets the checked shapes
        ctx.set_option("relay_test_corrupt", 1)
        capfd.readouterr()
        rgb, s = ctx.render_brute(pm, pc, 4096, 100.0, 0.05)
        assert np.array_equal(rgb, want) and (s.rays, s.steps, s.n_pos, s.n_neg, s.n_none) == (st.rays, st.steps, st.n_pos, st.n_neg, st.n_none)
        assert ctx.get_option("relay_mismatches") == 1 and ctx.get_option("relay_disabled") == 1
        assert "differs from the static kernel" in capfd.readouterr().err
        rgb, _ = ctx.render_brute(pm, pc, 4096, 100.0, 0.05)
        assert np.array_equal(rgb, want) and ctx.get_option("last_relay_launches") == 0                  # static from now on
    finally:
        ctx.close()


def test_batch_framebuffer_beyond_4_gib(gpu_ctx):
    """One launch whose frames fill more than 2^32 bytes of framebuffer (100 frames of 5120x2880, 4.4 GB): frame
    offsets, pixel indices and the per-frame counters must be 64-bit clean.  The escape radius is pulled in to 6 so
    that rays take ~20-250 steps (the arithmetic is the same; the launch costs a fraction of a second).  Every
    frame is read back on its own (44 MB at a time) and must equal frame 0, whose comb of rows equals the oracle."""
    import ctypes as C
    W, H, N, CAP, R = 5120, 2880, 100, 400, 6.0
    assert W * H * 3 * N > 2 ** 32
    sp, sn = common.make_skies(1024, 512, "check")
    om, oc, pm, pc = common.scene("ellis", res=(W, H))
    gpu_ctx.set_sky(0, curvis_amd.SphericalImage(sp))
    gpu_ctx.set_sky(1, curvis_amd.SphericalImage(sn))
    _, st = gpu_ctx.render_brute(pm, [pc] * N, CAP, R, 0.05, download=False)
    per = gpu_ctx.frame_stats()
    assert len(per) == N and st.rays == W * H * N
    f0 = per[0]
    assert f0.rays == W * H and f0.n_pos + f0.n_neg + f0.n_none == W * H and f0.n_pos > 0 and f0.n_neg > 0
    for k, s in enumerate(per):
        assert (s.rays, s.steps, s.n_pos, s.n_neg, s.n_none, s.n_oob) == (f0.rays, f0.steps, f0.n_pos, f0.n_neg, f0.n_none, f0.n_oob), k
    assert st.steps == f0.steps * N
    dev, nbytes = gpu_ctx.framebuffer()
    frame_bytes = W * H * 3
    assert nbytes == frame_bytes * N
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.restype = C.c_int
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]

    def frame(k):
        out = np.empty(frame_bytes, np.uint8)
        assert hip.hipMemcpy(out.ctypes.data, C.c_void_p(dev + k * frame_bytes), frame_bytes, 2) == 0   # hipMemcpyDeviceToHost
        return out.reshape(H, W, 3)
    first = frame(0)
    osp, osn = O.sky(sp), O.sky(sn)
    want, _, _ = O.render_image(O.CV, om, oc, osp, osn, CAP, R, 0.05, row_begin=7, row_step=96)
    assert np.array_equal(first[7::96], want[7::96])
    assert len(np.unique(first[::16, ::16].reshape(-1, 3), axis=0)) > 50   # a picture, not a constant
    for k in (1, 2, 31, 32, 33, 50, 97, 98, 99):                       # around the 2^32-byte line (frame 32) and the ends
        assert np.array_equal(frame(k), first), k


def test_efficient_batch_framebuffer_beyond_4_gib(gpu_ctx):
    """the same for the efficient renderer's batch path (render_image_efficient x 100 frames of 5120x2880 in one call)"""
    import ctypes as C
    W, H, N = 5120, 2880, 100
    sp, sn = common.make_skies(1024, 512, "check")
    om, oc, pm, pc = common.scene("ellis", res=(W, H))
    gpu_ctx.set_sky(0, curvis_amd.SphericalImage(sp))
    gpu_ctx.set_sky(1, curvis_amd.SphericalImage(sn))
    gpu_ctx.render_efficient(pm, [pc] * N, 4096, 100.0, 0.05, 100, 100, 1e-5, 1e-5, download=False)
    dev, nbytes = gpu_ctx.framebuffer()
    frame_bytes = W * H * 3
    assert nbytes == frame_bytes * N and nbytes > 2 ** 32
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.restype = C.c_int
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]

    def frame(k):
        out = np.empty(frame_bytes, np.uint8)
        assert hip.hipMemcpy(out.ctypes.data, C.c_void_p(dev + k * frame_bytes), frame_bytes, 2) == 0
        return out.reshape(H, W, 3)
    first = frame(0)
    single, _ = gpu_ctx.render_efficient(pm, pc, 4096, 100.0, 0.05, 100, 100, 1e-5, 1e-5)
    assert np.array_equal(first, single)
    assert len(np.unique(first[::16, ::16].reshape(-1, 3), axis=0)) > 50
    gpu_ctx.render_efficient(pm, [pc] * N, 4096, 100.0, 0.05, 100, 100, 1e-5, 1e-5, download=False)
    dev, _ = gpu_ctx.framebuffer()
    for k in (1, 31, 32, 33, 99):
        assert np.array_equal(frame(k), first), k


def test_single_frame_of_half_a_billion_rays(gpu_ctx):
    """32768x16384 = 5.4e8 rays in ONE frame (1.6 GB of RGB8; tile and ray indices close to 2^32 / 8), escape radius
    pulled in to 6 to keep the launch short: counters consistent, a comb of rows read back from HBM equals the oracle.
    A frame whose padded ray count no longer fits 32 bits is refused, not rendered wrongly."""
    import ctypes as C
    W, H, CAP, R = 32768, 16384, 400, 6.0
    sp, sn = common.make_skies(1024, 512, "check")
    om, oc, pm, pc = common.scene("ellis", res=(W, H))
    gpu_ctx.set_sky(0, curvis_amd.SphericalImage(sp))
    gpu_ctx.set_sky(1, curvis_amd.SphericalImage(sn))
    _, st = gpu_ctx.render_brute(pm, pc, CAP, R, 0.05, download=False)
    assert st.rays == W * H and st.n_pos + st.n_neg + st.n_none == W * H and st.n_pos > 0 and st.n_neg > 0
    dev, nbytes = gpu_ctx.framebuffer()
    assert nbytes == W * H * 3
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.restype = C.c_int
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    rows = list(range(3, H, 1024)) + [H - 1]
    T = len(rows)
    osp, osn = O.sky(sp), O.sky(sn)
    from concurrent.futures import ThreadPoolExecutor

    def check_row(y):
        got = np.empty(W * 3, np.uint8)
        assert hip.hipMemcpy(got.ctypes.data, C.c_void_p(dev + y * W * 3), W * 3, 2) == 0
        want, _, _ = O.render_image(O.CV, om, oc, osp, osn, CAP, R, 0.05, row_begin=y, row_step=H)   # row y only
        return np.array_equal(got.reshape(W, 3), want[y])
    with ThreadPoolExecutor(common.host_threads(16)) as ex:
        assert all(ex.map(check_row, rows)), rows
    assert T == 17
    big = curvis_amd.Camera((0.0, 5.0, common.HALF_PI, 0.0), (-1.0, 0.0, 0.0), (0.0, 0.0, 1.0), 15.0, 43.0, 65536, 65536)
    with pytest.raises(curvis_amd.CurvisError):
        gpu_ctx.render_brute(pm, big, CAP, R, 0.05, download=False)


def _efficient_call(ctx, pm, cams, cap, n0, maxit, t1, t2):
    rgb, st = ctx.render_efficient(pm, cams, cap, 100.0, 0.05, n0, maxit, t1, t2)
    per = []
    for f in range(len(cams)):
        si = ctx.sampling_info(f)
        a, e, s = ctx.samples(f)
        fs = ctx.frame_stats(f)
        per.append((si.n_samples, si.rounds, si.calls, si.steps, si.warned_max_iterations, a.tobytes(), e.tobytes(), s.tobytes(),
                    fs.rays, fs.steps, fs.n_pos, fs.n_neg, fs.n_none, fs.n_oob))
    return np.asarray(rgb).tobytes(), (st.rays, st.steps, st.n_pos, st.n_neg, st.n_none, st.n_oob), per


def test_device_resident_sampler_equals_the_host_paced_sampler(gpu_ctx):
    """sampler_kernel (one workgroup runs the reference's whole adaptive sampler in LDS: kernels_efficient.h, cv_sampler_dev.h)
    against the host-paced sampler with speculation (cv_sampler.h, efficient_host.h): frames, sample tables (every alpha, escape
    angle and escape space, bit for bit), rounds, integrator calls and steps as the reference would count them per frame, per-frame
    pixel statistics.  Batches mix camera radii -- equal radii share a job on the device --, both metrics, both sides of the throat,
    a camera inside the Interstellar throat, few initial points, a cap that leaves not-escaped (NaN) samples to be cleaned out,
    max_iterations 0 / 1 (the warned path), a refine-nothing threshold, the strict (non-fast) Euler step."""
    sp, sn = common.make_skies(512, 256, "check")
    gpu_ctx.set_sky(0, curvis_amd.SphericalImage(sp))
    gpu_ctx.set_sky(1, curvis_amd.SphericalImage(sn))
    res = (96, 54)

    def cams_of(ls):
        return [curvis_amd.Camera((0.0, l, common.HALF_PI + 0.05 * k, 0.3 * k), (-1.0 if l > 0 else 1.0, 0.1 * k, 0.02 * k), (0.0, 0.0, 1.0), 15.0, 43.0,
                                  res[0], res[1]) for k, l in enumerate(ls)]
    ellis, inter = curvis_amd.EllisMetric(1.0), curvis_amd.InterstellarMetric(0.1, 1e-4, 1.0)
    cases = [(ellis, [5.0, 3.0, 3.0, 5.0, -2.5, 3.0, 0.3, 40.0, 5.0], 4096, 100, 100, 1e-5, 1e-5, 1),
             (inter, [5.0, -4.0, 0.01, 0.00005, 2.0, 2.0, -0.5], 8192, 100, 100, 1e-5, 2e-5, 1),
             (ellis, [5.0, 3.0], 4096, 3, 50, 1e-5, 1e-5, 1), (ellis, [5.0, 1.0, 7.0], 2000, 100, 100, 1e-5, 1e-5, 1),
             (ellis, [5.0, 2.0], 4096, 100, 0, 1e-5, 1e-5, 1), (ellis, [5.0, 2.0], 4096, 100, 1, 1e-5, 1e-5, 1),
             (ellis, [5.0, 2.0], 4096, 100, 100, 10.0, 10.0, 1), (inter, [3.0, -3.0, 0.2], 8192, 60, 100, 1e-5, 1e-5, 0),
             (curvis_amd.FlatSphericalMetric(), [5.0, 2.0], 4096, 100, 100, 1e-2, 1e-2, 1)]
    try:
        for pm, ls, cap, n0, maxit, t1, t2, fast in cases:
            gpu_ctx.set_option("fast_math", fast)
            cams = cams_of(ls)
            gpu_ctx.set_option("device_sampler", 0)
            host = _efficient_call(gpu_ctx, pm, cams, cap, n0, maxit, t1, t2)
            assert gpu_ctx.get_option("last_sampler_path") == 0
            gpu_ctx.set_option("device_sampler", 1)
            dev = _efficient_call(gpu_ctx, pm, cams, cap, n0, maxit, t1, t2)
            assert gpu_ctx.get_option("last_sampler_path") == 1 and gpu_ctx.get_option("last_sampling_launches") == 1
            assert dev[1] == host[1], (ls, dev[1], host[1])
            for f, (d, h) in enumerate(zip(dev[2], host[2])):
                assert d[:5] == h[:5], ("sampling info", ls, f, d[:5], h[:5])
                assert d[5:8] == h[5:8], ("sample table", ls, f)
                assert d[8:] == h[8:], ("frame statistics", ls, f, d[8:], h[8:])
            assert dev[0] == host[0], ls
        # a table that outgrows the kernel's fixed arrays (everything refined): the call falls back to the host-paced sampler
        cams = cams_of([5.0, 3.0])
        gpu_ctx.set_option("device_sampler", 0)
        host = _efficient_call(gpu_ctx, ellis, cams, 4096, 7, 100, 1e-9, 1e-9)
        gpu_ctx.set_option("device_sampler", 1)
        dev = _efficient_call(gpu_ctx, ellis, cams, 4096, 7, 100, 1e-9, 1e-9)
        assert gpu_ctx.get_option("last_sampler_path") == 2 and dev == host and host[2][0][0] > 1536
        # the reference's panic (fewer than three finite samples: nothing escapes within 10 steps) is the same error on both paths
        for flag in (0, 1):
            gpu_ctx.set_option("device_sampler", flag)
            with pytest.raises(curvis_amd.CurvisError) as e:
                gpu_ctx.render_efficient(ellis, cams, 10, 100.0, 0.05, 100, 100, 1e-5, 1e-5)
            assert e.value.code == curvis_amd._abi.E_SAMPLING
        # automatic choice: the device from device_sampler_min_frames frames on
        gpu_ctx.set_option("device_sampler", -1)
        gpu_ctx.render_efficient(ellis, cams_of([5.0] * 3), 4096, 100.0, 0.05, 100, 100, 1e-5, 1e-5)
        assert gpu_ctx.get_option("last_sampler_path") == 0
        gpu_ctx.render_efficient(ellis, cams_of([5.0] * gpu_ctx.get_option("device_sampler_min_frames")), 4096, 100.0, 0.05, 100, 100, 1e-5, 1e-5)
        assert gpu_ctx.get_option("last_sampler_path") == 1
    finally:
        gpu_ctx.set_option("device_sampler", -1)
        gpu_ctx.set_option("fast_math", 1)


def test_prefetched_sampler_gives_the_same_call(gpu_ctx):
    """curvis_ctx_prefetch_efficient: the sampler of a future render call launched ahead of time on its own stream.  The call that
    matches (metric, settings, the l of every frame) consumes the prefetched tables -- same frames, tables, counts as a call that
    samples itself --; one that does not match samples itself; batches in flight take turns in two slots."""
    sp, sn = common.make_skies(512, 256, "check")
    gpu_ctx.set_sky(0, curvis_amd.SphericalImage(sp))
    gpu_ctx.set_sky(1, curvis_amd.SphericalImage(sn))
    res = (96, 54)
    inter = curvis_amd.InterstellarMetric(0.1, 1e-4, 1.0)

    def batch(k):
        return [curvis_amd.Camera((0.0, -3.0 + 0.37 * (4 * k + j), common.HALF_PI, 0.2 * j), (1.0, 0.1, 0.0), (0.0, 0.0, 1.0), 15.0, 43.0, res[0], res[1])
                for j in range(4)]
    args = (8192, 100.0, 0.05, 100, 100, 1e-5, 1e-5)
    gpu_ctx.set_option("device_sampler", 1)
    try:
        plain = [_efficient_call(gpu_ctx, inter, batch(k), 8192, 100, 100, 1e-5, 1e-5) for k in range(3)]
        assert gpu_ctx.get_option("last_sampling_prefetched") == 0
        hits0 = gpu_ctx.get_option("prefetch_hits")
        # the loop of a video worker: the next batch's sampler is in flight while this batch renders
        gpu_ctx.prefetch_efficient(inter, batch(0), *args)
        got = []
        for k in range(3):
            if k + 1 < 3:
                gpu_ctx.prefetch_efficient(inter, batch(k + 1), *args)
            got.append(_efficient_call(gpu_ctx, inter, batch(k), 8192, 100, 100, 1e-5, 1e-5))
            assert gpu_ctx.get_option("last_sampling_prefetched") == 1 and gpu_ctx.get_option("last_sampler_path") == 1
        assert gpu_ctx.get_option("prefetch_hits") == hits0 + 3 and got == plain
        # a prefetch for other cameras / other settings is not taken; the call samples itself and is still right
        gpu_ctx.prefetch_efficient(inter, batch(1), *args)
        assert _efficient_call(gpu_ctx, inter, batch(0), 8192, 100, 100, 1e-5, 1e-5) == plain[0] and gpu_ctx.get_option("last_sampling_prefetched") == 0
        gpu_ctx.prefetch_efficient(inter, batch(2), 8192, 100.0, 0.05, 100, 100, 1e-5, 2e-5)
        assert _efficient_call(gpu_ctx, inter, batch(2), 8192, 100, 100, 1e-5, 1e-5) == plain[2] and gpu_ctx.get_option("last_sampling_prefetched") == 0
        # ... and the still-pending prefetch of batch 1 (two slots) is taken by its call
        gpu_ctx.prefetch_efficient(inter, batch(1), *args)
        assert _efficient_call(gpu_ctx, inter, batch(1), 8192, 100, 100, 1e-5, 1e-5) == plain[1] and gpu_ctx.get_option("last_sampling_prefetched") == 1
        # the tables of a call stay readable for ONE more submission into the other slot; the second one takes their slot: an error, not stale data
        gpu_ctx.render_efficient(inter, batch(0), *args)
        gpu_ctx.prefetch_efficient(inter, batch(1), *args)
        assert gpu_ctx.samples(1)[0].tobytes() == plain[0][2][1][5]
        gpu_ctx.prefetch_efficient(inter, batch(2), *args)
        assert gpu_ctx.samples(1)[0].tobytes() == plain[0][2][1][5]          # (already fetched: kept)
        with pytest.raises(curvis_amd.CurvisError) as e:
            gpu_ctx.samples(2)
        assert "prefetch" in str(e.value)
        # prefetches nobody consumes, then calls that sample themselves into the same slots (other stream): still right
        for k in range(3):
            assert _efficient_call(gpu_ctx, inter, [batch(k)[0]] * 4, 8192, 100, 100, 1e-5, 1e-5)[2][0][5:8] == plain[k][2][0][5:8]
            gpu_ctx.prefetch_efficient(inter, batch((k + 1) % 3), 8192, 100.0, 0.05, 100, 100, 1e-5, 3e-5)
        # a context destroyed with a prefetch in flight
        c2 = curvis_amd.Context(0)
        c2.set_sky(0, curvis_amd.SphericalImage(sp))
        c2.set_sky(1, curvis_amd.SphericalImage(sn))
        c2.prefetch_efficient(inter, batch(0), *args)
        c2.close()
    finally:
        gpu_ctx.set_option("device_sampler", -1)
