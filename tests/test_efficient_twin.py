"""Efficient renderer (the CLI's variant): host sampler (cv_sampler.h) + per-ray / per-pixel functions
(cv_efficient.h) compiled for x86 against the oracle's restatement of src/systems.rs:333-527,
src/sampling.rs and interp 1.0.3 -- sample tables and pixels bit for bit.  CPU only."""
import numpy as np
import pytest

import common
import oracle_lib as O

CASES = [
    ("ellis", (0.0, 5.0, common.HALF_PI, 0.0), (-1.0, 0.0, 0.0), 4096, 100),   # KAT-5 configuration
    ("ellis", (0.0, 3.0, common.HALF_PI, 0.9), (-1.0, 0.0, 0.0), 4096, 100),   # path_orbit pose
    ("interstellar", (0.0, 5.0, common.HALF_PI, 0.0), (-1.0, 0.0, 0.0), 4096, 100),
    ("ellis", (0.0, -2.5, 1.1, 2.0), (1.0, 0.3, 0.1), 3000, 60),               # -l side, tilted camera
]


@pytest.mark.parametrize("dev_sampler", [0, 1, 2], ids=["host-sampler", "device-sampler-control-flow", "device-sampler-with-speculation"])
@pytest.mark.parametrize("fast", [0, 1])
@pytest.mark.parametrize("metric,pos,fwd,cap,n0", CASES)
def test_twin_efficient_equals_oracle(metric, pos, fwd, cap, n0, fast, dev_sampler):
    sp, sn = common.make_skies(256, 128, "check")
    om, oc, pm, pc = common.scene(metric, res=(48, 27), pos=pos, fwd=fwd)
    # the reference wires max_iterations_sampling to sampling_initial_nums (src/main.rs:47,107)
    want_rgb, want, st = O.render_image_efficient(O.CV, om, oc, O.sky(sp), O.sky(sn), cap, 100.0, 0.05, n0, n0, 1e-5, 1e-5)
    got_rgb, got = common.twin_render_efficient(pm, pc, sp, sn, cap, 100.0, 0.05, n0, n0, 1e-5, 1e-5, fast=fast, dev_sampler=dev_sampler)
    assert got["calls"] == want["calls"] and got["steps"] == want["steps"]
    for k in ("a", "e", "s"):
        assert np.array_equal(common.bits(got[k]), common.bits(want[k])), k
    assert np.array_equal(got_rgb, want_rgb)


def test_kat5_counts_through_the_batched_sampler():
    sp, sn = common.make_skies(64, 32, "check")
    om, oc, pm, pc = common.scene("ellis", res=(8, 6))
    _, got = common.twin_render_efficient(pm, pc, sp, sn, 4096, 100.0, 0.05, 100, 100, 1e-5, 1e-5, fast=1)
    assert len(got["a"]) == 678 and got["calls"] == 712 and got["steps"] == 1496307


def test_nan_direction_maps_to_texel_zero():
    """A NaN rotation (axis == 0 -> 0/0, src/systems.rs:504) reaches the sky lookup as a NaN direction and
    lands on texel (0, 0) through `NaN as u32 == 0` (src/images.rs:118-119): checked on the lookup itself."""
    import ctypes as C
    img = np.arange(64 * 32 * 4, dtype=np.uint32).astype(np.uint8).reshape(32, 64, 4)
    s = O.sky(np.ascontiguousarray(img))
    px = (C.c_uint8 * 4)()
    O.lib().cvo_sky_pixel(O.CV, C.byref(s), O._dp(O.vec(np.nan, np.nan, np.nan)), px)
    assert list(px) == list(img[0, 0])


def test_device_sampler_control_flow_equals_the_host_sampler_on_hard_settings():
    """cv_sampler_dev.h (plan / store / consume over fixed arrays: what lane 0 of sampler_kernel runs) against cv_sampler.h on
    settings that stress the control flow: few initial points, thresholds that refine everywhere / nowhere, a cap that leaves
    NaN samples to be cleaned out (not escaped), max_iterations 0 / 1 / 3 (the warned path), a camera deep in the throat"""
    sp, sn = common.make_skies(64, 32, "check")
    cases = [("ellis", 5.0, 4096, 100, 100, 1e-5, 1e-5), ("ellis", 5.0, 4096, 3, 50, 1e-5, 1e-5), ("ellis", 5.0, 4096, 7, 100, 1e-9, 1e-9),
             ("ellis", 5.0, 4096, 100, 100, 10.0, 10.0), ("ellis", 5.0, 2000, 100, 100, 1e-5, 1e-5), ("ellis", 0.3, 4096, 100, 3, 1e-5, 1e-5),
             ("ellis", 5.0, 4096, 100, 0, 1e-5, 1e-5), ("ellis", 5.0, 4096, 100, 1, 1e-5, 1e-5), ("interstellar", 0.01, 8192, 100, 100, 1e-5, 2e-5),
             ("interstellar", -3.0, 8192, 50, 100, 1e-6, 1e-5), ("ellis", 40.0, 4096, 100, 100, 1e-5, 1e-5)]
    overflowed = 0
    for metric, l, cap, n0, maxit, t1, t2 in cases:
        _, _, pm, pc = common.scene(metric, res=(6, 4), pos=(0.0, l, common.HALF_PI, 0.3))
        res = []
        for dev in (0, 1, 2):
            try:
                rgb, got = common.twin_render_efficient(pm, pc, sp, sn, cap, 100.0, 0.05, n0, maxit, t1, t2, fast=1, dev_sampler=dev)
                res.append((rgb.tobytes(), got["calls"], got["steps"]) + tuple(got[k].tobytes() for k in "aes"))
            except RuntimeError as exc:
                res.append(str(exc))
        if res[1] == "twin efficient render failed: -4":   # the fixed arrays ran out (the product then falls back to the host sampler)
            assert len(res[0][3]) // 8 > 700 and res[2] == res[1], (metric, l, n0, t1)     # ... which may only happen to tables that really are large
            overflowed += 1
            continue
        assert res[0] == res[1] == res[2], (metric, l, cap, n0, maxit, t1, t2)
    assert overflowed == 1


def test_bucket_grid_lookup_equals_the_full_search():
    """cv_efficient.h interp_index_grid (what the per-pixel kernel uses) == interp_index (interp 1.0.3's prev_index, what the
    reference's interp_slice does per pixel, src/systems.rs:491-497 via src/interpolation.rs) for every query: tables spread over
    the sampled interval, clustered inside ONE bucket, reaching outside the interval, with two and three entries; queries on the
    abscissae themselves, one ulp either side of them, on bucket edges, outside the interval, infinities and NaN."""
    import ctypes as C
    lib = common.twin()
    lib.twin_interp_grid_check.restype = C.c_size_t
    lib.twin_interp_grid_check.argtypes = [C.c_void_p, C.c_uint, C.c_void_p, C.c_size_t, C.c_void_p]
    rng = np.random.default_rng(77)
    a0, a1 = -0.1 * np.pi, 1.1 * np.pi
    edges = a0 + (a1 - a0) * np.arange(0, 1025) / 1024.0
    tables = []
    for n in (2, 3, 17, 100, 700, 1536):
        tables.append(np.sort(rng.uniform(a0, a1, n)))                                  # spread
        tables.append(np.sort(1.3 + rng.uniform(0.0, 1e-4, n)))                          # inside one bucket
        tables.append(np.sort(np.concatenate([rng.uniform(a0 - 1.0, a1 + 1.0, n - 2), [a0, a1]])))   # beyond both ends
        x = np.sort(rng.uniform(1.0, 1.2, n)); x[: n // 2] = np.sort(rng.uniform(a0, a0 + 1e-9, n // 2)); tables.append(np.sort(x))
    x = edges[100:140].copy(); tables.append(x)                                        # abscissae ON bucket edges
    x = np.nextafter(edges[100:140], 10.0); tables.append(x)
    for x in tables:
        x = np.unique(x)                                                                 # strictly increasing, as the sampler's tables are
        q = np.concatenate([x, np.nextafter(x, -10.0), np.nextafter(x, 10.0), edges, np.nextafter(edges, -10.0), np.nextafter(edges, 10.0),
                            rng.uniform(a0 - 0.5, a1 + 0.5, 20000), rng.uniform(x[0], x[-1], 20000),
                            [0.0, -0.0, np.pi, a0, a1, -1e300, 1e300, np.inf, -np.inf, np.nan, 5e-324]])
        G = np.zeros(1025, np.uint32)
        x = np.ascontiguousarray(x); q = np.ascontiguousarray(q)
        bad = lib.twin_interp_grid_check(x.ctypes.data, x.size, q.ctypes.data, q.size, G.ctypes.data)
        assert bad == 0, (x.size, bad)
        assert G[0] == 0 and G[1024] == x.size and np.all(np.diff(G.astype(np.int64)) >= 0)


def test_pixel_row_by_magic_multiplication_is_the_division():
    """efficient_pixel_kernel takes the pixel's row as the high 64 bits of index x (floor((2^64 - 1) / W) + 1) (efficient_host.h
    w_magic) instead of dividing: exact for every index below 2^32 and every width from 2 up"""
    rng = np.random.default_rng(5)
    widths = [2, 3, 4, 5, 7, 64, 96, 100, 255, 256, 257, 1000, 1080, 1920, 2160, 3840, 4096, 65535, 65536, 65537, 2 ** 31 - 1, 2 ** 31, 2 ** 32 - 1]
    widths += [int(w) for w in rng.integers(2, 2 ** 32, 200)]
    for W in widths:
        M = (2 ** 64 - 1) // W + 1
        assert M < 2 ** 64
        idx = [0, 1, W - 1, W, W + 1, 2 * W - 1, 2 * W, 2 ** 32 - 1, 2 ** 32 - W, (2 ** 32 - 1) // W * W, (2 ** 32 - 1) // W * W - 1]
        idx += [int(v) for v in rng.integers(0, 2 ** 32, 300)]
        for n in idx:
            if 0 <= n < 2 ** 32:
                assert (n * M) >> 64 == n // W, (W, n)
