"""Overlapped download (option "async_download", curvis_ctx_download_wait; render_host.h fb_download / fb_begin_write).

The reference's render_image returns an owned host image (src/systems.rs:314-329): a host that renders frame after frame
pays the PCIe copy behind every kernel.  With the option set the copy of call k runs under the kernels of call k + 1 (second
frame buffer, copy stream).  What must hold: every frame that arrives is byte for byte the frame the synchronous path
returns (and therefore the oracle's), the contract "complete when the next call returns / after download_wait" is kept,
statistics are those of the call, and everything that reads "the frames of the last render" -- curvis_ctx_download,
curvis_ctx_deflate_frames -- still sees the last render, whichever of the two device buffers it went to."""
import zlib

import numpy as np
import pytest

import common
import curvis_amd
import oracle_lib as O

pytestmark = pytest.mark.gpu

W, H = 320, 180
CAP, R, DELTA = 4096, 100.0, 0.05


def poses(n):
    """n cameras that give n different frames: radial positions from both sides of the throat, the view tilted a little more
    each time (a camera's phi alone would not do: the per-pixel path looks the sky up by the photon's final MOMENTUM
    direction, src/metrics.rs:339-349, so frames along the orbit of configs[3] are identical)"""
    out = []
    for k in range(n):
        l = (-1.0) ** k * (1.5 + 0.6 * k)
        out.append(((0.0, l, common.HALF_PI, 0.3 * k), (-1.0, 0.05 * k, 0.02 * k), (0.0, 0.0, 1.0)))
    return out


@pytest.fixture(scope="module")
def ctx():
    c = curvis_amd.Context(0)
    sp, sn = common.make_skies(1024, 512, "check")
    c.set_sky(0, curvis_amd.SphericalImage(sp))
    c.set_sky(1, curvis_amd.SphericalImage(sn))
    yield c, sp, sn
    c.close()


def cams_of(metric, n):
    res = []
    for pos, fwd, up in poses(n):
        _, _, pm, pc = common.scene(metric, (W, H), pos=pos, fwd=fwd, up=up)
        res.append((pm, pc))
    return res


@pytest.mark.parametrize("metric", ["ellis", "interstellar"])
def test_pipelined_frames_equal_synchronous_frames(ctx, metric):
    c, sp, sn = ctx
    n = 7
    scenes = cams_of(metric, n)
    want, want_st = [], []
    for pm, pc in scenes:                                             # synchronous reference (option off)
        rgb, st = c.render_brute(pm, pc, CAP, R, DELTA)
        want.append(rgb.copy())
        want_st.append((st.rays, st.steps, st.n_pos, st.n_neg, st.n_none))
    assert all(not np.array_equal(want[i], want[j]) for i in range(n) for j in range(i))   # the frames really differ
    bufs = [curvis_amd.HostBuffer(W * H * 3) for _ in range(2)]
    c.set_option("async_download", 1)
    try:
        before = c.get_option("downloads_overlapped")
        for k, (pm, pc) in enumerate(scenes):
            bufs[k % 2].array[:] = 0xAB                               # stale contents must not survive
            rgb, st = c.render_brute(pm, pc, CAP, R, DELTA, out=bufs[k % 2].array)
            assert (st.rays, st.steps, st.n_pos, st.n_neg, st.n_none) == want_st[k]   # statistics are this call's, valid on return
            assert c.get_option("download_pending") == 1
            if k:                                                     # contract: frame k - 1 is complete now
                assert np.array_equal(bufs[(k - 1) % 2].array.reshape(H, W, 3), want[k - 1]), k
        c.download_wait()
        assert c.get_option("download_pending") == 0
        assert np.array_equal(bufs[(n - 1) % 2].array.reshape(H, W, 3), want[n - 1])
        assert c.get_option("downloads_overlapped") == before + n
        c.download_wait()                                             # idempotent
    finally:
        c.set_option("async_download", 0)
        for b in bufs:
            b.close()


def test_last_render_stays_visible_to_download_and_deflate(ctx):
    """d_fb must always be the buffer of the LAST render: curvis_ctx_download and the PNG front end read it, while the
    pending copy of the previous call still drains the other buffer"""
    c, sp, sn = ctx
    scenes = cams_of("ellis", 5)
    sync = [c.render_brute(pm, pc, CAP, R, DELTA)[0].copy() for pm, pc in scenes]
    assert all(not np.array_equal(sync[i], sync[j]) for i in range(len(sync)) for j in range(i))
    bufs = [curvis_amd.HostBuffer(W * H * 3) for _ in range(2)]
    c.set_option("async_download", 1)
    try:
        ptrs = set()
        for k, (pm, pc) in enumerate(scenes):
            c.render_brute(pm, pc, CAP, R, DELTA, out=bufs[k % 2].array)
            ptrs.add(c.framebuffer()[0])
            assert np.array_equal(c.download_frames(W, H, 1)[0], sync[k]), k          # explicit download: the last render
            streams, _ = c.deflate_frames(W, H, 1)                                     # device PNG front end: the last render
            raw = np.frombuffer(zlib.decompress(streams[0]), np.uint8).reshape(H, W * 3 + 1)
            got = np.cumsum(raw[:, 1:].astype(np.uint32), axis=0).astype(np.uint8).reshape(H, W, 3)
            assert np.array_equal(got, sync[k]), k
        assert len(ptrs) == 2                                         # two device buffers took turns
        # a render WITHOUT an output buffer in between (frames stay in HBM) must not disturb the download in flight
        c.render_brute(scenes[0][0], scenes[0][1], CAP, R, DELTA, download=False)
        assert np.array_equal(c.download_frames(W, H, 1)[0], sync[0])
        c.download_wait()
        assert np.array_equal(bufs[(len(scenes) - 1) % 2].array.reshape(H, W, 3), sync[-1])
        # frames uploaded by the host go to the buffer that is not being drained, too
        c.render_brute(scenes[1][0], scenes[1][1], CAP, R, DELTA, out=bufs[0].array)
        c.upload_frames(sync[2][None])
        assert np.array_equal(c.download_frames(W, H, 1)[0], sync[2])
        c.download_wait()
        assert np.array_equal(bufs[0].array.reshape(H, W, 3), sync[1])
    finally:
        c.set_option("async_download", 0)
        for b in bufs:
            b.close()


def test_other_render_modes_batches_and_the_oracle(ctx):
    """efficient, direct, multi-frame batches and row bands share the mechanism; the frames are the oracle's"""
    c, sp, sn = ctx
    om, oc, pm, pc = common.scene("ellis", (W, H), pos=(0.0, 3.0, common.HALF_PI, 0.7))
    eff_sync, _ = c.render_efficient(pm, pc, CAP, R, DELTA, 100, 50, 1e-5, 1e-5)
    dir_sync, _ = c.render_direct(pm, pc, CAP, R, DELTA)
    scenes = cams_of("ellis", 3)
    batch_sync, _ = c.render_brute(scenes[0][0], [s[1] for s in scenes], CAP, R, DELTA)
    ref, _, _ = O.render_image(O.CV, om, oc, O.sky(sp), O.sky(sn), CAP, R, DELTA)   # oracle, same elementary functions
    b_eff, b_dir, b_one = (curvis_amd.HostBuffer(W * H * 3) for _ in range(3))
    b_batch = curvis_amd.HostBuffer(3 * W * H * 3)
    c.set_option("async_download", 1)
    try:
        c.render_efficient(pm, pc, CAP, R, DELTA, 100, 50, 1e-5, 1e-5, out=b_eff.array)
        c.render_direct(pm, pc, CAP, R, DELTA, out=b_dir.array)
        assert np.array_equal(b_eff.array.reshape(H, W, 3), eff_sync)               # complete: the next call has returned
        c.render_brute(scenes[0][0], [s[1] for s in scenes], CAP, R, DELTA, out=b_batch.array)
        assert np.array_equal(b_dir.array.reshape(H, W, 3), dir_sync)
        c.render_brute(pm, pc, CAP, R, DELTA, out=b_one.array)
        assert np.array_equal(b_batch.array.reshape(3, H, W, 3), batch_sync)
        c.set_option("async_download", 0)                              # switching the option off waits
        assert c.get_option("download_pending") == 0
        assert np.array_equal(b_one.array.reshape(H, W, 3), ref)
    finally:
        c.set_option("async_download", 0)
        for b in (b_eff, b_dir, b_one, b_batch):
            b.close()


def test_context_destroyed_with_a_download_in_flight():
    c = curvis_amd.Context(0)
    sp, sn = common.make_skies(512, 256, "check")
    c.set_sky(0, curvis_amd.SphericalImage(sp))
    c.set_sky(1, curvis_amd.SphericalImage(sn))
    _, _, pm, pc = common.scene("ellis", (W, H))
    want, _ = c.render_brute(pm, pc, CAP, R, DELTA)
    buf = curvis_amd.HostBuffer(W * H * 3)
    c.set_option("async_download", 1)
    c.render_brute(pm, pc, CAP, R, DELTA, out=buf.array)
    c.close()                                                          # waits for the copy engine before freeing the frame buffers
    assert np.array_equal(buf.array.reshape(H, W, 3), want)
    buf.close()


def test_wrapper_arrays_are_complete_and_kept_alive(ctx):
    """with the option on, a render call WITHOUT a caller's buffer hands back an array the wrapper allocated itself
    (pageable): it must be complete on return -- nobody calls download_wait for an array they were just handed -- and a caller's
    buffer whose only other reference is dropped must stay alive (the context keeps it) until the copy has been waited for."""
    import gc
    import weakref
    c, sp, sn = ctx
    _, _, pm, pc = common.scene("ellis", (W, H))
    want, _ = c.render_brute(pm, pc, CAP, R, DELTA)
    c.set_option("async_download", 1)
    try:
        for _ in range(3):
            got, _ = c.render_brute(pm, pc, CAP, R, DELTA)                 # no `out`: waited for inside the wrapper
            assert c.get_option("download_pending") == 0 and np.array_equal(got, want)
            eff, _ = c.render_efficient(pm, pc, CAP, R, DELTA, 100, 50, 1e-5, 1e-5)
            assert c.get_option("download_pending") == 0 and eff.shape == want.shape
        buf = np.empty(W * H * 3, np.uint8)                               # a caller's (pageable) buffer, dropped right after the call
        ref = weakref.ref(buf)
        c.render_brute(pm, pc, CAP, R, DELTA, out=buf)
        del buf
        gc.collect()
        assert ref() is not None                                           # the context holds it while the copy may be in flight
        kept = ref()
        c.download_wait()
        assert np.array_equal(kept.reshape(H, W, 3), want)
        del kept
        gc.collect()
        assert ref() is None                                               # ... and lets go afterwards
    finally:
        c.set_option("async_download", 0)
