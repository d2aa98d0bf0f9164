"""PNG front end on the device (curvis_ctx_deflate_frames, kernels_png.h): the frames a render call left in HBM come back as
finished zlib streams; wrapped into PNG files they must decode -- with Python's zlib (which also checks the Adler-32 the
device computed), with this repository's two PNG decoders -- to exactly the pixels the same render call downloads.
Reference seam: `image::DynamicImage::save` after every frame (src/rendering.rs:110, :311)."""
import ctypes as C
import os
import zlib

import numpy as np
import pytest

import common
import curvis_amd
from curvis_amd import _abi, pngio

pytestmark = pytest.mark.gpu


def unfilter_up(raw, w, h):
    rows = np.frombuffer(raw, np.uint8).reshape(h, w * 3 + 1)
    assert (rows[:, 0] == 2).all()                                   # filter type Up on every row
    return np.cumsum(rows[:, 1:].astype(np.uint32), axis=0).astype(np.uint8).reshape(h, w, 3)   # Up: running sum mod 256


def product_decode(path):
    p, w, h = C.POINTER(C.c_uint8)(), C.c_uint32(), C.c_uint32()
    _abi.check(_abi.lib().curvis_image_load(str(path).encode(), C.byref(p), C.byref(w), C.byref(h)))
    try:
        return np.ctypeslib.as_array(p, shape=(h.value, w.value, 4)).copy()
    finally:
        _abi.lib().curvis_image_free(p)


_LEN_BASE = [3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258]
_LEN_EXTRA = [0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0]
_CL_ORDER = [16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15]


def deflate_tokens(z, lengths_out=None):
    """The TOKENS of a zlib stream holding one final dynamic-Huffman block (RFC 1951), read by a decoder written for this test:
    literal v -> v, match of length L at distance 1 -> 1000 + L (any other distance fails).  The independent checker of WHAT the
    device emitted, not only of what it decodes to (until round 5 the three-pass kernels played that part)."""
    data = int.from_bytes(z[2:-4], "little")
    pos = 0

    def bits(n):
        nonlocal pos
        v = (data >> pos) & ((1 << n) - 1)
        pos += n
        return v

    def table(lengths):
        code, out = 0, {}
        for ln in range(1, 16):
            for sym, l in enumerate(lengths):
                if l == ln:
                    out[(ln, code)] = sym
                    code += 1
            code <<= 1
        return out

    def symbol(t):
        code = 0
        for ln in range(1, 16):
            code = (code << 1) | bits(1)
            if (ln, code) in t:
                return t[(ln, code)]
        raise AssertionError("bad code")
    assert z[:2] == b"\x78\x01" and bits(1) == 1 and bits(2) == 2          # final block, dynamic Huffman
    hlit, hdist, hclen = bits(5) + 257, bits(5) + 1, bits(4) + 4
    cl = [0] * 19
    for k in range(hclen):
        cl[_CL_ORDER[k]] = bits(3)
    clt, lens = table(cl), []
    while len(lens) < hlit + hdist:
        sy = symbol(clt)
        if sy < 16:
            lens.append(sy)
        elif sy == 16:
            lens += [lens[-1]] * (3 + bits(2))
        elif sy == 17:
            lens += [0] * (3 + bits(3))
        else:
            lens += [0] * (11 + bits(7))
    lt, dt = table(lens[:hlit]), table(lens[hlit:])
    if lengths_out is not None:
        lengths_out.extend(lens[:hlit])
    toks = []
    while True:
        sy = symbol(lt)
        if sy < 256:
            toks.append(sy)
        elif sy == 256:
            break
        else:
            ln = _LEN_BASE[sy - 257] + bits(_LEN_EXTRA[sy - 257])
            assert symbol(dt) == 0, "only distance 1 is ever emitted"
            toks.append(1000 + ln)
    assert (len(z) - 6) * 8 - pos < 8, "the stream ends with the end-of-block code"
    return toks


def assert_code_is_the_cheapest(toks, lens, what):
    """the code in the block header is what png_codes_kernel is meant to build: the cheapest one of at most 12 bits for the
    stream's own token counts + 1 (+ 1 more for the end-of-block symbol) -- compared with the host's construction (png_codes.h
    through the x86 twin; tests/test_png_stream_host.py checks THAT against an independent package-merge).  Ties may be broken
    differently; the cost cannot differ."""
    import ctypes as C
    lib = common.twin()
    lib.twin_huffman_lengths.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    lib.twin_huffman_lengths.restype = None
    freq = np.ones(286, np.uint32)
    freq[256] += 1
    for t in toks:
        if t < 1000:
            freq[t] += 1
        else:
            ln = t - 1000
            freq[257 + max(i for i in range(29) if _LEN_BASE[i] <= ln)] += 1
    best = np.zeros(286, np.uint8)
    lib.twin_huffman_lengths(freq.ctypes.data, 286, 12, best.ctypes.data)
    got = np.array(lens[:286], np.int64)
    assert got.size == 286 and got.min() >= 1 and got.max() <= 12, what
    assert int((freq * got).sum()) == int((freq.astype(np.int64) * best).sum()), what


def model_tokens(frame):
    """the stream format of kernels_png.h, restated: filter Up on every row (type byte 2 leads the row), every 64 image bytes of a
    row tokenised on their own -- a run of L zeros is L literals when L <= 3, else one literal zero and a distance-1 match of L - 1"""
    h, w, _ = frame.shape
    rows = frame.reshape(h, w * 3).astype(np.int16)
    filt = rows.copy()
    filt[1:] -= rows[:-1]
    filt = (filt & 255).astype(np.uint8)
    toks = []
    for r in range(h):
        toks.append(2)
        for x0 in range(0, w * 3, 64):
            run = 0
            for v in filt[r, x0:x0 + 64].tolist() + [None]:
                if v == 0:
                    run += 1
                    continue
                if run:
                    toks += [0] * run if run <= 3 else [0, 1000 + run - 1]
                    run = 0
                if v is not None:
                    toks.append(v)
    return toks


def check_streams(ctx, tmp_path, frames, w, h, tag):
    streams, ms = ctx.deflate_frames(w, h, len(frames))
    assert ms > 0 and len(streams) == len(frames)
    for k, (z, want) in enumerate(zip(streams, frames)):
        assert z[:2] == b"\x78\x01"
        raw = zlib.decompress(z)                                      # raises on a wrong Adler-32 or a malformed block
        assert len(raw) == h * (w * 3 + 1)
        assert np.array_equal(unfilter_up(raw, w, h), want), (tag, k)
        path = tmp_path / ("%s_%d.png" % (tag, k))
        buf = np.frombuffer(z, np.uint8)
        _abi.check(_abi.lib().curvis_image_save_zlib_rgb8(str(path).encode(), buf.ctypes.data, buf.size, w, h))
        assert np.array_equal(pngio.read_png(path), want), (tag, k)              # decoder 1: tests' own (zlib + unfilter)
        got = product_decode(path)                                               # decoder 2: the product's (host/png_io.h)
        assert np.array_equal(got[..., :3], want) and (got[..., 3] == 255).all(), (tag, k)
    # WHAT was emitted, token by token, against the restated format (small frames: the checker is a Python loop per token).
    # One device path serves every width since round 6 (rows that are a multiple of 64 bytes are staged with coalesced loads,
    # ragged rows chunk by chunk); until then the three-pass kernels were compared bit for bit here.
    if w * h * 3 <= 160_000:
        for k, (z, want) in enumerate(zip(streams, frames)):
            lens = []
            toks = deflate_tokens(z, lens)
            assert toks == model_tokens(want), (tag, k)
            assert_code_is_the_cheapest(toks, lens, (tag, k))
    # the PNG chunk's CRC-32 ("IDAT" + stream) from the device: equal to zlib's over the same bytes, for every frame; the file
    # written with it is byte-identical to the one whose CRC the host computed
    s2, _, crcs = ctx.deflate_frames_crc(w, h, len(frames))
    assert s2 == streams
    assert crcs is not None and [zlib.crc32(b"IDAT" + z) for z in streams] == crcs, tag
    buf = np.frombuffer(streams[0], np.uint8)
    pa, pb = tmp_path / ("%s_crc_dev.png" % tag), tmp_path / ("%s_0.png" % tag)
    _abi.check(_abi.lib().curvis_image_save_zlib_rgb8_crc(str(pa).encode(), buf.ctypes.data, buf.size, w, h, crcs[0]))
    assert pa.read_bytes() == pb.read_bytes()
    return streams, ms


@pytest.mark.parametrize("metric,res", [("ellis", (96, 54)), ("interstellar", (64, 36)), ("ellis", (50, 31)), ("ellis", (7, 3)), ("ellis", (341, 17))])
def test_brute_frames_round_trip(gpu_ctx, tmp_path, metric, res):
    """rows that are a multiple of 64 bytes (staged: coalesced 16-byte loads), of 4 bytes (96, 64 pixels... 288-byte rows: a ragged last
    chunk read word by word) and of neither (50, 7, 341 pixels: byte loads, a last chunk of 22 / 21 / 63 bytes)"""
    sp, sn = common.make_skies(512, 256, "check")
    gpu_ctx.set_sky(0, curvis_amd.SphericalImage(sp))
    gpu_ctx.set_sky(1, curvis_amd.SphericalImage(sn))
    _, _, pm, pc = common.scene(metric, res=res)
    rgb, _ = gpu_ctx.render_brute(pm, pc, 4096, 100.0, 0.05)
    check_streams(gpu_ctx, tmp_path, [rgb], res[0], res[1], "brute")


def test_batch_of_frames_and_black_frame(gpu_ctx, tmp_path):
    """a multi-frame launch (every frame its own code and stream) and an all-black frame (cap 0: nothing but zero runs,
    cut into matches of 63 at the 64-byte chunk boundaries)"""
    sp, sn = common.make_skies(512, 256, "smooth")
    gpu_ctx.set_sky(0, curvis_amd.SphericalImage(sp))
    gpu_ctx.set_sky(1, curvis_amd.SphericalImage(sn))
    cams = []
    for k in range(5):
        _, _, pm, pc = common.scene("ellis", res=(128, 72), pos=(0.0, 3.0 + k, common.HALF_PI, 0.5 * k))
        cams.append(pc)
    rgb, _ = gpu_ctx.render_brute(pm, cams, 4096, 100.0, 0.05)
    streams, _ = check_streams(gpu_ctx, tmp_path, list(rgb), 128, 72, "batch")
    assert len(set(streams)) == 5
    black, _ = gpu_ctx.render_brute(pm, cams[0], 0, 100.0, 0.05)
    assert not black.any()
    (z,), _ = check_streams(gpu_ctx, tmp_path, [black], 128, 72, "black")
    assert len(z) < 128 * 72 * 3 // 30                                 # 72 rows x (1 literal + 6 x (literal + match)) + header


def test_frames_that_do_not_compress(gpu_ctx, tmp_path):
    """white-noise skies: every pixel differs from its neighbours, literals cost ~8 bits each; the emit pass's LDS image of a
    workgroup's piece of the stream holds the worst case (65 literals of 12 bits per thread) -- same decoded pixels; the stream is
    about as long as the raw frame"""
    rng = np.random.default_rng(11)
    skies_ = []
    for _ in range(2):
        t = rng.integers(0, 256, (2048, 4096, 4), dtype=np.uint8)      # ~3 texels per pixel: no two neighbours share one
        t[..., 3] = 255
        skies_.append(t)
    gpu_ctx.set_sky(0, curvis_amd.SphericalImage(skies_[0]))
    gpu_ctx.set_sky(1, curvis_amd.SphericalImage(skies_[1]))
    for res in ((256, 144), (200, 77)):            # staged (768-byte rows) and ragged (600-byte rows: word loads, a last chunk of 24 bytes)
        _, _, pm, pc = common.scene("ellis", res=res)
        rgb, _ = gpu_ctx.render_brute(pm, [pc, pc], 4096, 100.0, 0.05)
        streams, _ = check_streams(gpu_ctx, tmp_path, list(rgb), res[0], res[1], "noise%d" % res[0])
        ratios = [len(z) / (res[0] * res[1] * 3) for z in streams]
        assert all(r > 0.4 for r in ratios), ratios                      # measured 0.48: noise outside the throat, flat inside


def test_efficient_frames_and_full_hd(gpu_ctx, tmp_path):
    """what `curvis video` (default mode) saves: render_image_efficient frames, 1920x1080, left in HBM (download=False) and
    compressed there; the stream is a small fraction of the 6.2 MB frame and decodes to the frame a download gives"""
    sp, sn = common.make_skies(2048, 1024, "smooth")
    gpu_ctx.set_sky(0, curvis_amd.SphericalImage(sp))
    gpu_ctx.set_sky(1, curvis_amd.SphericalImage(sn))
    cams = []
    for k in range(3):
        _, _, pm, pc = common.scene("ellis", res=(1920, 1080), pos=(0.0, 4.0 + k, common.HALF_PI, 0.3 * k))
        cams.append(pc)
    want, _ = gpu_ctx.render_efficient(pm, cams, 4096, 100.0, 0.05, 100, 100, 1e-5, 1e-5)
    gpu_ctx.render_efficient(pm, cams, 4096, 100.0, 0.05, 100, 100, 1e-5, 1e-5, download=False)
    streams, ms = check_streams(gpu_ctx, tmp_path, list(want), 1920, 1080, "eff")
    assert all(len(z) < 1920 * 1080 * 3 // 4 for z in streams)
    print("device PNG front end: 3 x 1080p in %.3f ms, streams %s bytes" % (ms, [len(z) for z in streams]))


def test_errors(gpu_ctx):
    sp, sn = common.make_skies(64, 32, "check")
    gpu_ctx.set_sky(0, curvis_amd.SphericalImage(sp))
    gpu_ctx.set_sky(1, curvis_amd.SphericalImage(sn))
    _, _, pm, pc = common.scene("ellis", res=(32, 18))
    gpu_ctx.render_brute(pm, pc, 100, 100.0, 0.05)
    with pytest.raises(curvis_amd.CurvisError) as e:
        gpu_ctx.deflate_frames(32, 18, 2)                              # only one frame of that size is there
    assert e.value.code == _abi.E_INVALID
    with pytest.raises(curvis_amd.CurvisError) as e:
        gpu_ctx.deflate_frames(32, 18, 1, out=np.empty(64, np.uint8))
    assert e.value.code == _abi.E_INVALID and "too small" in str(e.value)


@pytest.mark.parametrize("w,h", [(64, 5), (192, 33), (320, 47), (1280, 9), (960, 235), (512, 143), (100, 41), (85, 23), (23, 90), (1, 7), (277, 3)])
def test_tokeniser_on_synthetic_contents(gpu_ctx, tmp_path, w, h):
    """the word-parallel tokeniser -- staged widths and ragged ones (300 / 255 / 69 / 3 / 831-byte rows: partial last chunks of 44,
    63, 5, 3 and 63 bytes, word and byte loads) -- on contents chosen to hit its cases: runs of every length at
    every alignment (1..70 zeros between non-zero bytes), bytes 1 / 255 / others in every position of a word, whole zero
    chunks, rows that differ from the row above in one byte only, white noise; frames put into the context's framebuffer
    with curvis_ctx_upload -- tokens identical to the restated format's (small frames), decoded pixels identical"""
    rng = np.random.default_rng(w * 1000 + h)
    frames = []
    n = w * h * 3
    f = np.zeros(n, np.uint8)                       # runs of growing length separated by single non-zero bytes (filter Up of row 0 = the row)
    i, run = 0, 1
    while i < n:
        f[i] = rng.choice([1, 255, 2, 128, 254, 77])
        i += 1 + run
        run = run % 70 + 1
    frames.append(f.reshape(h, w, 3))
    frames.append(np.cumsum(frames[0].astype(np.uint32), axis=0).astype(np.uint8))           # the same after the Up filter of every row
    frames.append(rng.choice(np.array([0, 0, 0, 0, 1, 255], np.uint8), n).reshape(h, w, 3))  # hot symbols, short runs
    frames.append(rng.integers(0, 256, n, dtype=np.uint8).reshape(h, w, 3))                  # noise: 12-bit literals, no runs
    g = np.zeros((h, w, 3), np.uint8)
    g[:, :, :] = (np.arange(w)[None, :, None] // 3).astype(np.uint8)                         # rows identical: all zero after row 0
    g[h // 2, w // 2, 1] += 1
    frames.append(g)
    frames.append(np.zeros((h, w, 3), np.uint8))
    batch = np.ascontiguousarray(np.stack(frames))
    # the scratch (stream buffers, counters) is re-used from call to call: leave the longest possible streams of OTHER
    # contents in it first -- a path that relies on memory being zero where it did not clear it shows up here
    gpu_ctx.upload_frames(rng.integers(0, 256, batch.shape, dtype=np.uint8))
    gpu_ctx.deflate_frames(w, h, len(frames))
    gpu_ctx.upload_frames(batch)
    assert np.array_equal(gpu_ctx.download_frames(w, h, len(frames)), batch)
    first, _ = check_streams(gpu_ctx, tmp_path, list(batch), w, h, "synthetic%d" % w)
    for dirt in (255, 0x55):                                            # ... and again over all-ones / alternating bits
        gpu_ctx.upload_frames(np.full(batch.shape, dirt, np.uint8) ^ rng.integers(0, 2, batch.shape, dtype=np.uint8))
        gpu_ctx.deflate_frames(w, h, len(frames))
        gpu_ctx.upload_frames(batch)
        again, _ = gpu_ctx.deflate_frames(w, h, len(frames))
        assert again == first


def test_async_streams_option(gpu_ctx):
    """option "async_streams": a deflate call returns with its streams still travelling to the caller's (page-locked) buffer -- offsets,
    Adler-32 trailers and chunk CRCs are final on return, the bytes after curvis_ctx_download_wait --; the next deflate call waits
    for them by itself before it reuses the scratch they are read from, and switching the option off waits too.  Same bytes as the
    synchronous call."""
    sp, sn = common.make_skies(2048, 1024, "smooth")
    gpu_ctx.set_sky(0, curvis_amd.SphericalImage(sp))
    gpu_ctx.set_sky(1, curvis_amd.SphericalImage(sn))
    w, h, n = 1280, 720, 6
    cams = []
    for k in range(n):
        _, _, pm, pc = common.scene("ellis", res=(w, h), pos=(0.0, 3.0 + k, common.HALF_PI, 0.4 * k))
        cams.append(pc)
    gpu_ctx.render_efficient(pm, cams, 4096, 100.0, 0.05, 100, 100, 1e-5, 1e-5, download=False)
    want, _, want_crc = gpu_ctx.deflate_frames_crc(w, h, n)
    buf = curvis_amd.HostBuffer(n * w * h * 3)
    try:
        gpu_ctx.set_option("async_streams", 1)
        for rep in range(3):
            buf.array[:] = 0xAB
            offs, ms, crc = gpu_ctx.deflate_frames_into(w, h, n, buf.array)
            assert gpu_ctx.get_option("streams_pending") == 1 and crc == want_crc and ms > 0
            if rep == 0:
                gpu_ctx.download_wait()
            elif rep == 1:
                gpu_ctx.render_efficient(pm, cams, 4096, 100.0, 0.05, 100, 100, 1e-5, 1e-5, download=False)   # a render call does not wait for them ...
                assert gpu_ctx.get_option("streams_pending") == 1
                other = curvis_amd.HostBuffer(n * w * h * 3)     # (kept alive until its own streams have arrived)
                gpu_ctx.deflate_frames_into(w, h, n, other.array)                                               # ... the next deflate call does, before its own
                gpu_ctx.download_wait()
                other.close()
            else:
                gpu_ctx.set_option("async_streams", 0)                                                          # switching it off waits
            assert gpu_ctx.get_option("streams_pending") == 0
            got = [bytes(buf.array[offs[k]:offs[k + 1]]) for k in range(n)]
            assert got == want, rep
            assert all(zlib.crc32(b"IDAT" + z) == c for z, c in zip(got, crc))
    finally:
        gpu_ctx.set_option("async_streams", 0)
        buf.close()
