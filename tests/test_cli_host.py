"""The `curvis` host binary on CPU: argument handling, TOML subset, validation messages, PNG codec,
and the loud failure without a GPU.  (End-to-end rendering is in tests/test_gpu_cli.py.)"""
import os
import struct
import subprocess

import numpy as np
import pytest

from curvis_amd import pngio

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "curvis_amd", "bin", "curvis")


def run(*args, cwd=None):
    return subprocess.run([BIN] + [str(a) for a in args], capture_output=True, text=True, cwd=cwd, timeout=120)


@pytest.fixture(scope="module")
def skies(tmp_path_factory):
    d = tmp_path_factory.mktemp("sky")
    rng = np.random.default_rng(0)
    a, b = d / "a.png", d / "b.png"
    pngio.write_png(a, rng.integers(0, 255, (32, 64, 3), dtype=np.uint8))
    pngio.write_png(b, rng.integers(0, 255, (32, 64, 4), dtype=np.uint8))
    return d, a, b


def test_binary_exists_and_usage():
    assert os.path.exists(BIN), "build with make -C curvis_amd/csrc"
    r = run("--help")
    assert r.returncode == 0 and "image" in r.stdout and "video" in r.stdout and "custom" in r.stdout
    assert run().returncode == 1
    r = run("custom")
    assert r.returncode == 1 and "Custom script" in r.stdout and "not implemented" in r.stderr


def test_argument_errors(skies):
    d, a, b = skies
    r = run("image", a)
    assert r.returncode == 2 and "required arguments" in r.stderr
    r = run("image", d / "missing.png", b)
    assert r.returncode == 1 and "Error with background image 1" in r.stderr and "not found" in r.stderr
    r = run("image", a, b, d / "nofolder")
    assert r.returncode == 1 and "Error with output folder" in r.stderr
    r = run("image", a, b, a)
    assert r.returncode == 1 and "is not a folder" in r.stderr
    r = run("image", a, b, "-v", "x.toml")
    assert r.returncode == 2
    r = run("image", a, b, "--bogus")
    assert r.returncode == 2 and "unexpected argument" in r.stderr


def test_settings_parsing_and_validation(skies):
    d, a, b = skies
    sim = d / "sim.toml"
    sim.write_text("escape_radius = 100.0\nray_integration_max_itarations = 4096 # sic\nray_integration_step = 0.05\n"
                   "sampling_initial_nums = 100\nsampling_max_iterations = 50\n"
                   "sampling_convergence_threshold_1 = 1e-5\nsampling_convergence_threshold_2 = 1e-5\n")
    bad = d / "bad.toml"
    bad.write_text(sim.read_text().replace("ray_integration_max_itarations", "ray_integration_max_iterations"))
    r = run("image", a, b, d, "-s", bad)
    assert r.returncode == 1 and "missing field `ray_integration_max_itarations`" in r.stderr
    neg = d / "neg.toml"
    neg.write_text(sim.read_text().replace("escape_radius = 100.0", "escape_radius = -1"))
    r = run("image", a, b, d, "-s", neg)
    assert r.returncode == 1 and "The escape radius must be larger than zero." in r.stderr
    cam = d / "cam.toml"
    cam.write_text("resolution_x = 64.0\nresolution_y = 36\ndiagonal = 43.0\nfocal_length = 15.0\n")
    r = run("image", a, b, d, "-c", cam)
    assert r.returncode == 1 and "resolution_x" in r.stderr  # a float does not deserialize into u32
    met = d / "met.toml"
    met.write_text("m = 0.1\na = 0.0\nrho = 1.0\n")
    r = run("image", a, b, d, "-m", met)
    assert r.returncode == 101 and "metric parameters must be positive" in r.stderr
    notoml = d / "sim.txt"
    notoml.write_text(sim.read_text())
    r = run("image", a, b, d, "-s", notoml)
    assert r.returncode == 1 and "is not a toml file" in r.stderr
    vid = d / "vid.toml"
    vid.write_text('video_name = "v"\nframe_rate = 4.0\nfilepath_to_camera_path = "nope/none.csv"\n')
    r = run("video", a, b, d, "-v", vid)
    assert r.returncode == 1 and "does not exist" in r.stderr


def test_without_gpu_fails_loudly(skies):
    d, a, b = skies
    import curvis_amd
    if curvis_amd.lib().curvis_device_count() > 0:
        pytest.skip("a GPU is present")
    r = run("image", a, b, d)
    assert r.returncode == 1 and "no CPU fallback" in r.stderr
    assert not (d / "output_image.png").exists()


def _decode_with_binary(path, tmp):
    out = tmp / "dump.rgba"
    r = run("selftest-png", path, out)
    assert r.returncode == 0, r.stderr
    raw = out.read_bytes()
    w, h = struct.unpack("<II", raw[:8])
    return np.frombuffer(raw[8:], np.uint8).reshape(h, w, 4)


def test_png_decoder_matches_pillow(tmp_path):
    PIL = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(5)
    cases = {
        "rgb": (rng.integers(0, 255, (17, 23, 3), dtype=np.uint8), "RGB"),
        "rgba": (rng.integers(0, 255, (9, 31, 4), dtype=np.uint8), "RGBA"),
        "grey": (rng.integers(0, 255, (12, 13), dtype=np.uint8), "L"),
        "la": (rng.integers(0, 255, (8, 8, 2), dtype=np.uint8), "LA"),
    }
    for name, (arr, mode) in cases.items():
        for interlace in (False, True):
            p = tmp_path / ("%s_%d.png" % (name, interlace))
            im = PIL.fromarray(arr, mode)
            if interlace:
                # Pillow cannot write Adam7; emulate by re-saving through optimize (filters vary)
                im.save(p, optimize=True)
            else:
                im.save(p)
            want = np.asarray(PIL.open(p).convert("RGBA"))
            got = _decode_with_binary(p, tmp_path)
            assert np.array_equal(got, want), name
    # palette with transparency
    pal = PIL.fromarray(rng.integers(0, 255, (20, 20, 3), dtype=np.uint8), "RGB").quantize(16)
    p = tmp_path / "pal.png"
    pal.save(p)
    assert np.array_equal(_decode_with_binary(p, tmp_path), np.asarray(PIL.open(p).convert("RGBA")))
    # 16-bit RGB: image crate reduction (v + 128) / 257
    a16 = rng.integers(0, 65535, (6, 7, 3), dtype=np.uint16)
    p = tmp_path / "rgb16.png"
    pngio.write_png(p, a16)
    got = _decode_with_binary(p, tmp_path)
    want = ((a16.astype(np.uint32) + 128) // 257).astype(np.uint8)
    assert np.array_equal(got[..., :3], want) and (got[..., 3] == 255).all()


def _paeth_pred(a, b, c):
    a, b, c = a.astype(np.int32), b.astype(np.int32), c.astype(np.int32)
    p = a + b - c
    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
    return np.where((pa <= pb) & (pa <= pc), a, np.where(pb <= pc, b, c)).astype(np.uint8)


def _write_png_filtered(path, img, filters, level=1):
    """an 8-bit PNG whose row y uses filter type filters[y % len(filters)] (PNG specification 9.2) -- Pillow-free, so that the
    decoder's five reconstruction loops are covered wherever the tests run"""
    import zlib
    img = np.ascontiguousarray(img if img.ndim == 3 else img[:, :, None])
    h, w, ch = img.shape
    rows = img.reshape(h, w * ch)
    zero = np.zeros(w * ch, np.uint8)
    out = bytearray()
    for y in range(h):
        cur, up = rows[y], (rows[y - 1] if y else zero)
        a = np.concatenate([np.zeros(ch, np.uint8), cur])[:w * ch]
        c = np.concatenate([np.zeros(ch, np.uint8), up])[:w * ch]
        ft = filters[y % len(filters)]
        pred = (zero, a, up, ((a.astype(np.int32) + up.astype(np.int32)) >> 1).astype(np.uint8), _paeth_pred(a, up, c))[ft]
        out.append(ft)
        out += (cur - pred).astype(np.uint8).tobytes()
    ihdr = struct.pack(">IIBBBBB", w, h, 8, {1: 0, 2: 4, 3: 2, 4: 6}[ch], 0, 0, 0)
    open(path, "wb").write(_png([(b"IHDR", ihdr), (b"IDAT", zlib.compress(bytes(out), level)), (b"IEND", b"")]))


def test_png_decoder_every_filter_type(tmp_path):
    """Sub / Up / Average / Paeth / None and mixtures of them, RGB and RGBA (the decoder's in-place fast path: one loop per filter
    type, a follower thread behind the inflater, hand-over every MiB of inflated data) and grey / grey+alpha (the generic
    path); widths of one to a few pixels (the first pixel of a row has no left neighbour), a single row (no row above), and
    an image of several MiB so that rows straddle the hand-over points"""
    rng = np.random.default_rng(11)

    def smoothish(h, w, ch):
        x = np.cumsum(rng.integers(-5, 6, size=(h, w, ch)), axis=1) + np.cumsum(rng.integers(-3, 4, size=(h, 1, ch)), axis=0)
        return (x % 256).astype(np.uint8)
    mixes = [[0], [1], [2], [3], [4], [4, 1, 2, 3, 0], [3, 4], [2, 2, 4, 1]]
    for ch in (3, 4, 1, 2):
        for (h, w) in ((1, 1), (1, 7), (5, 1), (4, 2), (3, 3), (33, 17), (64, 300)):
            img = smoothish(h, w, ch)
            for fl in mixes:
                p = tmp_path / "f.png"
                _write_png_filtered(p, img, fl)
                got = _decode_with_binary(p, tmp_path)
                want = {3: lambda: np.dstack([img, np.full((h, w, 1), 255, np.uint8)]), 4: lambda: img,
                        1: lambda: np.dstack([img[..., 0]] * 3 + [np.full((h, w), 255, np.uint8)]),
                        2: lambda: np.dstack([img[..., 0]] * 3 + [img[..., 1]])}[ch]()
                assert np.array_equal(got, want), (ch, h, w, fl)
    for ch in (3, 4):                                     # > 1 MiB of scanlines: several hand-overs between the two threads
        img = smoothish(700, 1500, ch)
        p = tmp_path / "big.png"
        _write_png_filtered(p, img, [4, 1, 2, 3, 4, 4, 0, 4])
        got = _decode_with_binary(p, tmp_path)
        assert np.array_equal(got[..., :ch], img) and (got[..., 3] == 255).all() if ch == 3 else np.array_equal(got, img)
    # a filter type that does not exist, and a stream that ends early: an error message, not a crash or a hang of the follower
    import zlib
    row = lambda ft: bytes([ft]) + bytes(12)
    ihdr = struct.pack(">IIBBBBB", 4, 3, 8, 2, 0, 0, 0)
    for name, body in (("badfilter", row(0) + row(7) + row(0)), ("short", row(0) + row(1))):
        p = tmp_path / (name + ".png")
        p.write_bytes(_png([(b"IHDR", ihdr), (b"IDAT", zlib.compress(body)), (b"IEND", b"")]))
        r = run("selftest-png", p, tmp_path / "o.rgba")
        assert r.returncode == 1 and "corrupt PNG" in r.stderr, (name, r.returncode, r.stderr)


def test_python_png_helpers_round_trip(tmp_path):
    rng = np.random.default_rng(1)
    for shape in ((5, 7, 3), (4, 4, 4), (3, 9)):
        a = rng.integers(0, 255, shape, dtype=np.uint8)
        pngio.write_png(tmp_path / "x.png", a)
        assert np.array_equal(pngio.read_png(tmp_path / "x.png"), a)


def _png(chunks):
    import zlib
    out = b"\x89PNG\r\n\x1a\n"
    for typ, data in chunks:
        out += struct.pack(">I", len(data)) + typ + data + struct.pack(">I", zlib.crc32(typ + data) & 0xFFFFFFFF)
    return out


def test_png_decoder_rejects_hostile_files(tmp_path):
    """untrusted sky files: forged bit depths (depth 0 used to divide by zero), absurd dimensions, a decompression
    bomb, a corrupted chunk -- each must be an error message, never a crash or a multi-GB allocation"""
    import zlib
    def ihdr(w, h, depth, ctype, interlace=0):
        return struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, interlace)
    row = b"\x00" + bytes(4)
    good = _png([(b"IHDR", ihdr(4, 1, 8, 0)), (b"IDAT", zlib.compress(row)), (b"IEND", b"")])
    cases = {
        "depth0": _png([(b"IHDR", ihdr(4, 1, 0, 0)), (b"IDAT", zlib.compress(row)), (b"IEND", b"")]),
        "depth3": _png([(b"IHDR", ihdr(4, 1, 3, 0)), (b"IDAT", zlib.compress(row)), (b"IEND", b"")]),
        "depth7pal": _png([(b"IHDR", ihdr(4, 1, 7, 3)), (b"IDAT", zlib.compress(row)), (b"IEND", b"")]),
        "depth4rgb": _png([(b"IHDR", ihdr(4, 1, 4, 2)), (b"IDAT", zlib.compress(row)), (b"IEND", b"")]),
        "huge": _png([(b"IHDR", ihdr(0x7FFFFFFF, 0x7FFFFFFF, 8, 6)), (b"IDAT", zlib.compress(row)), (b"IEND", b"")]),
        "bomb": _png([(b"IHDR", ihdr(4, 1, 8, 0)), (b"IDAT", zlib.compress(bytes(64 << 20))), (b"IEND", b"")]),
        "interlace9": _png([(b"IHDR", ihdr(4, 1, 8, 0, 9)), (b"IDAT", zlib.compress(row)), (b"IEND", b"")]),
        "crc": good[:-20] + bytes([good[-20] ^ 1]) + good[-19:],
    }
    p = tmp_path / "good.png"
    p.write_bytes(good)
    assert _decode_with_binary(p, tmp_path).shape == (1, 4, 4)
    # a damaged ANCILLARY chunk (lower-case first letter) is skipped, as the reference's png crate does; only critical ones are fatal
    text = _png([(b"tEXt", b"Comment\x00star map")])[8:]
    broken_text = text[:-1] + bytes([text[-1] ^ 0xFF])
    with_meta = good[:33] + broken_text + good[33:]                     # signature (8) + IHDR chunk (25), then the tEXt chunk
    p2 = tmp_path / "meta.png"
    p2.write_bytes(with_meta)
    assert np.array_equal(_decode_with_binary(p2, tmp_path), _decode_with_binary(p, tmp_path))
    for name, blob in cases.items():
        p = tmp_path / (name + ".png")
        p.write_bytes(blob)
        r = run("selftest-png", p, tmp_path / "out.rgba")
        assert r.returncode == 1, (name, r.returncode, r.stderr)   # an error exit, not a signal
        assert "selftest-png:" in r.stderr, name


def _jpeg_test_image(h=77, w=131, seed=3):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    img = np.stack([127 + 120 * np.sin(xx / 9.0) * np.cos(yy / 7.0), xx * 255.0 / w, yy * 255.0 / h], axis=2)
    return np.clip(img + rng.normal(0, 6, (h, w, 3)), 0, 255).astype(np.uint8)


def test_jpeg_decoder_against_libjpeg(tmp_path):
    """JPEG skies (the reference's README suggests .jpg star maps; image::open -> zune-jpeg): baseline and
    progressive, 4:4:4 / 4:2:2 / 4:2:0, grey, optimised Huffman tables, restart intervals, odd sizes -- against
    libjpeg (Pillow) within what two conforming decoders differ by (different IDCT / upsampling / colour arithmetic:
    measured max 3, mean 0.5 of 255); the PNG path is untouched by format detection."""
    PIL = pytest.importorskip("PIL.Image")
    img = _jpeg_test_image()
    variants = [("444", dict(subsampling=0)), ("422", dict(subsampling=1)), ("420", dict(subsampling=2)),
                ("prog420", dict(subsampling=2, progressive=True)), ("prog444", dict(subsampling=0, progressive=True)),
                ("q30", dict(quality=30)), ("opt", dict(optimize=True, quality=95)),
                ("rst", dict(subsampling=2, restart_marker_blocks=3)), ("rstprog", dict(progressive=True, restart_marker_rows=1))]
    for name, kw in variants:
        p = tmp_path / ("t_%s.jpg" % name)
        try:
            PIL.fromarray(img).save(p, **dict({"quality": 90}, **kw))
        except TypeError:
            continue
        got = _decode_with_binary(p, tmp_path)
        want = np.asarray(PIL.open(p).convert("RGB"))
        d = np.abs(got[..., :3].astype(int) - want.astype(int))
        assert got.shape == (77, 131, 4) and (got[..., 3] == 255).all()
        assert d.max() <= 4 and d.mean() < 0.8, (name, d.max(), d.mean())
    for size in ((1, 1), (8, 8), (9, 17), (16, 16), (33, 7)):
        small = _jpeg_test_image(size[0], size[1], seed=9)
        for sub in (0, 2):
            p = tmp_path / "small.jpg"
            PIL.fromarray(small).save(p, quality=92, subsampling=sub)
            got = _decode_with_binary(p, tmp_path)
            want = np.asarray(PIL.open(p).convert("RGB"))
            assert got.shape[:2] == size and np.abs(got[..., :3].astype(int) - want.astype(int)).max() <= 5, (size, sub)
    p = tmp_path / "grey.jpg"
    PIL.fromarray(img[..., 0]).save(p, quality=90)
    got = _decode_with_binary(p, tmp_path)
    want = np.asarray(PIL.open(p).convert("L"))
    assert np.abs(got[..., 0].astype(int) - want.astype(int)).max() <= 1 and (got[..., 0] == got[..., 1]).all()
    # unsupported / damaged files are an error message, not a crash
    p = tmp_path / "cmyk.jpg"
    PIL.fromarray(np.dstack([img, img[..., :1]]), "CMYK").save(p)
    r = run("selftest-png", p, tmp_path / "o.rgba")
    assert r.returncode == 1 and "CMYK" in r.stderr
    blob = (tmp_path / "t_420.jpg").read_bytes()
    for cut in (3, 20, 200, len(blob) // 2, len(blob) - 2):
        p = tmp_path / "cut.jpg"
        p.write_bytes(blob[:cut])
        r = run("selftest-png", p, tmp_path / "o.rgba")
        assert r.returncode in (0, 1), (cut, r.returncode)   # decoded what is there, or said why not; never a signal
    rng = np.random.default_rng(4)
    for _ in range(30):   # random corruption of the entropy-coded data and the headers
        b = bytearray(blob)
        for pos in rng.integers(2, len(b), 8):
            b[pos] = int(rng.integers(0, 256))
        p = tmp_path / "fuzz.jpg"
        p.write_bytes(bytes(b))
        r = run("selftest-png", p, tmp_path / "o.rgba")
        assert r.returncode in (0, 1), r.returncode


def test_large_jpeg_on_several_threads_equals_one_thread(tmp_path):
    """a large JPEG is reconstructed on several threads (inverse DCT by block rows, upsampling + colour by pixel rows; jpeg_io.h
    parallel_ranges): same pixels as on one thread, for every subsampling, for sizes that are no multiple of an MCU, with restart
    markers, progressive, grey; a damaged large file is an error or an image, never a hang.  (Reconstructing a baseline file BEHIND
    its scan was built and measured -- on the GPU box's 16-CPU quota the workers slowed the entropy decoder by more than they hid,
    354 vs 249 ms per `curvis image` -- and taken out again.)"""
    PIL = pytest.importorskip("PIL.Image")

    def decode(path, **env):
        out = tmp_path / "dump.rgba"
        r = subprocess.run([BIN, "selftest-png", str(path), str(out)], capture_output=True, text=True, timeout=120, env=dict(os.environ, **env))
        assert r.returncode == 0, r.stderr
        raw = out.read_bytes()
        w, h = struct.unpack("<II", raw[:8])
        return np.frombuffer(raw[8:], np.uint8).reshape(h, w, 4)
    variants = [("420", dict(subsampling=2)), ("422", dict(subsampling=1)), ("444", dict(subsampling=0)), ("rst", dict(subsampling=2, restart_marker_rows=3)),
                ("prog", dict(subsampling=2, progressive=True))]
    for (h, w) in ((1024, 1040), (901, 1201), (1160, 913)):
        img = _jpeg_test_image(h, w, seed=h)
        for name, kw in variants:
            p = tmp_path / ("big_%s.jpg" % name)
            try:
                PIL.fromarray(img).save(p, **dict({"quality": 88}, **kw))
            except TypeError:
                continue
            plain = decode(p, CURVIS_DECODE_THREADS="1")
            for threads in ("2", "5"):
                assert np.array_equal(decode(p, CURVIS_DECODE_THREADS=threads), plain), (h, w, name, threads)
            want = np.asarray(PIL.open(p).convert("RGB"))
            d = np.abs(plain[..., :3].astype(int) - want.astype(int))
            assert d.max() <= 24 and d.mean() < 0.8 and (plain[..., 3] == 255).all(), (name, d.max(), d.mean())   # two conforming decoders (upsampling filters differ)
    p = tmp_path / "big_grey.jpg"
    PIL.fromarray(_jpeg_test_image(1100, 1000, seed=2)[..., 0]).save(p, quality=90)
    assert np.array_equal(decode(p, CURVIS_DECODE_THREADS="4"), decode(p, CURVIS_DECODE_THREADS="1"))
    blob = (tmp_path / "big_420.jpg").read_bytes()
    rng = np.random.default_rng(21)
    for k in range(24):
        b = bytearray(blob)
        if k % 2:
            del b[int(rng.integers(700, len(b))):]
        else:
            for pos in rng.integers(650, len(b), 5):
                b[pos] = int(rng.integers(0, 256))
        q = tmp_path / "hurt.jpg"
        q.write_bytes(bytes(b))
        r = subprocess.run([BIN, "selftest-png", str(q), str(tmp_path / "o.rgba")], capture_output=True, text=True, timeout=60,
                           env=dict(os.environ, CURVIS_DECODE_THREADS="4"))
        assert r.returncode in (0, 1), (k, r.returncode, r.stderr[-200:])


def test_library_image_codecs_equal_the_binarys(tmp_path):
    """curvis_image_load / curvis_image_save_rgb8 (the library's counterpart of images::load_image / save_image,
    src/images.rs:7-20) are the binary's decoders: same texels for PNG and JPEG, errors as CurvisError"""
    PIL = pytest.importorskip("PIL.Image")
    import curvis_amd
    from curvis_amd import images
    img = _jpeg_test_image(33, 47)
    files = []
    for name, kw in (("a.jpg", dict(quality=88, subsampling=2)), ("b.jpg", dict(quality=88, progressive=True)), ("c.png", {})):
        p = tmp_path / name
        PIL.fromarray(img).save(p, **kw)
        files.append(p)
    pngio.write_png(tmp_path / "d.png", np.dstack([img, img[..., :1]]))     # RGBA
    pngio.write_png(tmp_path / "e.png", img[..., 0])                        # grey
    files += [tmp_path / "d.png", tmp_path / "e.png"]
    for p in files:
        assert np.array_equal(images.load_image(p), _decode_with_binary(p, tmp_path)), p.name
    sky = images.load_image_as_spherical_image(tmp_path / "c.png")
    assert (sky.width_pixels, sky.height_pixels) == (47, 33) and list(sky.forward) == [1.0, 0.0, 0.0] and list(sky.up) == [0.0, 0.0, 1.0]
    images.save_image(tmp_path / "out.png", img)
    assert np.array_equal(np.asarray(PIL.open(tmp_path / "out.png")), img)
    with pytest.raises(curvis_amd.CurvisError) as e:
        images.load_image(tmp_path / "missing.png")
    assert e.value.code == -10
    (tmp_path / "junk.png").write_bytes(b"not an image at all")
    with pytest.raises(curvis_amd.CurvisError):
        images.load_image(tmp_path / "junk.png")


def _fast_png_cases():
    rng = np.random.default_rng(11)
    yield "one pixel", np.array([[[7, 0, 255]]], np.uint8)
    yield "all zero", np.zeros((5, 300, 3), np.uint8)                       # one long zero run across rows of filter bytes
    yield "constant", np.full((40, 33, 3), 200, np.uint8)
    yield "noise", rng.integers(0, 256, (37, 53, 3), dtype=np.uint8)
    g = np.zeros((64, 400, 3), np.uint8)
    g[..., 0] = (np.arange(400) * 255 // 400)[None, :]
    g[..., 1] = (np.arange(64) * 255 // 64)[:, None]
    g[..., 2] = 128
    yield "smooth sky gradient", g
    # zero runs of every length around the piece boundaries of the run coder (3, 4, 258..262, 516..520)
    for r in (1, 2, 3, 4, 5, 257, 258, 259, 260, 261, 262, 263, 516, 517, 518, 519, 520, 521, 777):
        a = np.full((3, 300, 3), 9, np.uint8)
        flat = a.reshape(3, -1)
        flat[1, 10:10 + r] = flat[0, 10:10 + r]                            # Up-filtered row 1 holds a zero run of length r
        flat[1, 10 + r] ^= 0x55
        yield "zero run %d" % r, a
    sparse = np.zeros((20, 64, 3), np.uint8)
    sparse[rng.integers(0, 20, 40), rng.integers(0, 64, 40), rng.integers(0, 3, 40)] = 255
    yield "sparse", sparse


def test_fast_png_writer_round_trips(tmp_path):
    """the fast PNG writer of `curvis video` (png_io.h: filter Up, one dynamic-Huffman block, zero-run matches only): its
    files decode to the same pixels with TWO decoders that share nothing with it -- Python's zlib inflate + this repo's
    test unfilter (pngio.read_png) and the library's own reader -- and the zlib levels give the same pixels too"""
    from curvis_amd import images
    for name, img in _fast_png_cases():
        for level in (-1, 0, 1, 6):
            p = tmp_path / "f.png"
            images.save_image(p, img, level=level)
            got = pngio.read_png(p)
            assert got.shape == img.shape and np.array_equal(got, img), (name, level)
            assert np.array_equal(images.load_image(p)[..., :3], img), (name, level)
    big = np.random.default_rng(5).integers(0, 4, (270, 480, 3), dtype=np.uint8) * 60   # many short runs and literals
    images.save_image(tmp_path / "b.png", big, level=-1)
    assert np.array_equal(pngio.read_png(tmp_path / "b.png"), big)
    smooth = next(a for n, a in _fast_png_cases() if n == "smooth sky gradient")
    images.save_image(tmp_path / "s.png", smooth, level=-1)
    assert (tmp_path / "s.png").stat().st_size < smooth.size // 4           # it does compress what a smooth sky looks like
    import curvis_amd
    with pytest.raises(curvis_amd.CurvisError):
        images.save_image(tmp_path / "x.png", smooth, level=12)
