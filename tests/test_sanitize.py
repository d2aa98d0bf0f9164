"""ASan + UBSan pass over the native test infrastructure (SURVEY.md section 5 "race detection / sanitizers": the
reference is safe Rust; the restatements here are C and C++): oracle/curvis_oracle.c and tests/host_twin/twin.cpp
-- i.e. the product's per-ray headers cv_device.h / cv_math.h / cv_efficient.h / cv_sampler.h compiled for x86 --
are built with -fsanitize=address,undefined and driven through both renderers, both flavours, all metrics and
adversarial cameras by tests/sanitize/san_driver.cpp, which also cross-checks oracle(cv) == twin."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
D = os.path.join(ROOT, "tests", "sanitize")


def test_oracle_and_host_twin_are_clean_under_asan_and_ubsan():
    subprocess.run(["make", "-s", "-C", D], check=True)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([os.path.join(D, "san_driver")], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, (r.stdout + r.stderr)[-4000:]
    assert "sanitize ok" in r.stdout and "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr


def test_image_decoders_are_clean_under_asan_and_ubsan_on_hostile_files(tmp_path):
    """PNG and JPEG decoders of the host binary on ~400 mutated files (bit flips in headers, tables and entropy-coded
    data, truncations): any outcome is fine except a sanitizer report"""
    import numpy as np
    PIL = __import__("pytest").importorskip("PIL.Image")
    subprocess.run(["make", "-s", "-C", D, "san_images"], check=True)
    rng = np.random.default_rng(11)
    yy, xx = np.mgrid[0:40, 0:56]
    img = np.clip(np.stack([127 + 100 * np.sin(xx / 5.0), xx * 4.0, yy * 6.0], axis=2) + rng.normal(0, 8, (40, 56, 3)), 0, 255).astype(np.uint8)
    seeds = []
    for name, kw in (("a.jpg", dict(quality=85, subsampling=2)), ("b.jpg", dict(quality=85, progressive=True)),
                     ("c.jpg", dict(quality=60, subsampling=1, optimize=True)), ("d.png", {}), ("e.png", dict(optimize=True))):
        p = tmp_path / name
        PIL.fromarray(img).save(p, **kw)
        seeds.append(p.read_bytes())
    PIL.fromarray(img[..., 0]).save(tmp_path / "g.jpg")
    seeds.append((tmp_path / "g.jpg").read_bytes())
    files = []
    for k in range(400):
        b = bytearray(seeds[k % len(seeds)])
        mode = k % 4
        if mode == 0:      # a few random bytes anywhere
            for pos in rng.integers(0, len(b), 6):
                b[pos] = int(rng.integers(0, 256))
        elif mode == 1:    # damage concentrated in the first 600 bytes (headers, tables)
            for pos in rng.integers(0, min(600, len(b)), 4):
                b[pos] ^= 1 << int(rng.integers(0, 8))
        elif mode == 2:    # truncation
            b = b[:int(rng.integers(1, len(b)))]
        else:              # a run of 0xFF / zeros
            pos = int(rng.integers(0, len(b) - 8))
            b[pos:pos + 8] = bytes([0xFF if k % 8 == 3 else 0x00]) * 8
        f = tmp_path / ("m%03d.bin" % k)
        f.write_bytes(bytes(b))
        files.append(str(f))
    r = subprocess.run([os.path.join(D, "san_images")] + files, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    assert "decoded" in r.stdout and "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr


def test_large_images_on_several_threads_are_clean_under_asan_and_ubsan(tmp_path):
    """large files take the decoders' multi-threaded paths -- a JPEG reconstructed on several threads, a PNG inflated
    by one stage, reconstructed in place by a second, written as RGBA by a third -- intact and damaged, under ASan + UBSan"""
    import numpy as np
    PIL = __import__("pytest").importorskip("PIL.Image")
    subprocess.run(["make", "-s", "-C", D, "san_images"], check=True)
    rng = np.random.default_rng(13)
    yy, xx = np.mgrid[0:1000, 0:1100]
    img = np.clip(np.stack([127 + 100 * np.sin(xx / 50.0), xx * 0.2, yy * 0.25], axis=2) + rng.normal(0, 6, (1000, 1100, 3)), 0, 255).astype(np.uint8)
    seeds = []
    for name, kw in (("a.jpg", dict(quality=85, subsampling=2)), ("b.jpg", dict(quality=85, subsampling=0, restart_marker_rows=2)), ("c.png", {})):
        p = tmp_path / name
        try:
            PIL.fromarray(img).save(p, **kw)
        except TypeError:
            PIL.fromarray(img).save(p)
        seeds.append(p.read_bytes())
    files = []
    for k in range(27):
        b = bytearray(seeds[k % 3])
        if k >= 3:
            if k % 2:
                b = b[:int(rng.integers(600, len(b)))]
            else:
                for pos in rng.integers(600, len(b), 5):
                    b[pos] = int(rng.integers(0, 256))
        f = tmp_path / ("big%02d.bin" % k)
        f.write_bytes(bytes(b))
        files.append(str(f))
    env = dict(os.environ, CURVIS_DECODE_THREADS="3", ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([os.path.join(D, "san_images")] + files, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    assert "decoded" in r.stdout and "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr
    assert int(r.stdout.split("decoded")[1].split(",")[0]) >= 3          # the intact ones at least


def test_fast_png_writer_is_clean_under_asan_and_ubsan():
    """the fast PNG writer of `curvis video` (png_io.h) on 400 small images of awkward shapes and contents (all zero, noise,
    long runs, gradients, short runs), each decoded again and compared, under ASan + UBSan: its bit writer stores eight
    bytes at a time and its run coder splits runs at 258 -- the places an off-by-one would hide"""
    subprocess.run(["make", "-s", "-C", D, "san_images"], check=True)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([os.path.join(D, "san_images"), "--encode"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, (r.stdout + r.stderr)[-4000:]
    assert "encoded and decoded 400 images" in r.stdout and "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr


def test_inflater_is_clean_under_asan_and_ubsan():
    """inflate_fast.h (the PNG backgrounds' DEFLATE decoder: a 64-bit bit reader that looks ahead, the output buffer as its own
    window, word-wise match copies) on 300 intact and 3 600 damaged zlib streams, input and output in heap buffers of exactly the
    size it is told: AddressSanitizer sees a read or write one byte too far, UBSan a shift or an overflow"""
    subprocess.run(["make", "-s", "-C", D, "san_images"], check=True)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([os.path.join(D, "san_images"), "--inflate"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, (r.stdout + r.stderr)[-4000:]
    assert "inflated 300 intact streams" in r.stdout and "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr

