"""inflate_fast.h -- the DEFLATE decoder behind the PNG backgrounds of `curvis image|video` (the reference: image 0.25.2 ->
png 0.17.13 -> fdeflate / miniz_oxide, src/images.rs:8) -- against zlib, through the host twin: every conforming inflater
yields the same bytes, so on every VALID stream the two must agree byte for byte; on a damaged one this decoder must say so
(or produce what zlib produces, where the damage happens to leave a valid stream) -- never write outside its buffer, never
hang.  Streams of every compression level and strategy (stored, fixed-Huffman, dynamic blocks; long codes; matches at the
window's edge), contents from all-zero to incompressible, sizes from nothing to several MiB."""
import ctypes as C
import os
import zlib

import numpy as np
import pytest

import common

OK, E_DATA, E_TRUNCATED, E_OUTPUT_FULL, E_HEADER = 0, -1, -2, -3, -4


def inflate(z, cap):
    L = common.twin()
    L.twin_inflate_zlib.restype = C.c_int
    L.twin_inflate_zlib.argtypes = [C.c_char_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(C.c_uint32)]
    out = np.full(cap + 64, 0xA5, np.uint8)           # 64 guard bytes behind the capacity the decoder is told
    n, ad = C.c_size_t(0), C.c_uint32(0)
    rc = L.twin_inflate_zlib(z, len(z), out.ctypes.data, cap, C.byref(n), C.byref(ad))
    assert (out[cap:] == 0xA5).all(), "wrote past the end of the output buffer"
    return rc, out[:n.value].tobytes() if rc == OK else b"", ad.value


def contents(rng):
    yield "empty", b""
    yield "one byte", b"x"
    yield "zeros", bytes(300000)
    yield "one long run then noise", bytes(70000) + rng.integers(0, 256, 5000, dtype=np.uint8).tobytes()
    yield "noise", rng.integers(0, 256, 200000, dtype=np.uint8).tobytes()
    yield "text-like", (b"the quick brown fox jumps over the lazy dog. " * 5000)[:180001]
    yield "period 1..40", b"".join(bytes(range(p)) * (3000 // p) for p in range(1, 41))
    yield "far matches", rng.integers(0, 256, 32768, dtype=np.uint8).tobytes() * 5      # distance = the whole window
    x = np.cumsum(rng.integers(-3, 4, size=(300, 1200, 3)), axis=1) % 256                 # filtered-scanline-like
    yield "smooth image rows", b"".join(b"\x01" + np.diff(r.astype(np.uint8), axis=0, prepend=np.zeros((1, 3), np.uint8)).astype(np.uint8).tobytes() for r in x.astype(np.uint8))
    yield "skewed alphabet (long codes)", bytes(rng.choice(256, size=400000, p=np.r_[[0.5, 0.2, 0.1], np.full(253, 0.2 / 253)]).astype(np.uint8))
    yield "several MiB", rng.integers(0, 4, 5 << 20, dtype=np.uint8).tobytes()


def streams(data):
    for level in (0, 1, 2, 4, 6, 9):
        yield "level %d" % level, zlib.compress(data, level)
    for name, strat in (("filtered", zlib.Z_FILTERED), ("huffman only", zlib.Z_HUFFMAN_ONLY), ("rle", zlib.Z_RLE), ("fixed", zlib.Z_FIXED)):
        co = zlib.compressobj(6, zlib.DEFLATED, 15, 9, strat)
        yield name, co.compress(data) + co.flush()
    co = zlib.compressobj(6, zlib.DEFLATED, 9)                                        # a 512-byte window
    yield "small window", co.compress(data) + co.flush()
    co = zlib.compressobj(5)                                                          # many blocks, sync flushes (empty stored blocks) between
    parts = [co.compress(data[i:i + 7000]) + co.flush(zlib.Z_SYNC_FLUSH) for i in range(0, len(data), 7000)]
    yield "sync-flushed", b"".join(parts) + co.flush()


def test_every_valid_stream_decodes_to_what_zlib_decodes():
    rng = np.random.default_rng(2)
    n = 0
    for cname, data in contents(rng):
        for sname, z in streams(data):
            rc, got, ad = inflate(z, len(data) + 1)
            assert rc == OK and got == data, (cname, sname, rc, len(got), len(data))
            assert ad == zlib.adler32(data), (cname, sname)
            if len(data):                                                            # exactly enough room is enough; one byte less is an error
                assert inflate(z, len(data))[0] == OK and inflate(z, len(data) - 1)[0] == E_OUTPUT_FULL, (cname, sname)
            n += 1
    assert n > 100


def test_damaged_streams_are_errors_or_what_zlib_makes_of_them():
    """a flipped bit, a cut, junk: the decoder answers with an error or -- where the damage leaves a stream zlib accepts too, or only
    the checksum is off -- with zlib's bytes; it never writes past its buffer (guard bytes) and always returns"""
    rng = np.random.default_rng(3)
    base = []
    for cname, data in contents(rng):
        if len(data) > 400000:
            continue
        for sname, z in streams(data):
            if sname in ("level 1", "level 6", "fixed", "sync-flushed", "level 0"):
                base.append((data, z))
    errors = agreed = checksum_only = 0
    for trial in range(3000):
        data, z = base[int(rng.integers(0, len(base)))]
        b = bytearray(z)
        kind = trial % 3
        if kind == 0 and len(b) > 8:
            for _ in range(int(rng.integers(1, 4))):
                b[int(rng.integers(2, len(b)))] ^= 1 << int(rng.integers(0, 8))
        elif kind == 1:
            del b[int(rng.integers(0, len(b))):]
        else:
            pos = int(rng.integers(2, max(3, len(b))))
            b[pos:pos + 4] = rng.integers(0, 256, 4, dtype=np.uint8).tobytes()
        b = bytes(b)
        cap = len(data) + 64
        rc, got, ad = inflate(b, cap)
        d = zlib.decompressobj()
        try:
            ref = d.decompress(b, cap)
            verdict = "ok" if d.eof else ("full" if len(ref) == cap else "incomplete")
        except zlib.error as e:
            ref, verdict = None, ("checksum" if "incorrect data check" in str(e) else "error")
        if verdict == "ok":                       # the damage left a valid stream (or hit nothing that matters)
            assert rc == OK and got == ref and ad == zlib.adler32(ref), trial
            agreed += 1
        elif verdict == "checksum":               # valid DEFLATE data, wrong Adler-32: this decoder hands the stored value to its caller
            assert rc == OK and ad != zlib.adler32(got), trial
            checksum_only += 1
        elif verdict == "full":                   # more output than there is room for
            assert rc in (E_OUTPUT_FULL, E_DATA, E_TRUNCATED), (trial, rc)
            errors += 1
        else:                                     # zlib: invalid data, or the stream stops before its end
            assert rc in (E_DATA, E_TRUNCATED, E_OUTPUT_FULL), (trial, verdict, rc)
            errors += 1
    assert errors > 1500 and agreed + checksum_only > 10, (errors, agreed, checksum_only)


def test_headers_and_tiny_inputs():
    assert inflate(b"", 10)[0] == E_TRUNCATED
    assert inflate(b"\x78", 10)[0] == E_TRUNCATED
    good = zlib.compress(b"abc")
    assert inflate(good, 10)[:2] == (OK, b"abc")
    assert inflate(b"\x79" + good[1:], 10)[0] == E_HEADER            # compression method 9
    assert inflate(b"\x78\x9d" + good[2:], 10)[0] == E_HEADER        # FCHECK off
    assert inflate(bytes([0x78, 0xBB]) + good[2:], 10)[0] == E_HEADER  # preset dictionary
    assert inflate(good[:-1], 10)[0] == E_TRUNCATED                  # the Adler-32 is cut
    assert inflate(b"\x78\x9c\x07", 10)[0] in (E_DATA, E_TRUNCATED)  # block type 3
    assert inflate(b"\x78\x9c\x01\x01\x00\xff\xff", 10)[0] in (E_DATA, E_TRUNCATED)   # stored: LEN / NLEN do not match
