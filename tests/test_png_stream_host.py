"""Host half of the device PNG front end, without a GPU: curvis_image_save_zlib_rgb8 wraps a finished zlib stream of
Up-filtered scanlines (what curvis_ctx_deflate_frames returns) into a PNG file -- signature, IHDR, IDAT + CRC-32, IEND --
that both decoders of this repository read back to the pixels."""
import ctypes as C
import zlib

import numpy as np
import pytest

from curvis_amd import _abi, pngio


def up_filtered(img):
    h, w, _ = img.shape
    flat = img.reshape(h, w * 3).astype(np.int16)
    prev = np.vstack([np.zeros((1, w * 3), np.int16), flat[:-1]])
    rows = ((flat - prev) & 255).astype(np.uint8)
    return np.hstack([np.full((h, 1), 2, np.uint8), rows]).tobytes()


@pytest.mark.parametrize("shape", [(1, 1), (5, 3), (64, 36), (333, 17)])
def test_wrap_zlib_stream_into_png(tmp_path, shape):
    rng = np.random.default_rng(shape[0])
    w, h = shape
    img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    img[h // 2:] = img[h // 2]                       # zero runs after the Up filter
    z = zlib.compress(up_filtered(img), 1)
    buf = np.frombuffer(z, np.uint8)
    path = tmp_path / "x.png"
    _abi.check(_abi.lib().curvis_image_save_zlib_rgb8(str(path).encode(), buf.ctypes.data, buf.size, w, h))
    assert np.array_equal(pngio.read_png(path), img)
    p, ww, hh = C.POINTER(C.c_uint8)(), C.c_uint32(), C.c_uint32()
    _abi.check(_abi.lib().curvis_image_load(str(path).encode(), C.byref(p), C.byref(ww), C.byref(hh)))
    got = np.ctypeslib.as_array(p, shape=(hh.value, ww.value, 4)).copy()
    _abi.lib().curvis_image_free(p)
    assert (ww.value, hh.value) == (w, h) and np.array_equal(got[..., :3], img) and (got[..., 3] == 255).all()
    # with the chunk's CRC-32 supplied (what curvis_ctx_deflate_frames_crc computes on the device): the same file, byte for byte,
    # and the decoders -- which verify the CRC of critical chunks -- refuse a wrong one
    with_crc = tmp_path / "x_crc.png"
    _abi.check(_abi.lib().curvis_image_save_zlib_rgb8_crc(str(with_crc).encode(), buf.ctypes.data, buf.size, w, h, zlib.crc32(b"IDAT" + z)))
    assert with_crc.read_bytes() == path.read_bytes()
    _abi.check(_abi.lib().curvis_image_save_zlib_rgb8_crc(str(with_crc).encode(), buf.ctypes.data, buf.size, w, h, zlib.crc32(b"IDAT" + z) ^ 1))
    assert _abi.lib().curvis_image_load(str(with_crc).encode(), C.byref(p), C.byref(ww), C.byref(hh)) != 0


def test_wrap_rejects_nonsense(tmp_path):
    assert _abi.lib().curvis_image_save_zlib_rgb8(str(tmp_path / "y.png").encode(), None, 10, 4, 4) == _abi.E_INVALID
    b = np.zeros(3, np.uint8)
    assert _abi.lib().curvis_image_save_zlib_rgb8(str(tmp_path / "y.png").encode(), b.ctypes.data, 3, 4, 4) == _abi.E_INVALID
    b = np.frombuffer(zlib.compress(b"\x02" + bytes(12)), np.uint8)
    assert _abi.lib().curvis_image_save_zlib_rgb8(b"/nonexistent-dir/y.png", b.ctypes.data, b.size, 4, 1) == _abi.E_IO


def test_length_limited_codes_of_the_shared_construction():
    """png_codes.h (the host writer's and the device's png_codes_kernel's code construction): every code is complete (Kraft sum
    exactly 1), within the limit, monotone in the counts, Huffman's when the limit does not bind and as cheap as an independent
    package-merge (written out below) when it does; length symbols follow RFC 1951 3.2.5."""
    import ctypes as C
    import heapq
    import common
    lib = common.twin()
    lib.twin_huffman_lengths.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    lib.twin_huffman_lengths.restype = None

    def lengths(freq, maxlen):
        f = np.ascontiguousarray(freq, dtype=np.uint32)
        out = np.zeros(f.size, np.uint8)
        lib.twin_huffman_lengths(f.ctypes.data, f.size, maxlen, out.ctypes.data)
        return out

    def optimal_cost(freq):
        h = [int(v) for v in freq if v]
        heapq.heapify(h)
        cost = 0
        while len(h) > 1:
            a, b = heapq.heappop(h), heapq.heappop(h)
            cost += a + b
            heapq.heappush(h, a + b)
        return cost

    def limited_optimal_cost(freq, maxlen):
        """package-merge (Larmore & Hirschberg): the cheapest prefix code with no code longer than maxlen"""
        idx = [i for i, v in enumerate(freq) if v]
        leaves = sorted((int(freq[i]), (i,)) for i in idx)
        if len(leaves) < 2:
            return int(sum(freq))
        packages = list(leaves)
        for _ in range(maxlen - 1):
            merged = [(packages[k][0] + packages[k + 1][0], packages[k][1] + packages[k + 1][1]) for k in range(0, len(packages) - 1, 2)]
            packages = sorted(leaves + merged, key=lambda t: t[0])
        lens = {}
        for wgt, syms in packages[: 2 * len(leaves) - 2]:
            for sy in syms:
                lens[sy] = lens.get(sy, 0) + 1
        assert max(lens.values()) <= maxlen
        return sum(int(freq[i]) * l for i, l in lens.items())

    rng = np.random.default_rng(2026)
    cases = []
    for n in (2, 3, 19, 30, 286, 288):
        cases.append(np.ones(n, np.uint32))
        cases.append(rng.integers(1, 1000, n).astype(np.uint32))
        cases.append((rng.pareto(0.5, n) * 3 + 1).clip(1, 4e6).astype(np.uint32))          # heavy tail: deep trees
        cases.append(np.minimum(2 ** np.arange(n, dtype=np.float64), 2.0 ** 31).astype(np.uint32))   # Fibonacci-like: depth n - 1
        z = rng.integers(0, 50, n).astype(np.uint32); z[rng.random(n) < 0.5] = 0; z[0] = 7; z[1] = 9; cases.append(z)   # absent symbols
    img = np.ones(286, np.uint32); img[0] = 5_000_000; img[1] = img[255] = 400_000; img[2:16] += 30_000; img[240:255] += 30_000; img[276] += 90_000
    cases.append(img)                                                                          # what a filtered 1080p frame looks like
    for freq in cases:
        for maxlen in (12, 15):
            if freq.size > 256 and maxlen < 9:
                continue
            L = lengths(freq, maxlen)
            used = freq > 0
            assert (L[~used] == 0).all() and (L[used] >= 1).all() and L.max() <= maxlen
            if used.sum() >= 2:
                assert sum(2 ** (maxlen - int(l)) for l in L[used]) == 2 ** maxlen, "not a complete prefix code"
            order = np.argsort(freq[used], kind="stable")
            assert (np.diff(L[used][order].astype(int)) <= 0).all(), "a rarer symbol with a shorter code"
            cost, best = int((freq.astype(np.int64) * L).sum()), optimal_cost(freq)
            assert cost >= best
            if int(freq.max()) <= 1000 and maxlen == 15:   # counts this close cannot make a tree of 286 leaves deeper than 15
                assert cost == best, "the limit does not bind: the code must be Huffman's"
            lim = limited_optimal_cost(freq, maxlen)
            assert best <= lim == cost, (freq.size, maxlen, cost, lim, best)      # package-merge: THE cheapest code under the limit
    tab = [(257 + k, e) for k, e in enumerate([0] * 8 + [1] * 4 + [2] * 4 + [3] * 4 + [4] * 4 + [5] * 4 + [0])]
    base = [3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258]
    for length in range(3, 259):
        k = max(i for i in range(29) if base[i] <= length)
        s, e, v = C.c_int(), C.c_int(), C.c_int()
        lib.twin_length_symbol(length, C.byref(s), C.byref(e), C.byref(v))
        assert (s.value, e.value, v.value) == (tab[k][0], tab[k][1], length - base[k]), length
