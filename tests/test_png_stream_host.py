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
