"""images::{load_image, save_image} and load_image_as_spherical_image (src/images.rs:7-20, :186-193) over the
library's own PNG / JPEG codecs (curvis_image_load / curvis_image_save_rgb8): the same decoders the `curvis` binary
uses for its background arguments, so a sky loaded here and one passed on the command line are the same texels."""
import ctypes as C

import numpy as np

from ._abi import CurvisError, lib
from .systems import SphericalImage


def load_image(path):
    """image::open(path) as Rgba8 (what DynamicImage::get_pixel returns): H x W x 4 uint8"""
    p = C.POINTER(C.c_uint8)()
    w, h = C.c_uint32(0), C.c_uint32(0)
    rc = lib().curvis_image_load(str(path).encode(), C.byref(p), C.byref(w), C.byref(h))
    if rc != 0:
        raise CurvisError(rc, (lib().curvis_last_error(None) or b"").decode())
    try:
        return np.ctypeslib.as_array(p, (h.value, w.value, 4)).copy()
    finally:
        lib().curvis_image_free(p)


def save_image(path, rgb, level=None):
    """DynamicImage::save as PNG for an H x W x 3 uint8 array.  level: None = the library's default, -1 = its fast
    writer (what `curvis video` uses), 0..9 = zlib; the decoded pixels are the same whatever the level."""
    rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
    if rgb.ndim != 3 or rgb.shape[2] != 3:
        raise ValueError("rgb must be HxWx3 uint8")
    if level is None:
        rc = lib().curvis_image_save_rgb8(str(path).encode(), rgb.ctypes.data, rgb.shape[1], rgb.shape[0])
    else:
        rc = lib().curvis_image_save_rgb8_level(str(path).encode(), rgb.ctypes.data, rgb.shape[1], rgb.shape[0], int(level))
    if rc != 0:
        raise CurvisError(rc, (lib().curvis_last_error(None) or b"").decode())


def save_zlib_stream(path, zlib_stream, width, height):
    """A PNG file around a finished zlib stream of Up-filtered RGB8 scanlines -- what Context.deflate_frames returns per frame:
    signature, IHDR, IDAT + CRC-32, IEND (curvis_image_save_zlib_rgb8)."""
    buf = np.frombuffer(zlib_stream, dtype=np.uint8)
    rc = lib().curvis_image_save_zlib_rgb8(str(path).encode(), buf.ctypes.data, buf.size, int(width), int(height))
    if rc != 0:
        raise CurvisError(rc, (lib().curvis_last_error(None) or b"").decode())


def load_image_as_spherical_image(path, forward=None, up=None):
    """src/images.rs:186-193: forward / up default to x / z"""
    return SphericalImage(load_image(path), forward, up)
