"""Procedural equirectangular sky textures (synthetic inputs for tests and bench.py).

S_smooth: R = floor(255*x/W), G = floor(255*y/H), B = 128 (+l) / 32 (-l): adjacent texels differ by
          <= 1 LSB per channel, so a +-1-texel disagreement is <= 1 per channel.
S_check:  64-px checkerboard cells with a per-cell xorshift32 colour: any texel-index disagreement
          across a cell border is visible, used to measure exact-index agreement.
"""
import numpy as np


def smooth(width, height, blue):
    x = (np.arange(width, dtype=np.uint64) * 255 // width).astype(np.uint8)
    y = (np.arange(height, dtype=np.uint64) * 255 // height).astype(np.uint8)
    img = np.empty((height, width, 4), dtype=np.uint8)
    img[..., 0] = x[None, :]
    img[..., 1] = y[:, None]
    img[..., 2] = blue
    img[..., 3] = 255
    return img


def _xorshift32(v):
    v = v.astype(np.uint32)
    v ^= (v << np.uint32(13)) & np.uint32(0xFFFFFFFF)
    v ^= v >> np.uint32(17)
    v ^= (v << np.uint32(5)) & np.uint32(0xFFFFFFFF)
    return v


def checker(width, height, seed=0xC0FFEE, cell=64):
    cx = np.arange(width, dtype=np.uint32) // cell
    cy = np.arange(height, dtype=np.uint32) // cell
    ncx = (width + cell - 1) // cell
    cid = cy[:, None] * np.uint32(ncx) + cx[None, :]
    h = _xorshift32(np.uint32(seed) ^ (cid + np.uint32(1)))
    h = _xorshift32(h)
    img = np.empty((height, width, 4), dtype=np.uint8)
    img[..., 0] = (h & 0xFF).astype(np.uint8)
    img[..., 1] = ((h >> 8) & 0xFF).astype(np.uint8)
    img[..., 2] = ((h >> 16) & 0xFF).astype(np.uint8)
    img[..., 3] = 255
    return img
