#!/usr/bin/env python3
"""Inputs of tools/ref_dump/dump_golden.rs: the procedural checker skies of the committed fixtures (tests/golden/make_golden.py:
512x256, seeds of tests/common.make_skies) and the two "index" skies whose colours encode the texel -- see dump_golden.rs.
    python tools/ref_dump/make_inputs.py [dir]      (default tools/ref_dump/inputs)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
W, H = 512, 256


def index_sky(negative):
    x = np.arange(W, dtype=np.uint32)[None, :].repeat(H, 0)
    y = np.arange(H, dtype=np.uint32)[:, None].repeat(W, 1)
    img = np.empty((H, W, 4), np.uint8)
    img[..., 0] = x & 255
    img[..., 1] = (x >> 8) | ((y >> 8) << 4) | 64 | (128 if negative else 0)
    img[..., 2] = y & 255
    img[..., 3] = 255
    return img


def decode_index(rgb):
    """(tx, ty, code) of every pixel of a render on the index skies; code 0 = black = not escaped"""
    r, g, b = (rgb[..., k].astype(np.uint32) for k in range(3))
    esc = (g & 64) != 0
    return np.where(esc, r | ((g & 15) << 8), 0), np.where(esc, b | (((g >> 4) & 3) << 8), 0), np.where(esc, np.where(g & 128, -1, 1), 0)


def main():
    import common
    from curvis_amd import pngio
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tools", "ref_dump", "inputs")
    os.makedirs(out, exist_ok=True)
    sp, sn = common.make_skies(W, H, "check")
    for name, img in (("pos.png", sp), ("neg.png", sn), ("index_pos.png", index_sky(False)), ("index_neg.png", index_sky(True))):
        pngio.write_png(os.path.join(out, name), img)
        print("wrote", os.path.join(out, name))


if __name__ == "__main__":
    main()
