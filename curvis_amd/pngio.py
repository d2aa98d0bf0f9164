"""Small PNG helpers for tests and tools (stdlib zlib + numpy): write 8-bit grey/RGB/RGBA, read 8-bit
non-interlaced PNGs.  The product's PNG codec is curvis_amd/csrc/host/png_io.h; this is test plumbing."""
import struct
import zlib

import numpy as np


def _chunk(tag, data):
    return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)


def write_png(path, img, level=6):
    img = np.ascontiguousarray(img)
    if img.ndim == 2:
        img = img[:, :, None]
    h, w, ch = img.shape
    ctype = {1: 0, 2: 4, 3: 2, 4: 6}[ch]
    depth = 8 if img.dtype == np.uint8 else 16
    if depth == 16:
        img = img.astype(">u2")
    raw = b"".join(b"\x00" + img[y].tobytes() for y in range(h))
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n")
        f.write(_chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, 0)))
        f.write(_chunk(b"IDAT", zlib.compress(raw, level)))
        f.write(_chunk(b"IEND", b""))


def read_png(path):
    data = open(path, "rb").read()
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat, w = 8, b"", None
    while pos < len(data):
        n, tag = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        # every chunk carries the CRC-32 of type + data; a reader that skipped it would not notice a writer that got it wrong
        if zlib.crc32(tag + body) != struct.unpack(">I", data[pos + 8 + n:pos + 12 + n])[0]:
            raise ValueError("%s: CRC mismatch in chunk %s" % (path, tag.decode("latin-1")))
        if tag == b"IHDR":
            w, h, depth, ctype, _, _, inter = struct.unpack(">IIBBBBB", body)
            assert depth == 8 and inter == 0 and ctype in (0, 2, 4, 6)
        elif tag == b"IDAT":
            idat += body
        pos += 12 + n
    ch = {0: 1, 2: 3, 4: 2, 6: 4}[ctype]
    raw = np.frombuffer(zlib.decompress(idat), dtype=np.uint8).reshape(h, w * ch + 1)
    out = np.zeros((h, w * ch), dtype=np.uint8)
    for y in range(h):
        ft, line = int(raw[y, 0]), raw[y, 1:].astype(np.int32)
        up = out[y - 1].astype(np.int32) if y else np.zeros(w * ch, np.int32)
        if ft == 0:
            cur = line
        elif ft == 2:
            cur = (line + up) & 255
        else:
            cur = np.zeros(w * ch, np.int32)
            for x in range(w * ch):
                a = cur[x - ch] if x >= ch else 0
                b = up[x]
                c = up[x - ch] if x >= ch else 0
                if ft == 1:
                    pr = a
                elif ft == 3:
                    pr = (a + b) >> 1
                else:
                    p = a + b - c
                    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
                    pr = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
                cur[x] = (line[x] + pr) & 255
        out[y] = cur.astype(np.uint8)
    return out.reshape(h, w, ch) if ch > 1 else out.reshape(h, w)
