"""curvis_amd.video_merge: frame ordering of the reference's utils/video_merge.py (integer after the underscore),
lossless containers (Y4M 4:4:4, APNG) written without a codec library."""
import os
import struct
import zlib

import numpy as np
import pytest

from curvis_amd import pngio, video_merge


def _frames(tmp, n=12, w=10, h=6):
    rng = np.random.default_rng(3)
    imgs = []
    for k in range(n):
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        pngio.write_png(os.path.join(tmp, "frame_%d.png" % k), img)
        imgs.append(img)
    return imgs


def test_frames_are_ordered_by_index_not_by_name(tmp_path):
    _frames(str(tmp_path))
    names = [os.path.basename(f) for f in video_merge.frame_files(str(tmp_path))]
    assert names == ["frame_%d.png" % k for k in range(12)]          # frame_10 after frame_9, not after frame_1
    with pytest.raises(FileNotFoundError):
        video_merge.frame_files(str(tmp_path / "nope"))


def test_y4m_layout(tmp_path):
    imgs = _frames(str(tmp_path), n=3)
    path = video_merge.merge(str(tmp_path), fmt="y4m", fps=24, verbose=False)
    data = open(path, "rb").read()
    head, rest = data.split(b"\n", 1)
    assert head == b"YUV4MPEG2 W10 H6 F24:1 Ip A1:1 C444 XCOLORRANGE=FULL"
    assert len(rest) == 3 * (6 + 3 * 60)
    y = np.frombuffer(rest[6:6 + 60], np.uint8).reshape(6, 10).astype(int)
    want = np.rint(0.299 * imgs[0][..., 0] + 0.587 * imgs[0][..., 1] + 0.114 * imgs[0][..., 2]).astype(int)
    assert np.array_equal(y, want)


def test_apng_frames_decode_to_the_input_pixels(tmp_path):
    imgs = _frames(str(tmp_path), n=4)
    path = video_merge.merge(str(tmp_path), fmt="apng", fps=30, verbose=False)
    data = open(path, "rb").read()
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    pos, chunks = 8, []
    while pos < len(data):
        n, tag = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        assert struct.unpack(">I", data[pos + 8 + n:pos + 12 + n])[0] == (zlib.crc32(tag + body) & 0xFFFFFFFF)
        chunks.append((tag, body))
        pos += 12 + n
    assert [t for t, _ in chunks][:3] == [b"IHDR", b"acTL", b"fcTL"] and chunks[-1][0] == b"IEND"
    assert struct.unpack(">II", chunks[1][1]) == (4, 0)
    frames = [b for t, b in chunks if t == b"IDAT"] + [b[4:] for t, b in chunks if t == b"fdAT"]
    assert len(frames) == 4
    for img, comp in zip(imgs, frames):
        raw = np.frombuffer(zlib.decompress(comp), np.uint8).reshape(6, 1 + 30)
        assert np.all(raw[:, 0] == 0) and np.array_equal(raw[:, 1:].reshape(6, 10, 3), img)
    # sequence numbers of fcTL / fdAT are consecutive
    seqs = [struct.unpack(">I", b[:4])[0] for t, b in chunks if t in (b"fcTL", b"fdAT")]
    assert seqs == list(range(len(seqs)))
