"""Merge the frames written by `curvis video` (``<out>/tmp/frame_{k}.png``) into one video file.

Counterpart of the reference's ``utils/video_merge.py`` (an example script built on OpenCV's mp4v writer):
same inputs and ordering -- every ``*.png`` of the tmp folder, sorted by the integer after the underscore,
a frame rate, an output folder and a file stem -- but without a codec dependency (neither OpenCV nor ffmpeg
exists in this image).  Two lossless containers are written with the standard library + numpy only:

* ``.y4m``  YUV4MPEG2, 4:4:4 full-range BT.601 (plays in mpv / vlc / ffplay, `ffmpeg -i x.y4m x.mp4` re-encodes);
* ``.apng`` animated PNG (plays in browsers), pixels bit-identical to the frames.

    python -m curvis_amd.video_merge <tmp folder> [--out DIR] [--stem merged_video] [--fps 30] [--format y4m|apng]
"""
import argparse
import os
import struct
import zlib

import numpy as np

from . import pngio


def frame_files(tmp_folder):
    """all png files of the folder, ordered by the integer in ``<name>_<k>.png`` (reference: sorted by
    int(stem.split('_')[1]))"""
    if not os.path.isdir(tmp_folder):
        raise FileNotFoundError("Folder %s does not exist." % tmp_folder)
    files = [f for f in os.listdir(tmp_folder) if f.lower().endswith(".png")]
    if not files:
        raise FileNotFoundError("No png files found within %s" % tmp_folder)
    files.sort(key=lambda f: int(os.path.splitext(f)[0].split("_")[1]))
    return [os.path.join(tmp_folder, f) for f in files]


def _rgb_to_yuv444(rgb):
    """full-range BT.601, rounded to nearest"""
    r, g, b = [rgb[..., k].astype(np.float64) for k in range(3)]
    y = 0.299 * r + 0.587 * g + 0.114 * b
    u = -0.168736 * r - 0.331264 * g + 0.5 * b + 128.0
    v = 0.5 * r - 0.418688 * g - 0.081312 * b + 128.0
    return [np.clip(np.rint(p), 0, 255).astype(np.uint8) for p in (y, u, v)]


def write_y4m(path, files, fps):
    num, den = (int(fps), 1) if float(fps).is_integer() else (int(round(fps * 1000)), 1000)
    first = pngio.read_png(files[0])
    h, w = first.shape[:2]
    with open(path, "wb") as f:
        f.write(("YUV4MPEG2 W%d H%d F%d:%d Ip A1:1 C444 XCOLORRANGE=FULL\n" % (w, h, num, den)).encode())
        for name in files:
            img = first if name == files[0] else pngio.read_png(name)
            if img.shape[:2] != (h, w):
                raise ValueError("%s: frame size differs from the first frame" % name)
            f.write(b"FRAME\n")
            for plane in _rgb_to_yuv444(img[..., :3]):
                f.write(plane.tobytes())
    return w, h


def _chunk(tag, data):
    return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)


def write_apng(path, files, fps, level=1):
    num, den = (1, int(fps)) if float(fps).is_integer() else (1000, int(round(fps * 1000)))
    first = pngio.read_png(files[0])
    h, w = first.shape[:2]
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n")
        f.write(_chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)))
        f.write(_chunk(b"acTL", struct.pack(">II", len(files), 0)))
        seq = 0
        for k, name in enumerate(files):
            img = first if k == 0 else pngio.read_png(name)
            if img.shape[:2] != (h, w):
                raise ValueError("%s: frame size differs from the first frame" % name)
            rgb = np.ascontiguousarray(img[..., :3])
            raw = np.concatenate([np.zeros((h, 1), np.uint8), rgb.reshape(h, w * 3)], axis=1).tobytes()
            data = zlib.compress(raw, level)
            f.write(_chunk(b"fcTL", struct.pack(">IIIIIHHBB", seq, w, h, 0, 0, num, den, 0, 0)))
            seq += 1
            if k == 0:
                f.write(_chunk(b"IDAT", data))
            else:
                f.write(_chunk(b"fdAT", struct.pack(">I", seq) + data))
                seq += 1
        f.write(_chunk(b"IEND", b""))
    return w, h


def merge(tmp_folder, out_folder=None, stem="merged_video", fps=30.0, fmt="y4m", verbose=True):
    files = frame_files(tmp_folder)
    out_folder = out_folder or os.path.dirname(os.path.abspath(tmp_folder))
    if not os.path.isdir(out_folder):
        raise FileNotFoundError("Folder %s does not exist." % out_folder)
    path = os.path.join(out_folder, stem + "." + fmt)
    if verbose:
        print("Merging %d frames to %s" % (len(files), path))
    (write_y4m if fmt == "y4m" else write_apng)(path, files, fps)
    return path


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("tmp_folder")
    ap.add_argument("--out", default=None)
    ap.add_argument("--stem", default="merged_video")
    ap.add_argument("--fps", type=float, default=30.0)
    ap.add_argument("--format", choices=["y4m", "apng"], default="y4m")
    a = ap.parse_args(argv)
    merge(a.tmp_folder, a.out, a.stem, a.fps, a.format)


if __name__ == "__main__":
    main()
