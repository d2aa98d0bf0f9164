#!/usr/bin/env python3
"""Device PNG front end against Python's zlib on random frames: random widths and heights (staged rows of a multiple of 64
bytes, 16-byte-aligned rows, ragged rows; 1-pixel frames up to 1000 pixels wide), 1-6 frames per call, skies of white noise /
hash checker / smooth gradient (streams from ~1.0 x down to ~0.01 x of the pixels), cap 0 (black frame) now and then.
Every stream must inflate (Adler-32 checked by zlib) to the Up-filtered scanlines of the frame a download returns.
Since round 6 ONE device path serves every width (the three-pass kernels that ragged widths took, and that staged frames were
compared with, are gone); frames of up to 120 000 bytes are also checked TOKEN BY TOKEN against the restated stream format
(tests/test_gpu_png.py deflate_tokens / model_tokens), and every chunk CRC-32 against zlib's.
python tools/gpu_png_fuzz.py [cases] [seed]   -> profiles/round6_png_fuzz.txt"""
import os, sys, time, zlib
import numpy as np
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
sys.path.insert(0, os.path.join(root, "tests"))
import curvis_amd
from curvis_amd import skies
import test_gpu_png as T
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
ctx = curvis_amd.Context(0)
noise = []
for _ in range(2):
    t = rng.integers(0, 256, (1024, 2048, 4), dtype=np.uint8); t[..., 3] = 255; noise.append(t)
SKIES = {"noise": noise, "check": [skies.checker(1024, 512, 3), skies.checker(1024, 512, 4)], "smooth": [skies.smooth(2048, 1024, 128), skies.smooth(2048, 1024, 32)]}
bad = 0; token_checked = 0; t0 = time.time(); kinds = {"staged": 0, "aligned": 0, "ragged": 0}; ratio_min, ratio_max = 9.0, 0.0
for it in range(N):
    mode = it % 3
    if mode == 0: w = int(rng.choice([64, 128, 192, 256, 320, 448, 640, 960]))                                                          # 3w % 64 == 0
    elif mode == 1: w = int(rng.integers(1, 60)) * 16                                                                                   # 3w % 16 == 0
    else: w = int(rng.integers(1, 1000))
    h = int(rng.integers(1, 400)); nf = int(rng.integers(1, 7))
    if w * h * nf > 1_500_000: nf = 1
    kinds["staged" if (3 * w) % 64 == 0 else "aligned" if (3 * w) % 16 == 0 else "ragged"] += 1
    sk = str(rng.choice(list(SKIES)))
    ctx.set_sky(0, curvis_amd.SphericalImage(SKIES[sk][0])); ctx.set_sky(1, curvis_amd.SphericalImage(SKIES[sk][1]))
    m = curvis_amd.EllisMetric(1.0) if rng.random() < 0.7 else curvis_amd.FlatSphericalMetric()
    cams = [curvis_amd.Camera((0.0, float(rng.uniform(1.5, 9)), float(rng.uniform(0.5, 2.6)), float(rng.uniform(0, 6))),
                              (float(rng.normal()) - 1.0, float(rng.normal()), float(rng.normal())), (0.0, 0.0, 1.0), float(rng.uniform(8, 40)), 43.0, w, h) for _ in range(nf)]
    cap = 0 if rng.random() < 0.05 else 3000
    rgb, _ = ctx.render_brute(m, cams, cap, 100.0, 0.05)
    streams, _, crcs = ctx.deflate_frames_crc(w, h, nf)
    if crcs is None or crcs != [zlib.crc32(b"IDAT" + z) for z in streams]:   # the chunk CRC-32 the device computed
        bad += 1; print("CRC DIFFERS case %d: %dx%d x%d" % (it, w, h, nf), flush=True)
    if w * h * 3 <= 120_000:
        token_checked += 1
        for k in range(nf):
            try:
                same = T.deflate_tokens(streams[k]) == T.model_tokens(rgb[k])
            except AssertionError as exc:
                same = False; print("TOKEN READER:", exc)
            if not same:
                bad += 1; print("TOKENS DIFFER case %d frame %d: %dx%d x%d %s cap %d" % (it, k, w, h, nf, sk, cap), flush=True)
    for k in range(nf):
        try:
            raw = np.frombuffer(zlib.decompress(streams[k]), np.uint8).reshape(h, 3 * w + 1)
            ok = bool((raw[:, 0] == 2).all()) and np.array_equal(np.cumsum(raw[:, 1:].astype(np.uint32), axis=0).astype(np.uint8).reshape(h, w, 3), rgb[k])
        except Exception as exc:
            ok = False; print("EXC", exc)
        r = len(streams[k]) / (w * h * 3)
        if w * h > 20000: ratio_min, ratio_max = min(ratio_min, r), max(ratio_max, r)
        if not ok:
            bad += 1; print("MISMATCH case %d frame %d: %dx%d x%d %s cap %d" % (it, k, w, h, nf, sk, cap), flush=True)
print("cases %d (%s; %d of them also token by token), stream / pixels between %.4f and %.3f (frames > 20 000 pixels), mismatches %d, %.0f s" % (
    N, ", ".join("%s %d" % kv for kv in kinds.items()), token_checked, ratio_min, ratio_max, bad, time.time() - t0))
sys.exit(1 if bad else 0)
