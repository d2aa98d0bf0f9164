"""end-to-end timing of `curvis image` with 8192x4096 PNG skies (decode + upload + render + encode)."""
import os, subprocess, sys, time, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from curvis_amd import pngio, skies
BIN = os.path.join(ROOT, "curvis_amd", "bin", "curvis")
d = tempfile.mkdtemp(prefix="curvis_image_")
t0 = time.perf_counter()
pngio.write_png(os.path.join(d, "pos.png"), skies.checker(8192, 4096, 1)[..., :3], level=1)
pngio.write_png(os.path.join(d, "neg.png"), skies.checker(8192, 4096, 2)[..., :3], level=1)
print("wrote 2 skies (%.1f MB, %.1f MB) in %.1f s" % (os.path.getsize(os.path.join(d, "pos.png")) / 1e6, os.path.getsize(os.path.join(d, "neg.png")) / 1e6, time.perf_counter() - t0))
open(os.path.join(d, "cam.toml"), "w").write("resolution_x = 1920\nresolution_y = 1080\ndiagonal = 43.0\nfocal_length = 15.0\n")
for mode in ("efficient", "brute"):
    for rep in range(2):
        t0 = time.perf_counter()
        r = subprocess.run([BIN, "image", os.path.join(d, "pos.png"), os.path.join(d, "neg.png"), d, "-c", os.path.join(d, "cam.toml"), "--mode", mode, "--stats", os.path.join(d, "st.json")], capture_output=True, text=True)
        dt = time.perf_counter() - t0
        print("curvis image --mode %s: rc %d, %.2f s wall; stats %s" % (mode, r.returncode, dt, open(os.path.join(d, "st.json")).read().strip()[-60:]))
