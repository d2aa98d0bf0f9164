mkdir -p gpurun_out
python - <<'PY'
import numpy as np, os
from PIL import Image
rng = np.random.default_rng(5)
yy = np.linspace(-1, 1, 4096)[:, None]; xx = np.linspace(0, 1, 8192)[None, :]
band = np.exp(-(yy * 3 + 0.3 * np.sin(xx * 6.28)) ** 2) * 60
a = np.zeros((4096, 8192, 3), np.float32) + band[..., None] * np.array((1.0, 0.9, 0.8), np.float32)
n = 200000
ys, xs, br = rng.integers(0, 4096, n), rng.integers(0, 8192, n), rng.pareto(2.0, n) * 40
for c in range(3): np.add.at(a[..., c], (ys, xs), br * rng.uniform(0.7, 1.0, n))
a += rng.normal(0, 1.5, a.shape)
Image.fromarray(np.clip(a, 0, 255).astype(np.uint8)).save('/dev/shm/stars.png')
import sys; sys.path.insert(0, '.')
from curvis_amd import pngio, skies
pngio.write_png('/dev/shm/smooth.png', skies.smooth(8192, 4096, 128))
PY
for rep in 1 2 3 4; do echo -n "no populate: "; CURVIS_NO_POPULATE=1 build/pngab/b_new3 /dev/shm/stars.png /dev/shm/smooth.png | tr '\n' ' '; echo; echo -n "populate:    "; build/pngab/b_new3 /dev/shm/stars.png /dev/shm/smooth.png | tr '\n' ' '; echo; done > gpurun_out/png_decode_ab3.txt 2>&1
echo "two files at once (as the binary decodes them):" >> gpurun_out/png_decode_ab3.txt
for rep in 1 2 3; do for m in 1 0; do if [ $m = 1 ]; then export CURVIS_NO_POPULATE=1; else unset CURVIS_NO_POPULATE; fi; s=$(date +%s.%N); build/pngab/b_new3 /dev/shm/stars.png > /dev/null & build/pngab/b_new3 /dev/shm/stars.png > /dev/null; wait; e=$(date +%s.%N); echo "no_populate=$m pair of star maps: $(echo "($e - $s)*1000" | bc) ms"; done; done >> gpurun_out/png_decode_ab3.txt 2>&1
cat gpurun_out/png_decode_ab3.txt
