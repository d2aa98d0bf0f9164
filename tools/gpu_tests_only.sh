#!/bin/bash
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT; cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q -s 2>&1 | tail -25 > $OUT/pytest_gpu.log
python - > $OUT/eff_timing.log 2>&1 <<'PY'
import sys, time, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import curvis_amd, common
from curvis_amd import skies
ctx = curvis_amd.Context(0)
sp, sn = skies.smooth(8192, 4096, 128), skies.smooth(8192, 4096, 32)
ctx.set_sky(0, curvis_amd.SphericalImage(sp)); ctx.set_sky(1, curvis_amd.SphericalImage(sn))
m = curvis_amd.EllisMetric(1.0)
for res in ((960, 540), (1920, 1080)):
    cam = curvis_amd.Camera((0, 5, np.pi/2, 0), (-1, 0, 0), (0, 0, 1), 15, 43, res[0], res[1])
    for _ in range(2):
        t0 = time.perf_counter(); rgb, st = ctx.render_efficient(m, cam, 40000, 100.0, 0.05, 100, 100, 1e-5, 1e-5, download=False); dt = time.perf_counter() - t0
    info = ctx.sampling_info(0)
    print(res, 'single frame: wall %.2f ms, sampling kernels %.2f ms, pixel kernel %.3f ms, samples %d calls %d steps %d rounds %d' % (dt*1e3, st.integrate_ms, st.shade_ms, info.n_samples, info.calls, info.steps, info.rounds))
cams = [curvis_amd.Camera((0, 3, np.pi/2, 0.026*k), (-1, 0, 0), (0, 0, 1), 15, 43, 1920, 1080) for k in range(32)]
for _ in range(2):
    t0 = time.perf_counter(); rgb, st = ctx.render_efficient(m, cams, 40000, 100.0, 0.05, 100, 100, 1e-5, 1e-5, download=False); dt = time.perf_counter() - t0
print('batch of 32 1080p frames: wall %.2f ms (%.2f ms/frame), sampling kernels %.2f ms, pixel kernel %.3f ms' % (dt*1e3, dt*1e3/32, st.integrate_ms, st.shade_ms))
PY
