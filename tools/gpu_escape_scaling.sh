#!/bin/bash
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out; cd /tmp
cat > /tmp/esc.py <<'PY'
import os, sys, time
import numpy as np
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import curvis_amd
ctx = curvis_amd.Context(0)
m = curvis_amd.EllisMetric(1.0)
for n in (65536, 131072, 262144):
    al = np.full(n, 0.7)
    ctx.compute_escape_angles_range(m, 5.0, al, 0.05, 40000, 100.0)
    t0 = time.perf_counter(); ctx.compute_escape_angles_range(m, 5.0, al, 0.05, 40000, 100.0); print(n, (time.perf_counter() - t0) * 1e3, "ms", flush=True)
PY
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/esc_prof -o esc -- python /tmp/esc.py > $OUT/esc_prof.log 2>&1
grep -E "escape_angle" $OUT/esc_prof/esc_kernel_trace.csv | awk -F, '{print $0}' | cut -c1-400 | head -8
