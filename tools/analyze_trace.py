#!/usr/bin/env python3
"""Summarise a CURVIS_TRACE_FILE dump of the static kernel (u64 x 4 per wave: start, end in 100 MHz
wall_clock64 ticks, HW_ID, XCC_ID): residency over time, per-SIMD finish times, wave durations.

    python tools/analyze_trace.py gpurun_out/trace_config2.bin
"""
import collections
import sys

import numpy as np

t = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 4)
t0 = t[:, 0].min()
s = (t[:, 0] - t0).astype(np.float64) / 100e3   # ms
e = (t[:, 1] - t0).astype(np.float64) / 100e3
hw = t[:, 2].astype(np.uint32)
xcc = t[:, 3].astype(np.uint32) & 0xF
simd, cu, sh, se = (hw >> 4) & 3, (hw >> 8) & 0xF, (hw >> 12) & 1, (hw >> 13) & 7
key = ((((xcc * 8 + se) * 2 + sh) * 16 + cu) * 4 + simd).astype(np.int64)
fin = collections.defaultdict(float)
cnt = collections.Counter()
for k, ee in zip(key, e):
    fin[k] = max(fin[k], ee)
    cnt[k] += 1
f = np.array(list(fin.values()))
c = np.array([cnt[k] for k in fin])
print("waves %d on %d SIMDs; kernel span %.3f ms; last wave started at %.3f ms" % (len(t), len(f), e.max(), s.max()))
print("waves per SIMD: min %d median %d max %d" % (c.min(), np.median(c), c.max()))
print("per-SIMD finish (ms): min %.2f p10 %.2f median %.2f p90 %.2f max %.2f" % (
    f.min(), np.percentile(f, 10), np.median(f), np.percentile(f, 90), f.max()))
d = e - s
print("wave duration (ms): min %.2f median %.2f p90 %.2f max %.2f" % (d.min(), np.median(d), np.percentile(d, 90), d.max()))
print("resident waves per SIMD over time:")
for x in np.linspace(0, e.max(), 21):
    n = ((s <= x) & (e > x)).sum()
    print("  t = %5.2f ms  %.2f" % (x, n / len(f)))
