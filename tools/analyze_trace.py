#!/usr/bin/env python3
"""Summarise a CURVIS_TRACE_FILE dump (u64 x 4 per wave; times in 100 MHz wall_clock64 ticks): residency over
time, per-SIMD finish times, wave durations.

  static kernel: {start, end, HW_ID, XCC_ID}
  relay kernel:  {start, end, HW_ID | XCC_ID << 32 | fresh << 40 | parked << 41 | k0 << 44, time the wave had its
                  tile}; relay workgroups that left at once (every tile finished) leave all-zero records

    python tools/analyze_trace.py gpurun_out/trace_config2.bin
"""
import collections
import sys

import numpy as np

t = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 4)
t = t[t[:, 0] != 0]
relay = bool((t[:, 3] > (1 << 40)).any())
if relay:
    flags = t[:, 2] >> np.uint64(32)
    t = t.copy()
    t_work = t[:, 3].copy()
    t[:, 3] = flags & np.uint64(0xF)
    t[:, 2] = t[:, 2] & np.uint64(0xFFFFFFFF)
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
if relay:
    fresh = ((flags >> np.uint64(8)) & np.uint64(1)).astype(bool)
    parked = ((flags >> np.uint64(9)) & np.uint64(1)).astype(bool)
    wait = (t_work - t[:, 0]).astype(np.float64) / 100e3
    print("relay kernel: %d fresh waves, %d relay waves that got a tile (%d hand-overs = parked tiles); relay wave wait for a "
          "tile (ms): median %.3f p90 %.3f max %.3f" % (fresh.sum(), (~fresh).sum(), parked.sum(),
          np.median(wait[~fresh]) if (~fresh).any() else 0, np.percentile(wait[~fresh], 90) if (~fresh).any() else 0,
          wait[~fresh].max() if (~fresh).any() else 0))
    print("last fresh wave started at %.3f ms; first hand-over at %.3f ms" % (s[fresh].max(), e[parked].min() if parked.any() else -1))
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
