"""Reads the rocprofv3 kernel trace of a `curvis video --mode efficient` run (tools/gpu_eff_trace.sh): per kernel the total and mean
duration, and how the GPU's wall time divides into: no kernel running, exactly one, several at once."""
import csv
import glob
import json
import os
import re
import sys

import numpy as np

d = sys.argv[1]
rows = []
for p in glob.glob(os.path.join(d, "trace", "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(p)):
        m = re.search(r"(?:\)::|^)(\w+)\s*(?:<|\()", r["Kernel_Name"].replace("void ", ""))
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), m.group(1) if m else r["Kernel_Name"][:40], r.get("Queue_Id", "")))
rows.sort()
t0, t1 = rows[0][0], max(r[1] for r in rows)
span = (t1 - t0) * 1e-9
try:
    summ = json.load(open(os.path.join(d, "summary.json")))
    print("run: %d frames in %.2f s = %.0f frames/s, %d worker threads; kernel timeline spans %.2f s" % (summ["frames"], summ["wall_s"], summ["frames_per_s"], len(summ["devices"]), span))
except OSError:
    print("kernel timeline spans %.2f s" % span)
by = {}
for s, e, n, q in rows:
    b = by.setdefault(n, [0, 0.0])
    b[0] += 1
    b[1] += (e - s) * 1e-9
print("| kernel | launches | total s | mean ms | share of the span |\n|---|---|---|---|---|")
for n, (c, t) in sorted(by.items(), key=lambda kv: -kv[1][1]):
    print("| %s | %d | %.3f | %.4f | %.3f |" % (n, c, t, t / c * 1e3, t / span))
# concurrency: sweep over start / end events
ev = []
for s, e, n, q in rows:
    ev.append((s, 1))
    ev.append((e, -1))
ev.sort()
depth, last, hist = 0, t0, {}
for t, dlt in ev:
    hist[depth] = hist.get(depth, 0) + (t - last)
    depth += dlt
    last = t
tot = sum(hist.values())
print("time with k kernels in flight: " + ", ".join("%d: %.3f" % (k, v / tot) for k, v in sorted(hist.items())))
print("queues seen: %d" % len({q for _, _, _, q in rows}))
