#!/usr/bin/env python3
"""Print the results of tools/gpu_quick2.sh (gpurun_out/)."""
import collections, csv, json, os
G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
print(open(os.path.join(G, "pytest_gpu.log")).read().strip())
for f, d in (("q_bench_v1", "q_pmc"), ("q_bench_interstellar", "q_pmc_inter")):
    b = json.loads(open(os.path.join(G, f + ".json")).read())
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(os.path.join(G, d, "pmc_counter_collection.csv"))):
        if "geodesic" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    A = {c: sum(v) / len(v) for c, v in agg.items()}
    ws = b["config"]["executed_steps_per_frame"] / 64.0
    gui = A["GRBM_GUI_ACTIVE"] / 8
    print("%-22s %.1f G  %.3f ms | VALU/ws %.1f SALU/ws %.1f busy %.3f cyc/ws %.0f Mcyc/frame %.2f" % (
        f, b["value"] / 1e3, b["ms_per_step"], A["SQ_INSTS_VALU"] / ws, A["SQ_INSTS_SALU"] / ws,
        4 * A["SQ_ACTIVE_INST_VALU"] / (1024 * gui), gui * 1024 / ws, gui / 1e6))
