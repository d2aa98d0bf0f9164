import json, csv, collections, sys, os
G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
print(open(os.path.join(G, "pytest_gpu.log")).read().strip().splitlines()[-1])
for f in sorted(os.listdir(G)):
    if f.startswith("q_bench") and f.endswith(".json"):
        try:
            d = json.load(open(os.path.join(G, f)))
            print("%-28s value %9.1f  ms/step %7.3f  integ %7.3f shade %.3f frac %.4f" % (f, d["value"], d["ms_per_step"], d["roofline"]["kernel_ms_avg"], d["roofline"]["shade_kernel_ms_avg"], d["roofline"]["frac"]))
        except Exception as e:
            print(f, "ERR", e)
p = os.path.join(G, "q_pmc", "pmc_counter_collection.csv")
if os.path.exists(p):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(p)):
        agg[(r["Kernel_Name"][:46], r["Counter_Name"])].append(float(r["Counter_Value"]))
    ws = 4048633421 / 64
    for (k, c), v in sorted(agg.items()):
        if "geodesic" in k:
            m = sum(v) / len(v)
            print("%-46s %-20s %.5g  (%.1f / wave-step)" % (k, c, m, m / ws))
