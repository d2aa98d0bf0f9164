#!/usr/bin/env python3
"""Assemble profiles/roundN_* from gpurun_out/final (output of tools/gpu_session_final.sh)."""
import collections
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "gpurun_out", "final")
P = os.path.join(ROOT, "profiles")
rnd = sys.argv[1] if len(sys.argv) > 1 else "round1"


def pmc(kind):
    rows = list(csv.DictReader(open(os.path.join(G, "pmc_%s" % kind, "pmc_counter_collection.csv"))))
    agg = collections.defaultdict(list)
    for r in rows:
        name = r["Kernel_Name"]
        short = "geodesic_static" if "geodesic_static" in name else "geodesic_persistent" if "geodesic_persistent" in name \
            else "geodesic_relay" if "geodesic_relay" in name else "shade_kernel" if "shade_kernel" in name else None
        if short:
            agg[(short, r["Counter_Name"])].append(float(r["Counter_Value"]))
    # median over the dispatches of the run: one launch in a dozen shows a several-fold FETCH_SIZE (first touch of
    # a buffer by that process); the mean of three or four launches would report that, not the kernel
    return {k: sorted(v)[len(v) // 2] if len(v) % 2 else 0.5 * (sorted(v)[len(v) // 2 - 1] + sorted(v)[len(v) // 2]) for k, v in agg.items()}


bench = json.loads(open(os.path.join(G, "bench_default.json")).read().strip().splitlines()[-1])
steps = bench["config"]["executed_steps_per_frame"]
rays = bench["config"]["rays_per_frame"]
ws = steps / 64.0
sq, fe, wr = pmc("sq"), pmc("fetch"), pmc("write")
kern = ("geodesic_relay" if ("geodesic_relay", "SQ_INSTS_VALU") in sq else
        "geodesic_static" if ("geodesic_static", "SQ_INSTS_VALU") in sq else "geodesic_persistent")
KIB = 1024.0
fetch_i = fe[(kern, "FETCH_SIZE")] * KIB * 2   # gfx950: FETCH_SIZE counts 128-B requests as 64 B (MI355X_MICROARCH.md)
write_i = wr[(kern, "WRITE_SIZE")] * KIB
fetch_s = fe.get(("shade_kernel", "FETCH_SIZE"), 0.0) * KIB * 2   # absent when shading is fused into the epilogue
write_s = wr.get(("shade_kernel", "WRITE_SIZE"), 0.0) * KIB
traffic = {
    "ellis_1920x1080_cap4096_%s" % kern: {
        "integrate_kernel_bytes": int(fetch_i + write_i), "integrate_fetch_bytes": int(fetch_i),
        "integrate_write_bytes": int(write_i), "shade_kernel_bytes": int(fetch_s + write_s),
        "algorithmic_bytes": 7 * rays,
        "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, KiB -> bytes, FETCH doubled per the "
                  "gfx950 note in MI355X_MICROARCH.md), profiles/%s_pmc_*.csv" % rnd,
    }
}
_gui = sq[(kern, "GRBM_GUI_ACTIVE")] / 8.0
traffic["ellis_1920x1080_cap4096_%s" % kern].update({
    "valu_busy": round(4 * sq[(kern, "SQ_ACTIVE_INST_VALU")] / (1024 * _gui), 4),
    "valu_instr_per_wave_step": round(sq[(kern, "SQ_INSTS_VALU")] / ws, 1),
    "salu_instr_per_wave_step": round(sq[(kern, "SQ_INSTS_SALU")] / ws, 1),
    "shader_cycles_per_frame": int(_gui),
})
# second entry: the Interstellar metric at the same frame size (configs[2]/[4] use this kernel instantiation)
inter_path = os.path.join(G, "bench_interstellar_1080p.json")
inter = None
if os.path.exists(inter_path) and os.path.exists(os.path.join(G, "pmc_sq_inter", "pmc_counter_collection.csv")):
    inter = json.loads(open(inter_path).read().strip().splitlines()[-1])
    isq, ife, iwr = pmc("sq_inter"), pmc("fetch_inter"), pmc("write_inter")
    iws = inter["config"]["executed_steps_per_frame"] / 64.0
    igui = isq[(kern, "GRBM_GUI_ACTIVE")] / 8.0
    ifetch, iwrite = ife[(kern, "FETCH_SIZE")] * KIB * 2, iwr[(kern, "WRITE_SIZE")] * KIB
    traffic["interstellar_1920x1080_cap4096_%s" % kern] = {
        "integrate_kernel_bytes": int(ifetch + iwrite), "integrate_fetch_bytes": int(ifetch),
        "integrate_write_bytes": int(iwrite), "shade_kernel_bytes": 0, "algorithmic_bytes": 7 * rays,
        "source": "as above, profiles/%s_pmc_*_interstellar.csv" % rnd,
        "valu_busy": round(4 * isq[(kern, "SQ_ACTIVE_INST_VALU")] / (1024 * igui), 4),
        "valu_instr_per_wave_step": round(isq[(kern, "SQ_INSTS_VALU")] / iws, 1),
        "salu_instr_per_wave_step": round(isq[(kern, "SQ_INSTS_SALU")] / iws, 1),
        "shader_cycles_per_frame": int(igui),
    }
    for kind in ("sq", "fetch", "write"):
        shutil.copy(os.path.join(G, "pmc_%s_inter" % kind, "pmc_counter_collection.csv"),
                    os.path.join(P, "%s_pmc_%s_interstellar.csv" % (rnd, kind)))
    shutil.copy(inter_path, os.path.join(P, "%s_bench_interstellar_1080p.json" % rnd))
json.dump(traffic, open(os.path.join(P, "traffic.json"), "w"), indent=1)

for kind in ("sq", "fetch", "write"):
    shutil.copy(os.path.join(G, "pmc_%s" % kind, "pmc_counter_collection.csv"), os.path.join(P, "%s_pmc_%s.csv" % (rnd, kind)))
shutil.copy(os.path.join(G, "stats", "bench_kernel_stats.csv"), os.path.join(P, "%s_kernel_stats.csv" % rnd))
shutil.copy(os.path.join(G, "ubench.log"), os.path.join(P, "%s_ubench_fp64.txt" % rnd))
shutil.copy(os.path.join(G, "configs.md"), os.path.join(P, "%s_configs.md" % rnd))
for src_name, dst_name in (("cli_video.txt", "cli_video.txt"), ("efficient_sweep.txt", "efficient_sweep.txt"),
                           ("tail.txt", "frames_per_launch.txt"), ("deep_fuzz.txt", "deep_fuzz.txt"), ("efficient_phases.txt", "efficient_phases.txt"), ("wave_trace_config2.txt", "wave_trace_config2.txt"), ("wave_trace_config2_relay.txt", "wave_trace_config2_relay.txt")):
    if os.path.exists(os.path.join(G, src_name)):
        shutil.copy(os.path.join(G, src_name), os.path.join(P, "%s_%s" % (rnd, dst_name)))
for n in ("bench_default", "bench_persistent", "bench_strict", "bench_download", "bench_config3"):
    src = os.path.join(G, n + ".json")
    if os.path.exists(src):
        shutil.copy(src, os.path.join(P, "%s_%s.json" % (rnd, n)))

gui = sq[(kern, "GRBM_GUI_ACTIVE")] / 8.0
kstats = {r["Name"]: r for r in csv.DictReader(open(os.path.join(G, "stats", "bench_kernel_stats.csv")))}
kname = [k for k in kstats if kern in k][0]
avg_ms = float(kstats[kname]["AverageNs"]) / 1e6
lines = []
lines.append("# %s profile summary (MI355X gfx950, ROCm 7.2) -- `python bench.py` (configs[1]: Ellis 1920x1080 cap 4096)\n" % rnd)
lines.append("Collected by tools/gpu_session_final.sh; assembled by tools/make_profiles.py.\n")
lines.append("## bench.py line (un-profiled run)\n```\n%s\n```\n" % json.dumps({k: bench[k] for k in ("value", "unit", "ms_per_step", "roofline", "cpu_baseline") if k in bench}, indent=1))
lines.append("## rocprofv3 --kernel-trace --stats\n```\n%s```\n" % open(os.path.join(G, "stats", "bench_kernel_stats.csv")).read())
lines.append("## PMC (median over the dispatches; SQ set, FETCH_SIZE and WRITE_SIZE in separate passes)\n```")
for d in (sq, fe, wr):
    for (k, c), v in sorted(d.items()):
        lines.append("%-22s %-22s %.6g" % (k, c, v))
lines.append("```\n")
lines.append("## Derived for the integration kernel `%s`\n" % kern)
lines.append("| quantity | value |\n|---|---|")
lines.append("| executed Euler steps / frame | %d (= %.4g wave-steps of 64 lanes) |" % (steps, ws))
lines.append("| average duration, kernel-trace | %.3f ms (bench.py HIP events: %.3f ms) |" % (avg_ms, bench["roofline"]["kernel_ms_avg"]))
lines.append("| VALU instructions / wave-step | %.1f |" % (sq[(kern, "SQ_INSTS_VALU")] / ws))
lines.append("| SALU instructions / wave-step | %.1f |" % (sq[(kern, "SQ_INSTS_SALU")] / ws))
lines.append("| VALU busy = 4*SQ_ACTIVE_INST_VALU / (1024 SIMD x GRBM_GUI_ACTIVE/8) | %.3f |" % (4 * sq[(kern, "SQ_ACTIVE_INST_VALU")] / (1024 * gui)))
lines.append("| shader cycles per wave-step per SIMD | %.0f |" % (gui * 1024 / ws))
lines.append("| algorithmic FP64 rate (35 flop/step) | %.2f TFLOP/s = %.3f of 78.6 |" % (bench["roofline"]["achieved"], bench["roofline"]["frac"]))
lines.append("| HBM traffic, integration kernel | read %.2f MB + write %.2f MB |" % (fetch_i / 1e6, write_i / 1e6))
lines.append("| HBM traffic, shade kernel | read %.2f MB + write %.2f MB |" % (fetch_s / 1e6, write_s / 1e6))
lines.append("| algorithmic HBM bytes (7 B/ray) | %.2f MB |" % (7 * rays / 1e6))
lines.append("")
if inter is not None:
    t = traffic["interstellar_1920x1080_cap4096_%s" % kern]
    lines.append("## Interstellar metric, same frame (`bench.py --metric interstellar`)\n")
    lines.append("| quantity | value |\n|---|---|")
    lines.append("| throughput | %.1f %s, %.3f ms per frame |" % (inter["value"], inter["unit"], inter["ms_per_step"]))
    lines.append("| VALU / SALU instructions per wave-step | %.1f / %.1f |" % (t["valu_instr_per_wave_step"], t["salu_instr_per_wave_step"]))
    lines.append("| VALU busy | %.3f |" % t["valu_busy"])
    lines.append("| shader cycles per wave-step per SIMD | %.0f |" % (t["shader_cycles_per_frame"] * 1024 / iws))
    lines.append("| algorithmic FP64 rate (46 flop/step) | %.2f TFLOP/s = %.3f of 78.6 |" % (inter["roofline"]["achieved"], inter["roofline"]["frac"]))
    lines.append("| HBM traffic | read %.2f MB + write %.2f MB |" % (t["integrate_fetch_bytes"] / 1e6, t["integrate_write_bytes"] / 1e6))
    lines.append("")
lines.append("## All BASELINE configurations on one GPU\n")
lines.append(open(os.path.join(G, "configs.md")).read())
open(os.path.join(P, "%s_summary.md" % rnd), "w").write("\n".join(lines))
print("\n".join(lines[-30:]))
