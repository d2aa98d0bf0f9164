#!/usr/bin/env python3
"""Assemble profiles/<round>_* and profiles/traffic.json from gpurun_out/<round>/ (output of
tools/gpu_profile_round.sh): per workload the bench line, the rocprofv3 kernel statistics and the three PMC
passes.  HBM bytes follow MI355X_MICROARCH.md: FETCH_SIZE / WRITE_SIZE are in KiB, separate passes, and on gfx950
FETCH_SIZE counts 128-byte requests as 64 bytes (doubled here)."""
import collections
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1] if len(sys.argv) > 1 else "round2"
G = os.path.join(ROOT, "gpurun_out", rnd)
P = os.path.join(ROOT, "profiles")
KIB = 1024.0
KERNELS = ("geodesic_static", "geodesic_persistent", "geodesic_relay", "shade_kernel")


def pmc(path):
    rows = list(csv.DictReader(open(path)))
    agg = collections.defaultdict(list)
    for r in rows:
        short = next((k for k in KERNELS if k in r["Kernel_Name"]), None)
        if short:
            agg[(short, r["Counter_Name"])].append(float(r["Counter_Value"]))
    # median over the dispatches of the run: one launch in a dozen shows a several-fold FETCH_SIZE (first touch of
    # a buffer by that process); the mean of three or four launches would report that, not the kernel
    med = lambda v: sorted(v)[len(v) // 2] if len(v) % 2 else 0.5 * (sorted(v)[len(v) // 2 - 1] + sorted(v)[len(v) // 2])
    return {k: med(v) for k, v in agg.items()}


traffic_path = os.path.join(P, "traffic.json")
traffic = json.load(open(traffic_path)) if os.path.exists(traffic_path) else {}
lines = ["# %s profile summary (MI355X gfx950, ROCm 7.2)\n" % rnd,
         "Collected on the GPU box by tools/gpu_profile_round.sh, assembled by tools/make_profiles.py.  Per workload: the "
         "un-profiled `bench.py` line, `rocprofv3 --kernel-trace --stats`, and three separate `--pmc` passes (SQ set, "
         "FETCH_SIZE, WRITE_SIZE) of the same command.\n"]
for name in sorted(os.listdir(G)):
    d = os.path.join(G, name)
    if not os.path.isdir(d) or not os.path.exists(os.path.join(d, "bench.json")):
        continue
    bench = json.loads(open(os.path.join(d, "bench.json")).read().strip().splitlines()[-1])
    kern = bench["roofline"]["kernel"]
    steps, rays = bench["config"]["executed_steps_per_frame"], bench["config"]["rays_per_frame"]
    ws = steps / 64.0
    sq = pmc(os.path.join(d, "pmc_sq", "pmc_counter_collection.csv"))
    fe = pmc(os.path.join(d, "pmc_fetch", "pmc_counter_collection.csv"))
    wr = pmc(os.path.join(d, "pmc_write", "pmc_counter_collection.csv"))
    fetch_b, write_b = fe[(kern, "FETCH_SIZE")] * KIB * 2, wr[(kern, "WRITE_SIZE")] * KIB
    gui = sq[(kern, "GRBM_GUI_ACTIVE")] / 8.0   # summed over the 8 XCDs
    wl = bench["config"]["workload"]
    metric = "interstellar" if "interstellar" in wl else "ellis"
    res = wl.split(",")[1].strip()
    cap = int(wl.split("cap")[1].split()[0])
    key = "%s_%s_cap%d_%s" % (metric, res, cap, kern)
    flop = bench["roofline"]["flop_per_step"]
    traffic[key] = {
        "integrate_kernel_bytes": int(fetch_b + write_b), "integrate_fetch_bytes": int(fetch_b),
        "integrate_write_bytes": int(write_b),
        "shade_kernel_bytes": int(fe.get(("shade_kernel", "FETCH_SIZE"), 0.0) * KIB * 2 + wr.get(("shade_kernel", "WRITE_SIZE"), 0.0) * KIB),
        "algorithmic_bytes": 7 * rays,
        "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, KiB -> bytes, FETCH doubled per the gfx950 "
                  "note in MI355X_MICROARCH.md), profiles/%s_%s_pmc_*.csv" % (rnd, name),
        "valu_busy": round(4 * sq[(kern, "SQ_ACTIVE_INST_VALU")] / (1024 * gui), 4),
        "valu_instr_per_wave_step": round(sq[(kern, "SQ_INSTS_VALU")] / ws, 1),
        "salu_instr_per_wave_step": round(sq[(kern, "SQ_INSTS_SALU")] / ws, 1),
        "shader_cycles_per_frame": int(gui),
    }
    for kind in ("sq", "fetch", "write"):
        shutil.copy(os.path.join(d, "pmc_%s" % kind, "pmc_counter_collection.csv"), os.path.join(P, "%s_%s_pmc_%s.csv" % (rnd, name, kind)))
    shutil.copy(os.path.join(d, "stats", "bench_kernel_stats.csv"), os.path.join(P, "%s_%s_kernel_stats.csv" % (rnd, name)))
    shutil.copy(os.path.join(d, "bench.json"), os.path.join(P, "%s_%s_bench.json" % (rnd, name)))
    kstats = {r["Name"]: r for r in csv.DictReader(open(os.path.join(d, "stats", "bench_kernel_stats.csv")))}
    kname = [k for k in kstats if kern in k][0]
    avg_ms, calls = float(kstats[kname]["AverageNs"]) / 1e6, int(kstats[kname]["Calls"])
    t = traffic[key]
    lines.append("## %s: `%s` -- %s\n" % (name, kern, wl))
    lines.append("| quantity | value |\n|---|---|")
    lines.append("| bench.py value (un-profiled) | %.1f %s, %.3f ms per step%s |" % (
        bench["value"], bench["unit"], bench["ms_per_step"],
        (", multi-frame launches %.1f (%.3f ms per frame)" % (bench["value_multi_frame"]["value"], bench["value_multi_frame"]["ms_per_frame"]))
        if bench.get("value_multi_frame") else ""))
    lines.append("| executed Euler steps / frame | %d (= %.4g wave-steps of 64 lanes), %d rays |" % (steps, ws, rays))
    lines.append("| average kernel duration: rocprofv3 kernel trace (%d calls, warm-up included) / bench.py HIP events | %.3f ms / %.3f ms |" % (
        calls, avg_ms, bench["roofline"]["kernel_ms_avg"]))
    lines.append("| VALU / SALU instructions per wave-step | %.1f / %.1f |" % (t["valu_instr_per_wave_step"], t["salu_instr_per_wave_step"]))
    lines.append("| VALU busy = 4 SQ_ACTIVE_INST_VALU / (1024 SIMD x GRBM_GUI_ACTIVE/8) | %.3f |" % t["valu_busy"])
    lines.append("| shader cycles per wave-step per SIMD | %.0f |" % (gui * 1024 / ws))
    lines.append("| algorithmic FP64 rate (%d flop/step) | %.2f TFLOP/s = %.4f of 78.6 |" % (flop, bench["roofline"]["achieved"], bench["roofline"]["frac"]))
    lines.append("| HBM traffic of the kernel (PMC) vs algorithmic (7 B/ray) | read %.2f MB + write %.2f MB vs %.2f MB |" % (
        fetch_b / 1e6, write_b / 1e6, 7 * rays / 1e6))
    mix = {}
    for part in ("pmc_mix1", "pmc_mix2"):
        f = os.path.join(d, part, "pmc_counter_collection.csv")
        if os.path.exists(f):
            mix.update({c: v for (k, c), v in pmc(f).items() if k == kern})
            shutil.copy(f, os.path.join(P, "%s_%s_%s.csv" % (rnd, name, part)))
    if "SQ_INSTS_VALU_FMA_F64" in mix:
        fma, mul, add, trans = (mix.get("SQ_INSTS_VALU_" + n, 0.0) / ws for n in ("FMA_F64", "MUL_F64", "ADD_F64", "TRANS_F64"))
        lines.append("| FP64 instructions per wave-step: fma / mul / add / transcendental (rcp, rsq) | %.1f / %.1f / %.1f / %.1f |" % (fma, mul, add, trans))
        hw = (2 * fma + mul + add + trans) * 64 * ws / (bench["roofline"]["kernel_ms_avg"] * 1e-3) / 1e12
        lines.append("| FP64 operations the hardware executed (fma = 2) | %.1f per lane-step = %.2f TFLOP/s = %.3f of 78.6 |" % (
            2 * fma + mul + add + trans, hw, hw / 78.6))
        traffic[key].update({"fp64_fma_per_wave_step": round(fma, 2), "fp64_mul_per_wave_step": round(mul, 2),
                             "fp64_add_per_wave_step": round(add, 2), "fp64_trans_per_wave_step": round(trans, 2),
                             "fp64_ops_executed_per_lane_step": round(2 * fma + mul + add + trans, 2)})
    if "SQ_INSTS_VALU_INT32" in mix:
        lines.append("| other VALU per wave-step: int32 / conversions; LDS instructions per wave-step | %.1f / %.1f; %.1f |" % (
            mix["SQ_INSTS_VALU_INT32"] / ws, mix.get("SQ_INSTS_VALU_CVT", 0.0) / ws, mix.get("SQ_INSTS_LDS", 0.0) / ws))
        if mix.get("SQ_ACTIVE_INST_LDS"):
            lines.append("| LDS bank-conflict cycles / LDS active cycles | %.3f |" % (mix.get("SQ_LDS_BANK_CONFLICT", 0.0) / mix["SQ_ACTIVE_INST_LDS"]))
    if "cpu_baseline" in bench:
        lines.append("| CPU baseline (oracle, 1 thread) | %.2f %s; %s |" % (bench["cpu_baseline"]["value"], bench["cpu_baseline"]["unit"], bench["cpu_baseline"]["sample"]))
    lines.append("")
    lines.append("```\n%s```\n" % open(os.path.join(d, "stats", "bench_kernel_stats.csv")).read())
# ---- device PNG front end: per-kernel duration (trace) and HBM bytes (PMC), per call of 8 efficient-mode 1080p frames
pd = os.path.join(G, "png_front_end")
if os.path.exists(os.path.join(pd, "stats", "png_kernel_stats.csv")):
    # two-pass path (round 5) first: "png_hist_kernel" is no prefix of "png_hist2_kernel", but list the longer names first anyway
    PNGK = ("png_hist2_kernel", "png_blockbits_kernel", "png_offsets_kernel", "png_emit2_kernel", "png_crc_kernel",
            "png_hist_kernel", "png_count_kernel", "png_scan_kernel", "png_zero_kernel", "png_emit_kernel")

    def pmc_png(path):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(path)):
            short = next((k for k in PNGK if k in r["Kernel_Name"]), None)
            if short:
                agg[(short, r["Counter_Name"])].append(float(r["Counter_Value"]))
        return {k: sorted(v)[len(v) // 2] for k, v in agg.items()}
    ks = {next(k for k in PNGK if k in r["Name"]): r for r in csv.DictReader(open(os.path.join(pd, "stats", "png_kernel_stats.csv"))) if any(k in r["Name"] for k in PNGK)}
    fe, wr = pmc_png(os.path.join(pd, "pmc_fetch", "pmc_counter_collection.csv")), pmc_png(os.path.join(pd, "pmc_write", "pmc_counter_collection.csv"))
    sq = pmc_png(os.path.join(pd, "pmc_sq", "pmc_counter_collection.csv")) if os.path.exists(os.path.join(pd, "pmc_sq", "pmc_counter_collection.csv")) else {}
    shutil.copy(os.path.join(pd, "stats", "png_kernel_stats.csv"), os.path.join(P, "%s_png_kernel_stats.csv" % rnd))
    for kind in ("fetch", "write", "sq"):
        f = os.path.join(pd, "pmc_%s" % kind, "pmc_counter_collection.csv")
        if os.path.exists(f):
            shutil.copy(f, os.path.join(P, "%s_png_pmc_%s.csv" % (rnd, kind)))
    run = open(os.path.join(pd, "run.txt")).read().strip().splitlines()
    nf, frame_b = 8, 1920 * 1080 * 3
    lines.append("## PNG front end on the device (kernels_png.h): 8 efficient-mode 1920x1080 frames per call (`tools/gpu_png_front_end.py profile`)\n")
    lines.append("```\n%s\n```\n" % "\n".join(ln for ln in run if "frames/call" in ln))
    lines.append("| kernel | avg duration (trace) | HBM read + write per call (PMC, FETCH doubled) | bytes / duration | VALU busy | LDS bank-conflict share |\n|---|---|---|---|---|---|")
    tot_ms, tot_b = 0.0, 0.0
    for k in PNGK:
        if k not in ks:
            continue
        ms = float(ks[k]["AverageNs"]) / 1e6
        fb, wb = fe.get((k, "FETCH_SIZE"), 0.0) * KIB * 2, wr.get((k, "WRITE_SIZE"), 0.0) * KIB
        gui = sq.get((k, "GRBM_GUI_ACTIVE"), 0.0) / 8.0
        busy = "%.2f" % (4 * sq[(k, "SQ_ACTIVE_INST_VALU")] / (1024 * gui)) if gui else "-"
        conf = "%.2f" % (sq.get((k, "SQ_LDS_BANK_CONFLICT"), 0.0) / sq[(k, "SQ_ACTIVE_INST_LDS")]) if sq.get((k, "SQ_ACTIVE_INST_LDS")) else "-"
        lines.append("| `%s` | %.4f ms | %.2f + %.2f MB | %.0f GB/s | %s | %s |" % (k, ms, fb / 1e6, wb / 1e6, (fb + wb) / (ms * 1e-3) / 1e9 if ms else 0.0, busy, conf))
        tot_ms += ms
        tot_b += fb + wb
    lines.append("| all kernels | %.4f ms per call = %.4f ms per frame | %.2f MB moved vs %.2f MB of pixels (algorithmic: + the streams, ~5 MB here) | %.0f GB/s moved, %.0f GB/s algorithmic of 8000 | | |\n" % (
        tot_ms, tot_ms / nf, tot_b / 1e6, nf * frame_b / 1e6, tot_b / (tot_ms * 1e-3) / 1e9, nf * frame_b / (tot_ms * 1e-3) / 1e9))
json.dump(traffic, open(traffic_path, "w"), indent=1)
cfg = os.path.join(G, "configs.md")
if os.path.exists(cfg):
    shutil.copy(cfg, os.path.join(P, "%s_configs.md" % rnd))
    lines.append("## All BASELINE configurations on one GPU (tools/bench_configs.py)\n")
    lines.append(open(cfg).read())
open(os.path.join(P, "%s_summary.md" % rnd), "w").write("\n".join(lines))
print("\n".join(lines))
