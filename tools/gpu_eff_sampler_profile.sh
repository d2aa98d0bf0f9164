#!/bin/bash
# PMC passes over sampler_kernel (the device-resident adaptive sampler of the default mode): waves, instructions, VALU busy, LDS, on a launch
# of 128 jobs (the first 128 fly-through poses: every frame its own camera radius, Interstellar metric, cap 8192) and one of a single job
# (128 orbit poses: one shared radius).  bash tools/gpu_eff_sampler_profile.sh -> gpurun_out/effsamp/* -> profiles/<round>_eff_sampler_pmc.txt
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
D=$ROOT/gpurun_out/effsamp; rm -rf $D; mkdir -p $D
cat > /tmp/samp_workload.py <<'PY'
import os, sys
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import refpaths, curvis_amd
from curvis_amd import rendering, skies
ctx = curvis_amd.Context(0)
ctx.set_sky(0, curvis_amd.SphericalImage(skies.smooth(2048, 1024, 128))); ctx.set_sky(1, curvis_amd.SphericalImage(skies.smooth(2048, 1024, 32)))
ctx.set_option("device_sampler", 1)
for csv, fps, metric, cap in (("path_through.csv", 24.0, curvis_amd.InterstellarMetric(0.1, 1e-4, 1.0), 8192), ("path_orbit.csv", 4.0, curvis_amd.EllisMetric(1.0), 4096)):
    it = rendering.Interpolator.from_file(refpaths.reference_path_file(csv))
    times = rendering.times_of_frames(it.min_time(), it.max_time(), fps)[:128]
    cams = [curvis_amd.Camera(tuple(it.camera_position(t)), tuple(it.camera_forward(t)), tuple(it.camera_up(t)), 15.0, 43.0, 480, 270) for t in times]
    for rep in range(int(sys.argv[1])):
        _, st = ctx.render_efficient(metric, cams, cap, 100.0, 0.05, 100, 100, 1e-5, 1e-5, download=False)
    print("%s: sampler kernel %.3f ms per launch of 128 frames, %d Euler chains, %d points integrated, %d steps consumed" % (
        csv, st.integrate_ms, ctx.get_option("last_sampling_chains"), ctx.get_option("last_sampling_evaluated"), st.steps))
PY
python /tmp/samp_workload.py 3 > $D/plain.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $D/stats -o t -- python /tmp/samp_workload.py 3 > $D/stats.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE --output-format csv -d $D/pmc_sq -o pmc -- python /tmp/samp_workload.py 2 > $D/pmc_sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 --output-format csv -d $D/pmc_mix1 -o pmc -- python /tmp/samp_workload.py 2 > $D/pmc_mix1.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS --output-format csv -d $D/pmc_mix2 -o pmc -- python /tmp/samp_workload.py 2 > $D/pmc_mix2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $D/pmc_fetch -o pmc -- python /tmp/samp_workload.py 2 > $D/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $D/pmc_write -o pmc -- python /tmp/samp_workload.py 2 > $D/pmc_write.log 2>&1
cat $D/plain.txt
python - <<PY
import csv, glob, collections
for sub in ("pmc_sq", "pmc_mix1", "pmc_mix2", "pmc_fetch", "pmc_write"):
    for f in glob.glob("$D/%s/**/*counter_collection.csv" % sub, recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "sampler_kernel" in r["Kernel_Name"]:
                acc[(r["Kernel_Name"].split("sampler_kernel")[1][:8], r["Counter_Name"])].append(float(r["Counter_Value"]))
        for k, v in sorted(acc.items()):
            print("sampler_kernel%-8s %-26s per launch: %.5g  (n=%d)" % (k[0], k[1], sum(v) / len(v), len(v)))
PY
grep -h 'sampler_kernel\|efficient_pixel' $D/stats/*/*kernel_stats.csv 2>/dev/null | head -4
