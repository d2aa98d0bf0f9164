#!/bin/bash
# PMC passes over efficient_pixel_kernel (the per-pixel kernel of the default mode): instructions per wave, VALU busy, FP64 mix, LDS.
# bash tools/gpu_eff_pixel_profile.sh -> gpurun_out/effpix/*  (summarised by hand into profiles/<round>_eff_pixel_pmc.txt)
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
D=$ROOT/gpurun_out/effpix; mkdir -p $D
python $ROOT/tools/gpu_eff_pixel_profile.py 6 > $D/plain.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $D/stats -o t -- python $ROOT/tools/gpu_eff_pixel_profile.py 4 > $D/stats.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE --output-format csv -d $D/pmc_sq -o pmc -- python $ROOT/tools/gpu_eff_pixel_profile.py 3 > $D/pmc_sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 --output-format csv -d $D/pmc_mix1 -o pmc -- python $ROOT/tools/gpu_eff_pixel_profile.py 3 > $D/pmc_mix1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $D/pmc_mix2 -o pmc -- python $ROOT/tools/gpu_eff_pixel_profile.py 3 > $D/pmc_mix2.log 2>&1
cat $D/plain.txt
python - <<PY
import csv, glob, collections
for sub in ("pmc_sq", "pmc_mix1", "pmc_mix2"):
    for f in glob.glob("$D/%s/**/*counter_collection.csv" % sub, recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "efficient_pixel" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in sorted(acc.items()):
            print("%-28s per launch (32 frames): %.4g  (n=%d)" % (k, sum(v) / len(v), len(v)))
PY
grep -h 'efficient_pixel\|escape_angle' $D/stats/*/*kernel_stats.csv 2>/dev/null | head -4
