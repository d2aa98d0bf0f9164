#!/bin/bash
# first measurement session on the GPU box
set -x
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $OUT/pytest_gpu.log
./build/ubench_fp64 > $OUT/ubench.log 2>&1
for v in 0 1; do python bench.py --steps 5 --warmup 2 --variant $v --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_variant$v.json; done
for t in 1 2 4 8 16 32; do python bench.py --steps 5 --warmup 2 --refill-threshold $t --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_thr$t.json; done
for b in 1 2 3 4; do python bench.py --steps 5 --warmup 2 --blocks-per-cu $b --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_bpc$b.json; done
python bench.py --steps 5 --warmup 2 --metric interstellar --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_interstellar.json
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/prof_stats -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/prof_stats.log 2>&1
rocprofv3 -L > $OUT/counters_list.txt 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE -d $OUT/prof_pmc1 -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/prof_pmc1.log 2>&1
ls -R $OUT | head -50
