#!/bin/bash
set -x
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $OUT/pytest_gpu.log
for v in 0 1; do for f in 0 1; do python bench.py --steps 5 --warmup 2 --variant $v --fast-math $f --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_v${v}_f${f}.json; done; done
for t in 4 8 16 32; do python bench.py --steps 5 --warmup 2 --refill-threshold $t --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_thr$t.json; done
python bench.py --steps 5 --warmup 2 --metric interstellar --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_interstellar_v0.json
python bench.py --steps 5 --warmup 2 --metric interstellar --variant 1 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_interstellar_v1.json
cd /tmp
for v in 0 1; do
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE -d $OUT/prof_pmc_v$v -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --variant $v --no-cpu-baseline > $OUT/prof_pmc_v$v.log 2>&1
done
