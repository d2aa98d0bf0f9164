#!/bin/bash
# round-1 profile collection: kernel-trace stats + PMC passes (separate runs), default bench line
set -x
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > $OUT/pytest_gpu.log
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
for v in 0 1; do python bench.py --steps 10 --warmup 2 --variant $v --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_v${v}.json; done
cd /tmp
for v in 0 1; do
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats_v$v -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --variant $v --no-cpu-baseline > $OUT/prof_stats_v$v.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/prof_pmc_sq_v$v -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --variant $v --no-cpu-baseline > $OUT/prof_pmc_sq_v$v.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/prof_pmc_fetch_v$v -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --variant $v --no-cpu-baseline > $OUT/prof_pmc_fetch_v$v.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/prof_pmc_write_v$v -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --variant $v --no-cpu-baseline > $OUT/prof_pmc_write_v$v.log 2>&1
done
find $OUT -name '*.csv' | head -40
