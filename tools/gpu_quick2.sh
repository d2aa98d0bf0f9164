#!/bin/bash
# quick loop: gpu tests + bench (Ellis, Interstellar) + SQ counters of both default kernels
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT; cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -15 > $OUT/pytest_gpu.log
python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/q_bench_v1.json
python bench.py --steps 5 --warmup 2 --metric interstellar --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/q_bench_interstellar.json
cd /tmp
C="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE"
rocprofv3 --pmc $C --output-format csv -d $OUT/q_pmc -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/q_pmc.log 2>&1
rocprofv3 --pmc $C --output-format csv -d $OUT/q_pmc_inter -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --metric interstellar --no-cpu-baseline > $OUT/q_pmc_inter.log 2>&1
