#!/bin/bash
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out; cd /tmp
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_inter -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --metric interstellar --no-cpu-baseline > $OUT/pmc_inter.log 2>&1
