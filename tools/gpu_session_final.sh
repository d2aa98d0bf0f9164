#!/bin/bash
# Round profile collection (run on the GPU box): default bench line, kernel-trace stats, PMC passes
# (SQ set, FETCH_SIZE, WRITE_SIZE in SEPARATE runs), all-configs table, FP64 micro-benchmarks.
set -x
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/final
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|error" | tail -5 > $OUT/pytest_gpu.log
python __graft_entry__.py smoke > $OUT/smoke.log 2>&1
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
python bench.py --variant 0 --no-cpu-baseline > $OUT/bench_persistent.json 2>/dev/null
python bench.py --fast-math 0 --no-cpu-baseline > $OUT/bench_strict.json 2>/dev/null
python bench.py --download --no-cpu-baseline > $OUT/bench_download.json 2>/dev/null
python bench.py --metric interstellar --width 3840 --height 2160 --max-iter 8192 --steps 5 --warmup 1 --no-cpu-baseline > $OUT/bench_config3.json 2>/dev/null
python tools/bench_configs.py > $OUT/configs.md 2> $OUT/configs.err
make -s -C curvis_amd/csrc ubench > /dev/null 2>&1; ./build/ubench_fp64 > $OUT/ubench.log 2>&1
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $OUT/stats.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/pmc_sq.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/pmc_write.log 2>&1
INTER="--metric interstellar --steps 2 --warmup 1 --no-cpu-baseline"
python $GRAFT_REPO_ROOT/bench.py --metric interstellar --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_interstellar_1080p.json 2>/dev/null
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq_inter -o pmc -- python $GRAFT_REPO_ROOT/bench.py $INTER > $OUT/pmc_sq_inter.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_inter -o pmc -- python $GRAFT_REPO_ROOT/bench.py $INTER > $OUT/pmc_fetch_inter.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_inter -o pmc -- python $GRAFT_REPO_ROOT/bench.py $INTER > $OUT/pmc_write_inter.log 2>&1
cd $GRAFT_REPO_ROOT
(python tools/gpu_cli_video.py; python tools/gpu_cli_image.py) > $OUT/cli_video.txt 2>&1
SPECS=0,4,6,8 BATCHES=8,30 python tools/gpu_efficient_sweep.py 2>&1 | grep spec > $OUT/efficient_sweep.txt
python tools/gpu_tail.py 2>&1 | grep frames > $OUT/tail.txt
python tools/gpu_trace.py > /dev/null 2>&1 && python tools/analyze_trace.py gpurun_out/trace_config2.bin > $OUT/wave_trace_config2.txt && python tools/analyze_trace.py gpurun_out/trace_config2_relay.bin > $OUT/wave_trace_config2_relay.txt
(timeout 900 python tools/gpu_deep_fuzz.py 2000 7; timeout 900 python tools/gpu_deep_fuzz.py 2000 11; timeout 900 python tools/gpu_deep_fuzz.py 2000 23) 2>&1 | grep -E "MISMATCH|scenes" | tail -9 > $OUT/deep_fuzz.txt
python tools/gpu_eff_phases.py 2>&1 | grep -E "frames|curvis\]" > $OUT/efficient_phases.txt
ls -R $OUT | head -80
