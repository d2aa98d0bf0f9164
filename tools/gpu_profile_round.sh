#!/bin/bash
# Profile collection of a round (run ON the GPU box):  bash tools/gpu_profile_round.sh round2
# For every workload: the un-profiled bench line, rocprofv3 --kernel-trace --stats, and THREE separate PMC passes
# (SQ set, FETCH_SIZE, WRITE_SIZE -- never combined with a trace domain) of the same bench command.
# Output: gpurun_out/<round>/<workload>/...; tools/make_profiles.py <round> turns it into profiles/<round>_*.
RND=${1:-round2}
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$RND
rm -rf "$OUT"; mkdir -p "$OUT"
SQ="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE"
run_workload() { # name, bench arguments...
  local name=$1; shift
  local d=$OUT/$name; mkdir -p "$d"
  cd "$ROOT"
  python bench.py "$@" --no-traffic --no-live-traffic > "$d/bench.json" 2> "$d/bench.err"
  cd /tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d "$d/stats" -o bench -- python "$ROOT/bench.py" "$@" --no-traffic --no-cpu-baseline --multi-frame 0 --sustained-seconds 0 > "$d/stats.log" 2>&1
  rocprofv3 --pmc $SQ --output-format csv -d "$d/pmc_sq" -o pmc -- python "$ROOT/bench.py" "$@" --steps 3 --warmup 1 --no-traffic --no-cpu-baseline --multi-frame 0 --sustained-seconds 0 > "$d/pmc_sq.log" 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$d/pmc_fetch" -o pmc -- python "$ROOT/bench.py" "$@" --steps 3 --warmup 1 --no-traffic --no-cpu-baseline --multi-frame 0 --sustained-seconds 0 > "$d/pmc_fetch.log" 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$d/pmc_write" -o pmc -- python "$ROOT/bench.py" "$@" --steps 3 --warmup 1 --no-traffic --no-cpu-baseline --multi-frame 0 --sustained-seconds 0 > "$d/pmc_write.log" 2>&1
  # instruction mix of the FP64 pipe and LDS behaviour (two more passes; what the 83 / 116 VALU instructions are)
  rocprofv3 --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 --output-format csv -d "$d/pmc_mix1" -o pmc -- python "$ROOT/bench.py" "$@" --steps 3 --warmup 1 --no-traffic --no-cpu-baseline --multi-frame 0 --sustained-seconds 0 > "$d/pmc_mix1.log" 2>&1
  rocprofv3 --pmc SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS --output-format csv -d "$d/pmc_mix2" -o pmc -- python "$ROOT/bench.py" "$@" --steps 3 --warmup 1 --no-traffic --no-cpu-baseline --multi-frame 0 --sustained-seconds 0 > "$d/pmc_mix2.log" 2>&1
  cd "$ROOT"
}
# configs[1] (the headline): the library's automatic kernel (relay) and the static kernel
run_workload ellis_1080p
run_workload ellis_1080p_static --variant 1 --no-cpu-baseline
# Interstellar metric at the same frame, and BASELINE configs[2] at full size (3840x2160, cap 8192): automatic and static
run_workload interstellar_1080p --metric interstellar --steps 8 --warmup 2 --no-cpu-baseline
run_workload interstellar_4k --metric interstellar --width 3840 --height 2160 --max-iter 8192 --steps 4 --warmup 1 --no-cpu-baseline
run_workload interstellar_4k_static --metric interstellar --width 3840 --height 2160 --max-iter 8192 --steps 4 --warmup 1 --variant 1 --no-cpu-baseline
# device PNG front end (kernels_png.h): 8 efficient-mode 1080p frames per call; kernel trace and the two HBM passes
PD=$OUT/png_front_end; mkdir -p "$PD"; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$PD/stats" -o png -- python "$ROOT/tools/gpu_png_front_end.py" profile > "$PD/run.txt" 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$PD/pmc_fetch" -o pmc -- python "$ROOT/tools/gpu_png_front_end.py" profile > "$PD/pmc_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$PD/pmc_write" -o pmc -- python "$ROOT/tools/gpu_png_front_end.py" profile > "$PD/pmc_write.log" 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d "$PD/pmc_sq" -o pmc -- python "$ROOT/tools/gpu_png_front_end.py" profile > "$PD/pmc_sq.log" 2>&1
cd "$ROOT"
python tools/bench_configs.py > "$OUT/configs.md" 2> "$OUT/configs.err"
ls -R "$OUT" | head -60
