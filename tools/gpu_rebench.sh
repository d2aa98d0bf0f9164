# Bench lines + kernel-trace stats of the default bench in ONE call (same box), after profiles/traffic.json was refreshed
set -x
OUT=gpurun_out/final
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
python bench.py --variant 0 --no-cpu-baseline > $OUT/bench_persistent.json 2>/dev/null
python bench.py --fast-math 0 --no-cpu-baseline > $OUT/bench_strict.json 2>/dev/null
python bench.py --download --no-cpu-baseline > $OUT/bench_download.json 2>/dev/null
python bench.py --metric interstellar --width 3840 --height 2160 --max-iter 8192 --steps 5 --warmup 1 --no-cpu-baseline > $OUT/bench_config3.json 2>/dev/null
python bench.py --metric interstellar --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_interstellar_1080p.json 2>/dev/null
tail -c 300 $OUT/bench_default.json
export TMPDIR=/tmp
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/stats -o bench -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $GRAFT_REPO_ROOT/$OUT/stats.log 2>&1 )
cat $OUT/stats/bench_kernel_stats.csv | head -3
