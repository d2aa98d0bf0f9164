#!/bin/bash
# Re-validation of the round's final kernels beyond the test-suite (run on the GPU box):
# deep parity fuzz against the oracle, relay-kernel soak (incl. per-frame counters), fast step vs IEEE step.
# usage: bash tools/gpu_soak_round.sh <round>   (output: gpurun_out/soak_<round>/)
# SEED_BASE=<n> shifts every seed (default 0 = the seeds of the committed profiles/round2_* runs).
cd ${GRAFT_REPO_ROOT:-.}
B=${SEED_BASE:-0}
RND=${1:-round3}
OUT=gpurun_out/soak_$RND; mkdir -p $OUT
(timeout 900 python tools/gpu_deep_fuzz.py 2000 $((B+101)); timeout 900 python tools/gpu_deep_fuzz.py 2000 $((B+202)); timeout 900 python tools/gpu_deep_fuzz.py 2000 $((B+303))) 2>&1 | grep -E "MISMATCH|scenes" | tail -9 > $OUT/deep_fuzz.txt
(timeout 900 python tools/gpu_relay_soak.py 4000 $((B+31)); timeout 900 python tools/gpu_relay_soak.py 4000 $((B+32))) 2>&1 | grep -E "MISMATCH|launches" | tail -12 > $OUT/relay_soak.txt
timeout 1200 python tools/gpu_fast_vs_strict.py 600 $((B+77)) 2>&1 | tail -6 > $OUT/fast_vs_strict.txt
(timeout 900 python tools/gpu_eff_fuzz.py 5000 $((B+5)); timeout 900 python tools/gpu_eff_fuzz.py 5000 $((B+6))) 2>&1 | grep -E "MISMATCH|MISSING|scenes" | tail -8 > $OUT/eff_fuzz.txt
cat $OUT/*.txt
