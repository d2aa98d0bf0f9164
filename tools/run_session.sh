mkdir -p gpurun_out
for rep in 1 2; do
echo "== sampling_priority=0"; CURVIS_CTX_OPTIONS="sampling_priority=0" SWEEP_C=1,4 SWEEP_B=32 timeout 600 python tools/gpu_eff_contexts_sweep.py 1 2>&1 | grep -E '^round'
echo "== sampling_priority=1"; SWEEP_C=1,4 SWEEP_B=32 timeout 600 python tools/gpu_eff_contexts_sweep.py 1 2>&1 | grep -E '^round'
done > gpurun_out/eff_priority_ab.txt 2>&1; cat gpurun_out/eff_priority_ab.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "efficient" 2>&1 | tail -2
