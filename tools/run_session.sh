mkdir -p gpurun_out
SWEEP_C=0 SWEEP_B=0 timeout 600 python tools/gpu_eff_contexts_sweep.py 3 > gpurun_out/eff_auto.txt 2>&1; grep -E '^round' gpurun_out/eff_auto.txt
(time timeout 900 python -m pytest tests/test_gpu_cli.py tests/test_gpu_configs.py -m gpu -x -q) 2>&1 | tail -5
