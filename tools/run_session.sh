mkdir -p gpurun_out
SWEEP_C=1,2,4 SWEEP_B=32 timeout 600 python tools/gpu_eff_contexts_sweep.py 2 > gpurun_out/eff_sweep_final.txt 2>&1; tail -14 gpurun_out/eff_sweep_final.txt
