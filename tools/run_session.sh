mkdir -p gpurun_out
build/ubench_issue > gpurun_out/ubench_issue.txt 2>&1; tail -40 gpurun_out/ubench_issue.txt
(time timeout 1100 python -m pytest tests -m gpu -x -q) > gpurun_out/s2_gpu_tests.txt 2>&1; tail -3 gpurun_out/s2_gpu_tests.txt
python bench.py > gpurun_out/s2_bench.json 2> gpurun_out/s2_bench.err; tail -c 300 gpurun_out/s2_bench.json
SWEEP_MODE=brute SWEEP_FPS=8 SWEEP_C=1,2 SWEEP_B=4,8,16 timeout 300 python tools/gpu_eff_contexts_sweep.py 2 > gpurun_out/s2_brute_sweep.txt 2>&1; tail -12 gpurun_out/s2_brute_sweep.txt
