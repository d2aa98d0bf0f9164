mkdir -p gpurun_out
for s in 1 2 3 4; do timeout 300 python tools/gpu_async_fuzz.py 500 $s; done > gpurun_out/async_fuzz.txt 2>&1; cat gpurun_out/async_fuzz.txt | tail -20
