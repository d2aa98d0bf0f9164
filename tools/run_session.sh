mkdir -p gpurun_out
(time timeout 900 python -m pytest tests/test_gpu_cli.py -m gpu -x -q) > gpurun_out/s7_tests.txt 2>&1; tail -6 gpurun_out/s7_tests.txt
