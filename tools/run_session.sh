mkdir -p gpurun_out
(time timeout 900 python -m pytest tests/test_gpu_cli.py tests/test_gpu_png.py tests/test_gpu_configs.py -m gpu -x -q) > gpurun_out/s4_tests.txt 2>&1; tail -15 gpurun_out/s4_tests.txt
python tools/gpu_cli_startup.py 3 > gpurun_out/cli_startup.txt 2>&1; cat gpurun_out/cli_startup.txt
