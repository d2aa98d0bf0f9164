mkdir -p gpurun_out
python tools/gpu_cli_startup.py 3 2>&1 | grep -E '^##|^wall'
echo SLOW; CURVIS_SLOW_EXIT=1 python tools/gpu_cli_startup.py 3 2>&1 | grep -E '^##|^wall'
echo FAST; python tools/gpu_cli_startup.py 3 2>&1 | grep -E '^##|^wall'
