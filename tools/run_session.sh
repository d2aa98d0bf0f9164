mkdir -p gpurun_out
python tools/gpu_cli_startup.py 3 > gpurun_out/cli_startup.txt 2>&1; cat gpurun_out/cli_startup.txt
echo; echo "# JPEG reconstruction on one thread (CURVIS_DECODE_THREADS=1):"; CURVIS_DECODE_THREADS=1 python tools/gpu_cli_startup.py 2 2>&1 | grep -A12 'JPEG backgrounds$' | grep -E 'wall|jpeg|settings'
