mkdir -p gpurun_out
python tools/gpu_cli_startup.py 4 > gpurun_out/cli_startup5.txt 2>&1; grep -E '^##|^wall|jpeg|settings' gpurun_out/cli_startup5.txt | sed -n 1,40p
echo NO_STREAM; CURVIS_NO_JPEG_STREAM=1 python tools/gpu_cli_startup.py 4 2>&1 | grep -A9 'JPEG backgrounds$' | grep -E 'wall|jpeg|settings'
