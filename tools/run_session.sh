mkdir -p gpurun_out
python tools/gpu_cli_startup.py 4 > gpurun_out/cli_startup4.txt 2>&1; grep -E '^##|^wall|decode|settings' gpurun_out/cli_startup4.txt
echo NO_POPULATE; CURVIS_NO_POPULATE=1 python tools/gpu_cli_startup.py 3 2>&1 | grep -E '^##|^wall|settings'
