mkdir -p gpurun_out
for m in 0 1 0 1; do if [ $m = 1 ]; then export CURVIS_EARLY_INIT=1; else unset CURVIS_EARLY_INIT; fi; echo "EARLY_INIT=$m"; python tools/gpu_cli_startup.py 4 2>&1 | grep -E '^##|^wall|settings|context' | head -9; done
