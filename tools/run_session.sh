mkdir -p gpurun_out
(time timeout 1200 python -m pytest tests -m gpu -x -q) > gpurun_out/s9_gpu_tests.txt 2>&1; tail -4 gpurun_out/s9_gpu_tests.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python bench.py > gpurun_out/s9_bench.json 2> gpurun_out/s9_bench.err; tail -c 200 gpurun_out/s9_bench.json
python tools/gpu_cli_startup.py 4 > gpurun_out/cli_startup6.txt 2>&1; grep -E '^##|^wall' gpurun_out/cli_startup6.txt
