mkdir -p gpurun_out
(time timeout 1200 python -m pytest tests -m gpu -x -q) > gpurun_out/s6_gpu_tests.txt 2>&1; tail -4 gpurun_out/s6_gpu_tests.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python bench.py > gpurun_out/s6_bench.json 2> gpurun_out/s6_bench.err; tail -c 400 gpurun_out/s6_bench.json
python tools/gpu_cli_startup.py 3 > gpurun_out/cli_startup3.txt 2>&1; grep -E '^##|^wall' gpurun_out/cli_startup3.txt
