mkdir -p gpurun_out
(time timeout 1200 python -m pytest tests -m gpu -x -q) > gpurun_out/s5_gpu_tests.txt 2>&1; tail -4 gpurun_out/s5_gpu_tests.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python bench.py > gpurun_out/s5_bench.json 2> gpurun_out/s5_bench.err; python -c "
import json
d=json.loads(open('gpurun_out/s5_bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['effective_sclk_mhz'], d['roofline']['frac'], d['value_with_download']['fraction_of_value'], d['value_with_download']['overlapped'])"
