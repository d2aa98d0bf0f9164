mkdir -p gpurun_out
(time timeout 600 python -m pytest tests/test_gpu_async_download.py -m gpu -x -q) > gpurun_out/s3_tests.txt 2>&1; tail -25 gpurun_out/s3_tests.txt
python tools/gpu_async_download.py 3 > gpurun_out/async_download.txt 2>&1; cat gpurun_out/async_download.txt
HSA_ENABLE_SDMA=0 python tools/gpu_async_download.py 2 > gpurun_out/async_download_nosdma.txt 2>&1; cat gpurun_out/async_download_nosdma.txt
