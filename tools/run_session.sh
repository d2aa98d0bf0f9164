mkdir -p gpurun_out
python tools/gpu_png_buffer_paths.py > gpurun_out/png_buffer_paths.txt 2>&1; cat gpurun_out/png_buffer_paths.txt
