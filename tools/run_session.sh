bash tools/gpu_eff_pixel_profile.sh 2>&1 | tail -40
