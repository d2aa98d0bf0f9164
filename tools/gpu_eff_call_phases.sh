#!/bin/bash
# Where ONE context's time per call goes in `curvis video --mode efficient` (CURVIS_DEBUG_TIMING=1: the PNG front end's phases per
# deflate call, the device sampler's jobs): usage (on the GPU box): bash tools/gpu_eff_call_phases.sh [contexts] [batch] [fps]
C=${1:-1}; B=${2:-128}; FPS=${3:-100}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
D=$(mktemp -d /dev/shm/curvis_phases_XXXX 2>/dev/null || mktemp -d)
cd "$ROOT"
python - "$D" "$FPS" <<'PY'
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import refpaths
from curvis_amd import pngio, skies
d, fps = sys.argv[1], float(sys.argv[2])
pngio.write_png(os.path.join(d, "pos.png"), skies.smooth(4096, 2048, 128)[..., :3], level=1)
pngio.write_png(os.path.join(d, "neg.png"), skies.smooth(4096, 2048, 32)[..., :3], level=1)
open(os.path.join(d, "sim.toml"), "w").write("escape_radius = 100.0\nray_integration_max_itarations = 4096\nray_integration_step = 0.05\nsampling_initial_nums = 100\nsampling_max_iterations = 50\nsampling_convergence_threshold_1 = 1e-5\nsampling_convergence_threshold_2 = 1e-5\n")
open(os.path.join(d, "cam.toml"), "w").write("resolution_x = 1920\nresolution_y = 1080\ndiagonal = 43.0\nfocal_length = 15.0\n")
open(os.path.join(d, "vid.toml"), "w").write('video_name = "v"\nframe_rate = %r\nfilepath_to_camera_path = "%s"\n' % (fps, refpaths.reference_path_file("path_orbit.csv")))
PY
mkdir -p "$D/out"
CURVIS_DEBUG_TIMING=1 "$ROOT/curvis_amd/bin/curvis" video "$D/pos.png" "$D/neg.png" "$D/out" -v "$D/vid.toml" -s "$D/sim.toml" -c "$D/cam.toml" \
  --mode efficient --contexts-per-device "$C" --batch "$B" --writers 16 --stats "$D/st.jsonl" > "$D/run.txt" 2>&1
grep -E "deflate|efficient call" "$D/run.txt" | sed -n "9,20p"
python - "$D" <<'PY'
import json, sys, re
d = sys.argv[1]
s = json.load(open(d + "/st.jsonl.summary.json"))
print("frames %d, %.0f frames/s, wall %.3f s" % (s["frames"], s["frames_per_s"], s["wall_s"]))
for dv in s["devices"]:
    print({k: dv[k] for k in ("frames", "batches", "kernel_ms_per_frame", "render_call_ms_per_frame", "gpu_png_kernel_ms_per_frame", "busy_s", "buffer_wait_s", "hand_over_s", "wait_s") if k in dv})
PY
rm -rf "$D"
