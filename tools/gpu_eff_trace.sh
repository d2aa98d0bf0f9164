#!/bin/bash
# Kernel timeline of `curvis video --mode efficient` at its default operating point (4 contexts per GPU x 32 frames per launch) under
# rocprofv3 --kernel-trace: where the GPU's time goes when four contexts share it (tools/analyze_eff_trace.py reads the CSV).
# usage (on the GPU box): bash tools/gpu_eff_trace.sh [contexts] [batch] [fps]
C=${1:-4}; B=${2:-32}; FPS=${3:-50}
export TMPDIR=/tmp
export CURVIS_SLOW_EXIT=1   # the binary leaves through _Exit otherwise, and the profiler never writes its trace
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
D=$(mktemp -d /dev/shm/curvis_trace_XXXX 2>/dev/null || mktemp -d)
OUT=$ROOT/gpurun_out/eff_trace_c${C}_b${B}; rm -rf "$OUT"; mkdir -p "$OUT"
cd "$ROOT"
python - "$D" "$FPS" <<'PY'
import sys, os
sys.path.insert(0, os.getcwd())
sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
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
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace" -o eff -- "$ROOT/curvis_amd/bin/curvis" video "$D/pos.png" "$D/neg.png" "$D/out" \
  -v "$D/vid.toml" -s "$D/sim.toml" -c "$D/cam.toml" --mode efficient --contexts-per-device "$C" --batch "$B" --writers 16 --stats "$D/st.jsonl" > "$OUT/run.txt" 2>&1
cp "$D/st.jsonl.summary.json" "$OUT/summary.json" 2>/dev/null
cd "$ROOT"
python tools/analyze_eff_trace.py "$OUT" | tee "$OUT/analysis.txt"
rm -rf "$D"
