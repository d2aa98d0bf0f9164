"""Host logic of the video path: Interpolator (with the reference's off-by-one), times_of_frames, frame
sharding, and the world_size-2 gloo gather.  CPU only; the renderer is replaced by a recording stub
(the product has no CPU renderer)."""
import ctypes as C
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np
import pytest

import oracle_lib as O
import refpaths
from curvis_amd import paths, rendering

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORBIT = refpaths.reference_path_file("path_orbit.csv")      # the reference's own bytes (tests/golden/paths)
THROUGH = refpaths.reference_path_file("path_through.csv")


def test_interpolator_matches_oracle_including_off_by_one():
    it = rendering.Interpolator.from_file(ORBIT)
    p = O.Path()
    assert O.lib().cvo_load_path(ORBIT.encode(), C.byref(p)) == 0
    times = rendering.times_of_frames(it.min_time(), it.max_time(), 30.0)
    assert len(times) == 1801
    pos, f, u = np.zeros(4), np.zeros(3), np.zeros(3)
    for k, t in enumerate(times):
        rc = O.lib().cvo_path_camera(C.byref(p), t, O._dp(pos), O._dp(f), O._dp(u))
        if rc == 0:
            assert np.array_equal(it.camera_position(t), pos)
            assert np.array_equal(it.camera_forward(t), f) and np.array_equal(it.camera_up(t), u)
        else:
            assert k >= 1799  # README.md:107 "sometimes panics on the last frame"
            with pytest.raises(rendering.InterpolatorPanic):
                it.camera_position(t)
    O.lib().cvo_path_free(C.byref(p))


def test_frame_counts_of_the_baseline_configs():
    o = rendering.Interpolator.from_file(ORBIT)
    t = rendering.Interpolator.from_file(THROUGH)
    assert len(rendering.times_of_frames(o.min_time(), o.max_time(), 4.0)) == 240     # config 4
    assert len(rendering.times_of_frames(t.min_time(), t.max_time(), 24.0)) == 480    # config 5
    assert len(rendering.times_of_frames(t.min_time(), t.max_time(), 30.0)) == 600


def test_sharding_is_a_partition():
    for n, w in [(240, 8), (480, 8), (7, 2), (3, 8), (0, 4)]:
        shards = [rendering.frames_of_rank(n, r, w) for r in range(w)]
        flat = sorted(k for s in shards for k in s)
        assert flat == list(range(n))
        assert max(len(s) for s in shards) - min(len(s) for s in shards) <= 1


WORKER = textwrap.dedent("""
    import os, sys, json
    sys.path.insert(0, %(root)r)
    import numpy as np
    import torch.distributed as dist
    from curvis_amd import rendering, paths, systems

    class StubStats:
        def __init__(self, n, k=0): self.rays = 64 * n; self.steps = 1000 * n + k; self.n_pos = 60 * n; self.n_neg = 3 * n; self.n_none = n; self.n_oob = 0; self.kernel_ms = 1.0 * n
    class StubContext:   # the product has no CPU renderer: the host logic is exercised over a stub of Context
        def __init__(self): self.calls = []; self.eff_calls = 0
        def render_brute(self, metric, cams, max_it, R, delta, download=True):
            cams = [cams] if isinstance(cams, systems.Camera) else list(cams)
            self.calls.append([tuple(c.position) for c in cams])
            return None, StubStats(len(cams))
        def render_efficient(self, metric, cams, max_it, R, delta, n0, maxit, thr1, thr2, download=True):
            assert n0 == maxit == 100 and thr1 == thr2 == 1e-5   # src/main.rs:107, src/rendering.rs:305-306
            self.eff_calls += 1
            return self.render_brute(metric, cams, max_it, R, delta, download)
        def frame_stats(self):   # per-frame counters of the last launch: frame j of the launch gets steps 1000 + j
            return [StubStats(1, j) for j in range(len(self.calls[-1]))]

    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    it = rendering.Interpolator.from_file(%(orbit)r)      # the reference's own path_orbit.csv (tests/golden/paths)
    ctx = StubContext()
    v = rendering.VideoRenderingSystem(None, ctx, it, 4.0, (8, 8), 43.0, 15.0, 100.0, 64, 0.05, rank=rank, world_size=world, batch=7)
    assert v.mode == "efficient"          # what the reference's video loop calls (src/rendering.rs:299-307)
    local = v.render(download=False)
    allstats = rendering.gather_frame_stats(local, dist)
    vb = rendering.VideoRenderingSystem(None, ctx, it, 4.0, (8, 8), 43.0, 15.0, 100.0, 64, 0.05, rank=rank, world_size=world, batch=7, mode="brute")
    n_eff = ctx.eff_calls
    vb.render(download=False)
    assert ctx.eff_calls == n_eff         # brute mode never takes the efficient entry point
    if rank == 0:
        print(json.dumps({"frames": [d["frame"] for d in allstats], "ranks": [d["rank"] for d in allstats],
                          "steps": sum(d["steps"] for d in allstats), "launches0": n_eff,
                          "per_frame_steps": [d["steps"] for d in allstats]}))
    dist.barrier()
    dist.destroy_process_group()
""")


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_two_rank_gloo_video_sharding(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT, "orbit": ORBIT})
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), str(script)]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    import json
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    assert out["frames"] == list(range(240))
    assert out["ranks"] == [k % 2 for k in range(240)]
    assert out["launches0"] == 18  # 120 frames of rank 0 in batches of 7
    # per-frame statistics come from the per-frame counters of each launch, not from batch totals: the j-th frame of
    # a launch carries 1000 + j steps in the stub
    shard_pos = [(k // 2) % 7 for k in range(240)]
    assert out["per_frame_steps"] == [1000 + j for j in shard_pos]
    assert out["steps"] == sum(1000 + j for j in shard_pos)


def test_row_bands_are_a_partition():
    for h, w in [(1080, 8), (2160, 8), (7, 2), (3, 8), (1, 1), (1081, 4)]:
        bands = [rendering.rows_of_rank(h, r, w) for r in range(w)]
        assert bands[0][0] == 0 and sum(c for _, c in bands) == h
        for (b0, c0), (b1, _) in zip(bands, bands[1:]):
            assert b1 == b0 + c0
        assert max(c for _, c in bands) - min(c for _, c in bands) <= 1


ROW_WORKER = textwrap.dedent("""
    import os, sys, json
    sys.path.insert(0, %(root)r)
    import numpy as np
    import torch.distributed as dist
    from curvis_amd import rendering, systems

    class StubStats:
        def __init__(self, rows, w): self.rays = rows * w; self.steps = 10 * rows * w
    class StubContext:   # the product has no CPU renderer: a band is filled with its absolute row numbers
        def render_brute_rows(self, metric, cam, row_begin, row_count, max_it, R, delta, download=True):
            rgb = np.zeros((row_count, cam.resolution_width, 3), np.uint8)
            rgb[:, :, 0] = (np.arange(row_begin, row_begin + row_count) %% 256)[:, None]
            rgb[:, :, 1] = dist.get_rank() + 1
            return rgb, StubStats(row_count, cam.resolution_width)

    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    cam = systems.Camera((0.0, 5.0, 1.5707963267948966, 0.0), (-1.0, 0.0, 0.0), (0.0, 0.0, 1.0), 15.0, 43.0, 16, 37)
    img, info = rendering.render_image_sharded(StubContext(), None, cam, 64, 100.0, 0.05, rank, world, dist)
    if rank == 0:
        ok = img.shape == (37, 16, 3) and bool(np.all(img[:, 0, 0] == np.arange(37) %% 256))
        print(json.dumps({"ok": ok, "owners": [int(v) for v in img[:, 0, 1]], "steps": sum(d["steps"] for d in info),
                          "bands": [[d["row_begin"], d["rows"]] for d in sorted(info, key=lambda d: d["rank"])]}))
    else:
        assert img is None
    dist.barrier()
    dist.destroy_process_group()
""")


def test_two_rank_gloo_row_band_image(tmp_path):
    """single image split by rows over two ranks and assembled on rank 0 (optional split)"""
    script = tmp_path / "row_worker.py"
    script.write_text(ROW_WORKER % {"root": ROOT})
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), str(script)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=dict(os.environ, OMP_NUM_THREADS="1"))
    assert r.returncode == 0, r.stderr[-2000:]
    import json
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["ok"] and out["bands"] == [[0, 19], [19, 18]]
    assert out["owners"] == [1] * 19 + [2] * 18 and out["steps"] == 10 * 37 * 16
