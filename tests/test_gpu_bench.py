"""bench.py on the GPU box: the N > 1 control flow with both ranks on the one GPU a test box has (the driver measures
real scaling on an 8-GPU node; this keeps the code that produces that line exercised every round)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def run_bench(extra, env_extra, timeout=900):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, env=env, capture_output=True, text=True,
                       timeout=timeout)
    return r


def test_two_ranks_sharing_the_device():
    """`CURVIS_BENCH_SHARE_DEVICE=1 bench.py --gpus 2`: self-launch under torch.distributed.run, process group (gloo, as RCCL
    refuses two ranks on one GPU), sky broadcast from rank 0 with read-back check on rank 1, barriers, max-over-ranks
    timing, ONE JSON line carrying n_gpus = 2, the collective's rank count and the per-rank table."""
    r = run_bench(["--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-traffic", "--no-live-traffic",
                   "--sustained-seconds", "0.5"], {"CURVIS_BENCH_SHARE_DEVICE": "1"})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout                      # the JSON line and nothing else on stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["collective"]["ranks"] == 2 and out["collective"]["allreduce_of_ones"] == 2
    assert len(out["collective"]["sky_broadcast_gbps"]) == 2
    pr = out["per_rank"]
    assert [p["rank"] for p in pr] == [0, 1]
    assert pr[0]["pci_bus_id"] == pr[1]["pci_bus_id"] != "" and out["distinct_gpus"] == 1   # the share hook, and it shows
    for p in pr:
        assert p["ms_per_step"] > 0 and p["kernel_ms_avg"] > 0 and p["value"] > 0
    assert out["value_per_gpu_min"] <= out["value_per_gpu_max"]
    # whole-job value = all ranks' steps / the slowest rank's time
    assert out["value"] <= sum(p["value"] for p in pr) * 1.001
    assert out["value_sustained"]["launches"] >= 1 and out["value_sustained"]["all_ranks"] > 0


def test_two_ranks_on_one_gpu_are_refused_without_the_hook():
    """without the hook a 2-rank run on a 1-GPU box must not produce a line (it would claim n_gpus = 2 for one GPU)"""
    # NOT `import torch` here: torch brings its own copies of the HIP runtime and RCCL into the pytest process, after which
    # tests that open librccl.so / libamdhip64.so through ctypes find "no ROCm-capable device"
    from curvis_amd import _abi
    if _abi.lib().curvis_device_count() >= 2:
        pytest.skip("this box has two GPUs")
    r = run_bench(["--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-traffic", "--no-live-traffic"], {})
    assert r.returncode != 0 and r.stdout.strip() == "" and "refusing" in r.stderr


def test_single_rank_line_carries_sustained_figure_and_device_identity(gpu_ctx):
    st = gpu_ctx.device_status()
    assert len(st["pci_bus_id"].split(":")) == 3          # dddd:bb:dd.f
    r = run_bench(["--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-traffic", "--no-live-traffic", "--multi-frame", "0",
                   "--sustained-seconds", "1"], {})
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    s = out["value_sustained"]
    assert s["seconds"] >= 1.0 and s["launches"] >= 50 and s["value"] > 0.5 * out["value"]
    assert s["samples"] >= 5                              # the sampler ran; the clock itself may be unreadable in a container


def test_single_rank_over_rccl_takes_the_multi_gpu_code_path():
    """CURVIS_BENCH_FORCE_DIST=1: process group on RCCL (torch `nccl`), all-reduce span check, object gathers of the
    per-rank records, device-to-device sky broadcast, barriers and reductions -- the N > 1 code path with the real
    collective library, on the one GPU a test box has"""
    r = run_bench(["--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-traffic", "--no-live-traffic", "--multi-frame", "0",
                   "--sustained-seconds", "0.5"], {"CURVIS_BENCH_FORCE_DIST": "1"})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout                      # RCCL's banner went to stderr
    out = json.loads(lines[0])
    assert out["n_gpus"] == 1 and out["collective"]["backend"].startswith("rccl") and out["collective"]["ranks"] == 1
    assert len(out["collective"]["sky_broadcast_gbps"]) == 2 and all(v is None or v > 0 for v in out["collective"]["sky_broadcast_gbps"])
    assert len(out["per_rank"]) == 1 and out["per_rank"][0]["pci_bus_id"] and out["distinct_gpus"] == 1


def test_rccl_failure_falls_back_to_gloo_and_says_so():
    """a node on which RCCL cannot be brought up must still produce a (flagged) line: control collectives over gloo, skies
    staged through host memory, `collective.fallback_from` set"""
    r = run_bench(["--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-traffic", "--no-live-traffic", "--multi-frame", "0",
                   "--sustained-seconds", "0"], {"CURVIS_BENCH_FORCE_DIST": "1", "CURVIS_BENCH_TEST_RCCL_FAIL": "1"})
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["collective"]["backend"] == "gloo" and "injected RCCL failure" in out["collective"]["fallback_from"]
    assert "RCCL unavailable" in r.stderr and out["value"] > 0
