"""bench.py on a host without a GPU: the contract's defaults, the loud failure (there is no CPU path to fall back to),
and the CPU-baseline leg on a small frame.  The measured paths themselves need an MI355X (`profiles/`)."""
import importlib.util
import os
import subprocess
import sys
import types

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_defaults_follow_the_contract(monkeypatch):
    b = load_bench()
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    a = b.parse()
    assert (a.gpus, a.width, a.height, a.max_iter, a.metric) == (1, 1920, 1080, 4096, "ellis")   # BASELINE configs[1]
    assert a.steps * 11e-3 < 60 and a.warmup >= 1                                                # minutes, not hours
    assert b.FLOP_PER_STEP == {"ellis": 35, "interstellar": 46} and b.ALGO_BYTES_PER_RAY == 7


@pytest.mark.parametrize("extra", [[], ["--gpus", "2"]])
def test_no_gpu_is_a_loud_failure(extra):
    import torch
    if torch.cuda.is_available():
        pytest.skip("this host has a GPU")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "needs a GPU" in r.stderr and r.stdout.strip() == ""          # no JSON line from a run that measured nothing


def test_world_size_must_match_gpus():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr and r.stdout.strip() == ""


def test_cpu_baseline_leg_small_frame():
    b = load_bench()
    from curvis_amd import skies
    args = types.SimpleNamespace(metric="ellis", width=96, height=54, max_iter=4096, cpu_row_step=8)
    out = b.cpu_baseline(args, (skies.smooth(256, 128, 128), skies.smooth(256, 128, 32)))
    assert out["cores"] == 1 and out["kind"] == "port" and out["value"] > 0
    assert out["config1_full"]["rays"] == 256 * 144 and out["config1_full"]["steps"] == 72225185   # config 1, SURVEY KAT
    allc = out["all_cores"]
    assert "failed" not in allc and allc["threads"] >= 1 and allc["value"] > 0
