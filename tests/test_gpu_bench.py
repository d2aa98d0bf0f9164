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


QUICK = ["--no-traffic", "--no-live-traffic", "--cpu-row-step", "64", "--sky", "2048"]


def the_line(r):
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout                      # the JSON line and nothing else on stdout
    return json.loads(lines[0])


def check_extras_of_a_multi_rank_line(out, world, one_device):
    """what every N > 1 line must carry: the CPU baseline, the strong-scaling figure of one image, the
    product's own multi-GPU binary end to end, the per-rank table with the clock of the timed region"""
    cb = out["cpu_baseline"]
    assert cb["value"] > 0 and cb["cores"] == 1 and cb["kind"] == "port" and "every 64th row" in cb["sample"]
    rows = out["value_single_image_rows"]
    assert rows["scaling"] == "strong" and rows["value"] > 0 and len(rows["per_rank"]) == world
    assert [p["rows"] for p in rows["per_rank"]] == [[r * 1080 // world, (r + 1) * 1080 // world] for r in range(world)]
    assert sum(p["steps_per_image"] for p in rows["per_rank"]) == out["config"]["executed_steps_per_frame"]   # the bands ARE the frame
    e2e = out["video_e2e"]
    assert "failed" not in e2e, e2e
    assert e2e["frames"] == e2e["frames_on_disk"] >= 4 * world and e2e["frames_per_s"] > 0 and e2e["value"] > 0
    assert len(e2e["per_device"]) == world and all(dv["frames"] > 0 for dv in e2e["per_device"])
    assert e2e["writer_drain_s"] >= 0 and e2e["sky_distribution"]["seconds"] > 0
    assert e2e["distinct_gpus"] == (1 if one_device else world)
    eff = out["video_e2e_efficient"]                       # the reference's default renderer through the same binary
    assert "failed" not in eff, eff
    assert eff["frames"] == eff["frames_on_disk"] >= 100 * world and eff["gpu_png"] is True and eff["workers"] % world == 0 and world <= eff["workers"] <= 4 * world   # up to four contexts per GPU in this mode (two CPUs per worker thread)
    pr = out["per_rank"]
    assert [p["rank"] for p in pr] == list(range(world))
    for p in pr:
        assert p["ms_per_step"] > 0 and p["kernel_ms_avg"] > 0 and p["value"] > 0 and len(p["sclk_mhz_sysfs_before_after_timed_region"]) == 2
    assert "effective_sclk_mhz" in out and len(out["sclk_mhz_sysfs_before_after"]) == 2
    dl = out["value_with_download"]                          # the reference's output contract: the frame in host memory
    assert 0 < dl["value"] <= out["value"] * 1.05 and dl["ms_per_step"] > 0 and dl["bytes_per_frame"] == 1920 * 1080 * 3
    col = out["collective"]
    assert col["ranks"] == world and col["allreduce_of_ones"] == world and col["readback_verified_on_every_rank"] is True
    assert col["sky_broadcast_ms"] > 0 and col["control_plane"].startswith("gloo")


def test_two_ranks_sharing_the_device():
    """`CURVIS_BENCH_SHARE_DEVICE=1 bench.py --gpus 2`: self-launch under torch.distributed.run, control plane on the
    launcher's store, sky distribution (host-staged: RCCL refuses two ranks on one GPU, and the line says that RCCL was not
    used), read-back check on both ranks, barriers, max-over-ranks timing, ONE JSON line carrying n_gpus = 2, the per-rank
    table, cpu_baseline, value_single_image_rows and video_e2e."""
    r = run_bench(["--gpus", "2", "--steps", "2", "--warmup", "1", "--sustained-seconds", "0.5"] + QUICK, {"CURVIS_BENCH_SHARE_DEVICE": "1"})
    out = the_line(r)
    assert out["n_gpus"] == 2
    check_extras_of_a_multi_rank_line(out, 2, one_device=True)
    pr = out["per_rank"]
    assert pr[0]["pci_bus_id"] == pr[1]["pci_bus_id"] != "" and out["distinct_gpus"] == 1   # the share hook, and it shows
    assert out["collective"]["backend"] == "gloo" and "not attempted" in out["collective"]["fallback_from"][0]
    assert out["value_per_gpu_min"] <= out["value_per_gpu_max"]
    # whole-job value = all ranks' steps / the slowest rank's time
    assert out["value"] <= sum(p["value"] for p in pr) * 1.001
    assert out["value_sustained"]["launches"] >= 1 and out["value_sustained"]["all_ranks"] > 0


def test_two_ranks_agree_on_the_fallback_when_rccl_fails_on_one_of_them():
    """the fall-back is decided TOGETHER.  Two ranks under torchrun (the agent's store: nothing may move
    MASTER_PORT); rank 1 is made to fail before ncclCommInitRank, rank 0 gets out of it by the time limit; both then find
    torch's nccl group unusable (two ranks on one GPU) and both take the host-staged broadcast -- one line, flagged."""
    r = run_bench(["--gpus", "2", "--steps", "1", "--warmup", "0", "--sustained-seconds", "0", "--no-cpu-baseline", "--no-video-e2e",
                   "--no-rows-split", "--multi-frame", "0"] + QUICK,
                  {"CURVIS_BENCH_SHARE_DEVICE": "1", "CURVIS_BENCH_TRY_RCCL": "1", "CURVIS_BENCH_TEST_RCCL_FAIL": "1",
                   "CURVIS_BENCH_RCCL_INIT_TIMEOUT": "8"})
    out = the_line(r)
    col = out["collective"]
    assert out["n_gpus"] == 2 and col["backend"] == "gloo" and col["readback_verified_on_every_rank"] is True
    assert len(col["fallback_from"]) == 2
    assert col["fallback_from"][0].startswith("curvis_ctx_rccl_comm_init: rank ") and "torch nccl group" in col["fallback_from"][1]
    assert "injected RCCL failure" in col["fallback_from"][0] or "time limit" in col["fallback_from"][0]
    assert r.stderr.count("sky broadcast fell back") == 2 and out["value"] > 0      # both ranks said so


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
    out = the_line(r)
    s = out["value_sustained"]
    assert s["seconds"] >= 1.0 and s["launches"] >= 1 and s["value"] > 0
    assert s["samples"] >= 5                              # the sampler ran; the clock itself may be unreadable in a container
    assert "sclk_mhz_median_second_half" in s and len(out["sclk_mhz_sysfs_before_after"]) == 2
    assert out["effective_sclk_mhz"] is None               # no live SQ pass in this run (--no-live-traffic)
    assert out["value_with_download"]["fraction_of_value"] <= 1.05
    assert "collective" not in out and "video_e2e" not in out


def test_default_line_clock_agrees_with_itself_and_carries_the_download_figure():
    """the clock of the timed region comes from the kernel's own cycle count (live SQ pass / this region's
    kernel time) and must agree with what the 10 s run's sysfs samples say; `value_with_download` is on the default line."""
    r = run_bench(["--no-cpu-baseline", "--multi-frame", "0", "--sustained-seconds", "4"], {}, timeout=1200)
    out = the_line(r)
    rf = out["roofline"]
    assert rf["traffic_detail"]["measured_in_this_run"] is True, rf["traffic_detail"].get("live")
    eff = out["effective_sclk_mhz"]
    assert eff == rf["effective_sclk_mhz"] and 1200 < eff < 2600
    # the identity the review checked by hand: value x cycles per wave-step per SIMD / (1024 SIMDs x 64 lanes) = clock
    sq = rf["traffic_detail"]["live"]["sq"]
    implied = out["config"]["executed_steps_per_frame"] / (rf["kernel_ms_avg"] * 1e-3) * sq["cycles_per_wave_step_per_simd"] / (1024 * 64) / 1e6
    assert abs(implied - eff) / eff < 0.002
    assert abs(rf["frac_at_effective_clock"] - rf["frac"] * 2400.0 / eff) < 2e-3
    sus = out["value_sustained"]["sclk_mhz_median_second_half"]
    if sus:                                                 # sysfs readable on this box
        assert abs(eff - sus) / sus < 0.05, (eff, sus)
    dl = out["value_with_download"]
    assert dl["steps"] == out["steps"] and 0.5 < dl["fraction_of_value"] <= 1.02 and dl["delta_ms_per_step"] > -0.5
    ov = dl["overlapped"]                                   # option async_download: the DMA of frame k under the kernel of frame k + 1
    assert "failed" not in ov, ov
    assert ov["value"] >= dl["value"] * 0.99 and 0.5 < ov["fraction_of_value"] <= 1.03, (ov, dl)
    # the reference's DEFAULT renderer on the default N = 1 line -- 1080p frames/s with kernels only (one context,
    # GPU-idle share from its own HIP events) and end to end through `curvis video --mode efficient` on the reference's path_orbit.csv
    ve = out["value_efficient"]
    assert "failed" not in ve, ve
    for key, sampler in (("kernels_only", "device-resident (sampler_kernel)"), ("kernels_only_distinct_radii", "device-resident (sampler_kernel)"),
                         ("host_paced_32", "host-paced")):
        ko = ve[key]
        assert ko["frames"] == 240 and ko["contexts"] == 1 and ko["value"] > 100 and 0 < ko["kernel_ms_per_frame"] < ko["ms_per_frame"] * 1.001
        assert 0.0 <= ko["gpu_idle_share"] < 1.0 and abs(ko["gpu_idle_share"] - (1 - ko["kernel_ms_per_frame"] / ko["ms_per_frame"])) < 2e-3
        assert ko["sampler"].startswith(sampler), (key, ko["sampler"])
    assert ve["kernels_only_prefetched"]["sampler_prefetched"] and ve["kernels_only_prefetched"]["value"] > ve["kernels_only"]["value"]
    assert ve["kernels_only_distinct_radii_prefetched"]["value"] > ve["kernels_only_distinct_radii"]["value"]
    assert ve["kernels_only"]["gpu_idle_share"] < 0.10 and ve["kernels_only"]["value"] > ve["host_paced_32"]["value"]   # < 10 % of the span without a kernel
    ee = ve["end_to_end"]
    assert "failed" not in ee, ee
    assert ee["frames_per_s_workers"] >= ee["frames_per_s"] * 0.9 and ee["frames_requested"] in (3840, 15360) and 0.99 * ee["frames_requested"] <= ee["frames"] == ee["frames_on_disk"] <= ee["frames_requested"] and ee["frames_per_s"] > 100 and ee["gpu_png"] is True and 0.0 <= ee["gpu_idle_share"] < 1.0
    assert "--mode efficient" in ee["command"] and "path_orbit.csv" in ee["command"]


def test_single_rank_over_rccl_takes_the_multi_gpu_code_path():
    """CURVIS_BENCH_FORCE_DIST=1: control plane, the PRODUCT's RCCL path (curvis_rccl_unique_id -> broadcast of the id ->
    curvis_ctx_rccl_comm_init -> curvis_ctx_bcast_skies -> read-back), rows split, `curvis video --devices 1` with its
    single-rank ncclCommInitAll -- the N > 1 code path with the real collective library, on the one GPU a test box has"""
    r = run_bench(["--steps", "2", "--warmup", "1", "--multi-frame", "0", "--sustained-seconds", "0.5"] + QUICK, {"CURVIS_BENCH_FORCE_DIST": "1"})
    out = the_line(r)                                     # RCCL's banner went to stderr
    assert out["n_gpus"] == 1
    check_extras_of_a_multi_rank_line(out, 1, one_device=True)
    col = out["collective"]
    assert col["backend"] == "rccl (product ABI)" and "curvis_ctx_bcast_skies" in col["via"] and "fallback_from" not in col
    assert col["sky_broadcast_gbps"] > 0
    assert out["video_e2e"]["sky_distribution"]["via"].startswith("rccl")
    assert len(out["per_rank"]) == 1 and out["per_rank"][0]["pci_bus_id"] and out["distinct_gpus"] == 1


@pytest.mark.parametrize("inject,backend", [("1", "rccl (torch nccl)"), ("all", "gloo")])
def test_rccl_failure_falls_back_and_says_so(inject, backend):
    """a node on which the product's communicator (or RCCL altogether) cannot be brought up must still produce a (flagged)
    line: torch's nccl broadcast, else skies staged through host memory; `collective.fallback_from` says why"""
    r = run_bench(["--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-video-e2e", "--no-rows-split", "--multi-frame", "0",
                   "--sustained-seconds", "0"] + QUICK, {"CURVIS_BENCH_FORCE_DIST": "1", "CURVIS_BENCH_TEST_RCCL_FAIL": inject})
    out = the_line(r)
    col = out["collective"]
    assert col["backend"] == backend and col["readback_verified_on_every_rank"] is True and out["value"] > 0
    if inject == "1":
        assert "injected RCCL failure" in col["fallback_from"][0] and "fell back" in r.stderr
