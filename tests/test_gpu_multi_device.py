"""Multi-GPU paths on REAL devices.  The first test runs everywhere (one process, one GPU: the RCCL communicator entry points
of the C ABI with a single rank).  Every other test needs two GPUs, uses NO share-device hook, and switches itself on
whenever `curvis_device_count() >= 2` -- on the 1-GPU boxes of this pool they are reported as skipped by name; on a
multi-GPU node they are the proof that RCCL (ncclCommInitRank / ncclCommInitAll + ncclBroadcast inside
curvis_ctx_bcast_skies) moved the textures between two devices and that frames rendered on the second device equal the
oracle's (src/rendering.rs:291-316: frames are independent, k mod N)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import common
import oracle_lib as O
import refpaths
from curvis_amd import paths, pngio, rendering, skies

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HERE = os.path.dirname(os.path.abspath(__file__))
BIN = os.path.join(ROOT, "curvis_amd", "bin", "curvis")
two_gpus = pytest.mark.skipif(common.device_count() < 2, reason="needs two GPUs (this box has %d)" % common.device_count())

sys.path.insert(0, HERE)
import multi_device_worker as W  # noqa: E402  (constants and poses only; importing it starts nothing)


def clean_env():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT",
                                                            "CURVIS_TEST_SHARE_DEVICE", "CURVIS_BENCH_SHARE_DEVICE")}
    return env


def rank_report(tmp_path, world):
    """what every rank's status file says: the stage it reached, its error, RCCL's warnings, its links"""
    lines = []
    for r in range(world):
        try:
            st = json.loads((tmp_path / ("status_%d.json" % r)).read_text())
        except (OSError, ValueError):
            lines.append("rank %d: no status file (the process died before its first stage)" % r)
            continue
        if st.get("ok"):
            lines.append("rank %d (device %s, %s): done; broadcast %.3f s = %.2f GB/s; links %s" % (
                r, st.get("device"), st.get("pci"), st.get("bcast_s", 0.0), st.get("sky_broadcast_gbps", 0.0), json.dumps(st.get("links"))))
        else:
            lines.append("rank %d (device %s, %s): FAILED at stage `%s`: %s\n    links: %s\n    RCCL log: %s" % (
                r, st.get("device"), st.get("pci"), st.get("stage"), st.get("error", "still inside the stage (killed by the time limit?)"),
                json.dumps(st.get("links")), (st.get("rccl_log") or "(empty)").strip()[-1200:]))
    return "\n".join(lines)


def run_ranks(tmp_path, world, env_extra=None, expect_failure=False):
    env = clean_env()
    env.update(env_extra or {})
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "multi_device_worker.py"), str(r), str(world), str(r), str(tmp_path)],
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(world)]
    outs = []
    try:
        for p in procs:
            try:
                outs.append(p.communicate(timeout=600))
            except subprocess.TimeoutExpired:
                outs.append(("", "time limit"))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()  # the exact processes this test started
    report = rank_report(tmp_path, world)
    print(report)                                           # the link table and the broadcast rate of every rank (pytest -s / on failure)
    if expect_failure:
        return report
    for r, p in enumerate(procs):
        assert p.returncode == 0, "first contact failed -- per rank:\n%s\nrank %d stderr: %s" % (report, r, outs[r][1][-1500:])
    return [np.load(os.path.join(tmp_path, "rank_%d.npz" % r)) for r in range(world)]


def check_against_oracle(results, world):
    sp, sn = skies.checker(W.SKY[0], W.SKY[1], seed=0xC0FFEE), skies.checker(W.SKY[0], W.SKY[1], seed=0xBADC0DE)
    om = O.ellis(1.0)
    seen = set()
    for r, res in enumerate(results):
        assert bool(res["readback_ok"]), "rank %d: textures in HBM differ from the root's" % r
        for k, (pos, fwd, up) in enumerate(W.poses()):
            if k % world != r:
                assert "frame_%d" % k not in res
                continue
            oc = O.camera(pos, fwd, up, 15.0, 43.0, W.RES)
            want, _, st = O.render_image(O.CV, om, oc, O.sky(sp), O.sky(sn), 4096, 100.0, 0.05)
            assert np.array_equal(res["frame_%d" % k], want), (r, k)
            assert tuple(res["stats_%d" % k]) == (st.rays, st.steps, st.n_pos, st.n_neg, st.n_none, st.n_oob), (r, k)
            seen.add(k)
    assert seen == set(range(W.N_FRAMES))


def test_one_process_per_gpu_host_single_rank(tmp_path):
    """curvis_rccl_unique_id -> curvis_ctx_rccl_comm_init -> curvis_ctx_bcast_skies -> curvis_rccl_comm_destroy with one
    rank on GPU 0 (what a 1-GPU box can host), frames against the oracle"""
    res = run_ranks(tmp_path, 1)
    check_against_oracle(res, 1)


def test_device_link_query():
    """curvis_device_link: a device to itself, an index that does not exist, and -- when the box has them -- two devices
    (the link type and hop count printed: what the first measured broadcast rate is read against)"""
    import curvis_amd
    me = curvis_amd.Context.device_link(0, 0)
    assert me["link"] == "same device" and me["hops"] == 0 and me["peer_access"] == 1
    with pytest.raises(curvis_amd.CurvisError):
        curvis_amd.Context.device_link(0, 64)
    if common.device_count() >= 2:
        ln = curvis_amd.Context.device_link(0, 1)
        print("device 0 <-> 1:", ln)
        assert ln["link"] in ("xGMI", "PCIe") and ln["hops"] >= 1


@pytest.mark.parametrize("stage", ["ncclCommInitRank", "read-back", "render on device 0"])
def test_a_failing_rank_says_which_stage_broke(tmp_path, stage):
    """the first contact between two devices must be self-diagnosing.  A rank made to fail at a given
    stage leaves a status file naming the stage, the error and its own RCCL log; the test's failure message is built from it."""
    report = run_ranks(tmp_path, 1, {"CURVIS_WORKER_FAIL_STAGE": stage}, expect_failure=True)
    assert "FAILED at stage `%s`: RuntimeError: injected failure at stage %s" % (stage, stage) in report
    assert "RCCL log:" in report and "links:" in report


@two_gpus
def test_two_processes_two_devices_broadcast_and_render(tmp_path):
    """two processes x two DEVICES: the unique id travels through a file, rank 1 holds no skies before the broadcast, both
    read head / middle / tail of both textures back from their own HBM, frames k mod 2 against the oracle"""
    res = run_ranks(tmp_path, 2)
    assert str(res[0]["pci"]) != str(res[1]["pci"]), "both ranks sat on one physical GPU"
    check_against_oracle(res, 2)


@pytest.fixture()
def scene_files(tmp_path):
    sp, sn = common.make_skies(512, 256, "check")
    pngio.write_png(tmp_path / "pos.png", sp)
    pngio.write_png(tmp_path / "neg.png", sn)
    (tmp_path / "sim.toml").write_text("escape_radius = 100.0\nray_integration_max_itarations = 4096\nray_integration_step = 0.05\n"
                                       "sampling_initial_nums = 100\nsampling_max_iterations = 50\n"
                                       "sampling_convergence_threshold_1 = 1e-5\nsampling_convergence_threshold_2 = 2e-5\n")
    (tmp_path / "cam.toml").write_text("resolution_x = 96\nresolution_y = 54\ndiagonal = 43.0\nfocal_length = 15.0\n")
    return tmp_path, sp, sn


@two_gpus
def test_video_two_devices_rccl_broadcast_against_oracle(scene_files):
    """`curvis video --mode brute --devices 2 --sky-broadcast rccl`: one process, two device threads, ncclCommInitAll,
    device 0 uploads and broadcasts, device 1 renders the odd frames from what arrived over xGMI"""
    d, sp, sn = scene_files
    orbit = refpaths.reference_path_file("path_orbit.csv")
    (d / "vid.toml").write_text('video_name = "v"\nframe_rate = 0.2\nfilepath_to_camera_path = "%s"\n' % orbit)
    out = d / "out"
    out.mkdir()
    r = subprocess.run([BIN, "video", str(d / "pos.png"), str(d / "neg.png"), str(out), "-v", str(d / "vid.toml"), "-s", str(d / "sim.toml"),
                        "-c", str(d / "cam.toml"), "--mode", "brute", "--devices", "2", "--batch", "2", "--sky-broadcast", "rccl",
                        "--stats", str(d / "st.jsonl")], env=clean_env(), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    summ = json.loads((d / "st.jsonl.summary.json").read_text())
    assert len(summ["devices"]) == 2 and summ["devices"][0]["pci_bus_id"] != summ["devices"][1]["pci_bus_id"]
    assert all(dv["frames"] > 0 for dv in summ["devices"])
    assert summ["sky_distribution"]["via"].startswith("rccl") and summ["sky_distribution"]["sky_broadcast_gbps"] > 0
    it = rendering.Interpolator.from_file(orbit)
    times = rendering.times_of_frames(it.min_time(), it.max_time(), 0.2)
    recs = {json.loads(ln)["frame"]: json.loads(ln) for ln in (d / "st.jsonl").read_text().splitlines()}
    assert sorted(recs) == list(range(len(times))) and {rc["device"] for rc in recs.values()} == {0, 1}
    om = O.ellis(1.0)
    for k, t in enumerate(times):
        oc = O.camera(tuple(it.camera_position(t)), tuple(it.camera_forward(t)), tuple(it.camera_up(t)), 15.0, 43.0, (96, 54))
        want, _, st = O.render_image(O.CV, om, oc, O.sky(sp), O.sky(sn), 4096, 100.0, 0.05)
        assert np.array_equal(pngio.read_png(out / "tmp" / ("frame_%d.png" % k)), want), k
        assert (recs[k]["rays"], recs[k]["steps"]) == (st.rays, st.steps), k


@two_gpus
def test_image_rows_over_two_devices_against_oracle(scene_files):
    """`curvis image --mode brute --devices 2`: the two row bands of ONE frame rendered on two GPUs"""
    d, sp, sn = scene_files
    out = d / "out_img"
    out.mkdir()
    r = subprocess.run([BIN, "image", str(d / "pos.png"), str(d / "neg.png"), str(out), "-s", str(d / "sim.toml"), "-c", str(d / "cam.toml"),
                        "--mode", "brute", "--devices", "2", "--stats", str(out / "st.json")], env=clean_env(), capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    om, oc, _, _ = common.scene("ellis", res=(96, 54))
    want, _, st = O.render_image(O.CV, om, oc, O.sky(sp), O.sky(sn), 4096, 100.0, 0.05)
    assert np.array_equal(pngio.read_png(out / "output_image.png"), want)
    got = json.loads((out / "st.json").read_text())
    assert (got["rays"], got["steps"]) == (st.rays, st.steps)


@two_gpus
def test_bench_two_gpus_over_the_products_rccl_path():
    """`bench.py --gpus 2` on two devices: no share hook, the skies through curvis_ctx_bcast_skies, every extra of the N > 1
    line present"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--no-traffic",
                        "--no-live-traffic", "--sustained-seconds", "1", "--cpu-row-step", "64", "--sky", "2048"],
                       env=clean_env(), capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["distinct_gpus"] == 2
    col = out["collective"]
    assert col["backend"].startswith("rccl (product ABI)") and "curvis_ctx_bcast_skies" in col["via"] and "fallback_from" not in col
    assert col["readback_verified_on_every_rank"] and col["sky_broadcast_gbps"] > 0
    assert out["cpu_baseline"]["value"] > 0 and out["cpu_baseline"]["cores"] == 1
    rows = out["value_single_image_rows"]
    assert rows["value"] > 0 and [p["rows"] for p in rows["per_rank"]] == [[0, 540], [540, 1080]]
    e2e = out["video_e2e"]
    assert "failed" not in e2e and e2e["frames"] == e2e["frames_on_disk"] >= 8 and e2e["distinct_gpus"] == 2
    assert e2e["sky_distribution"]["via"].startswith("rccl")
