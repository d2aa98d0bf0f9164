"""bench.py's N > 1 control plane on CPU: two ranks under torch.distributed.run (the launcher's store), gloo process group,
`agree`, `product_comm` and `distribute_skies` with a stub context in place of the GPU -- the decisions the ranks must take
TOGETHER: one rank failing before ncclCommInitRank, the broadcast failing on one rank, and the clean path.
The data plane itself (RCCL) needs GPUs: tests/test_gpu_bench.py, tests/test_gpu_multi_device.py."""
import json
import os
import socket
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import importlib.util, json, os, sys, types
    sys.path.insert(0, %(root)r)
    import numpy as np
    import torch
    import torch.distributed as dist
    import curvis_amd
    from curvis_amd import skies
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(%(root)r, "bench.py"))
    bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
    torch.cuda.synchronize = lambda: None            # no GPU here: the stub context below stands in for HBM

    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    bench.rccl_log_begin(rank)
    SCEN = os.environ["SCENARIO"]
    curvis_amd.Context.rccl_unique_id = staticmethod(lambda: bytes(range(128)))
    curvis_amd.Context.rccl_comm_destroy = staticmethod(lambda comm: None)

    class StubContext:
        # one "GPU": holds the two textures as numpy arrays; the communicator is a token
        def __init__(self): self.sky = [None, None]; self.joined = None
        def rccl_comm_init(self, uid, n, r):
            assert uid == bytes(range(128)) and n == world and r == rank
            if SCEN == "init_fails_on_rank1" and rank == 1: raise RuntimeError("no RCCL on this rank")
            if SCEN == "init_wedges_on_rank1" and rank == 1:
                import time; time.sleep(3600)        # ncclCommInitRank that never returns (daemon thread: dies with the process)
            self.joined = (n, r); return "comm-token"
        def set_sky(self, which, image):
            if SCEN == "upload_fails_on_rank0": raise RuntimeError("hipMalloc failed")
            self.sky[which] = np.ascontiguousarray(image.rgba)
        def bcast_skies(self, comm, root):
            assert comm == "comm-token"
            for w in range(2):                       # the stub moves the bytes over the control plane
                box = [self.sky[w] if rank == root else None]
                dist.broadcast_object_list(box, src=root); self.sky[w] = box[0]
            # a rank that sees its part of the collective fail AFTER taking part (a failed stream synchronisation, say): its
            # peers returned fine -- only the agreement afterwards tells them.  (A rank that never enters a collective leaves its
            # peers inside it; no protocol on top can repair that, with RCCL or with this stub.)
            if SCEN == "bcast_fails_on_rank1" and rank == 1:
                with open(bench.rccl_log_path(rank), "w") as f: f.write("stub NCCL WARN Cuda failure 'invalid device ordinal'\\n")
                raise RuntimeError("sky broadcast, stage texture_broadcast(-l sky): ncclBroadcast failed")
        def set_sky_device(self, which, ptr, w, h, copy=False): self.sky[which] = StubContext.tensors[ptr]
        def read_sky(self, which, off, n): return self.sky[which].reshape(-1)[off:off + n]

    # the fall-back path copies into "device" tensors: map data_ptr -> numpy view (CPU tensors stand in)
    StubContext.tensors = {}
    real_empty = torch.empty
    def fake_empty(*a, **k):
        k.pop("device", None); t = real_empty(*a, **k); StubContext.tensors[t.data_ptr()] = t.numpy(); return t
    torch.empty = fake_empty
    real_ones = torch.ones
    torch.ones = lambda *a, **k: real_ones(*a, **{kk: vv for kk, vv in k.items() if kk != "device"})
    dist_new_group = dist.new_group
    def failing_new_group(*a, **k): raise RuntimeError("no nccl backend on a CPU host")
    dist.new_group = failing_new_group

    ok, why = bench.agree(dist, world, rank == 0, "rank %%d says no" %% rank)
    assert (ok, why) == (False, "rank 1: rank 1 says no")
    assert bench.agree(dist, world, True) == (True, None)

    sw, sh = 256, 128
    host = (skies.smooth(sw, sh, 128), skies.smooth(sw, sh, 32)) if rank == 0 else None
    ctx, keep = StubContext(), []
    info = bench.distribute_skies(ctx, dist, torch, world, rank, host, sw, sh, True, keep)
    for w, blue in ((0, 128), (1, 32)):
        assert np.array_equal(np.asarray(ctx.sky[w]).reshape(sh, sw, 4), skies.smooth(sw, sh, blue))
    out = [None] * world
    dist.all_gather_object(out, {"backend": info["backend"], "fell": info.get("fallback_from"), "verified": info["readback_verified_on_every_rank"],
                                 "detail": info.get("failure_detail")})
    if rank == 0: print(json.dumps(out + [dict(bench.WEDGED)]), flush=True)
    dist.barrier(); dist.destroy_process_group()
    if bench.WEDGED["any"]: os._exit(0)             # what bench.py does after its line: a stuck join must not block the exit
""")


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def run_scenario(tmp_path, scenario):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT})
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), str(script)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=dict(os.environ, OMP_NUM_THREADS="1", SCENARIO=scenario,
                                                                                   CURVIS_BENCH_RCCL_INIT_TIMEOUT="8"))
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("[")][-1])
    run_scenario.wedged = out.pop()
    return out, r.stderr


def test_clean_path_takes_the_products_broadcast(tmp_path):
    out, _ = run_scenario(tmp_path, "clean")
    assert [o["backend"] for o in out] == ["rccl (product ABI)"] * 2 and all(o["fell"] is None and o["verified"] for o in out)


def test_both_ranks_fall_back_when_one_cannot_join(tmp_path):
    """rank 1 fails before ncclCommInitRank: rank 0 (which joined) must NOT go on alone; both report the same reasons"""
    out, err = run_scenario(tmp_path, "init_fails_on_rank1")
    assert [o["backend"] for o in out] == ["gloo"] * 2 and out[0]["fell"] == out[1]["fell"] and all(o["verified"] for o in out)
    assert out[0]["fell"][0] == "curvis_ctx_rccl_comm_init: rank 1: RuntimeError: no RCCL on this rank" and "torch nccl group" in out[0]["fell"][1]
    d = out[0]["detail"]
    assert d == out[1]["detail"] and [(x["rank"], x["stage"]) for x in d] == [(1, "ncclCommInitRank")] and "no RCCL on this rank" in d[0]["error"]
    assert err.count("sky broadcast fell back") == 2


def test_both_ranks_fall_back_when_the_broadcast_fails_on_one(tmp_path):
    out, _ = run_scenario(tmp_path, "bcast_fails_on_rank1")
    assert [o["backend"] for o in out] == ["gloo"] * 2 and out[0]["fell"] == out[1]["fell"]
    assert out[0]["fell"][0].startswith("curvis_ctx_bcast_skies: rank 1: RuntimeError: sky broadcast, stage texture_broadcast(-l sky): ncclBroadcast failed")
    # failure attribution: WHICH rank, WHICH stage, the error, and that rank's own RCCL warnings -- the same
    # table on every rank
    assert out[0]["detail"] == out[1]["detail"] and len(out[0]["detail"]) == 1
    d = out[0]["detail"][0]
    assert d["rank"] == 1 and d["stage"] == "texture_broadcast(-l sky)" and "ncclBroadcast failed" in d["error"]
    assert "invalid device ordinal" in d["rccl_log"]


def test_upload_failure_on_rank0_does_not_strand_the_other_rank(tmp_path):
    """rank 0's upload raises after the communicator is up -- every rank must still issue the same sequence of
    control-plane collectives (no barrier skipped by the rank that raised) and both fall back together"""
    out, _ = run_scenario(tmp_path, "upload_fails_on_rank0")
    assert [o["backend"] for o in out] == ["gloo"] * 2 and out[0]["fell"] == out[1]["fell"] and all(o["verified"] for o in out)
    assert out[0]["fell"][0] == "curvis_ctx_bcast_skies: rank 0: upload on rank 0: RuntimeError: hipMalloc failed"
    assert [(x["rank"], x["stage"]) for x in out[1]["detail"]] == [(0, "upload on the root")]


def test_wedged_join_skips_the_second_rccl_stage_on_every_rank(tmp_path):
    """a rank whose ncclCommInitRank never returns still holds its GPU context: no rank may open torch's nccl group
    then; all go straight to the host-staged broadcast and know that the process must leave through os._exit"""
    out, _ = run_scenario(tmp_path, "init_wedges_on_rank1")
    assert [o["backend"] for o in out] == ["gloo"] * 2 and out[0]["fell"] == out[1]["fell"] and all(o["verified"] for o in out)
    assert "time limit" in out[0]["fell"][0] and out[0]["fell"][1].startswith("torch nccl group: not attempted")
    assert run_scenario.wedged == {"any": True, "here": False}     # rank 0's view: somebody is wedged, not me
    assert [(x["rank"], x["stage"]) for x in out[0]["detail"]] == [(1, "ncclCommInitRank")] and "time limit" in out[0]["detail"][0]["error"]


def test_clean_path_reports_no_failure_detail(tmp_path):
    out, _ = run_scenario(tmp_path, "clean")
    assert all(o["detail"] is None for o in out)


def test_stage_names_are_parsed_from_the_librarys_messages():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_module_stage", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    assert bench.stage_of("CurvisError: sky broadcast, stage header_broadcast: ncclBroadcast: unhandled system error", "x") == "header_broadcast"
    assert bench.stage_of("sky broadcast, stage texture_broadcast(+l sky): stream synchronised; RCCL: ...", "x") == "texture_broadcast(+l sky)"
    assert bench.stage_of("something else", "sky broadcast") == "sky broadcast"
    v, why, wedged = bench.with_time_limit(lambda: 7, 5.0)
    assert (v, why, wedged) == (7, None, False)
    v, why, wedged = bench.with_time_limit(lambda: __import__("time").sleep(30), 0.2)
    assert v is None and wedged and "did not return" in why
