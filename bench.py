#!/usr/bin/env python3
"""bench.py -- headline benchmark of the per-pixel geodesic hot path on MI355X.

A "step" is one pass of the hot path over one frame of BASELINE.json configs[1]:
Ellis wormhole (rho = 1), 1920x1080, cap 4096 Euler steps, default camera (l = 5, theta = pi/2,
focal 15, diagonal 43), escape radius 100, delta 0.05, two procedural 8192x4096 RGBA8 skies
resident in HBM.  The frame stays in HBM (no D2H inside the timed region).

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Started WITHOUT a launcher (no WORLD_SIZE in the environment) and with --gpus N > 1 it launches itself under
`python -m torch.distributed.run --nproc-per-node N` (one rank per GPU, rendezvous on 127.0.0.1).  It refuses to
run when the node has fewer than N GPUs or when the rank count it ends up with is not --gpus: a run never
reports fewer GPUs than it was asked for.

With N > 1 every rank renders its own K frames (video frames are independent: weak scaling, no
data-path collective); the only communication is the RCCL broadcast of the two sky textures from
rank 0 before the timed region.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import socket
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_PER_STEP = {"ellis": 35, "interstellar": 46}   # live FP64 flop per Euler step
FP64_VECTOR_PEAK_TFLOPS = 78.6                      # MI355X vector FP64 (spec), MFMA not applicable
HBM_PEAK_GBPS = 8000.0                              # MI355X_MICROARCH.md: 8 TB/s spec
ALGO_BYTES_PER_RAY = 7                              # 3 B RGB8 store + 4 B sky texel


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--max-iter", type=int, default=4096)
    ap.add_argument("--metric", default="ellis", choices=["ellis", "interstellar"])
    ap.add_argument("--sky", type=int, default=8192, help="sky width (height = width/2)")
    ap.add_argument("--variant", type=int, default=-1,
                    help="-1 library default (static kernel; relay kernel for big single frames), 1 static, 2 relay, 0 persistent lane-refill")
    ap.add_argument("--refill-threshold", type=int, default=None)
    ap.add_argument("--blocks-per-cu", type=int, default=None)
    ap.add_argument("--fast-math", type=int, default=1, help="1 shared-reciprocal step, 0 compiler IEEE div/sqrt")
    ap.add_argument("--fuse-shade", type=int, default=1, help="1 shade in the integration kernel's epilogue (default)")
    ap.add_argument("--download", action="store_true",
                    help="copy every frame to host memory inside the timed region (PCIe-inclusive rate, "
                         "reported in DESIGN.md; never the headline value)")
    ap.add_argument("--no-traffic", action="store_true",
                    help="report roofline.traffic as null when profiles/traffic.json has no entry for this workload "
                         "(default: such a run fails, so a missing PMC profile cannot go unnoticed)")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="do not re-run this workload under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` after the "
                         "timed region (two short child runs, N = 1 only); roofline.traffic then comes from the "
                         "committed profile alone")
    ap.add_argument("--multi-frame", type=int, default=6,
                    help="frames per launch of the secondary multi-frame measurement (value_multi_frame); 0 = skip")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--sustained-seconds", type=float, default=3.0,
                    help="length of the secondary back-to-back measurement behind `value_sustained` (clock and power "
                         "sampled during it); 0 = skip")
    ap.add_argument("--cpu-row-step", type=int, default=8)
    return ap.parse_args()


def flush_c_stdio():
    """fflush(NULL): whatever native libraries left in the C stdio buffers goes out now, not at exit"""
    import ctypes
    try:
        ctypes.CDLL(None).fflush(None)
    except OSError:
        pass


def self_launch(args):
    """--gpus N > 1 from a plain shell: become `python -m torch.distributed.run --nproc-per-node N bench.py ...`"""
    import torch
    share = os.environ.get("CURVIS_BENCH_SHARE_DEVICE") == "1"
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if have < 1:
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    if have < args.gpus and not share:
        raise SystemExit("bench.py: --gpus %d but this node has %d GPU(s); refusing to measure fewer GPUs than asked "
                         "for (CURVIS_BENCH_SHARE_DEVICE=1 maps every rank to GPU 0 for control-flow tests)" % (args.gpus, have))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stderr.write("bench.py: no launcher in the environment, starting %d ranks: %s\n" % (args.gpus, " ".join(cmd)))
    sys.stderr.flush()
    os.execv(sys.executable, cmd)


def main():
    args = parse()
    if args.gpus < 1:
        raise SystemExit("--gpus must be at least 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        self_launch(args)  # does not return
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d: the JSON line would not describe the run that was "
                         "asked for" % (args.gpus, world))

    import torch  # device memory / streams / torch.distributed plumbing only
    import curvis_amd
    from curvis_amd import skies

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    if os.environ.get("CURVIS_BENCH_SHARE_DEVICE") != "1" and torch.cuda.device_count() < world:
        raise SystemExit("bench.py: %d ranks but %d GPU(s) visible" % (world, torch.cuda.device_count()))
    # test hooks (1-GPU boxes): CURVIS_BENCH_SHARE_DEVICE=1 maps every rank to GPU 0 and
    # CURVIS_BENCH_BACKEND=gloo replaces RCCL, so the N>1 control flow can be exercised on one GPU.
    share_device = os.environ.get("CURVIS_BENCH_SHARE_DEVICE") == "1"
    device_index = 0 if share_device else local_rank
    # RCCL refuses two ranks on one GPU ("Duplicate GPU detected"), so the share-device hook implies gloo
    backend = os.environ.get("CURVIS_BENCH_BACKEND", "gloo" if share_device else "nccl")
    torch.cuda.set_device(device_index)
    dist = None
    # test hook (1-GPU boxes): CURVIS_BENCH_FORCE_DIST=1 takes the N > 1 code path -- process group on RCCL, all-reduce
    # check, sky broadcast, barriers, reductions -- with a single rank
    use_dist = world > 1 or os.environ.get("CURVIS_BENCH_FORCE_DIST") == "1"
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:
            os.environ.setdefault("MASTER_PORT", "29533")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        # RCCL writes a version banner to the C stdout when a communicator comes up: while that happens (process group,
        # first collective below) file descriptor 1 points at stderr, so the job's stdout carries the JSON line only
        flush_c_stdio()
        sys.stdout.flush()
        saved_stdout_fd = os.dup(1)
        os.dup2(2, 1)
        backend_fallback = None
        if backend == "nccl":
            # RCCL first.  If the communicator cannot be brought up on this node (init or the first collective raises), the
            # run is NOT lost: every rank falls back to gloo for the control collectives and the skies go through host
            # memory; the line then says so (`collective.fallback_from`), loudly -- a scaling curve with a flagged
            # broadcast is worth more than no curve.  (A hang inside RCCL cannot be caught this way.)
            try:
                if os.environ.get("CURVIS_BENCH_TEST_RCCL_FAIL") == "1":  # test hook: take the fallback branch
                    raise RuntimeError("injected RCCL failure (CURVIS_BENCH_TEST_RCCL_FAIL)")
                import datetime
                # a collective that cannot complete aborts after 3 minutes instead of the default 10
                dist.init_process_group("nccl", device_id=torch.device("cuda", device_index), timeout=datetime.timedelta(seconds=180))
                probe = torch.ones(1, dtype=torch.float64, device="cuda")
                dist.all_reduce(probe)
                torch.cuda.synchronize()
                ok_local = int(probe.item()) == world
            except Exception as exc:  # noqa: BLE001 -- whatever RCCL / torch raise here
                backend_fallback = "%s: %s" % (type(exc).__name__, str(exc).splitlines()[0][:200] if str(exc) else "")
                ok_local = False
            if not ok_local:
                try:
                    if dist.is_initialized():
                        dist.destroy_process_group()
                except Exception:  # noqa: BLE001
                    pass
                os.environ["MASTER_PORT"] = str(int(os.environ.get("MASTER_PORT", "29533")) + 1)
                dist.init_process_group("gloo")
                backend = "gloo"
                backend_fallback = backend_fallback or "all-reduce of ones did not span the ranks"
                sys.stderr.write("bench.py: rank %d: RCCL unavailable (%s); continuing over gloo\n" % (rank, backend_fallback))
        else:
            dist.init_process_group(backend)

    ctx = curvis_amd.Context(device_index)
    ctx.set_option("variant", args.variant)
    if args.refill_threshold is not None:
        ctx.set_option("refill_threshold", args.refill_threshold)
    if args.blocks_per_cu is not None:
        ctx.set_option("blocks_per_cu", args.blocks_per_cu)
    ctx.set_option("fast_math", args.fast_math)
    ctx.set_option("fuse_shade", args.fuse_shade)
    pci_bus_id = ctx.device_status()["pci_bus_id"]
    if use_dist and world > 1:
        # one rank per PHYSICAL GPU: two ranks on one device would time-share it and the line would still say n_gpus = N
        ids = [None] * world
        dist.all_gather_object(ids, pci_bus_id)
        if len(set(ids)) != world and not share_device:
            raise SystemExit("bench.py: %d ranks on %d distinct GPU(s) %s; refusing (CURVIS_BENCH_SHARE_DEVICE=1 allows it "
                             "for control-flow tests)" % (world, len(set(ids)), sorted(set(ids))))

    # ---- inputs resident in HBM before the timed region: two skies (rank 0 generates, RCCL broadcast)
    sw, sh = args.sky, args.sky // 2
    host_skies = None
    if rank == 0:
        host_skies = (skies.smooth(sw, sh, 128), skies.smooth(sw, sh, 32))
    sky_dev = []
    comm_info = None
    if use_dist:
        # how many ranks the collective backend really spans (an all-reduce of ones on the device), before it is
        # trusted with the skies
        one = torch.ones(1, dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(one)
        comm_info = {"backend": "rccl (torch nccl)" if backend == "nccl" else backend, "ranks": dist.get_world_size(),
                     "allreduce_of_ones": int(one.item()), "sky_bytes_each": sw * sh * 4, "sky_broadcast_ms": []}
        if backend_fallback:
            comm_info["fallback_from"] = "rccl (torch nccl) -> gloo: " + backend_fallback
        if comm_info["allreduce_of_ones"] != world:
            raise SystemExit("bench.py: the %s communicator spans %d ranks, not %d" % (backend, comm_info["allreduce_of_ones"], world))
        flush_c_stdio()
        os.dup2(saved_stdout_fd, 1)  # banner written (to stderr); stdout is the job's again
        os.close(saved_stdout_fd)
    for which in range(2):
        if use_dist:
            t = torch.empty((sh, sw, 4), dtype=torch.uint8, device="cuda")
            if rank == 0:
                t.copy_(torch.from_numpy(host_skies[which]))
            torch.cuda.synchronize()
            dist.barrier()
            tb = time.perf_counter()
            if backend == "nccl":
                dist.broadcast(t, src=0)  # RCCL over xGMI, w*h*4 bytes
            else:  # test hook: stage through host memory
                h = t.cpu()
                dist.broadcast(h, src=0)
                t.copy_(h)
            torch.cuda.synchronize()
            comm_info["sky_broadcast_ms"].append(round((time.perf_counter() - tb) * 1e3, 3))
            if rank != 0:  # the texture really arrived: same closed form as rank 0 generated (first and last row)
                want = skies.smooth(sw, sh, 128 if which == 0 else 32)
                got0, got1 = t[0].cpu().numpy(), t[sh - 1].cpu().numpy()
                if not (np.array_equal(got0, want[0]) and np.array_equal(got1, want[sh - 1])):
                    raise SystemExit("bench.py: rank %d received a corrupted sky texture" % rank)
            ctx.set_sky_device(which, t.data_ptr(), sw, sh, copy=False)
            sky_dev.append(t)  # keep alive
        else:
            ctx.set_sky(which, curvis_amd.SphericalImage(host_skies[which]))

    if args.metric == "ellis":
        metric = curvis_amd.EllisMetric(1.0)
    else:
        metric = curvis_amd.InterstellarMetric(0.1, 1e-4, 1.0)
    cam = curvis_amd.Camera((0.0, 5.0, np.pi / 2, 0.0), (-1.0, 0.0, 0.0), (0.0, 0.0, 1.0), 15.0, 43.0,
                            args.width, args.height)
    R, DELTA = 100.0, 0.05

    # --download: the frame is copied into page-locked host memory (one DMA transfer), as `curvis video` does it
    host_frame = curvis_amd.HostBuffer(args.width * args.height * 3) if args.download else None

    def step():
        _, st = ctx.render_brute(metric, cam, args.max_iter, R, DELTA, download=args.download,
                                 out=host_frame.array if host_frame else None)
        return st

    # setup, not a warm-up step: the first relay launch of a launch shape is checked once against the static kernel by the
    # library ("seat belt"); that one-off check must not land in the timed region when --warmup is 0
    step()
    for _ in range(args.warmup):
        step()

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    fence()
    t0 = time.perf_counter()
    steps_executed = 0
    rays = 0
    kernel_ms = 0.0
    shade_ms = 0.0
    for _ in range(args.steps):
        st = step()
        steps_executed += st.steps
        rays += st.rays
        kernel_ms += st.integrate_ms  # HIP events on the context's own stream, inside the C ABI
        shade_ms += st.shade_ms
    fence()
    elapsed = time.perf_counter() - t0

    # secondary figure, outside the contract's timed region: launches of several frames amortise the ramp and the
    # end-game of a launch (~5 % of a single 1080p frame), which is what a video shard runs as
    multi = None
    if args.multi_frame > 1 and world == 1:
        nf = args.multi_frame
        reps = max(2, (args.steps + nf - 1) // nf)
        ctx.render_brute(metric, [cam] * nf, args.max_iter, R, DELTA, download=False)
        torch.cuda.synchronize()
        tm = time.perf_counter()
        m_steps, m_kernel = 0, 0.0
        for _ in range(reps):
            _, stm = ctx.render_brute(metric, [cam] * nf, args.max_iter, R, DELTA, download=False)
            m_steps += stm.steps
            m_kernel += stm.integrate_ms
        torch.cuda.synchronize()
        dtm = time.perf_counter() - tm
        multi = {"value": round(m_steps / dtm / 1e6, 1), "frames_per_launch": nf, "launches": reps,
                 "ms_per_frame": round(dtm / (reps * nf) * 1e3, 3), "kernel_ms_per_frame": round(m_kernel / (reps * nf), 4),
                 "kernel": ("geodesic_persistent" if args.variant == 0 else
                            "geodesic_relay" if ctx.get_option("last_relay_launches") > 0 else "geodesic_static") +
                           ("<fast>" if args.fast_math else "<strict>"),
                 "note": "same frame %d times per launch; not the contract's `value` (one frame per step)" % nf}
        ctx.render_brute(metric, cam, args.max_iter, R, DELTA, download=False)  # so that last_relay_launches below describes a single-frame launch

    # secondary figure: >= `sustained_seconds` of back-to-back single-frame launches with the shader clock and the board
    # power sampled from sysfs while they run -- the chip is power-limited under sustained FP64 load, and boxes differ
    # by a few per cent in the clock they hold; this is where that shows.  Not the contract's `value`.
    sustained = None
    if args.sustained_seconds > 0:
        sustained = sustained_run(ctx, step, args.sustained_seconds, torch)

    own_elapsed = elapsed
    per_rank = None
    if dist is not None:
        red_dev = "cuda" if backend == "nccl" else "cpu"
        mine = {"rank": rank, "pci_bus_id": pci_bus_id, "device_index": device_index,
                "ms_per_step": round(own_elapsed / args.steps * 1e3, 4),
                "kernel_ms_avg": round(kernel_ms / args.steps, 4),
                "value": round(steps_executed / own_elapsed / 1e6, 1),
                "sclk_mhz": sustained["sclk_mhz_median"] if sustained else None,
                "power_w": sustained["power_w_median"] if sustained else None,
                "value_sustained": sustained["value"] if sustained else None}
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)
        tt = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        agg = torch.tensor([float(steps_executed), float(rays), kernel_ms, shade_ms], dtype=torch.float64,
                           device=red_dev)
        dist.all_reduce(agg, op=dist.ReduceOp.SUM)
        total_steps, total_rays, total_kernel_ms, total_shade_ms = [float(v) for v in agg.tolist()]
    else:
        total_steps, total_rays, total_kernel_ms, total_shade_ms = float(steps_executed), float(rays), kernel_ms, shade_ms

    if rank == 0:
        n_launches = args.steps * world
        kernel_s = total_kernel_ms / 1e3 / n_launches            # average launch duration
        per_launch_steps = total_steps / n_launches
        per_launch_rays = total_rays / n_launches
        flop = FLOP_PER_STEP[args.metric]
        achieved_tflops = per_launch_steps * flop / kernel_s / 1e12
        hbm_gbps = per_launch_rays * ALGO_BYTES_PER_RAY / kernel_s / 1e9
        value = total_steps / elapsed / 1e6
        nominal = total_rays * args.max_iter / elapsed / 1e6
        info = ctx.device_info()
        # which integration kernel the library chose (automatic: relay for big single frames, else static)
        kernel_name = ("geodesic_persistent" if args.variant == 0 else
                       "geodesic_relay" if ctx.get_option("last_relay_launches") > 0 else "geodesic_static")
        traffic, traffic_note = pmc_traffic(args, kernel_name)
        being_profiled = any(k in os.environ for k in ("ROCP_TOOL_LIBRARIES", "ROCPROF_OUTPUT_PATH", "ROCPROFILER_LIBRARY_CTOR"))
        if world == 1 and not args.no_live_traffic and not being_profiled:  # never a profiler inside a profiler
            live_bytes, live = live_traffic(args, kernel_name, per_launch_steps)
            if not isinstance(traffic_note, dict):
                traffic_note = {}
            if live_bytes is not None:
                traffic_note["committed_profile_bytes"] = traffic
                traffic_note.update({k: live[k] for k in ("integrate_kernel_bytes", "integrate_fetch_bytes", "integrate_write_bytes")})
                traffic_note["live"] = live
                traffic_note["measured_in_this_run"] = True
                traffic_note["origin"] = ("bytes, VALU busy and instructions per wave-step: PMC passes made by this run (see "
                                          "`live`); FP64 instruction mix: committed profile (see `source`)")
                traffic = live_bytes
                if "sq" in live:  # the live SQ pass replaces the committed figures below
                    traffic_note["committed_profile_sq"] = {k: traffic_note.get(k) for k in ("valu_busy", "valu_instr_per_wave_step", "salu_instr_per_wave_step")}
                    traffic_note.update({k: live["sq"][k] for k in ("valu_busy", "valu_instr_per_wave_step", "salu_instr_per_wave_step")})
            else:
                traffic_note["live"] = {"failed": live}
        if traffic is None and not args.no_traffic:
            raise SystemExit("bench.py: no HBM traffic figure for this workload: %s; live PMC passes: %s (collect a profile "
                             "with tools/gpu_profile_round.sh + tools/make_profiles.py, or pass --no-traffic to report null)"
                             % (traffic_note.get("note"), traffic_note.get("live", "not attempted")))
        out = {
            "metric": "Mrays/s (pixels x steps/s) at 1920x1080, 4096 steps",
            "value": round(value, 1),
            "unit": "Mray-steps/s (executed Euler steps, all GPUs)",
            "value_nominal_cap": round(nominal, 1),
            "value_note": "single-frame launches (one frame per step): each carries the ramp and end-game tail of a launch",
            "value_multi_frame": multi,
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": ("synthetic (procedural %dx%d RGBA8 skies, default camera/metric settings)" % (sw, sh)) +
                    ("; frames copied to host inside the timed region" if args.download else ""),
            "config": {
                "workload": "%s: %s wormhole, %dx%d, cap %d Euler steps, R=100, delta=0.05, single image; "
                            "%d frame(s)/GPU/step" % (
                                "configs[1]" if (args.metric, args.width, args.height, args.max_iter) == ("ellis", 1920, 1080, 4096)
                                else "configs[2]" if (args.metric, args.width, args.height, args.max_iter) == ("interstellar", 3840, 2160, 8192)
                                else "non-BASELINE variant", args.metric, args.width, args.height, args.max_iter, 1),
                "kernel": kernel_name + ("<fast>" if args.fast_math else "<strict>"),
                "frames_per_gpu": args.steps,
                "rays_per_frame": int(per_launch_rays),
                "executed_steps_per_frame": int(per_launch_steps),
                "device": info["name"],
                "compute_units": info["compute_units"],
            },
            "roofline": {
                "bound": "fp64-valu",
                "achieved": round(achieved_tflops, 3),
                "peak": FP64_VECTOR_PEAK_TFLOPS,
                "unit": "TFLOP/s",
                "frac": round(achieved_tflops / FP64_VECTOR_PEAK_TFLOPS, 4),
                "flop_per_step": flop,
                "kernel": kernel_name,
                "kernel_ms_avg": round(kernel_s * 1e3, 4),
                "shade_kernel_ms_avg": round(total_shade_ms / n_launches, 4),
                "traffic": traffic,
                "traffic_detail": traffic_note,
                "valu_busy_pmc": traffic_note.get("valu_busy") if isinstance(traffic_note, dict) else None,
                "valu_instr_per_wave_step_pmc": traffic_note.get("valu_instr_per_wave_step") if isinstance(traffic_note, dict) else None,
                # what the FP64 pipe really executed: instruction mix from the committed PMC profile (fma counted as
                # two operations) x the steps and the kernel time of THIS run
                "executed_fp64": ({"ops_per_lane_step": traffic_note["fp64_ops_executed_per_lane_step"],
                                   "achieved": round(per_launch_steps * traffic_note["fp64_ops_executed_per_lane_step"] / kernel_s / 1e12, 3),
                                   "unit": "TFLOP/s",
                                   "frac": round(per_launch_steps * traffic_note["fp64_ops_executed_per_lane_step"] / kernel_s / 1e12 / FP64_VECTOR_PEAK_TFLOPS, 4),
                                   "mix_per_wave_step": {k: traffic_note.get("fp64_%s_per_wave_step" % k) for k in ("fma", "mul", "add", "trans")},
                                   "note": "SQ_INSTS_VALU_{FMA,MUL,ADD,TRANS}_F64 of the committed profile; the algorithmic figure above counts a division, a square root or a sine as ONE flop"}
                                  if isinstance(traffic_note, dict) and "fp64_ops_executed_per_lane_step" in traffic_note else None),
                "hbm": {"achieved": round(hbm_gbps, 4), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                        "frac": round(hbm_gbps / HBM_PEAK_GBPS, 8),
                        "note": "7 B/ray algorithmic; the loop is register-resident, HBM fraction is ~0 by construction"},
            },
        }
        if sustained is not None:
            out["value_sustained"] = sustained
        if comm_info is not None:
            bms = comm_info["sky_broadcast_ms"]
            comm_info["sky_broadcast_gbps"] = [round(comm_info["sky_bytes_each"] / (ms * 1e-3) / 1e9, 2) if ms > 0 else None for ms in bms]
            out["collective"] = comm_info
        if per_rank is not None:
            # who is the straggler: the timed region ends when the slowest rank does (max over ranks), so `value` is
            # N x the slowest GPU's rate; the table says which GPU that was and at which clock it ran
            per_rank.sort(key=lambda r: r["rank"])
            out["per_rank"] = per_rank
            out["value_per_gpu_min"] = min(r["value"] for r in per_rank)
            out["value_per_gpu_max"] = max(r["value"] for r in per_rank)
            out["distinct_gpus"] = len(set(r["pci_bus_id"] for r in per_rank))
            if sustained is not None:
                out["value_sustained"]["all_ranks"] = round(sum(r["value_sustained"] or 0.0 for r in per_rank), 1)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args, host_skies)

    # The JSON line must be the LAST thing on the job's stdout: libraries (RCCL's banner) write to the C stdout, which is
    # block-buffered when redirected and would otherwise be flushed at exit, after the line.  So: tear everything down,
    # flush the C streams on every rank, meet once more, and only then rank 0 prints.
    ctx.close()
    if dist is not None:
        dist.barrier()
        flush_c_stdio()
        dist.barrier()
        dist.destroy_process_group()
    flush_c_stdio()
    if rank == 0:
        sys.stdout.write(json.dumps(out) + "\n")
        sys.stdout.flush()


def sustained_run(ctx, step, seconds, torch):
    """back-to-back single-frame launches for at least `seconds`; a sampler thread reads the shader clock and the board
    power from sysfs (curvis_ctx_device_status) every 100 ms meanwhile"""
    import threading
    samples, stop = [], threading.Event()

    def sampler():
        while not stop.is_set():
            st = ctx.device_status()
            samples.append((st["sclk_mhz"], st["power_w"]))
            stop.wait(0.1)
    th = threading.Thread(target=sampler, daemon=True)
    torch.cuda.synchronize()
    th.start()
    t0 = time.perf_counter()
    n, n_steps, k_ms = 0, 0, 0.0
    while True:
        st = step()
        n += 1
        n_steps += st.steps
        k_ms += st.integrate_ms
        if time.perf_counter() - t0 >= seconds:
            break
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    stop.set()
    th.join()
    sclk = sorted(v[0] for v in samples if v[0] > 0)
    powr = sorted(v[1] for v in samples if v[1] > 0)

    def med(v):
        return v[len(v) // 2] if v else None
    return {"value": round(n_steps / dt / 1e6, 1), "unit": "Mray-steps/s (executed), this rank", "launches": n,
            "seconds": round(dt, 2), "ms_per_step": round(dt / n * 1e3, 4), "kernel_ms_avg": round(k_ms / n, 4),
            "sclk_mhz_median": med(sclk), "sclk_mhz_min": sclk[0] if sclk else None, "sclk_mhz_max": sclk[-1] if sclk else None,
            "power_w_median": med(powr), "power_w_max": powr[-1] if powr else None, "samples": len(samples),
            "note": "back-to-back single-frame launches after the contract's timed region; clock and power from sysfs "
                    "(pp_dpm_sclk, hwmon power1_average) every 100 ms; null where sysfs does not tell"}


def pmc_traffic(args, kernel_name):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
    (profiles/traffic.json, written by tools/make_profiles.py from separate --pmc FETCH_SIZE / WRITE_SIZE
    runs of this same command).  Returns (bytes, note) or (None, reason): counters cannot be read from
    inside an un-profiled run."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(path) as f:
            t = json.load(f)
        key = "%s_%dx%d_cap%d_%s" % (args.metric, args.width, args.height, args.max_iter, kernel_name)
        e = t.get(key)
        if e is None:  # the caller decides: the live PMC passes may still deliver, else the run fails unless --no-traffic
            return None, {"measured_in_this_run": False, "note": "no PMC profile committed for " + key, "missing_key": key}
        e = dict(e)
        if kernel_name == "geodesic_relay":  # why the relay kernel moves more than the algorithmic bytes
            st = t.get(key.replace("geodesic_relay", "geodesic_static"))
            e["excess_over_algorithmic"] = (
                "deliberate: in the end-game of a launch unfinished 8x8 tiles are handed from wave to wave through HBM "
                "(64 rays x 48 B parked and reloaded per hand-over, write-through / cache-bypassing) -- the price of "
                "the 4-6 % the hand-over takes off a single-frame launch; it is not re-reading of inputs" +
                ("; the static kernel on the same workload moves %d bytes" % st["integrate_kernel_bytes"] if st else ""))
        e["measured_in_this_run"] = False  # PMC counters cannot be read from inside an un-profiled run
        e["origin"] = "committed profile: rocprofv3 --pmc passes of this same command (see `source`)"
        return e["integrate_kernel_bytes"], e
    except (OSError, ValueError, KeyError) as exc:
        return None, {"measured_in_this_run": False, "note": "profiles/traffic.json unavailable: %s" % exc, "missing_key": "*"}


def live_traffic(args, kernel_name, steps_per_launch):
    """HBM bytes per launch of `kernel_name`, observed in THIS run: two child runs of this same workload (4 launches
    each) under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` -- separate passes, no trace domain next to the
    counters, KiB -> bytes, FETCH doubled (gfx950: 128-byte requests are counted as 64; MI355X_MICROARCH.md).  Runs
    after the timed region; every failure (no rocprofv3, time limit, unreadable output) returns (None, reason) and the
    committed profile is reported instead."""
    import csv
    import glob
    import shutil
    import signal
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    child = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--steps", "3", "--warmup", "1",
             "--width", str(args.width), "--height", str(args.height), "--max-iter", str(args.max_iter),
             "--metric", args.metric, "--sky", str(args.sky), "--variant", str(args.variant),
             "--fast-math", str(args.fast_math), "--fuse-shade", str(args.fuse_shade), "--multi-frame", "0",
             "--sustained-seconds", "0", "--no-traffic", "--no-live-traffic", "--no-cpu-baseline"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "CURVIS_BENCH_FORCE_DIST")}
    env["TMPDIR"] = "/tmp"
    SQ = ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_ACTIVE_INST_VALU", "GRBM_GUI_ACTIVE")

    def one_pass(counters):
        """{counter: (median over the launches, launches)} of one child run, or a string saying why not"""
        d = tempfile.mkdtemp(prefix="curvis_pmc_", dir="/tmp")
        try:
            proc = subprocess.Popen([exe, "--pmc"] + list(counters) + ["--output-format", "csv", "-d", d, "-o", "pmc", "--"] + child,
                                    cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                                    start_new_session=True)
            try:
                rc = proc.wait(timeout=150)
            except subprocess.TimeoutExpired:
                os.killpg(proc.pid, signal.SIGKILL)  # the group this call started, nothing else
                proc.wait()
                return "rocprofv3 --pmc %s exceeded 150 s" % " ".join(counters)
            if rc != 0:
                return "rocprofv3 --pmc %s exited with %d" % (" ".join(counters), rc)
            vals = {c: [] for c in counters}
            for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                with open(path) as f:
                    for r in csv.DictReader(f):
                        if kernel_name in r.get("Kernel_Name", "") and r.get("Counter_Name") in vals:
                            vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
            res = {}
            for c, v in vals.items():
                if not v:
                    return "no %s rows for %s in the rocprofv3 output" % (c, kernel_name)
                v.sort()  # median over the launches: the first touch of a buffer shows a several-fold FETCH_SIZE
                n = len(v)
                res[c] = (v[n // 2] if n % 2 else 0.5 * (v[n // 2 - 1] + v[n // 2]), n)
            return res
        except OSError as exc:
            return "rocprofv3 --pmc %s: %s" % (" ".join(counters), exc)
        finally:
            shutil.rmtree(d, ignore_errors=True)

    t0 = time.perf_counter()
    got = {}
    for counters in (("FETCH_SIZE",), ("WRITE_SIZE",)):
        res = one_pass(counters)
        if isinstance(res, str):
            return None, res
        got.update(res)
    fetch_b, write_b = got["FETCH_SIZE"][0] * 1024.0 * 2.0, got["WRITE_SIZE"][0] * 1024.0
    live = {
        "integrate_kernel_bytes": int(fetch_b + write_b), "integrate_fetch_bytes": int(fetch_b), "integrate_write_bytes": int(write_b),
        "launches_sampled": {"FETCH_SIZE": got["FETCH_SIZE"][1], "WRITE_SIZE": got["WRITE_SIZE"][1]},
        "source": "this run: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, two separate child runs of the same workload "
                  "(1 warm-up + 3 launches each, median over the launches, KiB -> bytes, FETCH doubled per the gfx950 note in "
                  "MI355X_MICROARCH.md)"}
    # third pass, SQ counters: what the issue-bound roofline is argued from (VALU instructions per wave-step, VALU busy)
    sq = one_pass(SQ)
    if isinstance(sq, str):
        live["sq_failed"] = sq
    else:
        wave_steps = steps_per_launch / 64.0
        gui = sq["GRBM_GUI_ACTIVE"][0] / 8.0  # summed over the 8 XCDs
        live["sq"] = {"valu_instr_per_wave_step": round(sq["SQ_INSTS_VALU"][0] / wave_steps, 1),
                      "salu_instr_per_wave_step": round(sq["SQ_INSTS_SALU"][0] / wave_steps, 1),
                      "valu_busy": round(4 * sq["SQ_ACTIVE_INST_VALU"][0] / (1024 * gui), 4),
                      "shader_cycles_per_launch": int(gui),
                      "cycles_per_wave_step_per_simd": round(gui * 1024 / wave_steps, 1),
                      "source": "this run: rocprofv3 --pmc " + " ".join(SQ) + " (third child run); busy = 4 x "
                                "SQ_ACTIVE_INST_VALU / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs)"}
    live["seconds"] = round(time.perf_counter() - t0, 1)
    return int(fetch_b + write_b), live


def cpu_baseline(args, host_skies):
    """The oracle (libm flavour = what a Linux build of the single-threaded Rust reference calls), one
    thread, on every `cpu_row_step`-th row of the same frame.  Checker code used as a timed baseline only."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    om = O.ellis(1.0) if args.metric == "ellis" else O.interstellar(0.1, 1e-4, 1.0)
    oc = O.camera(res=(args.width, args.height))
    sp, sn = O.sky(host_skies[0]), O.sky(host_skies[1])
    t0 = time.perf_counter()
    _, _, st = O.render_image(O.LIBM, om, oc, sp, sn, args.max_iter, 100.0, 0.05, row_begin=0,
                              row_step=args.cpu_row_step)
    dt = time.perf_counter() - t0
    # "C1 in full" -- the reference's default image (256x144, cap 40 000) end to end on the same core
    c1 = None
    if args.metric == "ellis":
        t1 = time.perf_counter()
        _, _, st1 = O.render_image(O.LIBM, O.ellis(1.0), O.camera(res=(256, 144)), sp, sn, 40000, 100.0, 0.05)
        d1 = time.perf_counter() - t1
        c1 = {"workload": "config 1 in full: 256x144, cap 40000", "rays": int(st1.rays), "steps": int(st1.steps),
              "seconds": round(d1, 2), "value": round(st1.steps / d1 / 1e6, 2)}
    # SURVEY 8d, optional: the same restatement on the host cores this container may use (rows striped over threads,
    # ctypes drops the GIL; as many threads as the cgroup's CPU quota allows) -- NOT the reference's configuration
    # (it is single-threaded, README.md:110), shown for scale only
    allc = None
    try:
        from concurrent.futures import ThreadPoolExecutor
        quota = None  # what the container may really use: CPU affinity and the cgroup's CPU quota, if any
        try:
            with open("/sys/fs/cgroup/cpu.max") as f:
                q = f.read().split()
            quota = None if q[0] == "max" else round(float(q[0]) / float(q[1]), 2)
        except (OSError, ValueError, IndexError):
            try:  # cgroup v1
                with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f1, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f2:
                    q1, q2 = float(f1.read()), float(f2.read())
                quota = round(q1 / q2, 2) if q1 > 0 else None
            except (OSError, ValueError):
                pass
        usable = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        T = max(1, min(64, int(quota) if quota else usable // 2))  # the quota if there is one, else one per physical core
        stride = T * 2  # every second row of the frame in all: thread i takes rows 2i, 2i + 2T, ...

        def band(i):
            _, _, s_ = O.render_image(O.LIBM, om, oc, sp, sn, args.max_iter, 100.0, 0.05, row_begin=2 * i, row_step=stride)
            return int(s_.rays), int(s_.steps)
        ta = time.perf_counter()
        with ThreadPoolExecutor(T) as ex:
            parts = list(ex.map(band, range(T)))
        da = time.perf_counter() - ta
        allc = {"value": round(sum(p[1] for p in parts) / da / 1e6, 1), "threads": T,
                "speedup_over_one_thread": round(sum(p[1] for p in parts) / da / (st.steps / dt), 1),
                "cpus_in_affinity_mask": usable,
                "cgroup_cpu_quota": quota,
                "sample": "every 2nd row of the frame striped over %d threads: %d rays, %d Euler steps, %.1f s" % (
                    T, sum(p[0] for p in parts), sum(p[1] for p in parts), da),
                "note": "not the reference's configuration (single-threaded); for scale only; the thread count follows the "
                        "container's CPU quota, not the machine's core count"}
    except Exception as exc:  # a baseline extra must never cost the bench line
        allc = {"failed": str(exc)}
    model = ""
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return {
        "value": round(st.steps / dt / 1e6, 2),
        "unit": "Mray-steps/s (executed)",
        "cores": 1,
        "kind": "port",
        "sample": "every %dth row of the same %dx%d frame: %d rays, %d Euler steps, %.1f s" % (
            args.cpu_row_step, args.width, args.height, st.rays, st.steps, dt),
        "host_cpu": model,
        "host_logical_cpus": os.cpu_count(),
        "config1_full": c1,
        "all_cores": allc,
    }


if __name__ == "__main__":
    main()
