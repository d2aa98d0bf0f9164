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
data-path collective); the only communication on the data path is the RCCL broadcast of the two sky
textures from rank 0 before the timed region, made by the PRODUCT's entry point (curvis_ctx_bcast_skies
on a communicator from curvis_ctx_rccl_comm_init; the ncclUniqueId travels over the control plane).
The control plane (barriers, the max-over-ranks reduction, the per-rank table) is a gloo process group
on the launcher's store: it comes up wherever torchrun does, and it is what the ranks use to AGREE on a
fall-back (torch's nccl broadcast, then host-staged gloo) should RCCL fail on any of them -- the line
then says so in `collective.via` / `collective.fallback_from`.  Rank 0 prints ONE JSON line; for N > 1
it also carries `cpu_baseline`, `value_single_image_rows` (ONE configs[1] image split by rows over the
N GPUs) and `video_e2e` (`curvis video --mode brute --devices N` on a 16N-frame rendition of configs[3]).
At N = 1 the line also carries `value_efficient`: the reference's DEFAULT renderer (render_image_efficient, what its CLI
runs) at configs[3]'s poses -- frames/s with kernels only (one context, GPU-idle share from its HIP events) and end to end
through `curvis video --mode efficient` on the reference's own path_orbit.csv; outside the contract's timed region.
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
    ap.add_argument("--no-video-e2e", action="store_true",
                    help="N > 1: skip the `curvis video --mode brute --devices N` end-to-end figure (video_e2e)")
    ap.add_argument("--video-e2e-frames-per-gpu", type=int, default=16)
    ap.add_argument("--no-rows-split", action="store_true",
                    help="N > 1: skip value_single_image_rows (one image split by rows over the ranks)")
    ap.add_argument("--sustained-seconds", type=float, default=10.0,
                    help="length of the secondary back-to-back measurement behind `value_sustained` (clock and power "
                         "sampled during it); 0 = skip")
    ap.add_argument("--cpu-row-step", type=int, default=8)
    ap.add_argument("--no-value-efficient", action="store_true",
                    help="N = 1: skip value_efficient (the reference's default renderer: kernels only + `curvis video --mode efficient`)")
    ap.add_argument("--value-efficient-frames", type=int, default=0,
                    help="frames of the end-to-end leg of value_efficient (path_orbit.csv resampled to that many frames); 0 = 15 360 "
                         "when /dev/shm has 32 GB free for the 9.5 GB of PNG files and the container's memory limit is above 64 GB (about 1.3 s of rendering: long enough for the binary's fixed "
                         "start-up costs, ~0.15 s, to stop dominating), else 3 840")
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
    # the product's own device count (no `import torch` in a process that is about to exec the launcher: on a box with a cold
    # page cache that import alone costs tens of seconds)
    from curvis_amd import _abi
    share = os.environ.get("CURVIS_BENCH_SHARE_DEVICE") == "1"
    have = max(0, int(_abi.lib().curvis_device_count()))
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

    os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")  # one node: RCCL's bootstrap needs no network interface (torch's nccl group included)
    if world > 1 or os.environ.get("CURVIS_BENCH_FORCE_DIST") == "1":
        rccl_log_begin(rank)  # RCCL's own warnings per rank, quoted by `collective.failure_detail` should the data plane fail
    # The job's stdout carries the JSON line and nothing else: gloo ("[Gloo] Rank 0 is connected to ..."), RCCL (version
    # banner) and whatever else writes to the C stdout go to stderr -- file descriptor 1 points there until the line is due.
    flush_c_stdio()
    sys.stdout.flush()
    job_stdout_fd = os.dup(1)
    os.dup2(2, 1)

    import torch  # device memory / streams / torch.distributed plumbing only
    import curvis_amd
    from curvis_amd import skies

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    # a launcher may give every rank ONE visible device of its own (HIP_VISIBLE_DEVICES per rank): then each rank takes device 0,
    # and the PCI identity check below still refuses ranks that really share a GPU
    visible = torch.cuda.device_count()
    one_device_per_rank = world > 1 and visible == 1 and os.environ.get("CURVIS_BENCH_SHARE_DEVICE") != "1"
    if os.environ.get("CURVIS_BENCH_SHARE_DEVICE") != "1" and visible < world and not one_device_per_rank:
        raise SystemExit("bench.py: %d ranks but %d GPU(s) visible" % (world, visible))
    # test hooks (1-GPU boxes): CURVIS_BENCH_SHARE_DEVICE=1 maps every rank to GPU 0; RCCL refuses two ranks on one GPU
    # ("Duplicate GPU detected"), so with it RCCL is only attempted when CURVIS_BENCH_TRY_RCCL=1 (which then exercises
    # the agreed fall-back with two real ranks); CURVIS_BENCH_BACKEND=gloo skips RCCL altogether.
    share_device = os.environ.get("CURVIS_BENCH_SHARE_DEVICE") == "1"
    device_index = 0 if (share_device or one_device_per_rank) else local_rank
    try_rccl = os.environ.get("CURVIS_BENCH_BACKEND", "nccl") != "gloo" and (
        not share_device or os.environ.get("CURVIS_BENCH_TRY_RCCL") == "1")
    torch.cuda.set_device(device_index)
    dist = None
    # test hook (1-GPU boxes): CURVIS_BENCH_FORCE_DIST=1 takes the N > 1 code path -- control plane, RCCL communicator,
    # sky broadcast through the product's entry point, barriers, reductions -- with a single rank
    use_dist = world > 1 or os.environ.get("CURVIS_BENCH_FORCE_DIST") == "1"
    if use_dist:
        import datetime
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1 and "MASTER_PORT" not in os.environ:
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
        if world == 1:
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        # Control plane: gloo on the launcher's store (under torchrun every rank is a client of the agent's TCPStore, so
        # nothing here may move MASTER_PORT or re-initialise the default group).  The long timeout covers rank 0's
        # CPU baseline and the video end-to-end run, during which the other ranks wait at a barrier.
        dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=1800))

    # where the wall time of the whole run goes (rank 0's view; seconds since the process started its work), in the line as
    # `phase_seconds`: start-up of a communicator, the CPU baseline and the extras dwarf the 0.2 s timed region
    t_run0 = time.perf_counter()
    phases = {}

    def phase(name, _last=[t_run0]):
        now = time.perf_counter()
        phases[name] = round(phases.get(name, 0.0) + now - _last[0], 3)
        _last[0] = now
    ctx = curvis_amd.Context(device_index)
    phase("imports_context_process_group")
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
        comm_info = distribute_skies(ctx, dist, torch, world, rank, host_skies, sw, sh, try_rccl, sky_dev)
    else:
        for which in range(2):
            ctx.set_sky(which, curvis_amd.SphericalImage(host_skies[which]))
    phase("skies_generated_and_distributed")

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
    render_why = None
    try:
        step()
    except Exception as exc:  # noqa: BLE001 -- first render on this device: say which rank / device, on every rank
        render_why = short(exc)
    if dist is not None:
        bad = gather_failures(dist, world, rank, None if render_why is None else {
            "stage": "%s on device %s" % (STAGES[7], pci_bus_id), "error": render_why})
        if bad:
            raise SystemExit("bench.py: the first render failed: %s" % json.dumps(bad + FAILURES))
    elif render_why is not None:
        raise SystemExit("bench.py: the first render failed on device %s: %s" % (pci_bus_id, render_why))
    for _ in range(args.warmup):
        step()

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    fence()
    phase("priming_and_warmup")
    # shader clock / board power around the timed region: ONE sysfs read before and one after (no sampler thread inside the
    # region: it would add host jitter to `value`; the sysfs level lags anyway -- the clock the region really ran at comes
    # from the kernel's cycle count, `effective_sclk_mhz` below)
    status_before = ctx.device_status()
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
    status_after = ctx.device_status()
    timed_clock = {"sclk_mhz_sysfs_before_after": [status_before["sclk_mhz"], status_after["sclk_mhz"]],
                   "power_w_before_after": [status_before["power_w"], status_after["power_w"]]}
    phase("timed_region")

    # The reference's output contract: render_image RETURNS an owned host image (src/systems.rs:314-329).  The same
    # launches with the frame copied into page-locked host memory (curvis_host_alloc: one DMA transfer per frame, as
    # `curvis video` does it) -- outside the contract's timed region, reported beside `value`, never as `value`.
    with_download = None
    if not args.download:
        hb = curvis_amd.HostBuffer(args.width * args.height * 3)

        def step_dl():
            _, st_dl = ctx.render_brute(metric, cam, args.max_iter, R, DELTA, download=True, out=hb.array)
            return st_dl
        step_dl()
        fence()
        td = time.perf_counter()
        dl_steps = 0
        for _ in range(args.steps):
            dl_steps += step_dl().steps
        fence()
        dl_elapsed = time.perf_counter() - td
        if dist is not None:
            tdl = torch.tensor([dl_elapsed], dtype=torch.float64)
            dist.all_reduce(tdl, op=dist.ReduceOp.MAX)
            dl_elapsed = float(tdl.item())
            sdl = torch.tensor([float(dl_steps)], dtype=torch.float64)
            dist.all_reduce(sdl, op=dist.ReduceOp.SUM)
            dl_steps = float(sdl.item())
        with_download = {"value": round(dl_steps / dl_elapsed / 1e6, 1), "ms_per_step": round(dl_elapsed / args.steps * 1e3, 3),
                         "bytes_per_frame": args.width * args.height * 3, "steps": args.steps,
                         "note": "the same %d single-frame launches, each frame DMA'd into a curvis_host_alloc (page-locked) buffer before the "
                                 "next launch: what RelativisticSystem::render_image's caller gets (an owned host image, "
                                 "src/systems.rs:314-329); measured right after the contract's timed region, max over ranks" % args.steps}
        # the same again with the copy of frame k running under the kernel of frame k + 1 (option "async_download": copy
        # stream + second frame buffer; two host buffers take turns, the last download is waited for INSIDE the region)
        hb2 = curvis_amd.HostBuffer(args.width * args.height * 3)
        try:
            ctx.set_option("async_download", 1)
            bufs = (hb, hb2)
            ctx.render_brute(metric, cam, args.max_iter, R, DELTA, download=True, out=hb2.array)
            ctx.download_wait()
            fence()
            to = time.perf_counter()
            ov_steps = 0
            for k in range(args.steps):
                _, st_ov = ctx.render_brute(metric, cam, args.max_iter, R, DELTA, download=True, out=bufs[k & 1].array)
                ov_steps += st_ov.steps
            ctx.download_wait()
            fence()
            ov_elapsed = time.perf_counter() - to
            ctx.set_option("async_download", 0)
            if dist is not None:
                tov = torch.tensor([ov_elapsed], dtype=torch.float64)
                dist.all_reduce(tov, op=dist.ReduceOp.MAX)
                ov_elapsed = float(tov.item())
                sov = torch.tensor([float(ov_steps)], dtype=torch.float64)
                dist.all_reduce(sov, op=dist.ReduceOp.SUM)
                ov_steps = float(sov.item())
            with_download["overlapped"] = {
                "value": round(ov_steps / ov_elapsed / 1e6, 1), "ms_per_step": round(ov_elapsed / args.steps * 1e3, 3),
                "note": "option async_download = 1: the DMA of frame k runs on the context's copy stream under the kernel of frame "
                        "k + 1 (second frame buffer in HBM, two page-locked host buffers taking turns; the last download is waited "
                        "for inside the region)"}
        except Exception as exc:  # noqa: BLE001 -- a secondary figure must not cost the line
            with_download["overlapped"] = {"failed": short(exc)}
            try:
                ctx.set_option("async_download", 0)
            except Exception:  # noqa: BLE001
                pass
        del hb, hb2
        phase("with_download")

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
    phase("multi_frame_and_sustained")

    # secondary figure, N > 1: ONE image of the same workload split by rows over the ranks (curvis_render_brute_rows, bit
    # for bit the rows of the whole frame) -- strong scaling of a single image, next to the weak scaling of `value`
    rows_split = None
    if dist is not None and (world > 1 or os.environ.get("CURVIS_BENCH_FORCE_DIST") == "1") and not args.no_rows_split:
        rows_split = rows_split_run(ctx, dist, torch, world, rank, metric, cam, args, R, DELTA, fence)
        phase("single_image_rows")

    own_elapsed = elapsed
    per_rank = None
    if dist is not None:
        mine = {"rank": rank, "pci_bus_id": pci_bus_id, "device_index": device_index,
                "ms_per_step": round(own_elapsed / args.steps * 1e3, 4),
                "kernel_ms_avg": round(kernel_ms / args.steps, 4),
                "value": round(steps_executed / own_elapsed / 1e6, 1),
                "sclk_mhz_sysfs_before_after_timed_region": timed_clock["sclk_mhz_sysfs_before_after"],
                "power_w_before_after_timed_region": timed_clock["power_w_before_after"],
                "sclk_mhz": sustained["sclk_mhz_median"] if sustained else None,
                "power_w": sustained["power_w_median"] if sustained else None,
                "value_sustained": sustained["value"] if sustained else None}
        # how this rank's GPU is connected to the other ranks' (xGMI or PCIe, hops): what the first measured
        # sky_broadcast_gbps has to be read against (xGMI: 7 links x ~153 GB/s per GPU, point to point)
        devs = [None] * world
        dist.all_gather_object(devs, device_index)
        if one_device_per_rank or share_device:
            mine["links"] = None  # this process sees one device only (launcher's device mask / share hook): ask rocm-smi --showtopo
        else:
            try:
                mine["links"] = {"to_rank_%d" % r: curvis_amd.Context.device_link(device_index, devs[r]) for r in range(world) if r != rank}
            except Exception as exc:  # noqa: BLE001
                mine["links"] = {"failed": short(exc)}
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)
        tt = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        agg = torch.tensor([float(steps_executed), float(rays), kernel_ms, shade_ms], dtype=torch.float64)
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
            # the clock the timed region ran at = the kernel's shader cycles per launch (live SQ pass: GRBM_GUI_ACTIVE / 8 XCDs,
            # a property of the instruction stream) / the kernel time of THIS region (HIP events); boxes differ by 2-6 % in
            # the clock they hold under this load.  The sysfs level (read once before and once after the region) lags a
            # 0.2 s region and is kept only as a cross-reference; the 10 s run below samples it properly.
            "effective_sclk_mhz": None,
            "sclk_mhz_sysfs_before_after": timed_clock["sclk_mhz_sysfs_before_after"],
            "power_w_before_after": timed_clock["power_w_before_after"],
            "value_with_download": with_download,
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
        if with_download is not None:
            with_download["delta_ms_per_step"] = round(with_download["ms_per_step"] - out["ms_per_step"], 3)
            with_download["fraction_of_value"] = round(with_download["value"] / out["value"], 4)
            if "value" in with_download.get("overlapped", {}):
                with_download["overlapped"]["fraction_of_value"] = round(with_download["overlapped"]["value"] / out["value"], 4)
        live_sq = traffic_note.get("live", {}).get("sq") if isinstance(traffic_note, dict) and isinstance(traffic_note.get("live"), dict) else None
        if live_sq and kernel_s > 0:
            eff = live_sq["shader_cycles_per_launch"] / kernel_s / 1e6
            out["effective_sclk_mhz"] = round(eff, 1)
            out["roofline"]["effective_sclk_mhz"] = round(eff, 1)
            # the FP64 peak AT THAT CLOCK (78.6 TF is 256 CU x 4 SIMD x 16 lanes x 2 flop at 2400 MHz)
            out["roofline"]["peak_at_effective_clock"] = round(FP64_VECTOR_PEAK_TFLOPS * eff / 2400.0, 2)
            out["roofline"]["frac_at_effective_clock"] = round(achieved_tflops / (FP64_VECTOR_PEAK_TFLOPS * eff / 2400.0), 4)
            out["roofline"]["effective_clock_note"] = (
                "shader cycles per launch of the live SQ pass (%d, GRBM_GUI_ACTIVE / 8) / this region's average kernel time (%.4f ms); "
                "the profiled child's own dispatches ran at %s MHz (same cycles / their own duration: a 4-launch process is "
                "still ramping its clock)" % (live_sq["shader_cycles_per_launch"], kernel_s * 1e3, live_sq.get("sclk_mhz_profiled_dispatches")))
        if sustained is not None:
            out["value_sustained"] = sustained
        if comm_info is not None:
            out["collective"] = comm_info
        if rows_split is not None:
            out["value_single_image_rows"] = rows_split
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
        # the reference's DEFAULT renderer (what its CLI runs): secondary figure at N = 1, outside the contract's timed region
        if dist is None and args.metric == "ellis" and not args.no_value_efficient:
            try:
                out["value_efficient"] = efficient_kernels(ctx, torch, args)
            except Exception as exc:  # noqa: BLE001 -- an extra must never cost the bench line
                out["value_efficient"] = {"failed": short(exc)}
            phase("value_efficient_kernels")
        # the reference's CPU path beside EVERY line (north_star: "timed on the node's own host cores in the same run"):
        # rank 0 runs it after the timed region while the other ranks wait at the barrier below
        if not args.no_cpu_baseline:
            phase("reductions_and_live_pmc_passes")
            out["cpu_baseline"] = cpu_baseline(args, host_skies)
            phase("cpu_baseline")

    # The JSON line must be the LAST thing on the job's stdout: libraries (RCCL's banner) write to the C stdout, which is
    # block-buffered when redirected and would otherwise be flushed at exit, after the line.  So: tear everything down,
    # flush the C streams on every rank, meet once more, and only then rank 0 prints.
    if not WEDGED["here"]:
        ctx.close()
        sky_dev.clear()
    # one visible device per rank (HIP_VISIBLE_DEVICES set by the launcher): rank 0's child would see ONE GPU and
    # `curvis video --devices N` could only fail -- the binary gets the launcher's restriction lifted instead
    if dist is not None and not args.no_video_e2e and not WEDGED["any"] and (world > 1 or os.environ.get("CURVIS_BENCH_FORCE_DIST") == "1"):
        # the product's own multi-GPU design -- ONE process, N device threads, ncclCommInitAll + curvis_ctx_bcast_skies, shared
        # retry queue, page-locked batch buffers, PNG writer pool -- end to end on a 16N-frame rendition of configs[3]; every
        # rank has released its context, the others wait at the barrier while rank 0 runs the binary over all N GPUs
        torch.cuda.synchronize()
        dist.barrier()
        if rank == 0:
            try:
                out["video_e2e"] = video_e2e(args, world, host_skies, os.environ.get("CURVIS_BENCH_SHARE_DEVICE") == "1")
            except Exception as exc:  # noqa: BLE001 -- an extra must never cost the bench line
                out["video_e2e"] = {"failed": short(exc)}
            try:  # the reference's DEFAULT renderer on the same node: ~0.3 ms of GPU per frame, where the host used to be the limit
                out["video_e2e_efficient"] = video_e2e(args, world, host_skies, os.environ.get("CURVIS_BENCH_SHARE_DEVICE") == "1",
                                                       mode="efficient", frames_per_gpu=32 * max(1, args.video_e2e_frames_per_gpu))
            except Exception as exc:  # noqa: BLE001
                out["video_e2e_efficient"] = {"failed": short(exc)}
            phase("video_e2e_both_modes")
    if dist is None and rank == 0 and isinstance(out.get("value_efficient"), dict) and "failed" not in out["value_efficient"]:
        # ... and end to end through the binary: `curvis video --mode efficient` on the reference's path_orbit.csv, files in -> PNG
        # frames out (the context above is closed: the binary makes its own, two per GPU for a video of this length)
        try:
            n_eff = args.value_efficient_frames
            if n_eff <= 0:
                n_eff = 3840
                try:  # 15 360 frames are 9.5 GB of PNG files in /dev/shm, which counts against the container's memory limit
                    stv = os.statvfs("/dev/shm")
                    if stv.f_bavail * stv.f_frsize > (32 << 30) and cgroup_memory_limit() > (64 << 30):
                        n_eff = 15360
                except OSError:
                    pass
            e2e = video_e2e(args, 1, host_skies, False, mode="efficient", frames_per_gpu=n_eff)
            if "failed" not in e2e:
                e2e["frames_requested"] = n_eff
                e2e["frames_per_s_workers"] = round(sum(float(dv.get("frames_per_s", 0.0)) for dv in e2e["per_device"]), 1)
                e2e["frames_per_s_workers_note"] = ("sum over the binary's workers of frames / the worker's own busy time: the run without its start-up "
                                                    "(context creation, sky decode and upload: ~0.1 s of this ~1 s run); `frames_per_s` is the whole run")
                e2e["frames_note"] = ("times_of_frames stops at t < 60 s of a path whose last row is short of 60 s, so a few frames fewer than "
                                      "requested are rendered, as with the reference (src/rendering.rs:224-238)")
                dev = e2e["per_device"][0]
                kern_s = e2e["frames"] * (dev["kernel_ms_per_frame"] + dev["gpu_png_kernel_ms_per_frame"]) / 1e3
                e2e["gpu_idle_share"] = round(max(0.0, 1.0 - kern_s / e2e["wall_s"]), 4) if e2e["wall_s"] > 0 else None
                e2e["gpu_idle_share_note"] = ("1 - (render + PNG kernels' HIP-event time, summed over the binary's contexts) / the binary's "
                                              "wall time for the frames; contexts overlap on the device, so this is a lower bound of the idle share")
            out["value_efficient"]["end_to_end"] = e2e
        except Exception as exc:  # noqa: BLE001
            out["value_efficient"]["end_to_end"] = {"failed": short(exc)}
        phase("value_efficient_end_to_end")
    if dist is not None:
        dist.barrier()
        flush_c_stdio()
        dist.barrier()
        dist.destroy_process_group()
    flush_c_stdio()
    sys.stdout.flush()
    os.dup2(job_stdout_fd, 1)
    os.close(job_stdout_fd)
    if rank == 0:
        phase("teardown")
        out["phase_seconds"] = phases
        sys.stdout.write(json.dumps(out) + "\n")
        sys.stdout.flush()
    if WEDGED["any"]:
        flush_c_stdio()
        os._exit(0)  # a thread stuck inside ncclCommInitRank would keep the interpreter from exiting


def agree(dist, world, ok, why=None):
    """every rank contributes (ok, reason) over the control plane; all get (everybody ok?, the first reason why not)"""
    flags = [None] * world
    dist.all_gather_object(flags, (bool(ok), why))
    bad = ["rank %d: %s" % (r, w) for r, (o, w) in enumerate(flags) if not o]
    return not bad, (bad[0] if bad else None)


def short(exc):
    t = str(exc).splitlines()
    return "%s: %s" % (type(exc).__name__, t[0][:200] if t else "")


FAILURES = []  # failure records of the product's data plane, the same list on every rank (gather_failures)
WEDGED = {"any": False, "here": False}  # set by product_comm / distribute_skies: some rank's RCCL call never returned

# First contact between two devices must be self-diagnosing: every step of the data plane has a name,
# and a failure is reported as {rank, stage, error, RCCL's own warnings} in `collective.failure_detail`.
STAGES = ("bootstrap (ncclGetUniqueId, id over the control plane)", "ncclCommInitRank", "upload on the root",
          "header_broadcast", "texture_broadcast(+l sky)", "texture_broadcast(-l sky)", "read-back", "render")


def rccl_log_path(rank):
    return os.environ.get("NCCL_DEBUG_FILE") or os.path.join("/tmp", "curvis_bench_rccl_rank%d_%d.log" % (rank, os.getpid()))


def rccl_log_begin(rank):
    """RCCL's warnings of THIS rank into a file of its own (NCCL_DEBUG=WARN unless the user asked for more), so that a
    failure can quote them; must run before the first RCCL call of the process"""
    os.environ.setdefault("NCCL_DEBUG", "WARN")
    if "NCCL_DEBUG_FILE" not in os.environ:
        os.environ["NCCL_DEBUG_FILE"] = rccl_log_path(rank)


def rccl_log_tail(rank, limit=1500):
    try:
        with open(rccl_log_path(rank), "r", errors="replace") as f:
            t = f.read()
        return t[-limit:].strip() or None
    except OSError:
        return None


def stage_of(message, default):
    """the stage a library error names ("sky broadcast, stage header_broadcast: ...") or `default`"""
    import re
    m = re.search(r"stage ([a-z_]+(?:\([^)]*\))?)", message or "")
    return m.group(1) if m else default


def with_time_limit(fn, seconds):
    """fn() on a helper thread: (result, None, False) | (None, reason, wedged) -- a collective whose peer never arrives
    must not hold the rank for ever"""
    import threading
    res = {}

    def run():
        try:
            res["value"] = fn()
        except Exception as exc:  # noqa: BLE001
            res["why"] = short(exc)
    th = threading.Thread(target=run, daemon=True)
    th.start()
    th.join(seconds)
    if th.is_alive():
        return None, "did not return within %g s" % seconds, True
    if "why" in res:
        return None, res["why"], False
    return res.get("value"), None, False


def gather_failures(dist, world, rank, mine):
    """every rank contributes its own failure record (or None); all get the list of the records that exist"""
    if mine is not None:
        mine = dict(mine, rank=rank, rccl_log=rccl_log_tail(rank))
    table = [None] * world
    dist.all_gather_object(table, mine)
    return [t for t in table if t]


def product_comm(ctx, dist, world, rank, timeout_s=600.0):
    """An RCCL communicator over the ranks' contexts, made by the product's entry points: rank 0 draws the ncclUniqueId
    (curvis_rccl_unique_id), it travels over the control plane, every rank joins with curvis_ctx_rccl_comm_init.  The
    join runs on a helper thread with a time limit (ncclCommInitRank waits for all ranks: a rank that failed early would
    leave the others inside it for ever; the limit is generous because RCCL's start-up alone was seen to take 3 s on one
    host and > 100 s on another), and the ranks agree on the outcome.  (comm, None) on every rank or
    (None, reason) on every rank."""
    import threading
    import curvis_amd
    uid, why = [None], None
    if rank == 0:
        try:
            if os.environ.get("CURVIS_BENCH_TEST_RCCL_FAIL") == "id":  # test hook
                raise RuntimeError("injected RCCL failure (CURVIS_BENCH_TEST_RCCL_FAIL=id)")
            uid[0] = curvis_amd.Context.rccl_unique_id()
        except Exception as exc:  # noqa: BLE001
            why = short(exc)
    dist.broadcast_object_list(uid, src=0)
    if uid[0] is None:
        FAILURES.extend(gather_failures(dist, world, rank, {"stage": STAGES[0], "error": why} if rank == 0 else None))
        return None, agree(dist, world, rank != 0, why)[1]
    res = {}

    def join():
        try:
            # test hook: the LAST rank never reaches ncclCommInitRank -- the others must get out by the time limit / error
            if os.environ.get("CURVIS_BENCH_TEST_RCCL_FAIL") == "1" and rank == world - 1:
                raise RuntimeError("injected RCCL failure (CURVIS_BENCH_TEST_RCCL_FAIL)")
            res["comm"] = ctx.rccl_comm_init(uid[0], world, rank)
        except Exception as exc:  # noqa: BLE001
            res["why"] = short(exc)
    th = threading.Thread(target=join, daemon=True)
    th.start()
    th.join(float(os.environ.get("CURVIS_BENCH_RCCL_INIT_TIMEOUT", timeout_s)))
    mine_ok = "comm" in res
    wedged = th.is_alive()   # the helper thread is still inside ncclCommInitRank, holding the context
    ok, why = agree(dist, world, mine_ok, res.get("why") or ("ncclCommInitRank did not return within the time limit" if wedged else None))
    # a rank whose join never came back must not open another RCCL communicator on the same GPU (torch's nccl group) nor
    # destroy the context under the wedged thread: every rank learns of it and the run goes straight to the host-staged
    # broadcast, leaks the context and leaves through os._exit once the line is out
    flags = [None] * world
    dist.all_gather_object(flags, bool(wedged))
    WEDGED["any"] = any(flags)
    WEDGED["here"] = bool(wedged)
    FAILURES.extend(gather_failures(dist, world, rank, None if mine_ok else {
        "stage": STAGES[1], "error": res.get("why") or "ncclCommInitRank did not return within the time limit"}))
    if ok:
        return res["comm"], None
    if mine_ok and world == 1:  # with peers missing, destroying a half-connected communicator may block: leave it
        curvis_amd.Context.rccl_comm_destroy(res["comm"])
    return None, why


def distribute_skies(ctx, dist, torch, world, rank, host_skies, sw, sh, try_rccl, keep_alive):
    """Both sky textures from rank 0 into every rank's context.  First choice: the PRODUCT's path (curvis_ctx_bcast_skies =
    ncclBroadcast on the context's stream, textures allocated on the receiving ranks from the broadcast shapes).  If any
    rank cannot take it, ALL ranks fall back together -- to torch's nccl broadcast into borrowed tensors, then to a
    host-staged gloo broadcast -- and the line says so.  Every rank then reads head, middle and tail of both textures
    back from ITS context and compares them with the closed form rank 0 generated from."""
    import datetime
    import curvis_amd
    from curvis_amd import skies
    nbytes = sw * sh * 4
    one = torch.ones(1, dtype=torch.float64)
    dist.all_reduce(one)
    info = {"control_plane": "gloo (torch.distributed on the launcher's store)", "ranks": dist.get_world_size(),
            "allreduce_of_ones": int(one.item()), "sky_bytes_each": nbytes}
    if info["allreduce_of_ones"] != world:
        raise SystemExit("bench.py: the control plane spans %d ranks, not %d" % (info["allreduce_of_ones"], world))
    fell = []
    done = False
    fail3 = os.environ.get("CURVIS_BENCH_TEST_RCCL_FAIL") == "all"  # test hook: no RCCL at all

    def timed_ms(t0):
        t = torch.tensor([(time.perf_counter() - t0) * 1e3], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return round(float(t.item()), 3)

    if fail3:
        fell.append("injected RCCL failure (CURVIS_BENCH_TEST_RCCL_FAIL=all): no RCCL on this node")
    if try_rccl and not fail3:
        comm, why = product_comm(ctx, dist, world, rank)
        if comm is None:
            fell.append("curvis_ctx_rccl_comm_init: " + str(why))
        else:
            ok, why, t0 = True, None, time.perf_counter()
            try:
                if rank == 0:
                    for which in range(2):
                        ctx.set_sky(which, curvis_amd.SphericalImage(host_skies[which]))
                torch.cuda.synchronize()
            except Exception as exc:  # noqa: BLE001
                ok, why = False, "upload on rank 0: " + short(exc)
            # every rank issues the SAME sequence of control-plane collectives whatever happened on rank 0 (a barrier
            # inside the try was skipped by a rank that raised, and the others waited in it for ever): agree, barrier, broadcast
            # only if the upload stood, timing reduction, agree
            up_ok, _ = agree(dist, world, ok, why)
            mine = None if ok else {"stage": STAGES[2], "error": why}
            dist.barrier()
            t0 = time.perf_counter()
            if up_ok:
                # header + 2 x ncclBroadcast on the context's stream, synchronised per stage inside; on a helper thread with a
                # time limit: a peer that never enters the collective must not hold this rank for ever
                _, bwhy, wedged = with_time_limit(lambda: ctx.bcast_skies(comm, 0),
                                                  float(os.environ.get("CURVIS_BENCH_RCCL_BCAST_TIMEOUT", "300")))
                if bwhy is not None:
                    ok, why = False, bwhy
                    mine = {"stage": stage_of(bwhy, "sky broadcast"), "error": bwhy}
                if wedged:
                    WEDGED["here"] = True
            ms = timed_ms(t0)  # (the upload's failure stays rank 0's own outcome: the agreement below reports it once)
            ok, why = agree(dist, world, ok, why)
            FAILURES.extend(gather_failures(dist, world, rank, mine))
            flags = [None] * world
            dist.all_gather_object(flags, bool(WEDGED["here"]))
            WEDGED["any"] = WEDGED["any"] or any(flags)
            if not WEDGED["any"]:  # destroying a communicator a thread is still inside of may block
                try:
                    curvis_amd.Context.rccl_comm_destroy(comm)
                except Exception:  # noqa: BLE001
                    pass
            if ok:
                done = True
                info.update({"backend": "rccl (product ABI)",
                             "via": "curvis_ctx_bcast_skies on a communicator from curvis_ctx_rccl_comm_init "
                                    "(ncclUniqueId from curvis_rccl_unique_id, shipped over the control plane)",
                             "sky_broadcast_ms": ms, "sky_broadcast_gbps": round(2 * nbytes / (ms * 1e-3) / 1e9, 2) if ms > 0 else None})
            else:
                fell.append("curvis_ctx_bcast_skies: " + str(why))
    group = None
    if not done and try_rccl and not fail3 and WEDGED["any"]:
        fell.append("torch nccl group: not attempted (a rank is still inside ncclCommInitRank on its GPU)")
    if not done and try_rccl and not fail3 and not WEDGED["any"]:  # second choice: torch's RCCL
        ok, why = True, None
        try:
            group = dist.new_group(backend="nccl", timeout=datetime.timedelta(seconds=180))
            probe = torch.ones(1, dtype=torch.float64, device="cuda")
            dist.all_reduce(probe, group=group)
            torch.cuda.synchronize()
            if int(probe.item()) != world:
                ok, why = False, "all-reduce of ones over torch's nccl group gave %d" % int(probe.item())
        except Exception as exc:  # noqa: BLE001
            ok, why = False, short(exc)
        ok, why = agree(dist, world, ok, why)
        if not ok:
            fell.append("torch nccl group: " + str(why))
            group = None
    if not done:
        ms_all = 0.0
        for which in range(2):
            t = torch.empty((sh, sw, 4), dtype=torch.uint8, device="cuda")
            if rank == 0:
                t.copy_(torch.from_numpy(host_skies[which]))
            torch.cuda.synchronize()
            dist.barrier()
            t0 = time.perf_counter()
            if group is not None:
                dist.broadcast(t, src=0, group=group)
            else:  # last resort: through host memory over the control plane
                h = t.cpu()
                dist.broadcast(h, src=0)
                t.copy_(h)
            torch.cuda.synchronize()
            ms_all += timed_ms(t0)
            ctx.set_sky_device(which, t.data_ptr(), sw, sh, copy=False)
            keep_alive.append(t)
        info.update({"backend": "rccl (torch nccl)" if group is not None else "gloo",
                     "via": "torch.distributed.broadcast into borrowed device tensors (curvis_ctx_set_sky_device)" +
                            ("" if group is not None else ", staged through host memory"),
                     "sky_broadcast_ms": round(ms_all, 3),
                     "sky_broadcast_gbps": round(2 * nbytes / (ms_all * 1e-3) / 1e9, 2) if ms_all > 0 else None})
    if fell:
        info["fallback_from"] = fell
        sys.stderr.write("bench.py: rank %d: sky broadcast fell back (%s); continuing via %s\n" % (rank, "; ".join(fell), info["backend"]))
    elif not try_rccl:
        info["fallback_from"] = ["RCCL not attempted (CURVIS_BENCH_SHARE_DEVICE / CURVIS_BENCH_BACKEND test hook)"]
    # what sits in THIS rank's HBM now, against the closed form (head, middle, tail of both textures)
    ok, why = True, None
    piece = min(nbytes, 1 << 16)
    for which in range(2):
        want = skies.smooth(sw, sh, 128 if which == 0 else 32).reshape(-1)
        for off in (0, (nbytes - piece) // 2 // 4 * 4, nbytes - piece):
            got = ctx.read_sky(which, off, piece)
            if not np.array_equal(got, want[off:off + piece]):
                ok, why = False, "sky %d differs at byte offset %d" % (which, off)
    FAILURES.extend(gather_failures(dist, world, rank, None if ok else {"stage": STAGES[6], "error": why}))
    ok, why = agree(dist, world, ok, why)
    if FAILURES:
        info["failure_detail"] = list(FAILURES)
    if not ok:
        raise SystemExit("bench.py: a sky texture arrived corrupted (%s); failure detail: %s" % (why, json.dumps(FAILURES)))
    info["readback_verified_on_every_rank"] = True
    return info


class ClockSampler:
    """shader clock and board power of a context's GPU, read from sysfs (curvis_ctx_device_status) on a helper thread
    every `period` seconds between construction and stop()"""

    def __init__(self, ctx, period):
        import threading
        self.samples, self._stop, self._ctx, self._period = [], threading.Event(), ctx, period
        self._th = threading.Thread(target=self._run, daemon=True)
        self._th.start()

    def _run(self):
        while True:
            st = self._ctx.device_status()
            self.samples.append((st["sclk_mhz"], st["power_w"]))
            if self._stop.wait(self._period):
                break

    def stop(self):
        self._stop.set()
        self._th.join()
        sclk = sorted(v[0] for v in self.samples if v[0] > 0)
        powr = sorted(v[1] for v in self.samples if v[1] > 0)

        def med(v):
            return v[len(v) // 2] if v else None
        return {"sclk_mhz_median": med(sclk), "sclk_mhz_min": sclk[0] if sclk else None, "sclk_mhz_max": sclk[-1] if sclk else None,
                "power_w_median": med(powr), "power_w_max": powr[-1] if powr else None, "samples": len(self.samples)}


def rows_split_run(ctx, dist, torch, world, rank, metric, cam, args, R, DELTA, fence):
    """ONE frame of the workload, rows [rank*H/N, (rank+1)*H/N) on each rank (curvis_render_brute_rows), repeated
    args.steps times between barriers; time = max over ranks.  Rows near the image centre need more steps than rows at
    the edge, so the split is not perfectly balanced -- the per-rank times say by how much."""
    H = args.height
    r0, r1 = rank * H // world, (rank + 1) * H // world
    ctx.render_brute_rows(metric, cam, r0, r1 - r0, args.max_iter, R, DELTA, download=False)  # shape priming
    fence()
    t0 = time.perf_counter()
    n_steps, k_ms = 0, 0.0
    for _ in range(args.steps):
        _, st = ctx.render_brute_rows(metric, cam, r0, r1 - r0, args.max_iter, R, DELTA, download=False)
        n_steps += st.steps
        k_ms += st.integrate_ms
    torch.cuda.synchronize()
    own = time.perf_counter() - t0
    fence()
    dt = time.perf_counter() - t0
    tt = torch.tensor([dt], dtype=torch.float64)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    agg = torch.tensor([float(n_steps)], dtype=torch.float64)
    dist.all_reduce(agg, op=dist.ReduceOp.SUM)
    parts = [None] * world
    dist.all_gather_object(parts, {"rank": rank, "rows": [r0, r1], "ms_per_image": round(own / args.steps * 1e3, 4),
                                   "kernel_ms_avg": round(k_ms / args.steps, 4), "steps_per_image": n_steps // max(1, args.steps)})
    return {"value": round(float(agg.item()) / float(tt.item()) / 1e6, 1), "unit": "Mray-steps/s (executed), ONE image over all GPUs",
            "scaling": "strong", "images": args.steps, "ms_per_image": round(float(tt.item()) / args.steps * 1e3, 4),
            "per_rank": parts,
            "note": "one %dx%d image per step, split by rows over the %d ranks with curvis_render_brute_rows; the frame "
                    "stays in HBM (a host would gather the bands: %d bytes per image); not the contract's `value`" % (
                        args.width, args.height, world, args.width * args.height * 3)}


def efficient_kernels(ctx, torch, args):
    """value_efficient, kernels only: RelativisticSystem::render_image_efficient (src/systems.rs:333-527) -- the renderer the
    reference's CLI runs (src/rendering.rs:97-106, :299-307) -- with the CLI's sampler settings (sampling_initial_nums = 100 for
    BOTH alphas_num and max_iterations_sampling, threshold_1 for both thresholds: src/main.rs:91-110), ONE context, frames left in
    HBM, args.width x args.height:
      kernels_only                 the 240 poses of configs[3] (the reference's own path_orbit.csv at 4 fps), cap args.max_iter, 120
                                   frames per call -> the device-resident sampler (sampler_kernel; every pose has l = 3: ONE job per call)
      kernels_only_distinct_radii  the first 240 poses of configs[4] (path_through.csv at 24 fps: every frame its own l), Interstellar
                                   metric, cap 8192, 120 frames per call -> one sampler job per frame
      ..._prefetched               the same two with the next call's sampler started before the current call renders
                                   (curvis_ctx_prefetch_efficient: what `curvis video` does)
      host_paced_32                configs[3] again at 32 frames per call: the host-paced sampler (what round 5 measured)
    GPU-idle share from the context's own HIP events."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import refpaths
    import curvis_amd
    from curvis_amd import rendering

    def poses(csv, fps, n):
        it = rendering.Interpolator.from_file(refpaths.reference_path_file(csv))
        times = rendering.times_of_frames(it.min_time(), it.max_time(), fps)[:n]
        return [curvis_amd.Camera(tuple(it.camera_position(t)), tuple(it.camera_forward(t)), tuple(it.camera_up(t)), 15.0, 43.0,
                                  args.width, args.height) for t in times]

    def leg(metric, cams, cap, per_call, prefetch=False):
        sets = (cap, 100.0, 0.05, 100, 100, 1e-5, 1e-5)

        def run():
            kernel_ms = steps = 0
            paths, chains = set(), 0
            if prefetch:
                ctx.prefetch_efficient(metric, cams[:per_call], *sets)
            for k in range(0, len(cams), per_call):
                if prefetch and k + per_call < len(cams):  # the next call's sampler starts before this call renders
                    ctx.prefetch_efficient(metric, cams[k + per_call:k + 2 * per_call], *sets)
                _, st = ctx.render_efficient(metric, cams[k:k + per_call], *sets, download=False)
                kernel_ms += st.kernel_ms
                steps += st.steps
                paths.add(ctx.get_option("last_sampler_path"))
                chains = max(chains, ctx.get_option("last_sampling_chains") if 1 in paths else 0)
            return kernel_ms, steps, paths, chains
        run()                                   # allocations, first-launch checks
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        kernel_ms, steps, paths, chains = run()
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        n = len(cams)
        return {"value": round(n / wall, 1), "frames": n, "frames_per_call": per_call, "contexts": 1, "sampler_prefetched": bool(prefetch),
                "sampler": {0: "host-paced (cv_sampler.h, speculating launches)", 1: "device-resident (sampler_kernel)",
                            2: "device-resident, fell back to the host"}[max(paths)],
                "euler_chains_per_call": chains or None,
                "ms_per_frame": round(wall / n * 1e3, 4), "kernel_ms_per_frame": round(kernel_ms / n, 4),
                # with the prefetch the sampler's kernel runs BESIDE the previous call's per-pixel kernel: summed kernel time can exceed the wall time
                "gpu_idle_share": round(max(0.0, 1.0 - kernel_ms / 1e3 / wall), 4), "integrator_steps_per_frame": int(steps / n)}
    orbit = poses("path_orbit.csv", 4.0, 240)
    through = poses("path_through.csv", 24.0, 240)
    out = {"unit": "%dx%d frames/s, reference's default renderer (render_image_efficient)" % (args.width, args.height),
           "kernels_only": leg(curvis_amd.EllisMetric(1.0), orbit, args.max_iter, 120),
           "kernels_only_distinct_radii": leg(curvis_amd.InterstellarMetric(0.1, 1e-4, 1.0), through, 8192, 120),
           "kernels_only_prefetched": leg(curvis_amd.EllisMetric(1.0), orbit, args.max_iter, 120, prefetch=True),
           "kernels_only_distinct_radii_prefetched": leg(curvis_amd.InterstellarMetric(0.1, 1e-4, 1.0), through, 8192, 120, prefetch=True),
           "host_paced_32": leg(curvis_amd.EllisMetric(1.0), orbit, args.max_iter, 32)}
    out["kernels_only"]["note"] = ("configs[3] poses (Ellis, l = 3 in every frame: the frames of a call share ONE sampler job); one context, frames "
                                   "stay in HBM; gpu_idle_share = 1 - kernels' HIP-event time / wall")
    out["kernels_only_distinct_radii"]["note"] = "configs[4] poses (Interstellar, cap 8192): every frame its own camera radius, one sampler job per frame"
    return out


def cgroup_memory_limit():
    """bytes this container may use (cgroup v2 memory.max or v1 memory.limit_in_bytes); a very large number when unlimited or unknown"""
    for p in ("/sys/fs/cgroup/memory.max", "/sys/fs/cgroup/memory/memory.limit_in_bytes"):
        try:
            with open(p) as f:
                t = f.read().strip()
            return (1 << 62) if t == "max" else int(t)
        except (OSError, ValueError):
            continue
    return 1 << 62


def video_e2e(args, world, host_skies, share_device, mode="brute", frames_per_gpu=None):
    """`curvis video --mode brute --devices N --stats` on a 16N-frame rendition of configs[3] (Ellis, path_orbit.csv,
    1920x1080, cap 4096): files in, PNG frames out, the binary's own per-device table back."""
    import shutil
    import subprocess
    import tempfile
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import refpaths  # the reference's own path_orbit.csv, committed as a data fixture (tests/golden/paths)
    exe = os.path.join(ROOT, "curvis_amd", "bin", "curvis")
    base = None  # a RAM file system when it has room (a 1080p frame of the smooth skies is a 0.6 MB PNG), else the default temporary directory
    try:
        stv = os.statvfs("/dev/shm")
        if stv.f_bavail * stv.f_frsize > (8 << 30):
            base = "/dev/shm"
    except OSError:
        pass
    d = tempfile.mkdtemp(prefix="curvis_e2e_", dir=base)
    try:
        t_files = time.perf_counter()
        from curvis_amd import _abi
        for name, sky in (("pos.png", host_skies[0]), ("neg.png", host_skies[1])):  # alpha is 255 throughout: RGB8 decodes to the same RGBA8
            rgb = np.ascontiguousarray(sky[..., :3])
            _abi.check(_abi.lib().curvis_image_save_rgb8(os.path.join(d, name).encode(), rgb.ctypes.data, sky.shape[1], sky.shape[0]))
        n_frames = max(1, frames_per_gpu or args.video_e2e_frames_per_gpu) * world
        # path_orbit.csv runs for 60 s; times_of_frames pushes t = 0, 1/fps, ... while t < 60 (src/rendering.rs:224-238)
        fps = n_frames / 60.0
        with open(os.path.join(d, "vid.toml"), "w") as f:
            f.write('video_name = "e2e"\nframe_rate = %r\nfilepath_to_camera_path = "%s"\n' % (fps, refpaths.reference_path_file("path_orbit.csv")))
        with open(os.path.join(d, "sim.toml"), "w") as f:
            f.write("escape_radius = 100.0\nray_integration_max_itarations = %d\nray_integration_step = 0.05\n"
                    "sampling_initial_nums = 100\nsampling_max_iterations = 50\n"
                    "sampling_convergence_threshold_1 = 1e-5\nsampling_convergence_threshold_2 = 2e-5\n" % args.max_iter)
        with open(os.path.join(d, "cam.toml"), "w") as f:
            f.write("resolution_x = %d\nresolution_y = %d\ndiagonal = 43.0\nfocal_length = 15.0\n" % (args.width, args.height))
        t_files = time.perf_counter() - t_files
        cmd = [exe, "video", os.path.join(d, "pos.png"), os.path.join(d, "neg.png"), os.path.join(d, "out"),
               "-v", os.path.join(d, "vid.toml"), "-s", os.path.join(d, "sim.toml"), "-c", os.path.join(d, "cam.toml"),
               "--mode", mode, "--devices", str(world), "--stats", os.path.join(d, "st.jsonl")] + (["--batch", "4"] if mode == "brute" else [])  # efficient: the binary's defaults (two contexts per GPU and 128 frames per call for a long video)
        env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
        if world > 1 and not share_device:  # a per-rank device mask is the launcher's, not the binary's: it drives all N GPUs itself
            for k in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
                env.pop(k, None)
        if share_device:
            env["CURVIS_TEST_SHARE_DEVICE"] = "1"
        elif world == 1 and mode == "brute":
            env["CURVIS_FORCE_RCCL"] = "1"  # single-rank communicator: the broadcast entry point still runs (once: RCCL's start-up takes 3-100 s depending on the host)
        os.makedirs(os.path.join(d, "out"), exist_ok=True)
        t0 = time.perf_counter()
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
        wall = time.perf_counter() - t0
        if r.returncode not in (0, 101):  # 101 = the reference's own panic in the last segment of a camera path; the frames before it are written
            return {"failed": "curvis video exited with %d: %s" % (r.returncode, r.stderr.strip().splitlines()[-1][:300] if r.stderr.strip() else "")}
        with open(os.path.join(d, "st.jsonl.summary.json")) as f:
            summ = json.load(f)
        with open(os.path.join(d, "st.jsonl")) as f:
            recs = [json.loads(ln) for ln in f if ln.strip()]
        frames_on_disk = len([n for n in os.listdir(os.path.join(d, "out", "tmp")) if n.endswith(".png")])
        steps = sum(rc["steps"] for rc in recs)
        return {"command": "curvis video --mode %s --devices %d%s --stats (configs[3]: Ellis, path_orbit.csv at %.4g fps, %dx%d, cap %d)" % (
                    mode, world, " --batch 4" if mode == "brute" else "", fps, args.width, args.height, args.max_iter),
                "frames": summ["frames"], "frames_on_disk": frames_on_disk,
                "frames_per_s": round(summ["frames_per_s"], 2), "wall_s": round(summ["wall_s"], 3),
                "process_wall_s": round(wall, 3),
                "value": round(steps / summ["wall_s"] / 1e6, 1), "unit": "Mray-steps/s (executed), files in -> PNG frames out",
                "gpu_png": summ.get("gpu_png"), "workers": len(summ["devices"]),
                "sky_distribution": summ.get("sky_distribution"),
                "writer_drain_s": round(summ["writer_drain_s"], 3), "writers": summ["writers"],
                "per_device": summ["devices"], "encode": summ.get("encode"),
                "distinct_gpus": len(set(dv["pci_bus_id"] for dv in summ["devices"])),
                "input_files_s": round(t_files, 2),
                "note": "one process, one host thread + context per GPU (in --mode efficient one or two per GPU by the length of the video, each taking up to 128 consecutive frames per call), frames k mod workers, skies decoded once and broadcast from "
                        "device 0 (ncclCommInitAll + curvis_ctx_bcast_skies), PNG frames written by the writer pool; wall_s "
                        "includes context creation, sky decode/upload/broadcast and the first-launch check of the relay kernel"}
    finally:
        shutil.rmtree(d, ignore_errors=True)


def sustained_run(ctx, step, seconds, torch):
    """back-to-back single-frame launches for at least `seconds`; a sampler thread reads the shader clock and the board
    power from sysfs (curvis_ctx_device_status) every 100 ms meanwhile"""
    torch.cuda.synchronize()
    clock = ClockSampler(ctx, 0.1)
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
    ck = clock.stop()
    # the second half alone: what the chip holds once the boost budget of the first seconds is spent
    half = clock.samples[len(clock.samples) // 2:]
    late = sorted(v[0] for v in half if v[0] > 0)
    out = {"value": round(n_steps / dt / 1e6, 1), "unit": "Mray-steps/s (executed), this rank", "launches": n,
           "seconds": round(dt, 2), "ms_per_step": round(dt / n * 1e3, 4), "kernel_ms_avg": round(k_ms / n, 4)}
    out.update(ck)
    out["sclk_mhz_median_second_half"] = late[len(late) // 2] if late else None
    out["note"] = ("back-to-back single-frame launches after the contract's timed region; clock and power from sysfs "
                   "(pp_dpm_sclk, hwmon power1_average) every 100 ms; null where sysfs does not tell")
    return out


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
                "(64 rays x 40 B parked and reloaded per hand-over, ~5 300 hand-overs per 1080p frame, write-through / cache-bypassing) "
                "-- the price of the 3-5 % the hand-over takes off a single-frame launch; it is not re-reading of inputs" +
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
             "--sustained-seconds", "0", "--no-traffic", "--no-live-traffic", "--no-cpu-baseline", "--no-value-efficient"]
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
            durations = {}
            for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                with open(path) as f:
                    for r in csv.DictReader(f):
                        if kernel_name in r.get("Kernel_Name", "") and r.get("Counter_Name") in vals:
                            vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
                            try:
                                durations[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
                            except (KeyError, ValueError):
                                pass
            res = {}
            if durations:
                dv = sorted(durations.values())
                res["_dispatch_seconds"] = (dv[len(dv) // 2], len(dv))
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
                      "sclk_mhz_profiled_dispatches": (round(gui / sq["_dispatch_seconds"][0] / 1e6, 1)
                                                       if sq.get("_dispatch_seconds", (0,))[0] > 0 else None),
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
    # the LITERAL Euler step: every r(l) / r_squared(l) / r_derivative(l) call the reference makes per step (the tests run the
    # oracle with each evaluated once -- identical bits, fewer libm calls; a timed baseline must not)
    O.lib().cvo_set_metric_memo(0)
    try:
        return _cpu_baseline_timed(args, O, om, oc, sp, sn)
    finally:
        O.lib().cvo_set_metric_memo(1)


def _cpu_baseline_timed(args, O, om, oc, sp, sn):
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
        "step": "literal (oracle/curvis_oracle_impl.inc update: the reference's calls, one for one; cvo_set_metric_memo(0))",
        "sample": "every %dth row of the same %dx%d frame: %d rays, %d Euler steps, %.1f s" % (
            args.cpu_row_step, args.width, args.height, st.rays, st.steps, dt),
        "host_cpu": model,
        "host_logical_cpus": os.cpu_count(),
        "config1_full": c1,
        "all_cores": allc,
    }


if __name__ == "__main__":
    main()
