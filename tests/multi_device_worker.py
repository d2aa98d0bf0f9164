"""One rank of a multi-process, one-process-per-GPU host over the C ABI, the way a Rust host without an RCCL binding
would do it: rank 0 draws the ncclUniqueId with curvis_rccl_unique_id and leaves it in a file, every
rank joins with curvis_ctx_rccl_comm_init on ITS device, rank 0 uploads the two skies, all call curvis_ctx_bcast_skies,
read head / middle / tail of both textures back from their own HBM, and render frames k = rank (mod world) of a small
orbit.  No torch, no launcher: plain processes started by tests/test_gpu_multi_device.py.

  python multi_device_worker.py <rank> <world> <device> <work dir>
"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)

import curvis_amd  # noqa: E402
from curvis_amd import skies  # noqa: E402

SKY = (1024, 512)
RES = (96, 54)
N_FRAMES = 6


def poses():
    """a small orbit at l = 3: (position, forward, up) per frame"""
    out = []
    for k in range(N_FRAMES):
        out.append(((0.0, 3.0, np.pi / 2, 2 * np.pi * k / N_FRAMES), (-1.0, 0.0, 0.0), (0.0, 0.0, 1.0)))
    return out


def main():
    """every step has a name; the rank's status file always says which one it reached and, if one failed, why -- with RCCL's
    own warnings (NCCL_DEBUG=WARN into a file per rank) -- so that the first run on two real devices that breaks says WHERE:
    bootstrap / ncclCommInitRank / upload / header or texture broadcast (named by the library's error) / read-back /
    render on this device."""
    import json
    import traceback
    rank, world, device, work = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    log = os.path.join(work, "rccl_rank_%d.log" % rank)
    os.environ.setdefault("NCCL_DEBUG", "WARN")
    os.environ.setdefault("NCCL_DEBUG_FILE", log)
    status = {"rank": rank, "device": device, "stage": "context", "ok": False}

    def save():
        with open(os.path.join(work, "status_%d.json.part" % rank), "w") as f:
            json.dump(status, f)
        os.replace(os.path.join(work, "status_%d.json.part" % rank), os.path.join(work, "status_%d.json" % rank))

    def stage(name):
        status["stage"] = name
        save()
    fail_at = os.environ.get("CURVIS_WORKER_FAIL_STAGE")    # test hook: raise on entering this stage

    def enter(name):
        stage(name)
        if fail_at == name:
            raise RuntimeError("injected failure at stage %s" % name)
    try:
        enter("context")
        ctx = curvis_amd.Context(device)
        status["pci"] = ctx.device_status()["pci_bus_id"]
        status["device_name"] = ctx.device_info()["name"]
        # how this rank's GPU is connected to the others' (xGMI / PCIe, hops): the first measured broadcast rate is read against it
        status["links"] = {}
        from curvis_amd import _abi
        if int(_abi.lib().curvis_device_count()) >= world:
            for other in range(world):
                if other != device:
                    status["links"]["to_device_%d" % other] = curvis_amd.Context.device_link(device, other)
        enter("bootstrap (unique id through a file)")
        id_file = os.path.join(work, "rccl_id.bin")
        if rank == 0:
            uid = curvis_amd.Context.rccl_unique_id()
            with open(id_file + ".part", "wb") as f:
                f.write(uid)
            os.rename(id_file + ".part", id_file)
        else:
            t0 = time.time()
            while not os.path.exists(id_file):
                if time.time() - t0 > 120:
                    raise RuntimeError("no unique id from rank 0 within 120 s")
                time.sleep(0.05)
            with open(id_file, "rb") as f:
                uid = f.read()
        enter("ncclCommInitRank")
        comm = ctx.rccl_comm_init(uid, world, rank)
        sp, sn = skies.checker(SKY[0], SKY[1], seed=0xC0FFEE), skies.checker(SKY[0], SKY[1], seed=0xBADC0DE)
        enter("upload on the root")
        if rank == 0:  # only the root holds the textures before the broadcast
            ctx.set_sky(0, curvis_amd.SphericalImage(sp))
            ctx.set_sky(1, curvis_amd.SphericalImage(sn))
        enter("sky broadcast (the library's error names header_broadcast / texture_broadcast)")
        t0 = time.perf_counter()
        ctx.bcast_skies(comm, 0)
        bcast_s = time.perf_counter() - t0
        status["bcast_s"] = bcast_s
        status["sky_broadcast_gbps"] = 2 * SKY[0] * SKY[1] * 4 / bcast_s / 1e9
        curvis_amd.Context.rccl_comm_destroy(comm)
        enter("read-back")
        nbytes = SKY[0] * SKY[1] * 4
        piece = 1 << 14
        readback_ok = True
        for which, want in ((0, sp.reshape(-1)), (1, sn.reshape(-1))):
            for off in (0, (nbytes - piece) // 2 // 4 * 4, nbytes - piece):
                readback_ok &= bool(np.array_equal(ctx.read_sky(which, off, piece), want[off:off + piece]))
        enter("render on device %d" % device)
        metric = curvis_amd.EllisMetric(1.0)
        frames, stats = {}, {}
        for k, (pos, fwd, up) in enumerate(poses()):
            if k % world != rank:
                continue
            cam = curvis_amd.Camera(pos, fwd, up, 15.0, 43.0, RES[0], RES[1])
            rgb, st = ctx.render_brute(metric, cam, 4096, 100.0, 0.05)
            frames["frame_%d" % k] = rgb
            stats["stats_%d" % k] = np.array([st.rays, st.steps, st.n_pos, st.n_neg, st.n_none, st.n_oob], dtype=np.int64)
        np.savez(os.path.join(work, "rank_%d.npz" % rank), readback_ok=readback_ok, bcast_s=bcast_s, pci=np.array(status["pci"]), **frames, **stats)
        ctx.close()
        status["ok"] = True
        stage("done")
    except BaseException as exc:  # noqa: BLE001 -- the status file must say what happened, whatever it was
        status["error"] = "%s: %s" % (type(exc).__name__, exc)
        status["traceback"] = traceback.format_exc()[-1500:]
        try:
            with open(log, "r", errors="replace") as f:
                status["rccl_log"] = f.read()[-2000:]
        except OSError:
            status["rccl_log"] = None
        save()
        raise SystemExit("rank %d failed at stage `%s`: %s" % (rank, status["stage"], status["error"]))


if __name__ == "__main__":
    main()
