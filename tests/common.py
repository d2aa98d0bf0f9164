"""Shared scene builders for the parity tests (same inputs to the oracle, the host twin and the GPU)."""
import ctypes as C
import os

import numpy as np

import oracle_lib as O
import curvis_amd
from curvis_amd import _abi, skies

HALF_PI = np.pi / 2
DEFAULTS = dict(max_radius=100.0, delta=0.05)  # settings/defaults/simulation_settings.toml


def make_skies(w=512, h=256, kind="check"):
    if kind == "check":
        return skies.checker(w, h, seed=0xC0FFEE), skies.checker(w, h, seed=0xBADC0DE)
    return skies.smooth(w, h, 128), skies.smooth(w, h, 32)


def scene(metric="ellis", res=(64, 36), pos=(0.0, 5.0, HALF_PI, 0.0), fwd=(-1.0, 0.0, 0.0), up=(0.0, 0.0, 1.0),
          focal=15.0, diag=43.0):
    """returns (oracle metric, oracle camera, product metric, product camera)"""
    if metric == "ellis":
        om, pm = O.ellis(1.0), curvis_amd.EllisMetric(1.0)
    elif metric == "interstellar":
        om, pm = O.interstellar(0.1, 1e-4, 1.0), curvis_amd.InterstellarMetric(0.1, 1e-4, 1.0)
    else:
        om, pm = O.flat(), curvis_amd.FlatSphericalMetric()
    oc = O.camera(pos, fwd, up, focal, diag, res)
    pc = curvis_amd.Camera(pos, fwd, up, focal, diag, res[0], res[1])
    return om, oc, pm, pc


_twin = None


def twin():
    global _twin
    if _twin is None:
        p = os.path.join(os.path.dirname(os.path.abspath(__file__)), "host_twin", "libtwin.so")
        L = C.CDLL(p)
        L.twin_math_array.restype = None
        L.twin_math_array.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
        L.twin_math3_array.restype = None
        L.twin_math3_array.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
        L.twin_render.restype = None
        L.twin_render.argtypes = [C.POINTER(_abi.Metric), C.POINTER(_abi.CameraC), C.c_void_p, C.c_uint, C.c_uint,
                                  C.c_void_p, C.c_uint, C.c_uint, C.c_uint, C.c_double, C.c_double, C.c_void_p,
                                  C.c_void_p, C.c_int]
        L.twin_render_efficient.restype = C.c_int
        L.twin_render_efficient.argtypes = [C.POINTER(_abi.Metric), C.POINTER(_abi.CameraC), C.c_void_p, C.c_uint,
                                            C.c_uint, C.c_void_p, C.c_uint, C.c_uint, C.c_uint, C.c_double,
                                            C.c_double, C.c_uint, C.c_uint, C.c_double, C.c_double, C.c_void_p,
                                            C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                                            C.POINTER(C.c_size_t), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64),
                                            C.c_int, C.c_int]
        _twin = L
    return _twin


def twin_math(op, a, b=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    out = np.empty_like(a)
    bb = np.ascontiguousarray(b, dtype=np.float64) if b is not None else None
    twin().twin_math_array(op, a.ctypes.data, bb.ctypes.data if bb is not None else None, out.ctypes.data, a.size)
    return out


def twin_math3(op, a, b=None, c=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    bb = np.ascontiguousarray(b, dtype=np.float64) if b is not None else None
    cc = np.ascontiguousarray(c, dtype=np.float64) if c is not None else None
    out = np.empty_like(a)
    twin().twin_math3_array(op, a.ctypes.data, bb.ctypes.data if bb is not None else None,
                            cc.ctypes.data if cc is not None else None, out.ctypes.data, a.size)
    return out


def twin_render(pm, pc, sky_pos, sky_neg, max_iter, max_radius, delta, fast=0):
    W, H = pc.resolution_width, pc.resolution_height
    rgb = np.zeros((H, W, 3), np.uint8)
    dbg = np.zeros((H, W), _abi.RAY_DEBUG)
    m = pm._c()
    twin().twin_render(C.byref(m), C.byref(pc._c), sky_pos.ctypes.data, sky_pos.shape[1], sky_pos.shape[0],
                       sky_neg.ctypes.data, sky_neg.shape[1], sky_neg.shape[0], max_iter, max_radius, delta,
                       rgb.ctypes.data, dbg.ctypes.data, int(fast))
    return rgb, dbg


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def assert_debug_equal(got, want, check_t=True):
    """bit-exact comparison of two RAY_DEBUG arrays (NaN payloads included)."""
    assert got.shape == want.shape
    for f in ("steps", "code", "tx", "ty"):
        bad = np.nonzero(got[f] != want[f])
        assert bad[0].size == 0, "%s differs at %d rays, first %s: %s vs %s" % (
            f, bad[0].size, [b[0] for b in bad], got[f][bad][0], want[f][bad][0])
    lo = 0 if check_t else 1
    for f in ("x", "p"):
        g, w = bits(got[f])[..., lo:], bits(want[f])[..., lo:]
        bad = np.nonzero(g != w)
        assert bad[0].size == 0, "%s differs at %d entries, first index %s: %r vs %r" % (
            f, bad[0].size, [b[0] for b in bad], got[f][..., lo:][bad][0], want[f][..., lo:][bad][0])


def twin_render_efficient(pm, pc, sky_pos, sky_neg, max_iter, max_radius, delta, alpha_nums, max_it_sampling, thr1, thr2,
                          fast=0, cap=8192, dev_sampler=0):
    """dev_sampler=1: the control flow of the device-resident sampler (cv_sampler_dev.h) instead of cv_sampler.h's"""
    W, H = pc.resolution_width, pc.resolution_height
    rgb = np.zeros((H, W, 3), np.uint8)
    a, e, s = np.zeros(cap), np.zeros(cap), np.zeros(cap)
    n, calls, steps = C.c_size_t(0), C.c_uint64(0), C.c_uint64(0)
    m = pm._c()
    rc = twin().twin_render_efficient(C.byref(m), C.byref(pc._c), sky_pos.ctypes.data, sky_pos.shape[1], sky_pos.shape[0],
                                      sky_neg.ctypes.data, sky_neg.shape[1], sky_neg.shape[0], max_iter, max_radius,
                                      delta, alpha_nums, max_it_sampling, thr1, thr2, rgb.ctypes.data, a.ctypes.data,
                                      e.ctypes.data, s.ctypes.data, cap, C.byref(n), C.byref(calls), C.byref(steps),
                                      int(fast), int(dev_sampler))
    if rc != 0:
        raise RuntimeError("twin efficient render failed: %d" % rc)
    k = n.value
    return rgb, dict(a=a[:k].copy(), e=e[:k].copy(), s=s[:k].copy(), calls=calls.value, steps=steps.value)


def device_count():
    """GPUs the product sees (curvis_device_count; not torch -- importing torch here would bring a second HIP runtime
    into the pytest process)"""
    return int(_abi.lib().curvis_device_count())


def share_env(n_devices, **extra):
    """environment for a `curvis ... --devices n` run: the devices themselves when the box has them, the
    CURVIS_TEST_SHARE_DEVICE hook (every worker on GPU 0) ONLY when it has fewer"""
    env = dict(os.environ, **extra)
    env.pop("CURVIS_TEST_SHARE_DEVICE", None)
    if device_count() < n_devices:
        env["CURVIS_TEST_SHARE_DEVICE"] = "1"
    return env


def host_cpus():
    """(cgroup CPU quota or None, CPUs in the affinity mask, logical CPUs of the machine)"""
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q = f.read().split()
        quota = None if q[0] == "max" else float(q[0]) / float(q[1])
    except (OSError, ValueError, IndexError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f1, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f2:
                q1, q2 = float(f1.read()), float(f2.read())
            quota = q1 / q2 if q1 > 0 else None
        except (OSError, ValueError):
            pass
    usable = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    return quota, usable, os.cpu_count() or 1


_reported = False


def host_threads(cap=128):
    """threads worth starting for the striped oracle renders: the container's CPU quota when the cgroup sets one (the
    GPU boxes show 256 logical CPUs behind a 16-CPU quota; more threads than that only add throttling), else the CPUs
    of the affinity mask.  The first call says what it found (pytest shows it with -s, and in the captured output of a
    failing test): the GPU tests' wall time is oracle time, i.e. proportional to 1 / this number."""
    global _reported
    quota, usable, logical = host_cpus()
    n = int(quota + 0.5) if quota else usable
    n = max(1, min(n, usable))
    if not _reported:
        _reported = True
        print("[host] cgroup CPU quota %s, %d CPUs in the affinity mask, %d logical CPUs -> %d oracle threads" % (
            "%.2f" % quota if quota else "none", usable, logical, n), flush=True)
    return max(1, min(cap, n))


# ---- oracle time: memo of repeated renders, and a budget that fails FAST instead of being killed by the driver's limit ----
_memo = {}
_memo_lock = None
_prefetch_thread = None


def _memo_slot(key):
    import threading
    global _memo_lock
    if _memo_lock is None:
        _memo_lock = threading.Lock()
    with _memo_lock:
        if key not in _memo:
            _memo[key] = {"lock": threading.Lock(), "done": False, "value": None}
        return _memo[key]


def oracle_memo(key, make):
    """session-wide memo of an oracle result several tests need (key = everything the result depends on: flavour, scene,
    size, cap, sky); the first caller pays -- or the background prefetcher already has (oracle_prefetch): a test that asks
    for a result being computed there waits for it instead of computing it twice"""
    slot = _memo_slot(key)
    with slot["lock"]:
        if not slot["done"]:
            slot["value"] = make()
            slot["done"] = True
    return slot["value"]


def prefetch_threads():
    """threads the background prefetcher may use: the quota minus six, so that the tests running meanwhile (the bench and CLI
    tests: subprocesses that mostly wait for the GPU and RCCL, but whose timing assertions must not be disturbed) keep CPUs and
    the cgroup stays under its quota -- CFS throttles EVERY thread of a cgroup that exceeds it"""
    return max(1, host_threads(64) - 6)


def oracle_prefetch(thunks):
    """Run `thunks` (callables that end in an oracle_memo call), one after the other, on ONE background thread, starting now.
    The GPU suite runs its files in alphabetical order: async download, bench and CLI tests come first and leave the host's
    CPUs idle for ~100 s, while the tests that need the heaviest oracle renders (full-size frames of the video configs, the 4K
    frame of configs[2]) come later and used to compute them while the GPU sat idle.  The renders start here instead; nothing
    about WHAT is computed or asserted changes.  A thunk that fails is dropped: the test then computes (and fails) itself."""
    import threading
    global _prefetch_thread
    if _prefetch_thread is not None or not thunks or os.environ.get("CURVIS_TEST_NO_PREFETCH"):
        return

    def run():
        for t in thunks:
            try:
                t()
            except Exception as exc:  # noqa: BLE001
                print("[oracle prefetch] dropped a job: %s" % exc, flush=True)
    _prefetch_thread = threading.Thread(target=run, name="oracle-prefetch", daemon=True)
    _prefetch_thread.start()


_rate = {}
_oracle_seconds = [0.0]   # projected oracle wall time of the heavy tests that have asked so far


def oracle_rate(kind, fl):
    """Euler steps per second of ONE oracle thread on this host (measured once per metric kind and flavour on a 64x36 frame)"""
    import time
    if (kind, fl) not in _rate:
        om, oc, _, _ = scene(kind, res=(64, 36))
        sp, sn = make_skies(64, 32, "check")
        t0 = time.perf_counter()
        _, _, st = O.render_image(fl, om, oc, O.sky(sp), O.sky(sn), 8192, 100.0, 0.05)
        _rate[(kind, fl)] = st.steps / max(time.perf_counter() - t0, 1e-6)
    return _rate[(kind, fl)]


def oracle_budget(what, kind, fl, n_steps, threads=None):
    """Called by a test BEFORE an oracle render of ~n_steps Euler steps: projects its wall time from the measured single-thread
    rate and the threads this lease really gives, prints it, and fails the test at once -- with the numbers -- when that one
    render would take longer than CURVIS_TEST_ORACLE_LIMIT_S (default 420 s) or the heavy renders of the session together
    longer than CURVIS_TEST_ORACLE_TOTAL_S (default 1000 s; the driver kills the GPU suite at 1200 s, and a green product must
    not read as `killed_at_limit` because the lease had a small CPU quota)."""
    import pytest
    T = threads or host_threads(64)
    rate = oracle_rate(kind, fl)
    projected = n_steps / (rate * T * 0.85)      # 0.85: what striping over T threads of a throttled cgroup delivers
    _oracle_seconds[0] += projected
    quota, usable, _ = host_cpus()
    line = "[oracle budget] %s: %.3g steps at %.1f Msteps/s/thread x %d threads (quota %s) -> ~%.0f s; heavy renders so far ~%.0f s" % (
        what, n_steps, rate / 1e6, T, "%.1f" % quota if quota else "none", projected, _oracle_seconds[0])
    print(line, flush=True)
    limit, total = float(os.environ.get("CURVIS_TEST_ORACLE_LIMIT_S", 420)), float(os.environ.get("CURVIS_TEST_ORACLE_TOTAL_S", 1000))
    if projected > limit or _oracle_seconds[0] > total:
        pytest.fail("oracle time over budget on this host, not a product failure -- " + line +
                    " (limits: %.0f s per render, %.0f s per session; raise CURVIS_TEST_ORACLE_LIMIT_S / _TOTAL_S or run on a "
                    "lease with more CPUs)" % (limit, total), pytrace=False)
    return projected


def oracle_full_frame(fl, om, oc, sky_pos, sky_neg, cap, threads=None):
    """whole-frame oracle render with the rows striped over host threads (ctypes drops the GIL)."""
    from concurrent.futures import ThreadPoolExecutor
    T = threads or host_threads(64)
    W, H = oc.res_x, oc.res_y
    rgb = np.zeros((H, W, 3), np.uint8)
    dbg = np.zeros((H, W), O.RAY_DEBUG)
    sp, sn = O.sky(sky_pos), O.sky(sky_neg)

    def work(i):
        r, d, st = O.render_image(fl, om, oc, sp, sn, cap, 100.0, 0.05, row_begin=i, row_step=T, debug=True)
        rgb[i::T] = r[i::T]
        dbg[i::T] = d[i::T]
        return st.steps
    with ThreadPoolExecutor(T) as ex:
        steps = sum(ex.map(work, range(T)))
    return rgb, dbg, steps


def oracle_full_frame_stats(fl, om, oc, sky_pos, sky_neg, cap, threads=None):
    """as oracle_full_frame, without the per-ray dump (88 B per ray and thread): returns (rgb, (rays, steps, n_pos,
    n_neg, n_none, n_oob)) -- what a 4K frame needs to stay inside a test's time budget."""
    from concurrent.futures import ThreadPoolExecutor
    T = threads or host_threads(64)
    W, H = oc.res_x, oc.res_y
    rgb = np.zeros((H, W, 3), np.uint8)
    sp, sn = O.sky(sky_pos), O.sky(sky_neg)

    def work(i):
        r, _, st = O.render_image(fl, om, oc, sp, sn, cap, 100.0, 0.05, row_begin=i, row_step=T)
        rgb[i::T] = r[i::T]
        return (st.rays, st.steps, st.n_pos, st.n_neg, st.n_none, st.n_oob)
    with ThreadPoolExecutor(T) as ex:
        parts = list(ex.map(work, range(T)))
    return rgb, tuple(int(sum(p[k] for p in parts)) for k in range(6))


def random_scene(rng, res=(16, 9)):
    """a random but valid scene: (oracle metric, oracle camera, product metric, product camera, delta, cap, R)"""
    kind = rng.choice(["ellis", "interstellar", "flat"], p=[0.5, 0.4, 0.1])
    if kind == "ellis":
        rho = float(10 ** rng.uniform(-1, 1))
        om, pm = O.ellis(rho), curvis_amd.EllisMetric(rho)
        scale = rho
    elif kind == "interstellar":
        m, a, rho = float(10 ** rng.uniform(-2, 0)), float(10 ** rng.uniform(-4, 0.3)), float(10 ** rng.uniform(-0.5, 0.7))
        om, pm = O.interstellar(m, a, rho), curvis_amd.InterstellarMetric(m, a, rho)
        scale = rho
    else:
        om, pm = O.flat(), curvis_amd.FlatSphericalMetric()
        scale = 1.0
    R = float(scale * rng.uniform(20, 120))
    l = float(rng.uniform(-0.5, 0.5) * R * 0.2)
    if kind == "flat":
        l = abs(l) + 0.5
    pos = (float(rng.uniform(-3, 3)), l, float(rng.uniform(0.15, 3.0)), float(rng.uniform(-7, 7)))
    fwd = tuple(rng.uniform(-1, 1, 3))
    up = tuple(rng.uniform(-1, 1, 3))
    focal, diag = float(rng.uniform(8, 60)), float(rng.uniform(15, 60))
    delta = float(scale * 10 ** rng.uniform(-2, -0.7))
    cap = int(rng.integers(200, 3000))
    oc = O.camera(pos, fwd, up, focal, diag, res)
    pc = curvis_amd.Camera(pos, fwd, up, focal, diag, res[0], res[1])
    return om, oc, pm, pc, delta, cap, R


def table_edge_inputs():
    """Arguments on and next to every row boundary of the cv_math.h tables: atan rows (u = j/128 +- 1/256 for the
    direct branch, u = j/256 +- 1/512 with x = -1/u for the reciprocal branch, the 0.4375 / 2 / 2^66 switches) and log slices
    (z = 1 + i/512, all exponents incl. subnormal scaling)."""
    import math
    at, lg = [], []
    for j in range(56, 257):          # direct branch: rows at j/128
        for d in (-1.0 / 256, 0.0, 1.0 / 256):
            for e in (-1, 0, 1):
                c = j / 128.0 + d
                c = math.nextafter(c, math.inf) if e > 0 else math.nextafter(c, -math.inf) if e < 0 else c
                if 0.4375 <= c < 2.0:
                    at += [c, -c]
    for j in range(-128, 1):          # reciprocal branch: rows at j/256, x = -1/u
        for d in (-1.0 / 512, 0.0, 1.0 / 512):
            for e in (-1, 0, 1):
                c = j / 256.0 + d
                c = math.nextafter(c, math.inf) if e > 0 else math.nextafter(c, -math.inf) if e < 0 else c
                if -0.5 <= c < 0.0:
                    at += [-1.0 / c, 1.0 / c]
    for x in (0.4375, 2.0, 2.0 ** 66, 128.0, 256.0):
        at += [x, math.nextafter(x, 0.0), math.nextafter(x, math.inf), -x]
    for i in range(513):
        z = 1.0 + i / 512.0
        for zz in (math.nextafter(z, 0.0), z, math.nextafter(z, math.inf)):
            for k in (-1070, -1022, -60, -1, 0, 1, 2, 10, 1023):
                try:
                    v = math.ldexp(zz, k)
                except OverflowError:
                    continue
                if 0.0 < v < math.inf:
                    lg.append(v)
    lg += [5e-324, 2.2250738585072014e-308, math.nextafter(2.2250738585072014e-308, 0.0), 1.7976931348623157e308]
    return np.array(at, dtype=np.float64), np.array(lg, dtype=np.float64)
