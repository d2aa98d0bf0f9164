#!/usr/bin/env python3
"""Deep parity fuzz of the EFFICIENT renderer (render_image_efficient, what `curvis image|video` run): N random
scenes (tests/common.random_scene) at 24x16 -- the adaptive sample table (alphas, escape angles, spaces), the sampler
bookkeeping (calls, steps) and every pixel against the oracle (cv flavour); scenes on which the reference panics
(fewer than three finite samples, undefined tangent rotation) must be reported as errors by the ABI too.
    python tools/gpu_eff_fuzz.py [scenes] [seed] [device]      "device": through the device-resident sampler (sampler_kernel) instead of
                                                               the host-paced one a single-frame call takes by default"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import common, oracle_lib as O, curvis_amd
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 11)
ctx = curvis_amd.Context(0)
DEVICE = len(sys.argv) > 3 and sys.argv[3] == "device"
ctx.set_option("device_sampler", 1 if DEVICE else 0)
paths = {0: 0, 1: 0, 2: 0}
sp, sn = common.make_skies(128, 64, "check")
ctx.set_sky(0, curvis_amd.SphericalImage(sp)); ctx.set_sky(1, curvis_amd.SphericalImage(sn))
bad = panics = done = 0; steps = 0; kinds = {}
t0 = time.time()
while done + panics < N:
    om, oc, pm, pc, delta, cap, R = common.random_scene(rng, res=(24, 16))
    if om.kind == O.FLAT:
        continue
    n0 = int(rng.integers(20, 80)); thr = float(10 ** rng.uniform(-5, -3))
    try:
        with np.errstate(all="ignore"):
            want_rgb, want, _ = O.render_image_efficient(O.CV, om, oc, O.sky(sp), O.sky(sn), cap, R, delta, n0, n0, thr, thr)
    except RuntimeError:
        try:
            ctx.render_efficient(pm, pc, cap, R, delta, n0, n0, thr, thr)
            bad += 1; print("MISSING PANIC", done + panics, flush=True)
        except curvis_amd.CurvisError:
            pass
        panics += 1
        continue
    rgb, st = ctx.render_efficient(pm, pc, cap, R, delta, n0, n0, thr, thr)
    paths[ctx.get_option("last_sampler_path")] += 1
    a, e, s = ctx.samples(0); info = ctx.sampling_info(0)
    ok = (np.array_equal(common.bits(a), common.bits(want["a"])) and np.array_equal(common.bits(e), common.bits(want["e"]))
          and np.array_equal(common.bits(s), common.bits(want["s"])) and np.array_equal(rgb, want_rgb)
          and (info.calls, info.steps) == (want["calls"], want["steps"]))
    if not ok:
        bad += 1; print("MISMATCH scene", done + panics, "kind", om.kind, flush=True)
    done += 1; steps += int(want["steps"]); kinds[om.kind] = kinds.get(om.kind, 0) + 1
print("efficient renderer, %s sampler (calls by path: host-paced %d, device %d, device -> host fall-back %d): scenes %d (by kind %s) + %d on which the reference panics "
      "(reported as errors), sampler Euler steps %d, mismatching scenes %d, %.0f s" % (
          "device-resident" if DEVICE else "host-paced", paths[0], paths[1], paths[2], done, kinds, panics, steps, bad, time.time() - t0))
sys.exit(1 if bad else 0)
