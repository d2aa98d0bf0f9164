#!/usr/bin/env python3
"""render_image_efficient on ONE context: the host-paced sampler (speculating, several launches and host round trips per call)
against the device-resident one (sampler_kernel: one launch per call, a workgroup per distinct camera radius), frames per call
swept, on the poses of the two video configs (reference's CSVs): orbit = every frame at l = 3 (ONE job per call on the device),
fly-through = every frame its own l.  1920x1080, frames stay in HBM; optionally + the device PNG front end (deflate_frames).
python tools/gpu_eff_device_sampler.py [png]  -> profiles/round6_eff_device_sampler.txt"""
import os, sys, time
import numpy as np
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import refpaths
import curvis_amd
from curvis_amd import rendering, skies
PNG = len(sys.argv) > 1 and sys.argv[1] == "png"
W, H = 1920, 1080
ctx = curvis_amd.Context(0)
ctx.set_sky(0, curvis_amd.SphericalImage(skies.smooth(4096, 2048, 128))); ctx.set_sky(1, curvis_amd.SphericalImage(skies.smooth(4096, 2048, 32)))
print("# %s, %dx%d, one context, frames left in HBM%s" % (ctx.device_info()["name"], W, H, " + device PNG front end (streams to host)" if PNG else ""))
for video, metric, csv, fps, cap in (("orbit (Ellis, every frame l = 3)", curvis_amd.EllisMetric(1.0), "path_orbit.csv", 4.0, 4096),
                                     ("fly-through (Interstellar, every frame its own l)", curvis_amd.InterstellarMetric(0.1, 1e-4, 1.0), "path_through.csv", 24.0, 8192)):
    it = rendering.Interpolator.from_file(refpaths.reference_path_file(csv))
    times = rendering.times_of_frames(it.min_time(), it.max_time(), fps)
    cams = [curvis_amd.Camera(tuple(it.camera_position(t)), tuple(it.camera_forward(t)), tuple(it.camera_up(t)), 15.0, 43.0, W, H) for t in times]
    cams = cams[:480 - 480 % 32]
    print("\n## %s: %d frames" % (video, len(cams)))
    print("sampler  frames/call  frames/s  ms/frame  sampler kernel ms/frame  pixel kernel ms/frame  GPU idle share  launches/call")
    for per_call in (8, 32, 64, 128, 240):
        for dev in (0, 1):
            ctx.set_option("device_sampler", dev)
            best = None
            for rep in range(3):
                t0 = time.perf_counter(); samp = pix = png = 0.0; launches = 0; n = 0
                for k in range(0, len(cams), per_call):
                    part = cams[k:k + per_call]
                    _, st = ctx.render_efficient(metric, part, cap, 100.0, 0.05, 100, 100, 1e-5, 1e-5, download=False)
                    samp += st.integrate_ms; pix += st.shade_ms; launches += ctx.get_option("last_sampling_launches"); n += len(part)
                    if PNG:
                        _, ms = ctx.deflate_frames(W, H, len(part)); png += ms
                dt = time.perf_counter() - t0
                if best is None or dt < best[0]: best = (dt, samp, pix, png, launches)
            dt, samp, pix, png, launches = best
            calls = (len(cams) + per_call - 1) // per_call
            print("%-7s  %-11d  %-8.0f  %-8.4f  %-23.4f  %-21.4f  %-14.3f  %.1f%s" % ("device" if dev else "host", per_call, n / dt, dt / n * 1e3, samp / n, pix / n,
                  1 - (samp + pix + png) / 1e3 / dt, launches / calls, ("  (PNG kernels %.4f ms/frame)" % (png / n)) if PNG else ""), flush=True)
ctx.set_option("device_sampler", -1)
