#!/usr/bin/env python3
"""Measure the five BASELINE.json configurations on ONE GPU (configs 4-5 are video shards: the frames a
single rank of an 8-GPU run would render, i.e. frames k = 0 mod 8, plus the whole-video extrapolation).
Writes a markdown table to stdout; used for profiles/roundN_configs.md."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import refpaths  # noqa: E402  (the reference's own camera paths: tests/golden/paths)
import curvis_amd  # noqa: E402
from curvis_amd import paths, rendering, skies  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, default=8, help="simulated world size for the video shards")
    ap.add_argument("--variant", type=int, default=-1)
    ap.add_argument("--c5-frames", type=int, default=12, help="frames of the config-5 shard to render")
    args = ap.parse_args()
    ctx = curvis_amd.Context(0)
    ctx.set_option("variant", args.variant)
    sp, sn = skies.smooth(8192, 4096, 128), skies.smooth(8192, 4096, 32)
    ctx.set_sky(0, curvis_amd.SphericalImage(sp))
    ctx.set_sky(1, curvis_amd.SphericalImage(sn))
    ellis = curvis_amd.EllisMetric(1.0)
    inter = curvis_amd.InterstellarMetric(0.1, 1e-4, 1.0)
    pose = ((0.0, 5.0, np.pi / 2, 0.0), (-1.0, 0.0, 0.0), (0.0, 0.0, 1.0), 15.0, 43.0)
    rows = []

    def single(name, metric, res, cap, reps=5):
        cam = curvis_amd.Camera(*pose, res[0], res[1])
        ctx.render_brute(metric, cam, cap, 100.0, 0.05, download=False)
        t0 = time.perf_counter()
        for _ in range(reps):
            _, st = ctx.render_brute(metric, cam, cap, 100.0, 0.05, download=False)
        dt = (time.perf_counter() - t0) / reps
        rows.append((name, "%dx%d" % res, cap, 1, st.rays, st.steps, st.steps / st.rays, st.n_none, dt * 1e3,
                     st.steps / dt / 1e9, st.integrate_ms, st.shade_ms))

    single("config 1 (defaults, cap 40000)", ellis, (256, 144), 40000)
    single("config 2 (Ellis 1080p)", ellis, (1920, 1080), 4096)
    single("config 3 (Interstellar 4K)", inter, (3840, 2160), 8192, reps=3)

    def video(name, metric, csv, fps, res, cap, batch, max_frames=None):
        it = rendering.Interpolator.from_file(refpaths.reference_path_file(csv))
        v = rendering.VideoRenderingSystem(metric, ctx, it, fps, res, 43.0, 15.0, 100.0, cap, 0.05, rank=0,
                                           world_size=args.world, batch=batch, mode="brute")
        n_total = len(v.times_of_frames())
        mine = rendering.frames_of_rank(n_total, 0, args.world)
        if max_frames:
            # render only the first max_frames of the shard: shrink the world view accordingly
            v_times = v.times_of_frames()
            keep = mine[:max_frames]
            cams = [v.camera_at(v_times[k]) for k in keep]
            t0 = time.perf_counter()
            tot_steps = tot_rays = tot_none = 0
            integ = shade = 0.0
            for b0 in range(0, len(cams), batch):
                _, st = ctx.render_brute(metric, cams[b0:b0 + batch], cap, 100.0, 0.05, download=False)
                tot_steps += st.steps; tot_rays += st.rays; tot_none += st.n_none
                integ += st.integrate_ms; shade += st.shade_ms
            dt = time.perf_counter() - t0
            nfr = len(cams)
        else:
            t0 = time.perf_counter()
            stats = v.render(download=False)
            dt = time.perf_counter() - t0
            nfr = len(stats)
            tot_steps = sum(d["steps"] for d in stats); tot_rays = sum(d["rays"] for d in stats)
            tot_none = sum(d["n_none"] for d in stats)
            integ = sum(d["kernel_ms"] for d in stats); shade = 0.0
        rows.append(("%s: rank 0 of %d, %d of %d frames, batch %d" % (name, args.world, nfr, n_total, batch),
                     "%dx%d" % res, cap, nfr, tot_rays, tot_steps, tot_steps / max(tot_rays, 1), tot_none,
                     dt * 1e3 / nfr, tot_steps / dt / 1e9, integ / nfr, shade / nfr))

    video("config 4 (orbit video, Ellis)", ellis, "path_orbit.csv", 4.0, (1920, 1080), 4096, batch=6)
    video("config 5 (through video, Interstellar 4K)", inter, "path_through.csv", 24.0, (3840, 2160), 8192, batch=2,
          max_frames=args.c5_frames)

    print("| workload | resolution | cap | frames | rays | executed steps | steps/ray | capped rays | ms/frame (wall) | "
          "G ray-steps/s | integrate ms/frame | shade ms/frame |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|")
    for r in rows:
        print("| %s | %s | %d | %d | %d | %d | %.0f | %d | %.2f | %.1f | %.2f | %.2f |" % r)
    ctx.close()


if __name__ == "__main__":
    main()
