#!/usr/bin/env python3
"""Generate the golden fixtures of tests/golden/ with the CPU oracle (both math flavours).

The reference (Rust) cannot be built or run in this environment and holds no golden vectors for this path,
so these fixtures pin the ORACLE (regression) and give the GPU tests committed expected outputs; the
oracle itself is pinned by the reference's own known answers and the survey's independent KAT values in
tests/test_oracle.py.  Inputs are procedural (curvis_amd.skies), so only outputs are stored.

    python tests/golden/make_golden.py        # rewrites tests/golden/*.npz
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import common  # noqa: E402
import oracle_lib as O  # noqa: E402

HALF_PI = np.pi / 2
# name -> (metric, resolution, camera position, forward, cap)
BRUTE = {
    "brute_ellis_default_64x36": ("ellis", (64, 36), (0.0, 5.0, HALF_PI, 0.0), (-1.0, 0.0, 0.0), 4096),
    "brute_interstellar_default_64x36": ("interstellar", (64, 36), (0.0, 5.0, HALF_PI, 0.0), (-1.0, 0.0, 0.0), 8192),
    "brute_ellis_orbit_l3_48x27": ("ellis", (48, 27), (0.0, 3.0, HALF_PI, 1.0), (-1.0, 0.0, 0.0), 4096),
}
EFFICIENT = {
    "efficient_ellis_default_96x54": ("ellis", (96, 54), (0.0, 5.0, HALF_PI, 0.0), (-1.0, 0.0, 0.0), 40000),
    "efficient_interstellar_default_64x36": ("interstellar", (64, 36), (0.0, 5.0, HALF_PI, 0.0), (-1.0, 0.0, 0.0), 40000),
}
SKY = (512, 256)


def main():
    sp, sn = common.make_skies(SKY[0], SKY[1], "check")
    for name, (metric, res, pos, fwd, cap) in BRUTE.items():
        om, oc, _, _ = common.scene(metric, res=res, pos=pos, fwd=fwd)
        out = {}
        for fl, tag in ((O.CV, "cv"), (O.LIBM, "libm")):
            rgb, dbg, st = O.render_image(fl, om, oc, O.sky(sp), O.sky(sn), cap, 100.0, 0.05, debug=True)
            out["rgb_" + tag] = rgb
            out["steps_" + tag] = dbg["steps"]
            out["code_" + tag] = dbg["code"].astype(np.int8)
            out["tx_" + tag] = dbg["tx"].astype(np.uint16)
            out["ty_" + tag] = dbg["ty"].astype(np.uint16)
            if tag == "cv":
                out["x_cv"] = dbg["x"]
                out["p_cv"] = dbg["p"]
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        print(name, "steps", int(out["steps_cv"].sum()))
    for name, (metric, res, pos, fwd, cap) in EFFICIENT.items():
        om, oc, _, _ = common.scene(metric, res=res, pos=pos, fwd=fwd)
        out = {}
        for fl, tag in ((O.CV, "cv"), (O.LIBM, "libm")):
            # the CLI's wiring: alphas_num = max_iterations_sampling = 100 (src/main.rs:46-47), thr 1e-5 / 1e-5
            rgb, smp, st = O.render_image_efficient(fl, om, oc, O.sky(sp), O.sky(sn), cap, 100.0, 0.05, 100, 100, 1e-5, 1e-5)
            out["rgb_" + tag] = rgb
            out["a_" + tag], out["e_" + tag], out["s_" + tag] = smp["a"], smp["e"], smp["s"]
            out["calls_" + tag] = np.array([smp["calls"], smp["steps"]], dtype=np.uint64)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        print(name, "samples", len(out["a_cv"]))


if __name__ == "__main__":
    main()
