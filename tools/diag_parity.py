"""diagnostic: dump rays whose GPU state differs from the oracle (run on the GPU box)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import common, oracle_lib as O, curvis_amd

res = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (256, 144)
cap = int(sys.argv[3]) if len(sys.argv) > 3 else 40000
sp, sn = common.make_skies(512, 256, "smooth")
om, oc, pm, pc = common.scene("ellis", res=res)
ctx = curvis_amd.Context(0)
s = curvis_amd.RelativisticSystem(pm, curvis_amd.SphericalImage(sp), curvis_amd.SphericalImage(sn), pc, context=ctx)
rgb, dbg = s.render_image_debug(cap, 100.0, 0.05)
rgb2 = s.render_image(cap, 100.0, 0.05)
wrgb, wdbg, st = O.render_image(O.CV, om, oc, O.sky(sp), O.sky(sn), cap, 100.0, 0.05, debug=True)
print("debug rgb equal:", np.array_equal(rgb, wrgb), " nondebug rgb equal:", np.array_equal(rgb2, wrgb), " debug==nondebug:", np.array_equal(rgb, rgb2))
bad = np.argwhere((dbg["steps"] != wdbg["steps"]) | (dbg["code"] != wdbg["code"]) | (dbg["tx"] != wdbg["tx"]) | (dbg["ty"] != wdbg["ty"])
                  | (common.bits(dbg["x"])[..., 1:] != common.bits(wdbg["x"])[..., 1:]).any(-1) | (common.bits(dbg["p"]) != common.bits(wdbg["p"])).any(-1))
print("rays with differing state:", len(bad))
for (j, i) in bad[:12]:
    print((i, j), "gpu", dbg[j, i], "\n      cpu", wdbg[j, i])
badpx = np.argwhere((rgb2 != wrgb).any(-1))
print("nondebug differing pixels:", len(badpx), badpx[:20].tolist())
for (j, i) in badpx[:8]:
    print((i, j), rgb2[j, i], wrgb[j, i], "steps", wdbg[j, i]["steps"], "code", wdbg[j, i]["code"], "x", wdbg[j, i]["x"], "p", wdbg[j,i]["p"])
