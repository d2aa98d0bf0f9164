"""diagnostics: how often does the corrupt-hand-over hook of the relay kernel change the frame?"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import common, curvis_amd
sp, sn = common.make_skies(2048, 1024, "smooth")
_, _, pm, pc = common.scene("ellis", res=(480, 270))
ctx = curvis_amd.Context(0)
ctx.set_sky(0, curvis_amd.SphericalImage(sp)); ctx.set_sky(1, curvis_amd.SphericalImage(sn))
ctx.set_option("variant", 2); ctx.set_option("relay_min_blocks", 0); ctx.set_option("relay_auto_verify", 0)
for seg in (64, 0):
    ctx.set_option("relay_segment", seg)
    want, _ = ctx.render_brute(pm, pc, 4096, 100.0, 0.05)
    for k in range(12):
        ctx.set_option("relay_test_corrupt", 1)
        bad, _ = ctx.render_brute(pm, pc, 4096, 100.0, 0.05)
        print("seg %d run %d: differs %s (%d pixels), launches %d parks %d waiters %d disabled %d fallbacks %d" % (
            seg, k, not np.array_equal(bad, want), int((bad != want).any(axis=2).sum()), ctx.get_option("last_relay_launches"),
            ctx.get_option("last_relay_parks"), ctx.get_option("last_relay_waiters"), ctx.get_option("relay_disabled"), ctx.get_option("relay_fallbacks")), flush=True)
ctx.close()
