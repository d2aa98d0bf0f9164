"""stability soak: repeated renders with changing resolutions / modes / variants, two contexts used from two
threads on the same GPU, device memory watched for growth."""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import curvis_amd
from curvis_amd import skies

def mem():
    free, total = torch.cuda.mem_get_info(0)
    return (total - free) / 2**20

sp, sn = skies.checker(2048, 1024, 1), skies.checker(2048, 1024, 2)
m_e, m_i = curvis_amd.EllisMetric(1.0), curvis_amd.InterstellarMetric(0.1, 1e-4, 1.0)
errors = []

def worker(tag, n):
    try:
        ctx = curvis_amd.Context(0)
        ctx.set_sky(0, curvis_amd.SphericalImage(sp)); ctx.set_sky(1, curvis_amd.SphericalImage(sn))
        ref = {}
        rng = np.random.default_rng(hash(tag) % 1000)
        for it in range(n):
            res = [(320, 180), (641, 359), (1280, 720), (96, 54)][it % 4]
            metric = m_e if it % 3 else m_i
            cam = curvis_amd.Camera((0, 4.0 + (it % 5), np.pi / 2, 0.1 * (it % 7)), (-1, 0, 0), (0, 0, 1), 15, 43, *res)
            ctx.set_option("variant", it % 3)
            ctx.set_option("relay_min_blocks", 0 if it % 2 else -1)
            ctx.set_option("fast_math", (it // 2) % 2)
            key = (res, it % 3 != 0, it % 5, it % 7)
            if it % 4 == 3:
                rgb, st = ctx.render_efficient(metric, cam, 4096, 100.0, 0.05, 100, 100, 1e-5, 1e-5)
                key = key + ("eff",)
            else:
                rgb, st = ctx.render_brute(metric, cam, 2500, 100.0, 0.05)
            h = hash(rgb.tobytes())
            if key in ref and ref[key] != h:
                errors.append("%s: non-deterministic frame for %r" % (tag, key))
            ref[key] = h
        ctx.close()
    except Exception as e:  # noqa
        errors.append("%s: %r" % (tag, e))

m0 = mem()
t0 = time.time()
ths = [threading.Thread(target=worker, args=("A", 140)), threading.Thread(target=worker, args=("B", 140))]
for t in ths: t.start()
for t in ths: t.join()
m1 = mem()
worker("C", 60)
m2 = mem()
print("soak done in %.1f s; device memory in use: start %.0f MiB, after two threaded contexts %.0f MiB, after third %.0f MiB" % (time.time() - t0, m0, m1, m2))
print("errors:", errors)
sys.exit(1 if errors or (m2 - m0) > 512 else 0)
