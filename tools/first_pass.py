"""Development timer: one pass right after a camera change (nothing cached, batch of 1) on both trees, and a cached
single pass.  usage: python tools/first_pass.py"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401
from raytracer_amd import Context, scenes
s = scenes.sponza_class()
cams = [s.camera, scenes.make_camera((2.0, 3.0, 1.0), (-1.0, -0.1, 0.3)), scenes.make_camera((-4.0, 5.0, -1.0), (1.0, -0.2, 0.1))]
for wide in (False, True):
    ctx = Context(s, 1920, 1080, max_depth=8, seed=0, wide_bvh=wide)
    ctx.set_camera(cams[0]); ctx.render(2); ctx.synchronize()
    first, cached = [], []
    for k in range(6):
        ctx.set_camera(cams[(k + 1) % 3]); ctx.synchronize()
        t0 = time.perf_counter(); ctx.render(1); ctx.synchronize(); first.append((time.perf_counter() - t0) * 1e3)
        t0 = time.perf_counter(); ctx.render(1); ctx.synchronize(); cached.append((time.perf_counter() - t0) * 1e3)
    print("wide" if wide else "binary", "first pass after a camera change %.2f ms, next single pass %.2f ms" % (np.median(first), np.median(cached)))
    ctx.close()
