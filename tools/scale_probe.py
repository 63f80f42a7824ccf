"""How the closest-hit launch time scales with the number of rays: camera
rays at several resolutions and incoherent sets of several sizes on the
Sponza-class scene.  usage (GPU box): python tools/scale_probe.py [--wide]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracer_amd import Context, scenes  # noqa: E402

s = scenes.sponza_class()
ctx = Context(s, 64, 64, wide_bvh="--wide" in sys.argv)
dev = torch.device("cuda:0")


def run(name, rays):
    n = rays.shape[1]
    d = torch.from_numpy(rays).to(dev)
    of = torch.empty((3, n), dtype=torch.float32, device=dev)
    oi = torch.empty((2, n), dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    for rep in (3, 20):
        ctx.trace_closest_dev(n, d.data_ptr(), of[0].data_ptr(), oi[0].data_ptr(), oi[1].data_ptr(), of[1].data_ptr(), of[2].data_ptr(), rep)
    ms = ctx.last_trace_ms()
    print(f"{name:24s} n={n:9d}  {ms:8.3f} ms  {n / ms / 1e3:8.1f} Mrays/s", flush=True)


inc = scenes.incoherent_rays(s, 65536)
for n in (1, 64, 1024, 16384, 65536):   # the floor: what a launch costs however few rays it carries
    run("incoherent (small)", np.ascontiguousarray(inc[:, :n]))
for (W, H) in ((480, 270), (960, 540), (1920, 1080), (2880, 1620), (3840, 2160)):
    run(f"camera {W}x{H}", scenes.camera_rays_numpy(s.camera, W, H))
for n in (259200, 1036800, 2073600, 4147200, 8294400):
    run("incoherent", scenes.incoherent_rays(s, n))
