"""Upper bound of what reordering rays could buy: the incoherent and the
bounce set traced as they are, and sorted by (direction octant, Morton code of
the origin) on the host beforehand.  usage (GPU box): python tools/sort_probe.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracer_amd import Context, scenes  # noqa: E402

s = scenes.sponza_class()
ctx = Context(s, 64, 64)
dev = torch.device("cuda:0")
lo, hi = s.pos.min(0), s.pos.max(0)


def part(v):
    v = v.astype(np.uint64)
    v = (v | (v << 16)) & 0x030000FF
    v = (v | (v << 8)) & 0x0300F00F
    v = (v | (v << 4)) & 0x030C30C3
    v = (v | (v << 2)) & 0x09249249
    return v


def keys(r, bits):
    q = np.clip(((r[:3].T - lo) / (hi - lo) * (1 << bits)).astype(np.int64), 0, (1 << bits) - 1)
    m = (part(q[:, 0]) << 2) | (part(q[:, 1]) << 1) | part(q[:, 2])
    octant = ((r[3] < 0).astype(np.uint64) << 2) | ((r[4] < 0).astype(np.uint64) << 1) | (r[5] < 0).astype(np.uint64)
    return m, octant


def run(name, rays):
    n = rays.shape[1]
    d = torch.from_numpy(np.ascontiguousarray(rays)).to(dev)
    of = torch.empty((3, n), dtype=torch.float32, device=dev)
    oi = torch.empty((2, n), dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    for rep in (3, 20):
        ctx.trace_closest_dev(n, d.data_ptr(), of[0].data_ptr(), oi[0].data_ptr(), oi[1].data_ptr(), of[1].data_ptr(), of[2].data_ptr(), rep)
    ms = ctx.last_trace_ms()
    print(f"{name:40s} {ms:8.3f} ms  {n / ms / 1e3:8.1f} Mrays/s", flush=True)


n = 2073600
for label, rays in (("incoherent", scenes.incoherent_rays(s, n)), ("bounce", scenes.bounce_rays(s, 1920, 1080, n))):
    run(label, rays)
    for bits in (4, 6, 8):
        m, o = keys(rays, bits)
        run(f"{label} sorted octant|morton{3*bits}", rays[:, np.argsort((o << 40) | m, kind="stable")])
        run(f"{label} sorted morton{3*bits}|octant", rays[:, np.argsort((m << 3) | o, kind="stable")])
