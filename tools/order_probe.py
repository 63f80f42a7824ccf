"""Development probe: the coherent (camera) ray set traced in raster order, in the reference's tile order (64x64 tiles,
2x2 quads inside, src/cpu/PathTracer.cpp:31-34) and in 8x8 pixel blocks inside tiles (the tracer's own slot order)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from raytracer_amd import Context, scenes  # noqa: E402

W, H = 1920, 1080
s = scenes.sponza_class()
ctx = Context(s, W, H, max_depth=8, seed=0, batch=1)
rays = scenes.camera_rays_numpy(s.camera, W, H)
ys, xs = np.mgrid[0:H, 0:W]
ys, xs = ys.ravel(), xs.ravel()
tile = (ys // 64) * (-(-W // 64)) + xs // 64
orders = {
    "raster": np.arange(W * H),
    "tile_quads": np.lexsort((xs & 1, ys & 1, (xs % 64) // 2, (ys % 64) // 2, tile)),
    "tile_8x8": np.lexsort((xs & 7, ys & 7, (xs % 64) // 8, (ys % 64) // 8, tile)),
}
for name, o in orders.items():
    r = np.ascontiguousarray(rays[:, o])
    m = r.shape[1]
    d = torch.from_numpy(r).cuda()
    of = torch.empty((3, m), dtype=torch.float32, device="cuda")
    oi = torch.empty((2, m), dtype=torch.int32, device="cuda")
    a = (m, d.data_ptr(), of[0].data_ptr(), oi[0].data_ptr(), oi[1].data_ptr(), of[1].data_ptr(), of[2].data_ptr())
    ctx.trace_closest_dev(*a, 3)
    t = []
    for _ in range(3):
        ctx.trace_closest_dev(*a, 20)
        t.append(ctx.last_trace_ms())
    print(name, [round(x, 4) for x in t])
