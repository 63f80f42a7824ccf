"""Development A/B timer for the closest-hit launch: the three 1080p ray sets of the benchmark in one launch and each
alone, on the library RT_HIP_LIB names (or the in-tree one).  Prints one JSON line with the times and a checksum of
the hit ids (equal checksums = same hits).  usage: python tools/ab_trace.py [--host-bvh] [--scene sponza|bistro]"""
import argparse
import json
import os
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from raytracer_amd import Context, scenes  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--host-bvh", action="store_true")
ap.add_argument("--scene", default="sponza")
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--tag", default="")
a = ap.parse_args()
W, H = 1920, 1080
n = W * H
s = scenes.sponza_class() if a.scene == "sponza" else scenes.bistro_class()
ctx = Context(s, W, H, max_depth=8, seed=0, host_bvh=a.host_bvh, batch=1)
sets = {"coherent": scenes.camera_rays_numpy(s.camera, W, H), "incoherent": scenes.incoherent_rays(s, n, seed=0x5EED),
        "bounce": scenes.bounce_rays(s, W, H, n, seed=0x5EED, device=0)}
sets["all"] = np.ascontiguousarray(np.concatenate([sets[k] for k in ("coherent", "incoherent", "bounce")], axis=1))
out = {"tag": a.tag, "lib": os.environ.get("RT_HIP_LIB", "in-tree"), "host_bvh": a.host_bvh, "scene": a.scene}
for k in ("all", "coherent", "incoherent", "bounce"):
    r = sets[k]
    m = r.shape[1]
    d = torch.from_numpy(r).cuda()
    of = torch.empty((3, m), dtype=torch.float32, device="cuda")
    oi = torch.empty((2, m), dtype=torch.int32, device="cuda")
    args = (m, d.data_ptr(), of[0].data_ptr(), oi[0].data_ptr(), oi[1].data_ptr(), of[1].data_ptr(), of[2].data_ptr())
    ctx.trace_closest_dev(*args, 3)
    best = 1e9
    tot = 0.0
    for _ in range(3):
        ctx.trace_closest_dev(*args, a.reps)
        ms = ctx.last_trace_ms()
        best = min(best, ms)
        tot += ms
    out[k + "_ms"] = round(tot / 3, 4)
    out[k + "_best_ms"] = round(best, 4)
    out[k + "_crc"] = zlib.crc32(oi[1].cpu().numpy().tobytes()) ^ zlib.crc32(of[0].cpu().numpy().tobytes())
out["grays_per_s_all"] = round(3 * n / out["all_ms"] / 1e6, 3)
print(json.dumps(out))
