"""Development A/B of the binary and the 4-wide closest-hit / any-hit launches on the benchmark's three 1080p ray
sets: times of both, and how their results differ (ids, distance bits, occlusion flags).  Prints one JSON line.
usage: python tools/wide_ab.py [--scene sponza|bistro] [--reps N]"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from raytracer_amd import Context, scenes  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--scene", default="sponza")
ap.add_argument("--reps", type=int, default=10)
ap.add_argument("--tag", default="")
ap.add_argument("--skip-binary", action="store_true")
a = ap.parse_args()
W, H = 1920, 1080
n = W * H
s = scenes.sponza_class() if a.scene == "sponza" else scenes.bistro_class()
ctxs = {"wide": Context(s, W, H, max_depth=8, seed=0, batch=1, wide_bvh=True)}
if not a.skip_binary:
    ctxs["binary"] = Context(s, W, H, max_depth=8, seed=0, batch=1)
nodes, depth = ctxs["wide"].get_wide_bvh()
slots = nodes[:, 28].view(np.int32)
out = {"tag": a.tag, "scene": a.scene, "triangles": s.n_triangles, "wide_nodes": int(nodes.shape[0]), "wide_depth": depth,
       "slots_hist": np.bincount(slots, minlength=5).tolist(), "build_ms": {k: c.stats()["bvh_build_ms"] for k, c in ctxs.items()}}
sets = {"coherent": scenes.camera_rays_numpy(s.camera, W, H), "incoherent": scenes.incoherent_rays(s, n, seed=0x5EED),
        "bounce": scenes.bounce_rays(s, W, H, n, seed=0x5EED, device=0)}
sets["all"] = np.ascontiguousarray(np.concatenate([sets[k] for k in ("coherent", "incoherent", "bounce")], axis=1))
res = {}
for k in ("all", "coherent", "incoherent", "bounce"):
    r = sets[k]
    m = r.shape[1]
    d = torch.from_numpy(r).cuda()
    for name, ctx in ctxs.items():
        of = torch.empty((3, m), dtype=torch.float32, device="cuda")
        oi = torch.empty((2, m), dtype=torch.int32, device="cuda")
        st = torch.empty(m, dtype=torch.int32, device="cuda")
        args = (m, d.data_ptr(), of[0].data_ptr(), oi[0].data_ptr(), oi[1].data_ptr(), of[1].data_ptr(), of[2].data_ptr())
        ctx.trace_closest_dev(*args, 3)
        ms = []
        for _ in range(3):
            ctx.trace_closest_dev(*args, a.reps)
            ms.append(ctx.last_trace_ms())
        out[f"{k}_{name}_ms"] = round(float(np.mean(ms)), 4)
        ctx.trace_occluded_dev(m, d.data_ptr(), st.data_ptr(), 2)
        ctx.trace_occluded_dev(m, d.data_ptr(), st.data_ptr(), a.reps)
        out[f"{k}_{name}_any_ms"] = round(ctx.last_trace_ms(), 4)
        res[(k, name)] = (of.cpu().numpy(), oi.cpu().numpy(), st.cpu().numpy())
    if "binary" in ctxs:
        (bf, bi, bs), (wf, wi, ws) = res[(k, "binary")], res[(k, "wide")]
        idm = bi[1] != wi[1]
        tm = bf[0].view(np.uint32) != wf[0].view(np.uint32)
        out[f"{k}_diff"] = {"id": int(idm.sum()), "t_bits": int(tm.sum()), "id_with_equal_t": int((idm & ~tm).sum()),
                            "any": int((bs != ws).sum()),
                            "max_rel_t": float(np.max(np.abs(bf[0][tm] - wf[0][tm]) / np.abs(bf[0][tm]))) if tm.any() else 0.0,
                            "hit_miss_flips": int(((bi[1] < 0) != (wi[1] < 0)).sum())}
print(json.dumps(out))
