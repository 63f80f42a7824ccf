"""Development probe (needs a library built with -DRT_PROBE, RT_HIP_LIB=...): where the wave-steps of the
closest-hit launch go, before and after the work counter runs dry."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from raytracer_amd import Context, scenes, api  # noqa: E402

W, H = 1920, 1080
n = W * H
s = scenes.sponza_class()
WIDE = "--wide" in sys.argv
ctx = Context(s, W, H, max_depth=8, seed=0, wide_bvh=WIDE)
sets = {"coherent": scenes.camera_rays_numpy(s.camera, W, H), "incoherent": scenes.incoherent_rays(s, n, seed=0x5EED),
        "bounce": scenes.bounce_rays(s, W, H, n, seed=0x5EED, device=0)}
sets["all"] = np.ascontiguousarray(np.concatenate([sets[k] for k in ("coherent", "incoherent", "bounce")], axis=1))
L = api.load_library()
L.rt_debug_probe.argtypes = [C.POINTER(C.c_uint64), C.c_int]
buf = (C.c_uint64 * 16)()
for k, r in sets.items():
    m = r.shape[1]
    d = torch.from_numpy(r).cuda()
    of = torch.empty((3, m), dtype=torch.float32, device="cuda"); oi = torch.empty((2, m), dtype=torch.int32, device="cuda")
    ctx.trace_closest_dev(m, d.data_ptr(), of[0].data_ptr(), oi[0].data_ptr(), oi[1].data_ptr(), of[1].data_ptr(), of[2].data_ptr(), 1)
    ctx.synchronize()
    L.rt_debug_probe(buf, 1)
    ctx.trace_closest_dev(m, d.data_ptr(), of[0].data_ptr(), oi[0].data_ptr(), oi[1].data_ptr(), of[1].data_ptr(), of[2].data_ptr(), 1)
    ctx.synchronize()
    L.rt_debug_probe(buf, 1)
    v = [int(x) for x in buf]
    for ph, o in (("steady", 0), ("drain", 8)):
        refills, refill_lanes, inner, inner_lanes, leaf, leaf_lanes, deep_lanes, deep_steps = v[o:o + 8]
        print(f"{k:10s} {ph:6s} refills {refills/m:.4f}/ray ({refill_lanes/max(refills,1):.1f} idle lanes)  inner wave-steps {inner/m:.3f}/ray "
              f"({inner_lanes/max(inner,1):.1f} lanes, {deep_lanes/max(inner_lanes,1)*100:.0f}% of lane-steps below the LDS levels, "
              f"{deep_steps/max(inner,1)*100:.0f}% of wave-steps touch global)  leaf wave-steps {leaf/m:.3f}/ray ({leaf_lanes/max(leaf,1):.1f} lanes)")
