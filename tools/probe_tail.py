"""Development probe (needs a -DRT_PROBE_TIME build): where the time of one wide closest-hit launch goes -- from the
first wave's start to the last claim of work, and from there to the last wave's end (the tail).
usage: RT_HIP_LIB=<probe build> python tools/probe_tail.py [set,set,...]"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from raytracer_amd import Context, scenes, api
W, H = 1920, 1080
n = W * H
s = scenes.sponza_class()
ctx = Context(s, W, H, max_depth=8, seed=0, batch=1, wide_bvh=True)
sets = {"coherent": scenes.camera_rays_numpy(s.camera, W, H), "incoherent": scenes.incoherent_rays(s, n, seed=0x5EED),
        "bounce": scenes.bounce_rays(s, W, H, n, seed=0x5EED, device=0)}
order = sys.argv[1].split(",") if len(sys.argv) > 1 else ["coherent", "incoherent", "bounce"]
sets["all"] = np.ascontiguousarray(np.concatenate([sets[k] for k in order], axis=1))
L = api.load_library()
NW = 8192
L.rt_debug_tprobe.argtypes = [C.c_void_p, C.c_int]
for k in order + ["all"]:
    r = sets[k]
    m = r.shape[1]
    d = torch.from_numpy(r).cuda()
    of = torch.empty((3, m), dtype=torch.float32, device="cuda"); oi = torch.empty((2, m), dtype=torch.int32, device="cuda")
    a = (m, d.data_ptr(), of[0].data_ptr(), oi[0].data_ptr(), oi[1].data_ptr(), of[1].data_ptr(), of[2].data_ptr())
    ctx.trace_closest_dev(*a, 10)          # the stamps kept are those of the last of the ten launches
    ms = ctx.last_trace_ms()
    rec = np.zeros((NW, 4), dtype=np.uint64)
    assert L.rt_debug_tprobe(rec.ctypes.data, NW) == 0
    t = rec.astype(np.float64) / 100.0     # us
    t0 = t[:, 0].min()
    start, claim, dry, end = (t[:, i] - t0 for i in range(4))
    last_claim = claim.max()
    print(f"{k:10s} event mean {ms*1e3:7.1f} us | launch {end.max():7.1f} us; waves start within {start.max():5.1f} us; last claim of work at {last_claim:7.1f} us "
          f"-> tail {end.max()-last_claim:6.1f} us; waves end: 10% {np.percentile(end,10):7.1f} median {np.median(end):7.1f} 90% {np.percentile(end,90):7.1f} "
          f"99% {np.percentile(end,99):7.1f}; mean idle after own end {np.mean(end.max()-end):6.1f} us")
