"""Development timer: the three-set closest-hit launch and the any-hit launch on the 4-wide tree (library from RT_HIP_LIB)."""
import json, os, sys, zlib
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from raytracer_amd import Context, scenes
W, H = 1920, 1080
n = W * H
wide = "--binary" not in sys.argv
s = scenes.sponza_class()
ctx = Context(s, W, H, max_depth=8, seed=0, batch=1, wide_bvh=wide)
sets = [scenes.camera_rays_numpy(s.camera, W, H), scenes.incoherent_rays(s, n, seed=0x5EED), scenes.bounce_rays(s, W, H, n, seed=0x5EED, device=0)]
r = np.ascontiguousarray(np.concatenate(sets, axis=1))
m = r.shape[1]
d = torch.from_numpy(r).cuda()
of = torch.empty((3, m), dtype=torch.float32, device="cuda"); oi = torch.empty((2, m), dtype=torch.int32, device="cuda")
st = torch.empty(m, dtype=torch.int32, device="cuda")
a = (m, d.data_ptr(), of[0].data_ptr(), oi[0].data_ptr(), oi[1].data_ptr(), of[1].data_ptr(), of[2].data_ptr())
REPS = int(os.environ.get("REPS", "200"))   # the launch time settles after about 0.2 s of back-to-back launches
ctx.trace_closest_dev(*a, REPS)
ms = []
for _ in range(2):
    ctx.trace_closest_dev(*a, REPS); ms.append(ctx.last_trace_ms())
alone = []
for k in range(3):   # each set launched alone
    dk = torch.from_numpy(np.ascontiguousarray(sets[k])).cuda()
    ak = (n, dk.data_ptr(), of[0].data_ptr(), oi[0].data_ptr(), oi[1].data_ptr(), of[1].data_ptr(), of[2].data_ptr())
    ctx.trace_closest_dev(*ak, REPS // 4); ctx.trace_closest_dev(*ak, REPS // 2); alone.append(round(ctx.last_trace_ms(), 4))
ctx.trace_closest_dev(*a, 10)
ctx.trace_occluded_dev(m, d.data_ptr(), st.data_ptr(), REPS // 2)
ctx.trace_occluded_dev(m, d.data_ptr(), st.data_ptr(), REPS // 2)
print(json.dumps({"lib": os.path.basename(os.environ.get("RT_HIP_LIB", "in-tree")), "wide": wide, "closest_ms": round(float(np.mean(ms)), 4), "any_ms": round(ctx.last_trace_ms(), 4), "alone_ms": alone,
                  "crc_id": zlib.crc32(oi[1].cpu().numpy().tobytes()), "crc_t": zlib.crc32(of[0].cpu().numpy().tobytes()), "crc_any": zlib.crc32(st.cpu().numpy().tobytes())}))
