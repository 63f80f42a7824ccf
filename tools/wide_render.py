"""Development check: the path tracer on the binary and on the 4-wide tree (rt_params.flags): ms per pass of both and
how many film pixels differ.  usage: python tools/wide_render.py [sponza|bistro|cornell] [passes]"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401
from raytracer_amd import Context, scenes
name = sys.argv[1] if len(sys.argv) > 1 else "sponza"
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 32
s = {"sponza": scenes.sponza_class, "bistro": scenes.bistro_class, "cornell": scenes.cornell_box}[name]()
W, H, D = (1920, 1080, 8) if name != "cornell" else (512, 512, 4)
if name == "bistro":
    D = 12
out = {"scene": name, "passes": passes}
films = {}
for kind in ("binary", "wide"):
    ctx = Context(s, W, H, max_depth=D, seed=0, wide_bvh=(kind == "wide"))
    ctx.set_camera(s.camera)
    ctx.render(passes)
    ctx.reset_accumulation()
    ctx.synchronize()
    t0 = time.perf_counter()
    ctx.render(passes)
    ctx.synchronize()
    out[kind + "_ms_per_pass"] = round((time.perf_counter() - t0) / passes * 1e3, 4)
    films[kind] = ctx.read_film()
    ctx.close()
a, b = films["binary"].view(np.uint32), films["wide"].view(np.uint32)
out["pixels_differing"] = int((a != b).any(-1).sum())
print(json.dumps(out))
