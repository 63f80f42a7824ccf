"""Renders a few passes of the C3 workload (Sponza-class, 1080p, depth 8) and
nothing else: the program to put behind `rocprofv3 --kernel-trace`.
usage: python tools/render_only.py [passes] [variant: plain|tex|n4|bistro] [batch]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracer_amd import Context, scenes  # noqa: E402

passes = int(sys.argv[1]) if len(sys.argv) > 1 else 4
variant = sys.argv[2] if len(sys.argv) > 2 else "plain"
s = scenes.sponza_class()
if variant == "tex":
    s = scenes.with_textures(s, uv_scale=8.0)
if variant == "n4":
    panel = scenes.area_quad((-6, 14.5, -3), (6, 14.5, -3), (6, 14.5, 3), (-6, 14.5, 3), (60.0, 55.0, 45.0))
    s = scenes.with_glossy(scenes.replace(s, lights=s.lights + panel), scenes.GLOSSY_GGX)
depth = 8
if variant == "bistro":   # the C5 workload: 3.16 M triangles, 64 point lights + sky, depth 12
    s = scenes.bistro_class()
    depth = 12
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 0
ctx = Context(s, 1920, 1080, max_depth=depth, seed=0, batch=batch)
ctx.set_camera(s.camera)
ctx.render(max(1, batch))
ctx.synchronize()
ctx.render(passes)
ctx.synchronize()
st = ctx.stats()
print("ms/pass", st["last_render_ms"] / passes, {k: st[k] for k in ("closest_rays", "shadow_rays", "exact_closest_rays", "exact_shadow_rays")})
ctx.close()
