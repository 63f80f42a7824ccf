"""Development probe: cost of a 1080p depth-8 pass by light kind (point + sky, point only, sky only) on the
Sponza-class scene.  usage (GPU box): python tools/shade_probe.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd import Context, scenes
s = scenes.sponza_class()
for name, lights in (("point+env", s.lights), ("point", [l for l in s.lights if l.type == scenes.LIGHT_POINT]), ("env", [l for l in s.lights if l.type == scenes.LIGHT_ENV])):
    sc = scenes.replace(s, lights=lights)
    ctx = Context(sc, 1920, 1080, max_depth=8, seed=0)
    ctx.set_camera(sc.camera); ctx.render(16); ctx.synchronize()
    ctx.render(32); ctx.synchronize()
    st = ctx.stats()
    print(name, "ms/pass", st["last_render_ms"] / 32, "rays/pass", (st["closest_rays"] + st["shadow_rays"]) / 48 / 1e6)
    ctx.close()
