"""Soak run of the exact-film check (pytest does not collect this file; it runs the CPU oracle as checker): seeded random
configurations -- scene, camera, lights (point, environment, area), microfacet lobe, textures, film size, tile size, depth,
batch, passes -- rendered by the HIP tracer on the reference's binary tree and compared bit for bit with the shared-math
oracle's film; the same configuration on the 4-wide tree is compared with the binary film (pixels may differ there only
through distance ties: the count is reported, not asserted).
usage (GPU box): python tests/soak_films.py [first_seed] [n_seeds]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import binding as ob  # noqa: E402
from oracle.binding import OracleScene  # noqa: E402
from raytracer_amd import Context, scenes  # noqa: E402


def _config(rng, base):
    name = ("cornell", "cornell_env", "atrium", "cornell_area")[rng.integers(0, 4)]
    s = base[name]
    lo, hi = s.pos.min(0), s.pos.max(0)
    if name != "cornell_area" and rng.random() < 0.6:       # a light set of its own
        lights = []
        for _ in range(int(rng.integers(0, 5))):
            p = rng.uniform(lo + 0.1 * (hi - lo), hi - 0.1 * (hi - lo)).astype(np.float32)
            lights.append(scenes.Light(scenes.LIGHT_POINT, p, rng.uniform(0.2, 3.0, 3).astype(np.float32)))
        if rng.random() < 0.6:
            lights.append(scenes.Light(scenes.LIGHT_ENV, intensity=rng.uniform(0.3, 2.0, 3).astype(np.float32),
                                       env=scenes.sky_env(int(rng.choice([16, 64, 100])), int(rng.choice([8, 32, 50])), seed=int(rng.integers(1, 99)))))
        if rng.random() < 0.4:
            c = 0.5 * (lo + hi)
            y = float(hi[1] - 0.02 * (hi[1] - lo[1]))
            e = 0.15 * (hi - lo)
            lights += scenes.area_quad((c[0] - e[0], y, c[2] - e[2]), (c[0] + e[0], y, c[2] - e[2]), (c[0] + e[0], y, c[2] + e[2]),
                                       (c[0] - e[0], y, c[2] + e[2]), tuple(rng.uniform(5, 40, 3)))
        if lights:
            s = scenes.replace(s, lights=lights)
    if rng.random() < 0.4:
        s = scenes.with_glossy(s, int(rng.choice([scenes.GLOSSY_GGX, scenes.GLOSSY_BECKMANN])), every=int(rng.integers(1, 4)))
    if rng.random() < 0.4:
        s = scenes.with_textures(s, uv_scale=float(rng.choice([1.0, 3.0, 8.0])), every=int(rng.integers(1, 3)))
    cam = s.camera
    if rng.random() < 0.5:      # a camera somewhere inside the scene looking at a random point of it
        p = rng.uniform(lo + 0.2 * (hi - lo), hi - 0.2 * (hi - lo))
        q = rng.uniform(lo, hi)
        f = q - p
        if np.linalg.norm(f) > 1e-3 and abs(f[1]) < 0.95 * np.linalg.norm(f):
            cam = scenes.make_camera(tuple(p), tuple(f))
    W, H = int(rng.integers(5, 300)), int(rng.integers(5, 200))
    return dict(name=name, scene=s, cam=cam, W=W, H=H, depth=int(rng.integers(1, 9)), seed=int(rng.integers(0, 1 << 30)),
                passes=int(rng.integers(1, 9)), tile=int(rng.choice([8, 16, 32, 64])), batch=int(rng.choice([0, 1, 2, 3, 8])))


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    n_seeds = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    threads = max(1, min(16, len(os.sched_getaffinity(0))))
    base = {"cornell": scenes.cornell_box(), "cornell_env": scenes.cornell_box(env=True),
            "atrium": scenes.sponza_class(detail=0.3, env_size=(128, 64)), "cornell_area": scenes.cornell_area()}
    ob.set_math("shared")
    t0 = time.perf_counter()
    wide_diff = 0
    for seed in range(first, first + n_seeds):
        c = _config(np.random.default_rng(seed), base)
        s = c["scene"]
        films = {}
        for wide in (False, True):
            ctx = Context(s, c["W"], c["H"], max_depth=c["depth"], seed=c["seed"], tile_size=c["tile"], batch=c["batch"], wide_bvh=wide)
            ctx.set_camera(c["cam"])
            ctx.render(c["passes"])
            films[wide] = ctx.read_film()
            ctx.close()
        ref, _ = OracleScene(s).render(c["cam"], c["W"], c["H"], c["depth"], c["seed"], c["passes"], threads=threads, tile_size=c["tile"])
        a, b, w = films[False].view(np.uint32), ref.view(np.uint32), films[True].view(np.uint32)
        same = (a == b) | (np.isnan(films[False]) & np.isnan(ref))
        desc = f"{c['name']} {c['W']}x{c['H']} depth {c['depth']} passes {c['passes']} tile {c['tile']} batch {c['batch']} lights {len(s.lights)}"
        assert same.all(), (seed, desc, f"{(~same).any(-1).sum()} pixels differ")
        dw = int(((a != w) & ~(np.isnan(films[False]) & np.isnan(films[True]))).any(-1).sum())
        wide_diff += dw
        print(f"seed {seed}: {desc}: exact; wide tree differs in {dw} pixels  [{time.perf_counter() - t0:.0f} s]", flush=True)
    print("soak ok:", {"seeds": n_seeds, "pixels_differing_on_the_wide_tree": wide_diff})


if __name__ == "__main__":
    main()
