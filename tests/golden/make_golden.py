"""Generates tests/golden/*.npz from the CPU oracle (the reference itself
holds no fixtures and cannot be built here).  The vectors freeze the oracle's
behaviour so that later edits to it, or to the scenes, are caught; they pin
neither side to the reference.  Run from the repository root:
    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import binding as ob  # noqa: E402
from raytracer_amd import scenes  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def textured():
    # 4. textured Cornell box with a sky, 30x26 film in 16-pixel tiles (ragged
    # tiles, quads with missing neighbours), depth 3, 2 passes
    s = scenes.with_textures(scenes.cornell_box(env=True), uv_scale=3.0)
    o = ob.OracleScene(s)
    film, cnt = o.render(s.camera, 30, 26, 3, 5, 2, threads=1, tile_size=16)
    lv = [o.texture_level(1, k) for k in range(o.texture_levels(1))]
    tri = np.array([o.texture_trilinear(2, 0.3, 0.7, (d, 0.0), (0.0, d * 0.5)) for d in (0.0, 0.004, 0.01, 0.1, 10.0)], np.float32)
    np.savez_compressed(os.path.join(HERE, "cornell_tex_30x26.npz"), film=film, closest=np.uint64(cnt["closest"]),
                        shadow=np.uint64(cnt["shadow"]), brick_l3=lv[3], brick_last=lv[-1], n_levels=np.int64(len(lv)), tri=tri)


def glossy_area():
    # 5. Cornell box under a ceiling panel (two area lights), microfacet lobe on every model
    out = {}
    for key, kind in (("ggx", scenes.GLOSSY_GGX), ("beckmann", scenes.GLOSSY_BECKMANN)):
        s = scenes.cornell_area(glossy=kind)
        film, cnt = ob.OracleScene(s).render(s.camera, 24, 24, 3, 2, 2, threads=1)
        out[key] = film
        out[key + "_closest"] = np.uint64(cnt["closest"])
        out[key + "_shadow"] = np.uint64(cnt["shadow"])
    np.savez_compressed(os.path.join(HERE, "cornell_area_ggx_24.npz"), **out)


def shared_films():
    # 6. films rendered with the oracle's transcendentals in "shared" mode (the polynomials the HIP kernels
    # restate): the GPU must reproduce these bit for bit
    ob.set_math("shared")
    try:
        out = {}
        s = scenes.cornell_box(env=True)
        out["cornell_env_32"], _ = ob.OracleScene(s).render(s.camera, 32, 32, 4, 0, 2, threads=1)
        s = scenes.with_textures(scenes.cornell_box(env=True), uv_scale=3.0)
        out["cornell_tex_30x26"], _ = ob.OracleScene(s).render(s.camera, 30, 26, 3, 5, 2, threads=1, tile_size=16)
        s = scenes.cornell_area(glossy=scenes.GLOSSY_GGX)
        out["cornell_area_ggx_24"], _ = ob.OracleScene(s).render(s.camera, 24, 24, 3, 2, 2, threads=1)
        a = scenes.sponza_class(detail=0.2, env_size=(32, 16))
        out["atrium_d02_48x27"], _ = ob.OracleScene(a).render(a.camera, 48, 27, 6, 3, 2, threads=1)
        np.savez_compressed(os.path.join(HERE, "films_shared_math.npz"), **out)
    finally:
        ob.set_math("libm")


def main():
    L = ob.lib()
    if "--only-shared-films" in sys.argv:
        shared_films()
        return
    if "--only-textured" in sys.argv:
        textured()
        return
    if "--only-glossy-area" in sys.argv:
        glossy_area()
        return
    # 1. Cornell box, 32x32: primary hits and a 2-pass film (seed 0, depth 4)
    s = scenes.cornell_box(env=True)
    o = ob.OracleScene(s)
    rays = ob.camera_rays(s.camera, 32, 32)
    t, id0, id1, u, v = o.intersect(rays)
    film, cnt = o.render(s.camera, 32, 32, 4, 0, 2, threads=1)
    inc = scenes.incoherent_rays(s, 2048, seed=9)
    ti, i0, i1, ui, vi = o.intersect(inc)
    occ = o.occluded(inc)
    np.savez_compressed(os.path.join(HERE, "cornell_env_32.npz"), t=t, id0=id0, id1=id1, u=u, v=v, film=film,
                        closest=np.uint64(cnt["closest"]), shadow=np.uint64(cnt["shadow"]),
                        inc_t=ti, inc_id0=i0, inc_id1=i1, inc_u=ui, inc_v=vi, inc_occ=occ)
    # 2. sampler values
    keys = [(seed, pix, ps, dim) for seed in (0, 7) for pix in (0, 1, 1000, 2073599) for ps in (0, 1, 1023) for dim in (0, 1, 65)]
    vals = np.array([L.orc_rng_u01(*k) for k in keys], np.float32)
    np.savez_compressed(os.path.join(HERE, "rng.npz"), keys=np.array(keys, np.uint64), vals=vals)
    # 3. small atrium: tree digest + hits
    a = scenes.sponza_class(detail=0.2, env_size=(32, 16))
    oa = ob.OracleScene(a)
    boxes, left, right, tri, model, leaf = oa.bvh_nodes()
    ra = ob.camera_rays(a.camera, 48, 27)
    ta, a0, a1, ua, va, c = oa.intersect(ra, counters=True)
    np.savez_compressed(os.path.join(HERE, "atrium_d02.npz"), n_tris=np.int64(a.n_triangles), left=left, right=right, tri=tri,
                        box_sum=boxes.astype(np.float64).sum(0), t=ta, id0=a0, id1=a1, u=ua, v=va,
                        n_inner=np.uint64(c["n_inner"]), n_tri=np.uint64(c["n_tri"]))
    textured()
    glossy_area()
    shared_films()
    print("golden vectors written to", HERE)


if __name__ == "__main__":
    main()
