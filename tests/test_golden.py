"""Golden vectors (tests/golden, written by tests/golden/make_golden.py from
the CPU oracle; the reference has none).  The CPU half checks that the oracle
and the scenes still reproduce them; the GPU half checks the HIP path against
the same files without running the oracle."""
import os

import numpy as np
import pytest

from oracle import binding as ob
from raytracer_amd import Context, scenes

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def test_oracle_reproduces_cornell_golden():
    g = np.load(os.path.join(G, "cornell_env_32.npz"))
    s = scenes.cornell_box(env=True)
    o = ob.OracleScene(s)
    t, id0, id1, u, v = o.intersect(ob.camera_rays(s.camera, 32, 32))
    assert np.array_equal(id1, g["id1"]) and np.array_equal(id0, g["id0"])
    assert np.array_equal(bits(t), bits(g["t"])) and np.array_equal(bits(u), bits(g["u"]))
    film, cnt = o.render(s.camera, 32, 32, 4, 0, 2, threads=3)  # thread count must not matter
    assert np.array_equal(bits(film), bits(g["film"]))
    assert cnt["closest"] == int(g["closest"]) and cnt["shadow"] == int(g["shadow"])


def test_oracle_reproduces_rng_golden():
    g = np.load(os.path.join(G, "rng.npz"))
    L = ob.lib()
    vals = np.array([L.orc_rng_u01(int(k[0]), int(k[1]), int(k[2]), int(k[3])) for k in g["keys"]], np.float32)
    assert np.array_equal(bits(vals), bits(g["vals"]))


def test_oracle_reproduces_atrium_golden():
    g = np.load(os.path.join(G, "atrium_d02.npz"))
    a = scenes.sponza_class(detail=0.2, env_size=(32, 16))
    assert a.n_triangles == int(g["n_tris"])
    o = ob.OracleScene(a)
    boxes, left, right, tri, model, leaf = o.bvh_nodes()
    assert np.array_equal(left, g["left"]) and np.array_equal(right, g["right"]) and np.array_equal(tri, g["tri"])
    t, id0, id1, u, v, c = o.intersect(ob.camera_rays(a.camera, 48, 27), counters=True)
    assert np.array_equal(id1, g["id1"]) and np.array_equal(bits(t), bits(g["t"]))
    assert c["n_inner"] == int(g["n_inner"]) and c["n_tri"] == int(g["n_tri"])


@pytest.mark.gpu
def test_gpu_reproduces_cornell_golden():
    g = np.load(os.path.join(G, "cornell_env_32.npz"))
    s = scenes.cornell_box(env=True)
    ctx = Context(s, 32, 32, max_depth=4, seed=0)
    t, id0, id1, u, v = ctx.trace_closest(scenes.camera_rays_numpy(s.camera, 32, 32))
    assert np.array_equal(id1, g["id1"]) and np.array_equal(id0, g["id0"])
    assert np.array_equal(bits(t), bits(g["t"]))
    assert np.array_equal(bits(u), bits(g["u"])) and np.array_equal(bits(v), bits(g["v"]))
    inc = scenes.incoherent_rays(s, 2048, seed=9)
    t, id0, id1, u, v = ctx.trace_closest(inc)
    assert np.array_equal(id1, g["inc_id1"]) and np.array_equal(bits(t), bits(g["inc_t"]))
    assert np.array_equal(ctx.trace_occluded(inc), g["inc_occ"])
    ctx.set_camera(s.camera)
    ctx.render(2)
    film = ctx.read_film()
    ref = g["film"]
    # tolerance: sin/cos/acos/atan2 differ by ulps between glibc and the device library
    err = np.abs(film[..., :3].astype(np.float64) - ref[..., :3])
    assert np.sqrt((err ** 2).mean()) / np.sqrt((ref[..., :3].astype(np.float64) ** 2).mean()) < 1e-3
    assert np.array_equal(film[..., 3], ref[..., 3])
    ctx.close()


@pytest.mark.gpu
def test_gpu_reproduces_atrium_golden():
    g = np.load(os.path.join(G, "atrium_d02.npz"))
    a = scenes.sponza_class(detail=0.2, env_size=(32, 16))
    ctx = Context(a, 48, 27)
    boxes, left, right, tri, model, leaf = ctx.get_bvh()
    assert np.array_equal(left, g["left"]) and np.array_equal(right, g["right"]) and np.array_equal(tri, g["tri"])
    t, id0, id1, u, v = ctx.trace_closest(scenes.camera_rays_numpy(a.camera, 48, 27))
    assert np.array_equal(id1, g["id1"]) and np.array_equal(id0, g["id0"])
    assert np.array_equal(bits(t), bits(g["t"])) and np.array_equal(bits(v), bits(g["v"]))
    ctx.close()


# ---- films in the oracle's "shared" math mode: reproduced bit for bit by both sides
SHARED_CASES = {
    "cornell_env_32": (lambda: scenes.cornell_box(env=True), 32, 32, 4, 0, 2, {}),
    "cornell_tex_30x26": (lambda: scenes.with_textures(scenes.cornell_box(env=True), uv_scale=3.0), 30, 26, 3, 5, 2, {"tile_size": 16}),
    "cornell_area_ggx_24": (lambda: scenes.cornell_area(glossy=scenes.GLOSSY_GGX), 24, 24, 3, 2, 2, {}),
    "atrium_d02_48x27": (lambda: scenes.sponza_class(detail=0.2, env_size=(32, 16)), 48, 27, 6, 3, 2, {}),
}


def test_oracle_reproduces_shared_math_films():
    g = np.load(os.path.join(G, "films_shared_math.npz"))
    ob.set_math("shared")
    try:
        for key, (mk, W, H, depth, seed, passes, kw) in SHARED_CASES.items():
            s = mk()
            film, _ = ob.OracleScene(s).render(s.camera, W, H, depth, seed, passes, threads=2, **kw)
            assert np.array_equal(bits(film), bits(g[key])), key
    finally:
        ob.set_math("libm")


@pytest.mark.gpu
def test_gpu_reproduces_shared_math_films():
    """No oracle at run time: the HIP tracer against the committed films, bit for bit."""
    g = np.load(os.path.join(G, "films_shared_math.npz"))
    for key, (mk, W, H, depth, seed, passes, kw) in SHARED_CASES.items():
        s = mk()
        ctx = Context(s, W, H, max_depth=depth, seed=seed, **kw)
        ctx.set_camera(s.camera)
        ctx.render(passes)
        assert np.array_equal(bits(ctx.read_film()), bits(g[key])), key
        ctx.close()
