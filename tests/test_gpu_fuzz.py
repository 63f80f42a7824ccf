"""Seeded fuzz parity on the GPU: random triangle soups (clustered, duplicated, degenerate, axis-aligned) and ray
sets salted with the awkward cases of src/scene/Ray.hpp:10-24 and src/scene/Ray.cpp:17-67 (zero and denormal
direction components, origins on box planes, huge and non-finite values).  Closest hit: ids, distances and
barycentrics bit-exact against the oracle; any hit: status equal; the tree equal node for node."""
import numpy as np
import pytest

from raytracer_amd import Context, scenes
from oracle.binding import OracleScene

pytestmark = pytest.mark.gpu


def _soup(rng, n_tri, kind):
    c = rng.uniform(-10, 10, (n_tri, 1, 3)).astype(np.float32)
    if kind == "clustered":      # many triangles in a few Morton cells: identical codes, deep equal-code splits
        c = np.round(c / 5.0) * 5.0
    e = rng.normal(0, 0.6, (n_tri, 3, 3)).astype(np.float32)
    if kind == "axis":           # axis-aligned slivers: flat boxes (zero extent on one axis)
        e[:, :, rng.integers(0, 3)] = 0.0
    tris = (c + e).astype(np.float32)
    if kind == "dup":            # coincident triangles: exact-distance ties
        tris[1::2] = tris[0::2][: tris[1::2].shape[0]]
    if kind == "degenerate":     # zero-area and point triangles among the ordinary ones
        tris[::7, 1] = tris[::7, 0]
        tris[::11] = tris[::11, :1]
    pos = tris.reshape(-1, 3)
    n = pos.shape[0]
    z = np.zeros((n, 3), np.float32)
    nrm = z.copy(); nrm[:, 2] = 1
    tan = z.copy(); tan[:, 0] = 1
    bn = z.copy(); bn[:, 1] = 1
    tm = (np.arange(n_tri) % 5).astype(np.int32)
    mats = np.tile(np.array([[0.5, 0.5, 0.5, 0.5]], np.float32), (5, 1))
    f3 = lambda *v: np.asarray(v, np.float32)
    cam = scenes.Camera(f3(0, 0, 30), f3(1, 0, 0), f3(0, 1, 0), f3(0, 0, -1))
    return scenes.Scene(pos, nrm, np.zeros((n, 2), np.float32), tan, bn, np.arange(n, dtype=np.uint32), tm, mats, [], cam, "fuzz")


def _rays(rng, scene, n):
    lo, hi = scene.pos.min(0), scene.pos.max(0)
    o = rng.uniform(lo - 2, hi + 2, (n, 3)).astype(np.float32)
    d = rng.normal(0, 1, (n, 3)).astype(np.float32)
    r = np.ascontiguousarray(np.concatenate([o, d], 1).T)
    k = np.arange(n)
    m = k[k % 13 == 0]
    r[3 + (m % 3), m] = 0.0                                              # axis-parallel rays: miss every box
    m = k[k % 17 == 0]
    r[3 + ((m + 1) % 3), m] = np.float32(1e-42)                          # denormal direction: 1/d = inf
    r[0, (k % 19) == 0] = np.float32(lo[0])                              # origin exactly on a box plane
    r[1, (k % 23) == 0] = np.float32(hi[1])
    r[2, (k % 101) == 0] = np.float32(3e38)                              # huge origin
    r[4, (k % 103) == 0] = np.inf
    r[5, (k % 107) == 0] = np.nan
    r[0, (k % 109) == 0] = np.nan
    # rays aimed at triangle vertices and edge midpoints: hits on shared edges, u or v exactly 0
    tri = scene.pos.reshape(-1, 3, 3)
    t = rng.integers(0, tri.shape[0], n)
    target = np.where((k % 2 == 0)[:, None], tri[t, 0], 0.5 * (tri[t, 0] + tri[t, 1]))
    aim = (k % 5) == 0
    r[3:6, aim] = (target[aim] - r[0:3, aim].T).T.astype(np.float32)
    return r


@pytest.mark.parametrize("seed,n_tri,kind", [(1, 1, "plain"), (2, 2, "dup"), (3, 3, "plain"), (4, 37, "clustered"), (5, 200, "axis"),
                                              (6, 999, "dup"), (7, 1500, "degenerate"), (8, 4096, "clustered"), (9, 20000, "plain")])
def test_fuzz_soups_bit_exact(seed, n_tri, kind):
    rng = np.random.default_rng(seed)
    s = _soup(rng, n_tri, kind)
    ctx, orc = Context(s, 32, 32), OracleScene(s)
    # the tree, node for node (reference numbering)
    for g, r in zip(ctx.get_bvh(), orc.bvh_nodes()):
        assert np.array_equal(g, r)
    rays = _rays(rng, s, 30000)
    t, id0, id1, u, v = ctx.trace_closest(rays)
    rt, rid0, rid1, ru, rv = orc.intersect(rays)
    assert np.array_equal(id1, rid1) and np.array_equal(id0, rid0)
    for a, b in ((t, rt), (u, ru), (v, rv)):
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    assert np.array_equal(ctx.trace_occluded(rays), orc.occluded(rays))
    if n_tri >= 37:
        assert (id1 >= 0).sum() > 100   # the aimed rays do hit
    st = ctx.stats()
    assert st["exact_closest_rays"] > 0   # the salted rays went through the exact follow-up launch
    ctx.close()
