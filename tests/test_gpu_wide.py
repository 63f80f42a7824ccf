"""GPU parity of the 4-wide tree (RT_FLAG_WIDE_BVH) against the CPU oracle.

The wide traversal reaches the triangles the reference's recursion reaches and keeps the strict `<` of
src/scene/BVH.cpp:139 and the cull `dist < hit.dist` of src/scene/BVH.cpp:155-158, but visits children in order of
their own box distance.  Which of several triangles at (nearly) the same distance is kept therefore depends on the
order, on both sides.  The bar:
  * any hit: identical to the oracle (a boolean does not depend on the order);
  * closest hit, class 0 (every ray of the benchmark's Sponza-class sets): distance, id and barycentrics bit-equal;
  * class 1, exact ties: same distance bits, another id, on rays where two or more reachable triangles lie at
    exactly that distance (the oracle's order-independent per-ray tie flag);
  * class 2, near ties: the distance differs by at most 4 ulp (coplanar or abutting triangles whose computed
    distances differ in the last bits: the one found first culls the other's box, in either order);
    in classes 1 and 2 the reported triangle must be one the reference's own triangle test places at exactly the
    reported distance and barycentrics, and the recursion must be able to reach it;
  * the tree: every triangle below exactly one slot, every slot box the exact union of what hangs below it.
Classes 1 and 2 are counted; a scene without coplanar duplicates has none of class 2.
"""
import numpy as np
import pytest

from raytracer_amd import Context, scenes
from oracle.binding import OracleScene, camera_rays

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def _check_closest_wide(ctx, orc, rays, max_near=0.0):
    """Returns (rays with another id at the same distance, rays the oracle flags as tied, rays of class 2);
    max_near = the share of rays allowed in class 2."""
    t, id0, id1, u, v = ctx.trace_closest(rays)
    rt, rid0, rid1, ru, rv = orc.intersect(rays)
    near = _bits(t) != _bits(rt)
    assert near.sum() <= max_near * rays.shape[1], f"{int(near.sum())} rays with another distance"
    if near.any():
        ulps = np.abs(_bits(t[near]).astype(np.int64) - _bits(rt[near]).astype(np.int64))
        assert ulps.max() <= 4 and (id1[near] >= 0).all() and (rid1[near] >= 0).all()
    tied = orc.intersect_ties(rays).astype(bool)
    diff = (id1 != rid1) & ~near
    assert not (diff & ~tied).any(), "at equal distance ids may differ only on rays with an exact-distance tie"
    same = ~diff & ~near
    assert np.array_equal(id0[same], rid0[same])
    assert np.array_equal(_bits(u[same]), _bits(ru[same])) and np.array_equal(_bits(v[same]), _bits(rv[same]))
    other = diff | near
    if other.any():
        # the other triangle is a genuine hit at exactly the reported distance, reachable by the recursion
        sub = np.ascontiguousarray(rays[:, other])
        ct, cu, cv, ok = orc.check_candidates(sub, id1[other])
        assert (ok == 3).all()
        assert np.array_equal(_bits(ct), _bits(t[other]))
        assert np.array_equal(_bits(cu), _bits(u[other])) and np.array_equal(_bits(cv), _bits(v[other]))
    return int(diff.sum()), int(tied.sum()), int(near.sum())


def test_wide_tree_structure(atrium_small):
    ctx = Context(atrium_small, 64, 64, wide_bvh=True)
    nodes, depth = ctx.get_wide_bvh()
    T = atrium_small.n_triangles
    n = nodes.shape[0]
    assert 0 < n <= T - 1 and depth > 0
    boxes = nodes[:, :24].reshape(n, 6, 4)            # plane, slot
    links = nodes[:, 24:28].view(np.int32)
    used = nodes[:, 28].view(np.int32)
    assert used.min() >= 2 and used.max() <= 4
    slot = np.arange(4)[None, :]
    live = slot < used[:, None]
    assert (links[~live] == np.int32(-2**31)).all() and np.isposinf(boxes.transpose(0, 2, 1)[~live]).all()
    inner = live & (links >= 0)
    leaf = live & (links < 0)
    # a tree over all nodes: every node but the root is the child of exactly one slot, children come later (breadth first)
    kids = np.sort(links[inner])
    assert np.array_equal(kids, np.arange(1, n))
    parent_of_kid = np.nonzero(inner)[0]
    assert (links[inner] > parent_of_kid).all()
    # every triangle slot of the Morton-ordered triangle array below exactly one leaf link
    assert np.array_equal(np.sort(~links[leaf]), np.arange(T))
    # a slot that is an inner node carries the exact union of that node's slot boxes
    b = np.where(live[:, None, :], boxes, np.nan)
    lo = np.nanmin(b[:, :3, :], axis=2)
    hi = np.nanmax(b[:, 3:, :], axis=2)
    pi, ps = np.nonzero(inner)
    child = links[pi, ps]
    assert np.array_equal(boxes[pi, :3, ps], lo[child]) and np.array_equal(boxes[pi, 3:, ps], hi[child])
    # the leaf boxes are the triangles' own bounds (as a multiset)
    tri = atrium_small.pos[atrium_small.indices.reshape(-1, 3)]
    want = np.concatenate([tri.min(1), tri.max(1)], axis=1)
    li, ls = np.nonzero(leaf)
    got = boxes[li, :, ls]
    order = lambda a: a[np.lexsort(a.T[::-1])]
    assert np.array_equal(order(got), order(want))
    # the root covers the scene
    assert np.array_equal(lo[0], atrium_small.pos[atrium_small.indices].min(0)) and np.array_equal(hi[0], atrium_small.pos[atrium_small.indices].max(0))
    ctx.close()


def test_wide_closest_and_any_cornell(cornell):
    ctx, orc = Context(cornell, 128, 128, wide_bvh=True), OracleScene(cornell)
    rays = camera_rays(cornell.camera, 128, 128)
    _check_closest_wide(ctx, orc, rays)
    assert np.array_equal(ctx.trace_occluded(rays), orc.occluded(rays))
    # the two boxes stand ON the floor: their bottom faces and the floor are coplanar, the one scene here with class 2
    inc = scenes.incoherent_rays(cornell, 50000)
    _check_closest_wide(ctx, orc, inc, max_near=0.005)
    assert np.array_equal(ctx.trace_occluded(inc), orc.occluded(inc))
    ctx.close()


def test_wide_closest_and_any_atrium(atrium_small):
    ctx, orc = Context(atrium_small, 64, 64, wide_bvh=True), OracleScene(atrium_small)
    rays = camera_rays(atrium_small.camera, 320, 180)
    _check_closest_wide(ctx, orc, rays)
    assert np.array_equal(ctx.trace_occluded(rays), orc.occluded(rays))
    inc = scenes.incoherent_rays(atrium_small, 100000)
    _check_closest_wide(ctx, orc, inc)
    assert np.array_equal(ctx.trace_occluded(inc), orc.occluded(inc))
    # a far bound per ray (the shadow rays of the area lights): bounds below, at and above the first hit
    t, _, id1, _, _ = ctx.trace_closest(inc)
    rng = np.random.default_rng(5)
    tmax = np.where(id1 >= 0, t * rng.choice(np.float32([0.5, 0.9999, 1.0, 1.0001, 2.0]), t.shape), np.float32(50)).astype(np.float32)
    tmax[::97] = np.inf
    tmax[1::97] = 0.0
    tmax[2::97] = -1.0
    assert np.array_equal(ctx.trace_occluded_bounded(inc, tmax), orc.occluded_bounded(inc, tmax))
    ctx.close()


def test_wide_exact_ties_are_the_only_id_differences(cornell):
    """Every triangle twice, the copy in another model: each hit is an exact-distance tie between a triangle and its
    copy, so which of the two is reported depends on the visiting order -- and on nothing else."""
    T = cornell.n_triangles
    idx = np.concatenate([cornell.indices, cornell.indices]).astype(np.uint32)
    tm = np.concatenate([cornell.tri_model, (cornell.tri_model + 1) % cornell.n_models]).astype(np.int32)
    twice = scenes.replace(cornell, indices=idx, tri_model=tm, name="cornell twice")
    assert twice.n_triangles == 2 * T
    ctx, orc = Context(twice, 64, 64, wide_bvh=True), OracleScene(twice)
    rays = np.concatenate([camera_rays(twice.camera, 96, 96), scenes.incoherent_rays(twice, 20000, seed=3)], axis=1)
    n_diff, n_tied, _ = _check_closest_wide(ctx, orc, rays, max_near=0.005)
    _, _, rid1, _, _ = orc.intersect(rays)
    assert n_tied >= int((rid1 >= 0).sum()) * 0.99      # (almost) every hit ray is tied
    assert np.array_equal(ctx.trace_occluded(rays), orc.occluded(rays))
    ctx.close()


def test_wide_untame_rays_go_to_the_exact_launch(atrium_small):
    """Rays the hardware min/max box test cannot decide are re-traced on the binary tree by the exact launch:
    bit-equal to the oracle, ids included."""
    ctx, orc = Context(atrium_small, 64, 64, wide_bvh=True), OracleScene(atrium_small)
    rays = scenes.incoherent_rays(atrium_small, 4096, seed=21).copy()
    lo = atrium_small.pos.min(0)
    tiny = np.float32(1e-42)
    rays[3, 0::8] = tiny
    rays[4, 1::8] = -tiny
    rays[5, 2::8] = 0.0
    rays[0, 3::16] = np.float32(lo[0])
    rays[3, 3::16] = tiny
    rays[1, 4::64] = np.nan
    rays[3, 5::64] = np.inf
    rays[0, 6::64] = np.float32(3e38)
    rays[2, 7::64] = -np.inf
    _check_closest_wide(ctx, orc, rays)
    assert np.array_equal(ctx.trace_occluded(rays), orc.occluded(rays))
    tmax = np.full(rays.shape[1], 7.5, np.float32)
    tmax[::5] = np.nan
    tmax[1::5] = np.inf
    assert np.array_equal(ctx.trace_occluded_bounded(rays, tmax), orc.occluded_bounded(rays, tmax))
    assert ctx.stats()["exact_closest_rays"] > 0
    ctx.close()


def test_wide_tiny_scenes():
    """One triangle (no inner node at all: the root is a leaf), two and three triangles (a single wide node with
    unused slots), five (two levels)."""
    base = scenes.cornell_box()
    for T in (1, 2, 3, 5):
        s = scenes.replace(base, indices=np.ascontiguousarray(base.indices[:3 * T]), tri_model=np.ascontiguousarray(base.tri_model[:T]))
        ctx, orc = Context(s, 32, 32, wide_bvh=True), OracleScene(s)
        nodes, depth = ctx.get_wide_bvh()
        assert nodes.shape[0] == (0 if T == 1 else (1 if T <= 4 else nodes.shape[0]))
        rays = np.concatenate([camera_rays(s.camera, 64, 64), scenes.incoherent_rays(base, 5000, seed=T)], axis=1)
        _check_closest_wide(ctx, orc, rays)
        assert np.array_equal(ctx.trace_occluded(rays), orc.occluded(rays)), T
        ctx.close()


def test_wide_ray_counts_around_chunk_boundaries(atrium_small):
    ctx, orc = Context(atrium_small, 64, 64, wide_bvh=True), OracleScene(atrium_small)
    pool = scenes.incoherent_rays(atrium_small, 70000, seed=23)
    for n in (1, 2, 15, 17, 33, 63, 65, 95, 97, 127, 129, 257, 2049, 64 * 96 - 1, 64 * 96 + 1, 16 * 128 * 16 + 1, 65537):
        rays = np.ascontiguousarray(pool[:, :n])
        _check_closest_wide(ctx, orc, rays)
        assert np.array_equal(ctx.trace_occluded(rays), orc.occluded(rays)), n
    ctx.close()


def test_wide_deep_tree_spills_and_refills_the_lds_stack():
    """A long sliver-shaped scene: its LBVH is deep, so lanes overflow the LDS stack and go through the spill /
    refill path of the wide kernel."""
    rng = np.random.default_rng(9)
    n = 6000
    # thin triangles stacked along x, all crossing the ray corridor: many children pass at every level
    x = np.sort(rng.uniform(0, 100, n)).astype(np.float32)
    v0 = np.stack([x, rng.uniform(-1, -0.5, n), rng.uniform(-1, 1, n)], 1)
    v1 = np.stack([x + 0.01, rng.uniform(0.5, 1, n), rng.uniform(-1, 1, n)], 1)
    v2 = np.stack([x + 0.02, rng.uniform(-1, 1, n), rng.uniform(1.5, 2, n)], 1)
    pos = np.ascontiguousarray(np.stack([v0, v1, v2], 1).reshape(-1, 3), np.float32)
    base = scenes.cornell_box()
    nrm = np.tile(np.float32([1, 0, 0]), (3 * n, 1))
    s = scenes.Scene(pos=pos, normal=nrm, uv=np.zeros((3 * n, 2), np.float32), tangent=np.tile(np.float32([0, 1, 0]), (3 * n, 1)),
                     binormal=np.tile(np.float32([0, 0, 1]), (3 * n, 1)), indices=np.arange(3 * n, dtype=np.uint32),
                     tri_model=np.zeros(n, np.int32), materials=base.materials[:1].copy(), lights=base.lights, camera=base.camera)
    ctx, orc = Context(s, 32, 32, wide_bvh=True), OracleScene(s)
    m = 20000
    org = np.stack([rng.uniform(-5, 105, m), rng.uniform(-0.4, 0.4, m), rng.uniform(-0.5, 1.2, m)]).astype(np.float32)
    d = np.stack([rng.choice([-1.0, 1.0], m), rng.normal(0, 0.01, m), rng.normal(0, 0.01, m)]).astype(np.float32)
    rays = np.ascontiguousarray(np.concatenate([org, d], 0))
    _check_closest_wide(ctx, orc, rays)
    # the any-hit walk of rays that hit nothing visits (and stacks) everything along the corridor
    org2 = org.copy(); org2[1] = 0.95; org2[2] = -0.95
    rays2 = np.ascontiguousarray(np.concatenate([org2, d], 0))
    assert np.array_equal(ctx.trace_occluded(rays2), orc.occluded(rays2))
    assert np.array_equal(ctx.trace_occluded(rays), orc.occluded(rays))
    ctx.close()


def test_wide_film_equals_binary_film(cornell_env, atrium_small):
    """rt_render on the wide tree: shadow rays are order-independent and closest-hit rays differ from the binary
    tree's only on exact-distance ties, so the films agree up to those pixels (none in these scenes)."""
    for s, (W, H, D) in ((cornell_env, (96, 96, 4)), (atrium_small, (160, 96, 6))):
        films = []
        for wide in (False, True):
            ctx = Context(s, W, H, max_depth=D, seed=5, wide_bvh=wide)
            ctx.set_camera(s.camera)
            ctx.render(3)
            films.append(ctx.read_film())
            ctx.close()
        a, b = films
        differing = int((_bits(a) != _bits(b)).any(-1).sum())
        assert differing <= 0.001 * W * H, differing
        assert np.isfinite(b).all()
