"""Edge cases of the C-ABI on the GPU: empty and tiny inputs, ragged film
sizes, every rank's shard assembled into the unsharded film, statistics and
error paths."""
import ctypes as C

import numpy as np
import pytest

from oracle.binding import OracleScene
from raytracer_amd import Context, RtError, api, scenes
from tests.test_oracle_kats import rays_of, soup

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def test_zero_rays_and_single_ray(cornell):
    ctx = Context(cornell, 16, 16)
    empty = np.zeros((6, 0), np.float32)
    t, id0, id1, u, v = ctx.trace_closest(empty)
    assert t.size == 0 and ctx.trace_occluded(empty).size == 0
    one = rays_of(((0.1, 1.0, 3.0), (0.01, -0.02, -1.0)))
    t, id0, id1, u, v = ctx.trace_closest(one)
    rt, rid0, rid1, ru, rv = OracleScene(cornell).intersect(one)
    assert id1[0] == rid1[0] >= 0 and _bits(t)[0] == _bits(rt)[0]
    ctx.close()


def test_single_triangle_scene_root_is_leaf():
    s = soup([((-1, -1, 0), (1, -1, 0), (0, 1, 0))])
    ctx, orc = Context(s, 8, 8), OracleScene(s)
    r = rays_of(((0.1, 0.05, 1), (0.01, 0.02, -1)), ((0.1, 0.05, 0.00001), (0.01, 0.02, -1)), ((5, 5, 1), (0.01, 0.02, -1)),
                ((0.1, 0.05, 1), (0.0, 0.0, -1)))
    t, id0, id1, u, v = ctx.trace_closest(r)
    rt, rid0, rid1, ru, rv = orc.intersect(r)
    assert list(id1) == list(rid1) == [0, -1, -1, -1]   # hit, self-hit epsilon, outside, axis-parallel quirk
    assert np.array_equal(_bits(t), _bits(rt))
    assert list(ctx.trace_occluded(r)) == list(orc.occluded(r))
    ctx.close()


def test_coincident_triangles_first_visited_wins():
    t0 = ((-1, -1, 0), (1, -1, 0), (0, 1, 0))
    s = soup([t0, t0, t0], models=[0, 1, 2])
    ctx, orc = Context(s, 8, 8), OracleScene(s)
    r = rays_of(((0.1, 0.05, 1), (0.01, 0.02, -1)))
    t, id0, id1, u, v = ctx.trace_closest(r)
    rt, rid0, rid1, ru, rv = orc.intersect(r)
    assert id1[0] == rid1[0] and id0[0] == rid0[0]
    ctx.close()


@pytest.mark.parametrize("w,h,tile", [(100, 70, 64), (65, 129, 32), (7, 5, 8), (64, 64, 64)])
def test_ragged_film_sizes_match_oracle(cornell, w, h, tile):
    ctx = Context(cornell, w, h, max_depth=3, seed=1, tile_size=tile)
    ctx.set_camera(cornell.camera)
    ctx.render(2)
    film = ctx.read_film()
    ref, _ = OracleScene(cornell).render(cornell.camera, w, h, 3, 1, 2)
    assert np.array_equal(film[..., 3], ref[..., 3])
    err = np.abs(film[..., :3].astype(np.float64) - ref[..., :3])
    assert np.sqrt((err ** 2).mean()) < 1e-4 * max(1e-3, np.sqrt((ref[..., :3].astype(np.float64) ** 2).mean()))
    assert ctx.stats()["n_local_pixels"] == w * h
    ctx.close()


@pytest.mark.parametrize("world", [2, 3, 8])
def test_shards_assemble_to_the_unsharded_film(cornell_env, world):
    """Each rank renders only its interleaved tiles; packed slabs gathered in
    rank order and unpacked must equal the 1-rank film bit for bit (the
    sampler is keyed by the global pixel id)."""
    W, H, tile = 200, 136, 32
    full = Context(cornell_env, W, H, max_depth=4, seed=9, tile_size=tile)
    full.set_camera(cornell_env.camera)
    full.render(2)
    ref = full.read_film()
    full.close()
    per = max(api.slab_floats(W, H, tile, r, world) for r in range(world))
    slabs = np.zeros((world, per), np.float32)
    pixels = 0
    for r in range(world):
        ctx = Context(cornell_env, W, H, max_depth=4, seed=9, tile_size=tile, rank=r, world=world)
        ctx.set_camera(cornell_env.camera)
        ctx.render(2)
        ptr, n_tiles, n_floats = ctx.film_slab_info()
        assert n_floats == api.slab_floats(W, H, tile, r, world)
        part = ctx.read_film()
        slabs[r, :n_floats] = api.pack_slab(part, tile, r, world)
        pixels += ctx.stats()["n_local_pixels"]
        ctx.close()
    assert pixels == W * H
    out = api.unpack_slabs(W, H, tile, world, slabs, per)
    assert np.array_equal(_bits(out), _bits(ref))


def test_copy_film_slab_to_torch_tensor(cornell):
    import torch
    W, H = 96, 64
    ctx = Context(cornell, W, H)
    ctx.set_camera(cornell.camera)
    ctx.render(1)
    _, n_tiles, n_floats = ctx.film_slab_info()
    dst = torch.empty(n_floats, dtype=torch.float32, device="cuda:0")
    ctx.copy_film_slab(dst.data_ptr(), n_floats)
    out = api.unpack_slabs(W, H, 64, 1, dst.cpu().numpy(), n_floats)
    assert np.array_equal(_bits(out), _bits(ctx.read_film()))
    with pytest.raises(RtError):
        ctx.copy_film_slab(dst.data_ptr(), n_floats - 4)
    ctx.close()


def test_running_mean_and_reset(cornell_env):
    """film after k passes = fp32 running mean of the per-pass films in the
    reference's order (src/cpu/PathTracer.cpp:148-157)."""
    W, H = 64, 48
    per_pass = []
    ctx = Context(cornell_env, W, H, seed=4)
    ctx.set_camera(cornell_env.camera)
    ctx.render(3)
    acc = ctx.read_film()
    ctx.close()
    # pass p alone = a context whose sampler pass index is p: render p+1 passes and undo the mean is not
    # possible bitwise, so check against the oracle's accumulation instead
    ref, _ = OracleScene(cornell_env).render(cornell_env.camera, W, H, 4, 4, 3)
    err = np.abs(acc[..., :3].astype(np.float64) - ref[..., :3])
    assert np.sqrt((err ** 2).mean()) / np.sqrt((ref[..., :3].astype(np.float64) ** 2).mean()) < 1e-3
    ctx = Context(cornell_env, W, H, seed=4)
    ctx.set_camera(cornell_env.camera)
    ctx.render(2)
    ctx.reset_accumulation()
    ctx.render(3)
    again = ctx.read_film()
    assert np.array_equal(_bits(again), _bits(acc))  # reset restarts the pass index at 0
    assert ctx.stats()["passes"] == 3
    ctx.close()


def test_error_paths(cornell):
    ctx = Context(cornell, 32, 32)
    with pytest.raises(RtError) as e:
        ctx.render(1)  # no camera yet
    assert e.value.code == -5 and "camera" in str(e.value)
    ctx.close()
    with pytest.raises(RtError):
        Context(cornell, 32, 32, tile_size=12)
    with pytest.raises(RtError):
        Context(cornell, 32, 32, rank=2, world=2)
    with pytest.raises(RtError):
        Context(cornell, 0, 32)
    bad = scenes.cornell_box()
    bad.indices = bad.indices.copy()
    bad.indices[5] = 10_000
    with pytest.raises(RtError) as e:
        Context(bad, 32, 32)
    assert "index out of range" in str(e.value)


def test_no_frame_scene_matches_oracle(cornell):
    """Meshes without tangents (every glTF mesh in the reference,
    src/scene/Scene.cpp:191-229): both sides derive the frame from the normal."""
    W, H = 64, 64
    ctx = Context(cornell, W, H, seed=2, use_frame=False)
    ctx.set_camera(cornell.camera)
    ctx.render(2)
    film = ctx.read_film()
    ref, _ = OracleScene(cornell, use_frame=False).render(cornell.camera, W, H, 4, 2, 2)
    err = np.abs(film[..., :3].astype(np.float64) - ref[..., :3])
    assert np.sqrt((err ** 2).mean()) / np.sqrt((ref[..., :3].astype(np.float64) ** 2).mean()) < 1e-4
    ctx.close()


@pytest.mark.parametrize("size", [(100, 37), (7, 5), (256, 128)])
def test_env_sampling_sparse_maps(size):
    """Environment maps whose CDFs have long flat stretches and odd sizes: the
    guided interval search must land on the interval the reference's binary
    search finds (src/scene/Common.hpp:13-25, src/samplers/Distributions.cpp:26-35)."""
    W, H = size
    rng = np.random.default_rng(W)
    env = np.zeros((H, W, 4), np.float32)
    env[..., 3] = 1
    for _ in range(max(3, W * H // 200)):
        env[rng.integers(0, H // 2 + 1), rng.integers(0, W), :3] = rng.uniform(5, 50, 3)
    base = scenes.cornell_box()
    s = scenes.replace(base, lights=[scenes.Light(scenes.LIGHT_ENV, intensity=np.ones(3, np.float32), env=env)])
    ref, _ = OracleScene(s).render(s.camera, 48, 48, 3, 6, 3, threads=4)
    ctx = Context(s, 48, 48, max_depth=3, seed=6)
    ctx.set_camera(s.camera)
    ctx.render(3)
    film = ctx.read_film()
    ctx.close()
    assert ref[..., :3].max() > 0
    err = np.abs(film[..., :3] - ref[..., :3])
    assert float(np.sqrt((err ** 2).mean()) / np.sqrt((ref[..., :3] ** 2).mean())) < 1e-4


def test_batched_passes_leave_the_same_film(cornell_env):
    """rt_render may trace several passes as one wavefront (rt_params.batch);
    the film must be the one single passes leave, bit for bit, also when the
    pass count is not a multiple of the batch and for textured scenes."""
    s = scenes.with_textures(cornell_env, uv_scale=3.0)
    films = []
    for batch, chunks in ((1, (5,)), (4, (5,)), (3, (2, 3)), (8, (1, 1, 3))):
        ctx = Context(s, 50, 38, max_depth=3, seed=9, tile_size=16, batch=batch)
        ctx.set_camera(s.camera)
        for n in chunks:
            ctx.render(n)
        films.append(ctx.read_film())
        st = ctx.stats()
        assert st["passes"] == 5
        ctx.close()
    for f in films[1:]:
        assert np.array_equal(_bits(f), _bits(films[0]))
    ref, _ = OracleScene(s).render(s.camera, 50, 38, 3, 9, 5, threads=4, tile_size=16)
    err = np.abs(films[0][..., :3] - ref[..., :3])
    assert float(np.sqrt((err ** 2).mean()) / np.sqrt((ref[..., :3] ** 2).mean())) < 1e-4


def test_last_bounce_rays_match_oracle(cornell_env):
    """The workload-capture hook returns the rays the oracle's integrator
    traces after the primary hit (src/cpu/PathTracer.cpp:134-135): same
    origins bit for bit, directions up to the ulps of the device sin/cos."""
    s = cornell_env
    ctx = Context(s, 40, 30, max_depth=2, seed=3, batch=1)
    ctx.set_camera(s.camera)
    ctx.render(1)
    got = ctx.last_bounce_rays()
    ctx.close()
    from oracle.binding import lib as olib
    o = OracleScene(s)
    rays = np.empty((6, 1200), np.float32)
    flags = np.empty(1200, np.uint8)
    olib().orc_capture_bounce_rays(o._h, s.camera.packed().ctypes.data_as(C.POINTER(C.c_float)), 40, 30, 2, 3, 0, 2, 0,
                                   rays.ctypes.data_as(C.POINTER(C.c_float)), flags.ctypes.data_as(C.POINTER(C.c_uint8)))
    ref = rays[:, flags.astype(bool)]
    assert got.shape == ref.shape and ref.shape[1] > 300
    # both are in pixel order up to the tile/8x8 slot order of the GPU: sort by origin
    ko, kr = np.lexsort(got[:3][::-1]), np.lexsort(ref[:3][::-1])
    assert np.array_equal(_bits(got[:3, ko]), _bits(ref[:3, kr]))
    assert np.abs(got[3:, ko] - ref[3:, kr]).max() < 1e-5
    assert scenes.bounce_rays(s, 40, 30, 2000).shape == (6, 2000)


@pytest.mark.parametrize("depth", [1, 2])
def test_no_lights_and_shallow_depths(cornell, depth):
    """A scene without lights renders black where rays hit (nothing is ever
    added, src/cpu/PathTracer.cpp:96) and the depth loop still runs; depth 1
    has no bounce rays at all."""
    dark = scenes.replace(cornell, lights=[])
    ctx = Context(dark, 40, 24, max_depth=depth, seed=1, tile_size=8)
    ctx.set_camera(dark.camera)
    ctx.render(3)
    film = ctx.read_film()
    st = ctx.stats()
    ctx.close()
    ref, cnt = OracleScene(dark).render(dark.camera, 40, 24, depth, 1, 3, threads=2, tile_size=8)
    assert np.array_equal(_bits(film), _bits(ref))
    assert not film[..., :3].any() and st["shadow_rays"] == 0
    # the oracle also counts the closest-hit ray of the last loop iteration, which the GPU does not trace
    assert st["closest_rays"] <= cnt["closest"]
    lit = Context(cornell, 40, 24, max_depth=depth, seed=1, tile_size=8)
    lit.set_camera(cornell.camera)
    lit.render(2)
    ref, _ = OracleScene(cornell).render(cornell.camera, 40, 24, depth, 1, 2, threads=2, tile_size=8)
    err = np.abs(lit.read_film()[..., :3] - ref[..., :3])
    assert err.max() < 1e-5
    lit.close()


def test_more_ranks_than_tiles(cornell):
    """A rank without tiles owns an empty slab and renders nothing."""
    c = Context(cornell, 16, 16, max_depth=2, tile_size=16, rank=2, world=3)
    c.set_camera(cornell.camera)
    c.render(2)
    assert not c.read_film().any()
    assert c.stats()["n_local_pixels"] == 0 and c.film_slab_info()[2] == 0
    assert c.last_bounce_rays().shape == (6, 0)
    c.close()


@pytest.mark.parametrize("w,h", [(1, 1), (3, 5), (2, 9)])
def test_films_smaller_than_a_tile(cornell_env, w, h):
    s = scenes.with_textures(cornell_env, uv_scale=2.0)
    ctx = Context(s, w, h, max_depth=3, seed=2, tile_size=8)
    ctx.set_camera(s.camera)
    ctx.render(4)
    film = ctx.read_film()
    ctx.close()
    ref, _ = OracleScene(s).render(s.camera, w, h, 3, 2, 4, threads=1, tile_size=8)
    assert np.abs(film[..., :3] - ref[..., :3]).max() <= 1e-4 * max(1.0, float(np.abs(ref[..., :3]).max()))


def test_many_lights_match_oracle(cornell):
    """Light loop order and the per-light random-number slots with 24 lights of all three kinds."""
    rng = np.random.default_rng(4)
    lights = [scenes.Light(scenes.LIGHT_POINT, rng.uniform([-0.8, 0.3, -0.8], [0.8, 1.9, 0.8]).astype(np.float32),
                           rng.uniform(0.05, 0.4, 3).astype(np.float32)) for _ in range(18)]
    lights += scenes.area_quad((-0.3, 1.98, -0.3), (0.3, 1.98, -0.3), (0.3, 1.98, 0.3), (-0.3, 1.98, 0.3), (6.0, 5.0, 4.0))
    lights += scenes.area_quad((-0.99, 0.5, 0.4), (-0.99, 0.5, -0.4), (-0.99, 1.4, -0.4), (-0.99, 1.4, 0.4), (2.0, 3.0, 5.0))
    lights.insert(7, scenes.Light(scenes.LIGHT_ENV, intensity=np.ones(3, np.float32), env=scenes.sky_env(32, 16)))
    lights.insert(15, scenes.Light(scenes.LIGHT_ENV, intensity=np.full(3, 0.5, np.float32), env=scenes.sky_env(16, 8, seed=3)))
    s = scenes.replace(cornell, lights=lights)
    ctx = Context(s, 48, 40, max_depth=3, seed=8, tile_size=8, batch=2)
    ctx.set_camera(s.camera)
    ctx.render(3)
    film = ctx.read_film()
    ctx.close()
    ref, _ = OracleScene(s).render(s.camera, 48, 40, 3, 8, 3, threads=4, tile_size=8)
    err = np.abs(film[..., :3] - ref[..., :3])
    assert float(np.sqrt((err ** 2).mean()) / np.sqrt((ref[..., :3] ** 2).mean())) < 1e-4


def test_degenerate_triangles_are_harmless():
    """Zero-area and repeated-vertex triangles never report a hit (|det| < 1e-8,
    src/scene/Ray.cpp) and do not disturb the tree or the hits around them."""
    from tests.test_host_logic import _random_soup
    s = _random_soup(300, 6)
    idx = s.indices.copy()
    idx[0:3] = idx[0]                 # a point
    idx[3:6] = [idx[3], idx[4], idx[3]]   # a segment
    s2 = scenes.replace(s, indices=idx)
    ctx, orc = Context(s2, 16, 16), OracleScene(s2)
    rays = scenes.incoherent_rays(s2, 30000, seed=12)
    got, ref = ctx.trace_closest(rays), orc.intersect(rays)
    for g, r in zip(got, ref):
        assert np.array_equal(_bits(g), _bits(r))
    assert not np.isin(got[2], [0, 3]).any()
    for g, r in zip(ctx.get_bvh(), orc.bvh_nodes()):
        assert g.shape == r.shape
    ctx.close()
