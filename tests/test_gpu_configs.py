"""The two largest workload configurations at their own sizes, on one GPU:

* the Bistro-class street block (3.16 M triangles, depth 12, 64 point lights + sky, and the variant with 64 area
  lights): a 240x136 film against the shared-math oracle bit for bit, the GPU-built LBVH against the host-built one,
  and at 1920x1080 the size-independent properties (any-hit == closest-hit, determinism);
* the 3840x2160 depth-8 Sponza-class render split over 2 and 8 emulated ranks by interleaved tiles (one context per
  rank on the same GPU), assembled on the device the way the multi-rank read-out does, against the unsharded film.
"""
import numpy as np
import pytest

from oracle import binding as oracle_binding
from oracle.binding import OracleScene
from raytracer_amd import Context, api, scenes

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


@pytest.fixture(scope="module")
def bistro():
    return scenes.bistro_class()


@pytest.fixture(scope="module")
def bistro_area():
    return scenes.bistro_class(area_lights=True)


def _film_vs_shared_oracle(scene, W, H, depth, seed, passes):
    ctx = Context(scene, W, H, max_depth=depth, seed=seed)
    ctx.set_camera(scene.camera)
    ctx.render(passes)
    film = ctx.read_film()
    st = ctx.stats()
    ctx.close()
    orc = OracleScene(scene)
    oracle_binding.set_math("shared")
    try:
        ref, cnt = orc.render(scene.camera, W, H, depth, seed, passes, threads=16)
    finally:
        oracle_binding.set_math("libm")
    same = (_bits(film) == _bits(ref)) | (np.isnan(film) & np.isnan(ref))
    assert same.all(), f"{int((~same).any(-1).sum())} of {W*H} pixels differ"
    assert np.isfinite(film).all() and film[..., :3].max() > 0
    # the GPU skips the last iteration's closest-hit ray and shadow rays that can only add zeros
    assert 0 < st["closest_rays"] <= cnt["closest"] and 0 < st["shadow_rays"] <= cnt["shadow"]
    return st


def test_bistro_class_film_bit_identical_to_oracle(bistro):
    assert bistro.n_triangles > 2_800_000 and len(bistro.lights) >= 64
    st = _film_vs_shared_oracle(bistro, 240, 136, 12, 4, 2)
    assert st["n_triangles"] == bistro.n_triangles and st["n_nodes"] == 2 * bistro.n_triangles - 1


def test_bistro_class_area_lights_film_bit_identical_to_oracle(bistro_area):
    assert sum(1 for l in bistro_area.lights if l.type == scenes.LIGHT_AREA) >= 64
    _film_vs_shared_oracle(bistro_area, 240, 136, 12, 4, 2)


def test_bistro_class_gpu_tree_equals_host_tree(bistro):
    ctx = Context(bistro, 64, 64)
    boxes, left, right, tri, model, leaf = ctx.get_bvh()
    depth = ctx.stats()["bvh_max_depth"]
    ctx.close()
    md, hb, hl, hr, ht, hm, hleaf = api.lbvh_host(bistro)
    assert md == depth
    assert np.array_equal(leaf, hleaf)
    inner = leaf == 0
    assert np.array_equal(left[inner], hl[inner]) and np.array_equal(right[inner], hr[inner])
    assert np.array_equal(tri[~inner], ht[~inner]) and np.array_equal(model[~inner], hm[~inner])
    assert np.array_equal(boxes, hb)


def test_bistro_class_full_size_properties(bistro):
    """1920x1080: an unbounded shadow ray is occluded exactly when the closest-hit query of the same ray finds
    something; two depth-12 renders with all 65 lights agree bit for bit."""
    W, H = 1920, 1080
    ctx = Context(bistro, W, H, max_depth=12, seed=3, batch=1)
    for rays in (scenes.camera_rays_numpy(bistro.camera, W, H), scenes.incoherent_rays(bistro, W * H, seed=9)):
        occ = ctx.trace_occluded(rays)
        t, id0, id1, u, v = ctx.trace_closest(rays)
        hit = id1 >= 0
        assert np.array_equal(occ.astype(bool), hit)
        assert np.all(t[hit] > 1e-4) and np.all(np.isinf(t[~hit]))
        assert id1[hit].max() < 3 * bistro.n_triangles and np.all(id1[hit] % 3 == 0)
        assert np.array_equal(id0[hit], bistro.tri_model[id1[hit] // 3])
    ctx.set_camera(bistro.camera)
    ctx.render(1)
    a = ctx.read_film()
    ctx.reset_accumulation()
    ctx.render(1)
    b = ctx.read_film()
    st = ctx.stats()
    ctx.close()
    assert np.array_equal(_bits(a), _bits(b))
    assert np.isfinite(a).all() and a[..., :3].max() > 0
    assert st["shadow_rays"] > 10 * st["closest_rays"]   # 65 lights per vertex


def test_4k_tile_sharded_film_equals_unsharded_film():
    """3840x2160, depth 8, 64x64 tiles dealt to 2 and to 8 ranks: every rank renders only its tiles, the slabs are
    laid out the way the all-gather leaves them and scattered on the device; the film equals the unsharded one."""
    import torch
    scene = scenes.sponza_class()
    W, H, depth, passes = 3840, 2160, 8, 2
    whole = Context(scene, W, H, max_depth=depth, seed=7, batch=passes)
    whole.set_camera(scene.camera)
    whole.render(passes)
    want = whole.read_film()
    assert np.isfinite(want).all() and want[..., :3].max() > 0
    film = torch.empty((H, W, 4), dtype=torch.float32, device="cuda:0")
    for world in (2, 8):
        per_rank = max(api.slab_floats(W, H, 64, r, world) for r in range(world))
        gathered = torch.zeros(world * per_rank, dtype=torch.float32, device="cuda:0")
        torch.cuda.synchronize()
        traced = 0
        for r in range(world):
            c = Context(scene, W, H, max_depth=depth, seed=7, rank=r, world=world, batch=passes)
            c.set_camera(scene.camera)
            c.render(passes)
            _, n_tiles, nf = c.film_slab_info()
            assert nf == api.slab_floats(W, H, 64, r, world) and n_tiles * 64 * 64 * 4 == nf
            c.copy_film_slab(gathered[r * per_rank:].data_ptr(), nf)
            st = c.stats()
            traced += st["closest_rays"] + st["shadow_rays"]
            c.close()
        film.fill_(-1.0)
        torch.cuda.synchronize()
        whole.unpack_slabs_dev(world, gathered.data_ptr(), per_rank, film.data_ptr())
        got = film.cpu().numpy()
        assert np.array_equal(_bits(got), _bits(want)), f"world {world}: sharded film differs"
        st = whole.stats()
        assert traced == st["closest_rays"] + st["shadow_rays"]   # the ranks traced exactly the unsharded render's rays
    whole.close()


@pytest.mark.parametrize("maker,W,H,batch", [
    (lambda: scenes.cornell_box(env=True), 300, 200, 5),
    (lambda: scenes.with_textures(scenes.cornell_box()), 128, 128, 3),
    (lambda: scenes.cornell_area(), 100, 100, 16),
])
def test_path_state_allocations_match_the_stated_budget(maker, W, H, batch):
    """The device memory that grows with slots x batch is what rt_hip.h states: 144 B per slot + 41 B per slot and
    light (+4 B with area lights); on top, per slot of ONE pass: the film (16 B), the per-camera cache (primary ray
    and hit 44 B, one visibility byte per light when the scene has point lights) and the UV differences of
    textured scenes (16 B)."""
    s = maker()
    ctx = Context(s, W, H, max_depth=4, batch=batch)
    st = ctx.stats()
    ctx.close()
    assert st["batch"] == batch
    n1 = -(-W // 64) * -(-H // 64) * 64 * 64
    L = max(1, len(s.lights))
    area = any(l.type == scenes.LIGHT_AREA for l in s.lights)
    point = any(l.type == scenes.LIGHT_POINT for l in s.lights)
    per_slot = 144 + (41 + (4 if area else 0)) * L
    stated = per_slot * n1 * batch + (16 + 44 + (len(s.lights) if point else 0) + (16 if s.textures else 0)) * n1
    # the first-bounce queue holds only the slots inside the film (up to 4 B per slot fewer); the per-launch work
    # counters and queue counts are a few KiB on top
    assert stated - 4 * n1 * batch <= st["path_state_bytes"] <= stated + 65536
    assert st["device_bytes"] > st["path_state_bytes"]


def test_config3_sponza_ggx_nee_as_worded():
    """BASELINE.json configs[2]: Sponza-class, 1920x1080, depth 8, diffuse + GGX lobes on every model
    (src/scene/Material.cpp:205-209 switched on, Trowbridge-Reitz distribution), next-event estimation towards a point
    light, the sky and a two-triangle ceiling panel.  A 480x270 film against the shared-math oracle bit for bit; at
    1920x1080 two renders agree bit for bit and rank 0 of 2 renders exactly its tiles of the unsharded film."""
    sp = scenes.sponza_class()
    panel = scenes.area_quad((-6, 14.5, -3), (6, 14.5, -3), (6, 14.5, 3), (-6, 14.5, 3), (60.0, 55.0, 45.0))
    sg = scenes.with_glossy(scenes.replace(sp, lights=sp.lights + panel), scenes.GLOSSY_GGX)
    assert (sg.material_glossy == scenes.GLOSSY_GGX).all() and len(sg.lights) == len(sp.lights) + 2
    st = _film_vs_shared_oracle(sg, 480, 270, 8, 4, 2)
    assert st["n_triangles"] == sp.n_triangles
    W, H = 1920, 1080
    films = []
    for _ in range(2):
        ctx = Context(sg, W, H, max_depth=8, seed=2)
        ctx.set_camera(sg.camera)
        ctx.render(2)
        films.append(ctx.read_film())
        ctx.close()
    assert np.array_equal(_bits(films[0]), _bits(films[1]))
    assert np.isfinite(films[0]).all() and films[0][..., :3].max() > 0
    ctx = Context(sg, W, H, max_depth=8, seed=2, rank=0, world=2)
    ctx.set_camera(sg.camera)
    ctx.render(2)
    half = ctx.read_film()
    ctx.close()
    ty, tx = np.mgrid[0:H, 0:W]
    mine = ((ty // 64) * (-(-W // 64)) + tx // 64) % 2 == 0
    assert np.array_equal(_bits(half[mine]), _bits(films[0][mine]))
    assert not half[~mine].any()
