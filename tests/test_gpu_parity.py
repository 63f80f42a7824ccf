"""GPU parity: the HIP path (through the C-ABI) against the CPU oracle on the
same seeded inputs.  Bit-exact for hit ids, distances and barycentrics; film
within a stated tolerance (transcendentals differ by ulps between libm and the
device maths library)."""
import numpy as np
import pytest

from raytracer_amd import Context, scenes
from oracle.binding import OracleScene, camera_rays

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def _check_closest(ctx, orc, rays):
    t, id0, id1, u, v = ctx.trace_closest(rays)
    rt, rid0, rid1, ru, rv = orc.intersect(rays)
    assert np.array_equal(id1, rid1)
    assert np.array_equal(id0, rid0)
    assert np.array_equal(_bits(t), _bits(rt))
    assert np.array_equal(_bits(u), _bits(ru))
    assert np.array_equal(_bits(v), _bits(rv))
    return id1


def test_closest_hit_cornell_bit_exact(cornell):
    ctx, orc = Context(cornell, 128, 128), OracleScene(cornell)
    rays = camera_rays(cornell.camera, 128, 128)
    id1 = _check_closest(ctx, orc, rays)
    assert (id1 >= 0).mean() > 0.5
    inc = scenes.incoherent_rays(cornell, 50000)
    _check_closest(ctx, orc, inc)
    ctx.close()


def test_closest_hit_atrium_bit_exact(atrium_small):
    ctx, orc = Context(atrium_small, 64, 64), OracleScene(atrium_small)
    rays = camera_rays(atrium_small.camera, 320, 180)
    id1 = _check_closest(ctx, orc, rays)
    assert (id1 >= 0).mean() > 0.9
    inc = scenes.incoherent_rays(atrium_small, 100000)
    _check_closest(ctx, orc, inc)
    ctx.close()


def test_ray_counts_around_chunk_and_range_boundaries(atrium_small):
    # a launch is cut into 64 ranges handed out in chunks of 16..96 rays: counts below, at and just above those
    # sizes (ranges of 0 and 1 rays, a last chunk of one ray) must all come back complete and bit-exact
    ctx, orc = Context(atrium_small, 64, 64), OracleScene(atrium_small)
    pool = scenes.incoherent_rays(atrium_small, 70000, seed=23)
    for n in (1, 2, 15, 16, 17, 31, 33, 63, 64, 65, 95, 96, 97, 127, 128, 129, 255, 257, 2047, 2049, 64 * 96 - 1, 64 * 96, 64 * 96 + 1,
              16 * 128 * 16 + 1, 65537):
        rays = np.ascontiguousarray(pool[:, :n])
        _check_closest(ctx, orc, rays)
        assert np.array_equal(ctx.trace_occluded(rays), orc.occluded(rays)), n
    ctx.close()


def test_occluded_matches_oracle(atrium_small):
    ctx, orc = Context(atrium_small, 64, 64), OracleScene(atrium_small)
    inc = scenes.incoherent_rays(atrium_small, 100000, seed=11)
    st = ctx.trace_occluded(inc)
    ref = orc.occluded(inc)
    assert np.array_equal(st, ref)
    assert 0.05 < st.mean() < 1.0
    ctx.close()


def test_bvh_identical_to_oracle(atrium_small):
    ctx, orc = Context(atrium_small, 64, 64), OracleScene(atrium_small)
    got, ref = ctx.get_bvh(), orc.bvh_nodes()
    for g, r in zip(got, ref):
        assert np.array_equal(g, r)
    ctx.close()


def _film_check(film, ref, rmse_tol, frac_tol=2e-3):
    a, b = film[..., :3].astype(np.float64), ref[..., :3].astype(np.float64)
    assert np.all(np.isfinite(a)) == np.all(np.isfinite(b))
    fin = np.isfinite(a).all(-1) & np.isfinite(b).all(-1)
    err = np.abs(a - b)[fin]
    rel = err / (np.abs(b[fin]) + 1e-3)
    rmse = float(np.sqrt(np.mean(err ** 2)))
    scale = float(np.sqrt(np.mean(b[fin] ** 2))) + 1e-12
    bad = float((rel.max(-1) > 1e-3).mean())
    assert rmse / scale < rmse_tol, f"relative RMSE {rmse/scale:.3e}"
    assert bad < frac_tol, f"{bad:.3e} of pixels off by more than 1e-3 relative"
    return rmse / scale, bad


def test_render_cornell_matches_oracle(cornell):
    W = H = 128
    ctx, orc = Context(cornell, W, H, max_depth=4, seed=0), OracleScene(cornell)
    ctx.set_camera(cornell.camera)
    ctx.render(4)
    film = ctx.read_film()
    ref, cnt = orc.render(cornell.camera, W, H, 4, 0, 4)
    assert np.array_equal(film[..., 3], ref[..., 3])
    _film_check(film, ref, 1e-4)
    ctx.close()


def test_render_env_and_point_matches_oracle(cornell_env):
    W, H = 96, 64
    ctx, orc = Context(cornell_env, W, H, max_depth=4, seed=3), OracleScene(cornell_env)
    ctx.set_camera(cornell_env.camera)
    ctx.render(3)
    film = ctx.read_film()
    ref, _ = orc.render(cornell_env.camera, W, H, 4, 3, 3)
    _film_check(film, ref, 1e-3, frac_tol=1e-2)
    ctx.close()


def test_render_atrium_matches_oracle(atrium_small):
    W, H = 160, 96
    ctx, orc = Context(atrium_small, W, H, max_depth=4, seed=5), OracleScene(atrium_small)
    ctx.set_camera(atrium_small.camera)
    ctx.render(2)
    film = ctx.read_film()
    ref, _ = orc.render(atrium_small.camera, W, H, 4, 5, 2)
    _film_check(film, ref, 1e-3, frac_tol=1e-2)
    ctx.close()


def test_untame_rays_take_the_exact_path(atrium_small):
    """Rays whose origin or inverse direction is not finite and moderate are
    traced by the follow-up launch with the select-based box test; results
    must still equal the oracle's bit for bit (NaN/inf semantics included)."""
    ctx, orc = Context(atrium_small, 64, 64), OracleScene(atrium_small)
    rays = scenes.incoherent_rays(atrium_small, 4096, seed=21).copy()
    lo, hi = atrium_small.pos.min(0), atrium_small.pos.max(0)
    tiny = np.float32(1e-42)  # denormal direction component: 1/d = inf
    rays[3, 0::8] = tiny
    rays[4, 1::8] = -tiny
    rays[5, 2::8] = 0.0                      # exact zero: invdir forced to 0
    rays[0, 3::16] = np.float32(lo[0])       # origin exactly on a box plane
    rays[3, 3::16] = tiny                    # ... with an infinite invdir: inf * 0
    rays[1, 4::64] = np.nan
    rays[3, 5::64] = np.inf
    rays[0, 6::64] = np.float32(3e38)
    rays[2, 7::64] = -np.inf
    t, id0, id1, u, v = ctx.trace_closest(rays)
    rt, rid0, rid1, ru, rv = orc.intersect(rays)
    assert np.array_equal(id1, rid1) and np.array_equal(id0, rid0)
    assert np.array_equal(_bits(t), _bits(rt)) and np.array_equal(_bits(u), _bits(ru)) and np.array_equal(_bits(v), _bits(rv))
    assert np.array_equal(ctx.trace_occluded(rays), orc.occluded(rays))
    assert (id1 >= 0).sum() > 1000
    ctx.close()


def test_render_depth8_larger_film_matches_oracle(atrium_small):
    """Deeper paths on a film large enough that every bounce's merged launch
    (shadow rays of bounce d + closest-hit rays of bounce d+1) spans many workgroups."""
    W, H = 480, 270
    ctx, orc = Context(atrium_small, W, H, max_depth=8, seed=11), OracleScene(atrium_small)
    ctx.set_camera(atrium_small.camera)
    ctx.render(3)
    film = ctx.read_film()
    ref, cnt = orc.render(atrium_small.camera, W, H, 8, 11, 3)
    _film_check(film, ref, 1e-3, frac_tol=1e-2)
    st = ctx.stats()
    # the GPU skips the closest-hit ray of the last loop iteration and shadow
    # rays that could only add zeros, so it traces no more rays than the oracle
    assert 0 < st["closest_rays"] <= cnt["closest"] and 0 < st["shadow_rays"] <= cnt["shadow"]
    ctx.close()


def test_render_is_deterministic_at_full_size():
    """Two renders of the 1080p Sponza-class workload must agree bit for bit
    (a race between workgroups appending to the next-bounce arrays or marking
    shadow entries would show up here), and the
    tile-sharded render of rank 0 of 2 must equal the same pixels of the
    unsharded one."""
    scene = scenes.sponza_class(env_size=(512, 256))
    W, H = 1920, 1080
    films = []
    for _ in range(2):
        ctx = Context(scene, W, H, max_depth=8, seed=2)
        ctx.set_camera(scene.camera)
        ctx.render(2)
        films.append(ctx.read_film())
        ctx.close()
    assert np.array_equal(_bits(films[0]), _bits(films[1]))
    assert np.isfinite(films[0]).all() and films[0][..., :3].max() > 0
    ctx = Context(scene, W, H, max_depth=8, seed=2, rank=0, world=2)
    ctx.set_camera(scene.camera)
    ctx.render(2)
    half = ctx.read_film()
    ctx.close()
    ty, tx = np.mgrid[0:H, 0:W]
    mine = ((ty // 64) * (-(-W // 64)) + tx // 64) % 2 == 0
    assert np.array_equal(_bits(half[mine]), _bits(films[0][mine]))
    assert not half[~mine].any()


def test_direct_lighting_closed_form_on_the_gpu():
    """The integrator against an answer computed by hand, not by the oracle: Lambertian plane, one point light,
    depth 1 (tests/test_oracle_kats.py: direct_lighting_case): L = color / pi * I / dist^2, no cosine term
    (src/cpu/PathTracer.cpp:106-107)."""
    from tests.test_oracle_kats import direct_lighting_case
    s, want, ok = direct_lighting_case(16, 12)
    H, W = ok.shape
    ctx = Context(s, W, H, max_depth=1, seed=0, tile_size=8)
    ctx.set_camera(s.camera)
    ctx.render(3)        # the running mean of three identical passes
    film = ctx.read_film()
    ctx.close()
    assert np.allclose(film[ok][:, :3], want[ok], rtol=2e-5, atol=0)
    assert not film[~ok][:, :3].any()
