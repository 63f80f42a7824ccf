"""GPU parity for the glossy lobe (Beckmann / GGX) and triangle area lights
against the CPU oracle.  Neither exists as executable code in the reference
(src/scene/Material.cpp:205-209 is a comment; LightSource::Area has no class),
so the oracle's definition is the specification: parity unpinned.  Films are
compared with the tolerance of the other film tests; the bounded any-hit rays
are bit-exact."""
import numpy as np
import pytest

from raytracer_amd import Context, scenes
from raytracer_amd.api import RtError
from oracle.binding import OracleScene
from tests.test_gpu_parity import _film_check

pytestmark = pytest.mark.gpu


def _render_pair(s, W, H, depth, seed, passes, **kw):
    ref, cnt = OracleScene(s).render(s.camera, W, H, depth, seed, passes, threads=8, **kw)
    ctx = Context(s, W, H, max_depth=depth, seed=seed, **kw)
    ctx.set_camera(s.camera)
    ctx.render(passes)
    film = ctx.read_film()
    st = ctx.stats()
    ctx.close()
    return film, ref, cnt, st


def test_bounded_any_hit_bit_exact(atrium_small):
    ctx, orc = Context(atrium_small, 64, 64), OracleScene(atrium_small)
    inc = scenes.incoherent_rays(atrium_small, 100000, seed=21)
    t = orc.intersect(inc)[0]
    rng = np.random.default_rng(0)
    # bounds around the first hit, far beyond it, tiny, infinite
    tmax = np.where(np.isfinite(t), t, 10.0).astype(np.float32) * rng.choice(np.array([0.5, 0.999, 1.0, 1.001, 3.0], np.float32), t.size)
    tmax[::7] = np.inf
    tmax[1::7] = 1e-5
    got, ref = ctx.trace_occluded_bounded(inc, tmax), orc.occluded_bounded(inc, tmax)
    assert np.array_equal(got, ref)
    assert 0.1 < got.mean() < 0.9
    unb = ctx.trace_occluded(inc)
    assert np.array_equal(got[::7], unb[::7]) and not got[1::7].any()
    ctx.close()


def test_area_light_film():
    s = scenes.cornell_area()
    film, ref, cnt, st = _render_pair(s, 96, 96, 4, 0, 4)
    assert ref[..., :3].mean() > 0.02            # the bounded shadow rays do reach the panel
    _film_check(film, ref, 1e-4)
    # only zero-contribution shadow rays and the last bounce's closest-hit ray are skipped
    assert st["shadow_rays"] <= cnt["shadow"] and st["closest_rays"] <= cnt["closest"]


def test_area_and_reference_lights_together():
    base = scenes.cornell_box(env=True)
    s = scenes.replace(base, lights=base.lights + scenes.cornell_area().lights)
    film, ref, _, _ = _render_pair(s, 64, 64, 3, 2, 3)
    _film_check(film, ref, 1e-4)


@pytest.mark.parametrize("kind", [scenes.GLOSSY_GGX, scenes.GLOSSY_BECKMANN])
def test_glossy_film_cornell(kind):
    s = scenes.with_glossy(scenes.cornell_box(env=True), kind)
    film, ref, _, _ = _render_pair(s, 64, 64, 4, 1, 3)
    assert np.isfinite(ref).all()
    _film_check(film, ref, 5e-4, frac_tol=5e-3)


@pytest.mark.parametrize("kind", [scenes.GLOSSY_GGX, scenes.GLOSSY_BECKMANN])
def test_glossy_under_area_light(kind):
    s = scenes.cornell_area(glossy=kind)
    film, ref, _, _ = _render_pair(s, 64, 64, 3, 2, 3)
    _film_check(film, ref, 5e-4, frac_tol=5e-3)


def test_glossy_on_some_models_textured_atrium():
    s = scenes.with_textures(scenes.sponza_class(detail=0.3, env_size=(128, 64)), uv_scale=6.0, every=2)
    s = scenes.with_glossy(s, scenes.GLOSSY_GGX, every=3)
    film, ref, _, _ = _render_pair(s, 160, 90, 3, 2, 2)
    _film_check(film, ref, 3e-2, frac_tol=1e-2)


def test_zero_glossy_flag_is_bitwise_the_plain_path(cornell_env):
    a = Context(cornell_env, 48, 48, max_depth=3, seed=4)
    b = Context(scenes.with_glossy(cornell_env, scenes.GLOSSY_NONE), 48, 48, max_depth=3, seed=4)
    for c in (a, b):
        c.set_camera(cornell_env.camera)
        c.render(2)
    assert np.array_equal(a.read_film(), b.read_film())
    a.close(); b.close()


def test_n4_argument_checks(cornell):
    bad = scenes.replace(cornell, material_glossy=np.full(cornell.n_models, 7, np.int32))
    with pytest.raises(RtError) as e:
        Context(bad, 16, 16)
    assert e.value.code == -1 and "glossy" in str(e.value)
    z = np.zeros((3, 3), np.float32)
    bad = scenes.replace(cornell, lights=[scenes.Light(scenes.LIGHT_AREA, tri=z)])
    with pytest.raises(RtError) as e:
        Context(bad, 16, 16)
    assert e.value.code == -1 and "degenerate" in str(e.value)
    bad = scenes.replace(cornell, lights=[scenes.Light(1)])        # directional lights are not implemented anywhere
    with pytest.raises(RtError) as e:
        Context(bad, 16, 16)
    assert e.value.code == -1
