"""GPU parity for textured materials (Material::computeBSDF with a diffuse
map, src/scene/Material.cpp:195-225, fed by the quad UV differences of
src/cpu/PathTracer.cpp:83-90): the UV differences are float subtractions and
products, so bit-exact; the film within the tolerance of the other film tests
(log2f/libm and the transcendentals of the light sampling differ by ulps)."""
import numpy as np
import pytest

from raytracer_amd import Context, scenes
from raytracer_amd.api import RtError
from oracle.binding import OracleScene
from tests.test_gpu_parity import _film_check

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def cornell_tex():
    return scenes.with_textures(scenes.cornell_box(env=True), uv_scale=3.0)


@pytest.mark.parametrize("W,H,tile", [(30, 26, 16), (33, 17, 8), (64, 64, 64), (130, 70, 64)])
def test_quad_uv_differences_bit_exact(cornell_tex, W, H, tile):
    s = cornell_tex
    o = OracleScene(s)
    ref = np.zeros((H, W, 4), np.float32)
    o.render(s.camera, W, H, 1, 0, 1, threads=2, tile_size=tile, duv=ref)
    ctx = Context(s, W, H, max_depth=1, tile_size=tile)
    ctx.set_camera(s.camera)
    ctx.render(1)
    got = ctx.read_primary_duv()
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))
    assert np.abs(ref).max() > 0
    ctx.close()


def test_textured_first_hit_film(cornell_tex):
    """max_depth 1: the film is the direct light of the trilinearly filtered primary hit."""
    s = cornell_tex
    o = OracleScene(s)
    ref, _ = o.render(s.camera, 96, 80, 1, 3, 2, threads=4, tile_size=16)
    ctx = Context(s, 96, 80, max_depth=1, seed=3, tile_size=16)
    ctx.set_camera(s.camera)
    ctx.render(2)
    _film_check(ctx.read_film(), ref, 2e-5, frac_tol=1e-3)
    ctx.close()


def test_textured_film_depth4(cornell_tex):
    s = cornell_tex
    o = OracleScene(s)
    ref, cnt = o.render(s.camera, 64, 64, 4, 0, 3, threads=4)
    ctx = Context(s, 64, 64, max_depth=4, seed=0)
    ctx.set_camera(s.camera)
    ctx.render(3)
    _film_check(ctx.read_film(), ref, 1e-4)
    ctx.close()


def test_textured_atrium_film():
    s = scenes.with_textures(scenes.sponza_class(detail=0.3, env_size=(128, 64)), uv_scale=6.0, every=2)
    o = OracleScene(s)
    W, H = 160, 90
    ref, _ = o.render(s.camera, W, H, 3, 2, 2, threads=8)
    refd = np.zeros((H, W, 4), np.float32)
    o.render(s.camera, W, H, 1, 2, 1, threads=8, duv=refd)
    ctx = Context(s, W, H, max_depth=3, seed=2)
    ctx.set_camera(s.camera)
    ctx.render(2)
    assert np.array_equal(ctx.read_primary_duv().view(np.uint32), refd.view(np.uint32))
    _film_check(ctx.read_film(), ref, 2e-2, frac_tol=5e-3)
    ctx.close()


def test_textured_shards_equal_whole(cornell_tex):
    """Tiles are the unit of the quad bookkeeping, so sharding by tiles changes nothing."""
    s = cornell_tex
    whole = Context(s, 70, 50, max_depth=2, seed=1, tile_size=16)
    whole.set_camera(s.camera)
    whole.render(2)
    want = whole.read_film()
    got = np.zeros_like(want)
    for r in range(3):
        c = Context(s, 70, 50, max_depth=2, seed=1, tile_size=16, rank=r, world=3)
        c.set_camera(s.camera)
        c.render(2)
        part = c.read_film()
        assert not np.any((part != 0) & (got != 0))
        got += part
        c.close()
    assert np.array_equal(got, want)
    whole.close()


def test_texture_argument_checks(cornell):
    s = scenes.with_textures(cornell)
    bad = scenes.Scene(pos=s.pos, normal=s.normal, uv=s.uv, tangent=s.tangent, binormal=s.binormal, indices=s.indices,
                       tri_model=s.tri_model, materials=s.materials, lights=s.lights, camera=s.camera, textures=s.textures,
                       material_texture=np.full(s.n_models, 3, np.int32))
    with pytest.raises(RtError) as e:
        Context(bad, 16, 16)
    assert e.value.code == -1 and "texture" in str(e.value)
