"""The oracle's two transcendental modes.  "libm" is what the reference calls; "shared" is the set of fixed
single-precision polynomials the HIP kernels restate, under which films compare bit for bit
(tests/test_gpu_exact_film.py).  Here: how accurate the shared functions are, their edge cases, and how far a
shared-math film is from a libm film (so that exact GPU parity against the former bounds the distance to the
latter)."""
import numpy as np
import pytest

from oracle import binding as ob
from oracle.binding import OracleScene


def _ulps(got, ref64):
    sp = np.spacing(np.abs(ref64).astype(np.float32)).astype(np.float64)
    return np.abs(got.astype(np.float64) - ref64) / sp


def test_sin_cos_accuracy_over_the_path_range():
    # concentricSampleDisk: theta in [-pi/4, 3pi/4]; EnvLight::sampleLi: theta in [0, pi], phi in [0, 2pi]
    x = np.linspace(-0.8, 6.4, 1_000_001).astype(np.float32)
    for name, f in (("sin", np.sin), ("cos", np.cos)):
        got, ref = ob.dm_eval(name, x), f(x.astype(np.float64))
        assert np.abs(got - ref).max() < 1.2e-7          # absolute, also near the zeros
        away = np.abs(ref) > 0.05
        assert _ulps(got[away], ref[away]).max() < 2.0   # relative away from them


def test_acos_atan2_log2_accuracy():
    x = np.linspace(-1, 1, 1_000_001).astype(np.float32)
    assert _ulps(ob.dm_eval("acos", x), np.arccos(x.astype(np.float64))).max() < 2.0
    rng = np.random.default_rng(0)
    a, b = rng.normal(size=500_000).astype(np.float32), rng.normal(size=500_000).astype(np.float32)
    got, ref = ob.dm_eval("atan2", a, b), np.arctan2(a.astype(np.float64), b.astype(np.float64))
    assert np.abs(got - ref).max() < 4e-7 and _ulps(got, ref).max() < 4.0
    r = np.exp(rng.uniform(0, 12, 500_000)).astype(np.float32)   # the lod argument is > 1
    assert np.abs(ob.dm_eval("log2", r) - np.log2(r.astype(np.float64))).max() < 2e-6


def test_log_exp_pow_accuracy():
    rng = np.random.default_rng(1)
    x = np.exp(rng.uniform(-80, 80, 500_000)).astype(np.float32)
    assert _ulps(ob.dm_eval("log", x), np.log(x.astype(np.float64))).max() < 1.0
    x = rng.uniform(-85, 88, 500_000).astype(np.float32)
    assert _ulps(ob.dm_eval("exp", x), np.exp(x.astype(np.float64))).max() < 1.5
    x, y = rng.uniform(1e-6, 1, 500_000).astype(np.float32), rng.uniform(0.5, 2, 500_000).astype(np.float32)
    ref = np.power(x.astype(np.float64), y.astype(np.float64))
    assert (np.abs(ob.dm_eval("pow", x, y) - ref) / ref).max() < 4e-6
    f = lambda *v: np.asarray(v, np.float32)
    out = ob.dm_eval("log", f(0, 1, -1, np.inf, np.nan, 1e-40))
    assert out[0] == -np.inf and out[1] == 0 and np.isnan(out[2]) and out[3] == np.inf and np.isnan(out[4])
    assert abs(out[5] - np.log(1e-40)) < 1e-4                                     # denormal argument
    out = ob.dm_eval("exp", f(0, -np.inf, np.inf, np.nan, -104, 89))
    assert out[0] == 1 and out[1] == 0 and out[2] == np.inf and np.isnan(out[3]) and out[4] == 0 and out[5] == np.inf
    assert ob.dm_eval("exp", f(-100))[0] > 0                                       # denormal result, two-step scaling


def test_edge_cases():
    f = lambda *v: np.asarray(v, np.float32)
    pi = np.float32(np.pi)
    # atan2(y, x): axes, quadrants, both zero (defined as 0 whatever the signs)
    got = ob.dm_eval("atan2", f(0, 0, 1, -1, 0, -0.0, 1, -1), f(1, -1, 0, 0, 0, -0.0, -1, -1))
    assert np.allclose(got, [0, pi, pi / 2, -pi / 2, 0, 0, 3 * pi / 4, -3 * pi / 4], atol=3e-7)
    assert np.array_equal(ob.dm_eval("acos", f(1, -1, 0)), f(0, np.pi, np.pi / 2))
    assert np.isnan(ob.dm_eval("acos", f(np.nan))).all()
    out = ob.dm_eval("sin", f(np.inf, np.nan, 1e5, 0, -0.0))
    assert np.isnan(out[:3]).all() and out[3] == 0 and out[4] == 0
    assert np.array_equal(ob.dm_eval("cos", f(0.0)), f(1.0)) and np.array_equal(ob.dm_eval("log2", f(1, 2, 4, 1024)), f(0, 1, 2, 10))
    assert ob.dm_eval("log2", f(np.inf))[0] == 128.0


@pytest.fixture
def libm_after():
    yield
    ob.set_math("libm")


def test_shared_and_libm_films_agree(cornell_env, atrium_small, libm_after):
    assert ob.get_math() == "libm"
    for scene, W, H, depth in ((cornell_env, 64, 64, 4), (atrium_small, 96, 64, 4)):
        orc = OracleScene(scene)
        ob.set_math("libm")
        a, _ = orc.render(scene.camera, W, H, depth, 3, 2, threads=4)
        ob.set_math("shared")
        b, _ = orc.render(scene.camera, W, H, depth, 3, 2, threads=4)
        assert not np.array_equal(a, b)      # the switch does something
        err = np.abs(a[..., :3].astype(np.float64) - b[..., :3])
        rel_rmse = np.sqrt((err ** 2).mean()) / np.sqrt((a[..., :3].astype(np.float64) ** 2).mean())
        # an ulp in a sampled direction can flip a shadow ray or a bounce at a silhouette: a few pixels move
        # by a visible amount, the rest by 1e-6
        frac_off = float(((err / (np.abs(a[..., :3]) + 1e-3)).max(-1) > 1e-3).mean())
        assert rel_rmse < 1e-3 and frac_off < 5e-3, (rel_rmse, frac_off)   # measured: 3e-7 and 0
