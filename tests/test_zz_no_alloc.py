"""rt_trace_closest_noalloc (the entry point that allocates nothing) against rt_trace_closest_dev through the C-ABI:
ids, distances and barycentrics bit for bit, on two calls in a row, the second with fewer rays, on ray sets salted
with rays that go through the exact follow-up launch (so the scratch's deferred list is used as well as its work
counters); the size limit is an error, not a fault."""
import numpy as np
import pytest

from raytracer_amd import Context, api, scenes
from tests.test_gpu_fuzz import _rays, _soup

pytestmark = pytest.mark.gpu


def _trace(ctx, fn, rays):
    import torch
    n = rays.shape[1]
    d_rays = torch.from_numpy(np.ascontiguousarray(rays)).to("cuda:0")
    out_f = torch.full((3, n), -7.0, dtype=torch.float32, device="cuda:0")
    out_i = torch.full((2, n), -7, dtype=torch.int32, device="cuda:0")
    torch.cuda.synchronize()
    fn(n, d_rays.data_ptr(), out_f[0].data_ptr(), out_i[0].data_ptr(), out_i[1].data_ptr(), out_f[1].data_ptr(), out_f[2].data_ptr())
    ctx.synchronize()
    return out_f.cpu().numpy().view(np.uint32), out_i.cpu().numpy()


@pytest.mark.parametrize("wide", [False, True], ids=["binary_tree", "wide_tree"])
def test_noalloc_entry_point_matches_the_allocating_one(wide):
    """Both entry points walk the tree the context was made with (a ray's result does not depend on how the launch
    is scheduled, on either tree, so the comparison is bit for bit on the 4-wide tree too)."""
    rng = np.random.default_rng(42)
    scene = _soup(rng, 20000, "plain")
    ctx = Context(scene, 64, 64, wide_bvh=wide)
    big = np.concatenate([_rays(rng, scene, 300_000), scenes.camera_rays_numpy(scene.camera, 640, 360)], axis=1)
    small = np.ascontiguousarray(_rays(rng, scene, 70_001)[:, ::-1])
    for rays in (big, small):          # two calls in a row, the second smaller
        e0 = ctx.stats()["exact_closest_rays"]
        got_f, got_i = _trace(ctx, ctx.trace_closest_noalloc, rays)
        e1 = ctx.stats()["exact_closest_rays"]
        want_f, want_i = _trace(ctx, lambda *a: ctx.trace_closest_dev(*a, 1), rays)
        e2 = ctx.stats()["exact_closest_rays"]
        assert np.array_equal(got_i, want_i)
        assert np.array_equal(got_f, want_f)
        assert (want_i[1] >= 0).sum() > 1000
        assert e1 - e0 == e2 - e1 > 0      # the same rays took the exact follow-up launch on both paths
    # beyond the fixed scratch: a clear error and nothing launched
    lib = api.load_library()
    assert api.RT_NOALLOC_MAX_RAYS == 1 << 22
    rc = lib.rt_trace_closest_noalloc(ctx._h, api.RT_NOALLOC_MAX_RAYS + 1, 1, 1, 1, 1, 1, 1)
    assert rc == -4 and b"exceed" in lib.rt_last_error(ctx._h)
    ctx.close()


def test_noalloc_entry_point_on_a_second_context_after_the_first_is_gone(cornell_env):
    """The scratch belongs to the device, not to a context: a later context finds it in whatever state the last call
    left it and must still give the same hits."""
    rays = scenes.camera_rays_numpy(cornell_env.camera, 256, 256)
    for _ in range(2):
        ctx = Context(cornell_env, 32, 32)
        got_f, got_i = _trace(ctx, ctx.trace_closest_noalloc, rays)
        want_f, want_i = _trace(ctx, lambda *a: ctx.trace_closest_dev(*a, 1), rays)
        assert np.array_equal(got_i, want_i) and np.array_equal(got_f, want_f)
        ctx.close()
