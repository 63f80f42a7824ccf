"""The device side of the multi-rank path on one GPU: the film slab is copied
device to device, all-gathered over RCCL (backend "nccl", world size 1 here;
worlds of 2 and 3 run over gloo in test_dist_gloo.py) and unpacked; ranks of a
larger world are emulated by separate contexts on the same GPU."""
import os
import socket

import numpy as np
import pytest

from raytracer_amd import Context, api, scenes

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_gather_film_over_rccl(cornell_env):
    import torch
    import torch.distributed as dist
    from raytracer_amd.dist import gather_film
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    try:
        ctx = Context(cornell_env, 100, 70, max_depth=3, seed=2, tile_size=32)
        ctx.set_camera(cornell_env.camera)
        ctx.render(3)
        got = gather_film(ctx)
        assert np.array_equal(got, ctx.read_film())
        ctx.close()
    finally:
        dist.destroy_process_group()


def test_rank_slabs_assemble_on_device(cornell_env):
    """Every rank's device slab, copied out the way gather_film does and
    concatenated the way all_gather does, unpacks to the unsharded film."""
    import torch
    s, W, H, tile, world = cornell_env, 150, 90, 32, 4
    whole = Context(s, W, H, max_depth=3, seed=5, tile_size=tile)
    whole.set_camera(s.camera)
    whole.render(2)
    want = whole.read_film()
    whole.close()
    per_rank = max(api.slab_floats(W, H, tile, r, world) for r in range(world))
    gathered = torch.zeros(world * per_rank, dtype=torch.float32, device="cuda:0")
    for r in range(world):
        c = Context(s, W, H, max_depth=3, seed=5, tile_size=tile, rank=r, world=world)
        c.set_camera(s.camera)
        c.render(2)
        _, _, nf = c.film_slab_info()
        assert nf == api.slab_floats(W, H, tile, r, world)
        c.copy_film_slab(gathered[r * per_rank:].data_ptr(), nf)
        c.close()
    got = api.unpack_slabs(W, H, tile, world, gathered.cpu().numpy(), per_rank)
    assert np.array_equal(got, want)
    # the same scatter on the device (what gather_film_device runs after the all-gather)
    c = Context(s, W, H, max_depth=3, seed=5, tile_size=tile)
    film = torch.full((H, W, 4), -1.0, dtype=torch.float32, device="cuda:0")
    c.unpack_slabs_dev(world, gathered.data_ptr(), per_rank, film.data_ptr())
    assert np.array_equal(film.cpu().numpy(), want)
    with pytest.raises(api.RtError):
        c.unpack_slabs_dev(world, gathered.data_ptr(), per_rank // 2, film.data_ptr())
    c.close()


def test_any_hit_agrees_with_closest_hit_at_full_size():
    """Size-independent property: an unbounded shadow ray is occluded exactly
    when the closest-hit query of the same ray finds something (both walk the
    boxes the ray pierces; culling only ever drops farther hits)."""
    s = scenes.sponza_class()
    ctx = Context(s, 64, 64)
    n = 2073600
    for rays in (scenes.incoherent_rays(s, n, seed=77), scenes.camera_rays_numpy(s.camera, 1920, 1080)):
        occ = ctx.trace_occluded(rays)
        t, id0, id1, u, v = ctx.trace_closest(rays)
        assert np.array_equal(occ.astype(bool), id1 >= 0)
        # a second query returns the same bits
        t2, _, id1b, u2, v2 = ctx.trace_closest(rays)
        assert np.array_equal(id1, id1b) and np.array_equal(t.view(np.uint32), t2.view(np.uint32))
        hit = id1 >= 0
        assert np.all(t[hit] > 1e-4) and np.all(np.isinf(t[~hit]))
        assert np.all((u[hit] >= 0) & (v[hit] >= 0) & (u[hit] + v[hit] <= 1.0 + 1e-6))
    ctx.close()
