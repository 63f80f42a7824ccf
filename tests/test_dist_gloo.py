"""The multi-rank film path on CPU: world_size 2 and 3 over gloo.  Each rank
packs the slab of its own interleaved tiles (what a GPU rank's film buffer
holds), the slabs are all-gathered and unpacked, and the result must equal
the full film on every rank."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, w, h, tile, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from raytracer_amd import api
        from raytracer_amd.dist import gather_slabs
        rng = np.random.default_rng(1234)
        film = rng.random((h, w, 4), np.float32)  # what a 1-rank render would produce
        local = torch.from_numpy(api.pack_slab(film, tile, rank, world))
        assert local.numel() == api.slab_floats(w, h, tile, rank, world)
        out = gather_slabs(local, w, h, tile)
        q.put((rank, bool(np.array_equal(out, film))))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,w,h,tile", [(2, 200, 136, 64), (3, 100, 70, 32), (2, 64, 64, 64)])
def test_tile_sharded_film_gather(world, w, h, tile):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, w, h, tile, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert sorted(res) == [(r, True) for r in range(world)]
