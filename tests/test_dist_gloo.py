"""The multi-rank film path on CPU: world_size 2 and 3 over gloo.  Each rank
packs the slab of its own interleaved tiles (what a GPU rank's film buffer
holds) of a committed golden film (tests/golden/films_shared_math.npz: films
rendered by the oracle), the slabs are all-gathered and unpacked, and the
result must equal the golden film on every rank."""
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


GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "films_shared_math.npz")


def _worker(rank, world, port, name, tile, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from raytracer_amd import api
        from raytracer_amd.dist import gather_slabs
        film = np.ascontiguousarray(np.load(GOLDEN)[name])  # what a 1-rank render produces
        h, w = film.shape[:2]
        local = torch.from_numpy(api.pack_slab(film, tile, rank, world))
        assert local.numel() == api.slab_floats(w, h, tile, rank, world)
        out = gather_slabs(local, w, h, tile)
        q.put((rank, bool(np.array_equal(out, film))))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,name,tile", [(2, "atrium_d02_48x27", 8), (3, "cornell_tex_30x26", 8), (2, "cornell_env_32", 16),
                                             (2, "cornell_area_ggx_24", 64)])
def test_tile_sharded_film_gather(world, name, tile):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, name, tile, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert sorted(res) == [(r, True) for r in range(world)]
