"""Film gather across ranks (one process per GPU, torch.distributed).

The film is sharded by interleaved tiles (tile t belongs to rank t % world,
the GPU form of the reference's tile pool, src/cpu/Integrator.cpp:33-49).
Rendering needs no exchange; once per read-out every rank's packed tile slab
is all-gathered (RCCL over xGMI with the "nccl" backend, gloo on CPU) and
unpacked into the full film.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist

from .api import Context, slab_floats, unpack_slabs


def gather_slabs(local_slab: torch.Tensor, width: int, height: int, tile_size: int, group=None) -> np.ndarray:
    """All-gathers every rank's slab (padded to the largest) and returns the
    assembled (H, W, 4) film on every rank."""
    world = dist.get_world_size(group)
    per_rank = max(slab_floats(width, height, tile_size, r, world) for r in range(world))
    padded = torch.zeros(per_rank, dtype=torch.float32, device=local_slab.device)
    padded[: local_slab.numel()] = local_slab.reshape(-1)
    gathered = torch.empty(world * per_rank, dtype=torch.float32, device=local_slab.device)
    dist.all_gather_into_tensor(gathered, padded, group=group)
    return unpack_slabs(width, height, tile_size, world, gathered.cpu().numpy(), per_rank)


def _buffers(ctx: Context, world: int, per_rank: int, dev: torch.device):
    """The three device buffers of a read-out (own slab, gathered slabs, film), kept on the context between calls."""
    key = (world, per_rank, dev)
    cache = getattr(ctx, "_gather_buffers", None)
    if cache is None or cache[0] != key:
        local = torch.zeros(per_rank, dtype=torch.float32, device=dev)   # the padding behind the slab stays zero
        gathered = torch.empty(world * per_rank, dtype=torch.float32, device=dev)
        film = torch.empty((ctx.height, ctx.width, 4), dtype=torch.float32, device=dev)
        cache = (key, local, gathered, film)
        ctx._gather_buffers = cache
    return cache[1:]


def gather_film_device(ctx: Context, group=None) -> torch.Tensor:
    """Gathers the tile-sharded film of a multi-rank render and assembles it on the device: every rank ends up with
    the (H, W, 4) film as a CUDA tensor (owned by the context and overwritten by the next call).  One all-gather of
    the padded slabs, one scatter kernel, no host copy.

    The library works on a stream of its own that has no implicit ordering with torch's streams, so every hand-over
    between the two is an explicit wait: torch's stream is drained before the library writes into a torch buffer,
    and the library calls return only when their work is done (rt_copy_film_slab, rt_unpack_slabs_dev)."""
    world = dist.get_world_size(group)
    _, _, n_floats = ctx.film_slab_info()
    dev = torch.device("cuda", torch.cuda.current_device())
    per_rank = max(slab_floats(ctx.width, ctx.height, ctx.tile_size, r, world) for r in range(world))
    local, gathered, film = _buffers(ctx, world, per_rank, dev)
    stream = torch.cuda.current_stream(dev)
    stream.synchronize()          # the fill of `local` and earlier readers of the buffers are done
    if n_floats:
        ctx.copy_film_slab(local.data_ptr(), n_floats)   # returns after the copy
    dist.all_gather_into_tensor(gathered, local, group=group)
    stream.synchronize()          # the gather is done before the library's stream reads it
    ctx.unpack_slabs_dev(world, gathered.data_ptr(), per_rank, film.data_ptr())   # returns after the scatter
    return film


def gather_film(ctx: Context, group=None) -> np.ndarray:
    """Gathers the tile-sharded film of a multi-rank render into host memory (what the reference's
    `Integrator::sync()` hands to the presenter, src/cpu/Integrator.cpp:21-24).  The array is a pinned buffer kept on
    the context: the next call on the same context overwrites it."""
    film = gather_film_device(ctx, group)
    host = getattr(ctx, "_gather_host", None)
    if host is None or host.shape != film.shape:
        host = torch.empty(film.shape, dtype=torch.float32, pin_memory=True)   # kept: pinning 133 MB takes milliseconds
        ctx._gather_host = host
    host.copy_(film)
    return host.numpy()
