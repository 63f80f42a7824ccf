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


def gather_film(ctx: Context, group=None) -> np.ndarray:
    """Gathers the tile-sharded film of a multi-rank render."""
    _, _, n_floats = ctx.film_slab_info()
    dev = torch.device("cuda", torch.cuda.current_device())
    local = torch.empty(n_floats, dtype=torch.float32, device=dev)
    if n_floats:
        ctx.copy_film_slab(local.data_ptr(), n_floats)
    return gather_slabs(local, ctx.width, ctx.height, ctx.tile_size, group)
