"""Soak run of the tile sharding (pytest does not collect this file): seeded random film sizes, tile sizes, world sizes,
scenes and batches; every rank renders only its tiles (one context per emulated rank on this GPU), the slabs are laid out
the way the all-gather leaves them and scattered on the device; the film must equal the unsharded one bit for bit, and
the ranks together must trace exactly the unsharded render's rays.
usage (GPU box): python tests/soak_shards.py [first_seed] [n_seeds]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from raytracer_amd import Context, api, scenes  # noqa: E402


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    n_seeds = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    base = [scenes.cornell_box(env=True), scenes.with_textures(scenes.cornell_box()), scenes.cornell_area(),
            scenes.sponza_class(detail=0.3, env_size=(128, 64))]
    t0 = time.perf_counter()
    for seed in range(first, first + n_seeds):
        rng = np.random.default_rng(seed)
        s = base[rng.integers(0, len(base))]
        W, H = int(rng.integers(3, 400)), int(rng.integers(3, 300))
        tile = int(rng.choice([8, 16, 32, 64]))
        world = int(rng.integers(2, 10))
        depth, passes, batch = int(rng.integers(1, 7)), int(rng.integers(1, 5)), int(rng.choice([0, 1, 2, 4]))
        wide = bool(rng.integers(0, 2))
        kw = dict(max_depth=depth, seed=seed, tile_size=tile, batch=batch, wide_bvh=wide)
        whole = Context(s, W, H, **kw)
        whole.set_camera(s.camera)
        whole.render(passes)
        want = whole.read_film()
        per_rank = max(api.slab_floats(W, H, tile, r, world) for r in range(world))
        gathered = torch.zeros(world * max(per_rank, 1), dtype=torch.float32, device="cuda:0")
        film = torch.full((H, W, 4), -1.0, dtype=torch.float32, device="cuda:0")
        torch.cuda.synchronize()
        traced = 0
        for r in range(world):
            c = Context(s, W, H, rank=r, world=world, **kw)
            c.set_camera(s.camera)
            c.render(passes)
            _, n_tiles, nf = c.film_slab_info()
            assert nf == api.slab_floats(W, H, tile, r, world), (seed, r)
            if nf:
                c.copy_film_slab(gathered[r * per_rank:].data_ptr(), nf)
            st = c.stats()
            traced += st["closest_rays"] + st["shadow_rays"]
            c.close()
        whole.unpack_slabs_dev(world, gathered.data_ptr(), per_rank, film.data_ptr())
        got = film.cpu().numpy()
        same = (got.view(np.uint32) == want.view(np.uint32)) | (np.isnan(got) & np.isnan(want))
        desc = f"{s.name if hasattr(s, 'name') else ''} {W}x{H} tile {tile} world {world} depth {depth} passes {passes} batch {batch} {'wide' if wide else 'binary'}"
        assert same.all(), (seed, desc, int((~same).any(-1).sum()))
        st = whole.stats()
        assert traced == st["closest_rays"] + st["shadow_rays"], (seed, desc, traced, st["closest_rays"] + st["shadow_rays"])
        whole.close()
        print(f"seed {seed}: {desc}: sharded film equals the unsharded one  [{time.perf_counter() - t0:.0f} s]", flush=True)
    print("soak ok:", {"seeds": n_seeds})


if __name__ == "__main__":
    main()
