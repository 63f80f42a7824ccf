"""Soak run of the fuzz parity checks (pytest does not collect this file; it runs the CPU oracle as checker):
many seeded triangle soups of random size and kind, each traced on the binary tree (ids, distances and barycentrics bit
for bit, any-hit status equal, tree node for node) and on the 4-wide tree (any-hit status equal; closest hits classified
against the oracle: every difference must be an exact tie or a near tie with a genuine triangle).
usage (GPU box): python tests/soak_fuzz.py [first_seed] [n_seeds] [rays_per_seed]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import classify_against_oracle  # noqa: E402
from oracle.binding import OracleScene  # noqa: E402
from raytracer_amd import Context  # noqa: E402
from tests.test_gpu_fuzz import _rays, _soup  # noqa: E402

KINDS = ("plain", "clustered", "axis", "dup", "degenerate", "sliver")


def _sliver(rng, n_tri):
    """Thin triangles stacked along x, all crossing one corridor: a deep tree and long stacks (the wide kernel's LDS stack
    spills); returned as a soup in _soup's form."""
    s = _soup(rng, n_tri, "plain")
    x = np.sort(rng.uniform(0, 100, n_tri)).astype(np.float32)
    v0 = np.stack([x, rng.uniform(-1, -0.5, n_tri), rng.uniform(-1, 1, n_tri)], 1)
    v1 = np.stack([x + 0.01, rng.uniform(0.5, 1, n_tri), rng.uniform(-1, 1, n_tri)], 1)
    v2 = np.stack([x + 0.02, rng.uniform(-1, 1, n_tri), rng.uniform(1.5, 2, n_tri)], 1)
    s.pos[:] = np.stack([v0, v1, v2], 1).reshape(-1, 3).astype(np.float32)
    return s


def _corridor_rays(rng, m):
    org = np.stack([rng.uniform(-5, 105, m), rng.uniform(-0.4, 0.4, m), rng.uniform(-0.5, 1.2, m)]).astype(np.float32)
    d = np.stack([rng.choice([-1.0, 1.0], m), rng.normal(0, 0.01, m), rng.normal(0, 0.01, m)]).astype(np.float32)
    org[1, ::2] = 0.95; org[2, ::2] = -0.95      # every other ray passes under the triangles: hits nothing, stacks everything
    return np.ascontiguousarray(np.concatenate([org, d], 0))
SIZES = (1, 2, 3, 4, 5, 7, 8, 9, 16, 17, 33, 64, 100, 257, 1000, 4097, 20000, 60000)


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    n_seeds = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    n_rays = int(sys.argv[3]) if len(sys.argv) > 3 else 40000
    threads = max(1, min(16, len(os.sched_getaffinity(0))))
    tot = {"seeds": 0, "rays": 0, "wide_exact_ties": 0, "wide_near_ties": 0}
    t0 = time.perf_counter()
    for seed in range(first, first + n_seeds):
        rng = np.random.default_rng(seed)
        n_tri = int(SIZES[rng.integers(0, len(SIZES))])
        kind = KINDS[rng.integers(0, len(KINDS))]
        s = _sliver(rng, n_tri) if kind == "sliver" else _soup(rng, n_tri, kind)
        rays = _rays(rng, s, n_rays)
        if kind == "sliver":
            rays = np.ascontiguousarray(np.concatenate([rays[:, : n_rays // 2], _corridor_rays(rng, n_rays - n_rays // 2)], axis=1))
        orc = OracleScene(s)
        rt, rid0, rid1, ru, rv = orc.intersect(rays)
        rocc = orc.occluded(rays)
        ctx = Context(s, 32, 32, wide_bvh=False)
        for g, r in zip(ctx.get_bvh(), orc.bvh_nodes()):
            assert np.array_equal(g, r), (seed, "tree")
        t, id0, id1, u, v = ctx.trace_closest(rays)
        assert np.array_equal(id1, rid1) and np.array_equal(id0, rid0), (seed, n_tri, kind, "binary ids")
        for a, b in ((t, rt), (u, ru), (v, rv)):
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), (seed, n_tri, kind, "binary t/u/v")
        assert np.array_equal(ctx.trace_occluded(rays), rocc), (seed, n_tri, kind, "binary any-hit")
        ctx.close()
        ctx = Context(s, 32, 32, wide_bvh=True)
        t, id0, id1, u, v = ctx.trace_closest(rays)
        assert np.array_equal(ctx.trace_occluded(rays), rocc), (seed, n_tri, kind, "wide any-hit")
        c = classify_against_oracle(orc, rays, t, id1, rt, rid1, threads)
        assert c["unexplained"] == 0 and c["id_mismatches_outside_ties"] == 0, (seed, n_tri, kind, c)
        same = (id1 == rid1) & (t.view(np.uint32) == rt.view(np.uint32))
        for a, b in ((u, ru), (v, rv), (id0, rid0)):   # where the same triangle was kept, everything about the hit is equal
            assert np.array_equal(a.view(np.uint32)[same], b.view(np.uint32)[same]), (seed, n_tri, kind, "wide u/v/id0")
        ctx.close()
        tot["seeds"] += 1
        tot["rays"] += rays.shape[1]
        tot["wide_exact_ties"] += c["id_mismatches_on_exact_ties"]
        tot["wide_near_ties"] += c["near_ties"]
        print(f"seed {seed}: {n_tri} triangles ({kind}), hits {(rid1 >= 0).mean():.3f}, wide exact ties {c['id_mismatches_on_exact_ties']}, "
              f"near ties {c['near_ties']}  [{time.perf_counter() - t0:.0f} s]", flush=True)
    print("soak ok:", tot)


if __name__ == "__main__":
    main()
