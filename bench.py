"""Headline benchmark: closest-hit BVH traversal on the 1080p Sponza-class
scene, coherent (pinhole camera) + randomised incoherent + bounce ray sets,
inputs resident in HBM (BASELINE.json configs[1]).

One step = ONE launch of the closest-hit kernel over the three ray sets
(3 x 2,073,600 rays); the K timed steps are enqueued back to back.  With N > 1 every rank traces ray sets of that size on
its own GPU (scene replicated, no data-path collective): weak scaling.
A second, untimed-for-`value` leg renders sample passes of the path tracer at
1920x1080 and reports samples/s beside the headline.

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak
# fp32 vector peak of this path in lane-operations/s: 256 CUs x 4 SIMDs x 32 lanes per cycle x 2.4 GHz = 78.6 T.  A
# wave64 VALU instruction issues over 2 cycles on a SIMD-32 (tools/ubench/valu_ubench.hip measures 2.3-2.5 cycles at
# the nominal clock with 8 waves per SIMD: profiles/r02/valu_ubench.json).  An FMA would count twice (157.3 TFLOP/s
# is the datasheet figure), but the path may not contract a*b+c, so every multiply and add is an instruction of its
# own and one lane-operation per lane and issue slot is the ceiling.
VALU_PEAK_TOPS = 256 * 4 * 32 * 2.4e9 / 1e12


def algorithmic_bytes(n_rays: int, n_inner: int, n_tri: int) -> float:
    """Bytes the reference's binary-tree algorithm must move for a ray set:
    24 B ray in + 20 B hit out per ray, 56 B per inner-node visit (two child
    boxes + two links), 36 B per triangle test (three positions)."""
    return 44.0 * n_rays + 56.0 * n_inner + 36.0 * n_tri


def algorithmic_valu_ops(n_rays: int, n_inner: int, n_tri: int) -> float:
    """fp32 lane-operations of the same algorithm: two slab tests (27 each) per inner-node visit, one
    Moller-Trumbore test (57, the division counted as 10) per triangle test, 12 per ray for 1/dir."""
    return 12.0 * n_rays + 54.0 * n_inner + 57.0 * n_tri


def algorithmic_valu_cycles(n_rays: int, n_inner: int, n_tri: int) -> float:
    """SIMD cycles the same lane-operations take as full 64-lane wave instructions at the issue rates measured on
    this chip (tools/ubench/inst_ubench, profiles/r03/inst_ubench.json: v_add/sub/mul_f32 2.5 cycles per wave
    instruction, min/max/compare/select 4.2).  Per inner visit 24 subtract/multiply + 30 min/max/compare; per
    triangle test 40 + 17; the 12 per ray (three divisions) at 4.2."""
    return ((24.0 * 2.5 + 30.0 * 4.2) * n_inner + (40.0 * 2.5 + 17.0 * 4.2) * n_tri + 12.0 * 4.2 * n_rays) / 64.0


def classify_against_oracle(orc, rays, gpu_t, gpu_id1, ref_t, ref_id1, threads):
    """How a closest-hit result differs from the oracle's, by class: same distance and another id on a ray where two
    or more reachable triangles lie at exactly that distance (an exact tie: the strict `<` of src/scene/BVH.cpp:139
    keeps the first visited); another distance within 4 ulp whose triangle the reference's own test places exactly
    there (a near tie: the triangle found first culls the other's box); anything else."""
    bad_t = gpu_t.view(np.uint32) != ref_t.view(np.uint32)
    bad_id = gpu_id1 != ref_id1
    out = {"id_mismatches": int(bad_id.sum()), "dist_bit_mismatches": int(bad_t.sum()), "id_mismatches_on_exact_ties": 0,
           "near_ties": 0, "id_mismatches_outside_ties": 0, "unexplained": 0}
    idx = np.nonzero(bad_t | bad_id)[0]
    if idx.size == 0:
        return out
    sub = np.ascontiguousarray(rays[:, idx])
    tied = orc.intersect_ties(sub, threads=threads).astype(bool)
    ct, _, _, ok = orc.check_candidates(sub, gpu_id1[idx])
    genuine = (ok == 3) & (ct.view(np.uint32) == gpu_t[idx].view(np.uint32))
    ulps = np.abs(gpu_t[idx].view(np.int32).astype(np.int64) - ref_t[idx].view(np.int32).astype(np.int64))
    exact = ~bad_t[idx] & tied & genuine
    near = bad_t[idx] & (ulps <= 4) & genuine
    out["id_mismatches_on_exact_ties"] = int(exact.sum())
    out["near_ties"] = int(near.sum())
    out["id_mismatches_outside_ties"] = int((bad_id[idx] & ~exact & ~near).sum())
    out["unexplained"] = int((~exact & ~near).sum())
    return out


def strong_scaling_fields(world: int, sharded) -> dict:
    """Top-level keys of an N > 1 line beside `value` (weak scaling: every GPU traces ray sets of its own): the
    tile-sharded 4K film of the same run, gather included, is the strong-scaling number."""
    if world <= 1 or not sharded:
        return {}
    return {"value_strong": sharded["samples_per_sec"], "unit_strong": "samples/s", "samples_per_sec_strong": sharded["samples_per_sec"],
            "scaling_of": {"value": "weak", "value_strong": "strong"},
            "strong_workload": sharded["workload"]}


def cpu_model() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def launch_ranks(n_gpus: int, argv) -> int:
    """Starts one rank per GPU as fresh child processes (torch.distributed.run over 127.0.0.1) and returns their
    exit code.  Nothing here initialises a GPU: counting devices does not."""
    import socket
    import subprocess
    import torch
    have = torch.cuda.device_count()
    if have < n_gpus:
        print(f"bench.py: --gpus {n_gpus} asked for but only {have} GPU(s) are visible", file=sys.stderr)
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--settle-steps", type=int, default=200,
                    help="untimed launches before the warm-up steps: the launch time of a fresh process settles only after "
                         "about 0.2 s of back-to-back launches, whatever --warmup says")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--no-set-diagnostics", action="store_true",
                    help="skip the per-set launches made after the timed region (keeps a profile to the timed launches)")
    ap.add_argument("--render-passes", type=int, default=32, help="sample passes of the path-tracing leg (0 = skip)")
    ap.add_argument("--render-depth", type=int, default=8)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=0)
    ap.add_argument("--bvh", choices=("binary", "wide"), default="wide",
                    help="tree the closest-hit launch walks: the reference's binary LBVH (bit-exact ids) or its 4-wide collapse "
                         "(same distances; ids may differ where several triangles lie at exactly the same distance)")
    args = ap.parse_args()
    if args.gpus < 1:
        sys.exit("bench.py: --gpus must be at least 1")
    if args.gpus > 1 and "RANK" not in os.environ:
        # started as a plain script: become the launcher of N rank processes (before anything touches a GPU)
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))

    # stdout carries the ONE JSON line and nothing else: whatever libraries print there (RCCL's version banner)
    # goes to stderr instead
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    use_dist = world > 1 or "RANK" in os.environ  # launched by torch.distributed.run
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s)")
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))
    torch.cuda.set_device(local_rank)
    dev = torch.device(f"cuda:{local_rank}")

    from raytracer_amd import Context, scenes

    W, H = args.width, args.height
    scene = scenes.sponza_class()
    ctx = Context(scene, W, H, max_depth=args.render_depth, seed=0, device=local_rank, wide_bvh=(args.bvh == "wide"))
    # the other tree: timed after the headline for comparison; the path-tracing leg runs on the binary tree (exact films)
    other_name = "binary" if args.bvh == "wide" else "wide"
    compare = not args.no_set_diagnostics
    xctx = None
    if compare or (args.bvh == "wide" and args.render_passes > 0):
        xctx = Context(scene, W, H, max_depth=args.render_depth, seed=0, device=local_rank, wide_bvh=(other_name == "wide"))
    bctx = ctx if args.bvh == "binary" else xctx     # binary tree
    wctx = ctx if args.bvh == "wide" else xctx       # 4-wide tree (None when not built)
    n = W * H
    coherent = scenes.camera_rays_numpy(scene.camera, W, H)
    incoherent = scenes.incoherent_rays(scene, n, seed=0x5EED + rank)
    # the rays a path tracer sends after the primary hit, captured from depth-2 passes of the tracer itself
    bounce = scenes.bounce_rays(scene, W, H, n, seed=0x5EED + rank, device=local_rank)
    sets = {"coherent": coherent, "incoherent": incoherent, "bounce": bounce}
    NS = len(sets)
    # one batch = the three sets, resident in HBM as one SoA buffer of 3n rays: a step is ONE launch over it
    # (a launch costs about one worst-case ray latency, ~0.2 ms, whatever its size)
    both = np.ascontiguousarray(np.concatenate([sets[k] for k in sets], axis=1))
    d_both = torch.from_numpy(both).to(dev)
    d_rays = {k: torch.from_numpy(v).to(dev) for k, v in sets.items()}
    out_f = torch.empty((3, NS * n), dtype=torch.float32, device=dev)
    out_i = torch.empty((2, NS * n), dtype=torch.int32, device=dev)
    torch.cuda.synchronize()

    def run_steps(k):
        """k steps = k back-to-back launches on the library's stream, enqueued by one call (no host round trip between
        steps); the library brackets them with HIP events on that stream and keeps the mean launch duration."""
        ctx.trace_closest_dev(NS * n, d_both.data_ptr(), out_f[0].data_ptr(), out_i[0].data_ptr(), out_i[1].data_ptr(),
                              out_f[1].data_ptr(), out_f[2].data_ptr(), k)
        return ctx.last_trace_ms()

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        ctx.synchronize()

    if args.settle_steps > 0:
        run_steps(args.settle_steps)
    if args.warmup > 0:
        run_steps(args.warmup)
    barrier()
    t0 = time.perf_counter()
    kernel_ms = [run_steps(args.steps)] if args.steps > 0 else [float("nan")]
    barrier()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    rays_per_step = NS * n * world
    mrays = rays_per_step * args.steps / elapsed / 1e6

    # hit ids and distances of the last timed launch for the full-size parity check below
    gpu_id1 = out_i[1].cpu().numpy().copy()
    gpu_t = out_f[0].cpu().numpy().copy()

    # diagnostics outside the timed region: each set launched by itself
    set_ms = {k: float("nan") for k in sets}
    for k in () if args.no_set_diagnostics else tuple(sets):
        for reps in (5, max(args.steps // 4, 5)):   # a few untimed launches, then the timed ones
            ctx.trace_closest_dev(n, d_rays[k].data_ptr(), out_f[0].data_ptr(), out_i[0].data_ptr(), out_i[1].data_ptr(),
                                  out_f[1].data_ptr(), out_f[2].data_ptr(), reps)
        set_ms[k] = ctx.last_trace_ms()

    # the same launch on the other tree (outside the timed region)
    other_ms, other_id1, other_t = float("nan"), None, None
    if compare and xctx is not None:
        xctx.trace_closest_dev(NS * n, d_both.data_ptr(), out_f[0].data_ptr(), out_i[0].data_ptr(), out_i[1].data_ptr(),
                               out_f[1].data_ptr(), out_f[2].data_ptr(), max(args.warmup + args.settle_steps, 1))
        xctx.trace_closest_dev(NS * n, d_both.data_ptr(), out_f[0].data_ptr(), out_i[0].data_ptr(), out_i[1].data_ptr(),
                               out_f[1].data_ptr(), out_f[2].data_ptr(), max(args.steps, 1))
        other_ms = xctx.last_trace_ms()
        other_id1 = out_i[1].cpu().numpy().copy()
        other_t = out_f[0].cpu().numpy().copy()

    # ---- path-tracing leg: samples/s at 1920x1080 (BASELINE.json configs[2] shape)
    render = None
    if args.render_passes > 0:
        rctx = bctx
        rctx.set_camera(scene.camera)
        rctx.render(args.render_passes)  # warm-up; fills the per-camera cache (primary hits, point-light visibility)
        rctx.reset_accumulation()
        s0 = rctx.stats()
        barrier()
        rctx.synchronize()
        r0 = time.perf_counter()
        rctx.render(args.render_passes)
        rctx.synchronize()
        barrier()
        r_el = time.perf_counter() - r0
        if use_dist:
            t = torch.tensor([r_el], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            r_el = float(t.item())
        s1 = rctx.stats()
        traced = (s1["closest_rays"] - s0["closest_rays"]) + (s1["shadow_rays"] - s0["shadow_rays"])
        render = {
            "workload": f"sponza_class {W}x{H}, depth {args.render_depth}, point+env NEE, {args.render_passes} passes per GPU, still "
                        "camera: the per-camera cache is warm (primary rays and point-light shadow rays of the primary vertex "
                        "are not re-traced)",
            "bvh": "binary (bit-exact hit ids, films identical to the oracle's)",
            "samples_per_sec": W * H * args.render_passes * world / r_el,
            "mrays_per_sec": traced * world / r_el / 1e6,
            "traced_rays_per_pass": traced / args.render_passes,
            "ms_per_pass": r_el / args.render_passes * 1e3,
        }
        # what a host that moves the camera every frame pays: the first pass after a camera change fills the cache
        import copy
        cam2 = copy.deepcopy(scene.camera)
        cam2.pos = (np.asarray(cam2.pos, np.float32) + np.float32(1e-3)).astype(np.float32)
        firsts = []
        for rep in range(3):
            rctx.set_camera(cam2 if rep % 2 == 0 else scene.camera)
            rctx.synchronize()
            f0 = time.perf_counter()
            rctx.render(1)
            rctx.synchronize()
            firsts.append((time.perf_counter() - f0) * 1e3)
        render["ms_first_pass_after_camera_change"] = float(np.median(firsts))
        rctx.set_camera(scene.camera)
        if wctx is not None:
            wctx.set_camera(scene.camera)
            wctx.render(args.render_passes)
            wctx.reset_accumulation()
            wctx.synchronize()
            w0 = time.perf_counter()
            wctx.render(args.render_passes)
            wctx.synchronize()
            render["wide_tree_ms_per_pass"] = (time.perf_counter() - w0) / args.render_passes * 1e3
            rctx.reset_accumulation()
            rctx.render(args.render_passes)
            fw, fb = wctx.read_film(), rctx.read_film()
            render["wide_tree_pixels_differing_from_binary"] = int((fw.view(np.uint32) != fb.view(np.uint32)).any(-1).sum())

    # ---- tile-sharded leg (BASELINE.json configs[3] shape): ONE film split over the ranks by interleaved tiles,
    # no exchange while rendering, one all-gather of the tile slabs over RCCL at read-out.  Strong scaling.  Part of
    # every N > 1 run: a failure here fails the run.
    sharded = None
    if use_dist and args.render_passes > 0 and (world > 1 or os.environ.get("BENCH_FORCE_SHARDED") == "1"):
        from raytracer_amd.dist import gather_film
        SW, SH = 2 * W, 2 * H
        sctx = Context(scene, SW, SH, max_depth=args.render_depth, seed=0, device=local_rank, rank=rank, world=world)
        sctx.set_camera(scene.camera)
        sctx.render(args.render_passes)  # warm-up
        gather_film(sctx)
        sctx.reset_accumulation()
        barrier()
        g0 = time.perf_counter()
        sctx.render(args.render_passes)
        sctx.synchronize()
        g1 = time.perf_counter()
        film = gather_film(sctx)
        barrier()
        g2 = time.perf_counter()
        t = torch.tensor([g2 - g0, g2 - g1], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        tot, gat = float(t[0].item()), float(t[1].item())
        # the gathered film against rank-local renders, bit for bit: this rank's own tiles, and the tiles of the
        # next rank rendered here again by a second context
        ty, tx = np.mgrid[0:SH, 0:SW]
        owner = ((ty // 64) * (-(-SW // 64)) + tx // 64) % world
        mine = sctx.read_film()
        bad = int((film.view(np.uint32)[owner == rank] != mine.view(np.uint32)[owner == rank]).any(-1).sum())
        other = (rank + 1) % world
        if other != rank:
            octx = Context(scene, SW, SH, max_depth=args.render_depth, seed=0, device=local_rank, rank=other, world=world)
            octx.set_camera(scene.camera)
            octx.render(args.render_passes)
            theirs = octx.read_film()
            octx.close()
            bad += int((film.view(np.uint32)[owner == other] != theirs.view(np.uint32)[owner == other]).any(-1).sum())
        tb = torch.tensor([bad], dtype=torch.int64, device=dev)
        dist.all_reduce(tb, op=dist.ReduceOp.SUM)
        sharded = {"workload": f"sponza_class {SW}x{SH}, depth {args.render_depth}, {args.render_passes} passes, film sharded by "
                               f"64x64 tiles over {world} GPU(s), slabs all-gathered over RCCL; still camera: the per-camera cache is warm "
                               f"(primary rays and point-light shadow rays of the primary vertex are not re-traced)", "scaling": "strong",
                   "samples_per_sec": SW * SH * args.render_passes / tot, "ms_per_pass": tot / args.render_passes * 1e3,
                   "gather_ms": gat * 1e3, "film_finite": bool(np.isfinite(film).all()),
                   "film_mean": float(film[..., :3].mean()),
                   "pixels_differing_from_rank_local_renders": int(tb.item())}
        sctx.close()
        if int(tb.item()) != 0 or not sharded["film_finite"]:
            print(f"bench.py: tile-sharded film differs from the rank-local renders: {sharded}", file=sys.stderr)
            dist.destroy_process_group()
            sys.exit(3)

    if rank != 0:
        if use_dist:
            dist.destroy_process_group()
        return

    # ---- CPU side: the oracle on the same ray sets (rank 0): traversal
    # counters for the algorithmic byte count, the CPU baseline, and a
    # full-size bit-exact parity check of the hit ids.
    roofline = None
    cpu_baseline = None
    parity = None
    if not args.no_cpu_baseline:
        from oracle.binding import OracleScene
        threads = args.cpu_threads or max(1, min(16, len(os.sched_getaffinity(0))))
        orc = OracleScene(scene)
        counts = {}
        cpu_s = 0.0
        cpu_reps = 3  # about 20 CPU-seconds at 16 threads
        parity = {"rays": NS * n, "bvh": args.bvh, "id_mismatches": 0, "dist_bit_mismatches": 0, "id_mismatches_on_exact_ties": 0,
                  "near_ties": 0, "id_mismatches_outside_ties": 0, "unexplained": 0, "exact_distance_ties": 0, "per_set": {}}
        parity_other = None if other_id1 is None else {"rays": NS * n, "bvh": other_name, "id_mismatches": 0, "dist_bit_mismatches": 0,
                                                         "id_mismatches_on_exact_ties": 0, "near_ties": 0,
                                                         "id_mismatches_outside_ties": 0, "unexplained": 0}
        for j, k in enumerate(sets):
            for rep in range(cpu_reps):
                c0 = time.perf_counter()
                rt, rid0, rid1, ru, rv, cnt = orc.intersect(sets[k], counters=True, threads=threads)
                cpu_s += time.perf_counter() - c0
            counts[k] = cnt
            sl = slice(j * n, (j + 1) * n)
            cls = classify_against_oracle(orc, sets[k], gpu_t[sl], gpu_id1[sl], rt, rid1, threads)
            # triangle tests of the oracle's own traversal that tied the held hit exactly (first-visited wins on both sides)
            cls["exact_distance_ties"] = int(cnt["n_ties"])
            cls["hit_fraction"] = float((rid1 >= 0).mean())
            parity["per_set"][k] = cls
            for key in ("id_mismatches", "dist_bit_mismatches", "id_mismatches_on_exact_ties", "near_ties", "id_mismatches_outside_ties",
                        "unexplained", "exact_distance_ties"):
                parity[key] += cls[key]
            if parity_other is not None:
                clo = classify_against_oracle(orc, sets[k], other_t[sl], other_id1[sl], rt, rid1, threads)
                for key in clo:
                    parity_other[key] += clo[key]
        cpu_baseline = {"value": cpu_reps * NS * n / cpu_s / 1e6, "unit": "Mrays/s", "cores": threads, "kind": "port",
                        "cpu_model": cpu_model(), "cores_available": len(os.sched_getaffinity(0)),
                        "sample": f"the full ray sets ({NS*n} rays) x {cpu_reps}, oracle closest hit, {threads} threads, {cpu_s:.1f} s wall"}
        if render is not None:
            # the oracle's integrator on a bounded sample of the render workload: one full-resolution pass
            rw, rh = W, H
            c0 = time.perf_counter()
            _, rc = orc.render(scene.camera, rw, rh, args.render_depth, 0, 1, threads=threads)
            c1 = time.perf_counter() - c0
            render["cpu_baseline"] = {"samples_per_sec": rw * rh / c1, "mrays_per_sec": (rc["closest"] + rc["shadow"]) / c1 / 1e6,
                                      "cores": threads, "kind": "port",
                                      "sample": f"1 pass at {rw}x{rh}, depth {args.render_depth}, {c1:.1f} s wall"}
            # full-size film parity: one pass of the GPU tracer against one pass of the oracle with its transcendentals
            # switched to the polynomials the kernels restate ("shared" mode), compared bit for bit
            from oracle import binding as oracle_binding
            bctx.reset_accumulation()
            bctx.render(1)
            g_film = bctx.read_film()
            oracle_binding.set_math("shared")
            o_film, _ = orc.render(scene.camera, rw, rh, args.render_depth, 0, 1, threads=threads)
            oracle_binding.set_math("libm")
            same = (g_film.view(np.uint32) == o_film.view(np.uint32)) | (np.isnan(g_film) & np.isnan(o_film))
            render["film_parity"] = {"passes": 1, "pixels": rw * rh, "bit_identical": bool(same.all()),
                                     "pixels_differing": int((~same).any(-1).sum())}
            # Roofline of a sample pass: the reference's algorithm on the rays of ONE pass (the oracle's integrator counts
            # the node visits and triangle tests of its closest-hit and shadow rays) as fp32 lane-operations, over the
            # time of a pass (all kernels of the pass: traversal, shading, accumulation), against the fp32 vector rate.
            # A still camera lets the tracer keep the camera rays and the point-light shadow rays of the primary
            # vertex from the pass before (they are the same rays): the steady-state figure leaves their work out of
            # the numerator; the first pass after a camera change traces everything.
            r_rays = rc["closest"] + rc["shadow"]
            r_inner = rc["closest_inner"] + rc["shadow_inner"]
            r_tri = rc["closest_tri"] + rc["shadow_tri"]
            ops_all = algorithmic_valu_ops(r_rays, r_inner, r_tri)
            ops_steady = algorithmic_valu_ops(r_rays - rc["prim_rays"] - rc["prim_point_rays"], r_inner - rc["prim_inner"] - rc["prim_point_inner"],
                                              r_tri - rc["prim_tri"] - rc["prim_point_tri"])
            t_pass = render["ms_per_pass"] * 1e-3
            render["roofline"] = {
                "bound": "valu", "unit": "T lane-ops/s", "peak": VALU_PEAK_TOPS,
                "achieved": ops_steady / t_pass / 1e12, "frac": ops_steady / t_pass / 1e12 / VALU_PEAK_TOPS,
                "ops_per_pass": ops_steady, "ops_per_pass_with_cached_rays": ops_all,
                "oracle_counts_per_pass": rc,
                "first_pass_after_camera_change": {"achieved": ops_all / (render["ms_first_pass_after_camera_change"] * 1e-3) / 1e12,
                                                   "frac": ops_all / (render["ms_first_pass_after_camera_change"] * 1e-3) / 1e12 / VALU_PEAK_TOPS,
                                                   "note": "one pass traced alone (batch of 1): every launch of the pass pays its drain"},
                "note": "12 per ray + 54 per inner visit + 57 per triangle test, rays of one pass as the oracle's integrator traces them "
                        "(the tracer skips rays whose result cannot change the film: the last bounce's extension ray and shadow rays "
                        "with zero contribution); denominator = ms_per_pass, every kernel of the pass; the traversal launches' share "
                        "of a pass is in profiles/r03/render_kernel_stats.csv"}
        n_inner = sum(counts[k]["n_inner"] for k in counts)
        n_tri = sum(counts[k]["n_tri"] for k in counts)
        tot_bytes = algorithmic_bytes(NS * n, n_inner, n_tri)
        tot_ops = algorithmic_valu_ops(NS * n, n_inner, n_tri)
        launch_ms = float(np.mean(kernel_ms))
        secs = launch_ms * 1e-3
        # Candidate bounds of the launch, each as algorithmic work / launch time / peak:
        #  * valu: the slab and triangle arithmetic of the reference's algorithm against the fp32 vector rate;
        #  * hbm: the bytes that must cross HBM once per launch -- rays in, hits out, and every node and triangle
        #    once (the tree is cache-resident: 56 B per visit are NOT HBM bytes) -- against 8 TB/s;
        # plus, as a diagnostic, the per-visit byte model against HBM peak (above 1: a cache-less machine could not
        # run this launch in this time; it says the caches work, it is not a bound).
        st = ctx.stats()
        tree_bytes = 64.0 * (st["n_triangles"] - 1) + 48.0 * st["n_triangles"]
        compulsory = 44.0 * NS * n + tree_bytes
        headline_kernel = "k_trace_closest_wide" if args.bvh == "wide" else "k_trace_closest<false>"
        traffic = None
        counters = None   # pipe utilisation of the same kernel from committed rocprofv3 --pmc passes (not this run)
        tp = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tp):
            tj = json.load(open(tp))
            tj = tj.get(headline_kernel)
        else:
            tj = None
        if tj:
            traffic = tj.get("bytes_per_launch")
            counters = {k: tj[k] for k in ("tcc_hit_rate", "valu_pipe_busy", "ta_busy", "valu_active_lane_fraction",
                                           "valu_instructions_per_ray", "salu_instructions_per_ray", "simd_cycles_per_ray",
                                           "rocprof_avg_launch_us") if k in tj}
            counters["source"] = "profiles/traffic.json: " + tj.get("source", "")
        cand = {
            "valu": {"achieved": tot_ops / secs / 1e12, "peak": VALU_PEAK_TOPS, "unit": "T lane-ops/s",
                     "frac": tot_ops / secs / 1e12 / VALU_PEAK_TOPS, "ops_per_launch": tot_ops,
                     "note": "fp32 lane-operations (no FMA: the path may not contract); 54 per inner visit, 57 per triangle "
                             "test, 12 per ray; peak = 256 CUs x 4 SIMD-32 x 2.4 GHz"},
            "hbm": {"achieved": compulsory / secs / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": compulsory / secs / 1e9 / HBM_PEAK_GBS, "bytes_per_launch": compulsory,
                    "note": "compulsory bytes: 44 B per ray + the node and triangle arrays once"},
        }
        mix_cycles = algorithmic_valu_cycles(NS * n, n_inner, n_tri)
        cand["valu_at_measured_issue_rates"] = {
            "frac": mix_cycles / (1024.0 * 2.4e9 * secs), "simd_cycles_per_launch": mix_cycles,
            "note": "diagnostic, not the headline: the same lane-operations as full wave instructions at the issue rates measured "
                    "on this chip (add/sub/mul 2.5 cycles per wave instruction, min/max/compare/select 4.2; "
                    "profiles/r03/inst_ubench.json) over the launch's 1024 SIMDs x 2.4 GHz x time; the `valu` candidate prices "
                    "every operation at 2 cycles"}
        if traffic:
            cand["hbm_measured"] = {"achieved": traffic / (counters["rocprof_avg_launch_us"] * 1e-6) / 1e9, "peak": HBM_PEAK_GBS,
                                    "unit": "GB/s", "frac": traffic / (counters["rocprof_avg_launch_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS,
                                    "note": "bytes beyond L2 (FETCH_SIZE + WRITE_SIZE) of the profiled launch / its duration"}
            cand["l1_address_path"] = {"frac": counters.get("ta_busy"), "note": "TA_TA_BUSY / cycles of the profiled launch"}
        binding = max(("valu", "hbm"), key=lambda k: cand[k]["frac"])
        b = cand[binding]
        roofline = {"bound": binding, "kernel": headline_kernel, "achieved": b["achieved"], "peak": b["peak"], "unit": b["unit"],
                    "frac": b["frac"], "traffic": traffic, "candidates": cand, "counters": counters,
                    "per_visit_byte_model": {"bytes_per_launch": tot_bytes, "gbs": tot_bytes / secs / 1e9,
                                             "over_hbm_peak": tot_bytes / secs / 1e9 / HBM_PEAK_GBS,
                                             "note": "44 B per ray + 56 B per inner visit + 36 B per triangle test; the visits are "
                                                     "served by LDS, L1, L2 and Infinity Cache, so this is not a bound"},
                    "avg_launch_ms": launch_ms, "rays_per_launch": NS * n,
                    "per_set": {k: {"n_inner_per_ray": counts[k]["n_inner"] / n, "n_tri_per_ray": counts[k]["n_tri"] / n,
                                    "bytes_per_ray": algorithmic_bytes(n, counts[k]["n_inner"], counts[k]["n_tri"]) / n,
                                    "valu_ops_per_ray": algorithmic_valu_ops(n, counts[k]["n_inner"], counts[k]["n_tri"]) / n,
                                    "launched_alone_ms": set_ms[k],
                                    "launched_alone_mrays_per_sec": n / set_ms[k] / 1e3} for k in counts}}

    line = {
        "kernel_ms": {"all_sets_one_launch": float(np.mean(kernel_ms)), **{k + "_alone": v for k, v in set_ms.items()},
                      "all_sets_one_launch_" + other_name + "_tree": other_ms},
        "metric": "Mrays/sec (closest-hit, whole node)", "value": mrays, "unit": "Mrays/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "settle_steps": args.settle_steps, "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"primary-ray closest hit, sponza_class ({scene.n_triangles} triangles) {W}x{H}: "
                               f"coherent + incoherent + bounce sets of {n} rays each per GPU per step, one launch",
                   "rays_per_step": rays_per_step,
                   "bvh": "reference LBVH, 1 triangle per leaf" + (", collapsed to 4-wide nodes (same triangles reachable; ids may differ "
                                                                   "only where triangles tie in distance: see parity)" if args.bvh == "wide" else "")},
        "roofline": roofline, "cpu_baseline": cpu_baseline, "sharded_render": sharded, "parity": parity,
        "parity_" + other_name + "_tree": parity_other if not args.no_cpu_baseline else None, "render": render,
    }
    line.update(strong_scaling_fields(world, sharded))
    sys.stdout.flush()
    with os.fdopen(json_fd, "w") as out:
        out.write(json.dumps(line) + "\n")
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
