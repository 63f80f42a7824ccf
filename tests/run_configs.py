"""Measurement script (kept under tests/ because it runs the CPU oracle as checker and baseline; pytest does not
collect it).  Runs the BASELINE.json configurations that fit one GPU and writes
profiles/<tag>/configs.json: GPU time, CPU-oracle time on a bounded sample,
and film agreement where both sides render the same passes.
usage (on the GPU box): python tests/run_configs.py <out.json> [wide] [c1 c3 c4 c5 c5a tex n4]
("wide": every context walks the 4-wide tree, RT_FLAG_WIDE_BVH)"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import binding as oracle_binding  # noqa: E402
from oracle.binding import OracleScene  # noqa: E402
from raytracer_amd import Context, scenes  # noqa: E402

THREADS = max(1, min(16, len(os.sched_getaffinity(0))))
WIDE = False


def agree(film, ref):
    a, b = film[..., :3].astype(np.float64), ref[..., :3].astype(np.float64)
    err = np.abs(a - b)
    same = (film.view(np.uint32) == ref.view(np.uint32)) | (np.isnan(film) & np.isnan(ref))
    return {"bit_identical": bool(same.all()), "pixels_differing": int((~same).any(-1).sum()),
            "rel_rmse": float(np.sqrt((err ** 2).mean()) / (np.sqrt((b ** 2).mean()) + 1e-30)),
            "max_abs": float(err.max()),
            "frac_pixels_rel_gt_1e-3": float(((err / (np.abs(b) + 1e-3)).max(-1) > 1e-3).mean())}


def gpu_render(scene, W, H, depth, spp, seed=0):
    t0 = time.perf_counter()
    ctx = Context(scene, W, H, max_depth=depth, seed=seed, wide_bvh=WIDE)
    setup = time.perf_counter() - t0
    ctx.set_camera(scene.camera)
    ctx.render(1)
    ctx.reset_accumulation()
    s0 = ctx.stats()
    t0 = time.perf_counter()
    ctx.render(spp)
    ctx.synchronize()
    el = time.perf_counter() - t0
    s1 = ctx.stats()
    film = ctx.read_film()
    rays = (s1["closest_rays"] - s0["closest_rays"]) + (s1["shadow_rays"] - s0["shadow_rays"])
    res = {"seconds": el, "spp": spp, "samples_per_sec": W * H * spp / el, "mrays_per_sec": rays / el / 1e6,
           "ms_per_pass": el / spp * 1e3, "setup_seconds": setup, "bvh_build_ms": s1["bvh_build_ms"],
           "n_triangles": s1["n_triangles"], "bvh_max_depth": s1["bvh_max_depth"]}
    ctx.close()
    return film, res


def cpu_render(scene, W, H, depth, spp, seed=0):
    """Times the oracle with libm (what the reference calls), then renders the film that is compared with the
    GPU's in the shared-math mode, under which the two sides are comparable bit for bit."""
    orc = OracleScene(scene)
    oracle_binding.set_math("libm")
    t0 = time.perf_counter()
    film, cnt = orc.render(scene.camera, W, H, depth, seed, spp, threads=THREADS)
    el = time.perf_counter() - t0
    oracle_binding.set_math("shared")
    film, _ = orc.render(scene.camera, W, H, depth, seed, spp, threads=THREADS)
    oracle_binding.set_math("libm")
    return film, {"seconds": el, "spp": spp, "threads": THREADS, "samples_per_sec": W * H * spp / el,
                  "mrays_per_sec": (cnt["closest"] + cnt["shadow"]) / el / 1e6}


def main():
    out_path = sys.argv[1]
    global WIDE
    which = sys.argv[2:]
    if "wide" in which:
        WIDE = True
        which.remove("wide")
    which = which or ["c1", "c3", "c4", "c5"]
    out = {"cpu_threads": THREADS, "bvh": "4-wide" if WIDE else "binary"}
    if "c1" in which:
        s = scenes.cornell_box()
        film, g = gpu_render(s, 512, 512, 4, 16)
        ref, c = cpu_render(s, 512, 512, 4, 16)
        out["c1_cornell_512_16spp_d4"] = {"gpu": g, "cpu_oracle": c, "agreement": agree(film, ref)}
        print("c1", json.dumps(out["c1_cornell_512_16spp_d4"]), flush=True)
    if "c3" in which or "c4" in which:
        sp = scenes.sponza_class()
    if "c3" in which:
        film64, g = gpu_render(sp, 1920, 1080, 8, 64)
        film4, g4 = gpu_render(sp, 1920, 1080, 8, 4)
        ref4, c = cpu_render(sp, 1920, 1080, 8, 4)
        c["extrapolated_seconds_64spp"] = c["seconds"] * 16
        out["c3_sponza_1080p_64spp_d8"] = {"gpu": g, "gpu_4spp": g4, "cpu_oracle_4spp": c, "agreement_4spp": agree(film4, ref4)}
        print("c3", json.dumps(out["c3_sponza_1080p_64spp_d8"]), flush=True)
    if "c4" in which:
        film, g = gpu_render(sp, 3840, 2160, 8, 1024)
        out["c4_sponza_4k_1024spp_1gpu"] = {"gpu": g, "finite": bool(np.isfinite(film).all()), "mean_rgb": film[..., :3].mean((0, 1)).tolist()}
        print("c4", json.dumps(out["c4_sponza_4k_1024spp_1gpu"]), flush=True)
    if "c5" in which:
        t0 = time.perf_counter()
        b = scenes.bistro_class()
        gen = time.perf_counter() - t0
        film, g = gpu_render(b, 1920, 1080, 12, 4)
        g["extrapolated_seconds_256spp"] = g["seconds"] * 64
        fs, gs = gpu_render(b, 240, 136, 12, 1, seed=4)
        rs, c = cpu_render(b, 240, 136, 12, 1, seed=4)
        out["c5_bistro_1080p_d12_65lights"] = {"scene_gen_seconds": gen, "n_lights": len(b.lights), "gpu_4spp": g,
                                               "cpu_oracle_240x136_1spp": c, "agreement_240x136": agree(fs, rs),
                                               "finite": bool(np.isfinite(film).all())}
        print("c5", json.dumps(out["c5_bistro_1080p_d12_65lights"]), flush=True)
    if "c5a" in which:
        # C5 as worded in BASELINE.json ("many area lights"): the same street block with an emissive triangle in
        # place of each point light (bounded shadow rays, the build's own light kind)
        b = scenes.bistro_class(area_lights=True)
        film, g = gpu_render(b, 1920, 1080, 12, 4)
        g["extrapolated_seconds_256spp"] = g["seconds"] * 64
        fs, gs = gpu_render(b, 240, 136, 12, 1, seed=4)
        rs, c = cpu_render(b, 240, 136, 12, 1, seed=4)
        out["c5a_bistro_1080p_d12_64_area_lights"] = {"n_lights": len(b.lights), "gpu_4spp": g, "cpu_oracle_240x136_1spp": c,
                                                      "agreement_240x136": agree(fs, rs), "finite": bool(np.isfinite(film).all()),
                                                      "mean_rgb": film[..., :3].mean((0, 1)).tolist()}
        print("c5a", json.dumps(out["c5a_bistro_1080p_d12_64_area_lights"]), flush=True)
    if "tex" in which:
        # C3's scene with diffuse maps on every model: cost of k_quad_duv + the filtered fetches
        sp = scenes.sponza_class()
        st = scenes.with_textures(sp, uv_scale=8.0)
        _, g0 = gpu_render(sp, 1920, 1080, 8, 16)
        film4, g1 = gpu_render(st, 1920, 1080, 8, 16)
        f2, g2 = gpu_render(st, 480, 270, 8, 2, seed=4)
        r2, c2 = cpu_render(st, 480, 270, 8, 2, seed=4)
        out["sponza_textured_1080p_d8"] = {"gpu_untextured_16spp": g0, "gpu_textured_16spp": g1, "cpu_oracle_480x270_2spp": c2,
                                           "agreement_480x270": agree(f2, r2), "finite": bool(np.isfinite(film4).all())}
        print("tex", json.dumps(out["sponza_textured_1080p_d8"]), flush=True)
    if "n4" in which:
        # C3's scene with the GGX lobe on every model and a ceiling panel of two area lights added
        sp = scenes.sponza_class()
        panel = scenes.area_quad((-6, 14.5, -3), (6, 14.5, -3), (6, 14.5, 3), (-6, 14.5, 3), (60.0, 55.0, 45.0))
        sg = scenes.with_glossy(scenes.replace(sp, lights=sp.lights + panel), scenes.GLOSSY_GGX)
        film, g1 = gpu_render(sg, 1920, 1080, 8, 16)
        f2, g2 = gpu_render(sg, 480, 270, 8, 2, seed=4)
        r2, c2 = cpu_render(sg, 480, 270, 8, 2, seed=4)
        out["sponza_ggx_area_1080p_d8"] = {"n_lights": len(sg.lights), "gpu_16spp": g1, "cpu_oracle_480x270_2spp": c2,
                                           "agreement_480x270": agree(f2, r2), "finite": bool(np.isfinite(film).all())}
        print("n4", json.dumps(out["sponza_ggx_area_1080p_d8"]), flush=True)
    json.dump(out, open(out_path, "w"), indent=1)


if __name__ == "__main__":
    main()
