"""Writes profiles/traffic.json (read by bench.py for `roofline.traffic`) from summaries made by
profiles/summarize.py, one entry per headline kernel.
Usage: python profiles/make_traffic.py k_trace_closest_wide=profiles/r03/wide_summary.json "k_trace_closest<false>=profiles/r03/binary_summary.json"

HBM-side bytes per launch of the headline kernel = FETCH_SIZE + WRITE_SIZE (KiB units, separate --pmc
passes).  FETCH_SIZE is taken x1.0: the kernel's reads are random 64-byte node gathers, for which the counter
was calibrated against a known byte count (profiles/r01/fetch_size_calibration.log: x1.03); the halving that
applies to wide coalesced streams on gfx950 does not apply here."""
import json
import os
import sys

RAYS_PER_LAUNCH = 3 * 1920 * 1080


def one(kernel, path):
    d = json.load(open(path))
    k = [x for x in d["counters"] if kernel in x][0]
    c = d["counters"][k]
    fetch = c["FETCH_SIZE"]["mean_per_launch"] * 1024.0
    write = c["WRITE_SIZE"]["mean_per_launch"] * 1024.0
    hit, miss = c["TCC_HIT_sum"]["mean_per_launch"], c["TCC_MISS_sum"]["mean_per_launch"]
    cycles = c["GRBM_GUI_ACTIVE"]["mean_per_launch"] / 8.0   # the counter sums the 8 XCDs
    return {
        "bytes_per_launch": fetch + write,
        "fetch_bytes": fetch,
        "write_bytes": write,
        "rays_per_launch": RAYS_PER_LAUNCH,
        "source": f"{path} (rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes, mean over the "
                  f"{c['FETCH_SIZE']['launches']} launches of bench.py --steps 5 --warmup 2 --no-set-diagnostics: "
                  "one launch over the three ray sets)",
        "correction": "FETCH_SIZE x1.0 (random 64-byte gathers: calibrated x1.03, profiles/r01/fetch_size_calibration.log); "
                      "WRITE_SIZE taken as is. Counter unit: KiB.",
        "tcc_hit_rate": hit / (hit + miss),
        # pipe utilisation over the launch: SQ_ACTIVE_INST_VALU counts, per SIMD, quarter-rate issue slots in which a
        # VALU instruction executes (x4 = cycles; 1024 SIMDs); TA_TA_BUSY is summed over the 256 CUs
        "valu_pipe_busy": c["SQ_ACTIVE_INST_VALU"]["mean_per_launch"] * 4.0 / (1024.0 * cycles),
        "ta_busy": c["TA_TA_BUSY_sum"]["mean_per_launch"] / (256.0 * cycles),
        "valu_active_lane_fraction": c["SQ_THREAD_CYCLES_VALU"]["mean_per_launch"] / (64.0 * c["SQ_ACTIVE_INST_VALU"]["mean_per_launch"]),
        "valu_instructions_per_ray": c["SQ_INSTS_VALU"]["mean_per_launch"] / RAYS_PER_LAUNCH,
        "salu_instructions_per_ray": c["SQ_INSTS_SALU"]["mean_per_launch"] / RAYS_PER_LAUNCH,
        "simd_cycles_per_ray": cycles * 1024.0 / RAYS_PER_LAUNCH,
        "rocprof_avg_launch_us": d["kernels"][k]["avg_ns"] / 1e3,
        "rocprof_calls": d["kernels"][k]["calls"],
    }


def main(specs):
    out = {}
    for spec in specs:
        kernel, path = spec.split("=", 1)
        out[kernel] = one(kernel, path)
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "traffic.json")
    json.dump(out, open(dst, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(sys.argv[1:])
