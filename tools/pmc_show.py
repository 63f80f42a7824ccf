"""Prints the derived per-ray figures of a profiles/pmc_passes.sh summary for the kernels whose name contains argv[2]."""
import json, sys
d = json.load(open(sys.argv[1]))
pat = sys.argv[2] if len(sys.argv) > 2 else "k_trace_closest"
rays = float(sys.argv[3]) if len(sys.argv) > 3 else 6220800.0
for k, c in d["counters"].items():
    if pat not in k:
        continue
    g = lambda n: c[n]["mean_per_launch"] if n in c else float("nan")
    cyc = g("GRBM_GUI_ACTIVE") / 8
    print(k, "avg_ns", d["kernels"].get(k, {}).get("avg_ns"))
    print("  per ray: VALU %.1f SALU %.1f VMEM_RD %.2f LDS %.2f" % (g("SQ_INSTS_VALU") / rays, g("SQ_INSTS_SALU") / rays, g("SQ_INSTS_VMEM_RD") / rays, g("SQ_INSTS_LDS") / rays))
    print("  lane fill %.3f  cycles/launch %.4g  SIMD cycles per ray %.0f" % (g("SQ_THREAD_CYCLES_VALU") / g("SQ_INSTS_VALU") / 64, cyc, cyc * 1024 / rays))
    print("  wave cycles: wait_any %.3f issue-stalled %.3f issuing %.3f; SQ_ACTIVE_INST_VALU/ (4*cycles*1024) %.3f" % (
        g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES"), g("SQ_WAIT_INST_ANY") / g("SQ_WAVE_CYCLES"), g("SQ_ACTIVE_INST_ANY") / g("SQ_WAVE_CYCLES"),
        g("SQ_ACTIVE_INST_VALU") * 4 / (cyc * 1024)))
    print("  FETCH %.4g KB WRITE %.4g KB L2 hit %.3f TA_BUSY/cyc %.3f TCP->TCC %.4g LDS conflict %.3f" % (
        g("FETCH_SIZE"), g("WRITE_SIZE"), g("TCC_HIT_sum") / (g("TCC_HIT_sum") + g("TCC_MISS_sum")), g("TA_TA_BUSY_sum") / (cyc * 256 if cyc else 1),
        g("TCP_TCC_READ_REQ_sum"), g("SQ_LDS_BANK_CONFLICT") / g("SQ_LDS_IDX_ACTIVE")))
