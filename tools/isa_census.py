"""Instruction census of a kernel's ISA (the .s file `make asm` writes): per basic-block range, counts by class and a
weighted VALU cycle estimate (add/sub/mul/fma 2.5 cycles, every other VALU instruction 4.2: tools/ubench/inst_ubench).
usage: python tools/isa_census.py rt_hip.s <kernel-name-substring> [first_line last_line]   (line numbers inside the kernel)"""
import re, sys
path, pat = sys.argv[1], sys.argv[2]
lines = open(path).read().splitlines()
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S*" + re.escape(pat) + r"\S*:", l))
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
body = lines[start:end + 1]
lo, hi = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (0, len(body))
FAST = ("v_add_f32", "v_sub_f32", "v_mul_f32", "v_fma_f32", "v_subrev_f32", "v_mac_f32", "v_fmac_f32")
cls = {"valu_fast": 0, "valu_other": 0, "salu": 0, "vmem": 0, "lds": 0, "wait": 0, "branch": 0, "nop": 0}
for l in body[lo:hi]:
    t = l.strip().split()
    if not t or t[0].startswith((";", ".")) or t[0].endswith(":"):
        continue
    op = t[0]
    if op.startswith("s_waitcnt"): cls["wait"] += 1
    elif op.startswith("s_nop"): cls["nop"] += 1
    elif op.startswith(("s_cbranch", "s_branch")): cls["branch"] += 1
    elif op.startswith("s_"): cls["salu"] += 1
    elif op.startswith(("buffer_", "global_", "flat_", "scratch_")): cls["vmem"] += 1
    elif op.startswith("ds_"): cls["lds"] += 1
    elif op.startswith("v_"):
        cls["valu_fast" if op.split("_e")[0] in FAST or any(op.startswith(f) for f in FAST) else "valu_other"] += 1
cls["valu_cycles_est"] = round(cls["valu_fast"] * 2.5 + cls["valu_other"] * 4.2, 1)
print(pat, "lines", lo, hi, cls)
