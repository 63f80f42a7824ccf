"""Instruction census of a kernel's ISA (the .s file `make asm` writes): per basic block, counts by class and a
weighted VALU cycle estimate (add/sub/mul/fma 2.5 cycles per wave instruction, every other VALU instruction 4.2:
tools/ubench/inst_ubench).
usage: python tools/isa_census.py rt_hip.s <kernel-name-substring> [--sections name=3,7,12;name2=...] [--json out.json]
Block numbers are the N of the compiler's .LBB<k>_N labels ("entry" = the code before the first label); a section
is a set of blocks named by hand after reading the table (the labels move when the source changes)."""
import json
import re
import sys

FAST = ("v_add_f32", "v_sub_f32", "v_mul_f32", "v_fma_f32", "v_subrev_f32", "v_mac_f32", "v_fmac_f32")
KEYS = ("valu_fast", "valu_other", "salu", "vmem", "lds", "wait", "branch", "nop", "atomic")


def blocks_of(lines, pat):
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S*" + re.escape(pat) + r"\S*:", l))
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    out = []
    cur = dict(block="entry", line=0, to=[], **{k: 0 for k in KEYS})
    for i in range(start + 1, end + 1):
        l = lines[i].strip()
        m = re.match(r"^\.LBB\d+_(\d+):", l)
        if m:
            out.append(cur)
            cur = dict(block=m.group(1), line=i - start, to=[], **{k: 0 for k in KEYS})
            continue
        t = l.split()
        if not t or t[0].startswith((";", ".")):
            continue
        op = t[0]
        if op.startswith("s_waitcnt"): cur["wait"] += 1
        elif op.startswith("s_nop"): cur["nop"] += 1
        elif op.startswith(("s_cbranch", "s_branch")):
            cur["branch"] += 1
            cur["to"].append(t[-1].split("_")[-1])
        elif op.startswith("s_"): cur["salu"] += 1
        elif op.startswith(("buffer_", "global_", "flat_", "scratch_")):
            cur["vmem"] += 1
            cur["atomic"] += "atomic" in op
        elif op.startswith("ds_"): cur["lds"] += 1
        elif op.startswith("v_"):
            cur["valu_fast" if any(op.startswith(f) for f in FAST) else "valu_other"] += 1
    out.append(cur)
    return out


def total(bs):
    s = {k: sum(b[k] for b in bs) for k in KEYS}
    s["valu"] = s["valu_fast"] + s["valu_other"]
    s["valu_cycles_est"] = round(s["valu_fast"] * 2.5 + s["valu_other"] * 4.2, 1)
    return s


def main(argv):
    path, pat = argv[1], argv[2]
    sections, dst = None, None
    if "--sections" in argv:
        sections = {p.split("=")[0]: p.split("=")[1].split(",") for p in argv[argv.index("--sections") + 1].split(";") if p}
    if "--json" in argv:
        dst = argv[argv.index("--json") + 1]
    bs = blocks_of(open(path).read().splitlines(), pat)
    res = {"kernel": pat, "weights": "v_add/sub/mul/fma_f32 2.5 cycles, other VALU 4.2 (tools/ubench/inst_ubench)", "whole_kernel": total(bs)}
    if sections:
        res["sections"] = {}
        for name, ids in sections.items():
            sel = [b for b in bs if b["block"] in ids]
            missing = set(ids) - {b["block"] for b in sel}
            if missing:
                raise SystemExit(f"section {name}: no block {sorted(missing)}")
            res["sections"][name] = dict(blocks=ids, **total(sel))
        named = {i for ids in sections.values() for i in ids}
        res["sections"]["(unassigned)"] = total([b for b in bs if b["block"] not in named])
    else:
        for b in bs:
            print(b["block"], "line", b["line"], {k: b[k] for k in KEYS if b[k]}, "->", ",".join(b["to"]))
    print(json.dumps(res, indent=1))
    if dst:
        res["blocks"] = bs
        json.dump(res, open(dst, "w"), indent=1)


if __name__ == "__main__":
    main(sys.argv)
