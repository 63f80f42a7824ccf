"""Summarises a rocprofv3 kernel trace of tools/render_only.py: per kernel and
call index within a pass, mean duration, plus busy/idle/overlap of the
device over the last pass.  usage: python tools/timeline.py <kernel_trace.csv> [passes]"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 4
ev = []
for r in rows:
    name = r["Kernel_Name"].split("(")[0].replace("void rt::", "").replace("rt::", "")
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name))
ev.sort()
# a batch ends with k_film
ends = [i + 1 for i, e in enumerate(ev) if e[2].startswith("k_film")]
bounds = [0] + ends
segs = [ev[a:b] for a, b in zip(bounds, bounds[1:])][-passes:]
segs = [[e for e in seg if e[2].startswith("k_")] for seg in segs]
agg = defaultdict(list)
for seg in segs:
    cnt = defaultdict(int)
    for s, e, n in seg:
        agg[(n, cnt[n])].append((e - s) / 1e3)
        cnt[n] += 1
names = sorted({k[0] for k in agg})
for n in names:
    ks = sorted(k for k in agg if k[0] == n)
    print(f"{n:28s}", " ".join(f"{sum(agg[k]) / len(agg[k]):7.1f}" for k in ks), " | sum", f"{sum(sum(agg[k]) / len(agg[k]) for k in ks):8.1f} us")
seg = segs[-1]
t0, t1 = seg[0][0], max(e for _, e, _ in seg)
pts = sorted([(s, 1) for s, _, _ in seg] + [(e, -1) for _, e, _ in seg])
busy = both = 0
depth, last = 0, t0
for t, d in pts:
    if depth >= 1:
        busy += t - last
    if depth >= 2:
        both += t - last
    depth += d
    last = t
print(f"last pass: wall {(t1 - t0) / 1e3:.1f} us, busy {busy / 1e3:.1f}, two kernels in flight {both / 1e3:.1f}, idle {(t1 - t0 - busy) / 1e3:.1f}, kernels {len(seg)}")
