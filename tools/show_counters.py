import json,sys
d=json.load(open(sys.argv[1]))
k=[x for x in d["kernels"] if "k_trace_closest" in x]
for kk in k: print(kk, d["kernels"][kk])
for kk in d["counters"]:
    if "k_trace_closest" in kk:
        print(kk)
        for c,x in sorted(d["counters"][kk].items()): print("  ",c, "%.4g"%x["mean_per_launch"], x["launches"])
