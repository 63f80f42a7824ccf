#!/bin/bash
# Kernel A/B helper (development only): builds variants of librt_hip.so with
# different -D settings into build/variants/ and, on the GPU box, benches each.
#   tools/sweep.sh build "name1:-DRT_LEAF_MIN=8 -DRT_REFILL_MIN=8" "name2:..."
#   tools/sweep.sh run  name1 name2 ...
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
V=$R/tools/variants
mkdir -p $V
mode=$1; shift
if [ "$mode" = build ]; then
  for spec in "$@"; do
    name=${spec%%:*}; flags=${spec#*:}
    (mkdir -p $V/obj_$name && cd $R/raytracer_amd/csrc && make -s lbvh_host.o lbvh_device.o wide_bvh.o rt_node.o && make -s OBJ=$V/obj_$name OUT=$V/librt_$name.so EXTRA="$flags" $V/librt_$name.so) &
  done
  wait
  ls -la $V
else
  for name in "$@"; do
    RT_HIP_LIB=$V/librt_$name.so timeout -k 10 200 python3 -u $R/bench.py --steps 10 --warmup 2 --render-passes ${RENDER:-0} --no-cpu-baseline > $R/gpurun_out/sweep_$name.log 2>&1 || { echo "$name FAILED"; tail -3 $R/gpurun_out/sweep_$name.log; exit 1; }
    python3 - "$name" "$R/gpurun_out/sweep_$name.log" <<'PY'
import json,sys
d=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
r=d.get("render") or {}
print(sys.argv[1], d.get("kernel_ms"), "Mrays/s %.0f" % d["value"], "ms/step %.3f" % d["ms_per_step"], "render ms/pass %s" % r.get("ms_per_pass"))
PY
  done
fi
