set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in pinall pinmin; do
  export RT_HIP_LIB=$R/tools/variants/librt_$v.so
  timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $R/gpurun_out/ab_$v -- python3 $R/bench.py --steps 5 --warmup 2 --render-passes 0 --no-cpu-baseline --no-set-diagnostics > $R/gpurun_out/ab_$v.log 2>&1
done
echo done
