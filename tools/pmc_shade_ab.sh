#!/bin/bash
# TA / VALU counters of k_shade for two builds of the library (before / after a k_shade change).
# usage: tools/pmc_shade_ab.sh <before.so> ; results under gpurun_out/pmc_shade_{before,after}
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for v in before after; do
  if [ $v = before ]; then export RT_HIP_LIB=$R/$1; else unset RT_HIP_LIB; fi
  OUT=$R/gpurun_out/pmc_shade_$v
  mkdir -p $OUT
  CMD="python3 $R/tools/render_only.py 16 plain 16"
  timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq -- $CMD > $OUT/sq.log 2>&1
  timeout -s KILL 120 rocprofv3 --pmc TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE --output-format csv -d $OUT/ta -- $CMD > $OUT/ta.log 2>&1
  timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $CMD > $OUT/stats.log 2>&1
done
echo done
