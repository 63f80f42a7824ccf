#!/bin/bash
# Collects the profile evidence (headline kernels on both trees, render traces and counters, instruction microbenchmark)
# on the GPU box into gpurun_out/r03, to be copied into profiles/r03 afterwards.
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03
mkdir -p $O
cd $R
BENCH_ARGS="--bvh wide" bash profiles/pmc_passes.sh r03w > $O/pmc_wide.log 2>&1
python3 profiles/summarize.py gpurun_out/prof_r03w $O/wide_summary.json > /dev/null
cp $(find gpurun_out/prof_r03w/stats -name "*kernel_stats.csv" | head -1) $O/wide_kernel_stats.csv
BENCH_ARGS="--bvh binary" bash profiles/pmc_passes.sh r03b > $O/pmc_bin.log 2>&1
python3 profiles/summarize.py gpurun_out/prof_r03b $O/binary_summary.json > /dev/null
cp $(find gpurun_out/prof_r03b/stats -name "*kernel_stats.csv" | head -1) $O/binary_kernel_stats.csv
echo headline done
for v in plain n4 bistro; do
  P=32; B=32; if [ $v = bistro ]; then P=8; B=8; fi
  bash tools/prof_render.sh r03_$v $v $B $P
  cp gpurun_out/timeline_r03_$v.txt $O/render_timeline_$v.txt
  cp $(find gpurun_out/trace_r03_$v -name "*kernel_stats.csv" | head -1) $O/render_kernel_stats_$v.csv
  echo render trace $v done
done
cp $O/render_kernel_stats_plain.csv $O/render_kernel_stats.csv
for v in plain n4 bistro; do
  rm -rf gpurun_out/pmc_render
  bash tools/pmc_render.sh $v > /dev/null 2>&1
  python3 profiles/summarize.py gpurun_out/pmc_render $O/render_counters_$v.json > /dev/null
  echo render pmc $v done
done
timeout -k 5 60 tools/ubench/inst_ubench > $O/inst_ubench.json || true
echo all done
