#!/bin/bash
# rocprofv3 kernel trace of the render leg, summarised per kernel and bounce.  usage: tools/prof_render.sh <tag> [variant] [batch] [passes]
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-x}; VAR=${2:-plain}; B=${3:-16}; P=${4:-32}
OUT=$R/gpurun_out/trace_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/tools/render_only.py $P $VAR $B > $OUT/run.log 2>&1
F=$(find $OUT -name "*kernel_trace.csv" | head -1)
python3 $R/tools/timeline.py $F $((P / B)) > $R/gpurun_out/timeline_$TAG.txt
tail -2 $OUT/run.log >> $R/gpurun_out/timeline_$TAG.txt
