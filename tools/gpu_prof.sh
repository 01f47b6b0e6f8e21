#!/bin/bash
# usage (on the GPU box, from the repo root): tools/gpu_prof.sh <tag> [bench args...]
# kernel trace + stats of bench.py under rocprofv3, per-layer table printed and saved under gpurun_out/<tag>/
set -e
TAG=$1; shift
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/bench.py --steps 5 --warmup 2 --cpu-crop 0 --inflight 1 --no-extras "$@" > $OUT/run.log 2>&1
grep '^{' $OUT/run.log > $OUT/bench.json || true
TRACE=$(find $OUT -name '*_kernel_trace.csv' | head -1)
python3 $R/tools/trace_layers.py $TRACE | tee $OUT/per_layer.txt
cp $(find $OUT -name '*_kernel_stats.csv' | head -1) $OUT/kernel_stats.csv
rm -f $(find $OUT -name '*_kernel_trace.csv')   # large; the per-layer table is what we keep
