#!/bin/bash
# usage (GPU box, repo root): tools/ablate.sh <dtype> <n> <hw> <lib1.so> [lib2.so ...]
# one rocprofv3 kernel-trace run of tools/ablate.py per library; prints the per-shape kernel times (us)
set -e
DT=$1; N=$2; HW=$3; shift 3
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp
for L in "$@"; do
  TAG=$(basename $L .so)
  OUT=$R/gpurun_out/abl/$TAG
  rm -rf $OUT; mkdir -p $OUT
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $R/tools/ablate.py $R/$L --dtype $DT --n $N --hw $HW --reps 6 > $OUT/run.log 2>&1
  TRACE=$(find $OUT -name '*_kernel_trace.csv' | head -1)
  echo "$TAG: $(python3 $R/tools/trace_seq.py $TRACE 6)" | tee -a $R/gpurun_out/abl/summary.txt
  rm -f $TRACE
done
