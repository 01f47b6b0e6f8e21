#!/bin/bash
# usage (GPU box): tools/pmc_conv.sh <dtype> <n> <hw> <shapes> <lib.so> "<CTR1 CTR2 ...>" ["<CTRs of a second pass>" ...]
# rocprofv3 --pmc passes over tools/ablate.py (one conv layer shape, 6 launches), summary per kernel
set -e
DT=$1; N=$2; HW=$3; SH=$4; L=$5; shift 5
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp
i=0
for CTRS in "$@"; do
  OUT=$R/gpurun_out/pmc_conv/p$i
  rm -rf $OUT; mkdir -p $OUT
  timeout -k 10 200 rocprofv3 --pmc $CTRS --output-format csv -d $OUT -- python3 $R/tools/ablate.py $R/$L --dtype $DT --n $N --hw $HW --reps 6 --shapes $SH > $OUT/run.log 2>&1
  python3 $R/tools/pmc_summary.py $OUT | grep -v "^#" | grep conv3x3
  i=$((i+1))
done
