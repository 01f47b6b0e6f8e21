#!/bin/bash
# usage (GPU box, repo root): tools/gpu_pmc.sh <tag> "<CTR1 CTR2 ...>" [bench args...]
# one rocprofv3 --pmc pass over bench.py; per-kernel sums into gpurun_out/<tag>/pmc_<first counter>.txt
set -e
TAG=$1; CTRS=$2; shift 2
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
F=$(echo $CTRS | cut -d' ' -f1)
mkdir -p $OUT
cd /tmp
rm -rf $OUT/pmc_$F
timeout -k 10 300 rocprofv3 --pmc $CTRS --output-format csv -d $OUT/pmc_$F -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-crop 0 --inflight 1 --no-extras "$@" > $OUT/pmc_$F.log 2>&1
python3 $R/tools/pmc_summary.py $OUT/pmc_$F | tee $OUT/pmc_$F.txt
rm -rf $OUT/pmc_$F
