#!/bin/bash
# usage (GPU box, repo root): tools/gpu_round_profiles.sh <tag>
# everything profiles/<round>/ holds for the c2 workload, in one call:
#   1. rocprofv3 --kernel-trace --stats of the DEFAULT bench command (4 frames in flight: per-layer kernels)
#   2. the same for one frame at a time (--inflight 1: the fused dense-block kernel), with the per-kernel table
#   3. FETCH_SIZE / WRITE_SIZE passes of (2) -> profiles/pmc_traffic.json (stamped with the kernel sources' hash)
#   4. SQ_VALU_MFMA_BUSY_CYCLES / GRBM_GUI_ACTIVE pass of (2)
set -e
TAG=$1
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/default -- python3 $R/bench.py --steps 5 --warmup 2 --cpu-crop 0 --no-extras > $OUT/default_run.log 2>&1
cp $(find $OUT/default -name '*_kernel_stats.csv' | head -1) $OUT/default_kernel_stats.csv
grep '^{' $OUT/default_run.log > $OUT/default_bench.json || true
rm -rf $OUT/default
echo "== default bench command: kernel stats"; head -12 $OUT/default_kernel_stats.csv
cd $R
bash tools/gpu_prof.sh $TAG/single
bash tools/gpu_pmc_traffic.sh $TAG/single
bash tools/gpu_pmc.sh $TAG/single "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"
