#!/bin/bash
# usage (GPU box, repo root): tools/gpu_pmc_traffic.sh <tag> [bench args...]
# two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over bench.py; per-kernel summary into gpurun_out/<tag>/
set -e
TAG=$1; shift
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf $OUT/pmc_$C
  timeout -k 10 300 rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_$C -- python3 $R/bench.py --steps 5 --warmup 0 --cpu-crop 0 --inflight 1 --no-extras "$@" > $OUT/pmc_$C.log 2>&1
  python3 $R/tools/pmc_summary.py $OUT/pmc_$C | tee -a $OUT/pmc_fetch_write.txt
done
# bytes per conv launch -> gpurun_out/<tag>/pmc_traffic.json entry (copy into profiles/ by hand: KEY defaults to c2)
cd $R && python3 tools/pmc_traffic_json.py ${KEY:-c2} $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE | tee -a $OUT/pmc_fetch_write.txt && cp profiles/pmc_traffic.json $OUT/pmc_traffic.json
