#!/bin/bash
# usage (GPU box, repo root, after tools/gpu_round_profiles.sh <tag> in the SAME call so that profiles/pmc_traffic.json keeps its c2 key):
#   tools/gpu_round_profiles_4k.sh <tag>
# FETCH_SIZE / WRITE_SIZE passes of the 4K workloads (c3-stream, c4) -> keys c3 / c4 of pmc_traffic.json; MFMA-busy / LDS pass of c3-stream
set -e
TAG=$1
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
KEY=c3 bash tools/gpu_pmc_traffic.sh $TAG/c3 --workload c3-stream
KEY=c4 bash tools/gpu_pmc_traffic.sh $TAG/c4 --workload c4
bash tools/gpu_pmc.sh $TAG/c3 "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT" --workload c3-stream
cp profiles/pmc_traffic.json gpurun_out/$TAG/pmc_traffic.json
