set -e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/c3trace
rm -rf $OUT; mkdir -p $OUT
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -- python3 $R/bench.py --workload c3-stream --steps 3 --warmup 1 --cpu-crop 0 --no-parity > $OUT/run.log 2>&1
cp $(find $OUT/t -name '*_kernel_trace.csv' | head -1) $OUT/kernel_trace.csv
cd $R
python3 tools/trace_groups.py $OUT/kernel_trace.csv > $OUT/groups.txt
rm -rf $OUT/t $OUT/kernel_trace.csv
grep '^{' $OUT/run.log | cut -c1-400
head -30 $OUT/groups.txt
