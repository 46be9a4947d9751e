#!/bin/bash
# Round-5 kernel statistics on the GPU box:  gpurun -- 'bash tools/run_stats_r05.sh <tag> [bench args]'
# rocprofv3 --kernel-trace --stats of the bench command (closed loop incl. local BA); the summary goes to
# gpurun_out/r05/<tag>_kernel_stats.csv (copied into profiles/ by hand when it is one to keep).
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-x}
shift
OUT=$ROOT/gpurun_out/r05
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 $ROOT/bench.py --no-cpu-baseline --no-secondary $*"
timeout 600 $B > $OUT/${TAG}_prerender.log 2>&1   # (the same command first: its stream goes into the render cache OUTSIDE the profiler)
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$TAG -o bench -- $B > $OUT/${TAG}_stats.log 2>&1
cp $OUT/stats_$TAG/*/bench_kernel_stats.csv $OUT/${TAG}_kernel_stats.csv 2>/dev/null || cp $OUT/stats_$TAG/bench_kernel_stats.csv $OUT/${TAG}_kernel_stats.csv
grep -h "^{" $OUT/${TAG}_stats.log | tail -1 | cut -c1-200
head -30 $OUT/${TAG}_kernel_stats.csv | cut -c1-160
rm -rf $OUT/stats_$TAG
