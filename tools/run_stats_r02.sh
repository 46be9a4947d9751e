#!/bin/bash
# rocprofv3 --kernel-trace --stats of the headline loop alone (the roofline's launch duration must agree with it), in the
# default replay mode (4) and stream-ordered (1):  gpurun -- 'bash tools/run_stats_r02.sh'
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for M in 4 1; do
  OUT=$ROOT/gpurun_out/stats_m$M
  rm -rf $OUT; mkdir -p $OUT
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o bench -- python3 $ROOT/bench.py --no-cpu-baseline --no-secondary --strict-border $M > $OUT/bench.log 2>&1
  grep -h "^{" $OUT/bench.log | tail -1 | cut -c1-120
  grep -o '"avg_launch_us": [0-9.]*' $OUT/bench.log | tail -1
  head -4 $OUT/bench_kernel_stats.csv | cut -c1-120
done
