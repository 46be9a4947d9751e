#!/bin/bash
# Round-3 kernel statistics only (no counter passes), on the GPU box:  gpurun -- 'bash tools/run_stats_r03.sh'
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r03
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 $ROOT/bench.py --no-cpu-baseline --no-secondary"
timeout 600 $B --steps 20 --warmup 5 > $OUT/prerender.log 2>&1   # the stream is rendered (and cached) outside the profiler
for v in "default:" "mode1:--strict-border 1" "nolba:--lba 0"; do
  tag=${v%%:*}; extra=${v#*:}
  rm -rf $OUT/stats_$tag
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$tag -o bench -- $B $extra > $OUT/stats_$tag.log 2>&1
  grep -h "^{" $OUT/stats_$tag.log | tail -1 | cut -c1-140
  head -5 $OUT/stats_$tag/bench_kernel_stats.csv | cut -c1-120
done
