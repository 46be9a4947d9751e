#!/bin/bash
# HBM traffic counters (separate --pmc passes, counters only with --kernel-trace), run on the GPU box:
#   gpurun -- 'bash tools/run_pmc.sh'
set -e
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/cal_$C -o cal -- $ROOT/tools/pmccal > $OUT/cal_$C.log 2>&1
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/bench_$C -o bench -- python3 $ROOT/bench.py --no-cpu-baseline --strict-border 1 --steps 60 --warmup 10 > $OUT/bench_$C.log 2>&1
done
ls -R $OUT | head -40
