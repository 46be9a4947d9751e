#!/bin/bash
# kernel-trace timeline of a few frames in a given strict-border mode: gpurun -- 'bash tools/run_trace_m3.sh 3'
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
M=${1:-3}
OUT=$ROOT/gpurun_out/trace_m$M
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT -o tr -- python3 $ROOT/bench.py --no-cpu-baseline --no-secondary --steps 60 --warmup 10 --strict-border $M > $OUT/log.txt 2>&1
f=$(ls $OUT/*kernel_trace.csv $OUT/*/*kernel_trace.csv 2>/dev/null | tail -1)
python3 $ROOT/tools/tools_trace_timeline.py $f 30 26 > $OUT/timeline.txt 2>&1
tail -3 $OUT/timeline.txt
