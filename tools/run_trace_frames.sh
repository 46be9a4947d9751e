#!/bin/bash
# kernel-by-kernel timeline of a few steady-state frames: gpurun -- 'bash tools/run_trace_frames.sh <first> <count> [bench args]'
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/trace_fr
FIRST=${1:-200}; COUNT=${2:-3}; shift; shift
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 $ROOT/bench.py --no-cpu-baseline --no-secondary $*"
timeout 600 $B > $OUT/prerender.log 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT -o tr -- $B > $OUT/log.txt 2>&1
f=$(ls $OUT/*kernel_trace.csv $OUT/*/*kernel_trace.csv 2>/dev/null | tail -1)
python3 $ROOT/tools/tools_trace_frames.py $f $FIRST $COUNT all > $OUT/timeline.txt 2>&1
grep -h "^{" $OUT/log.txt | tail -1 | cut -c1-120
rm -f $f
