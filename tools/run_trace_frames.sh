#!/bin/bash
# kernel-by-kernel timeline of a few steady-state frames: gpurun -- 'bash tools/run_trace_frames.sh <first> <count> [bench args]'
# or, with first = batch:  gpurun -- 'bash tools/run_trace_frames.sh batch <S> [tail_ms]'  — S streams on one GPU through the
# batch driver (tools_batch.py), how full the device is over the last tail_ms of the trace (tools_trace_frames.py busy)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/trace_fr
FIRST=${1:-200}; COUNT=${2:-3}; shift; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
if [ "$FIRST" = batch ]; then
  B="python3 $ROOT/tools/tools_batch.py --S $COUNT --lba 1 --strict 4 --frames 128"
  timeout 600 $B > $OUT/batch_prerender.log 2>&1
  rm -rf $OUT/b; mkdir -p $OUT/b
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/b -o tr -- $B > $OUT/batch_S$COUNT.log 2>&1
  f=$(ls $OUT/b/*kernel_trace.csv $OUT/b/*/*kernel_trace.csv 2>/dev/null | tail -1)
  python3 $ROOT/tools/tools_trace_frames.py $f busy ${1:-40} > $OUT/batch_S${COUNT}_busy.txt 2>&1
  tail -2 $OUT/batch_S$COUNT.log; cat $OUT/batch_S${COUNT}_busy.txt
  rm -rf $OUT/b
  exit 0
fi
rm -rf $OUT; mkdir -p $OUT
B="python3 $ROOT/bench.py --no-cpu-baseline --no-secondary $*"
timeout 600 $B > $OUT/prerender.log 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT -o tr -- $B > $OUT/log.txt 2>&1
f=$(ls $OUT/*kernel_trace.csv $OUT/*/*kernel_trace.csv 2>/dev/null | tail -1)
python3 $ROOT/tools/tools_trace_frames.py $f $FIRST $COUNT all > $OUT/timeline.txt 2>&1
grep -h "^{" $OUT/log.txt | tail -1 | cut -c1-120
rm -f $f
