#!/bin/bash
# SQ counter passes of round 4 (the two passes of run_profiles_r04.sh that the first attempt lost: the stream of --steps 40
# --warmup 10 was not in the render cache, and a renderer pool forked under rocprofv3 --pmc does not come back).
# gpurun -- 'bash tools/run_sq_r04.sh'
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r04
mkdir -p $OUT/sq
cd /tmp && export TMPDIR=/tmp
B="python3 $ROOT/bench.py --no-cpu-baseline --no-secondary"
timeout 600 $B --strict-border 1 --steps 40 --warmup 10 > $OUT/sq/prerender.log 2>&1   # renders + caches the stream outside the profiler
timeout 900 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $OUT/sq/a -o sq -- $B --strict-border 1 --steps 40 --warmup 10 > $OUT/sq/a.log 2>&1
timeout 900 rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU --kernel-trace --output-format csv -d $OUT/sq/b -o sq -- $B --strict-border 1 --steps 40 --warmup 10 > $OUT/sq/b.log 2>&1
ls $OUT/sq/a $OUT/sq/b | head; grep -h "^{" $OUT/sq/a.log | tail -1 | cut -c1-160
