#!/bin/bash
# Profiles of one round on the GPU box (one script for all of them; the per-round copies of earlier rounds are gone):
#   gpurun -- 'bash tools/run_profiles.sh <round tag, e.g. r05> <what> [bench args]'
# what = stats     rocprofv3 --kernel-trace --stats of `bench.py [bench args]` -> gpurun_out/<round>/<name>_kernel_stats.csv
#                  (name = NAME from the environment, default "default")
#        counters  HBM traffic (--pmc FETCH_SIZE / WRITE_SIZE in separate passes, with --kernel-trace only) and the two SQ passes
#                  of `bench.py --strict-border 1 [bench args]` (counter collection serialises kernels across queues: the
#                  stream-ordered replay is the one that can be profiled) -> gpurun_out/<round>/{pmc,sq}<NAME>/; then
#                  tools/pmc_summarize.py / sq_summarize.py stamp them with the kernel-source hash bench.py checks.
# Every stream length is rendered once OUTSIDE the profiler first (a renderer pool forked under rocprofv3 --pmc does not come back).
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
RND=${1:-r05}; WHAT=${2:-stats}; shift; shift
NAME=${NAME:-default}
OUT=$ROOT/gpurun_out/$RND
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 $ROOT/bench.py --no-cpu-baseline --no-secondary $*"
if [ "$WHAT" = stats ]; then
  timeout 900 $B > $OUT/${NAME}_prerender.log 2>&1
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$NAME -o bench -- $B > $OUT/${NAME}_stats.log 2>&1
  cp $OUT/stats_$NAME/*/bench_kernel_stats.csv $OUT/${NAME}_kernel_stats.csv 2>/dev/null || cp $OUT/stats_$NAME/bench_kernel_stats.csv $OUT/${NAME}_kernel_stats.csv
  grep -h "^{" $OUT/${NAME}_stats.log | tail -1 | cut -c1-200
  head -24 $OUT/${NAME}_kernel_stats.csv | cut -c1-150
  rm -rf $OUT/stats_$NAME
  exit 0
fi
B="$B --strict-border 1"
P=$OUT/pmc_$NAME; S=$OUT/sq_$NAME
mkdir -p $P $S
timeout 900 $B > $P/prerender.log 2>&1
if [ ! -x $ROOT/tools/pmccal ]; then /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 $ROOT/tools/pmccal.hip -o $ROOT/tools/pmccal > $OUT/pmccal_build.log 2>&1; fi
for C in FETCH_SIZE WRITE_SIZE; do
  [ -x $ROOT/tools/pmccal ] && timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $P/cal_$C -o cal -- $ROOT/tools/pmccal > $P/cal_$C.log 2>&1
  for attempt in 1 2; do  # (a pass has been seen to end at once without output: once more then)
    timeout 900 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $P/bench_$C -o bench -- $B > $P/bench_$C.log 2>&1
    ls $P/bench_$C/*counter_collection.csv > /dev/null 2>&1 && break
  done
done
timeout 900 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $S/a -o sq -- $B > $S/a.log 2>&1
timeout 900 rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU --kernel-trace --output-format csv -d $S/b -o sq -- $B > $S/b.log 2>&1
grep -h "^{" $P/prerender.log | tail -1 | cut -c1-160
for d in $P/bench_FETCH_SIZE $P/bench_WRITE_SIZE $P/cal_FETCH_SIZE $P/cal_WRITE_SIZE $S/a $S/b; do  # (hostname sub-directory or not: flatten)
  for f in $d/*/*counter_collection.csv; do [ -f "$f" ] && mv $f $d/; done
  rm -f $d/*kernel_trace.csv $d/*/*kernel_trace.csv $d/*agent_info.csv $d/*/*agent_info.csv
done
ls $P $P/bench_FETCH_SIZE $S/a $S/b | head -30
