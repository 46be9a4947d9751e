#!/bin/bash
# Round-3 profiles, run on the GPU box:  gpurun -- 'bash tools/run_profiles_r04.sh'
#  1. rocprofv3 --kernel-trace --stats of the DEFAULT bench command (closed loop incl. local BA, replay mode 4), of the
#     same with the stream-ordered replay, and of the loop without the local BA
#  2. HBM traffic counters in their own passes (--pmc with --kernel-trace only), FETCH_SIZE and WRITE_SIZE separately;
#     counter collection serialises kernels across queues, so these passes use the stream-ordered replay (--strict-border 1)
#  3. SQ counters (two passes of 8)
# tools/pmc_summarize.py / tools/sq_summarize.py turn 2. and 3. into profiles/r04_frame_*.json (stamped with the kernel
# source hash bench.py checks).
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r04
mkdir -p $OUT/pmc $OUT/sq
cd /tmp && export TMPDIR=/tmp
B="python3 $ROOT/bench.py --no-cpu-baseline --no-secondary"
# the stream is rendered once (worker pool, before anything touches the GPU) and cached under /tmp: do that outside the profiler
timeout 600 $B > $OUT/prerender.log 2>&1                      # (the default command's own stream length first)
timeout 600 $B --steps 20 --warmup 5 >> $OUT/prerender.log 2>&1
timeout 600 $B --strict-border 1 --steps 60 --warmup 10 >> $OUT/prerender.log 2>&1   # (every stream length used below goes into the render cache outside the profiler:
timeout 600 $B --strict-border 1 --steps 40 --warmup 10 >> $OUT/prerender.log 2>&1   #  a renderer pool forked under rocprofv3 --pmc does not come back)
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_default -o bench -- $B > $OUT/stats_default.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_mode1 -o bench -- $B --strict-border 1 > $OUT/stats_mode1.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_nolba -o bench -- $B --lba 0 > $OUT/stats_nolba.log 2>&1
for f in default mode1 nolba; do grep -h "^{" $OUT/stats_$f.log | tail -1 | cut -c1-140; head -6 $OUT/stats_$f/bench_kernel_stats.csv | cut -c1-150; done
if [ ! -x $ROOT/tools/pmccal ]; then /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 $ROOT/tools/pmccal.hip -o $ROOT/tools/pmccal > $OUT/pmccal_build.log 2>&1; fi
for C in FETCH_SIZE WRITE_SIZE; do
  [ -x $ROOT/tools/pmccal ] && timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc/cal_$C -o cal -- $ROOT/tools/pmccal > $OUT/pmc/cal_$C.log 2>&1
  for attempt in 1 2; do  # (a pass has been seen to end at once without output: once more then)
    timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc/bench_$C -o bench -- $B --strict-border 1 --steps 60 --warmup 10 > $OUT/pmc/bench_$C.log 2>&1
    ls $OUT/pmc/bench_$C/*counter_collection.csv > /dev/null 2>&1 && break
  done
done
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $OUT/sq/a -o sq -- $B --strict-border 1 --steps 40 --warmup 10 > $OUT/sq/a.log 2>&1
timeout 600 rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU --kernel-trace --output-format csv -d $OUT/sq/b -o sq -- $B --strict-border 1 --steps 40 --warmup 10 > $OUT/sq/b.log 2>&1
ls $OUT/pmc $OUT/sq/a $OUT/sq/b | head -30
