#!/bin/bash
# Counter passes of BASELINE configs[4] (3840x2160, 8000 features; bench.py --config 4 = round 2's closed workload), on the
# GPU box:  gpurun -- 'bash tools/run_counters_cfg4.sh'  -> then tools/pmc_summarize.py / sq_summarize.py with tag r03_cfg4_closed
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r03
mkdir -p $OUT/pmc4 $OUT/sq4
cd /tmp && export TMPDIR=/tmp
B="python3 $ROOT/bench.py --config 4 --no-cpu-baseline --no-secondary --strict-border 1 --steps 30 --warmup 5"
timeout 900 $B > $OUT/pmc4/prerender.log 2>&1
if [ ! -x $ROOT/tools/pmccal ]; then /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 $ROOT/tools/pmccal.hip -o $ROOT/tools/pmccal > $OUT/pmccal_build.log 2>&1; fi
for C in FETCH_SIZE WRITE_SIZE; do
  [ -x $ROOT/tools/pmccal ] && timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc4/cal_$C -o cal -- $ROOT/tools/pmccal > $OUT/pmc4/cal_$C.log 2>&1
  for attempt in 1 2; do
    timeout 900 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc4/bench_$C -o bench -- $B > $OUT/pmc4/bench_$C.log 2>&1
    ls $OUT/pmc4/bench_$C/*counter_collection.csv > /dev/null 2>&1 && break
  done
done
timeout 900 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $OUT/sq4/a -o sq -- $B > $OUT/sq4/a.log 2>&1
timeout 900 rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU --kernel-trace --output-format csv -d $OUT/sq4/b -o sq -- $B > $OUT/sq4/b.log 2>&1
grep -h "^{" $OUT/pmc4/prerender.log | tail -1 | cut -c1-160
ls $OUT/pmc4 $OUT/sq4/a $OUT/sq4/b | head -30
