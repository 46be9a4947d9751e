#!/bin/bash
# Round-2 profiles, run on the GPU box:  gpurun -- 'bash tools/run_profiles_r02.sh <tag>'
#  1. rocprofv3 --kernel-trace --stats of the DEFAULT bench command (the summary the roofline's launch duration must agree with)
#  2. HBM traffic counters in their own passes (--pmc with --kernel-trace only), FETCH_SIZE and WRITE_SIZE separately;
#     counter collection serialises kernels across queues, so these passes use the stream-ordered replay (--strict-border 1)
# Summaries are copied to gpurun_out/<tag>_*; tools/pmc_summarize.py makes the JSON under profiles/ afterwards.
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT/pmc
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python3 $ROOT/bench.py --no-cpu-baseline > $OUT/stats_bench.log 2>&1
tail -1 $OUT/stats_bench.log | cut -c1-300
if [ ! -x $ROOT/tools/pmccal ]; then /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 $ROOT/tools/pmccal.hip -o $ROOT/tools/pmccal > $OUT/pmccal_build.log 2>&1; fi
for C in FETCH_SIZE WRITE_SIZE; do
  [ -x $ROOT/tools/pmccal ] && timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc/cal_$C -o cal -- $ROOT/tools/pmccal > $OUT/pmc/cal_$C.log 2>&1
  timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc/bench_$C -o bench -- python3 $ROOT/bench.py --no-cpu-baseline --strict-border 1 --no-secondary --steps 60 --warmup 10 > $OUT/pmc/bench_$C.log 2>&1
done
ls $OUT/stats | head; ls $OUT/pmc | head -20
