#!/bin/bash
# SQ counters of the frame kernel (counters only with --kernel-trace): gpurun -- 'bash tools/run_pmc_sq.sh'
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmc_sq
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $OUT/a -o sq -- python3 $ROOT/bench.py --no-cpu-baseline --strict-border 1 --no-secondary --steps 40 --warmup 10 > $OUT/a.log 2>&1
timeout 600 rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU --kernel-trace --output-format csv -d $OUT/b -o sq -- python3 $ROOT/bench.py --no-cpu-baseline --strict-border 1 --no-secondary --steps 40 --warmup 10 > $OUT/b.log 2>&1
tail -2 $OUT/a.log | cut -c1-200; tail -2 $OUT/b.log | cut -c1-200
ls $OUT/a $OUT/b
