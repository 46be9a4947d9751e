#!/bin/bash
# A/B of the per-landmark kernel of the local BA: landmark order (VO_SBA_NO_ORDER) and poses in LDS (VO_SBA_NO_LDS_T)
cd /tmp && export TMPDIR=/tmp
run() {
  rm -rf /tmp/abp
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abp -o b -- python3 $GRAFT_REPO_ROOT/bench.py --steps 60 --warmup 10 --lba 1 --no-secondary --no-cpu-baseline > /dev/null 2>&1
  grep -E "sba_update_point" /tmp/abp/b_kernel_stats.csv | cut -d, -f1-4
}
echo "order + lds"; run
echo "no order, lds"; VO_SBA_NO_ORDER=1 run
echo "order, no lds"; VO_SBA_NO_LDS_T=1 run
echo "no order, no lds"; VO_SBA_NO_ORDER=1 VO_SBA_NO_LDS_T=1 run
