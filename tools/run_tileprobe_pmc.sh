#!/bin/bash
# gpurun -- 'bash tools/run_tileprobe_pmc.sh': SQ counters of the tile kernels alone (tools/tileprobe.hip), two passes.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/tp_pmc
rm -rf $OUT; mkdir -p $OUT
cd $ROOT
python3 -c "
import bench
k, L, R = bench._render_one((bench.CONFIGS[1], 2, 30, 80))
L.tofile('/tmp/frame.raw')
"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I visual_odometry_ros_amd/csrc tools/tileprobe.hip -o /tmp/tp0 || exit 1
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $OUT/a -o sq -- /tmp/tp0 1241 376 /tmp/frame.raw > $OUT/a.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_INSTS_SMEM --kernel-trace --output-format csv -d $OUT/b -o sq -- /tmp/tp0 1241 376 /tmp/frame.raw > $OUT/b.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH SQ_ACTIVE_INST_MISC SQ_IFETCH SQ_INST_LEVEL_LDS SQ_INSTS_GDS SQ_INSTS_FLAT --kernel-trace --output-format csv -d $OUT/c -o sq -- /tmp/tp0 1241 376 /tmp/frame.raw > $OUT/c.log 2>&1
python3 - <<'PY'
import csv, glob, os, collections
out = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out/tp_pmc"
for sub in ("a", "b", "c"):
    per = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    for path in glob.glob(os.path.join(out, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            c = per[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]]
            c[0] += float(r["Counter_Value"]); c[1] += 1
    for k, cs in per.items():
        print(sub, k[:30], {n: round(v[0] / v[1]) for n, v in cs.items()})
PY
tail -3 $OUT/c.log
