#!/bin/bash
# gpurun -- '[CFG=4] bash tools/run_tileprobe.sh [extra hipcc flags]': the tile kernels of the ingestion chain alone on the GPU, on a
# frame of the bench's renderer (CFG = BASELINE config index: 1 = 1241 x 376, 4 = 3840 x 2160): launch times and, in a second
# build with -DTILE_STAMP, one workgroup's phases.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
CFG=${CFG:-1}
cd $ROOT
WH=$(python3 -c "
import bench
cfg = bench.CONFIGS[$CFG]
k, L, R = bench._render_one((cfg, 2, 30, 80))
L.tofile('/tmp/frame.raw')
print(cfg['W'], cfg['H'])
" | tail -1)
F="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I visual_odometry_ros_amd/csrc $*"
hipcc $F tools/tileprobe.hip -o /tmp/tp0 && /tmp/tp0 $WH /tmp/frame.raw
hipcc $F -DTILE_STAMP=1 tools/tileprobe.hip -o /tmp/tp1 && /tmp/tp1 $WH /tmp/frame.raw | grep phases
