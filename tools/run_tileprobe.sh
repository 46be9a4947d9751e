#!/bin/bash
# gpurun -- 'bash tools/run_tileprobe.sh [extra hipcc flags]': the tile kernels of the ingestion chain alone on the GPU, on a
# frame of the bench's renderer (1241 x 376): launch times and, in a second build with -DTILE_STAMP, one workgroup's phases.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
python3 -c "
import bench
k, L, R = bench._render_one((bench.CONFIGS[1], 2, 30, 80))
L.tofile('/tmp/frame.raw')
"
F="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I visual_odometry_ros_amd/csrc $*"
hipcc $F tools/tileprobe.hip -o /tmp/tp0 && /tmp/tp0 1241 376 /tmp/frame.raw
hipcc $F -DTILE_STAMP=1 tools/tileprobe.hip -o /tmp/tp1 && /tmp/tp1 1241 376 /tmp/frame.raw | grep phases
