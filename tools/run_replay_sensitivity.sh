#!/bin/bash
# How much of a replayed IC iteration's cost is on the frame's critical path? The library is rebuilt ON THE GPU BOX with every
# replayed iteration made longer by s_sleep(N) (64 cycles each, ~27 ns at 2.4 GHz) and the default loop is timed again:
# d(ordinary frame) / d(iteration cost) x (what a faster iteration would save) bounds what the helper-wavefront replay of
# DESIGN §9 could give.   gpurun -- 'bash tools/run_replay_sensitivity.sh'
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r04
mkdir -p $OUT
cd $ROOT
B="python3 bench.py --no-cpu-baseline --no-secondary"
$B --steps 20 --warmup 5 > /dev/null 2>&1   # renders + caches the stream
for N in 0 10 19 38; do
  if [ $N -gt 0 ]; then export VO_EXTRA_FLAGS="-DIC_REPLAY_EXTRA_SLEEP=$N"; else unset VO_EXTRA_FLAGS; fi
  python3 -c "from visual_odometry_ros_amd import build as B; B.build()" > $OUT/sens_build_$N.log 2>&1
  for strict in 4 1; do
    $B --strict-border $strict > $OUT/sens_${N}_m$strict.json 2> $OUT/sens_${N}_m$strict.err
    python3 - <<PY
import json
d=json.loads(open("$OUT/sens_${N}_m$strict.json").read().strip().splitlines()[-1])
print("sleep $N strict $strict:", d["value"], d["frame_ms_by_kind"]["mean_ms_other"], d["frame_ms_by_kind"]["mean_ms_keyframe"], d["frame_ms_by_kind"]["mean_replayed_features"])
PY
  done
done
unset VO_EXTRA_FLAGS
python3 -c "from visual_odometry_ros_amd import build as B; B.build()" > $OUT/sens_build_restore.log 2>&1
