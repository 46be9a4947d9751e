#!/bin/bash
# A/B of measurement builds on the GPU box: gpurun -- 'bash tools/run_variants.sh "<flags A>" "<flags B>" ...'
# every variant: rebuild libvo_hip.so with VO_EXTRA_FLAGS, then the headline loop and the no-look-ahead loop (device / host images)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/variants
mkdir -p $OUT
cd $ROOT
B="python3 bench.py --no-cpu-baseline --no-secondary --steps ${STEPS:-200}"
i=0
for F in "$@"; do
  i=$((i+1))
  VO_EXTRA_FLAGS="$F" python3 -c "import __graft_entry__ as g; g.build()" > $OUT/build_$i.log 2>&1 || { echo "build $i failed"; tail -5 $OUT/build_$i.log; continue; }
  for leg in "" "--no-prefetch --no-issue-ahead" "--no-prefetch --no-issue-ahead --host-images"; do
    VO_EXTRA_FLAGS="$F" timeout 600 $B $leg > $OUT/v${i}.json 2> $OUT/v${i}.err
    python3 - "$F" "$leg" $OUT/v${i}.json <<'P'
import json, sys
try:
    d = json.loads(open(sys.argv[3]).read().strip().splitlines()[-1])
    k = d["frame_ms_by_kind"]
    print(f"[{sys.argv[1]}] [{sys.argv[2]}] {d['value']:.0f} fps  ordinary {k['mean_ms_other']:.4f} keyframe {k['mean_ms_keyframe']:.4f}  frame kernel {d['roofline']['avg_launch_us']:.1f} us")
except Exception as e:
    print(f"[{sys.argv[1]}] [{sys.argv[2]}] failed: {e}")
P
  done
done
