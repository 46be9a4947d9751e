set -e
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_orb -o orb -- python3 $GRAFT_REPO_ROOT/tests/measure/orbbench.py > $GRAFT_REPO_ROOT/gpurun_out/prof_orb.log 2>&1
