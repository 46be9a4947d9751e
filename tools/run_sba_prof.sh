set -e
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_sba -o sba -- python3 $GRAFT_REPO_ROOT/tests/measure/sbabench.py > $GRAFT_REPO_ROOT/gpurun_out/prof_sba.log 2>&1
