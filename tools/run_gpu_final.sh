set -e
python -m pytest tests -m gpu -x -q 2>&1 | tail -2
python bench.py 2>&1 | tail -1
python bench.py --no-cpu-baseline --strict-border 0 2>&1 | tail -1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_d -o r01d -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_d.log 2>&1
tail -1 $GRAFT_REPO_ROOT/gpurun_out/prof_d.log | cut -c1-120
