set -e
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python bench.py --no-cpu-baseline 2>&1 | tail -1
python bench.py --no-cpu-baseline --strict-border 0 2>&1 | tail -1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_c -o r01c -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 100 --warmup 20 > $GRAFT_REPO_ROOT/gpurun_out/prof_c.log 2>&1
