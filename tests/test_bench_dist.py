"""N > 1 path of bench.py on CPU: two gloo ranks, one independent stream each, one
all_gather of {frames, seconds}; value = total frames / max seconds."""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import os, sys, json
    sys.path.insert(0, %r)
    import torch.distributed as dist
    import bench
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group(backend="gloo")
    frames, seconds = 100 + rank, 0.5 + 0.25 * rank
    tot, mx = bench.aggregate(frames, seconds, world)
    dist.barrier()
    print(json.dumps({"rank": rank, "tot": tot, "max": mx, "seed": bench.stream_seed(rank)}), flush=True)
    dist.destroy_process_group()
""") % ROOT


def test_two_rank_aggregation_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29571", WORLD_SIZE="2")
    procs = []
    for r in range(2):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=e, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        o, err = p.communicate(timeout=120)
        assert p.returncode == 0, err[-2000:]
        outs.append(o.strip().splitlines()[-1])
    import json
    res = sorted((json.loads(o) for o in outs), key=lambda d: d["rank"])
    for d in res:
        assert d["tot"] == 201.0 and d["max"] == 0.75  # every rank sees the whole-job totals
    assert res[0]["seed"] != res[1]["seed"]  # independent streams


def test_single_rank_aggregation_is_identity():
    sys.path.insert(0, ROOT)
    import bench
    assert bench.aggregate(7, 0.5, 1) == (7.0, 0.5)
