"""N > 1 path of bench.py on CPU (gloo): `bench.py --gpus 2` has to start two ranks by itself, give each its own
stream, gather once and print ONE JSON line on rank 0; and the gather used by the timed run must give every rank the
whole-job totals."""
import json
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _clean_env():
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "LOCAL_WORLD_SIZE")}
    return env


def test_gpus_2_launches_two_ranks_itself():
    """No launcher environment: bench.py spawns the ranks (fresh processes, before any GPU call)."""
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--rendezvous-check"], env=_clean_env(),
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines  # one JSON line, from rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and len(d["per_rank"]) == 2
    assert d["per_rank"][0]["stream_seed"] != d["per_rank"][1]["stream_seed"]  # independent streams
    assert [r["frames"] for r in d["per_rank"]] == [100.0, 101.0]
    assert d["value"] == 201.0 / 0.75  # total frames / max seconds over ranks


def test_ranks_get_disjoint_cpu_sets():
    """N > 1: every rank pins itself to a contiguous share of the visible CPUs before it renders or polls
    (--rendezvous-check reports what each rank ended up with)."""
    if len(os.sched_getaffinity(0)) < 2:
        import pytest
        pytest.skip("one visible CPU")
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--rendezvous-check"], env=_clean_env(),
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    d = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][0])
    a, b = (set(r["cpus"]) for r in d["per_rank"])
    assert a and b and not (a & b)
    assert (a | b) <= set(os.sched_getaffinity(0))


def test_a_rank_that_never_reaches_the_rendezvous_ends_the_job():
    """Rank 1 hangs in front of init_process_group: rank 0's rendezvous times out (--rendezvous-timeout), the launcher
    ends every child and returns non-zero with the ranks' stderr — well before this test's own limit."""
    import time
    env = dict(_clean_env(), BENCH_TEST_HANG_RANK="1")
    t0 = time.time()
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--rendezvous-check", "--rendezvous-timeout", "8"], env=env,
                       capture_output=True, text=True, timeout=200)
    assert p.returncode != 0
    assert time.time() - t0 < 90
    assert "[rank 0]" in p.stderr or "did not finish" in p.stderr, p.stderr[-1500:]


def test_gpus_1_needs_no_rendezvous():
    p = subprocess.run([sys.executable, BENCH, "--gpus", "1", "--rendezvous-check"], env=_clean_env(),
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    d = json.loads(p.stdout.strip().splitlines()[-1])
    assert d["n_gpus"] == 1 and len(d["per_rank"]) == 1


def test_under_torch_distributed_run():
    """The driver's launch line: torch.distributed.run owns the ranks, bench.py must not spawn again."""
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29573", BENCH, "--gpus", "2",
                        "--rendezvous-check"], env=_clean_env(), capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    assert json.loads(lines[0])["n_gpus"] == 2


def test_world_size_mismatch_is_an_error():
    env = dict(_clean_env(), WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29574")
    p = subprocess.run([sys.executable, BENCH, "--gpus", "4", "--rendezvous-check"], env=env, capture_output=True,
                       text=True, timeout=120)
    assert p.returncode != 0 and "WORLD_SIZE" in p.stderr


def test_a_failing_rank_ends_the_job(tmp_path):
    """A rank that dies must not leave the others waiting in the rendezvous."""
    sys.path.insert(0, ROOT)
    script = tmp_path / "driver.py"
    script.write_text(textwrap.dedent("""
        import sys, os
        sys.path.insert(0, %r)
        import bench
        bench.__file__ = %r
        sys.exit(bench.launch_ranks(2, ["--gpus", "2", "--rendezvous-check", "--no-such-flag"]))
    """) % (ROOT, BENCH))
    p = subprocess.run([sys.executable, str(script)], env=_clean_env(), capture_output=True, text=True, timeout=120)
    assert p.returncode != 0


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
    env = dict(_clean_env(), MASTER_ADDR="127.0.0.1", MASTER_PORT="29571", WORLD_SIZE="2")
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
    res = sorted((json.loads(o) for o in outs), key=lambda d: d["rank"])
    for d in res:
        assert d["tot"] == 201.0 and d["max"] == 0.75  # every rank sees the whole-job totals
    assert res[0]["seed"] != res[1]["seed"]  # independent streams


def test_single_rank_aggregation_is_identity():
    sys.path.insert(0, ROOT)
    import bench
    assert bench.aggregate(7, 0.5, 1) == (7.0, 0.5)
