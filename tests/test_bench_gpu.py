"""bench.py end to end on one GPU, shortened: the JSON line carries what the contract asks for (metric / value / roofline /
cpu_baseline / parity / secondary legs) and its own parity checks hold."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_default_bench_line_shortened():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "40", "--warmup", "5", "--cpu-frames", "2",
                        "--parity-frames", "13", "--streams-per-gpu", "2", "--batch-frames", "14"],
                       capture_output=True, text=True, timeout=1500, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["metric"].startswith("stereo VO frames/sec @1241x376") and d["unit"] == "frames/s" and d["value"] > 100
    assert (d["n_gpus"], d["steps"], d["warmup"], d["higher_is_better"], d["scaling"], d["vs_baseline"]) == (1, 40, 5, True, "weak", None)
    assert d["config"]["track_set"] == "closed loop" and d["config"]["playback"] == "forward" and d["config"]["local_ba"] is True
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["peak"] == 8000.0 and 0 < rf["frac"] < 1 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-4
    assert set(rf["alg_bytes_per_launch"]) == {"survey_8d_required", "survey_8d_speculative", "design_records"}
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["value"] > 0 and cb["cores"] >= 1 and "sample" in cb
    p = d["parity"]
    assert p["track_ids_bit_exact"] and p["track_sets_and_poses_bit_exact"] and p["pose_rel_frobenius_max"] < 1e-4
    assert p["frames_checked"] == 13 and p["local_ba_solves_checked"] >= 1
    s = d["secondary"]
    assert s["r02_ground_truth_track_sets_back_and_forth"]["value"] > 100 and s["class_surface"]["survivors_equal_fused_operator"]
    assert s["loop_host_images"]["poses_equal_resident_run"] and s["loop_host_images"]["value"] > 100
    assert s["streams_per_gpu"]["2"]["poses_and_track_ids_equal_single_stream_run"]
    assert d["loop"]["keyframes"] >= 5 and d["loop"]["lba_runs"] >= 3 and d["loop"]["end_point_error_m"] < 1.0


@pytest.mark.parametrize("flags", [["--steps", "5", "--warmup", "0"], ["--steps", "1", "--warmup", "0", "--no-secondary"],
                                   ["--config", "2", "--steps", "2", "--warmup", "0"], ["--mode", "closed", "--steps", "2", "--warmup", "0"]])
def test_bench_with_very_few_steps(flags):
    """Whatever K and W the caller picks: one JSON line (a warm-up of zero frames once left a frame in flight)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--cpu-frames", "2", "--parity-frames", "4", "--streams-per-gpu", ""] + flags,
                       capture_output=True, text=True, timeout=1200, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["value"] > 0 and d["steps"] == int(flags[flags.index("--steps") + 1]) and "roofline" in d


def test_rccl_process_group_and_gather_with_one_rank():
    """The N > 1 path of bench.py — init_process_group("nccl") with the rendezvous time-out, dist.barrier around the timed region,
    the device-tensor all_gather of gather_ranks, destroy_process_group — executed on THIS one GPU with a world of one rank
    (--force-collective; a fresh child process, nothing re-exec'ed): the first RCCL communicator this code creates is not the one
    on the driver's 8-GPU node. N > 1 itself stays unmeasured on hardware."""
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-collective", "--steps", "2", "--warmup", "0",
                        "--no-secondary", "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["collective"] == "nccl" and d["n_gpus"] == 1 and len(d["per_rank_fps"]) == 1
    assert abs(d["per_rank_fps"][0] - d["value"]) <= 0.02 * d["value"]
