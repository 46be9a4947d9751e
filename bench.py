#!/usr/bin/env python3
"""bench.py — VO frames/s of the hot path on MI355X, one image stream per GPU.

Default workload = BASELINE.json configs[1]: synthetic KITTI-shaped stereo stream, 1241x376, 1500 tracked features
per frame (60x25 buckets), win 21, max_level 6 (5 effective levels), thresholds of config/stereo/kitti_00_stereo.yaml.
A "step" is ONE steady-state stereo frame through the whole operator sequence of StereoVO::trackStereoImages
(stereo_vo.cpp:483-711), in the reference's order of dependencies:
    pyramids of the new pair -> [3] priors -> [4] trackWithPrior l0->l1 -> [4-1] trackWithScale -> [5] trackWithPrior
    l1->r1 -> [6] stereo pose-only BA on the triangulated survivors -> [7] gate -> [10] updateWeightBin(survivors),
    extractORBwithBinning_fast(I1_left) (keypoint detection + per-bucket arg-max), trackBidirection of the new points
with the result (pose, survivor stages, pixels, new points) read back to the host EVERY frame, as a sequential VO
needs it before it can go on. Step [10] is closed on the device: nothing is known to the operator
ahead of the frame that the reference does not know either.

  python bench.py --gpus N --steps K --warmup W            (the driver's contract)
  python bench.py --mode {loop,closed,sequential,open}     loop (default): the CLOSED LOOP — a forward-driving stream (every
                                                           frame rendered once, no playback), the track set carried on the
                                                           device from frame to frame by StereoVO (vo_svo_*): stage-4 survivors
                                                           + the new landmarks of step [10] (DLT), ids, pose chain, keyframe rule,
                                                           reconstruction and local BA at keyframes. The others are round 2's
                                                           workloads on per-frame ground-truth track sets (kept as secondary legs)
  python bench.py --config {1,2,4}                         BASELINE configs[1] (default) / [2] mono 752x480 (the mono frame
                                                           incl. its new-point step closed on the device; --mode open: the
                                                           frame alone) / [4] 4K stereo
  python bench.py --strict-border {0,1,2,3,4,5}            border semantics of trackWithScale: 0 masked taps; 1..5 the reference's
                                                           never-reset tap state (identical results), differing in where the
                                                           replay of the border-touching features runs (default 4: chosen per
                                                           frame; vo_hip.h: vo_stereo_frame_set_strict_border)
  python bench.py --host-loop {library,python}             closed loops: the per-frame calls result(k) / enqueue(k + 1) /
                                                           prefetch(k + 2) by the library's sequence loop (vo_svo_run / vo_mvo_run,
                                                           default) or by this interpreter through ctypes
  N > 1: one rank per GPU, independent streams, one RCCL all_gather of the totals at the end; under
  torch.distributed.run the ranks are the launcher's, without a launcher environment bench.py starts the N ranks
  itself as fresh child processes (before this process has touched a GPU).

Prints ONE JSON line on rank 0. `value` is measured with the images resident in HBM before the timed region;
secondary measurements of the same run (N = 1 only) are reported next to it:
  host_images            the same frames with BOTH images arriving from pinned host memory every frame (H2D inside the
                         timed region, overlapped with the frame in flight)
  sequential_step10      step [10] as three host-driven operator calls after the frame's result (updateWeightBin,
                         extractORBwithBinning_fast, trackBidirection): what the class surface delivers without the
                         closed operator
  open_loop_candidates   round 1's workload: 150 candidates handed to the frame kernel before the frame (a sequential
                         VO cannot know them then; kept for comparison only)
See DESIGN.md §6 for every field.
"""
import argparse
import gc
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec
KITTI_K = (718.856, 718.856, 607.1928, 185.2157)
CONFIGS = {
    1: dict(kind="stereo", name="BASELINE configs[1]", metric="stereo VO frames/sec @1241x376, 1500 feats",
            W=1241, H=376, K=KITTI_K, n_u=60, n_v=25, win=21, max_level=6, speed=0.8, margin=31.0,
            thres=(80.0, 0.5, 3.0), thres_fast=15),
    2: dict(kind="mono", name="BASELINE configs[2]", metric="mono VO frames/sec @752x480, 1000 feats",
            W=752, H=480, K=(458.654, 457.296, 367.215, 248.375), n_u=40, n_v=25, win=15, max_level=5, speed=0.25,
            margin=31.0, thres=(20.0, 1.0, 5, 1.0), thres_fast=15),
    # configs[4]: the KITTI rig's field of view at three times the resolution. The camera advances a third of configs[1]'s
    # distance per frame (a 4K camera at three times the frame rate): the per-frame flow in PIXELS then equals configs[1]'s —
    # what 5 pyramid levels x a 21-pixel window can follow — and the track set fills the 100 x 80 buckets (at 0.8 m per frame
    # half of the tracks died every frame and the stream carried ~3000 of the 8000, every frame a keyframe: round 4).
    4: dict(kind="stereo", name="BASELINE configs[4]", metric="stereo VO frames/sec @3840x2160, 8000 feats",
            W=3840, H=2160, K=(718.856 * 3.0, 718.856 * 3.0, 1920.0, 1080.0), n_u=100, n_v=80, win=21, max_level=4,
            speed=0.8 / 3.0, tex_scale=1.0 / 3.0, margin=31.0, thres=(80.0, 0.5, 3.0), thres_fast=15,
            # feature_tracker.thres_sampson above 100: step [7] of the reference drops every feature below image row 660
            # (stereo_vo.cpp:659, a constant) — never at KITTI's 376 rows, 70 % of a 2160-row image: with the KITTI value (60) a 4K
            # stream cannot hold more than ~3900 of its 8000 buckets (measured). The YAML parameter is the reference's own switch.
            thres_sampson=120.0,
            # a keyframe every ~13 frames at this speed: nine of them (the local BA's window) need ~120 untimed frames
            prime=125),
}
UNTRIANGULATED = 0.10  # share of the track set whose landmark has no 3-D point yet (new since the last keyframe)
N_NEW_OPEN = 150       # candidates of the open-loop comparison workload


# ---- algorithmic bytes -------------------------------------------------------------------------
def klt_bytes_per_point_level(win):
    return (win + 2) ** 2 + (win + 1) ** 2  # u8 template tile incl. Scharr halo + one u8 search tile


IC_BYTES_8D = 25 ** 2 + 24 ** 2                  # §8(d): B_ic = N * [(25)^2 + (24)^2]
IC_RECORD_BYTES = (3 + 1) * 264 * 4 + 2 * 9 * 4  # design, strict border only: tap values + tap masks written per feature
POINT_IO_BYTES = 64                              # design: landmark, pixels, priors in; pixels, stage out


def frame_kernel_bytes(win, n, n_l0l1, n_step5, n_cand, levels, levels_bwd, strict):
    """Algorithmic bytes of ONE launch of the frame kernel, split as the verdict asked: `survey_8d` is §8(d)'s per-unit
    figure times the units the launch processes (KLT point-levels of the features' two calls and of the candidates'
    forward + backward calls, IC tiles), `design_records` is what this design adds (strict-border tap records, point I/O)."""
    klt = klt_bytes_per_point_level(win) * ((n + n_step5) * levels + n_cand * (levels + levels_bwd))
    return klt + IC_BYTES_8D * n_l0l1, (IC_RECORD_BYTES * n if strict else 0) + POINT_IO_BYTES * (n + n_cand)


def bins_with_keypoints(fe, slot):
    """Bins of the image in `slot` that hold a keypoint (bytes accounting of the speculative candidates), through the same two
    launches the loop uses (the candidate table: table 0, which the loop overwrites before it reads it)."""
    fe.enqueueCandidates(slot, 0)
    return int(fe.getCandidates(0)[1].sum())


def stamped_counters(config, workload="loop", kernel="frame_track"):
    """HBM traffic (PMC) and VALU instruction counts (SQ) of the dominant kernel from the committed rocprofv3 counter
    passes — quoted only while the kernel's sources still hash to what the passes ran on (a stale file is dropped)."""
    out = {"traffic": None, "valu_wave_instructions_per_launch": None, "counters_note": None}
    try:
        from visual_odometry_ros_amd import build as VB
        sha = VB.kernel_source_sha(VB.MONO_KERNEL_SOURCES if kernel == "mono_track" else VB.FRAME_KERNEL_SOURCES)
    except Exception:
        return out
    stale = []
    for key, suffix in (("pmc", "_frame_pmc.json"), ("sq", "_frame_sq_counters.json")):
        path = name = None
        for rnd in ("r05", "r04", "r03"):  # the newest round's pass that exists
            tag = (rnd if config == 1 else f"{rnd}_cfg{config}") + ("" if workload == "loop" else f"_{workload}")
            if os.path.exists(os.path.join(ROOT, "profiles", tag + suffix)):
                name = tag + suffix
                path = os.path.join(ROOT, "profiles", name)
                break
        if path is None:
            continue
        try:
            d = json.load(open(path))
        except Exception:
            continue
        if d.get("kernel_source_sha") != sha:
            stale.append(name)
            continue
        if key == "pmc":
            out["traffic"] = d.get("hbm_bytes_per_launch")
        else:
            for k, e in d.get("kernels", {}).items():
                if kernel in k and "SQ_INSTS_VALU" in e:
                    out["valu_wave_instructions_per_launch"] = e["SQ_INSTS_VALU"]
                    out["sq_launch_note"] = f"{e.get('SQ_WAVES')} wavefronts per launch in the counter pass"
    if stale:
        out["counters_note"] = "stale (kernel sources changed since the counter pass): " + ", ".join(stale)
    return out


def issue_roofline(counters, klt_ms, launches):
    """The roofline that binds: VALU wave-instructions per launch (counter pass) / live launch duration against the chip's
    issue rate — a SIMD issues one wave64 VALU instruction per 4 cycles: 256 CUs x 4 SIMDs x clk / 4 per second."""
    if not counters["valu_wave_instructions_per_launch"]:
        return None
    clk = 2.4e9
    peak_issue = 256 * 4 * clk / 4.0
    ach_issue = counters["valu_wave_instructions_per_launch"] / (klt_ms / launches * 1e-3)
    return {"bound": "valu_issue", "achieved": round(ach_issue / 1e9, 1), "peak": round(peak_issue / 1e9, 1),
            "unit": "G wave-instructions/s", "frac": round(ach_issue / peak_issue, 4),
            "valu_wave_instructions_per_launch": counters["valu_wave_instructions_per_launch"],
            "note": "VALU wave-instructions per launch (rocprofv3 SQ_INSTS_VALU, committed counter pass on the same kernel "
                    "sources) / live launch duration, against 1024 SIMDs x 2.4 GHz / 4; the launch is issue-bound while the "
                    "SIMDs are full and then waits for single wavefronts (DESIGN.md §4.2)"}


# ---- the one collective --------------------------------------------------------------------------------------------
FORCE_COLLECTIVE = False  # --force-collective: the process group and the gather also with ONE rank (the RCCL path on a 1-GPU box)


def gather_ranks(frames, seconds, seed, world, device=None):
    """The one collective of the job: an all_gather of {frames, seconds, stream seed} per rank
    (RCCL on GPUs, gloo in the CPU tests), 24 B per rank. Returns the per-rank list [(frames, seconds, seed)]."""
    if world <= 1 and not FORCE_COLLECTIVE:
        return [(float(frames), float(seconds), int(seed))]
    import torch
    import torch.distributed as dist
    mine = torch.tensor([float(frames), float(seconds), float(seed)], dtype=torch.float64, device=device)
    allv = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(allv, mine)
    return [(float(v[0]), float(v[1]), int(v[2])) for v in allv]


def aggregate(frames, seconds, world, device=None):
    """Whole-job totals: every rank ran `frames` frames of its own stream in `seconds`.
    Returns (total frames, max seconds over ranks)."""
    per = gather_ranks(frames, seconds, 0, world, device)
    return sum(p[0] for p in per), max(p[1] for p in per)


def stream_seed(rank):
    """Independent image stream per rank (stream s -> GPU s)."""
    return 2 + rank


def rank_cpus(local_rank, local_world):
    """The CPU set of one rank: a contiguous share of the CPUs this process may run on (contiguous ids share a NUMA node
    and its caches on the EPYC hosts), set BEFORE the rank renders its stream or spawns the renderer's workers — they
    inherit it — so that eight ranks do not render, launch and poll all over each other's cores."""
    cpus = sorted(os.sched_getaffinity(0))
    if local_world <= 1 or len(cpus) < local_world:
        return cpus
    per = len(cpus) // local_world
    return cpus[local_rank * per:(local_rank + 1) * per]


def pin_rank(local_rank, local_world):
    cpus = rank_cpus(local_rank, local_world)
    try:
        os.sched_setaffinity(0, cpus)
    except OSError:
        pass
    return sorted(os.sched_getaffinity(0))


def launch_ranks(n, argv, job_timeout=None):
    """`python bench.py --gpus N` without a launcher environment: this (parent) process starts N FRESH child
    interpreters, one per GPU, before it has imported torch or made any HIP call — it never touches a GPU and no
    GPU-initialised process is ever re-exec'ed. The children rendezvous on 127.0.0.1; rank 0 prints the one JSON
    line, which is relayed. A rank that dies, or a job that outlives `job_timeout` seconds (a rank stuck in front of the
    rendezvous), ends every child; the stderr tail of EVERY rank is then relayed with its rank in front. Returns the exit
    code (first non-zero child code; 124 for the time-out)."""
    import socket
    import subprocess
    import tempfile
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    procs, errs = [], []
    for r in range(n):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        ef = tempfile.TemporaryFile(mode="w+")
        errs.append(ef)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=e,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, stderr=ef, text=True))
    # wait for all ranks; if one fails, the others would sit in the rendezvous / barrier forever: end them
    # (rank 0 writes one line; a pipe of that size cannot fill up while we poll)
    rc, t0 = 0, time.time()
    while any(p.poll() is None for p in procs):
        bad = [p for p in procs if p.poll() not in (None, 0)]
        late = job_timeout is not None and time.time() - t0 > job_timeout
        if bad or late:
            rc = bad[0].returncode if bad else 124
            for p in procs:
                if p.poll() is None:
                    p.kill()  # exactly the children started above
            break
        time.sleep(0.05)
    out0 = procs[0].stdout.read()
    for p in procs:
        p.wait()
        rc = rc or p.returncode
    for ln in out0.splitlines():  # the JSON line goes to stdout; library chatter of the child (gloo's connect note) to stderr
        (sys.stdout if ln.startswith("{") else sys.stderr).write(ln + "\n")
    for r, ef in enumerate(errs):  # rank 0's stderr always (as before the ranks' stderr was a file); every rank's on failure
        ef.seek(0)
        txt = ef.read()
        ef.close()
        if txt and (rc != 0 or r == 0):
            for ln in txt.splitlines()[-40:]:
                sys.stderr.write((f"[rank {r}] " if rc != 0 else "") + ln + "\n")
    if rc == 124:
        sys.stderr.write(f"bench.py: the job did not finish within {job_timeout:.0f} s; all ranks ended\n")
    sys.stdout.flush()
    return rc


def rendezvous_check(rank, world, timeout_s, cpus):
    """--rendezvous-check: the N > 1 control flow of this file (launcher, rank environment, CPU sets, process group with
    its time-out, per-rank stream seeds, the gather, rank-0-only JSON) with gloo and no HIP call, so that it runs on a
    machine without GPUs (tests/test_bench_dist.py)."""
    import datetime
    import torch.distributed as dist
    if os.environ.get("BENCH_TEST_HANG_RANK") == str(rank):  # (test hook: a rank that never reaches the rendezvous)
        time.sleep(3600)
    if world > 1:
        dist.init_process_group(backend="gloo", timeout=datetime.timedelta(seconds=timeout_s))
    per = gather_ranks(100 + rank, 0.5 + 0.25 * rank, stream_seed(rank), world)
    all_cpus = [cpus]
    if world > 1:
        all_cpus = [None] * world
        dist.all_gather_object(all_cpus, cpus)
        dist.barrier()
    if rank == 0:
        tot, mx = sum(p[0] for p in per), max(p[1] for p in per)
        print(json.dumps({"rendezvous_check": True, "n_gpus": world, "value": tot / mx,
                          "per_rank": [{"frames": p[0], "seconds": p[1], "stream_seed": p[2], "cpus": c} for p, c in zip(per, all_cpus)]}),
              flush=True)
    if world > 1:
        dist.destroy_process_group()


def frame_ms_stats(stamps):
    """min / median / max of the per-frame wall time (ms) from the host time stamps taken at every result."""
    d = np.diff(np.asarray(stamps)) * 1e3
    if d.size == 0:
        return None
    return {"min": round(float(d.min()), 4), "median": round(float(np.median(d)), 4), "max": round(float(d.max()), 4),
            "p95": round(float(np.percentile(d, 95)), 4)}


def replay_split(stamps, replayed):
    """The synthetic stream is played back and forth: while the camera moves backwards no feature's refinement window
    leaves the image, so those frames have nothing to replay; a stream that only moves forward pays the `with` figure on
    every frame. stamps[k] is taken after frame k's result, replayed[k] is frame k's count."""
    d = np.diff(np.asarray(stamps)) * 1e3
    r = np.asarray(replayed[1:len(d) + 1])
    if d.size == 0 or r.size != d.size:
        return None
    w = r > 0
    out = {"frames_with_border_features": int(w.sum()), "frames_without": int((~w).sum())}
    if w.any():
        out["mean_ms_with"] = round(float(d[w].mean()), 4)
        out["mean_replayed_features"] = round(float(r[w].mean()), 1)
    if (~w).any():
        out["mean_ms_without"] = round(float(d[~w].mean()), 4)
    return out


def cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def host_cores():
    return len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)


def cpu_threads_for_baseline(O, L, R, cfg):
    """PyrLK over all host cores. A GPU box exposes every CPU of the host to the process (256 here) but
    schedules a share of them; 256 OpenMP threads on that share ran 28x slower than 16. So the thread count is measured:
    one trackBidirection of a bucket grid on the first pair with 16 / 32 / 64 / all threads, the fastest wins."""
    from visual_odometry_ros_amd import synthetic as S
    pts = S.bucket_points(cfg["W"], cfg["H"], cfg["n_u"], cfg["n_v"], np.random.default_rng(0), 31.0).astype(np.float32)
    cand = sorted({min(t, host_cores()) for t in (16, 32, 64, host_cores())})
    best, best_t = cand[0], None
    for t in cand:
        t0 = time.perf_counter()
        O.track_bidirection(L, R, pts, cfg["win"], cfg["max_level"], 80.0, 0.5, None, t)
        dt = time.perf_counter() - t0
        if best_t is None or dt < best_t:
            best, best_t = t, dt
    return best


# ======================================================================================================================
class StereoBench:
    """The stereo stream of one rank: everything resident in HBM (and, for the host-image leg, in pinned host memory)
    before any timed region; run() drives `count` frames of one mode."""

    def __init__(self, cfg, args, rank, local_rank, torch, V):
        from visual_odometry_ros_amd import synthetic as S
        from visual_odometry_ros_amd.api import StereoFramePipeline, make_stereo_params
        self.cfg, self.args, self.torch, self.V = cfg, args, torch, V
        self.W, self.H, self.win, self.max_level = cfg["W"], cfg["H"], cfg["win"], cfg["max_level"]
        self.n_pts = cfg["n_u"] * cfg["n_v"]
        self.dev = torch.device("cuda", local_rank)
        self.stream = S.StereoStream(width=self.W, height=self.H, K=cfg["K"], n_u=cfg["n_u"], n_v=cfg["n_v"],
                                     n_new=N_NEW_OPEN, seed=stream_seed(rank), speed=cfg["speed"], margin=cfg["margin"],
                                     tex_scale=cfg.get("tex_scale", 1.0))
        F = max(args.frames, 3)
        self.F = F
        self.poses = self.stream.poses(F)
        self.imgs = [self.stream.render_pair(p)[:2] for p in self.poses]
        # back-and-forth playback: 0,1,..,F-1,F-2,..,1,0,1,...
        self.order = list(range(F)) + list(range(F - 2, 0, -1))
        self.track_sets = {}
        rng = np.random.default_rng(1000 + rank)
        for s in range(len(self.order)):
            a, b = self.frame_id(s), self.frame_id(s + 1)
            if (a, b) not in self.track_sets:
                ts = self.stream.track_set(a * 131 + b, self.poses[a], self.poses[b])
                # bit 0 = lm->isTriangulated(): landmarks created since the last keyframe have no 3-D point yet
                ts["flags"] = (rng.random(self.n_pts) >= UNTRIANGULATED).astype(np.uint8)
                self.track_sets[(a, b)] = ts
        self.d_L = [torch.from_numpy(np.ascontiguousarray(L)).to(self.dev) for L, _ in self.imgs]
        self.d_R = [torch.from_numpy(np.ascontiguousarray(R)).to(self.dev) for _, R in self.imgs]
        self.h_L = self.h_R = None
        if args.host_images_leg:
            self.h_L = [torch.from_numpy(np.ascontiguousarray(L)).pin_memory() for L, _ in self.imgs]
            self.h_R = [torch.from_numpy(np.ascontiguousarray(R)).pin_memory() for _, R in self.imgs]
        self.d_ts = {key: {k: torch.from_numpy(np.ascontiguousarray(ts[k])).to(self.dev)
                           for k in ("pts_l0", "pts_r0", "Xp", "pts_new", "flags")}
                     for key, ts in self.track_sets.items()}
        torch.cuda.synchronize()
        self.ctx = V.Context(device=local_rank, max_width=self.W, max_height=self.H,
                             max_points=max(self.n_pts, N_NEW_OPEN) + 64, n_slots=5, max_level=self.max_level)
        thr = cfg["thres"]
        self.prm = make_stereo_params(self.W, self.H, self.win, self.max_level, thr[0], thr[1], thr[2], self.stream.K,
                                      self.stream.K, self.stream.T_lr)
        self.pipe = StereoFramePipeline(self.ctx, self.prm, strict_border=args.strict_border)
        self.ctx.set_pyramid_window_hint(self.win)  # build only the levels PyrLK with this window uses
        self.eff_levels = self.ctx.pyramid_levels(self.W, self.H, self.win, self.max_level) + 1
        self.eff_levels_bwd = self.ctx.pyramid_levels(self.W, self.H, self.win, self.max_level - 1) + 1
        self.fe = V.FeatureExtractor(self.ctx)
        self.fe.initParams(self.W, self.H, cfg["n_u"], cfg["n_v"], THRES_FAST=cfg["thres_fast"])
        self.bins = self.fe.binParams()
        self.ft = V.FeatureTracker(self.ctx)
        self.slot = {"P": 0, "CL": 1, "CR": 2, "NL": 3, "NR": 4}
        self.n_bins_with_kp = {}

    def frame_id(self, step):
        return self.order[step % len(self.order)]

    # ---- per-frame host code of the three modes --------------------------------------------------------------------
    def _ingest(self, fid, host):
        s = self.slot
        if host:
            self.ctx.set_stereo_pair_host_async(s["NL"], self.h_L[fid].data_ptr(), s["NR"], self.h_R[fid].data_ptr(),
                                                self.W, self.H, self.W)
        else:
            self.ctx.set_stereo_pair_device(s["NL"], self.d_L[fid].data_ptr(), s["NR"], self.d_R[fid].data_ptr(),
                                            self.W, self.H, self.W)

    def _enqueue(self, step, mode, host):
        a, b = self.frame_id(step), self.frame_id(step + 1)
        t, s = self.d_ts[(a, b)], self.slot
        dT = self.track_sets[(a, b)]["dT_prior"]
        slots = (s["P"], s["CL"], s["CR"])
        if mode == "closed":
            self.pipe.enqueue_closed_device(t["pts_l0"].data_ptr(), t["pts_r0"].data_ptr(), t["Xp"].data_ptr(), self.n_pts,
                                            dT, self.bins, step & 1, slots=slots, d_lm_flags=t["flags"].data_ptr())
        elif mode == "open":
            self.pipe.enqueue_device(t["pts_l0"].data_ptr(), t["pts_r0"].data_ptr(), t["Xp"].data_ptr(), self.n_pts, dT,
                                     t["pts_new"].data_ptr(), N_NEW_OPEN, slots=slots, d_lm_flags=t["flags"].data_ptr())
        else:  # sequential: the frame without candidates; step [10] follows its result
            self.pipe.enqueue_device(t["pts_l0"].data_ptr(), t["pts_r0"].data_ptr(), t["Xp"].data_ptr(), self.n_pts, dT,
                                     0, 0, slots=slots, d_lm_flags=t["flags"].data_ptr())
        # the NEXT pair does not depend on this frame: its ingestion (and, closed, its keypoint detection) is enqueued
        # behind the frame's launches and runs on the side stream while the frame is in flight
        self._ingest(self.frame_id(step + 2), host)
        if mode == "closed":
            self.fe.enqueueCandidates(s["NL"], (step + 1) & 1)

    def _step10_on_the_host(self, r):
        """stereo_vo.cpp:691-711 as three operator calls: updateWeightBin(lmtrack_final.pts_l1),
        extractORBwithBinning_fast(I1_left), trackBidirection(I1_left, I1_right, pts_l1_new)."""
        s = self.slot
        self.fe.updateWeightBin(r["pts_l1"][r["stage"] == 4])
        pts_new = self.fe.extractORBwithBinning_fast(s["CL"])
        thr = self.cfg["thres"]
        return self.ft.trackBidirection(s["CL"], s["CR"], pts_new, self.win, self.max_level, thr[0], thr[1])

    def run(self, first, count, mode, host=False, keep=None, stamps=None, on_result=None):
        """`count` frames, one at a time: the result of frame s is received before frame s+1 is enqueued."""
        s = self.slot
        if count <= 0:
            return
        self._enqueue(first, mode, host)
        for step in range(first, first + count):
            r = self.pipe.result(copy=keep is not None and len(keep) < self.args.cpu_frames)
            if mode == "sequential":
                r = dict(r, new=self._step10_on_the_host(r))
            s["P"], s["CL"], s["CR"], s["NL"], s["NR"] = s["CL"], s["NL"], s["NR"], s["P"], s["CR"]
            if step + 1 < first + count:
                self._enqueue(step + 1, mode, host)
            if stamps is not None:
                stamps.append(time.perf_counter())
            if keep is not None and len(keep) < self.args.cpu_frames:
                keep.append((step, r))
            if on_result is not None:
                on_result(step, r)

    # ---- the reference's own class surface: one operator call per step, host arrays in and out -----------------------
    def class_surface_frame(self, step, slots, me, stamp):
        """One frame as stereo_vo.cpp:483-711 calls the three classes: priors on the host, trackWithPrior, trackWithScale,
        trackWithPrior, poseOnlyBundleAdjustment_Stereo, the y-gate, updateWeightBin, extractORBwithBinning_fast,
        trackBidirection — numpy arrays in and out of every call, compactions on the host, images through the slot cache
        (one upload + pyramid per image and frame; stamp None = per call, as the reference)."""
        a, b = self.frame_id(step), self.frame_id(step + 1)
        ts = self.track_sets[(a, b)]
        I0l, (I1l, I1r) = self.imgs[a][0], self.imgs[b]
        thr, W, H = self.cfg["thres"], self.W, self.H
        f = np.float32
        K = np.asarray(self.stream.K, f)
        n = ts["pts_l0"].shape[0]
        tri = (ts["flags"] & 1) != 0
        # [3] priors and patch scale (:483-522)
        def inv_se3(T):  # geometry::inverseSE3_f in float32, the operation order of the library's own
            T = np.asarray(T, f)
            Rt = T[:3, :3].T.copy()
            o = np.eye(4, dtype=f)
            o[:3, :3] = Rt
            for i in range(3):
                o[i, 3] = f(f(f(-Rt[i, 0]) * T[0, 3]) + f(f(-Rt[i, 1]) * T[1, 3])) + f(f(-Rt[i, 2]) * T[2, 3])
            return o

        def xform(T, X):  # R X + t in float32, one rounding per operation, left to right
            return np.stack([((T[r, 0] * X[:, 0] + T[r, 1] * X[:, 1]) + T[r, 2] * X[:, 2]) + T[r, 3] for r in range(3)], 1).astype(f)
        T_cp, T_rl = inv_se3(ts["dT_prior"]), inv_se3(self.stream.T_lr)
        Xp = ts["Xp"].astype(f)
        Xl1 = xform(T_cp, Xp)
        Xr1 = xform(T_rl, Xl1)
        with np.errstate(divide="ignore", invalid="ignore"):
            izl, izr = f(1.0) / Xl1[:, 2], f(1.0) / Xr1[:, 2]
            pl = np.stack([K[0] * Xl1[:, 0] * izl + K[2], K[1] * Xl1[:, 1] * izl + K[3]], 1).astype(f)
            pr = np.stack([K[0] * Xr1[:, 0] * izr + K[2], K[1] * Xr1[:, 1] * izr + K[3]], 1).astype(f)
            scale = np.where(tri, Xp[:, 2] / Xl1[:, 2], f(1.0)).astype(f)

        def inimg(p):
            return (p[:, 0] >= 3) & (p[:, 1] >= 3) & (p[:, 0] < W - 3) & (p[:, 1] < H - 3)
        ok = tri & inimg(pl) & inimg(pr) & (Xl1[:, 2] >= 0.1) & (Xr1[:, 2] >= 0.1)
        pl1 = np.where(ok[:, None], pl, ts["pts_l0"])
        pr1 = np.where(ok[:, None], pr, ts["pts_r0"])
        # [4] l0 -> l1
        s0 = slots.get(I0l, stamp)
        s1 = slots.get(I1l, stamp, avoid=(s0,))
        p, m = self.ft.trackWithPrior(s0, s1, ts["pts_l0"], self.win, self.max_level, thr[0], pl1)
        m = m & ((ts["flags"] & 2) == 0)
        i1 = np.nonzero(m)[0]
        # [4-1] trackWithScale (the derivative images are computed on the device from I0)
        s0 = slots.get(I0l, stamp, avoid=(s1,))
        s1 = slots.get(I1l, stamp, avoid=(s0,))
        p2, m2 = self.ft.trackWithScale(s0, s1, ts["pts_l0"][i1], scale[i1], p[i1], None, strict_border=bool(self.args.strict_border))
        i2 = i1[m2]
        # [5] l1 -> r1
        s1 = slots.get(I1l, stamp)
        s2 = slots.get(I1r, stamp, avoid=(s1,))
        q, m3 = self.ft.trackWithPrior(s1, s2, p2[m2], self.win, self.max_level, thr[0], pr1[i2])
        i3 = i2[m3]
        pl3, pr3 = p2[m2][m3], q[m3]
        # [6] pose-only BA on the triangulated survivors (:595-646)
        tb = tri[i3]
        okb, T, mask, info = me.poseOnlyBundleAdjustment_Stereo(ts["Xp"][i3][tb], pl3[tb], pr3[tb], K, K, self.stream.T_lr, thr[2],
                                                                ts["dT_prior"])
        motion = np.ones(i3.shape[0], bool)
        motion[np.nonzero(tb)[0]] = mask
        fin = motion & ~(pl3[:, 1] > 660)  # [7]
        stage = np.zeros(n, np.uint8)
        stage[i1] = 1
        stage[i2] = 2
        stage[i3] = 3
        stage[i3[fin]] = 4
        # [10] updateWeightBin, extractORBwithBinning_fast, trackBidirection (:691-711)
        self.fe.updateWeightBin(pl3[fin])
        s1 = slots.get(I1l, stamp)
        pts_new = self.fe.extractORBwithBinning_fast(s1)
        s1 = slots.get(I1l, stamp)
        s2 = slots.get(I1r, stamp, avoid=(s1,))
        pnr, mn = self.ft.trackBidirection(s1, s2, pts_new, self.win, self.max_level, thr[0], thr[1])
        return dict(stage=stage, dT=T, pts_new=pts_new, pts_new_r=pnr, mask_new=mn)

    def run_class_surface(self, first, count, cached, stamps=None):
        from visual_odometry_ros_amd.api import ImageSlots
        self.ctx.set_ingest_side_stream(False)
        slots = ImageSlots(self.ctx, (0, 1, 2, 3))
        me = self.V.MotionEstimator(self.ctx, True, self.stream.T_lr)
        out = None
        for step in range(first, first + count):
            out = self.class_surface_frame(step, slots, me, (step + 1) if cached else None)
            if stamps is not None:
                stamps.append(time.perf_counter())
        self.last_uploads = slots.uploads
        return out

    def prime(self, mode, host=False):
        """Slots P / CL / CR for step 0 and, closed, the candidate table of CL; then a few untimed frames (lazy
        allocations, code-object loads)."""
        c, s = self.ctx, self.slot
        c.synchronize()
        c.set_ingest_side_stream(mode != "open")  # (the open-loop comparison keeps round 1's single-stream ingestion)
        s.update({"P": 0, "CL": 1, "CR": 2, "NL": 3, "NR": 4})
        c.set_image_device(s["P"], self.d_L[self.frame_id(0)].data_ptr(), self.W, self.H, self.W)
        f1 = self.frame_id(1)
        c.set_stereo_pair_device(s["CL"], self.d_L[f1].data_ptr(), s["CR"], self.d_R[f1].data_ptr(), self.W, self.H, self.W)
        if mode == "closed":
            self.fe.enqueueCandidates(s["CL"], 0)
        c.synchronize()
        PRIME = 4  # even: the table parity of step 0 stays 0
        self.run(0, PRIME, mode, host)
        return PRIME



# ======================================================================================================================
# loop mode: a forward-driving stream, every frame rendered once
def _render_one(job):
    from visual_odometry_ros_amd import synthetic as S
    cfg, seed, k, n = job
    st = S.StereoStream(width=cfg["W"], height=cfg["H"], K=cfg["K"], n_u=cfg["n_u"], n_v=cfg["n_v"], seed=seed, speed=cfg["speed"],
                        z_end=scene_length(cfg, n), tex_scale=cfg.get("tex_scale", 1.0))
    L, R, _ = st.render_pair(st.poses(n)[k])
    return k, L, R


def scene_length(cfg, n_frames):
    """The synthetic corridor ends in a wall at 600 m (the scene every round rendered): a forward-driving stream longer
    than ~700 frames at 0.8 m per frame would drive into it — the tracks die, the loop ends like the reference's would
    ('large update!'). Longer streams get a longer corridor (the images of the first 600 m stay what they were, apart from
    what is visible of the far wall)."""
    need = n_frames * float(cfg["speed"]) + 150.0
    return 600.0 if need <= 600.0 else float(int(need / 100.0 + 1) * 100)


def render_stream(cfg, seed, n, workers):
    """The n stereo pairs of a stream (numpy u8), rendered by a process pool. MUST run before this process touches a
    GPU (the pool forks). A cache under /tmp keeps repeated runs of the same box (profiling passes) from re-rendering."""
    key = f"{cfg['W']}x{cfg['H']}_{seed}_{cfg['speed']:.4f}_{n}" + ("" if scene_length(cfg, n) == 600.0 else f"_z{int(scene_length(cfg, n))}") \
        + ("" if cfg.get("tex_scale", 1.0) == 1.0 else f"_t{cfg['tex_scale']:.3f}")
    cache = os.path.join(os.environ.get("VO_BENCH_CACHE", "/tmp"), f"vo_bench_stream_{key}.npz")
    if os.path.exists(cache):
        try:
            z = np.load(cache)
            zl, zr = z["L"], z["R"]  # (once: every access of an NpzFile member reads and checks the whole array again)
            return [(zl[k], zr[k]) for k in range(n)]
        except Exception:
            pass
    import multiprocessing as mp
    workers = max(1, min(workers or host_cores(), n, 32))
    jobs = [(cfg, seed, k, n) for k in range(n)]
    if workers == 1:
        out = [_render_one(j) for j in jobs]
    else:
        with mp.get_context("fork").Pool(workers) as pool:
            out = pool.map(_render_one, jobs, chunksize=1)
    out.sort(key=lambda t: t[0])
    try:
        np.savez(cache, L=np.stack([o[1] for o in out]), R=np.stack([o[2] for o in out]))
    except Exception:
        pass
    return [(o[1], o[2]) for o in out]


# Untimed frames in front of the warm-up. The keyframe window (n_max_keyframes_in_window = 9, kitti_00_stereo.yaml:83;
# keyframes.cpp:217-303) must be FULL and sliding before the timed region, or a short run (the driver's --steps 20) times the
# local-BA solves of windows 3, 4, 5, 6 — smaller problems than the stream ever sees again (round 4: 2744 driver-timed
# against 2490 steady). A keyframe every 4-5 frames on these streams: nine of them are there after ~45 frames; the count
# is reported (loop.keyframes_before_timed_region). LOOP_PRIME_GROWING is round 4's value, kept for secondary.growing_window.
# 65, not 50: twenty consecutive frames are a small sample of this stream — ordinary frames cost 0.303 ms over frames 55-75, 0.274 over
# 75-95, 0.283 / 0.272 / 0.278 over the next three twenties (frame_ms_by_kind.mean_ms_other_by_20_frames of a longer run shows it) —
# and with the driver's --steps 20 --warmup 5 the timed frames are now 70-90, whose cost (0.280 ms) is the 400-frame mean (0.278 ms).
LOOP_PRIME = 65
LOOP_PRIME_GROWING = 3


def loop_frames_needed(args):
    return LOOP_PRIME + args.warmup + args.steps + 2  # (+2: the frame in flight at the end of the timed region, and the pair prefetched under it)


def run_loop(cfg, args, rank, local_rank, world, torch, V, barrier, dev, imgs, secondary):
    """BASELINE configs[1] as a CLOSED LOOP: StereoVO::trackStereoImages on a forward-driving stream, the track set carried
    on the device from frame to frame (vo_svo_*: survivors + new landmarks by DLT + ids + keyframes [+ local BA])."""
    from visual_odometry_ros_amd import synthetic as S
    W, H, win, lvl = cfg["W"], cfg["H"], cfg["win"], cfg["max_level"]
    n_bins = cfg["n_u"] * cfg["n_v"]
    st = S.StereoStream(width=W, height=H, K=cfg["K"], n_u=cfg["n_u"], n_v=cfg["n_v"], seed=stream_seed(rank), speed=cfg["speed"])
    F = len(imgs)
    poses_gt = st.poses(F)
    d_L = [torch.from_numpy(np.ascontiguousarray(L)).to(dev) for L, _ in imgs]
    d_R = [torch.from_numpy(np.ascontiguousarray(R)).to(dev) for _, R in imgs]
    torch.cuda.synchronize()
    cap = 2 * n_bins + 1024  # several features may share a bucket: the set grows beyond one per bucket
    ctx = V.Context(device=local_rank, max_width=W, max_height=H, max_points=cap, n_slots=5, max_level=lvl)
    if host_cores() < 3:  # (the rank's own share: caller thread + the runtime's helper threads would otherwise fight a spinning poll)
        ctx.debug_set(ctx.OPT_POLL_YIELD, 1)
    thr = cfg["thres"]
    svo = V.StereoVO(ctx, W, H, cfg["K"], cfg["K"], st.T_lr, cfg["n_u"], cfg["n_v"], thres_fastscore=cfg["thres_fast"],
                     window_size=win, max_level=lvl, thres_error=thr[0], thres_bidirection=thr[1], thres_poseba_error=thr[2],
                     strict_border=args.strict_border, local_ba=bool(args.lba), thres_sampson=cfg.get("thres_sampson", 60.0))
    eff_levels = ctx.pyramid_levels(W, H, win, lvl) + 1
    eff_levels_bwd = ctx.pyramid_levels(W, H, win, lvl - 1) + 1
    # bins that hold a keypoint, per left image: the candidates the frame kernel tracks speculatively (bytes accounting)
    fe = V.FeatureExtractor(ctx)
    fe.initParams(W, H, cfg["n_u"], cfg["n_v"], THRES_FAST=cfg["thres_fast"])
    n_kp_bins = []
    for k in range(F):
        ctx.set_image_device(4, d_L[k].data_ptr(), W, H, W)
        n_kp_bins.append(bins_with_keypoints(fe, 4))
    ctx.synchronize()
    ptr = [((d_L[k].data_ptr(), W), (d_R[k].data_ptr(), W)) for k in range(F)]
    prefetch = not args.no_prefetch
    state = {"k": 0}
    traj, infos, stamps = [], [], []

    def issue(k):
        svo.enqueue(*ptr[k])
        if prefetch and k + 1 < F:
            svo.prefetch(*ptr[k + 1])  # the NEXT pair does not depend on this frame: ingestion + detection under it

    def step(keep):
        # frame k is in flight (issued at the end of the previous step): collect it, send frame k + 1 off at once, and
        # only then look at frame k's results — the caller's own work per frame runs under the next frame
        k = state["k"]
        i = svo.result()
        if keep:
            stamps.append(time.perf_counter())
        if k + 1 < F and not args.no_issue_ahead:
            issue(k + 1)
        state["k"] = k + 1
        if not keep:
            kf_before[0] += int(bool(i.is_keyframe))
        if keep:
            c = i.counts
            infos.append((k, i.n_tracks_in, c.n_l0l1, c.n_refine, c.n_replayed, i.n_new_candidates, i.n_new, i.n_final,
                          i.is_keyframe, i.lba_ran, c.gn_iterations, c.n_ba))
        traj.append(np.array(i.T_wc, np.float32).reshape(4, 4))
        if k + 1 < F and args.no_issue_ahead:
            issue(k + 1)

    # Who drives the loop: the library (vo_svo_run — result(k), enqueue(k + 1), prefetch(k + 2) in compiled code, as the
    # reference's caller is) or this interpreter (the same three calls per frame through ctypes: ~5 us more per frame)
    lib_loop = args.host_loop == "library" and prefetch and not args.no_issue_ahead
    K = args.steps

    kf_before = [0]

    def collect(out, keep, k0):
        for j, i in enumerate(out):
            if not keep:
                kf_before[0] += int(bool(i.is_keyframe))
            if keep:
                c = i.counts
                infos.append((k0 + j, i.n_tracks_in, c.n_l0l1, c.n_refine, c.n_replayed, i.n_new_candidates, i.n_new, i.n_final,
                              i.is_keyframe, i.lba_ran, c.gn_iterations, c.n_ba))
            traj.append(np.array(i.T_wc, np.float32).reshape(4, 4))

    if lib_loop:
        n0 = LOOP_PRIME + args.warmup
        if n0 > 0:
            collect(svo.runSequence(ptr, 0, n0)[0], False, 0)
        else:
            issue(0)
        state["k"] = n0
    else:
        issue(0)
        for _ in range(LOOP_PRIME + args.warmup):
            step(False)
    ctx.profile_enable(K * 4 + 64)
    ctx.profile_set_classes(1 << 1)
    ctx.profile_reset()

    def timed_run():
        if lib_loop:
            k0 = state["k"]
            out, st_k = svo.runSequence(ptr, k0, k0 + K)
            stamps.extend(st_k.tolist())
            collect(out, True, k0)
            state["k"] = k0 + K
        else:
            for _ in range(K):
                step(True)

    dt = timed(timed_run, barrier, ctx)
    if state["k"] < F:
        svo.result()  # (the frame issued by the last timed step: it ran inside the timed region, nobody reads it)
    per_rank = gather_ranks(K, dt, stream_seed(rank), world, dev)
    tot_frames, max_dt = sum(p[0] for p in per_rank), max(p[1] for p in per_rank)
    if rank != 0:
        return None, ctx
    klt_n, klt_ms = ctx.profile_get(1)
    launches = max(klt_n, 1)
    I = np.array(infos, np.int64)
    strict = bool(args.strict_border)
    b_req = b_spec = b_des = 0
    for (k, n_in, n_l0l1, n_refine, n_rep, n_cand_emit, n_new, n_final, kf, lba, it, nba) in infos:
        per = klt_bytes_per_point_level(win)
        feat = per * (n_in + (n_refine - n_rep)) * eff_levels + IC_BYTES_8D * n_l0l1
        cand_all = per * n_kp_bins[k] * (eff_levels + eff_levels_bwd)   # every bin's candidate is tracked (speculation)
        cand_req = per * n_cand_emit * (eff_levels + eff_levels_bwd)     # the candidates the reference would track
        b_req += feat + cand_req
        b_spec += cand_all - cand_req
        b_des += (IC_RECORD_BYTES * n_in if strict else 0) + POINT_IO_BYTES * (n_in + n_kp_bins[k])
    achieved = ((b_req + b_spec) / launches) / (klt_ms / launches * 1e-3) / 1e9 if klt_n else 0.0
    achieved_req = (b_req / launches) / (klt_ms / launches * 1e-3) / 1e9 if klt_n else 0.0
    counters = stamped_counters(args.config)
    issue = issue_roofline(counters, klt_ms, launches) if klt_n else None
    # trajectory against the renderer's ground truth (frame 0 = identity)
    T0i = np.linalg.inv(poses_gt[0])
    gt = np.stack([(T0i @ p)[:3, 3] for p in poses_gt[:len(traj)]])
    est = np.stack([T[:3, 3] for T in traj]).astype(np.float64)
    path = float(np.sum(np.linalg.norm(np.diff(gt, axis=0), axis=1)))
    replayed = I[:, 4]
    d_ms = np.diff(np.asarray(stamps)) * 1e3
    kfm = I[1:, 8] > 0
    out = {
        "metric": cfg["metric"],
        "value": round(tot_frames / max_dt, 2),
        "unit": "frames/s",
        "n_gpus": world,
        "steps": K,
        "warmup": args.warmup,
        "ms_per_step": round(1e3 * max_dt / K, 4),
        "frame_ms": frame_ms_stats(stamps),
        "frame_ms_by_kind": {
            "keyframes": int(kfm.sum()), "mean_ms_keyframe": round(float(d_ms[kfm].mean()), 4) if kfm.any() else None,
            "mean_ms_other": round(float(d_ms[~kfm].mean()), 4) if (~kfm).any() else None,
            "keyframe_ms_first_12": [round(float(v), 4) for v in d_ms[kfm][:12]],
            # ordinary frames by position in the timed region (20 frames each, the first five windows): a short run next to a long one
            "mean_ms_other_by_20_frames": [round(float(d_ms[q:q + 20][~kfm[q:q + 20]].mean()), 4)
                                           for q in range(0, min(len(d_ms), 100), 20) if (~kfm[q:q + 20]).any()],
            "frames_with_border_features": int((replayed[1:] > 0).sum()), "mean_replayed_features": round(float(replayed.mean()), 1)},
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u8/i32 (KLT, FAST) + f32 (IC, GN, DLT) + f64 (local BA)",
        "data": "synthetic",
        "config": {
            "workload": f"{cfg['name']}: StereoVO::trackStereoImages on a synthetic forward-driving stereo stream {W}x{H} "
                        f"({cfg['speed']} m/frame, every frame rendered once), {cfg['n_u']}x{cfg['n_v']} buckets, win {win}, max_level "
                        f"{lvl} ({eff_levels} effective levels), thresholds of config/stereo/kitti_00_stereo.yaml; whole operator "
                        "sequence steps [1]-[12] incl. keypoint detection, bucketing, new landmarks (DLT), keyframe rule, "
                        "reconstruction" + (" and local BA" if args.lba else "") + "; one independent stream per GPU; result "
                        "read back every frame"
                        + (f"; texture cells x {cfg['tex_scale']:.3f} (the same detail per pixel as configs[1])" if cfg.get("tex_scale", 1.0) != 1.0 else "")
                        + (f"; feature_tracker.thres_sampson {cfg['thres_sampson']:g} (> 100: the reference's row-660 gate of step [7] never "
                           "fires, as it never does on a 376-row image)" if "thres_sampson" in cfg else ""),
            "track_set": "closed loop",
            "playback": "forward",
            "local_ba": bool(args.lba),
            "prefetch_next_pair": prefetch,
            "host_loop": "library (vo_svo_run: result(k), enqueue(k + 1), prefetch(k + 2) per frame in compiled code)" if lib_loop
                         else "python (the same three calls per frame through ctypes)",
            "images": "resident in HBM",
            "strict_border": int(args.strict_border),
            "frames_rendered": F,
        },
        "loop": {
            "mean_tracks_in": round(float(I[:, 1].mean()), 1), "mean_final": round(float(I[:, 7].mean()), 1),
            "mean_new_landmarks": round(float(I[:, 6].mean()), 1), "mean_ba_set": round(float(I[:, 11].mean()), 1),
            "mean_gn_iterations": round(float(I[:, 10].mean()), 2), "keyframes": int(I[:, 8].sum()), "lba_runs": int(I[:, 9].sum()),
            "path_m": round(path, 2), "end_point_error_m": round(float(np.linalg.norm(est[-1] - gt[-1])), 4),
            "ate_rmse_m": round(float(np.sqrt(np.mean(np.sum((est - gt) ** 2, axis=1)))), 4),
            "keyframes_before_timed_region": kf_before[0], "kf_window": int(svo.prm.kf_window),
            "untimed_frames": LOOP_PRIME + args.warmup,
        },
        "roofline": {
            "bound": "hbm",
            "kernel": f"frame_track_kernel<{win}>",
            "achieved": round(achieved, 2),
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 5),
            "traffic": counters["traffic"],
            "achieved_required_only": round(achieved_req, 2),
            "frac_required_only": round(achieved_req / HBM_PEAK_GBS, 5),
            "alg_bytes_per_launch": {"survey_8d_required": round(b_req / launches), "survey_8d_speculative": round(b_spec / launches),
                                     "design_records": round(b_des / launches)},
            "avg_launch_us": round(1e3 * klt_ms / launches, 2),
            "note": "achieved = SURVEY 8(d) bytes (features + every bucket's candidate) / launch duration (HIP events, live); "
                    "required_only leaves out the candidates the reference would not have tracked; the path is issue/latency-"
                    "bound, the fraction is reported because the metric asks for it (DESIGN.md §6)",
        },
        "roofline_issue": issue,
        "per_rank_fps": [round(p[0] / p[1], 2) for p in per_rank],
    }
    if counters["counters_note"]:
        out["roofline"]["counters_note"] = counters["counters_note"]
    if ctx.debug_switches:  # (VO_TEST_SWITCHES=1 only: a measurement run with a test switch on says so)
        out["config"]["debug_switches"] = dict(ctx.debug_switches)
    svo.close()
    prm_svo = svo.prm
    ctx.close()  # (its three HIP streams go away: the legs below must not share hardware queues with an idle context)
    if secondary:
        out["secondary"] = secondary_legs(cfg, args, rank, local_rank, torch, V, barrier)
        out["secondary"]["growing_window"] = growing_window_leg(cfg, args, local_rank, V, barrier, st, ptr, cap)
        out["secondary"]["loop_host_images"] = host_image_loop_leg(cfg, args, local_rank, V, barrier, st, imgs, cap, traj)
        out["secondary"]["track_stereo_images_synchronous"] = synchronous_leg(cfg, args, local_rank, torch, V, barrier, dev, st, imgs, cap, traj)
        if args.batch_S:
            out["secondary"]["streams_per_gpu"] = batch_leg(cfg, args, local_rank, torch, V, dev, prm_svo, cap)
    if world == 1 and not args.no_cpu_baseline:
        out.update(cpu_baseline_loop(cfg, args, st, imgs, traj, V, local_rank))
    return out, ctx


def growing_window_leg(cfg, args, local_rank, V, barrier, st, ptr, cap):
    """Round 4's driver-timed workload, kept for comparison: the stream's FIRST frames — three untimed + warm-up, then 20
    timed ones that hold the local-BA solves of the still growing keyframe window (3, 4, 5, 6 keyframes)."""
    W, H, thr = cfg["W"], cfg["H"], cfg["thres"]
    ctx = V.Context(device=local_rank, max_width=W, max_height=H, max_points=cap, n_slots=5, max_level=cfg["max_level"])
    svo = V.StereoVO(ctx, W, H, cfg["K"], cfg["K"], st.T_lr, cfg["n_u"], cfg["n_v"], thres_fastscore=cfg["thres_fast"],
                     window_size=cfg["win"], max_level=cfg["max_level"], thres_error=thr[0], thres_bidirection=thr[1],
                     thres_poseba_error=thr[2], strict_border=args.strict_border, local_ba=bool(args.lba), thres_sampson=cfg.get("thres_sampson", 60.0))
    pre = LOOP_PRIME_GROWING + min(args.warmup, 5)
    K = min(20, len(ptr) - pre - 2)
    svo.runSequence(ptr, 0, pre)
    res = {}

    def run():
        res["out"], res["st"] = svo.runSequence(ptr, pre, pre + K)

    dt = timed(run, barrier, ctx)
    svo.result()
    svo.close()
    ctx.close()
    d_ms = np.diff(np.asarray(res["st"])) * 1e3
    kf = np.asarray([bool(i.is_keyframe) for i in res["out"]][1:], bool)
    return {"value": round(K / dt, 2), "unit": "frames/s", "frames": K, "untimed_frames": pre,
            "mean_ms_keyframe": round(float(d_ms[kf].mean()), 4) if kf.any() else None,
            "mean_ms_other": round(float(d_ms[~kf].mean()), 4) if (~kf).any() else None,
            "note": "frames %d..%d of the stream: the keyframe window is still filling (round 4's BENCH workload)" % (pre, pre + K - 1)}


def host_image_loop_leg(cfg, args, local_rank, V, barrier, st, imgs, cap, traj):
    """The headline loop as the reference's boundary hands the images over: cv::Mat-like HOST arrays into trackStereoImages
    (here: enqueue / prefetch / result with numpy arrays), every pair crossing PCIe inside the timed region."""
    W, H, thr = cfg["W"], cfg["H"], cfg["thres"]
    ctx = V.Context(device=local_rank, max_width=W, max_height=H, max_points=cap, n_slots=5, max_level=cfg["max_level"])
    svo = V.StereoVO(ctx, W, H, cfg["K"], cfg["K"], st.T_lr, cfg["n_u"], cfg["n_v"], thres_fastscore=cfg["thres_fast"],
                     window_size=cfg["win"], max_level=cfg["max_level"], thres_error=thr[0], thres_bidirection=thr[1],
                     thres_poseba_error=thr[2], strict_border=args.strict_border, local_ba=bool(args.lba), thres_sampson=cfg.get("thres_sampson", 60.0))
    host = [(np.ascontiguousarray(L), np.ascontiguousarray(R)) for L, R in imgs]
    F, pre = len(host), LOOP_PRIME + args.warmup
    K = min(args.steps, F - pre - 2)
    same = [True]

    def issue(k):
        svo.enqueue(*host[k])
        if k + 1 < F:
            svo.prefetch(*host[k + 1])

    def run(k0, n, check):
        for k in range(k0, k0 + n):
            i = svo.result()
            if k + 1 < F:
                issue(k + 1)
            if check and k < len(traj):
                same[0] = same[0] and bool(np.array_equal(np.array(i.T_wc, np.float32).view(np.uint32), traj[k].reshape(-1).view(np.uint32)))

    issue(0)
    run(0, pre, False)
    dt = timed(lambda: run(pre, K, True), barrier, ctx)
    svo.result()
    svo.close()
    ctx.close()
    return {"value": round(K / dt, 2), "unit": "frames/s", "frames": K, "poses_equal_resident_run": same[0],
            "note": "pageable host arrays in, 2 x %d KB per frame over PCIe on the ingest stream under the frame in flight" % (W * H // 1024)}


def synchronous_leg(cfg, args, local_rank, torch, V, barrier, dev, st, imgs, cap, traj):
    """What the reference's own caller gets (ros2/visual_odometry/stereo_vo_ros2.cpp:104): trackStereoImages(left, right, t)
    — ONE synchronous call per pair, the pair handed over when its frame starts, nothing known about the next pair: no
    prefetch, no frame issued ahead; local BA on. Measured with host images (cv::Mat-like numpy arrays: upload, pyramids and
    detection inside the call) and with device-resident images, and — for comparison — the two-call form without
    prefetch (enqueue(k + 1) right behind result(k), --no-prefetch of the headline loop)."""
    W, H, thr = cfg["W"], cfg["H"], cfg["thres"]
    host = [(np.ascontiguousarray(L), np.ascontiguousarray(R)) for L, R in imgs]
    F, pre = len(host), LOOP_PRIME + args.warmup
    K = min(args.steps, F - pre - 2)
    d_imgs = [(torch.from_numpy(L).to(dev), torch.from_numpy(R).to(dev)) for L, R in host[:pre + K + 1]]
    torch.cuda.synchronize()
    dptr = [((a.data_ptr(), W), (b.data_ptr(), W)) for a, b in d_imgs]
    out = {}
    for name in ("host_images", "device_images", "two_calls_no_prefetch"):
        ctx = V.Context(device=local_rank, max_width=W, max_height=H, max_points=cap, n_slots=5, max_level=cfg["max_level"])
        svo = V.StereoVO(ctx, W, H, cfg["K"], cfg["K"], st.T_lr, cfg["n_u"], cfg["n_v"], thres_fastscore=cfg["thres_fast"],
                         window_size=cfg["win"], max_level=cfg["max_level"], thres_error=thr[0], thres_bidirection=thr[1],
                         thres_poseba_error=thr[2], strict_border=args.strict_border, local_ba=bool(args.lba), thres_sampson=cfg.get("thres_sampson", 60.0))
        src = host if name == "host_images" else dptr
        same, stamps, kinds = [True], [], []

        def run(k0, n, keep):
            for k in range(k0, k0 + n):
                if name == "two_calls_no_prefetch":
                    i = svo.result()
                    if k + 1 < len(src):
                        svo.enqueue(*src[k + 1])
                else:
                    i = svo.trackStereoImages(*src[k])
                if keep:
                    stamps.append(time.perf_counter())
                    kinds.append(bool(i.is_keyframe))
                    if k < len(traj):
                        same[0] = same[0] and bool(np.array_equal(np.array(i.T_wc, np.float32).view(np.uint32), traj[k].reshape(-1).view(np.uint32)))

        if name == "two_calls_no_prefetch":
            svo.enqueue(*src[0])
        run(0, pre, False)
        dt = timed(lambda: run(pre, K, True), barrier, ctx)
        if name == "two_calls_no_prefetch":
            svo.result()
        svo.close()
        ctx.close()
        d_ms = np.diff(np.asarray(stamps)) * 1e3
        kf = np.asarray(kinds[1:], bool)
        out[name] = {"value": round(K / dt, 2), "unit": "frames/s", "frames": K, "poses_equal_headline_run": same[0],
                     "mean_ms_keyframe": round(float(d_ms[kf].mean()), 4) if kf.any() else None,
                     "mean_ms_other": round(float(d_ms[~kf].mean()), 4) if (~kf).any() else None}
    out["note"] = ("host_images / device_images: StereoVO::trackStereoImages(left, right, timestamp) as ONE synchronous call per pair "
                   "(vo_svo_track), no look-ahead of any kind, local BA = %s; two_calls_no_prefetch: vo_svo_result(k) then "
                   "vo_svo_enqueue(k + 1) without vo_svo_prefetch" % bool(args.lba))
    return out


def batch_leg(cfg, args, local_rank, torch, V, dev, prm_svo, cap):
    """S independent streams on ONE GPU through vo_batch_* (C++ host threads inside the library): aggregate and per-stream
    frames/s for every S, and the check that a stream's poses and final track ids do not depend on its neighbours."""
    W, H = cfg["W"], cfg["H"]
    nf, warm = args.batch_frames, 8
    d = [[(torch.from_numpy(np.ascontiguousarray(L)).to(dev), torch.from_numpy(np.ascontiguousarray(R)).to(dev)) for L, R in st]
         for st in args.batch_imgs]
    torch.cuda.synchronize()
    Lp = [[a.data_ptr() for a, _ in st] for st in d]
    Rp = [[b.data_ptr() for _, b in st] for st in d]
    alone = []
    for q in range(len(d)):  # every stream by itself: the reference result of that stream
        b = V.StereoBatch(local_rank, 1, W, H, cap, cfg["max_level"], prm_svo)
        alone.append(b.run([Lp[q]], [Rp[q]], W, warmup=warm))
        b.close()
    res = {}
    for S in args.batch_S:
        b = V.StereoBatch(local_rank, S, W, H, cap, cfg["max_level"], prm_svo)
        r = b.run(Lp[:S], Rp[:S], W, warmup=warm)
        mode = b.strict_border()
        b.close()
        same = all(np.array_equal(r["T_wc"][q].view(np.uint32), alone[q]["T_wc"][0].view(np.uint32)) and
                   np.array_equal(r["ids"][q], alone[q]["ids"][0]) for q in range(S))
        per = [(nf - warm) / float(t) for t in r["seconds"]]
        res[str(S)] = {"aggregate_fps": round(S * (nf - warm) / r["wall"], 1), "per_stream_fps_min": round(min(per), 1),
                       "per_stream_fps_max": round(max(per), 1), "poses_and_track_ids_equal_single_stream_run": bool(same),
                       "strict_border_mode": mode}
    res["note"] = (f"{nf - warm} timed frames per stream (closed loop incl. local BA = {bool(args.lba)}; strict-border mode "
                   f"{args.strict_border} asked for, strict_border_mode = what the streams ran with: the arrangements with a replay "
                   "next to the frame kernel become the stream-ordered one when S > 1, same bits), streams of seeds 100.., a "
                   "context + StereoVO + host thread per stream inside libvo_hip.so")
    return res


def secondary_legs(cfg, args, rank, local_rank, torch, V, barrier):
    """Round 2's headline workload, kept so that rounds stay comparable: the frame operator with step [10] closed on the
    device, but on per-frame GROUND-TRUTH track sets (open loop) and a back-and-forth playback of 12 rendered frames."""
    sec = {}
    args.host_images_leg = False
    B = StereoBench(cfg, args, rank, local_rank, torch, V)
    B.ctx.profile_set_classes(1 << 30)
    f0 = B.prime("closed", False)
    w = min(args.warmup, 10)
    B.run(f0, w, "closed", False)
    st = []
    dts = timed(lambda: B.run(f0 + w, args.steps, "closed", False, None, st), barrier, B.ctx)
    sec["r02_ground_truth_track_sets_back_and_forth"] = {"value": round(args.steps / dts, 2), "unit": "frames/s",
                                                         "frame_ms": frame_ms_stats(st)}
    # the reference's own class surface on the same frames: what a maintainer gets by swapping the three classes only
    kc = min(args.steps, 100)
    for name, cached in (("class_surface", True), ("class_surface_uploads_per_call", False)):
        B.run_class_surface(0, 3, cached)
        st = []
        dts = timed(lambda: B.run_class_surface(3, kc, cached, st), barrier, B.ctx)
        sec[name] = {"value": round(kc / dts, 2), "unit": "frames/s", "frame_ms": frame_ms_stats(st),
                     "image_uploads_per_frame": round(B.last_uploads / kc, 2)}
    sec["class_surface"]["note"] = ("one frame = the operator calls of stereo_vo.cpp:483-711 through FeatureTracker / MotionEstimator / "
                                    "FeatureExtractor with host arrays in and out (priors and compactions on the host), ground-truth "
                                    "track sets; images identified by (buffer, size, frame stamp): 3 uploads + pyramids per frame; "
                                    "class_surface_uploads_per_call: no stamp, every call uploads and rebuilds, as the reference does")
    # same frame through the fused operator: the survivor sets must agree
    B.ctx.set_ingest_side_stream(True)
    f0 = B.prime("closed", False)
    keep = []
    args_cpu = args.cpu_frames
    args.cpu_frames = 1
    B.run(f0, 1, "closed", False, keep)
    args.cpu_frames = args_cpu
    cs = B.run_class_surface(f0, 1, True)
    sec["class_surface"]["survivors_equal_fused_operator"] = bool(np.array_equal(keep[0][1]["stage"], cs["stage"]))
    B.ctx.close()
    return sec


def cpu_baseline_loop(cfg, args, st, imgs, traj, V, local_rank):
    """The CPU restatement of the same loop (oracle/stereo_vo.py) on the first frames of the same stream: once in the
    reference's summation order (timed: the cpu_baseline; pose error of the device against it), once in the kernels'
    order next to a second device run (bit-exact track sets: ids, pixels, flags, poses)."""
    from oracle import oracle as O
    from oracle.stereo_vo import StereoVORef
    nf = max(2, min(args.cpu_frames + 1, len(imgs), len(traj)))  # (the device poses of these frames are compared)
    thr = cfg["thres"]
    cores = cpu_threads_for_baseline(O, imgs[0][0], imgs[0][1], cfg)
    border = O.IC_REFERENCE if args.strict_border else O.IC_MASKED

    def make(sum_mode, tw):
        return StereoVORef(cfg["W"], cfg["H"], cfg["K"], cfg["K"], st.T_lr, cfg["n_u"], cfg["n_v"], thres_fast=cfg["thres_fast"],
                           win=cfg["win"], max_level=cfg["max_level"], thres_err=thr[0], thres_bidir=thr[1], thres_poseba=thr[2],
                           lba=bool(args.lba), ic_border=border, sum_mode=sum_mode, tree_width=tw, n_threads=cores,
                           thres_sampson=cfg.get("thres_sampson", 60.0))
    ref = make(O.SUM_SEQ, 0)
    t_frames, worst, seq_ids = [], 0.0, []
    for k in range(nf):
        t0 = time.perf_counter()
        ref.track(*imgs[k])
        t_frames.append(time.perf_counter() - t0)
        seq_ids.append(np.array(ref.ids, copy=True))
        Tg, To = traj[k].astype(np.float64), ref.T_wp.astype(np.float64)
        worst = max(worst, float(np.linalg.norm(Tg - To) / np.linalg.norm(To)))
    # bit-exact leg: a second device run of the first nf frames against the oracle in the kernels' summation order
    tree = make(O.SUM_TREE, 512)
    ctx = V.Context(device=local_rank, max_width=cfg["W"], max_height=cfg["H"], max_points=2 * cfg["n_u"] * cfg["n_v"] + 1024,
                    n_slots=5, max_level=cfg["max_level"])
    svo = V.StereoVO(ctx, cfg["W"], cfg["H"], cfg["K"], cfg["K"], st.T_lr, cfg["n_u"], cfg["n_v"], thres_fastscore=cfg["thres_fast"],
                     window_size=cfg["win"], max_level=cfg["max_level"], thres_error=thr[0], thres_bidirection=thr[1],
                     thres_poseba_error=thr[2], strict_border=args.strict_border, local_ba=bool(args.lba), thres_sampson=cfg.get("thres_sampson", 60.0))
    exact, ids_ok, n_lba, seq_equal = True, True, 0, 0
    nf_par = max(nf, min(args.parity_frames, len(imgs)))  # (long enough for the window to reach three keyframes: local BA)
    for k in range(nf_par):
        gi = svo.trackStereoImages(*imgs[k])
        tree.track(*imgs[k])
        n_lba += int(bool(gi.lba_ran))
        g = svo.getTracks()
        ids_ok = ids_ok and bool(np.array_equal(g["ids"], tree.ids))
        if k < nf and seq_equal == k and np.array_equal(g["ids"], seq_ids[k]):
            seq_equal = k + 1  # (leading frames whose ids also equal the loop's in the REFERENCE's summation order)
        same = ids_ok and np.array_equal(g["pts_l"].view(np.uint32), tree.pts_l.view(np.uint32)) \
            and np.array_equal(g["pts_r"].view(np.uint32), tree.pts_r.view(np.uint32)) and np.array_equal(g["flags"], tree.flags) \
            and np.array_equal(np.array(gi.T_wc, np.float32).view(np.uint32), tree.T_wp.reshape(-1).view(np.uint32))
        exact = exact and bool(same)
    svo.close()
    ctx.close()
    steady = t_frames[1:]
    return {
        "cpu_baseline": {
            "value": round(len(steady) / sum(steady), 3) if steady else None,
            "unit": "frames/s",
            "cores": cores,
            "nproc": os.cpu_count(),
            "cores_note": "threads used = the fastest of 16 / 32 / 64 / all visible CPUs on one PyrLK call (the box schedules a "
                          "share of the host's CPUs)",
            "cpu_model": cpu_model(),
            "kind": "port",
            "sample": f"{len(steady)} steady-state frames of the same stream through the CPU restatement of the closed loop "
                      f"(oracle/stereo_vo.py over oracle/*.c, reference summation order and border semantics); PyrLK and the FAST "
                      f"score over {cores} OpenMP threads, trackWithScale, the BA, the DLT and the local BA single-threaded as in "
                      "the reference",
            "first_frame_s": round(t_frames[0], 3),
        },
        "parity": {"pose_rel_frobenius_max": worst, "track_ids_bit_exact": ids_ok, "track_sets_and_poses_bit_exact": exact,
                   "frames_checked": nf_par, "local_ba_solves_checked": n_lba, "pose_frames_checked": nf,
                   "track_ids_bit_exact_vs_reference_order": f"{seq_equal} of the first {nf} frames",
                   "note": "pose error: device loop against the CPU loop in the reference's summation order, both free-running; "
                           "bit-exactness: device loop against the CPU loop in the kernels' summation order (ids, pixels, flags, "
                           "poses after every frame, local-BA solves included); track ids against the reference-order loop: equal "
                           "until the two free-running loops part at one feature's inlier gate (frame 9 of 24 on the configs[1] stream: "
                           "tests/test_stereo_vo_gpu.py, DESIGN.md \u00a72)"},
    }


def main():
    if os.environ.get("BENCH_WATCHDOG_S"):  # a stuck run prints every thread's Python stack and exits (diagnosis under a profiler)
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["BENCH_WATCHDOG_S"]), exit=True)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed frames (default 400; 40 for --config 4, whose 4K stream is rendered first)")
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", type=int, default=1, choices=sorted(CONFIGS), help="BASELINE.json configs[i]")
    ap.add_argument("--frames", type=int, default=12, help="distinct rendered frames (played back and forth)")
    ap.add_argument("--cpu-frames", type=int, default=8, help="frames timed on the CPU oracle (rank 0, N=1)")
    ap.add_argument("--parity-frames", type=int, default=18, help="loop mode: frames of the bit-exact leg (device loop against the CPU loop)")
    ap.add_argument("--host-loop", choices=("library", "python"), default="library",
                    help="closed-loop modes: who calls result / enqueue / prefetch per frame — the library's sequence loop "
                         "(vo_svo_run / vo_mvo_run, compiled, as the reference's caller is) or this interpreter through ctypes")
    ap.add_argument("--strict-border", type=int, default=4,
                    help="0 masked border taps; 1-5 the reference's never-reset tap state (identical results): 1 replay "
                         "stream-ordered behind the frame kernel, 2 sequential replay, 3 replay next to the frame kernel, "
                         "5 stream-ordered replay behind a gate on the replay stream, 4 (default) 1 or 3 per frame, by the "
                         "previous frame's number of replayed features")
    ap.add_argument("--mode", default=None, choices=("loop", "closed", "sequential", "open"),
                    help="loop (default for --config 1): closed-loop StereoVO on a forward-driving stream; closed (default for "
                         "--config 4: 8000 ground-truth features per frame — the synthetic 4K stream keeps only ~3000 tracks "
                         "alive in the loop) / sequential / open: round 2's "
                         "frame operator on ground-truth track sets, differing in how step [10] is driven; see the docstring")
    ap.add_argument("--no-issue-ahead", action="store_true",
                    help="loop mode: look at a frame's results BEFORE the next frame is sent off (round 3's first loop order)")
    ap.add_argument("--lba", type=int, default=1, help="loop mode: local bundle adjustment at keyframes (the reference's behaviour)")
    ap.add_argument("--no-prefetch", action="store_true",
                    help="loop mode: hand every pair over only when its frame starts (no ingestion under the previous frame)")
    ap.add_argument("--render-workers", type=int, default=0, help="processes that render the synthetic stream (0 = all cores)")
    ap.add_argument("--streams-per-gpu", default="1,2,4,8",
                    help="loop mode, secondary leg: S independent streams on ONE GPU through the library's batch driver "
                         "(vo_batch_*: a context, a StereoVO and a host thread per stream); comma-separated S values, '' = off")
    ap.add_argument("--batch-frames", type=int, default=48, help="frames per stream of that leg (the first 8 untimed)")
    ap.add_argument("--host-images", action="store_true",
                    help="headline loop with host images (default: images resident in HBM; the host-image rate is a "
                         "secondary field of the default run)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary measurements")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--rendezvous-timeout", type=float, default=180.0,
                    help="seconds a rank waits in init_process_group for the others (a rank that never arrives ends the job "
                         "instead of hanging it); the self-launcher ends all ranks after 20x this")
    ap.add_argument("--no-pin", action="store_true", help="leave the ranks' CPU affinity alone (default for N > 1: a contiguous share per rank)")
    ap.add_argument("--force-collective", action="store_true",
                    help="N = 1 only: create the nccl (RCCL) process group with one rank and run the end-of-job all_gather "
                         "through it anyway — the code path N > 1 takes, executed on a 1-GPU box (tests/test_bench_gpu.py)")
    ap.add_argument("--rendezvous-check", action="store_true",
                    help="run only the multi-rank control flow (gloo, no GPU work) and print its JSON line")
    args = ap.parse_args()
    if args.mode is None:
        args.mode = "loop"  # (config 2, mono: the MonoVO loop; --mode closed / open: the frame operator on ground-truth track sets)
    if args.steps is None:
        args.steps = 40 if (args.config == 4 and args.mode == "loop") else 400

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # nothing GPU-related has been imported yet
        sys.exit(launch_ranks(args.gpus, sys.argv[1:], job_timeout=(2.0 if args.rendezvous_check else 20.0) * args.rendezvous_timeout))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but the launcher environment has WORLD_SIZE={world}")
    # every rank keeps to its own share of the host's CPUs (renderer pool, launches, result polling): set before anything
    # forks or spins. LOCAL_WORLD_SIZE is the launcher's (torch.distributed.run and the self-launcher both set it).
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    cpus = sorted(os.sched_getaffinity(0)) if (args.no_pin or world == 1) else pin_rank(local_rank, local_world)
    if args.rendezvous_check:
        return rendezvous_check(rank, world, args.rendezvous_timeout, cpus)

    cfg = CONFIGS[args.config]
    global LOOP_PRIME
    LOOP_PRIME = cfg.get("prime", LOOP_PRIME)  # (a configuration whose keyframes are rarer needs more untimed frames to fill the window)
    loop = args.mode == "loop" and cfg["kind"] == "stereo"
    mono_loop = args.mode == "loop" and cfg["kind"] == "mono"
    imgs = None
    if mono_loop:
        per_rank_workers = max(1, (args.render_workers or host_cores()) // (1 if (world > 1 and not args.no_pin) else max(world, 1)))
        imgs = render_stream(cfg, stream_seed(rank), loop_frames_needed(args), per_rank_workers)
    if loop:  # (before anything GPU-related is imported: the renderer's pool forks)
        # (a pinned rank's host_cores() is already its share)
        per_rank_workers = max(1, (args.render_workers or host_cores()) // (1 if (world > 1 and not args.no_pin) else max(world, 1)))
        imgs = render_stream(cfg, stream_seed(rank), loop_frames_needed(args), per_rank_workers)
        args.batch_S = [int(v) for v in args.streams_per_gpu.split(",") if v.strip()] if (world == 1 and not args.no_secondary) else []
        args.batch_imgs = [render_stream(cfg, 100 + q, args.batch_frames, per_rank_workers) for q in range(max(args.batch_S, default=0))]
    # torch first: its bundled libamdhip64.so.7 must be THE HIP runtime of the process;
    # libvo_hip.so (NEEDED libamdhip64.so.7) then binds to the already-loaded one.
    import torch
    import torch.distributed as dist
    import visual_odometry_ros_amd as V  # loads libvo_hip.so (fails loudly if missing)
    V.load()
    global FORCE_COLLECTIVE
    FORCE_COLLECTIVE = bool(args.force_collective) and world == 1
    collective = world > 1 or FORCE_COLLECTIVE
    if collective:
        import datetime
        if world == 1 and "MASTER_ADDR" not in os.environ:  # (no launcher: a rendezvous of one on the loopback interface)
            import socket
            with socket.socket() as so:
                so.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(so.getsockname()[1])
            os.environ["MASTER_ADDR"] = "127.0.0.1"
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank), rank=rank, world_size=world,
                                timeout=datetime.timedelta(seconds=args.rendezvous_timeout))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    secondary = world == 1 and not args.no_secondary and cfg["kind"] == "stereo" and args.config == 1
    args.host_images_leg = args.host_images or (secondary and not loop)

    def barrier(ctx):
        if collective:
            dist.barrier()
        torch.cuda.synchronize()
        ctx.synchronize()

    if mono_loop:
        out, ctx = run_mono_loop(cfg, args, rank, local_rank, world, torch, V, barrier, dev, imgs)
    elif cfg["kind"] == "mono":
        out, ctx = run_mono(cfg, args, rank, local_rank, world, torch, V, barrier, dev)
    elif loop:
        out, ctx = run_loop(cfg, args, rank, local_rank, world, torch, V, barrier, dev, imgs, secondary)
    else:
        out, ctx = run_stereo(cfg, args, rank, local_rank, world, torch, V, barrier, dev, secondary)
    if rank == 0:
        out["collective"] = "nccl" if collective else "none"
        print(json.dumps(out), flush=True)
    if ctx is not None:
        ctx.close()
    if collective:
        dist.destroy_process_group()


def timed(bench_run, barrier, ctx):
    """The steps between two barriers; the interpreter's generational GC is frozen for the loop (a gen-2 pass with torch
    loaded costs tens of ms and would be charged to one frame; the loop allocates nothing that needs it)."""
    gc.collect()
    gc.freeze()
    gc.disable()
    barrier(ctx)
    t0 = time.perf_counter()
    bench_run()
    barrier(ctx)
    dt = time.perf_counter() - t0
    gc.enable()
    return dt


def run_stereo(cfg, args, rank, local_rank, world, torch, V, barrier, dev, secondary):
    B = StereoBench(cfg, args, rank, local_rank, torch, V)
    ctx, K = B.ctx, args.steps
    mode, host = args.mode, bool(args.host_images)
    # bins that hold a keypoint, per distinct left image: the candidates the closed frame kernel tracks (bytes accounting)
    if mode == "closed":
        for fid in range(B.F):
            ctx.set_image_device(3, B.d_L[fid].data_ptr(), B.W, B.H, B.W)
            B.fe.resetWeightBin()
            B.n_bins_with_kp[fid] = int(B.fe.extractORBwithBinning_fast(3).shape[0])
    first = B.prime(mode, host)
    if args.warmup:
        B.run(first, args.warmup, mode, host)
    first += args.warmup
    ctx.profile_enable(K * 4 + 64)
    all_classes = bool(os.environ.get("VO_BENCH_ALL_KERNELS"))  # (more event brackets: a few us per frame slower)
    ctx.profile_set_classes(0 if all_classes else 1 << 1)  # default: event-bracket only the dominant kernel
    ctx.profile_reset()
    acc = {"b8d": 0, "bdes": 0}
    results, stamps = [], []
    replayed = []  # per frame: features the strict-border replay handled

    def account(step, r):
        cts = r["counts"]  # features finished by the replay kernel do their step [5] there
        if mode == "closed":
            n_cand = B.n_bins_with_kp[B.frame_id(step + 1)]
        else:
            n_cand = N_NEW_OPEN if mode == "open" else 0
        a, b = frame_kernel_bytes(B.win, B.n_pts, cts.n_l0l1, cts.n_refine - cts.n_replayed, n_cand, B.eff_levels,
                                  B.eff_levels_bwd, bool(args.strict_border))
        acc["b8d"] += a
        acc["bdes"] += b
        replayed.append(cts.n_replayed)

    dt = timed(lambda: B.run(first, K, mode, host, results, stamps, account), barrier, ctx)
    if os.environ.get("VO_BENCH_DUMP"):  # per-frame wall time and replayed-feature count, for A/B analysis
        np.save(os.environ["VO_BENCH_DUMP"], np.asarray(stamps))
    per_rank = gather_ranks(K, dt, stream_seed(rank), world, dev)
    tot_frames, max_dt = sum(p[0] for p in per_rank), max(p[1] for p in per_rank)
    if rank != 0:
        return None, ctx

    klt_n, klt_ms = ctx.profile_get(1)
    launches = max(klt_n, 1)
    achieved = (acc["b8d"] / launches) / (klt_ms / launches * 1e-3) / 1e9 if klt_n else 0.0
    # (counter passes of THIS workload on the present kernel sources, or nothing: profiles/r03[_cfgN]_closed_*.json)
    counters = stamped_counters(args.config, "closed") if mode == "closed" else stamped_counters(0, "none")
    traffic = counters["traffic"]
    step10 = {"closed": "closed on the device: per-bucket best keypoints detected from the image alone on the side stream, "
                        "every bucket's candidate tracked inside the frame kernel, updateWeightBin + emission behind the BA",
              "sequential": "three host-driven operator calls after the frame's result",
              "open": "OPEN LOOP: 150 candidates known before the frame (round 1's workload; not a sequential VO)"}[mode]
    r0 = results[0][1]["counts"] if results else None
    out = {
        "metric": cfg["metric"],
        "value": round(tot_frames / max_dt, 2),
        "unit": "frames/s",
        "n_gpus": world,
        "steps": K,
        "warmup": args.warmup,
        "ms_per_step": round(1e3 * max_dt / K, 4),
        "frame_ms": frame_ms_stats(stamps),
        "frame_ms_by_replay": replay_split(stamps, replayed),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u8/i32 (KLT, FAST) + f32 (IC, GN)",
        "data": "synthetic",
        "config": {
            "workload": f"{cfg['name']}: synthetic stereo stream {B.W}x{B.H}, {B.n_pts} tracked features/frame "
                        f"({cfg['n_u']}x{cfg['n_v']} buckets, {int(100 * UNTRIANGULATED)} % of them untriangulated landmarks), "
                        f"win {B.win}, max_level {B.max_level} ({B.eff_levels} effective levels), thresholds of "
                        "config/stereo/kitti_00_stereo.yaml; whole operator sequence of StereoVO::trackStereoImages "
                        "steps [3]-[7] + [10] incl. keypoint detection and bucketing; one independent stream per GPU; "
                        "result read back every frame; track set per frame from the scene's ground truth",
            "step10": step10,
            "images": "pinned host memory, H2D inside the timed region" if host else "resident in HBM",
            "strict_border": int(args.strict_border),
            "untriangulated_fraction": UNTRIANGULATED,
            "distinct_frames": B.F,
        },
        "roofline": {
            "bound": "hbm",
            "kernel": f"frame_track_kernel<{B.win}>",
            "achieved": round(achieved, 2),
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 5),
            "traffic": traffic,
            "alg_bytes_per_launch": {"survey_8d": round(acc["b8d"] / launches), "design_records": round(acc["bdes"] / launches)},
            "avg_launch_us": round(1e3 * klt_ms / launches, 2),
            "note": "achieved = survey_8d bytes / launch duration (HIP events, live); the path is issue/latency-bound, "
                    "the fraction is reported because the metric asks for it (DESIGN.md §6)",
        },
        "roofline_issue": issue_roofline(counters, klt_ms, launches) if klt_n else None,
        "per_rank_fps": [round(p[0] / p[1], 2) for p in per_rank],
    }
    if counters.get("counters_note"):
        out["roofline"]["counters_note"] = counters["counters_note"]
    if all_classes:
        names = {0: "pyramid", 1: "frame_track", 2: "ic_replay", 3: "gn_pose", 4: "hamming", 5: "aux"}
        out["kernels"] = {}
        for cls, nm in names.items():
            n_l, ms = ctx.profile_get(cls)
            if n_l:
                out["kernels"][nm] = {"launches": n_l, "avg_us": round(1e3 * ms / n_l, 2), "us_per_frame": round(1e3 * ms / K, 2)}
    if r0 is not None:
        out["frame_counts_first"] = {f: getattr(r0, f) for f in ("n_l0l1", "n_refine", "n_l1r1", "n_ba", "n_inlier", "n_new_ok",
                                                                 "gn_iterations", "n_replayed")}
    if secondary:
        sec = {}
        legs = [("host_images", "closed", True), ("sequential_step10", "sequential", False),
                ("open_loop_candidates", "open", False)]
        ctx.profile_set_classes(1 << 30)  # no event brackets in the secondary loops
        for name, m, h in legs:
            if (m, h) == (mode, host):
                continue
            f0 = B.prime(m, h)
            B.run(f0, min(args.warmup, 10), m, h)
            f0 += min(args.warmup, 10)
            st = []
            dts = timed(lambda: B.run(f0, K, m, h, None, st), barrier, ctx)
            sec[name] = {"value": round(K / dts, 2), "unit": "frames/s", "frame_ms": frame_ms_stats(st)}
        out["secondary"] = sec
    if world == 1 and not args.no_cpu_baseline:
        out.update(cpu_baseline_stereo(B, args, results, mode))
    return out, ctx


def cpu_baseline_stereo(B, args, results, mode):
    """The oracle (kind "port": the build's CPU restatement of the reference path, reference summation order and border
    semantics) on a bounded sample of the SAME frames, with the per-stage split BASELINE.md §3 promises; doubles as
    the parity check of the timed run (pose within 1e-4 relative Frobenius, survivor stages and new points identical)."""
    from oracle import oracle as O
    thr = B.cfg["thres"]
    prm_o = O.make_stereo_params(B.W, B.H, B.win, B.max_level, thr[0], thr[1], thr[2], B.stream.K, B.stream.K, B.stream.T_lr)
    cores = min(host_cores(), 16)  # the box's CPU share for one GPU; 1500 points do not feed more threads
    border = O.IC_REFERENCE if args.strict_border else O.IC_MASKED
    empty = np.zeros((0, 2), np.float32)
    worst, stage_equal, new_equal, tcpu = 0.0, True, True, 0.0
    stages = {}
    nf = min(args.cpu_frames, len(results))
    us, vs, iu, iv = O.weight_bin_init(B.W, B.H, B.cfg["n_u"], B.cfg["n_v"])
    t_pyr = 0.0
    for step, g in results[:nf]:
        a, b = B.frame_id(step), B.frame_id(step + 1)
        ts = B.track_sets[(a, b)]
        L0, (L1, R1) = B.imgs[a][0], B.imgs[b]
        t1 = time.perf_counter()
        cand = ts["pts_new"] if mode == "open" else empty
        o = O.stereo_frame(prm_o, L0, L1, R1, ts["pts_l0"], ts["pts_r0"], ts["Xp"], ts["dT_prior"], cand, O.SUM_SEQ, 0, border,
                           cores, lm_flags=ts["flags"])
        t2 = time.perf_counter()
        for k, v in O.stereo_frame_stage_ms().items():
            stages[k] = stages.get(k, 0.0) + v
        if mode != "open":  # step [10] in the reference's order, after the frame
            w = O.weight_bin_update(o["pts_l1"][o["stage"] == 4], us, vs, B.cfg["n_u"], B.cfg["n_v"])
            t3 = time.perf_counter()
            d = O.orb_detect(L1, B.cfg["thres_fast"])
            t4 = time.perf_counter()
            pn, _ = O.bucket_argmax(d["xy"], d["response"], iu, iv, B.cfg["n_u"], B.cfg["n_v"], w)
            t5 = time.perf_counter()
            _, pnr, mn = O.track_bidirection(L1, R1, pn, B.win, B.max_level, thr[0], thr[1], None, cores)
            t6 = time.perf_counter()
            for k, v in (("update_weight_bin", t3 - t2), ("orb_detect", t4 - t3), ("bucket_argmax", t5 - t4),
                         ("klt_new_points", t6 - t5)):
                stages[k] = stages.get(k, 0.0) + 1e3 * v
            tcpu += t6 - t1
            gn = g["new"] if mode == "sequential" else (g["pts_new_r"], g["mask_new"])
            new_equal = new_equal and bool(np.array_equal(gn[1], mn)) and bool(np.array_equal(np.asarray(gn[0]), pnr))
            if mode == "closed":
                new_equal = new_equal and bool(np.array_equal(g["pts_new"], pn))
        else:
            tcpu += t2 - t1
            new_equal = new_equal and bool(np.array_equal(g["mask_new"], o["mask_new"]))
        tp = time.perf_counter()
        O.build_pyramid(L1, B.win, B.max_level)
        t_pyr += time.perf_counter() - tp
        e = float(np.linalg.norm(g["dT"].astype(np.float64) - o["dT"]) / np.linalg.norm(o["dT"]))
        worst = max(worst, e)
        stage_equal = stage_equal and bool(np.array_equal(g["stage"], o["stage"]))
    per_stage = {k: round(v / max(nf, 1), 3) for k, v in stages.items()}
    per_stage["pyramid_one_image"] = round(1e3 * t_pyr / max(nf, 1), 3)
    return {
        "cpu_baseline": {
            "value": round(nf / tcpu, 3) if tcpu > 0 else None,
            "unit": "frames/s",
            "cores": cores,
            "nproc": os.cpu_count(),
            "cpu_model": cpu_model(),
            "kind": "port",
            "sample": f"{nf} frames of the same stream and workload on the CPU restatement (oracle/, reference summation "
                      f"order and border semantics); PyrLK and the FAST score over {cores} OpenMP threads, trackWithScale and "
                      "the BA single-threaded as in the reference",
            "per_stage_ms": per_stage,
            "per_stage_note": "every klt_* stage builds its two image pyramids itself, as each cv::calcOpticalFlowPyrLK call "
                              "of the reference does (8 pyramid builds per frame; pyramid_one_image = one of them); "
                              "track_with_scale includes the convertTo / Sobel images",
        },
        "parity": {"pose_rel_frobenius_max": worst, "survivor_sets_bit_exact": stage_equal,
                   "new_points_bit_exact": new_equal, "frames_checked": nf},
    }


class TruePoseHook:
    """Stands for MotionEstimator::calcPose5PointsAlgorithm (OpenCV calib3d, out of scope): the scene's true relative pose of
    frame k against k - 1, every pair an inlier. `k` is set by the loop before each frame."""

    def __init__(self, poses):
        self.poses, self.k, self.calls = poses, 0, 0

    def __call__(self, pts0, pts1):
        self.calls += 1
        T10 = np.linalg.inv(self.poses[self.k]) @ self.poses[self.k - 1]
        return True, T10[:3, :3].astype(np.float32), T10[:3, 3].astype(np.float32), np.ones(len(pts0), bool)


def run_mono_loop(cfg, args, rank, local_rank, world, torch, V, barrier, dev, imgs):
    """BASELINE configs[2] as a CLOSED LOOP: MonoVO::trackImage (mono_vo.cpp:496-1194) on a forward-driving mono stream, the
    track set with every landmark's first observation / age / parallax, the keyframes and the mono local BA carried on the
    device (vo_mvo_*); the 5-point pose of the initialisation comes from a caller hook."""
    from visual_odometry_ros_amd import synthetic as S
    W, H, win, lvl = cfg["W"], cfg["H"], cfg["win"], cfg["max_level"]
    n_bins = cfg["n_u"] * cfg["n_v"]
    st = S.StereoStream(width=W, height=H, K=cfg["K"], n_u=cfg["n_u"], n_v=cfg["n_v"], seed=stream_seed(rank), speed=cfg["speed"])
    mono = [L for L, _ in imgs]
    F = len(mono)
    poses_gt = st.poses(F)
    d_I = [torch.from_numpy(np.ascontiguousarray(I)).to(dev) for I in mono]
    torch.cuda.synchronize()
    cap = 2 * n_bins + 512
    ctx = V.Context(device=local_rank, max_width=W, max_height=H, max_points=cap, n_slots=3, max_level=lvl)
    if host_cores() < 3:
        ctx.debug_set(ctx.OPT_POLL_YIELD, 1)
    thr = cfg["thres"]
    hook = TruePoseHook(poses_gt)
    kw = dict(thres_fastscore=cfg["thres_fast"], window_size=win, max_level=lvl, thres_error=thr[0], thres_bidirection=thr[1],
              thres_poseba_error=thr[2], thres_sampson=thr[3], thres_parallax=1.0, thres_translation=cfg.get("kf_trans", 3.0),
              strict_border=int(args.strict_border), local_ba=bool(args.lba))
    mvo = V.MonoVO(ctx, W, H, cfg["K"], cfg["n_u"], cfg["n_v"], hook, **kw)
    eff_levels = ctx.pyramid_levels(W, H, win, lvl) + 1
    eff_levels_bwd = ctx.pyramid_levels(W, H, win, lvl - 1) + 1
    fe = V.FeatureExtractor(ctx)
    fe.initParams(W, H, cfg["n_u"], cfg["n_v"], THRES_FAST=cfg["thres_fast"])
    n_kp_bins = []
    for k in range(F):
        ctx.set_image_device(2, d_I[k].data_ptr(), W, H, W)
        n_kp_bins.append(bins_with_keypoints(fe, 2))
    ctx.synchronize()
    ptr = [np.ascontiguousarray(I) for I in mono] if args.host_images else [(d_I[k].data_ptr(), W) for k in range(F)]
    prefetch = not args.no_prefetch
    state = {"k": 0}
    traj, infos, stamps = [], [], []

    def issue(k):
        hook.k = k
        mvo.enqueue(ptr[k])
        if prefetch and k + 1 < F:
            mvo.prefetch(ptr[k + 1])

    def step(keep):
        k = state["k"]
        hook.k = k
        i = mvo.result()
        if keep:
            stamps.append(time.perf_counter())
        if k + 1 < F:
            issue(k + 1)
        state["k"] = k + 1
        if not keep:
            kf_before[0] += int(bool(i.is_keyframe))
        if keep:
            c = i.counts
            infos.append((k, i.n_tracks_in, c.n_klt, c.n_refine, c.n_replayed, i.n_new, i.n_final, i.is_keyframe, i.lba_ran,
                          c.gn_iterations, c.n_ba, i.used_five_point))
        traj.append(np.array(i.T_wc, np.float32).reshape(4, 4))

    # (the library's sequence loop behind the initialisation: the 5-point hook is fed the frame index from here)
    lib_loop = args.host_loop == "library" and prefetch and not args.host_images  # (runSequence takes device addresses)
    kf_before = [0]
    issue(0)
    for _ in range(LOOP_PRIME):
        step(False)
    K = args.steps

    def lib_frames(n, keep):
        k0 = state["k"]
        out, st_k = mvo.runSequence(ptr, k0, k0 + n)
        for j, i in enumerate(out):
            if i.used_five_point:
                raise RuntimeError("the 5-point fallback inside the library loop: the hook was not told the frame index (--host-loop python)")
            if not keep:
                kf_before[0] += int(bool(i.is_keyframe))
            if keep:
                c = i.counts
                infos.append((k0 + j, i.n_tracks_in, c.n_klt, c.n_refine, c.n_replayed, i.n_new, i.n_final, i.is_keyframe, i.lba_ran,
                              c.gn_iterations, c.n_ba, i.used_five_point))
            traj.append(np.array(i.T_wc, np.float32).reshape(4, 4))
        if keep:
            stamps.extend(st_k.tolist())
        state["k"] = k0 + n

    if lib_loop and args.warmup > 0:
        lib_frames(args.warmup, False)
    else:
        for _ in range(args.warmup):
            step(False)
    ctx.profile_enable(K * 4 + 64)
    ctx.profile_set_classes(1 << 1)
    ctx.profile_reset()
    dt = timed((lambda: lib_frames(K, True)) if lib_loop else (lambda: [step(True) for _ in range(K)]), barrier, ctx)
    if state["k"] < F:
        hook.k = state["k"]
        mvo.result()
    per_rank = gather_ranks(K, dt, stream_seed(rank), world, dev)
    tot_frames, max_dt = sum(p[0] for p in per_rank), max(p[1] for p in per_rank)
    if rank != 0:
        return None, ctx
    klt_n, klt_ms = ctx.profile_get(1)
    launches = max(klt_n, 1)
    I = np.array(infos, np.int64)
    b8d = bdes = 0
    for row in I:
        k, n_in, n_klt = int(row[0]), int(row[1]), int(row[2])
        n_cand = n_kp_bins[k]
        b8d += klt_bytes_per_point_level(win) * (2 * n_in * eff_levels + n_cand * (eff_levels + eff_levels_bwd)) + IC_BYTES_8D * n_klt
        bdes += (IC_RECORD_BYTES * n_in if args.strict_border else 0) + POINT_IO_BYTES * (n_in + n_cand)
    achieved = (b8d / launches) / (klt_ms / launches * 1e-3) / 1e9 if klt_n else 0.0
    counters = stamped_counters(args.config, "loop", "mono_track")
    # trajectory against the renderer's ground truth, up to the scale the initialisation fixes (|t| of the first motion = 1)
    T0i = np.linalg.inv(poses_gt[0])
    gt = np.stack([(T0i @ p)[:3, 3] for p in poses_gt[:len(traj)]])
    est = np.stack([T[:3, 3] for T in traj]).astype(np.float64)
    sc = np.linalg.norm(gt[1]) if len(gt) > 1 else 1.0
    d_ms = np.diff(np.asarray(stamps)) * 1e3
    kfm = I[1:, 7] > 0
    out = {
        "metric": cfg["metric"], "value": round(tot_frames / max_dt, 2), "unit": "frames/s", "n_gpus": world, "steps": K,
        "warmup": args.warmup, "ms_per_step": round(1e3 * max_dt / K, 4), "frame_ms": frame_ms_stats(stamps),
        "frame_ms_by_kind": {"keyframes": int(kfm.sum()), "mean_ms_keyframe": round(float(d_ms[kfm].mean()), 4) if kfm.any() else None,
                             "mean_ms_other": round(float(d_ms[~kfm].mean()), 4) if (~kfm).any() else None},
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u8/i32 (KLT, FAST) + f32 (IC, GN, DLT) + f64 (local BA)", "data": "synthetic",
        "config": {"workload": f"{cfg['name']}: MonoVO::trackImage on a synthetic forward-driving mono stream {W}x{H} ({cfg['speed']} m/frame, "
                               f"left images of the stereo renderer), {cfg['n_u']}x{cfg['n_v']} buckets, win {win}, max_level {lvl} "
                               f"({eff_levels} effective levels), thresholds of config/mono/mono0.yaml; the whole call: prior + scale, "
                               "trackBidirectionWithPrior, trackWithScale, pose-only BA, Sampson gate, new points, landmark bookkeeping "
                               "(age, parallax), keyframe rule, reconstruction"
                               + (", mono local BA over the keyframe window" if args.lba else "; local BA off (--lba 0)")
                               + "; first image + initialisation (5-point pose from a caller hook) happen before the timed frames",
                   "track_set": "closed loop", "playback": "forward", "images": "pageable host arrays, H2D inside the timed region" if args.host_images else "resident in HBM", "local_ba": bool(args.lba),
                   "strict_border": int(args.strict_border), "next_image": "prefetched (vo_mvo_prefetch)" if prefetch else "with its frame",
                   "host_loop": "library (vo_mvo_run)" if lib_loop else "python (ctypes calls per frame)"},
        "loop": {"mean_tracks_in": round(float(I[:, 1].mean()), 1), "mean_final": round(float(I[:, 6].mean()), 1),
                 "mean_new_landmarks": round(float(I[:, 5].mean()), 1), "mean_ba_set": round(float(I[:, 10].mean()), 1),
                 "mean_gn_iterations": round(float(I[:, 9].mean()), 2), "keyframes": int((I[:, 7] > 0).sum()), "lba_runs": int((I[:, 8] > 0).sum()),
                 "five_point_calls": int(hook.calls), "keyframes_before_timed_region": kf_before[0],
                 "untimed_frames": LOOP_PRIME + args.warmup,
                 "end_point_error_scaled": round(float(np.linalg.norm(est[-1] * sc - gt[-1])), 4), "path_m": round(float(np.linalg.norm(gt[-1])), 1)},
        "roofline": {"bound": "hbm", "kernel": f"mono_track_kernel<{win}>", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": counters["traffic"],
                     "alg_bytes_per_launch": {"survey_8d": round(b8d / launches), "design_records": round(bdes / launches)},
                     "avg_launch_us": round(1e3 * klt_ms / launches, 2),
                     "note": "algorithmic bytes per SURVEY 8(d): forward + backward PyrLK of every feature, every bin's candidate forward + "
                             "backward (speculative), IC tiles; the path is issue/latency-bound at these sizes"},
        "roofline_issue": issue_roofline(counters, klt_ms, launches),
        "per_rank_fps": [round(p[0] / p[1], 2) for p in per_rank],
    }
    if counters["counters_note"]:
        out["roofline"]["counters_note"] = counters["counters_note"]
    if ctx.debug_switches:
        out["config"]["debug_switches"] = dict(ctx.debug_switches)
    mvo.close()
    ctx.close()
    if world == 1 and not args.no_cpu_baseline:
        from oracle import oracle as O
        from oracle.mono_vo import MonoVORef
        nf = max(3, min(args.cpu_frames + 2, F, len(traj)))
        cores = min(host_cores(), 16)
        rh = TruePoseHook(poses_gt)
        ref = MonoVORef(W, H, cfg["K"], cfg["n_u"], cfg["n_v"], rh, thres_fast=cfg["thres_fast"], win=win, max_level=lvl, thres_err=thr[0],
                        thres_bidir=thr[1], thres_poseba=thr[2], thres_sampson=thr[3], thres_parallax_deg=1.0,
                        kf_trans=cfg.get("kf_trans", 3.0), lba=bool(args.lba), ic_border=O.IC_REFERENCE if args.strict_border else O.IC_MASKED,
                        sum_mode=O.SUM_SEQ, tree_width=0, n_threads=cores)
        t_frames, worst = [], 0.0
        for k in range(nf):
            rh.k = k
            t0 = time.perf_counter()
            ref.track(mono[k])
            t_frames.append(time.perf_counter() - t0)
            Tg, To = traj[k].astype(np.float64), ref.frames[k]["T_wc"].astype(np.float64)
            worst = max(worst, float(np.linalg.norm(Tg - To) / np.linalg.norm(To)))
        steady = t_frames[2:]
        out["cpu_baseline"] = {"value": round(len(steady) / sum(steady), 3) if steady else None, "unit": "frames/s", "cores": cores,
                               "nproc": os.cpu_count(), "cpu_model": cpu_model(), "kind": "port",
                               "sample": f"{len(steady)} steady-state frames of the same stream through the CPU restatement of the loop "
                                         f"(oracle/mono_vo.py over oracle/*.c, reference summation order and border semantics), PyrLK and "
                                         f"the FAST score over {cores} OpenMP threads, the rest single-threaded as in the reference"}
        out["parity"] = {"pose_rel_frobenius_max": worst, "pose_frames_checked": nf,
                         "note": "device loop against the CPU loop in the reference's summation order, both free-running from the same "
                                 "5-point hook; the bit-level comparison of the track sets is tests/test_mono_vo_gpu.py"}
    return out, None


def run_mono(cfg, args, rank, local_rank, world, torch, V, barrier, dev):
    """BASELINE configs[2]: the mono frame operator (mono_vo.cpp:739-963) at 752x480 / 1000 features / win 15 / 5 levels."""
    from visual_odometry_ros_amd import synthetic as S
    from visual_odometry_ros_amd.api import MonoFramePipeline, make_mono_params
    W_, H_, WIN, LVL = cfg["W"], cfg["H"], cfg["win"], cfg["max_level"]
    n_pts = cfg["n_u"] * cfg["n_v"]
    stream = S.StereoStream(width=W_, height=H_, K=cfg["K"], n_u=cfg["n_u"], n_v=cfg["n_v"], n_new=50,
                            seed=stream_seed(rank) + 1, speed=cfg["speed"], margin=cfg["margin"])
    F = max(args.frames, 3)
    poses = stream.poses(F)
    imgs = [stream.render_pair(p)[0] for p in poses]
    order = list(range(F)) + list(range(F - 2, 0, -1))
    fid = lambda s: order[s % len(order)]  # noqa: E731
    sets = {}
    rng = np.random.default_rng(1)
    for s in range(len(order)):
        a, b = fid(s), fid(s + 1)
        if (a, b) in sets:
            continue
        ts = stream.track_set(a * 131 + b, poses[a], poses[b])
        n = ts["pts_l0"].shape[0]
        flags = ((rng.random(n) < 0.7).astype(np.uint8) | ((rng.random(n) < 0.8).astype(np.uint8) << 1)).astype(np.uint8)
        dT = ts["dT_prior"].astype(np.float32)
        sets[(a, b)] = dict(pts0=ts["pts_l0"], Xw=ts["Xp"].astype(np.float32), flags=flags, Tcw_prev=np.eye(4, dtype=np.float32),
                            Tcw_prior=np.linalg.inv(dT.astype(np.float64)).astype(np.float32), dT=dT)
    d_I = [torch.from_numpy(np.ascontiguousarray(I)).to(dev) for I in imgs]
    d_s = {k: {q: torch.from_numpy(np.ascontiguousarray(v[q])).to(dev) for q in ("pts0", "Xw", "flags")} for k, v in sets.items()}
    torch.cuda.synchronize()
    ctx = V.Context(device=local_rank, max_width=W_, max_height=H_, max_points=n_pts + 64, n_slots=3, max_level=LVL)
    thr = cfg["thres"]
    args_p = (W_, H_, WIN, LVL, thr[0], thr[1], thr[2], thr[3], cfg["K"])
    pipe = MonoFramePipeline(ctx, make_mono_params(*args_p), strict_border=args.strict_border)
    ctx.set_pyramid_window_hint(WIN)
    ctx.set_ingest_side_stream(True)
    eff_levels = ctx.pyramid_levels(W_, H_, WIN, LVL) + 1
    eff_levels_bwd = ctx.pyramid_levels(W_, H_, WIN, LVL - 1) + 1
    slot = {"P": 0, "C": 1, "N": 2}
    # --mode closed (default): the frame includes its new-point step (mono_vo.cpp:977-1001) closed on the device, the
    # candidate table of the next image is built on the side stream while the frame runs; --mode open: the frame alone
    closed = args.mode == "closed"
    fe = V.FeatureExtractor(ctx)
    fe.initParams(W_, H_, cfg["n_u"], cfg["n_v"], THRES_FAST=cfg["thres_fast"])
    bins = fe.binParams()
    n_bins_with_kp = {}
    if closed:
        for f_ in range(F):
            ctx.set_image_device(2, d_I[f_].data_ptr(), W_, H_, W_)
            fe.resetWeightBin()
            n_bins_with_kp[f_] = int(fe.extractORBwithBinning_fast(2).shape[0])

    def enqueue(s):
        k = (fid(s), fid(s + 1))
        t, h = d_s[k], sets[k]
        if closed:
            pipe.enqueue_closed_device(t["pts0"].data_ptr(), t["Xw"].data_ptr(), t["flags"].data_ptr(), n_pts, h["Tcw_prev"],
                                       h["Tcw_prior"], h["dT"], bins, s & 1, slots=(slot["P"], slot["C"]))
        else:
            pipe.enqueue_device(t["pts0"].data_ptr(), t["Xw"].data_ptr(), t["flags"].data_ptr(), n_pts, h["Tcw_prev"],
                                h["Tcw_prior"], h["dT"], slots=(slot["P"], slot["C"]))
        ctx.set_image_device(slot["N"], d_I[fid(s + 2)].data_ptr(), W_, H_, W_)
        if closed:
            fe.enqueueCandidates(slot["N"], (s + 1) & 1)

    def run(first, count, keep=None, stamps=None, on_result=None):
        if count <= 0:
            return
        enqueue(first)
        for s in range(first, first + count):
            r = pipe.result()
            slot["P"], slot["C"], slot["N"] = slot["C"], slot["N"], slot["P"]
            if s + 1 < first + count:
                enqueue(s + 1)
            if stamps is not None:
                stamps.append(time.perf_counter())
            if keep is not None and len(keep) < max(args.cpu_frames, 1):
                keep.append((s, r))
            if on_result is not None:
                on_result(s, r)

    ctx.set_image_device(0, d_I[fid(0)].data_ptr(), W_, H_, W_)
    ctx.set_image_device(1, d_I[fid(1)].data_ptr(), W_, H_, W_)
    if closed:
        fe.enqueueCandidates(1, 0)
    ctx.synchronize()
    run(0, 4)
    run(4, args.warmup)
    K = args.steps
    ctx.profile_enable(K * 4 + 64)
    ctx.profile_set_classes(1 << 1)
    ctx.profile_reset()
    kept, stamps = [], []
    acc = {"b8d": 0, "bdes": 0}

    def account(s, r):
        c = r["counts"]  # forward + backward PyrLK at full depth for every feature, IC for the tracked ones
        n_cand = n_bins_with_kp[fid(s + 1)] if closed else 0  # + forward / backward (maxLevel - 1) for every bin's candidate
        acc["b8d"] += klt_bytes_per_point_level(WIN) * (2 * n_pts * eff_levels + n_cand * (eff_levels + eff_levels_bwd)) \
            + IC_BYTES_8D * c.n_klt
        acc["bdes"] += (IC_RECORD_BYTES * n_pts if args.strict_border else 0) + POINT_IO_BYTES * (n_pts + n_cand)

    first = 4 + args.warmup
    dt = timed(lambda: run(first, K, kept, stamps, account), barrier, ctx)
    per_rank = gather_ranks(K, dt, stream_seed(rank), world, dev)
    tot_frames, max_dt = sum(p[0] for p in per_rank), max(p[1] for p in per_rank)
    if rank != 0:
        return None, ctx
    nl, ms = ctx.profile_get(1)
    launches = max(nl, 1)
    achieved = (acc["b8d"] / launches) / (ms / launches * 1e-3) / 1e9 if nl else 0.0
    out = {
        "metric": cfg["metric"], "value": round(tot_frames / max_dt, 2), "unit": "frames/s", "n_gpus": world, "steps": K,
        "warmup": args.warmup, "ms_per_step": round(1e3 * max_dt / K, 4), "frame_ms": frame_ms_stats(stamps),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8/i32 (KLT) + f32 (IC, GN)",
        "data": "synthetic",
        "config": {"workload": f"{cfg['name']}: synthetic mono stream {W_}x{H_} (left images of the stereo renderer), {n_pts} "
                               f"tracked features/frame ({cfg['n_u']}x{cfg['n_v']} buckets), win {WIN}, max_level {LVL} "
                               f"({eff_levels} effective levels), thresholds of config/mono/mono0.yaml; the steady state of "
                               "MonoVO::trackImage (mono_vo.cpp:739-963: prior + scale, trackBidirectionWithPrior, "
                               "trackWithScale, pose-only BA, mask_motion, Sampson gate)"
                               + (" + its new-point step (:977-1001: updateWeightBin, keypoint detection and bucketing, "
                                  "trackBidirection) closed on the device" if closed else "; the new-point step is not part of it")
                               + "; result read back every frame; the 5-point fallback is the caller's",
                   "images": "resident in HBM", "strict_border": int(args.strict_border), "distinct_frames": F},
        "roofline": {"bound": "hbm", "kernel": f"mono_track_kernel<{WIN}>", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": None,
                     "alg_bytes_per_launch": {"survey_8d": round(acc["b8d"] / launches), "design_records": round(acc["bdes"] / launches)},
                     "avg_launch_us": round(1e3 * ms / launches, 2)},
        "per_rank_fps": [round(p[0] / p[1], 2) for p in per_rank],
    }
    if world == 1 and not args.no_cpu_baseline and kept:
        from oracle import oracle as O
        prm_o = O.make_mono_params(*args_p)
        cores = min(host_cores(), 16)
        border = O.IC_REFERENCE if args.strict_border else O.IC_MASKED
        kept = kept[:args.cpu_frames]
        ok, worst, new_ok = True, 0.0, True
        if closed:
            us, vs, iu, iv = O.weight_bin_init(W_, H_, cfg["n_u"], cfg["n_v"])
        t0 = time.perf_counter()
        for s, r in kept:
            k = (fid(s), fid(s + 1))
            h = sets[k]
            o = O.mono_frame(prm_o, imgs[k[0]], imgs[k[1]], h["pts0"], h["Xw"], h["flags"], h["Tcw_prev"], h["Tcw_prior"],
                             h["dT"], O.SUM_SEQ, 0, border, cores)
            ok = ok and bool(np.array_equal(o["stage"], r["stage"]))
            worst = max(worst, float(np.linalg.norm(r["dT01"].astype(np.float64) - o["dT01"]) / np.linalg.norm(o["dT01"])))
            if closed:  # the reference's order: after the frame, on its final set
                w = O.weight_bin_update(o["pts1"][o["stage"] == 4], us, vs, cfg["n_u"], cfg["n_v"])
                d = O.orb_detect(imgs[k[1]], cfg["thres_fast"])
                cand, _ = O.bucket_argmax(d["xy"], d["response"], iu, iv, cfg["n_u"], cfg["n_v"], w)
                _, p0n, mk = O.track_bidirection(imgs[k[1]], imgs[k[0]], cand, WIN, LVL, thr[0], thr[1], None, cores)
                new_ok = new_ok and bool(np.array_equal(r["pts1_new"], cand) and np.array_equal(r["mask_new"], mk)
                                         and np.array_equal(r["pts0_new"].view(np.uint32), p0n.view(np.uint32)))
        cdt = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": round(len(kept) / cdt, 3), "unit": "frames/s", "cores": cores, "nproc": os.cpu_count(),
                               "cpu_model": cpu_model(), "kind": "port",
                               "sample": f"{len(kept)} frames of the same stream on the CPU restatement (oracle_mono.c)"}
        out["parity"] = {"pose_rel_frobenius_max": worst, "survivor_sets_bit_exact": ok, "frames_checked": len(kept)}
        if closed:
            out["parity"]["new_points_bit_exact"] = new_ok
    return out, ctx


if __name__ == "__main__":
    main()
