#!/usr/bin/env python3
"""bench.py — stereo VO frames/s on the synthetic KITTI-shaped stream (BASELINE.json
configs[1]: 1241x376, 1500 tracked features per frame), one image stream per GPU.

A "step" is one steady-state stereo frame through the hot path: pyramids of the new
left/right images, trackWithPrior l0->l1, trackWithScale, trackWithPrior l1->r1,
stereo pose-only GN, and trackBidirection of the new-point candidates — the operator
sequence of StereoVO::trackStereoImages (stereo_vo.cpp:483-711) — with the result
(pose, survivors) read back to the host every frame, as a sequential VO needs it.
Inputs (images and track sets) are resident in HBM before the timed region.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: one rank per GPU, RCCL gather of the totals; under torch.distributed.run the ranks are the launcher's,
   without a launcher environment bench.py starts the N ranks itself as fresh child processes)

Prints ONE JSON line on rank 0 (see DESIGN.md §Measurement for every field).
"""
import argparse
import gc
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WIN, MAX_LEVEL = 21, 6
THRES_ERR, THRES_BIDIR, THRES_POSEBA = 80.0, 0.5, 3.0
N_U, N_V, N_NEW = 60, 25, int(os.environ.get("VO_BENCH_NNEW", "150"))
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec


def klt_bytes_per_point_level(win):
    # u8 template tile incl. Scharr halo + one u8 search tile
    return (win + 2) ** 2 + (win + 1) ** 2


IC_BYTES_PER_POINT = 27 * 32 + 40 * 44          # u8 template tile (Sobel halo) + u8 search tile
IC_RECORD_BYTES = (3 + 1) * 264 * 4 + 2 * 9 * 4  # strict border: tap values + tap masks written per feature
POINT_IO_BYTES = 64                              # landmark, pixels, priors in; pixels, stage out


def frame_kernel_bytes(win, n, n_l0l1, n_step5, n_new, levels, levels_bwd, strict):
    """Algorithmic bytes of ONE frame_track_kernel launch (kernel table): the tiles every
    feature has to see once per pyramid level and pass, the IC tiles, the tap records."""
    klt = klt_bytes_per_point_level(win) * ((n + n_step5) * levels + n_new * (levels + levels_bwd))
    return klt + IC_BYTES_PER_POINT * n_l0l1 + (IC_RECORD_BYTES * n if strict else 0) + POINT_IO_BYTES * (n + n_new)


def gather_ranks(frames, seconds, seed, world, device=None):
    """The one collective of the job: an all_gather of {frames, seconds, stream seed} per rank
    (RCCL on GPUs, gloo in the CPU tests), 24 B per rank. Returns the per-rank list [(frames, seconds, seed)]."""
    if world <= 1:
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


def launch_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher environment: this (parent) process starts N FRESH child
    interpreters, one per GPU, before it has imported torch or made any HIP call — it never touches a GPU and no
    GPU-initialised process is ever re-exec'ed. The children rendezvous on 127.0.0.1; rank 0 prints the one JSON
    line, which is relayed. Returns the exit code (first non-zero child code)."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    procs = []
    for r in range(n):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=e,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    # wait for all ranks; if one fails, the others would sit in the rendezvous / barrier forever: end them
    # (rank 0 writes one line; a pipe of that size cannot fill up while we poll)
    rc = 0
    while any(p.poll() is None for p in procs):
        bad = [p for p in procs if p.poll() not in (None, 0)]
        if bad:
            rc = bad[0].returncode
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
    sys.stdout.flush()
    return rc


def rendezvous_check(rank, world):
    """--rendezvous-check: the N > 1 control flow of this file (launcher, rank environment, process group,
    per-rank stream seeds, the gather, rank-0-only JSON) with gloo and no HIP call, so that it runs on a
    machine without GPUs (tests/test_bench_dist.py)."""
    import torch.distributed as dist
    if world > 1:
        dist.init_process_group(backend="gloo")
    per = gather_ranks(100 + rank, 0.5 + 0.25 * rank, stream_seed(rank), world)
    if world > 1:
        dist.barrier()
    if rank == 0:
        tot, mx = sum(p[0] for p in per), max(p[1] for p in per)
        print(json.dumps({"rendezvous_check": True, "n_gpus": world, "value": tot / mx,
                          "per_rank": [{"frames": p[0], "seconds": p[1], "stream_seed": p[2]} for p in per]}),
              flush=True)
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--frames", type=int, default=12, help="distinct rendered frames (played back and forth)")
    ap.add_argument("--cpu-frames", type=int, default=8, help="frames timed on the CPU oracle (rank 0, N=1)")
    ap.add_argument("--strict-border", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--detect", action="store_true",
                    help="also run the keypoint detection + bucketing of the current left image every frame "
                         "(side stream, overlapping the frame operator); not part of the default workload")
    ap.add_argument("--rendezvous-check", action="store_true",
                    help="run only the multi-rank control flow (gloo, no GPU work) and print its JSON line")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))  # nothing GPU-related has been imported yet
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but the launcher environment has WORLD_SIZE={world}")
    if args.rendezvous_check:
        return rendezvous_check(rank, world)

    # torch first: its bundled libamdhip64.so.7 must be THE HIP runtime of the process;
    # libvo_hip.so (NEEDED libamdhip64.so.7) then binds to the already-loaded one.
    import torch
    import torch.distributed as dist
    import visual_odometry_ros_amd as V  # loads libvo_hip.so (fails loudly if missing)
    from visual_odometry_ros_amd import synthetic as S
    from visual_odometry_ros_amd.api import StereoFramePipeline, make_stereo_params
    V.load()

    if world > 1:
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    # ---- synthetic stream (one independent sequence per rank), rendered on the host ----
    stream = S.StereoStream(n_u=N_U, n_v=N_V, n_new=N_NEW, seed=stream_seed(rank))
    F = max(args.frames, 3)
    poses = stream.poses(F)
    imgs = [stream.render_pair(p)[:2] for p in poses]
    # back-and-forth playback: 0,1,..,F-1,F-2,..,1,0,1,...
    order = list(range(F)) + list(range(F - 2, 0, -1))
    W_, H_ = stream.width, stream.height

    def frame_id(step):
        return order[step % len(order)]

    track_sets = {}
    for s in range(len(order)):
        a, b = frame_id(s), frame_id(s + 1)
        if (a, b) not in track_sets:
            track_sets[(a, b)] = stream.track_set(a * 131 + b, poses[a], poses[b])

    # ---- everything resident in HBM before timing ----
    d_L = [torch.from_numpy(np.ascontiguousarray(L)).to(dev) for L, _ in imgs]
    d_R = [torch.from_numpy(np.ascontiguousarray(R)).to(dev) for _, R in imgs]
    d_ts = {}
    for key, ts in track_sets.items():
        d_ts[key] = {k: torch.from_numpy(np.ascontiguousarray(ts[k])).to(dev)
                     for k in ("pts_l0", "pts_r0", "Xp", "pts_new")}
    torch.cuda.synchronize()

    n_pts = N_U * N_V
    ctx = V.Context(device=local_rank, max_width=W_, max_height=H_, max_points=max(n_pts, N_NEW) + 64,
                    n_slots=5, max_level=MAX_LEVEL)
    prm = make_stereo_params(W_, H_, WIN, MAX_LEVEL, THRES_ERR, THRES_BIDIR, THRES_POSEBA, stream.K, stream.K,
                             stream.T_lr)
    pipe = StereoFramePipeline(ctx, prm, strict_border=bool(args.strict_border))
    ctx.set_pyramid_window_hint(WIN)  # build only the levels PyrLK with this window uses
    eff_levels = ctx.pyramid_levels(W_, H_, WIN, MAX_LEVEL) + 1
    eff_levels_bwd = ctx.pyramid_levels(W_, H_, WIN, MAX_LEVEL - 1) + 1

    # slot roles, rotated in place of copying pyramids: previous left, current pair, prefetched pair
    slot = {"P": 0, "CL": 1, "CR": 2, "NL": 3, "NR": 4}

    fe = None
    if args.detect:
        fe = V.FeatureExtractor(ctx)
        fe.initParams(W_, H_, N_U, N_V, THRES_FAST=15)

    def enqueue(s):
        a, b = frame_id(s), frame_id(s + 1)
        t = d_ts[(a, b)]
        pipe.enqueue_device(t["pts_l0"].data_ptr(), t["pts_r0"].data_ptr(), t["Xp"].data_ptr(), n_pts,
                            track_sets[(a, b)]["dT_prior"], t["pts_new"].data_ptr(), N_NEW,
                            slots=(slot["P"], slot["CL"], slot["CR"]))
        if fe is not None:  # extractORBwithBinning_fast of the current left image, off the main chain
            fe.enqueueExtract(slot["CL"])
        # the NEXT stereo pair does not depend on this frame's result: its pyramids are enqueued
        # behind the frame and build while the host waits for / consumes this frame's result
        nb = frame_id(s + 2)
        ctx.set_stereo_pair_device(slot["NL"], d_L[nb].data_ptr(), slot["NR"], d_R[nb].data_ptr(), W_, H_, W_)

    def run_frames(first, count, keep=None, on_result=None):
        """`count` frames, one at a time: the result of frame s (pose, survivors — what a sequential VO
        needs before it can go on) is received before frame s+1 is enqueued, and s+1 is enqueued at once."""
        enqueue(first)
        for s in range(first, first + count):
            r = pipe.result(copy=keep is not None and len(keep) < args.cpu_frames)
            if fe is not None:
                fe.resultExtract()
            slot["P"], slot["CL"], slot["CR"], slot["NL"], slot["NR"] = (slot["CL"], slot["NL"], slot["NR"], slot["P"],
                                                                           slot["CR"])
            if s + 1 < first + count:
                enqueue(s + 1)
            if keep is not None and len(keep) < args.cpu_frames:
                keep.append(r)
            if on_result is not None:
                on_result(r)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        ctx.synchronize()

    ctx.set_image_device(0, d_L[frame_id(0)].data_ptr(), W_, H_, W_)
    ctx.set_stereo_pair_device(1, d_L[frame_id(1)].data_ptr(), 2, d_R[frame_id(1)].data_ptr(), W_, H_, W_)
    ctx.synchronize()
    # one-time initialisation, not a step: the first frames allocate the context's frame state lazily and load the
    # code objects of every kernel; PRIME frames are pushed through before the W warmup steps the contract asks for
    PRIME = 3
    run_frames(0, PRIME)
    if args.warmup:
        run_frames(PRIME, args.warmup)
    K = args.steps
    ctx.profile_enable(K * 4 + 64)
    ctx.profile_set_classes(1 << 1)  # event-bracket only the dominant kernel (frame_track_kernel)
    ctx.profile_reset()
    acc = {"bytes": 0}
    results = []

    def account(r):
        cts = r["counts"]  # features finished by the replay kernel do their step [5] there
        acc["bytes"] += frame_kernel_bytes(WIN, n_pts, cts.n_l0l1, cts.n_refine - cts.n_replayed, N_NEW, eff_levels,
                                           eff_levels_bwd, bool(args.strict_border))

    # A generational GC pass of the interpreter (tens of ms with torch loaded) inside the timed loop
    # would be charged to a few frames; the loop allocates nothing that needs it.
    gc.collect()
    gc.freeze()
    gc.disable()
    barrier()
    t0 = time.perf_counter()
    run_frames(PRIME + args.warmup, K, results, account)
    barrier()
    dt = time.perf_counter() - t0
    gc.enable()

    per_rank = gather_ranks(K, dt, stream_seed(rank), world, dev)
    tot_frames, max_dt = sum(p[0] for p in per_rank), max(p[1] for p in per_rank)

    out = None
    if rank == 0:
        names = {0: "pyramid", 1: "frame_track", 2: "ic_replay", 3: "gn_pose", 4: "hamming", 5: "aux"}
        per_kernel = {}
        for cls, nm in names.items():
            n_l, ms = ctx.profile_get(cls)
            if n_l:
                per_kernel[nm] = {"launches": n_l, "total_ms": round(ms, 3), "avg_us": round(1e3 * ms / n_l, 2)}
        klt_n, klt_ms = ctx.profile_get(1)
        achieved = (acc["bytes"] / max(klt_n, 1)) / (klt_ms / max(klt_n, 1) * 1e-3) / 1e9 if klt_n else 0.0
        traffic = None
        pmc_path = os.path.join(ROOT, "profiles", "r01_frame_pmc.json")
        if os.path.exists(pmc_path):
            try:
                traffic = json.load(open(pmc_path)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "stereo VO frames/sec @1241x376, 1500 feats",
            "value": round(tot_frames / max_dt, 2),
            "unit": "frames/s",
            "n_gpus": world,
            "steps": K,
            "warmup": args.warmup,
            "ms_per_step": round(1e3 * max_dt / K, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8/i32 (KLT) + f32 (IC, GN)",
            "data": "synthetic",
            "config": {
                "workload": "BASELINE configs[1]: synthetic KITTI-shaped stereo stream 1241x376, 1500 tracked "
                            "features/frame (60x25 buckets) + 150 new-point candidates, win 21, max_level 6 "
                            "(5 effective levels), thresholds of config/stereo/kitti_00_stereo.yaml; one "
                            "independent stream per GPU; result read back every frame",
                "strict_border": int(args.strict_border),
                "with_detection": bool(args.detect),
                "distinct_frames": F,
            },
            "roofline": {
                "bound": "hbm",
                "kernel": "frame_track_kernel<21>",
                "achieved": round(achieved, 2),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5),
                "traffic": traffic,
                "alg_bytes_per_launch": round(acc["bytes"] / max(klt_n, 1)),
                "avg_launch_us": round(1e3 * klt_ms / max(klt_n, 1), 2),
            },
            "kernels": per_kernel,
            "per_rank_fps": [round(p[0] / p[1], 2) for p in per_rank],
        }

        if world == 1 and not args.no_cpu_baseline:
            from oracle import oracle as O
            prm_o = O.make_stereo_params(W_, H_, WIN, MAX_LEVEL, THRES_ERR, THRES_BIDIR, THRES_POSEBA, stream.K,
                                         stream.K, stream.T_lr)
            cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
            cores = min(cores, 16)  # the box's CPU share for one GPU; 1500 points do not feed more threads
            border = O.IC_REFERENCE if args.strict_border else O.IC_MASKED
            worst, stage_equal = 0.0, True
            tcpu = 0.0
            nf = min(args.cpu_frames, len(results))
            for i in range(nf):
                s = PRIME + args.warmup + i
                a, b = frame_id(s), frame_id(s + 1)
                ts = track_sets[(a, b)]
                t1 = time.perf_counter()
                o = O.stereo_frame(prm_o, imgs[a][0], imgs[b][0], imgs[b][1], ts["pts_l0"], ts["pts_r0"],
                                   ts["Xp"], ts["dT_prior"], ts["pts_new"], O.SUM_SEQ, 0, border, cores)
                tcpu += time.perf_counter() - t1
                g = results[i]
                e = float(np.linalg.norm(g["dT"].astype(np.float64) - o["dT"]) / np.linalg.norm(o["dT"]))
                worst = max(worst, e)
                stage_equal = stage_equal and bool(np.array_equal(g["stage"], o["stage"]))
            out["cpu_baseline"] = {
                "value": round(nf / tcpu, 3) if tcpu > 0 else None,
                "unit": "frames/s",
                "cores": cores,
                "kind": "port",
                "sample": f"{nf} frames of the same stream on the CPU restatement (oracle/, reference summation "
                          f"order); PyrLK over {cores} OpenMP threads, IC and GN single-threaded as in the reference",
            }
            out["parity"] = {"pose_rel_frobenius_max": worst, "survivor_sets_bit_exact": stage_equal,
                             "frames_checked": nf}
        print(json.dumps(out), flush=True)
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
