"""Throughput of the mono frame (vo_mono_frame_enqueue) at BASELINE configs[2]'s shape — 752x480,
1000 features (40x25 buckets), win 15, 5 levels, result read back every frame — next to the CPU
oracle on the same frames. Measurement tool (not the driver's bench line; bench.py stays on configs[1]).
usage: python tests/measure/monobench.py [--steps 400] [--strict-border 1] [--cpu-frames 6]"""
import argparse
import gc
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
MONO_K = (458.654, 457.296, 367.215, 248.375)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--frames", type=int, default=12)
    ap.add_argument("--strict-border", type=int, default=1)
    ap.add_argument("--cpu-frames", type=int, default=6)
    args = ap.parse_args()
    import torch
    import visual_odometry_ros_amd as V
    from visual_odometry_ros_amd import synthetic as S
    from visual_odometry_ros_amd.api import MonoFramePipeline, make_mono_params
    V.load()
    dev = torch.device("cuda", 0)
    W_, H_, WIN, LVL = 752, 480, 15, 5
    stream = S.StereoStream(width=W_, height=H_, K=MONO_K, n_u=40, n_v=25, n_new=50, seed=3, speed=0.25)
    F = max(args.frames, 3)
    poses = stream.poses(F)
    imgs = [stream.render_pair(p)[0] for p in poses]
    order = list(range(F)) + list(range(F - 2, 0, -1))
    fid = lambda s: order[s % len(order)]
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
        sets[(a, b)] = dict(pts0=ts["pts_l0"], Xw=ts["Xp"].astype(np.float32), flags=flags,
                            Tcw_prev=np.eye(4, dtype=np.float32), Tcw_prior=np.linalg.inv(dT.astype(np.float64)).astype(np.float32),
                            dT=dT)
    d_I = [torch.from_numpy(np.ascontiguousarray(I)).to(dev) for I in imgs]
    d_s = {k: {q: torch.from_numpy(np.ascontiguousarray(v[q])).to(dev) for q in ("pts0", "Xw", "flags")} for k, v in sets.items()}
    torch.cuda.synchronize()
    n_pts = 1000
    ctx = V.Context(device=0, max_width=W_, max_height=H_, max_points=n_pts + 64, n_slots=3, max_level=LVL)
    args_p = (W_, H_, WIN, LVL, 20.0, 1.0, 5, 1.0, MONO_K)
    pipe = MonoFramePipeline(ctx, make_mono_params(*args_p), strict_border=args.strict_border)
    ctx.set_pyramid_window_hint(WIN)
    slot = {"P": 0, "C": 1, "N": 2}

    def enqueue(s):
        k = (fid(s), fid(s + 1))
        t, h = d_s[k], sets[k]
        pipe.enqueue_device(t["pts0"].data_ptr(), t["Xw"].data_ptr(), t["flags"].data_ptr(), n_pts, h["Tcw_prev"],
                            h["Tcw_prior"], h["dT"], slots=(slot["P"], slot["C"]))
        ctx.set_image_device(slot["N"], d_I[fid(s + 2)].data_ptr(), W_, H_, W_)

    def run(first, count, keep=None):
        enqueue(first)
        for s in range(first, first + count):
            r = pipe.result()
            slot["P"], slot["C"], slot["N"] = slot["C"], slot["N"], slot["P"]
            if s + 1 < first + count:
                enqueue(s + 1)
            if keep is not None and len(keep) < max(args.cpu_frames, 1):
                keep.append((s, r))

    ctx.set_image_device(0, d_I[fid(0)].data_ptr(), W_, H_, W_)
    ctx.set_image_device(1, d_I[fid(1)].data_ptr(), W_, H_, W_)
    ctx.synchronize()
    run(0, args.warmup)
    ctx.profile_enable(args.steps * 8 + 64)
    ctx.profile_reset()
    kept = []
    gc.collect(); gc.freeze(); gc.disable()
    ctx.synchronize()
    t0 = time.perf_counter()
    run(args.warmup, args.steps, kept)
    ctx.synchronize()
    dt = time.perf_counter() - t0
    gc.enable()
    names = {0: "pyramid", 1: "mono_track", 2: "ic_replay", 3: "gn_pose", 5: "aux"}
    per = {}
    for cls, nm in names.items():
        nl, ms = ctx.profile_get(cls)
        if nl:
            per[nm] = {"launches": nl, "avg_us": round(1e3 * ms / nl, 2)}
    out = {"workload": "mono 752x480, 1000 features, win 15, 5 levels (BASELINE configs[2] shape)",
           "strict_border": args.strict_border, "frames_per_s": round(args.steps / dt, 1),
           "ms_per_frame": round(1e3 * dt / args.steps, 4), "kernels": per,
           "counts": {f: getattr(kept[0][1]["counts"], f) for f in ("n_klt", "n_refine", "n_ba", "n_motion", "n_final", "n_replayed")}}
    if args.cpu_frames:
        kept = kept[:args.cpu_frames]
        from oracle import oracle as O
        prm_o = O.make_mono_params(*args_p)
        cores = min(len(os.sched_getaffinity(0)), 16)
        border = O.IC_REFERENCE if args.strict_border else O.IC_MASKED
        t0 = time.perf_counter()
        for s, r in kept:
            k = (fid(s), fid(s + 1))
            h = sets[k]
            o = O.mono_frame(prm_o, imgs[k[0]], imgs[k[1]], h["pts0"], h["Xw"], h["flags"], h["Tcw_prev"], h["Tcw_prior"],
                             h["dT"], O.SUM_SEQ, 0, border, cores)
            assert np.array_equal(o["stage"], r["stage"]), "survivor sets differ from the CPU restatement"
        cdt = time.perf_counter() - t0
        out["cpu_oracle"] = {"frames_per_s": round(len(kept) / cdt, 2), "cores": cores}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
