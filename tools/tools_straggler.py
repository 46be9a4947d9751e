"""Upper bound for "move the stragglers of the frame kernel to a launch of their own": the frame kernel's duration on
bench.py's round-2 workload (1500 features + one candidate per bucket) next to the duration of a launch that holds ONLY
the K slowest features of that frame (same inputs, so the same iterations), i.e. what they cost when nothing shares
their SIMDs. FRAME_STAMP build:

    VO_EXTRA_FLAGS=-DFRAME_STAMP python -m visual_odometry_ros_amd.build --force
    python tools/tools_straggler.py [K=16]
"""
import ctypes as C
import sys
import types

import numpy as np

sys.path.insert(0, ".")
import torch  # noqa: E402
import visual_odometry_ros_amd as V  # noqa: E402
import bench as B  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 16
args = types.SimpleNamespace(frames=12, host_images_leg=False, strict_border=1, cpu_frames=0)
sb = B.StereoBench(B.CONFIGS[1], args, 0, 0, torch, V)
first = sb.prime("closed")
n, nb = sb.n_pts, sb.bins.n_bins_u * sb.bins.n_bins_v
ctx = sb.ctx
ctx.profile_enable(4096)
ctx.profile_set_classes(1 << 1)
rows = []
for k in range(first, first + 8):
    ctx.profile_reset()
    sb.run(k, 1, "closed")
    ctx.synchronize()
    _, full_ms = ctx.profile_get(1)
    d = np.zeros((n + nb, 8), np.int32)
    ctx.lib.vo_debug_frame_stamps(ctx.handle, d.ctypes.data_as(C.POINTER(C.c_int)), n + nb)
    end = np.maximum.reduce([d[:n, 0], d[:n, 1], d[:n, 2], d[:n, 3]])
    life = (end - d[:n, 0]) / 100.0
    slow = np.sort(np.argsort(-end)[:K])
    # the same frame with only those K features (open operator, no candidates): slots as run() left them before rotation
    a, b = sb.frame_id(k), sb.frame_id(k + 1)
    ts = sb.track_sets[(a, b)]
    sub = {q: torch.from_numpy(np.ascontiguousarray(ts[q][slow])).to(sb.dev) for q in ("pts_l0", "pts_r0", "Xp", "flags")}
    s = sb.slot  # after run(): P holds frame b's left image, i.e. we need the previous rotation: rebuild the three slots
    ctx.synchronize()
    ctx.set_image_device(0, sb.d_L[a].data_ptr(), sb.W, sb.H, sb.W)
    ctx.set_stereo_pair_device(1, sb.d_L[b].data_ptr(), 2, sb.d_R[b].data_ptr(), sb.W, sb.H, sb.W)
    ctx.synchronize()
    ctx.profile_reset()
    sb.pipe.enqueue_device(sub["pts_l0"].data_ptr(), sub["pts_r0"].data_ptr(), sub["Xp"].data_ptr(), K, ts["dT_prior"], 0, 0,
                           slots=(0, 1, 2), d_lm_flags=sub["flags"].data_ptr())
    r = sb.pipe.result()
    ctx.synchronize()
    _, alone_ms = ctx.profile_get(1)
    dd = np.zeros((K, 8), np.int32)
    ctx.lib.vo_debug_frame_stamps(ctx.handle, dd.ctypes.data_as(C.POINTER(C.c_int)), K)
    same = bool(np.array_equal(dd[:, 4:7], d[slow, 4:7]))
    rows.append((k, 1e3 * full_ms, 1e3 * alone_ms, float(life[slow].max()), int(d[slow, 4].max()), same))
    print(f"frame {k}: full launch {1e3 * full_ms:6.1f} us | its {K} slowest features alone {1e3 * alone_ms:6.1f} us | their longest "
          f"life inside the full launch {life[slow].max():6.1f} us | max KLT iterations {d[slow, 4].max()} | same iteration counts: {same}", flush=True)
    # restore the slots the bench loop expects for the next step
    sb.prime("closed")
f, al = np.mean([r[1] for r in rows]), np.mean([r[2] for r in rows])
print(f"mean: full launch {f:.1f} us, slowest {K} alone {al:.1f} us -> a perfect migration could save at most {f - al:.1f} us per frame "
      f"({100 * (f - al) / f:.0f} %) before its own costs (list hand-over, the restart's level set-up ~14 us, a second resident kernel)")
