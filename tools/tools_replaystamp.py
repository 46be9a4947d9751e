"""IC_STAMP builds: per-feature timeline of the strict-border replay (frame_replay_kernel) on bench-like frames:
for every replayed feature the final look (attempt start, writers found), the run (start, end, iterations) and the
publication — i.e. what a link of a dependency chain is made of."""
import sys, ctypes as C, numpy as np
sys.path.insert(0, '.')
import torch  # noqa: F401
import visual_odometry_ros_amd as V
from visual_odometry_ros_amd import synthetic as S
from visual_odometry_ros_amd.api import StereoFramePipeline, make_stereo_params
st = S.StereoStream(); poses = st.poses(8)
ctx = V.Context(max_width=1241, max_height=376, max_points=8192, n_slots=5, max_level=6)
prm = make_stereo_params(st.width, st.height, 21, 6, 80.0, 0.5, 3.0, st.K, st.K, st.T_lr)
pipe = StereoFramePipeline(ctx, prm, strict_border=int(next((a for a in sys.argv[1:] if a.isdigit()), 1)))
ctx.set_pyramid_window_hint(21)
for k in range(1, 7):
    Lp, Rp, _ = st.render_pair(poses[k - 1]); L, R, _ = st.render_pair(poses[k]); ts = st.track_set(k - 1, poses[k - 1], poses[k])
    ctx.set_image(0, Lp); ctx.set_image(1, L); ctx.set_image(2, R)
    dbg = np.zeros(80 + 2048, np.int32)
    for rep in range(2):
        pipe.enqueue(ts["pts_l0"], ts["pts_r0"], ts["Xp"], ts["dT_prior"], ts["pts_new"]); g = pipe.result()
        ctx.lib.vo_debug_ic_jac(ctx.handle, dbg.ctypes.data_as(C.POINTER(C.c_int)))  # (the second, warm frame is kept)
    d = dbg[16:]
    t0 = int(d[31])
    print(f"frame {k}: replayed {g['counts'].n_replayed:4d}  runs {d[32]:4d} publishes {d[33]:4d} predicted looks {d[34]:4d}  quiescent at {(int(d[0]) - t0) / 100:6.1f} us, tails done at {(int(d[2]) - t0) / 100:6.1f} us")
    rows = dbg[80:].reshape(256, 8); rows = rows[rows[:, 1] > 0]
    if not len(rows):
        continue
    us = lambda c: (rows[:, c] - t0) / 100.0
    att, fw, rs_, re_, pub, it, natt, pt = us(4), us(5), us(0), us(1), us(6), rows[:, 2], us(7), rows[:, 3]
    o = np.argsort(re_)
    print("   last finishers: pt | final look starts, writers found, run starts, run ends, published | iterations, loads landed")
    for i in (o if "--all" in sys.argv else o[-16:]):
        print(f"     {pt[i]:5d} | {att[i]:7.1f} {fw[i]:7.1f} {rs_[i]:7.1f} {re_[i]:7.1f} {pub[i]:7.1f} | {it[i]:3d} {natt[i]:7.1f}")
    print(f"   means: look->writers {np.mean(fw - att):.1f} us, writers->run start {np.mean(rs_ - fw):.1f} us, run {np.mean(re_ - rs_):.1f} us "
          f"({it.mean():.1f} iterations), run end->published {np.mean(pub - re_):.1f} us; look start->loads landed {np.mean(natt - att):.1f} us")
