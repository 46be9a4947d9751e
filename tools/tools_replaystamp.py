"""IC_STAMP builds: replay phases inside frame_replay_kernel on bench-like frames (strict border)."""
import sys, ctypes as C, numpy as np
sys.path.insert(0, '.')
import torch  # noqa: F401
import visual_odometry_ros_amd as V
from visual_odometry_ros_amd import synthetic as S
from visual_odometry_ros_amd.api import StereoFramePipeline, make_stereo_params
st = S.StereoStream(); poses = st.poses(8)
ctx = V.Context(max_width=1241, max_height=376, max_points=8192, n_slots=5, max_level=6)
prm = make_stereo_params(st.width, st.height, 21, 6, 80.0, 0.5, 3.0, st.K, st.K, st.T_lr)
pipe = StereoFramePipeline(ctx, prm, strict_border=True)
ctx.set_pyramid_window_hint(21)
for k in range(1, 7):
    Lp, Rp, _ = st.render_pair(poses[k - 1]); L, R, _ = st.render_pair(poses[k]); ts = st.track_set(k - 1, poses[k - 1], poses[k])
    ctx.set_image(0, Lp); ctx.set_image(1, L); ctx.set_image(2, R)
    for rep in range(2):
        pipe.enqueue(ts["pts_l0"], ts["pts_r0"], ts["Xp"], ts["dT_prior"], ts["pts_new"]); g = pipe.result()
    dbg = np.zeros(80 + 2048, np.int32)
    ctx.lib.vo_debug_ic_jac(ctx.handle, dbg.ctypes.data_as(C.POINTER(C.c_int)))
    d = dbg[16:]
    t0 = int(d[31])
    print(f"frame {k}: replayed {g['counts'].n_replayed:4d}  looks->reruns {d[32]:4d} publishes {d[33]:4d}  quiescent at {(int(d[0]) - t0) / 100:6.1f} us, tails done at {(int(d[2]) - t0) / 100:6.1f} us")
    rows = dbg[80:].reshape(512, 4); rows = rows[rows[:, 1] > 0]
    st_, en, it = (rows[:, 0] - t0) / 100.0, (rows[:, 1] - t0) / 100.0, rows[:, 2]
    o = np.argsort(en)[-14:]
    print("   last finishers (pt, start, end, iters):", [(int(rows[i, 3]), round(float(st_[i]), 1), round(float(en[i]), 1), int(it[i])) for i in o])
    print(f"   run time: mean {np.mean(en - st_):.1f} us, iters mean {it.mean():.1f}; runs with 30 iters: {(it >= 30).sum()}; first starts: {np.sort(st_)[:3].round(1)}")
