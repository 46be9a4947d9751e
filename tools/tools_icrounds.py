"""IC_STAMP builds: per-round statistics of the parallel strict replay on a few stream frames."""
import sys, ctypes as C, numpy as np
sys.path.insert(0, '.')
import torch  # noqa: F401  (HIP runtime first)
import visual_odometry_ros_amd as V
from visual_odometry_ros_amd import synthetic as S
st = S.StereoStream(); poses = st.poses(8)
ctx = V.Context(max_width=1241, max_height=376, max_points=8192, n_slots=4, max_level=6)
ft = V.FeatureTracker(ctx)
K = st.K
for k in range(1, 7):
    L0, R0, _ = st.render_pair(poses[k - 1]); L1, R1, _ = st.render_pair(poses[k])
    ts = st.track_set(k, poses[k - 1], poses[k])
    ctx.set_image(0, L0); ctx.set_image(1, L1)
    T_cp = np.linalg.inv(ts['dT_prior'].astype(np.float64))
    Xl1 = ts['Xp'] @ T_cp[:3, :3].T + T_cp[:3, 3]
    scale = (ts['Xp'][:, 2] / Xl1[:, 2]).astype(np.float32)
    prior = np.stack([K[0] * Xl1[:, 0] / Xl1[:, 2] + K[2], K[1] * Xl1[:, 1] / Xl1[:, 2] + K[3]], 1).astype(np.float32)
    p1, m1 = ft.trackWithPrior(0, 1, ts['pts_l0'], 21, 6, 80.0, prior)
    idx = np.nonzero(m1)[0]
    dbg = np.zeros(80, np.int32)
    ctx.lib.vo_debug_ic_jac(ctx.handle, dbg.ctypes.data_as(C.POINTER(C.c_int)))  # clear
    r = ft.trackWithScale(0, 1, ts['pts_l0'][idx], scale[idx], p1[idx], None, strict_border=True)
    ctx.lib.vo_debug_ic_jac(ctx.handle, dbg.ctypes.data_as(C.POINTER(C.c_int)))
    jac, d = dbg[:16], dbg[16:]
    print(f"frame {k}: n={len(idx)} touched={jac[14]} ovf={jac[15]} publishes(ver)={jac[13]} reruns={d[32]} wg0 span us={(int(d[0]) - int(d[31])) / 100:.1f}")
