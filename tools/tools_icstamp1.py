"""IC_STAMP builds: per-iteration cycle breakdown of ONE point running alone on the GPU."""
import sys, ctypes as C, numpy as np
sys.path.insert(0, '.')
import torch  # noqa: F401
import visual_odometry_ros_amd as V
from visual_odometry_ros_amd import synthetic as S
st = S.StereoStream(); poses = st.poses(3)
L0, R0, _ = st.render_pair(poses[0]); L1, R1, _ = st.render_pair(poses[1])
ts = st.track_set(1, poses[0], poses[1])
ctx = V.Context(max_width=1241, max_height=376, max_points=8192, n_slots=4, max_level=6)
ft = V.FeatureTracker(ctx)
ctx.set_image(0, L0); ctx.set_image(1, L1)
T_cp = np.linalg.inv(ts['dT_prior'].astype(np.float64))
Xl1 = ts['Xp'] @ T_cp[:3, :3].T + T_cp[:3, 3]
scale = (ts['Xp'][:, 2] / Xl1[:, 2]).astype(np.float32); K = st.K
prior = np.stack([K[0] * Xl1[:, 0] / Xl1[:, 2] + K[2], K[1] * Xl1[:, 1] / Xl1[:, 2] + K[3]], 1).astype(np.float32)
p1, m1 = ft.trackWithPrior(0, 1, ts['pts_l0'], 21, 6, 80.0, prior)
idx = np.nonzero(m1)[0]
rows = []
for i in idx[200:260]:
    sel = np.array([i])
    for rep in range(2):
        ft.trackWithScale(0, 1, ts['pts_l0'][sel], scale[sel], p1[sel], None, strict_border=True)
    out = np.zeros((1, 6), np.float32)
    ctx.lib.vo_debug_ic_rows(ctx.handle, out.ctypes.data_as(C.POINTER(C.c_float)), 6, 1)
    it = out[0, 0]
    if it >= 1:
        dur = ((out[0, 5] - out[0, 4]) % (1 << 24)) / 100
        rows.append((it, *(out[0, 1:4] / it), dur))
rows = np.array(rows)
print('alone: iters  sample  reduce+exch  solve+err  total  | WG dur us')
for r in rows[np.argsort(rows[:, 0])][::4]:
    print('  %3d  %6.0f %6.0f %6.0f %6.0f | %5.1f' % (r[0], r[1], r[2], r[3], r[1] + r[2] + r[3], r[4]))
print('mean cyc/iter: sample %.0f reduce+exch %.0f solve+err %.0f' % tuple(rows[:, 1:4].mean(0)))
A = np.vstack([rows[:, 0], np.ones(len(rows))]).T
k, b = np.linalg.lstsq(A, rows[:, 4], rcond=None)[0]
print('WG duration fit: %.2f us fixed + %.3f us/iter' % (b, k))
