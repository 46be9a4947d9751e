#!/usr/bin/env python3
"""Times single operators (IC refine, KLT) on one bench-shaped frame with HIP events."""
import sys, time, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, '..')
import visual_odometry_ros_amd as V
from visual_odometry_ros_amd import synthetic as S
from visual_odometry_ros_amd.api import make_stereo_params, StereoFramePipeline
st = S.StereoStream()
poses = st.poses(3)
L0, R0, _ = st.render_pair(poses[0]); L1, R1, _ = st.render_pair(poses[1])
ts = st.track_set(1, poses[0], poses[1])
ctx = V.Context(max_width=1241, max_height=376, max_points=4096, n_slots=4, max_level=6)
ft = V.FeatureTracker(ctx)
ctx.set_image(0, L0); ctx.set_image(1, L1); ctx.set_image(2, R1)
n = ts['pts_l0'].shape[0]
T_cp = np.linalg.inv(ts['dT_prior'].astype(np.float64))
Xl1 = ts['Xp'] @ T_cp[:3,:3].T + T_cp[:3,3]
scale = (ts['Xp'][:,2]/Xl1[:,2]).astype(np.float32)
K = st.K
prior = np.stack([K[0]*Xl1[:,0]/Xl1[:,2]+K[2], K[1]*Xl1[:,1]/Xl1[:,2]+K[3]],1).astype(np.float32)
p1, m1 = ft.trackWithPrior(0,1,ts['pts_l0'],21,6,80.0,prior)
ctx.profile_enable(4096)
for name, fn in [('klt_l0l1', lambda: ft.trackWithPrior(0,1,ts['pts_l0'],21,6,80.0,prior)),
                 ('ic_nostrict', lambda: ft.trackWithScale(0,1,ts['pts_l0'],scale,p1,m1,strict_border=False)),
                 ('ic_strict', lambda: ft.trackWithScale(0,1,ts['pts_l0'],scale,p1,m1,strict_border=True))]:
    for _ in range(3): fn()
    ctx.profile_reset()
    for _ in range(10): r = fn()
    for cls,nm in ((1,'klt'),(2,'ic')):
        k, ms = ctx.profile_get(cls)
        if k: print(f"{name:12s} {nm}: {k} launches, avg {1e3*ms/k:8.1f} us")
print('mask sum', r[1].sum(), 'of', m1.sum())
