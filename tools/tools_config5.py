"""BASELINE configs[4] shape (3840x2160, 8000 features): duration of the fused frame kernel (HIP events)."""
import sys, numpy as np
sys.path.insert(0, '.')
import torch  # noqa: F401
import visual_odometry_ros_amd as V
from visual_odometry_ros_amd import synthetic as S
from visual_odometry_ros_amd.api import StereoFramePipeline, make_stereo_params
K = tuple(v * 3.0 for v in S.KITTI_K[:2]) + (1920.0, 1080.0)
st = S.StereoStream(width=3840, height=2160, K=K, n_u=100, n_v=80, n_new=200, seed=3, margin=31.0)
poses = st.poses(2)
L0, R0, _ = st.render_pair(poses[0]); L1, R1, _ = st.render_pair(poses[1]); ts = st.track_set(0, poses[0], poses[1])
ctx = V.Context(max_width=3840, max_height=2160, max_points=8448, n_slots=3, max_level=4)
for strict in (0, 1):
    prm = make_stereo_params(3840, 2160, 21, 4, 80.0, 0.5, 3.0, st.K, st.K, st.T_lr)
    pipe = StereoFramePipeline(ctx, prm, strict_border=bool(strict))
    ctx.set_image(0, L0); ctx.set_image(1, L1); ctx.set_image(2, R1)
    ctx.profile_enable(256)
    for _ in range(3):
        pipe.enqueue(ts["pts_l0"], ts["pts_r0"], ts["Xp"], ts["dT_prior"], ts["pts_new"]); pipe.result()
    ctx.profile_reset()
    for _ in range(10):
        pipe.enqueue(ts["pts_l0"], ts["pts_r0"], ts["Xp"], ts["dT_prior"], ts["pts_new"]); g = pipe.result()
    for cls, nm in ((1, 'frame_track'), (2, 'replay'), (3, 'gn')):
        k, ms = ctx.profile_get(cls)
        if k: print(f"strict={strict} {nm:12s}: {1e3*ms/k:8.1f} us per launch")
    print("   counts:", g["counts"].n_l0l1, g["counts"].n_refine, g["counts"].n_l1r1, g["counts"].n_inlier, "replayed", g["counts"].n_replayed)
