#!/usr/bin/env python3
"""GN stereo solve: kernel time vs number of points (HIP events) -> per-iteration fixed cost and per-point slope."""
import sys, numpy as np
sys.path.insert(0, '.')
import torch  # noqa: F401
import visual_odometry_ros_amd as V
from visual_odometry_ros_amd import synthetic as S
ctx = V.Context(max_width=1241, max_height=376, max_points=8192, n_slots=2, max_level=6)
me = V.MotionEstimator(ctx, is_stereo_mode=True) if True else None
rng = np.random.default_rng(1)
K = np.array([718.856, 718.856, 607.19, 185.2], np.float32)
T_lr = np.eye(4, dtype=np.float32); T_lr[0, 3] = 0.54
def make(n):
    X = np.stack([rng.uniform(-10, 10, n), rng.uniform(-2, 2, n), rng.uniform(5, 40, n)], 1).astype(np.float32)
    T = np.eye(4); T[:3, 3] = [0.02, -0.01, -0.8]; c, s_ = np.cos(0.01), np.sin(0.01); T[:3, :3] = [[c, 0, s_], [0, 1, 0], [-s_, 0, c]]
    Ti = np.linalg.inv(T); Xc = X @ Ti[:3, :3].T + Ti[:3, 3]
    pl = np.stack([K[0] * Xc[:, 0] / Xc[:, 2] + K[2], K[1] * Xc[:, 1] / Xc[:, 2] + K[3]], 1)
    Tr = np.linalg.inv(T_lr.astype(np.float64)); Xr = Xc @ Tr[:3, :3].T + Tr[:3, 3]
    pr = np.stack([K[0] * Xr[:, 0] / Xr[:, 2] + K[2], K[1] * Xr[:, 1] / Xr[:, 2] + K[3]], 1)
    pl += rng.normal(0, 0.3, pl.shape); pr += rng.normal(0, 0.3, pr.shape)
    return X, pl.astype(np.float32), pr.astype(np.float32)
ctx.profile_enable(4096)
for n in (64, 512, 1024, 1200, 1536, 2048, 4096):
    X, pl, pr = make(n)
    for _ in range(3): r = me.poseOnlyBundleAdjustment_Stereo(X, pl, pr, K, K, T_lr, 3.0, np.eye(4, dtype=np.float32))
    ctx.profile_reset()
    for _ in range(10): r = me.poseOnlyBundleAdjustment_Stereo(X, pl, pr, K, K, T_lr, 3.0, np.eye(4, dtype=np.float32))
    k, ms = ctx.profile_get(3)
    it = r[3].iterations
    print(f"n={n:5d}: {1e3*ms/k:7.1f} us, {it} iterations -> {1e3*ms/k/it:6.2f} us/iteration")
