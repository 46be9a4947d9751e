"""FRAME_STAMP builds: per-feature phase times inside frame_track_kernel for a few stream frames."""
import sys, ctypes as C, numpy as np
sys.path.insert(0, '.')
import torch  # noqa: F401
import visual_odometry_ros_amd as V
from visual_odometry_ros_amd import synthetic as S
from visual_odometry_ros_amd.api import StereoFramePipeline, make_stereo_params
st = S.StereoStream(); poses = st.poses(6)
ctx = V.Context(max_width=1241, max_height=376, max_points=4096, n_slots=5, max_level=6)
prm = make_stereo_params(st.width, st.height, 21, 6, 80.0, 0.5, 3.0, st.K, st.K, st.T_lr)
pipe = StereoFramePipeline(ctx, prm, strict_border=False)
ctx.set_pyramid_window_hint(21)
Lp, Rp, _ = st.render_pair(poses[0]); ctx.set_image(0, Lp)
for k in range(1, 5):
    L, R, _ = st.render_pair(poses[k]); ts = st.track_set(k - 1, poses[k - 1], poses[k])
    ctx.set_image(1, L); ctx.set_image(2, R)
    for rep in range(3):
        pipe.enqueue(ts["pts_l0"], ts["pts_r0"], ts["Xp"], ts["dT_prior"], ts["pts_new"]); g = pipe.result()
    n, nn = len(ts["pts_l0"]), len(ts["pts_new"])
    d = np.zeros((n + nn, 8), np.int32)
    ctx.lib.vo_debug_frame_stamps(ctx.handle, d.ctypes.data_as(C.POINTER(C.c_int)), n + nn)
    t0 = d[:, 0].min()
    f = d[:n]; full = f[:, 3] > 0
    us = lambda a: a / 100.0
    start = us(d[:, 0] - t0)
    end = np.where(d[:, 3] > 0, d[:, 3], np.where(d[:, 2] > 0, d[:, 2], d[:, 1]))
    print(f"frame {k}: n={n}+{nn}  start spread: p50 {np.median(start):.1f} p99 {np.percentile(start,99):.1f} max {start.max():.1f} us ; kernel end {us(end.max()-t0):.1f} us")
    k1 = us(f[:, 1] - f[:, 0]); ic = us(f[full, 2] - f[full, 1]); k2 = us(f[full, 3] - f[full, 2])
    print(f"   KLT1: mean {k1.mean():.1f} p99 {np.percentile(k1,99):.1f} max {k1.max():.1f} us, iters mean {f[:,4].mean():.1f} max {f[:,4].max()} ; us/iter fit:", np.polyfit(f[:, 4], k1, 1).round(3))
    print(f"   IC  : mean {ic.mean():.1f} p99 {np.percentile(ic,99):.1f} max {ic.max():.1f} us, iters mean {f[full,6].mean():.1f} ; fit:", np.polyfit(f[full, 6], ic, 1).round(3))
    print(f"   KLT2: mean {k2.mean():.1f} p99 {np.percentile(k2,99):.1f} max {k2.max():.1f} us, iters mean {f[full,5].mean():.1f} max {f[full,5].max()} ; fit:", np.polyfit(f[full, 5], k2, 1).round(3))
    tot = us(end[:n] - d[:n, 0]); print(f"   total per feature: mean {tot.mean():.1f} p99 {np.percentile(tot,99):.1f} max {tot.max():.1f} us")
    c = d[n:]; ct = us(c[:, 3] - c[:, 0]); print(f"   candidates: mean {ct.mean():.1f} max {ct.max():.1f} us (fwd iters max {c[:,4].max()}, bwd max {c[:,5].max()})")
