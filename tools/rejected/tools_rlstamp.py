"""-DRL_STAMP builds: timeline of the run-local replay (ic_run_replay) on bench-like strict-border frames."""
import sys, ctypes as C, numpy as np
sys.path.insert(0, '.')
import torch  # noqa: F401
import visual_odometry_ros_amd as V
from visual_odometry_ros_amd import synthetic as S
from visual_odometry_ros_amd.api import StereoFramePipeline, make_stereo_params
st = S.StereoStream(); poses = st.poses(6)
ctx = V.Context(max_width=1241, max_height=376, max_points=8192, n_slots=5, max_level=6)
prm = make_stereo_params(st.width, st.height, 21, 6, 80.0, 0.5, 3.0, st.K, st.K, st.T_lr)
pipe = StereoFramePipeline(ctx, prm, strict_border=True)
ctx.set_pyramid_window_hint(21)
NW = 8 + 16 * 64 + 8 * 1024
for k in range(1, 5):
    Lp, Rp, _ = st.render_pair(poses[k - 1]); L, R, _ = st.render_pair(poses[k]); ts = st.track_set(k - 1, poses[k - 1], poses[k])
    ctx.set_image(0, Lp); ctx.set_image(1, L); ctx.set_image(2, R)
    dbg = np.zeros(NW, np.int32)
    pipe.enqueue(ts["pts_l0"], ts["pts_r0"], ts["Xp"], ts["dT_prior"], ts["pts_new"]); g = pipe.result()
    ctx.lib.vo_debug_rl(ctx.handle, dbg.ctypes.data_as(C.POINTER(C.c_int)), NW)   # discard the first (cold) frame
    pipe.enqueue(ts["pts_l0"], ts["pts_r0"], ts["Xp"], ts["dT_prior"], ts["pts_new"]); g = pipe.result()
    ctx.lib.vo_debug_rl(ctx.handle, dbg.ctypes.data_as(C.POINTER(C.c_int)), NW)
    nr, nm = int(dbg[0]), int(dbg[1])
    runs = dbg[8:8 + 16 * 64].reshape(64, 16)[:min(nr, 64)]
    mem = dbg[8 + 16 * 64:].reshape(1024, 8)[:min(nm, 1024)]
    print(f"frame {k}: replayed {g['counts'].n_replayed}, runs {nr}, member records {nm}")
    if nr == 0:
        continue
    t0 = int(runs[:, 6].min())
    big = runs[np.argsort(-runs[:, 2])][:4]
    for r in big:
        print(f"   run {r[0]:3d}: slots {r[1]:3d} touched {r[2]:3d} converged {r[3]} mask_changed {r[4]} bail {r[5]} passes {r[10]}  "
              f"{(r[6]-t0)/100:.1f} -> {(r[7]-t0)/100:.1f} us  head {r[8]} first {r[9]}")
    small = runs[runs[:, 2] <= 3]
    if len(small):
        print(f"   {len(small)} small runs: duration mean {np.mean(small[:,7]-small[:,6])/100:.1f} us, last end {(small[:,7].max()-t0)/100:.1f} us")
    # members of the biggest run
    r = big[0]
    lo, hi = r[9], r[9] + 400
    mm = mem[(mem[:, 0] >= r[9])]
    mm = mm[np.argsort(mm[:, 2])]
    sel = [m for m in mm if (m[1] >> 16) & 0xff == 1][:70]
    print("   members (pt slot wave | begin prepared resolved done | attempts iters mch):")
    for m in sel:
        print(f"     {m[0]:5d} s{m[1] & 0xffff:3d} w{(m[1] >> 24) & 0xff} | {(m[2]-t0)/100:7.1f} {(m[3]-t0)/100:7.1f} {(m[4]-t0)/100:7.1f} {(m[5]-t0)/100:7.1f} | {m[6] & 0xff:3d} {m[6] >> 8:3d} {m[7]}")
