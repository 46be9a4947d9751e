"""FRAME_STAMP builds: timeline of frame_track_kernel on bench.py's closed-loop workload (1500 features + one
candidate per bucket): when workgroups start and end, what the stragglers are doing, how many are resident over time.

    VO_EXTRA_FLAGS=-DFRAME_STAMP python visual_odometry_ros_amd/build.py --force
    python tools/tools_k1timeline.py [strict_border]
"""
import sys, types, ctypes as C, numpy as np
sys.path.insert(0, '.')
import torch
import visual_odometry_ros_amd as V
import bench as B

strict = int(sys.argv[1]) if len(sys.argv) > 1 else 1
args = types.SimpleNamespace(frames=12, host_images_leg=False, strict_border=strict, cpu_frames=0)
sb = B.StereoBench(B.CONFIGS[1], args, 0, 0, torch, V)
first = sb.prime("closed")
n = sb.n_pts
nb = sb.bins.n_bins_u * sb.bins.n_bins_v
us = lambda a: a / 100.0
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
if skip:
    sb.run(first, skip, "closed")
    first += skip
for k in range(first, first + 6):
    sb.run(k, 1, "closed")
    d = np.zeros((n + nb, 8), np.int32)
    sb.ctx.lib.vo_debug_frame_stamps(sb.ctx.handle, d.ctypes.data_as(C.POINTER(C.c_int)), n + nb)
    ran = d[:, 0] > 0
    t0 = d[ran, 0].min()
    end = np.maximum.reduce([d[:, 0], d[:, 1], d[:, 2], d[:, 3]])
    start = d[:, 0]
    feat = np.arange(n + nb) < n
    work = ran & (end > start)
    print(f"frame {k}: workgroups that did work: {int(work[:n].sum())} features + {int(work[n:].sum())} candidates; kernel span {us(end[ran].max() - t0):.1f} us")
    for name, sel in (("features", work & feat), ("candidates", work & ~feat)):
        s, e = us(start[sel] - t0), us(end[sel] - t0)
        print(f"   {name:10s} start p50 {np.median(s):6.1f} p90 {np.percentile(s, 90):6.1f} max {s.max():6.1f} | end p50 {np.median(e):6.1f} p90 {np.percentile(e, 90):6.1f} p99 {np.percentile(e, 99):6.1f} max {e.max():6.1f} | life mean {(e - s).mean():6.1f} max {(e - s).max():6.1f} us")
    # resident workgroups over time
    ts = np.arange(0, us(end[ran].max() - t0) + 10, 10.0)
    res = [(int(((us(start - t0) <= t) & (us(end - t0) > t) & work).sum())) for t in ts]
    print("   resident at t (10 us steps):", res)
    order = np.argsort(-end)[:12]
    for i in order:
        r = d[i]
        print(f"   straggler {'F' if i < n else 'C'}{i if i < n else i - n:5d}: start {us(r[0]-t0):6.1f} klt0 {us(r[1]-r[0]) if r[1] else -1:6.1f} ic {us(r[2]-r[1]) if r[2] else -1:6.1f} klt1 {us(r[3]-max(r[2], r[1])) if r[3] else -1:6.1f} end {us(end[i]-t0):6.1f} | iters klt0 {r[4]} ic {r[6]} klt1 {r[5]}")
    if hasattr(sb.ctx.lib, "vo_debug_gn_stamps"):
        st = (C.c_longlong * 8)()
        try:
            sb.ctx.lib.vo_debug_gn_stamps(sb.ctx.handle, st)
            g = [us((st[i] & 0x7fffffff) - t0) for i in range(5)]
            print(f"   GN kernel (same clock, us after the frame kernel's first stamp): start {g[0]:.1f} prologue end {g[1]:.1f} loads end {g[2]:.1f} iterations end {g[3]:.1f} epilogue end {g[4]:.1f}")
        except Exception as e:
            print("   (no GN stamps)", e)
    f = d[:n][work[:n]]
    full = f[:, 3] > 0
    print(f"   feature phase means: klt0 {us(f[:,1]-f[:,0]).mean():.1f} us ({f[:,4].mean():.1f} it), ic {us(f[f[:,2]>0,2]-f[f[:,2]>0,1]).mean():.1f} us ({f[f[:,2]>0,6].mean():.1f} it), klt1 {us(f[full,3]-f[full,2]).mean():.1f} us ({f[full,5].mean():.1f} it)")
print("done")
