"""Per-frame times of bench.py's closed-loop run next to what each frame had to do (touched features, GN iterations)."""
import sys, time, types, numpy as np
sys.path.insert(0, '.')
import torch
import visual_odometry_ros_amd as V
import bench as B

args = types.SimpleNamespace(frames=12, host_images_leg=False, strict_border=int(sys.argv[1]) if len(sys.argv) > 1 else 1, cpu_frames=0)
sb = B.StereoBench(B.CONFIGS[1], args, 0, 0, torch, V)
first = sb.prime("closed")
rows = []
def on_result(step, r):
    c = r["counts"]
    rows.append((time.perf_counter(), step, c.n_replayed, c.gn_iterations, c.n_ba, int((r["stage"] == 4).sum())))
sb.run(first, 20, "closed")
rows.clear()
sb.run(first + 20, 240, "closed", on_result=on_result)
t = np.array([r[0] for r in rows]); dt = np.diff(t) * 1e6
a = np.array([r[1:] for r in rows[1:]])
print("frame us: mean %.1f median %.1f" % (dt.mean(), np.median(dt)))
print("corr(frame us, n_replayed) = %.2f ; corr(frame us, gn_iterations) = %.2f" % (np.corrcoef(dt, a[:, 1])[0, 1], np.corrcoef(dt, a[:, 2])[0, 1]))
for lo, hi in ((0, 200), (200, 300), (300, 400), (400, 500), (500, 2000)):
    m = (dt >= lo) & (dt < hi)
    if m.any():
        print(f"  {lo:4d}-{hi:4d} us: {int(m.sum()):3d} frames, replayed mean {a[m,1].mean():6.1f}, GN iterations mean {a[m,2].mean():4.1f}, n_ba {a[m,3].mean():.0f}")
period = len(sb.order)
print("by position in the playback cycle (step % period): us, replayed, gn iters")
for k in range(period):
    m = (a[:, 0] % period) == k
    print(f"  {k:2d}: {dt[m].mean():6.1f} us  replayed {a[m,1].mean():6.1f}  gn {a[m,2].mean():4.1f}")
