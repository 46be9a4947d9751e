"""One-off robustness sweep of the strict-border replay: many seeds / orders / image sizes, GPU vs oracle."""
import sys, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import torch  # noqa: F401
import visual_odometry_ros_amd as V
from oracle import oracle as O
from util import image_pair, grid_points, move_points
O.build()
ctx = V.Context(device=0, max_width=1241, max_height=480, max_points=8192, n_slots=4, max_level=6)
ft = V.FeatureTracker(ctx)
bad = 0
for seed in range(40):
    rng = np.random.default_rng(1000 + seed)
    h, w = int(rng.integers(120, 300)), int(rng.integers(160, 420))
    motion = dict(dx=float(rng.uniform(-3, 3)), dy=float(rng.uniform(-3, 3)), scale=float(rng.uniform(0.95, 1.1)), angle=float(rng.uniform(-0.004, 0.004)))
    img0, img1 = image_pair(h, w, seed=seed, **motion)
    pts0 = grid_points(h, w, step=int(rng.integers(7, 13)), margin=int(rng.integers(1, 5)))
    gt = move_points(pts0.astype(np.float64), img0.shape, **motion).astype(np.float32)
    prior = (gt + rng.normal(0, 0.4, gt.shape)).astype(np.float32)
    scale = np.full(len(pts0), motion["scale"], np.float32) * (1 + rng.normal(0, 0.01, len(pts0))).astype(np.float32)
    perm = rng.permutation(len(pts0)) if seed % 3 == 0 else np.arange(len(pts0))
    pts0, prior, scale = pts0[perm], prior[perm], scale[perm]
    m_in = rng.random(len(pts0)) > 0.05
    ctx.set_image(0, img0); ctx.set_image(1, img1)
    p, m = ft.trackWithScale(0, 1, pts0, scale, prior, m_in, strict_border=True)
    rc, pr, mr, tb = O.track_with_scale(img0, img1, pts0, scale, prior, m_in, O.IC_REFERENCE, O.SUM_TREE)
    ok = rc == 0 and np.array_equal(m, mr) and np.array_equal(p.view(np.uint32), pr.view(np.uint32))
    bad += not ok
    print(f"seed {seed:2d}: {h}x{w} n={len(pts0):4d} touched={int(tb.sum()):4d} {'ok' if ok else 'MISMATCH'}")
print("mismatches:", bad)
