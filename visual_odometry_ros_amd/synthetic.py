"""Seeded synthetic workloads for the VO hot path.

No dataset is reachable offline, so every config is generated: the 2-view
point set of config 1, and a ray-cast textured-corridor stereo stream with
exact per-pixel depth for configs 2-5.  Generators are pure numpy (float64
geometry, uint8 images) so that the same bytes feed the HIP path and the oracle.
"""
import numpy as np

# config/stereo/kitti_00_stereo.yaml:11-14,48 in the reference
KITTI_K = (718.856, 718.856, 607.1928, 185.2157)
KITTI_BASELINE = 0.5371657189
KITTI_SIZE = (1241, 376)


def se3_exp(xi):
    """float64 SE(3) exponential, xi = (v, w)."""
    xi = np.asarray(xi, np.float64)
    v, w = xi[:3], xi[3:]
    th = np.linalg.norm(w)
    wx = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    if th < 1e-12:
        R = np.eye(3) + wx
        V = np.eye(3) + 0.5 * wx
    else:
        R = np.eye(3) + np.sin(th) / th * wx + (1 - np.cos(th)) / th**2 * wx @ wx
        V = np.eye(3) + (1 - np.cos(th)) / th**2 * wx + (th - np.sin(th)) / th**3 * wx @ wx
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = V @ v
    return T


def stereo_T_lr(baseline=KITTI_BASELINE):
    T = np.eye(4, dtype=np.float32)
    T[0, 3] = baseline
    return T


def two_view_points(n=500, seed=1, K=KITTI_K, noise_px=0.3, outlier_frac=0.10,
                    xi_true=(0.05, -0.02, 0.8, 0.004, -0.01, 0.002), baseline=KITTI_BASELINE):
    """BASELINE config 1: X ~ U([-10,10]x[-4,4]x[4,40]) in the previous
    camera frame, true motion xi_true (T01 = exp(xi): pose of camera 1 in frame 0),
    pixels = projection + N(0, noise^2), a fraction of uniform +-20 px outliers."""
    rng = np.random.default_rng(seed)
    X = np.stack([rng.uniform(-10, 10, n), rng.uniform(-4, 4, n), rng.uniform(4, 40, n)], 1)
    T01 = se3_exp(xi_true)
    T10 = np.linalg.inv(T01)
    X1 = X @ T10[:3, :3].T + T10[:3, 3]
    fx, fy, cx, cy = K
    pl = np.stack([fx * X1[:, 0] / X1[:, 2] + cx, fy * X1[:, 1] / X1[:, 2] + cy], 1)
    Xr = X1.copy()
    Xr[:, 0] -= baseline  # T_rl = inverse(T_lr), T_lr translates +baseline along x
    pr = np.stack([fx * Xr[:, 0] / Xr[:, 2] + cx, fy * Xr[:, 1] / Xr[:, 2] + cy], 1)
    pl += rng.normal(0, noise_px, pl.shape)
    pr += rng.normal(0, noise_px, pr.shape)
    n_out = int(round(outlier_frac * n))
    out_idx = rng.choice(n, n_out, replace=False)
    pl[out_idx] += rng.uniform(-20, 20, (n_out, 2))
    pr[out_idx] += rng.uniform(-20, 20, (n_out, 2))
    is_outlier = np.zeros(n, bool)
    is_outlier[out_idx] = True
    return dict(X=X.astype(np.float32), pts_l=pl.astype(np.float32), pts_r=pr.astype(np.float32),
                T01_true=T01, K=np.asarray(K, np.float32), T_lr=stereo_T_lr(baseline),
                is_outlier=is_outlier)


def random_descriptors(n, seed=0, flip_from=None, flip_bits=20):
    """n x 32 uint8 ORB-like descriptors; optionally noisy copies of `flip_from`."""
    rng = np.random.default_rng(seed)
    if flip_from is None:
        return rng.integers(0, 256, (n, 32), dtype=np.uint8)
    src = np.asarray(flip_from, np.uint8)
    bits = np.unpackbits(src[rng.integers(0, src.shape[0], n)], axis=1)
    for i in range(n):
        idx = rng.choice(256, flip_bits, replace=False)
        bits[i, idx] ^= 1
    return np.packbits(bits, axis=1)


# --------------------------------------------------------------------------
# Ray-cast corridor scene
# --------------------------------------------------------------------------
class CorridorScene:
    """Axis-aligned textured box the camera drives through (camera frame: x right,
    y down, z forward). Planes: ground y=+hg, ceiling y=-hc, walls x=-wl / x=+wr,
    end wall z=z_end. Texture = multi-octave value noise indexed by the two
    in-plane world coordinates, octave weights faded by pixel footprint so that
    the image is band-limited at every depth."""

    def __init__(self, seed=2, hg=1.65, hc=5.0, wl=7.0, wr=7.5, z_end=600.0, tex_size=512,
                 cells=(0.035, 0.09, 0.24, 0.65, 1.8, 5.0)):
        rng = np.random.default_rng(seed)
        self.planes = [
            (1, +hg, (0, 2)),   # ground: y = hg, texture coords (x, z)
            (1, -hc, (0, 2)),   # ceiling
            (0, -wl, (2, 1)),   # left wall: coords (z, y)
            (0, +wr, (2, 1)),   # right wall
            (2, z_end, (0, 1)),  # end wall
        ]
        self.cells = cells
        self.tex = [rng.random((tex_size, tex_size)).astype(np.float32) for _ in cells]
        self.offs = rng.uniform(0, tex_size, (len(self.planes), len(cells), 2))
        self.tex_size = tex_size

    def _sample(self, o, a, b):
        T = self.tex[o]
        n = self.tex_size
        a0 = np.floor(a)
        b0 = np.floor(b)
        fa = (a - a0).astype(np.float32)
        fb = (b - b0).astype(np.float32)
        ia = a0.astype(np.int64) % n
        ib = b0.astype(np.int64) % n
        ia1 = (ia + 1) % n
        ib1 = (ib + 1) % n
        # smoothstep-interpolated value noise
        fa = fa * fa * (3 - 2 * fa)
        fb = fb * fb * (3 - 2 * fb)
        v = (T[ib, ia] * (1 - fa) + T[ib, ia1] * fa) * (1 - fb) + \
            (T[ib1, ia] * (1 - fa) + T[ib1, ia1] * fa) * fb
        return v

    def cast(self, T_wc, K, width, height, pix=None):
        """Ray-cast. Returns (depth z in camera frame, plane id, hit point world) for the
        full image, or for the given pixel list `pix` (n x 2)."""
        fx, fy, cx, cy = K
        if pix is None:
            u, v = np.meshgrid(np.arange(width, dtype=np.float64),
                               np.arange(height, dtype=np.float64))
        else:
            u, v = np.asarray(pix, np.float64)[:, 0], np.asarray(pix, np.float64)[:, 1]
        d_c = np.stack([(u - cx) / fx, (v - cy) / fy, np.ones_like(u)], -1)
        R, o = T_wc[:3, :3], T_wc[:3, 3]
        d_w = d_c @ R.T
        best_t = np.full(u.shape, np.inf)
        best_p = np.full(u.shape, -1, np.int32)
        for pid, (axis, val, _) in enumerate(self.planes):
            with np.errstate(divide="ignore", invalid="ignore"):
                t = (val - o[axis]) / d_w[..., axis]
            ok = (t > 1e-6) & (t < best_t)
            best_t = np.where(ok, t, best_t)
            best_p = np.where(ok, pid, best_p)
        hit = o + d_w * best_t[..., None]
        return best_t, best_p, hit  # d_c has z=1 so t is the camera-frame depth

    def render(self, T_wc, K, width, height):
        depth, pid, hit = self.cast(T_wc, K, width, height)
        fx = K[0]
        foot = depth / fx  # metres per pixel (fronto-parallel approximation)
        img = np.zeros(depth.shape, np.float32)
        wsum = np.zeros(depth.shape, np.float32)
        for p, (axis, val, (ca, cb)) in enumerate(self.planes):
            sel = pid == p
            if not sel.any():
                continue
            a = hit[..., ca][sel]
            b = hit[..., cb][sel]
            f = foot[sel]
            acc = np.zeros(a.shape, np.float32)
            ws = np.zeros(a.shape, np.float32)
            for o, cell in enumerate(self.cells):
                w = np.clip((cell / f - 2.0) / 2.0, 0.0, 1.0).astype(np.float32)
                if not (w > 0).any():
                    continue
                s = self._sample(o, a / cell + self.offs[p, o, 0], b / cell + self.offs[p, o, 1])
                acc += w * (s - 0.5)
                ws += w * w
            img[sel] = acc
            wsum[sel] = ws
        img = img / np.sqrt(np.maximum(wsum, 1e-6))
        out = np.clip(128.0 + 150.0 * img, 0, 255)
        return np.rint(out).astype(np.uint8), depth


def camera_trajectory(n_frames, seed=3, speed=0.8):
    """T_wc for each frame: forward motion ~speed m/frame with gentle sway/yaw."""
    rng = np.random.default_rng(seed)
    ph = rng.uniform(0, 2 * np.pi, 4)
    poses = []
    for k in range(n_frames):
        z = speed * k
        x = 0.25 * np.sin(0.07 * k + ph[0])
        y = 0.05 * np.sin(0.11 * k + ph[1])
        yaw = 0.02 * np.sin(0.05 * k + ph[2])
        pitch = 0.006 * np.sin(0.09 * k + ph[3])
        T = se3_exp([0, 0, 0, pitch, yaw, 0])
        T[:3, 3] = [x, y, z]
        poses.append(T)
    return poses


def bucket_points(width, height, n_u, n_v, rng, margin=16.0):
    """One jittered pixel per bucket of an n_u x n_v grid (the reference keeps one
    feature per bucket: feature_extractor.cpp:250-279), clamped `margin` px inside."""
    su, sv = width / n_u, height / n_v
    gu, gv = np.meshgrid(np.arange(n_u), np.arange(n_v))
    u = (gu + rng.uniform(0.15, 0.85, gu.shape)) * su
    v = (gv + rng.uniform(0.15, 0.85, gv.shape)) * sv
    u = np.clip(u, margin, width - 1 - margin)
    v = np.clip(v, margin, height - 1 - margin)
    return np.stack([u.ravel(), v.ravel()], 1)


class StereoStream:
    """Synthetic stereo stream + per-frame track sets for the steady-state frame
    operator (open loop: the track set entering frame k+1 comes from the scene's
    ground truth at frame k, perturbed; see DESIGN.md §bench workload).
    `margin` = 31 px keeps features as far from the border as the reference's ORB
    detector does (extractor_orb_->setEdgeThreshold(31), feature_extractor.cpp:53)."""

    def __init__(self, width=KITTI_SIZE[0], height=KITTI_SIZE[1], K=KITTI_K,
                 baseline=KITTI_BASELINE, n_u=60, n_v=25, n_new=150, seed=2, speed=0.8,
                 depth_noise=0.01, prior_noise=(0.02, 0.002), margin=31.0, z_end=600.0, tex_scale=1.0):
        """z_end: the corridor's end wall [m] — a forward-driving stream of n frames needs n * speed well below it.
        tex_scale: factor on the texture's cell sizes (1/3 for a camera of three times the focal length: the same detail per PIXEL)."""
        self.width, self.height, self.K, self.baseline = width, height, K, baseline
        self.n_u, self.n_v, self.n_new = n_u, n_v, n_new
        self.seed, self.speed = seed, speed
        self.depth_noise, self.prior_noise, self.margin = depth_noise, prior_noise, margin
        self.scene = CorridorScene(seed=seed, z_end=z_end) if tex_scale == 1.0 else \
            CorridorScene(seed=seed, z_end=z_end, cells=tuple(c * tex_scale for c in (0.035, 0.09, 0.24, 0.65, 1.8, 5.0)))
        self.z_end = z_end
        self.T_lr = stereo_T_lr(baseline)

    def poses(self, n_frames):
        return camera_trajectory(n_frames, seed=self.seed + 1, speed=self.speed)

    def render_pair(self, T_wc):
        L, depth = self.scene.render(T_wc, self.K, self.width, self.height)
        T_wr = T_wc @ self.T_lr.astype(np.float64)
        R, _ = self.scene.render(T_wr, self.K, self.width, self.height)
        return L, R, depth

    def track_set(self, k, T_wc_prev, T_wc_cur):
        """Inputs of frame step k (prev -> cur): pts_l0, pts_r0, Xp (prev-camera frame),
        prior motion dT (T_pc, perturbed ground truth), candidate new points in cur-left."""
        rng = np.random.default_rng(self.seed * 7919 + k)
        fx, fy, cx, cy = self.K
        pts = bucket_points(self.width, self.height, self.n_u, self.n_v, rng, self.margin)
        z, _, _ = self.scene.cast(T_wc_prev, self.K, self.width, self.height, pix=pts)
        X = np.stack([(pts[:, 0] - cx) / fx * z, (pts[:, 1] - cy) / fy * z, z], 1)
        pr = np.stack([fx * (X[:, 0] - self.baseline) / X[:, 2] + cx, pts[:, 1]], 1)
        Xn = X * (1.0 + rng.normal(0, self.depth_noise, (X.shape[0], 1)))
        dT_true = np.linalg.inv(T_wc_prev) @ T_wc_cur
        pert = np.concatenate([rng.normal(0, self.prior_noise[0], 3),
                               rng.normal(0, self.prior_noise[1], 3)])
        dT_prior = dT_true @ se3_exp(pert)
        new = bucket_points(self.width, self.height, max(self.n_new // 10, 1), 10, rng,
                            self.margin)[: self.n_new]
        return dict(pts_l0=pts.astype(np.float32), pts_r0=pr.astype(np.float32),
                    Xp=Xn.astype(np.float32), dT_prior=dT_prior.astype(np.float32),
                    dT_true=dT_true, pts_new=new.astype(np.float32))


def ba_window(n_kf=8, n_points=600, stereo=True, seed=0, K=KITTI_K, baseline=0.537, n_fix=2, px_noise=0.3,
              pose_noise=(0.02, 0.004), point_noise=0.05, width=1241, height=376, scale=0.1, right_only_frac=0.0,
              T_lr=None):
    """A local-BA window as SparseBAParameters hands it to the solver (sparse_ba_parameters.h:283-420):
    keyframe poses T_jw relative to the first keyframe with translations scaled by 1/pose_scale (0.1), the
    first n_fix poses fixed, landmarks in that frame and scale, per-landmark observation lists in keyframe
    order (left observation, then the right one of the same stereo keyframe). Returns a dict with the
    perturbed problem (T_jw, X), the ground truth and the CSR observation arrays."""
    rng = np.random.default_rng(seed)
    fx, fy, cx, cy = K
    # forward-moving rig with small rotations; keyframe 0 is the reference frame
    T_wj = [np.eye(4)]
    for _ in range(1, n_kf):
        w = rng.normal(0, 0.01, 3)
        th = np.linalg.norm(w)
        k = w / th
        Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
        dT = np.eye(4)
        dT[:3, :3] = np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * Kx @ Kx
        dT[:3, 3] = [rng.normal(0, 0.03), rng.normal(0, 0.02), 0.9 + rng.normal(0, 0.05)]
        T_wj.append(T_wj[-1] @ dT)
    if T_lr is None:
        T_lr = np.eye(4)
        T_lr[0, 3] = baseline
    T_rl = np.linalg.inv(T_lr)
    Xw = np.stack([rng.uniform(-12, 12, n_points), rng.uniform(-3, 3, n_points),
                   rng.uniform(4, 45, n_points)], axis=1)
    obs_ptr, obs_frame, obs_right, obs_px, keep = [0], [], [], [], []
    for i in range(n_points):
        fr, rt, px = [], [], []
        for j in range(n_kf):
            Xc = np.linalg.inv(T_wj[j]) @ np.append(Xw[i], 1.0)
            for right in ((0, 1) if stereo else (0,)):
                Xr = T_rl @ Xc if right else Xc
                if Xr[2] < 0.5:
                    continue
                u, v = fx * Xr[0] / Xr[2] + cx, fy * Xr[1] / Xr[2] + cy
                if not (5 < u < width - 5 and 5 < v < height - 5):
                    continue
                if right == 0 and stereo and rng.random() < right_only_frac:
                    continue  # seen in the right image only
                fr.append(j)
                rt.append(right)
                px.append([u + rng.normal(0, px_noise), v + rng.normal(0, px_noise)])
        if len(fr) >= 2:  # THRES_MINIMUM_SEEN, sparse_ba_parameters.h:308
            keep.append(i)
            obs_frame += fr
            obs_right += rt
            obs_px += px
            obs_ptr.append(len(obs_frame))
    Xw = Xw[keep]
    T_jw_true = np.stack([np.linalg.inv(T) for T in T_wj])
    T_jw_true[:, :3, 3] *= scale
    X_true = Xw * scale
    T_jw = T_jw_true.copy()
    for j in range(n_fix, n_kf):
        xi = np.concatenate([rng.normal(0, pose_noise[0] * scale, 3), rng.normal(0, pose_noise[1], 3)])
        w = xi[3:]
        th = np.linalg.norm(w)
        k = w / th
        Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
        dT = np.eye(4)
        dT[:3, :3] = np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * Kx @ Kx
        dT[:3, 3] = xi[:3]
        T_jw[j] = dT @ T_jw[j]
    X = X_true + rng.normal(0, point_noise * scale, X_true.shape) * (X_true[:, 2:3] / (10 * scale))
    opt_index = np.array([-1] * n_fix + list(range(n_kf - n_fix)), np.int32)
    T_lr_s = T_lr.copy()
    T_lr_s[:3, 3] *= scale
    return dict(T_jw=T_jw, X=X, T_jw_true=T_jw_true, X_true=X_true, opt_index=opt_index,
                obs_ptr=np.array(obs_ptr, np.int32), obs_frame=np.array(obs_frame, np.int32),
                obs_right=np.array(obs_right, np.uint8), obs_px=np.array(obs_px, np.float64), K=np.array(K, np.float64),
                T_lr=T_lr_s, stereo=stereo)
