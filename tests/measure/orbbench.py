"""Time of FeatureExtractor::extractORBwithBinning_fast on the device (detection + bucketing, result read
back) and of the CPU restatement. usage: python tests/measure/orbbench.py"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def main():
    import visual_odometry_ros_amd as V
    from oracle import oracle as O
    from visual_odometry_ros_amd import synthetic as S
    V.load()
    out = []
    for (W, H, nu, nv, thr) in ((1241, 376, 60, 25, 15), (752, 480, 40, 25, 15), (3840, 2160, 100, 80, 15)):
        ctx = V.Context(device=0, max_width=W, max_height=H, max_points=nu * nv + 64, n_slots=2, max_level=4)
        K = (0.58 * W, 0.58 * W, W / 2, H / 2)
        stream = S.StereoStream(width=W, height=H, K=K, n_u=8, n_v=4, n_new=8, seed=4)
        img = stream.render_pair(stream.poses(1)[0])[0]
        fe = V.FeatureExtractor(ctx)
        fe.initParams(W, H, nu, nv, THRES_FAST=thr)
        ctx.set_image(0, img)
        for _ in range(5):
            pts = fe.extractORBwithBinning_fast(0)
        ctx.profile_enable(512)
        ctx.profile_reset()
        t0 = time.perf_counter()
        for _ in range(50):
            pts = fe.extractORBwithBinning_fast(0)
        dt = (time.perf_counter() - t0) / 50
        n, ms = ctx.profile_get(5)
        t0 = time.perf_counter()
        o = O.orb_detect(img, thr, max_kp=400000)
        dt_o = time.perf_counter() - t0
        out.append({"shape": f"{W}x{H}", "bins": nu * nv, "keypoints": fe.n_detected, "bucketed": int(pts.shape[0]),
                    "gpu_call_us": round(1e6 * dt, 1), "gpu_kernels_us": round(1e3 * ms / 50, 1),
                    "cpu_restatement_ms": round(1e3 * dt_o, 2)})
        ctx.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
