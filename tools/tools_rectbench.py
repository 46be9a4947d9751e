"""Pyramid build of a stereo pair with and without the rectification remap fused into level 0
(HIP-event time of the whole launch chain, vo_profile class 0). usage: python tools/tools_rectbench.py"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))


def main():
    import visual_odometry_ros_amd as V
    from visual_odometry_ros_amd.api import StereoCamera
    from util import DeviceBuffer
    V.load()
    out = []
    for (W, H, win, lvl) in ((1241, 376, 21, 6), (752, 480, 15, 5), (3840, 2160, 21, 4)):
        ctx = V.Context(device=0, max_width=W, max_height=H, max_points=64, n_slots=2, max_level=lvl)
        ctx.set_pyramid_window_hint(win)
        sc = StereoCamera(ctx)
        K = (0.58 * W, 0.58 * W, W / 2 - 3.0, H / 2 + 2.0)
        sc.initParams(W, H, K, (-0.28, 0.07, 0.0002, 1e-5, 0.0), K, (-0.28, 0.07, -0.0001, -3e-5, 0.0))
        T = np.eye(4, dtype=np.float32)
        T[0, 3] = 0.11
        T[:3, :3] = [[0.99995, 0.0, 0.01], [0.0, 1.0, 0.0], [-0.01, 0.0, 0.99995]]
        sc.setStereoPoseLeft2Right(T)
        sc.initStereoCameraToRectify()
        rng = np.random.default_rng(0)
        L, R = (rng.integers(0, 256, (H, W), dtype=np.uint8) for _ in range(2))
        dL, dR = DeviceBuffer(L), DeviceBuffer(R)
        res = {"shape": f"{W}x{H}", "levels": ctx.pyramid_levels(W, H, win, lvl) + 1}
        for name, fn in (("plain", lambda: ctx.set_stereo_pair_device(0, dL.data_ptr(), 1, dR.data_ptr(), W, H, W)),
                         ("rectified", lambda: ctx.set_stereo_pair_rectified_device(0, dL.data_ptr(), 1, dR.data_ptr(), W, H, W))):
            for _ in range(20):
                fn()
            ctx.synchronize()
            ctx.profile_enable(256)
            ctx.profile_reset()
            for _ in range(200):
                fn()
            ctx.synchronize()
            n, ms = ctx.profile_get(0)
            res[name + "_us"] = round(1e3 * ms / n, 2)
        # algorithmic bytes of the fused level 0: raw read + two float maps + padded write, both images
        res["remap_alg_MB"] = round(2 * W * H * (1 + 8 + 1) / 1e6, 2)
        dL.free(); dR.free()
        ctx.close()
        out.append(res)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
