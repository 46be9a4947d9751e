"""Soak of the strict-border stereo frame against the oracle: many seeds, features placed 4 px from the border
(long chains of border-touching features), every frame compared at every gate. Reuses the parity check of
tests/test_frame_gpu.py. usage: python tests/measure/frame_soak.py [--seeds 12] [--frames 12]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=12)
    ap.add_argument("--frames", type=int, default=12)
    ap.add_argument("--strict", type=int, default=1, help="strict-border mode of the pipeline: 1, 3 (concurrent replay), 4 (automatic)")
    a = ap.parse_args()
    import visual_odometry_ros_amd as V
    from oracle import oracle as O
    from visual_odometry_ros_amd import synthetic as S
    import test_frame_gpu as T
    V.load()
    O.build()
    ctx = V.Context(device=0, max_width=1241, max_height=480, max_points=8192, n_slots=4, max_level=6)
    t0 = time.time()
    n = 0
    for seed in range(100, 100 + a.seeds):
        for (w, h, nu, nv, win) in ((1241, 376, 60, 25, 21), (752, 480, 40, 25, 15)):
            K = S.KITTI_K if w == 1241 else (458.654, 457.296, 367.215, 248.375)
            stream = S.StereoStream(width=w, height=h, K=K, n_u=nu, n_v=nv, n_new=100, seed=seed, margin=4.0,
                                    speed=0.8 if w == 1241 else 0.3)
            T._run_stream(ctx, O, stream, a.frames, a.strict, win=win, max_level=6 if w == 1241 else 5, sanity=False)
            n += a.frames - 1
    print(f"{n} strict-border frames identical to the oracle at every gate ({time.time() - t0:.0f} s)")


if __name__ == "__main__":
    main()
