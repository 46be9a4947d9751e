"""Phase times of gn_pose_kernel in frame mode (needs the -DGN_STAMP build): prologue (compaction, counts, control
block), point loads, iterations, epilogue. usage: VO_EXTRA_FLAGS=-DGN_STAMP python visual_odometry_ros_amd/build.py --force;
python tools/tools_gnstamp.py"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import visual_odometry_ros_amd as V
    from visual_odometry_ros_amd import synthetic as S
    from visual_odometry_ros_amd.api import StereoFramePipeline, make_stereo_params
    V.load()
    stream = S.StereoStream(seed=2)
    poses = stream.poses(6)
    ctx = V.Context(device=0, max_width=stream.width, max_height=stream.height, max_points=1700, n_slots=3, max_level=6)
    pipe = StereoFramePipeline(ctx, make_stereo_params(stream.width, stream.height, 21, 6, 80.0, 0.5, 3.0, stream.K, stream.K, stream.T_lr), True)
    rows = []
    for k in range(1, 6):
        L0 = stream.render_pair(poses[k - 1])[0]
        L1, R1, _ = stream.render_pair(poses[k])
        ts = stream.track_set(k, poses[k - 1], poses[k])
        ctx.set_image(0, L0); ctx.set_image(1, L1); ctx.set_image(2, R1)
        for _ in range(3):
            pipe.enqueue(ts["pts_l0"], ts["pts_r0"], ts["Xp"], ts["dT_prior"], ts["pts_new"])
            r = pipe.result()
        st = (C.c_longlong * 12)()
        ctx.check(ctx.lib.vo_debug_gn_stamps(ctx.handle, st))
        t = [st[i] for i in range(5)]
        rows.append(dict(iterations=r["counts"].gn_iterations, prologue_us=(t[1] - t[0]) / 100.0, loads_us=(t[2] - t[1]) / 100.0,
                         iterations_us=(t[3] - t[2]) / 100.0, epilogue_us=(t[4] - t[3]) / 100.0))
    for r in rows:
        print(r)


if __name__ == "__main__":
    main()
