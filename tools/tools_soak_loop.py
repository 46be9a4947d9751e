"""The closed loop over a long stream: where (if anywhere) does it stop, and what did the frames before look like?
usage: python tools/tools_soak_loop.py [--frames 2000] [--lba 1]"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=2000)
    ap.add_argument("--lba", type=int, default=1)
    ap.add_argument("--strict-border", type=int, default=4)
    a = ap.parse_args()
    cfg = bench.CONFIGS[1]
    imgs = bench.render_stream(cfg, 2, a.frames, 0)
    import visual_odometry_ros_amd as V
    from visual_odometry_ros_amd import synthetic as S
    st = S.StereoStream(width=cfg["W"], height=cfg["H"], K=cfg["K"], n_u=cfg["n_u"], n_v=cfg["n_v"], seed=2, speed=cfg["speed"])
    gt = st.poses(a.frames)
    cap = 2 * cfg["n_u"] * cfg["n_v"] + 1024
    ctx = V.Context(device=0, max_width=cfg["W"], max_height=cfg["H"], max_points=cap, n_slots=5, max_level=cfg["max_level"])
    thr = cfg["thres"]
    svo = V.StereoVO(ctx, cfg["W"], cfg["H"], cfg["K"], cfg["K"], st.T_lr, cfg["n_u"], cfg["n_v"], thres_fastscore=cfg["thres_fast"],
                     window_size=cfg["win"], max_level=cfg["max_level"], thres_error=thr[0], thres_bidirection=thr[1],
                     thres_poseba_error=thr[2], strict_border=a.strict_border, local_ba=bool(a.lba))
    T0i = np.linalg.inv(gt[0])
    log = []
    try:
        for k in range(a.frames):
            i = svo.trackStereoImages(*imgs[k])
            T = np.array(i.T_wc, np.float64).reshape(4, 4)
            err = float(np.linalg.norm(T[:3, 3] - (T0i @ gt[k])[:3, 3]))
            log.append((k, i.n_tracks_in, i.n_final, i.n_new, int(i.is_keyframe), int(i.lba_ran), i.lba_landmarks, round(i.lba_err_first, 3),
                        round(i.lba_err_last, 3), i.gn.iterations, round(err, 3)))
            if k % 200 == 0:
                print("frame", log[-1], flush=True)
    except Exception as e:  # noqa: BLE001
        print("STOPPED at frame", len(log), ":", e)
    for r in log[-12:]:
        print(r)
    print("frames done", len(log), "keyframes", sum(r[4] for r in log), "lba", sum(r[5] for r in log))


if __name__ == "__main__":
    main()
