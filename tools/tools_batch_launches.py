"""Is it the NUMBER OF LAUNCHES per frame that keeps S streams on one GPU from scaling? The batch driver over S = 1, 2,
4 streams as bench.py's streams_per_gpu leg runs it, and again with the keypoint detection switched off after two fills of each
candidate table (vo_debug_set VO_DBG_SKIP_DETECT: 17 of a frame's ~35 launches go away; the results are those of stale
tables — a measurement, not a mode).  usage: python tools/tools_batch_launches.py"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    cfg = bench.CONFIGS[1]
    nf, warm, S_list = 48, 8, (1, 2, 4)
    imgs = [bench.render_stream(cfg, 100 + q, nf, bench.host_cores()) for q in range(max(S_list))]
    import torch
    import visual_odometry_ros_amd as V
    from visual_odometry_ros_amd import synthetic as S
    V.load()
    dev = torch.device("cuda", 0)
    W, H = cfg["W"], cfg["H"]
    cap = 2 * cfg["n_u"] * cfg["n_v"] + 1024
    st = S.StereoStream(width=W, height=H, K=cfg["K"], n_u=cfg["n_u"], n_v=cfg["n_v"], seed=2, speed=cfg["speed"])
    ctx = V.Context(device=0, max_width=W, max_height=H, max_points=cap, n_slots=5, max_level=cfg["max_level"])
    thr = cfg["thres"]
    svo = V.StereoVO(ctx, W, H, cfg["K"], cfg["K"], st.T_lr, cfg["n_u"], cfg["n_v"], thres_fastscore=cfg["thres_fast"], window_size=cfg["win"],
                     max_level=cfg["max_level"], thres_error=thr[0], thres_bidirection=thr[1], thres_poseba_error=thr[2], strict_border=4,
                     local_ba=True)
    prm = svo.prm
    svo.close()
    ctx.close()
    d = [[(torch.from_numpy(np.ascontiguousarray(L)).to(dev), torch.from_numpy(np.ascontiguousarray(R)).to(dev)) for L, R in s] for s in imgs]
    torch.cuda.synchronize()
    Lp = [[a.data_ptr() for a, _ in s] for s in d]
    Rp = [[b.data_ptr() for _, b in s] for s in d]
    out = {}
    for skip in (0, 1):
        for Sn in S_list:
            b = V.StereoBatch(0, Sn, W, H, cap, cfg["max_level"], prm)
            if skip:
                b.debug_set(3, 1)  # VO_DBG_SKIP_DETECT
            r = b.run(Lp[:Sn], Rp[:Sn], W, warmup=warm)
            b.close()
            out[f"S={Sn} skip_detect={skip}"] = round(Sn * (nf - warm) / r["wall"], 1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
