"""S streams per GPU through vo_batch_* for a few settings: where does the aggregate rate go?
usage: python tools/tools_batch.py [--frames 48]"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=48)
    ap.add_argument("--S", default="1,2,4,8")
    ap.add_argument("--lba", default="0,1")
    ap.add_argument("--strict", default="0,1,4")
    a = ap.parse_args()
    cfg = bench.CONFIGS[1]
    Ss = [int(v) for v in a.S.split(",")]
    imgs = [bench.render_stream(cfg, 100 + q, a.frames, 0) for q in range(max(Ss))]
    import torch
    import visual_odometry_ros_amd as V
    from visual_odometry_ros_amd import synthetic as S
    V.load()
    dev = torch.device("cuda", 0)
    d = [[(torch.from_numpy(np.ascontiguousarray(L)).to(dev), torch.from_numpy(np.ascontiguousarray(R)).to(dev)) for L, R in st] for st in imgs]
    torch.cuda.synchronize()
    Lp = [[x.data_ptr() for x, _ in st] for st in d]
    Rp = [[y.data_ptr() for _, y in st] for st in d]
    W, H = cfg["W"], cfg["H"]
    cap = 2 * cfg["n_u"] * cfg["n_v"] + 1024
    stt = S.StereoStream(width=W, height=H, K=cfg["K"], n_u=cfg["n_u"], n_v=cfg["n_v"], seed=2, speed=cfg["speed"])
    thr = cfg["thres"]
    for lba in [int(v) for v in a.lba.split(",")]:
        for strict in [int(v) for v in a.strict.split(",")]:
            c0 = V.Context(device=0, max_width=W, max_height=H, max_points=cap, n_slots=5, max_level=cfg["max_level"])
            svo = V.StereoVO(c0, W, H, cfg["K"], cfg["K"], stt.T_lr, cfg["n_u"], cfg["n_v"], thres_fastscore=cfg["thres_fast"], window_size=cfg["win"],
                             max_level=cfg["max_level"], thres_error=thr[0], thres_bidirection=thr[1], thres_poseba_error=thr[2],
                             strict_border=strict, local_ba=bool(lba))
            prm = svo.prm
            svo.close()
            c0.close()  # (its HIP streams must not hold hardware queues while the batch runs)
            row = []
            for Sn in Ss:
                b = V.StereoBatch(0, Sn, W, H, cap, cfg["max_level"], prm)
                t0 = time.time()
                r = b.run(Lp[:Sn], Rp[:Sn], W, warmup=8)
                b.close()
                row.append(round(Sn * (a.frames - 8) / r["wall"]))
            print(f"lba {lba} strict {strict}: aggregate fps for S={Ss}: {row}", flush=True)


if __name__ == "__main__":
    main()
