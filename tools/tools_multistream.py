"""Aggregate throughput of S independent stereo streams on ONE GPU (one vo_ctx, HIP stream and host thread per
sequence; ctypes releases the GIL during the calls). The bench metric is one stream per GPU — this shows how much
of the device a single latency-bound stream leaves idle. usage: python tools/tools_multistream.py [--streams 1,2,4,8]"""
import argparse
import gc
import json
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", default="1,2,4,8")
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--strict-border", type=int, default=1)
    args = ap.parse_args()
    import torch
    import visual_odometry_ros_amd as V
    from visual_odometry_ros_amd import synthetic as S
    from visual_odometry_ros_amd.api import StereoFramePipeline, make_stereo_params
    V.load()
    dev = torch.device("cuda", 0)
    N_U, N_V, N_NEW, WIN, LVL = 60, 25, 150, 21, 6
    F = 8
    order = list(range(F)) + list(range(F - 2, 0, -1))
    fid = lambda s: order[s % len(order)]

    class Seq:
        def __init__(self, seed):
            st = S.StereoStream(n_u=N_U, n_v=N_V, n_new=N_NEW, seed=seed)
            poses = st.poses(F)
            imgs = [st.render_pair(p)[:2] for p in poses]
            self.W, self.H = st.width, st.height
            self.dL = [torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a, _ in imgs]
            self.dR = [torch.from_numpy(np.ascontiguousarray(b)).to(dev) for _, b in imgs]
            self.ts, self.dts = {}, {}
            for s in range(len(order)):
                k = (fid(s), fid(s + 1))
                if k not in self.ts:
                    t = st.track_set(k[0] * 131 + k[1], poses[k[0]], poses[k[1]])
                    self.ts[k] = t
                    self.dts[k] = {q: torch.from_numpy(np.ascontiguousarray(t[q])).to(dev) for q in ("pts_l0", "pts_r0", "Xp", "pts_new")}
            self.ctx = V.Context(device=0, max_width=self.W, max_height=self.H, max_points=N_U * N_V + 64, n_slots=5, max_level=LVL)
            self.pipe = StereoFramePipeline(self.ctx, make_stereo_params(self.W, self.H, WIN, LVL, 80.0, 0.5, 3.0, st.K, st.K, st.T_lr),
                                            strict_border=int(args.strict_border))
            self.ctx.set_pyramid_window_hint(WIN)
            self.slot = {"P": 0, "CL": 1, "CR": 2, "NL": 3, "NR": 4}
            self.ctx.set_image_device(0, self.dL[fid(0)].data_ptr(), self.W, self.H, self.W)
            self.ctx.set_stereo_pair_device(1, self.dL[fid(1)].data_ptr(), 2, self.dR[fid(1)].data_ptr(), self.W, self.H, self.W)
            self.ctx.synchronize()

        def enqueue(self, s):
            k = (fid(s), fid(s + 1))
            t, sl = self.dts[k], self.slot
            self.pipe.enqueue_device(t["pts_l0"].data_ptr(), t["pts_r0"].data_ptr(), t["Xp"].data_ptr(), N_U * N_V,
                                     self.ts[k]["dT_prior"], t["pts_new"].data_ptr(), N_NEW, slots=(sl["P"], sl["CL"], sl["CR"]))
            nb = fid(s + 2)
            self.ctx.set_stereo_pair_device(sl["NL"], self.dL[nb].data_ptr(), sl["NR"], self.dR[nb].data_ptr(), self.W, self.H, self.W)

        def run(self, first, count):
            sl = self.slot
            self.enqueue(first)
            for s in range(first, first + count):
                self.pipe.result(copy=False)
                sl["P"], sl["CL"], sl["CR"], sl["NL"], sl["NR"] = sl["CL"], sl["NL"], sl["NR"], sl["P"], sl["CR"]
                if s + 1 < first + count:
                    self.enqueue(s + 1)

    counts = [int(v) for v in args.streams.split(",")]
    seqs = [Seq(2 + i) for i in range(max(counts))]
    torch.cuda.synchronize()
    out = []
    for n in counts:
        use = seqs[:n]
        for q in use:
            q.run(0, 10)
        bar = threading.Barrier(n + 1)

        def work(q):
            bar.wait()
            q.run(10, args.steps)
            q.ctx.synchronize()
            bar.wait()

        th = [threading.Thread(target=work, args=(q,)) for q in use]
        gc.collect(); gc.freeze(); gc.disable()
        for t in th:
            t.start()
        bar.wait()
        t0 = time.perf_counter()
        bar.wait()
        dt = time.perf_counter() - t0
        gc.enable()
        for t in th:
            t.join()
        out.append({"streams": n, "frames_per_s_total": round(n * args.steps / dt, 1), "per_stream": round(args.steps / dt, 1)})
    print(json.dumps({"strict_border": args.strict_border, "results": out}))


if __name__ == "__main__":
    main()
