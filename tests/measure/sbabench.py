"""Wall time of one local bundle adjustment (10 iterations) on the device vs the CPU restatement, and the
largest deviation between the two. usage: python tests/measure/sbabench.py"""
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
    from visual_odometry_ros_amd.api import SparseBundleAdjustmentSolver
    V.load()
    ctx = V.Context(device=0, max_width=64, max_height=64, max_points=64, n_slots=2, max_level=1)
    out = []
    for (n_kf, n_pts, stereo) in ((9, 1800, True), (9, 1800, False), (16, 6000, True)):
        p = S.ba_window(n_kf=n_kf, n_points=n_pts, stereo=stereo, seed=11)
        sol = SparseBundleAdjustmentSolver(ctx, stereo)
        (sol.setStereoCameras(p["K"], p["K"], p["T_lr"]) if stereo else sol.setCamera(p["K"]))
        sol.setHuberThreshold(0.5)
        args = (p["T_jw"], p["opt_index"], p["X"], p["obs_ptr"], p["obs_frame"], p["obs_right"], p["obs_px"])
        sol.solveForFiniteIterations(10, *args)
        ctx.profile_enable(64)
        ctx.profile_reset()
        t0 = time.perf_counter()
        for _ in range(5):
            ok, T, X, err = sol.solveForFiniteIterations(10, *args)
        dt = (time.perf_counter() - t0) / 5
        n, ms = ctx.profile_get(5)
        import ctypes as C
        ph = (C.c_int * 8)()
        ctx.lib.vo_debug_sba_phases(ctx.handle, ph)
        t0 = time.perf_counter()
        rc, T_o, X_o, err_o = O.sba_solve(*args, p["K"], p["K"] if stereo else None, p["T_lr"] if stereo else None, 0.5, 10)
        dt_o = time.perf_counter() - t0
        out.append({"keyframes": n_kf, "landmarks": int(p["X"].shape[0]), "observations": int(p["obs_px"].shape[0]),
                    "stereo": stereo, "gpu_call_ms": round(1e3 * dt, 3), "gpu_kernels_ms": round(ms / max(n, 1), 3),
                    "cpu_restatement_ms": round(1e3 * dt_o, 2), "max_abs_dev_pose": float(np.abs(T - T_o).max()),
                    "max_abs_dev_point": float(np.abs(X - X_o).max()), "solve_phases_us(assemble,ldlt,pose | own entries, wait, rank+rows, factorisation)": [round(v / 100.0, 1) for v in list(ph)[:7]], "err_first_last": [round(float(err[0]), 4), round(float(err[-1]), 4)]})
    print(json.dumps(out))


if __name__ == "__main__":
    main()
