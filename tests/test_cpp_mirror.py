"""The C++ host-side classes (same names / argument order / throw behaviour as the
reference's FeatureTracker and MotionEstimator) on top of the C ABI: compile check on
CPU, behaviour on the GPU."""
import os
import struct
import subprocess

import numpy as np
import pytest

from visual_odometry_ros_amd import synthetic as S
from util import grid_points, image_pair, move_points

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "visual_odometry_ros_amd", "lib")


def _compile(tmp_path):
    exe = str(tmp_path / "host_mirror_demo")
    cmd = ["g++", "-std=c++17", "-O2", "-I", ROOT, os.path.join(ROOT, "tests", "cpp", "host_mirror_demo.cpp"), "-o", exe,
           "-L", LIBDIR, "-lvo_hip", f"-Wl,-rpath,{LIBDIR}", "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib",
           "-lamdhip64"]
    subprocess.check_call(cmd)
    return exe


def test_cpp_mirror_compiles_and_links(vo, tmp_path):
    vo.load()
    exe = _compile(tmp_path)
    assert os.path.exists(exe)
    # the reference-typed adapter is gated on OpenCV/Eigen headers and must at least preprocess away
    subprocess.check_call(["g++", "-std=c++17", "-fsyntax-only", "-I", ROOT, "-x", "c++",
                           os.path.join(ROOT, "visual_odometry_ros_amd", "core", "visual_odometry",
                                        "reference_adapter.h")])


@pytest.mark.gpu
def test_cpp_mirror_matches_python_api_and_oracle(ctx, vo, oracle, tmp_path):
    exe = _compile(tmp_path)
    d = S.two_view_points(n=500, seed=1)
    motion = dict(dx=2.5, dy=-1.5, scale=1.0, angle=0.003)
    img0, img1 = image_pair(200, 280, seed=77, **motion)
    pts0 = grid_points(200, 280, step=16, margin=20)
    prior = (move_points(pts0.astype(np.float64), img0.shape, **motion) + 0.4).astype(np.float32)
    n, npt = 500, pts0.shape[0]
    inp, outp = tmp_path / "in.bin", tmp_path / "out.bin"
    with open(inp, "wb") as f:
        f.write(struct.pack("4i", n, 280, 200, npt))
        for a in (d["X"], d["pts_l"], d["pts_r"], d["K"], d["T_lr"].reshape(16), img0, img1, pts0, prior):
            f.write(np.ascontiguousarray(a).tobytes())
    subprocess.check_call([exe, str(inp), str(outp)])
    raw = open(outp, "rb").read()
    ok, threw, iters = struct.unpack_from("3i", raw, 0)
    off = 12
    T = np.frombuffer(raw, np.float32, 16, off).reshape(4, 4); off += 64
    inl = np.frombuffer(raw, np.uint8, n, off).astype(bool); off += n
    ptk = np.frombuffer(raw, np.float32, 2 * npt, off).reshape(-1, 2); off += 8 * npt
    mv = np.frombuffer(raw, np.uint8, npt, off).astype(bool); off += npt
    ref = np.frombuffer(raw, np.float32, 2 * npt, off).reshape(-1, 2); off += 8 * npt
    mv2 = np.frombuffer(raw, np.uint8, npt, off).astype(bool)
    assert ok == 1 and threw == 1  # stereo call on a mono estimator throws, as in the reference
    rc, T_o, mask_o, info_o = oracle.gn_pose_stereo(d["X"], d["pts_l"], d["pts_r"], d["K"], d["K"], d["T_lr"], 3.0,
                                                    np.eye(4, dtype=np.float32), oracle.SUM_TREE, 512)
    assert iters == info_o.iterations and np.array_equal(inl, mask_o)
    assert np.linalg.norm(T - T_o) / np.linalg.norm(T_o) < 1e-6
    rc, p_o, m_o = oracle.track_with_prior(img0, img1, pts0, prior, 21, 4, 80.0)
    assert np.array_equal(mv, m_o) and np.array_equal(ptk, p_o)
    rc, r_o, m2_o, _ = oracle.track_with_scale(img0, img1, pts0, np.ones(npt, np.float32), p_o, m_o,
                                               oracle.IC_REFERENCE, oracle.SUM_TREE)
    # the C++ call passed a fresh (all-true) mask, like the reference drivers do after compaction
    rc, r_o2, m2_o2, _ = oracle.track_with_scale(img0, img1, pts0, np.ones(npt, np.float32), p_o, None,
                                                 oracle.IC_REFERENCE, oracle.SUM_TREE)
    assert np.array_equal(mv2, m2_o2) and np.array_equal(ref, r_o2)
