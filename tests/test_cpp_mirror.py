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


STUBS = os.path.join(ROOT, "tests", "typecheck_stubs")


def _compile(tmp_path, name="host_mirror_demo", extra=()):
    exe = str(tmp_path / name)
    cmd = ["g++", "-std=c++17", "-O2", "-I", ROOT, *extra, os.path.join(ROOT, "tests", "cpp", name + ".cpp"), "-o", exe,
           "-L", LIBDIR, "-lvo_hip", f"-Wl,-rpath,{LIBDIR}", "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib",
           "-lamdhip64"]
    subprocess.check_call(cmd)
    return exe


def test_cpp_mirror_compiles_and_links(vo, tmp_path):
    vo.load()
    exe = _compile(tmp_path)
    assert os.path.exists(exe)
    assert os.path.exists(_compile(tmp_path, "frame_demo"))
    assert os.path.exists(_compile(tmp_path, "stereo_vo_demo"))
    assert os.path.exists(_compile(tmp_path, "mono_vo_demo"))
    # the reference-typed adapter, against the type-check stand-ins (tests/test_reference_adapter.py has the details)
    assert os.path.exists(_compile(tmp_path, "adapter_demo", ADAPTER_INC))


ADAPTER_INC = ("-I", os.path.join(STUBS, "thirdparty"), "-I", os.path.join(STUBS, "reference"))


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
    # the reference-typed adapter (cv::Mat / cv::Point2f / column-major Eigen::Matrix4f / CameraConstPtr signatures,
    # lazily sized contexts) makes the same calls and must produce the same bytes (its header has no GN info: word 2)
    exe_a = _compile(tmp_path, "adapter_demo", ADAPTER_INC)
    outa = tmp_path / "out_adapter.bin"
    subprocess.check_call([exe_a, str(inp), str(outa)])
    raw_a = open(outa, "rb").read()
    assert len(raw_a) == len(raw) and raw_a[:8] == raw[:8] and raw_a[12:] == raw[12:]


@pytest.mark.gpu
def test_cpp_frame_pipelines_match_python_api(vo, tmp_path):
    """vo::StereoFramePipeline / vo::MonoFramePipeline (frame_pipeline.h) give the Python mirrors' results
    bit for bit (those are checked against the oracle in test_frame_gpu.py / test_mono_frame_gpu.py)."""
    from visual_odometry_ros_amd.api import (MonoFramePipeline, StereoFramePipeline, make_mono_params,
                                             make_stereo_params)
    exe = _compile(tmp_path, "frame_demo")
    W, H, win, lvl = 640, 240, 21, 4
    K = (400.0, 400.0, 320.0, 120.0)
    stream = S.StereoStream(width=W, height=H, K=K, n_u=24, n_v=10, n_new=30, seed=11, margin=6.0)
    poses = stream.poses(3)
    L0, _, _ = stream.render_pair(poses[1])
    L1, R1, _ = stream.render_pair(poses[2])
    ts = stream.track_set(1, poses[1], poses[2])
    n, nn = ts["pts_l0"].shape[0], ts["pts_new"].shape[0]
    rng = np.random.default_rng(4)
    flags = ((rng.random(n) < 0.7).astype(np.uint8) | ((rng.random(n) < 0.8).astype(np.uint8) << 1)).astype(np.uint8)
    dT = ts["dT_prior"].astype(np.float32)
    Tcw_prev = np.eye(4, dtype=np.float32)
    Tcw_prior = np.linalg.inv(dT.astype(np.float64)).astype(np.float32)
    Xw = ts["Xp"].astype(np.float32)
    inp, outp = tmp_path / "fin.bin", tmp_path / "fout.bin"
    with open(inp, "wb") as f:
        f.write(struct.pack("6i", W, H, n, nn, win, lvl))
        for a in (np.asarray(K, np.float32), stream.T_lr.astype(np.float32).reshape(16), dT.reshape(16), L0, L1, R1,
                  ts["pts_l0"], ts["pts_r0"], ts["Xp"].astype(np.float32), ts["pts_new"], Xw, flags,
                  Tcw_prev.reshape(16), Tcw_prior.reshape(16)):
            f.write(np.ascontiguousarray(a).tobytes())
        ba = S.ba_window(n_kf=6, n_points=200, stereo=False, seed=8)
        f.write(struct.pack("3i", ba["T_jw"].shape[0], ba["X"].shape[0], ba["obs_px"].shape[0]))
        for a in (ba["T_jw"], ba["opt_index"], ba["X"], ba["obs_ptr"], ba["obs_frame"], ba["obs_right"], ba["obs_px"],
                  ba["K"]):
            f.write(np.ascontiguousarray(a).tobytes())
    subprocess.check_call([exe, str(inp), str(outp)])
    raw = open(outp, "rb").read()
    off = 0

    def take(dt, cnt):
        nonlocal off
        a = np.frombuffer(raw, dt, cnt, off)
        off += a.nbytes
        return a

    # ---- stereo
    c = vo.Context(device=0, max_width=W, max_height=H, max_points=n + nn + 64, n_slots=3, max_level=lvl)
    try:
        c.set_image(0, L0)
        c.set_image(1, L1)
        c.set_image(2, R1)
        pipe = StereoFramePipeline(c, make_stereo_params(W, H, win, lvl, 80.0, 0.5, 3.0, K, K, stream.T_lr), True)
        pipe.enqueue(ts["pts_l0"], ts["pts_r0"], ts["Xp"], dT, ts["pts_new"], lm_flags=flags & 1)
        g = pipe.result()
        head = take(np.int32, 5)
        assert head[0] == 1 and head[1] == g["counts"].n_inlier and head[2] == g["gn"].iterations and head[3] == 1
        assert head[4] == g["counts"].n_ba and 0 < head[4] < g["counts"].n_l1r1
        assert g["counts"].n_inlier > 0.5 * n
        assert np.array_equal(take(np.float32, 16).reshape(4, 4), g["dT"])
        assert np.array_equal(take(np.float32, 2 * n).reshape(-1, 2), g["pts_l1"])
        assert np.array_equal(take(np.float32, 2 * n).reshape(-1, 2), g["pts_r1"])
        assert np.array_equal(take(np.uint8, n), g["stage"])
        assert np.array_equal(take(np.float32, 2 * nn).reshape(-1, 2), g["pts_new_r"])
        assert np.array_equal(take(np.uint8, nn).astype(bool), g["mask_new"])
        # ---- mono
        mp = MonoFramePipeline(c, make_mono_params(W, H, win, lvl, 20.0, 1.0, 5, 1.0, K), True)
        mp.enqueue(ts["pts_l0"], Xw, flags, Tcw_prev, Tcw_prior, dT)
        m = mp.result()
        head = take(np.int32, 4)
        assert head[0] == 0 and head[1] == m["counts"].n_final and head[2] == m["counts"].n_ba
        assert head[3] == m["gn"].iterations and m["counts"].n_final > 0.5 * n
        assert np.array_equal(take(np.float32, 16).reshape(4, 4), m["dT01"])
        assert np.array_equal(take(np.float32, 2 * n).reshape(-1, 2), m["pts1"])
        assert np.array_equal(take(np.float32, n), m["scale"])
        assert np.array_equal(take(np.uint8, n), m["stage"])
        # ---- StereoCamera (camera.h) against the Python mirror
        from visual_odometry_ros_amd.api import StereoCamera
        sc = StereoCamera(c)
        sc.initParams(W, H, K, (-0.12, 0.03, 0.0004, -0.0002, 0.0),
                      tuple(np.float32(v) for v in (np.float32(K[0]) * np.float32(1.01), np.float32(K[1]) * np.float32(0.99),
                                                    np.float32(K[2]) - np.float32(3.0), np.float32(K[3]) + np.float32(2.0))),
                      (-0.11, 0.025, -0.0003, 0.0001, 0.0))
        sc.setStereoPoseLeft2Right(stream.T_lr)
        sc.initStereoCameraToRectify()
        sc.rectifyStereoImages(L1, R1, 0, 1)
        head = take(np.int32, 2)
        assert head[0] == 1
        assert np.array_equal(take(np.float32, 4), sc.getRectifiedCamera())
        assert np.array_equal(take(np.float32, 16).reshape(4, 4), sc.getRectifiedStereoPoseLeft2Right())
        assert np.array_equal(take(np.uint8, W * H).reshape(H, W), c.get_level(0, 0))
        assert np.array_equal(take(np.uint8, W * H).reshape(H, W), c.get_level(1, 0))
        # ---- FeatureExtractor::extractORBwithBinning_fast (feature_extractor.h) against the Python mirror
        fe = vo.FeatureExtractor(c)
        fe.initParams(W, H, 16, 8, THRES_FAST=15)
        fe.suppressCenterBins()
        c.set_image(0, L1)
        pts_p = fe.extractORBwithBinning_fast(0)
        npts = int(take(np.int32, 1)[0])
        assert npts == pts_p.shape[0] and npts > 40
        assert np.array_equal(take(np.float32, 2 * npts).reshape(-1, 2), pts_p)
        # ---- SparseBundleAdjustmentSolver (sparse_bundle_adjustment.h) against the Python mirror
        from visual_odometry_ros_amd.api import SparseBundleAdjustmentSolver
        sol = SparseBundleAdjustmentSolver(c, False)
        sol.setCamera(ba["K"])
        sol.setHuberThreshold(0.5)
        ok_p, T_p, X_p, err_p = sol.solveForFiniteIterations(10, ba["T_jw"], ba["opt_index"], ba["X"], ba["obs_ptr"],
                                                             ba["obs_frame"], ba["obs_right"], ba["obs_px"])
        head = take(np.int32, 2)
        assert head[0] == int(ok_p) == 1 and head[1] == 1
        assert np.array_equal(take(np.float64, 10), err_p)
        assert np.array_equal(take(np.float64, 16 * T_p.shape[0]).reshape(-1, 4, 4), T_p)
        assert np.array_equal(take(np.float64, 3 * X_p.shape[0]).reshape(-1, 3), X_p)
        assert off == len(raw)
    finally:
        c.close()


@pytest.mark.gpu
@pytest.mark.parametrize("hand_over", [1, 2])
def test_cpp_stereo_vo_writes_the_python_loops_trajectory(vo, tmp_path, hand_over):
    """vo::StereoVO (core/visual_odometry/stereo_vo.h: trackStereoImages / getStatistics on POD images) over 20 pairs, with
    the local BA and the one-frame-ahead hand-over: frame ids, keyframes, track-set sizes and poses equal to the Python
    mirror's (api.StereoVO) bit for bit, and the trajectory file it leaves behind has the same bytes as the one written
    from the Python loop's poses (the reference's dump format). hand_over 1: enqueue / prefetch / result called per frame by the
    program; 2: trackSequence (vo_svo_run, the loop inside the library)."""
    W, H, K = 640, 240, (400.0, 400.0, 320.0, 120.0)
    st = S.StereoStream(width=W, height=H, K=K, n_u=20, n_v=8, seed=5, speed=0.5)
    n = 20
    imgs = [st.render_pair(p)[:2] for p in st.poses(n)]
    kw = dict(thres_trans=1.2, thres_alive_ratio=0.6, thres_rotation=15.0)
    exe = _compile(tmp_path, "stereo_vo_demo")
    inp, outp, traj = tmp_path / "svo_in.bin", tmp_path / "svo_out.bin", tmp_path / "traj_cpp.txt"
    with open(inp, "wb") as f:
        f.write(struct.pack("9i", n, W, H, 20, 8, 21, 4, hand_over, 1))
        f.write(np.array(list(K) + list(np.asarray(st.T_lr, np.float32).reshape(16)) +
                         [80.0, 0.5, 3.0, kw["thres_alive_ratio"], kw["thres_trans"], kw["thres_rotation"]], np.float32).tobytes())
        for L, R in imgs:
            f.write(np.ascontiguousarray(L).tobytes())
            f.write(np.ascontiguousarray(R).tobytes())
    subprocess.check_call([exe, str(inp), str(outp), str(traj)])
    blob = np.fromfile(outp, np.uint8)
    raw, tail = blob[:n * 80].reshape(n, 16 + 64), blob[n * 80:].tobytes()
    rec = raw[:, :16].copy().view(np.int32)
    T_cpp = raw[:, 16:].copy().view(np.float32).reshape(n, 4, 4)
    c = vo.Context(device=0, max_width=W, max_height=H, max_points=4096, n_slots=5, max_level=4)
    svo = vo.StereoVO(c, W, H, K, K, st.T_lr, 20, 8, thres_fastscore=15, window_size=21, max_level=4, local_ba=True, **kw)
    ids, Ts = [], []
    for k, (L, R) in enumerate(imgs):
        i = svo.trackStereoImages(L, R)
        assert (i.frame_id, i.is_keyframe, i.n_tracks_out, i.lba_ran) == tuple(int(v) for v in rec[k]), k
        T = np.array(i.T_wc, np.float32).reshape(4, 4)
        assert np.array_equal(T.view(np.uint32), T_cpp[k].view(np.uint32)), k
        ids.append(i.frame_id)
        Ts.append(T)
    # stats_keyframe (what the ROS 2 node publishes): the C++ class's against the Python mirror's, byte for byte
    kfs = svo.getKeyframes()
    (nk,), off = struct.unpack_from("i", tail, 0), 4
    assert nk == len(kfs) >= 5
    for T, X in kfs:
        T_c = np.frombuffer(tail, np.float32, 16, off).reshape(4, 4); off += 64
        (m,) = struct.unpack_from("i", tail, off); off += 4
        X_c = np.frombuffer(tail, np.float32, 3 * m, off).reshape(m, 3); off += 12 * m
        assert np.array_equal(T_c.view(np.uint32), T.view(np.uint32)) and np.array_equal(X_c.view(np.uint32), X.view(np.uint32))
    assert off == len(tail)
    svo.close()
    c.close()
    assert rec[:, 3].sum() >= 3 and rec[:, 1].sum() >= 5
    vo.write_trajectory(str(tmp_path / "traj_py.txt"), ids, np.stack(Ts))
    assert open(traj, "rb").read() == open(tmp_path / "traj_py.txt", "rb").read()


@pytest.mark.gpu
def test_cpp_mono_vo_matches_python_mirror(vo, tmp_path):
    """vo::MonoVO (core/visual_odometry/mono_vo.h) over 12 images, its 5-point callable fed from the file, against the
    Python mirror on the same library: frame ids, keyframe decisions, track counts, local-BA runs, the hook's use, pose
    bits and stats_keyframe (poses + map points) byte for byte."""
    K = (458.654, 457.296, 367.215, 248.375)
    W, H, nu, nv, n = 752, 480, 40, 25, 12
    st = S.StereoStream(width=W, height=H, K=K, n_u=nu, n_v=nv, seed=5, speed=0.25)
    poses = st.poses(n)
    imgs = [st.render_pair(p)[0] for p in poses]
    T10 = np.stack([np.eye(4)] + [np.linalg.inv(poses[k]) @ poses[k - 1] for k in range(1, n)]).astype(np.float32)
    exe = _compile(tmp_path, "mono_vo_demo")
    inp, outp = tmp_path / "mvo_in.bin", tmp_path / "mvo_out.bin"
    with open(inp, "wb") as f:
        f.write(struct.pack("8i", n, W, H, nu, nv, 15, 5, 1))
        f.write(np.array(list(K) + [20.0, 1.0, 5.0, 1.0, 1.0, 2.5], np.float32).tobytes())
        f.write(T10.tobytes())
        for im in imgs:
            f.write(np.ascontiguousarray(im).tobytes())
    subprocess.check_call([exe, str(inp), str(outp)])
    blob = np.fromfile(outp, np.uint8)
    raw, tail = blob[:n * 84].reshape(n, 20 + 64), blob[n * 84:].tobytes()
    rec = raw[:, :20].copy().view(np.int32)
    T_cpp = raw[:, 20:].copy().view(np.float32).reshape(n, 4, 4)
    state = {"k": 0}

    def hook(p0, p1):
        T = T10[state["k"]]
        return True, T[:3, :3], T[:3, 3], np.ones(len(p0), bool)

    c = vo.Context(device=0, max_width=W, max_height=H, max_points=2 * nu * nv + 512, n_slots=3, max_level=5)
    mvo = vo.MonoVO(c, W, H, K, nu, nv, hook, thres_fastscore=15, window_size=15, max_level=5, thres_error=20.0, thres_bidirection=1.0,
                    thres_poseba_error=5, thres_sampson=1.0, thres_parallax=1.0, thres_translation=2.5, strict_border=1, local_ba=True)
    for k, im in enumerate(imgs):
        state["k"] = k
        i = mvo.trackImage(im)
        assert (i.frame_id, i.is_keyframe, i.n_tracks_out, i.lba_ran, i.used_five_point) == tuple(int(v) for v in rec[k]), k
        assert np.array_equal(np.array(i.T_wc, np.float32).view(np.uint32), T_cpp[k].reshape(-1).view(np.uint32)), k
    kfs = mvo.getKeyframes()
    (nk,), off = struct.unpack_from("i", tail, 0), 4
    assert nk == len(kfs) >= 3
    for T, X in kfs:
        T_c = np.frombuffer(tail, np.float32, 16, off).reshape(4, 4); off += 64
        (m,) = struct.unpack_from("i", tail, off); off += 4
        X_c = np.frombuffer(tail, np.float32, 3 * m, off).reshape(m, 3); off += 12 * m
        assert np.array_equal(T_c.view(np.uint32), T.view(np.uint32)) and np.array_equal(X_c.view(np.uint32), X.view(np.uint32))
    assert off == len(tail)
    mvo.close()
    c.close()
    assert rec[:, 3].sum() >= 1 and rec[1, 4] == 1
