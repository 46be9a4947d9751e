"""The reference's YAML configuration files (OpenCV FileStorage dialect) through visual_odometry_ros_amd.config: a file
written here with the reference's keys (the numbers of config/stereo/kitti_00_stereo.yaml), and — where the reference
tree is on disk — every file under config/stereo."""
import glob
import os

import numpy as np
import pytest

from visual_odometry_ros_amd import config

KITTI_LIKE = """%YAML:1.0
# Camera Parameters
flagDoUndistortion: 0 # 0: do not undistort an image (KITTI), 1: undistort an image
Camera.left.fx: 718.856
Camera.left.fy: 718.856
Camera.left.cx: 607.1928
Camera.left.cy: 185.2157
Camera.left.k1: 0.0
Camera.left.k2: 0.0
Camera.left.k3: 0.0
Camera.left.p1: 0.0
Camera.left.p2: 0.0
Camera.left.width: 1241
Camera.left.height: 376
Camera.right.fx: 718.856
Camera.right.fy: 718.856
Camera.right.cx: 607.1928
Camera.right.cy: 185.2157
Camera.right.k1: 0.0
Camera.right.k2: 0.0
Camera.right.k3: 0.0
Camera.right.p1: 0.0
Camera.right.p2: 0.0
Camera.right.width: 1241
Camera.right.height: 376
T_lr: !!opencv-matrix # this statement is necessary.
  rows: 4
  cols: 4
  dt: f
  data: [1,0,0,0.5371657189, 0,1,0,0, 0,0,1,0, 0,0,0,1]
feature_tracker.thres_error: 80.0
feature_tracker.thres_bidirection: 0.5
feature_tracker.thres_sampson: 60.0
feature_tracker.window_size: 21
feature_tracker.max_level: 6
map_update.thres_parallax: 1.0
feature_extractor.n_features: 2000
feature_extractor.n_bins_u: 24
feature_extractor.n_bins_v: 12
feature_extractor.thres_fastscore: 20.0
feature_extractor.radius: 5.0
motion_estimator.thres_1p_error: 120.0   # pixels
motion_estimator.thres_5p_error: 1.0     # pixels
motion_estimator.thres_poseba_error: 3.0 # pixels
keyframe_update.thres_alive_ratio: 0.6
keyframe_update.thres_mean_parallax: 1.0
keyframe_update.thres_trans: 10.0 # meters
keyframe_update.thres_rotation: 15.0 # degrees
keyframe_update.n_max_keyframes_in_window: 9
"""


def test_stereo_config_of_a_file_in_the_references_format(tmp_path):
    p = tmp_path / "kitti_like.yaml"
    p.write_text(KITTI_LIKE)
    for src in (str(p), KITTI_LIKE):
        c = config.load_stereo_config(src)
        assert c["flagDoUndistortion"] == 0
        assert np.allclose(c["camera"]["left"]["K"], [718.856, 718.856, 607.1928, 185.2157])
        assert np.array_equal(c["camera"]["right"]["D"], np.zeros(5, np.float32))
        assert (c["camera"]["left"]["width"], c["camera"]["left"]["height"]) == (1241, 376)
        assert c["T_lr"].dtype == np.float32 and c["T_lr"].shape == (4, 4) and c["T_lr"][0, 3] == np.float32(0.5371657189)
        assert c["feature_tracker"] == dict(thres_error=80.0, thres_bidirection=0.5, thres_sampson=60.0, window_size=21, max_level=6)
        assert c["feature_extractor"]["n_bins_u"] == 24 and c["feature_extractor"]["thres_fastscore"] == 20.0
        assert c["motion_estimator"]["thres_poseba_error"] == 3.0
        assert c["keyframe_update"]["n_max_keyframes_in_window"] == 9 and c["keyframe_update"]["thres_rotation"] == 15.0


def test_distortion_order_and_missing_keys():
    t = KITTI_LIKE.replace("Camera.left.k1: 0.0", "Camera.left.k1: -0.1").replace("Camera.left.k3: 0.0", "Camera.left.k3: 0.3") \
        .replace("Camera.left.p2: 0.0", "Camera.left.p2: 0.02").replace("flagDoUndistortion: 0", "flagDoUndistortion: 1") \
        .replace("feature_tracker.max_level: 6\n", "")
    c = config.load_stereo_config(t)
    assert c["flagDoUndistortion"] == 1
    assert np.allclose(c["camera"]["left"]["D"], [-0.1, 0.0, 0.0, 0.02, 0.3])  # k1, k2, p1, p2, k3 (stereo_vo.cpp:158-163)
    assert c["feature_tracker"]["max_level"] == 0  # a missing FileNode converts to 0 in the reference


@pytest.mark.skipif(not os.path.isdir("/root/reference/config/stereo"), reason="the reference tree is not on this machine")
def test_every_stereo_config_of_the_reference_parses():
    files = sorted(glob.glob("/root/reference/config/stereo/*.yaml"))
    assert files
    for f in files:
        c = config.load_stereo_config(f)
        assert c["camera"]["left"]["width"] > 0 and c["T_lr"].shape == (4, 4), f
        assert c["feature_tracker"]["window_size"] > 0 and c["feature_extractor"]["n_bins_u"] > 0, f
        # (exp_stereo.yaml has no n_max_keyframes_in_window: 0, with which the reference pops an empty list at its first
        # keyframe, keyframes.cpp:186-191; vo_svo_create refuses a window below 1)
        assert 0 <= c["keyframe_update"]["n_max_keyframes_in_window"] <= 16, f


def _cpp_reader(tmp_path):
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libdir = os.path.join(root, "visual_odometry_ros_amd", "lib")
    if not os.path.exists(os.path.join(libdir, "libvo_hip.so")):
        pytest.skip("libvo_hip.so is not built")
    exe = str(tmp_path / "config_demo")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", root, os.path.join(root, "tests", "cpp", "config_demo.cpp"), "-o", exe,
                           "-L", libdir, "-lvo_hip", f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib", "-lamdhip64"])
    return exe


def _cpp_read(exe, path, mode=None):
    import subprocess
    r = subprocess.run([exe, path] + ([mode] if mode else []), capture_output=True, text=True)
    return r.returncode, dict(line.split(" ", 1) for line in r.stdout.strip().splitlines())


def _flat(c):
    out = {"flagDoUndistortion": c["flagDoUndistortion"], "width": c["camera"]["left"]["width"], "height": c["camera"]["left"]["height"]}
    for k in range(4):
        out[f"Kl{k}"], out[f"Kr{k}"] = c["camera"]["left"]["K"][k], c["camera"]["right"]["K"][k]
    for k in range(5):
        out[f"Dl{k}"], out[f"Dr{k}"] = c["camera"]["left"]["D"][k], c["camera"]["right"]["D"][k]
    for k in range(16):
        out[f"T{k}"] = c["T_lr"].reshape(-1)[k]
    for sec in ("feature_tracker", "feature_extractor", "motion_estimator"):
        out.update(c[sec])
    ku = c["keyframe_update"]
    out.update(thres_alive_ratio=ku["thres_alive_ratio"], thres_trans=ku["thres_trans"], thres_rotation=ku["thres_rotation"],
               n_max_keyframes_in_window=ku["n_max_keyframes_in_window"])
    return out


def test_cpp_reader_agrees_with_the_python_reader(tmp_path):
    """vo::loadStereoVOParams (core/visual_odometry/stereo_vo_config.h: the (mode, YAML) constructor of the adapter) reads
    the same numbers; the mode argument behaves as in stereo_vo.cpp:6-20."""
    exe = _cpp_reader(tmp_path)
    files = [str(tmp_path / "a.yaml"), str(tmp_path / "b.yaml")]
    open(files[0], "w").write(KITTI_LIKE)
    open(files[1], "w").write(KITTI_LIKE.replace("flagDoUndistortion: 0", "flagDoUndistortion: 1").replace("Camera.right.k1: 0.0", "Camera.right.k1: -0.31")
                              .replace("data: [1,0,0,0.5371657189, 0,1,0,0, 0,0,1,0, 0,0,0,1]",
                                       "data: [ 0.9999, 0.01, 0., 0.12,\n     -0.01, 0.9999, 0., 0.001,\n     0., 0., 1., -0.002, 0., 0., 0., 1. ]"))
    if os.path.isdir("/root/reference/config/stereo"):
        files += sorted(glob.glob("/root/reference/config/stereo/*.yaml"))
    for f in files:
        rc, got = _cpp_read(exe, f)
        assert rc == 0, (f, got)
        want = _flat(config.load_stereo_config(f))
        assert set(got) == set(want), f
        for k, v in want.items():
            assert np.float32(float(got[k])) == np.float32(v), (f, k, got[k], v)
    assert _cpp_read(exe, files[0], "rosbag")[0] == 0
    assert _cpp_read(exe, files[0], "dataset") == (2, {"error": "StereoVO - 'dataset' mode is not supported now..."})
    assert _cpp_read(exe, files[0], "live") == (2, {"error": "StereoVO - unknown mode..."})
    assert _cpp_read(exe, str(tmp_path / "missing.yaml"))[0] == 2
