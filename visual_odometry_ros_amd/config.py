"""The reference's YAML configuration files (config/stereo/*.yaml) as StereoVO reads them:
StereoVO::loadStereoCameraIntrinsicAndUserParameters (core/visual_odometry/stereo_vo/stereo_vo.cpp:118-280) walks a
cv::FileStorage — OpenCV's YAML 1.0 dialect: a `%YAML:1.0` directive, flat `a.b.c: value` keys, `!!opencv-matrix`
nodes (rows, cols, dt, data). This module reads the same files with PyYAML and hands the numbers to
visual_odometry_ros_amd.StereoVO (StereoVO.from_yaml), so that a user of the reference keeps their configuration.

Keys (all required, as `fs["..."]` of a missing key yields 0 in the reference): flagDoUndistortion,
Camera.{left,right}.{fx,fy,cx,cy,k1,k2,k3,p1,p2,width,height}, T_lr (4x4, dt f), feature_tracker.{thres_error,
thres_bidirection,thres_sampson,window_size,max_level}, map_update.thres_parallax, feature_extractor.{n_features,n_bins_u,
n_bins_v,thres_fastscore,radius}, motion_estimator.{thres_1p_error,thres_5p_error,thres_poseba_error},
keyframe_update.{thres_alive_ratio,thres_mean_parallax,thres_trans,thres_rotation,n_max_keyframes_in_window}."""
import numpy as np
import yaml


class _Loader(yaml.SafeLoader):
    pass


def _opencv_matrix(loader, node):
    m = loader.construct_mapping(node, deep=True)
    dt = {"f": np.float32, "d": np.float64, "i": np.int32, "u": np.uint8}.get(str(m.get("dt", "f")), np.float32)
    a = np.array(m["data"], dtype=dt)
    return a.reshape(int(m["rows"]), int(m["cols"]))


_Loader.add_constructor("tag:yaml.org,2002:opencv-matrix", _opencv_matrix)


def load_yaml(path_or_text):
    """An OpenCV FileStorage YAML file (or its text) as a dict; matrices become numpy arrays."""
    text = path_or_text
    if "\n" not in path_or_text:
        with open(path_or_text) as f:
            text = f.read()
    lines = text.splitlines()
    if lines and lines[0].startswith("%YAML"):  # "%YAML:1.0" is OpenCV's spelling, not a directive PyYAML accepts
        lines = lines[1:]
    d = yaml.load("\n".join(lines), Loader=_Loader)
    return d or {}


def _num(d, key, kind=float):
    v = d.get(key, 0)  # (cv::FileNode of a missing key converts to 0)
    return kind(v)


def load_stereo_config(path_or_text):
    """The numbers of a config/stereo/*.yaml file, named as StereoVO::AlgorithmParameters names them
    (stereo_vo.h:57-103), cameras as (fx, fy, cx, cy) + (k1, k2, p1, p2, k3) — the order StereoCamera::initParams feeds to
    OpenCV (stereo_vo.cpp:147-160)."""
    d = load_yaml(path_or_text)
    cam = {}
    for side in ("left", "right"):
        p = f"Camera.{side}."
        cam[side] = dict(
            K=np.array([_num(d, p + "fx"), _num(d, p + "fy"), _num(d, p + "cx"), _num(d, p + "cy")], np.float32),
            D=np.array([_num(d, p + "k1"), _num(d, p + "k2"), _num(d, p + "p1"), _num(d, p + "p2"), _num(d, p + "k3")], np.float32),
            width=_num(d, p + "width", int), height=_num(d, p + "height", int))
    T_lr = np.asarray(d.get("T_lr", np.eye(4)), np.float32).reshape(4, 4)
    return dict(
        flagDoUndistortion=_num(d, "flagDoUndistortion", int),
        camera=cam, T_lr=T_lr,
        feature_tracker=dict(thres_error=_num(d, "feature_tracker.thres_error"), thres_bidirection=_num(d, "feature_tracker.thres_bidirection"),
                             thres_sampson=_num(d, "feature_tracker.thres_sampson"), window_size=_num(d, "feature_tracker.window_size", int),
                             max_level=_num(d, "feature_tracker.max_level", int)),
        map_update=dict(thres_parallax=_num(d, "map_update.thres_parallax")),
        feature_extractor=dict(n_features=_num(d, "feature_extractor.n_features", int), n_bins_u=_num(d, "feature_extractor.n_bins_u", int),
                               n_bins_v=_num(d, "feature_extractor.n_bins_v", int), thres_fastscore=_num(d, "feature_extractor.thres_fastscore"),
                               radius=_num(d, "feature_extractor.radius")),
        motion_estimator=dict(thres_1p_error=_num(d, "motion_estimator.thres_1p_error"), thres_5p_error=_num(d, "motion_estimator.thres_5p_error"),
                              thres_poseba_error=_num(d, "motion_estimator.thres_poseba_error")),
        keyframe_update=dict(thres_alive_ratio=_num(d, "keyframe_update.thres_alive_ratio"),
                             thres_mean_parallax=_num(d, "keyframe_update.thres_mean_parallax"),
                             thres_trans=_num(d, "keyframe_update.thres_trans"), thres_rotation=_num(d, "keyframe_update.thres_rotation"),
                             n_max_keyframes_in_window=_num(d, "keyframe_update.n_max_keyframes_in_window", int)),
    )
