"""visual_odometry_ros_amd — MI355X (gfx950) implementation of the per-frame
visual-odometry hot path of ChanghyeonKim93/visual_odometry_ros: pyramidal KLT,
scale-compensated IC patch refinement, ORB Hamming distance and the pose-only
Gauss-Newton motion estimator, as hand-written HIP kernels behind a C ABI
(include/vo_hip.h). This package is the Python host-side mirror of the
reference's FeatureTracker / MotionEstimator operator interface.
"""
from ._capi import VoError, load, LIB_PATH  # noqa: F401
from .api import (Context, FeatureTracker, MotionEstimator, FeatureExtractor,  # noqa: F401
                  StereoFramePipeline, MonoFramePipeline, Camera, StereoCamera,
                  SparseBundleAdjustmentSolver, TrackIds, se3Exp_f, write_trajectory, StereoVO, MonoVO, triangulateDLT, ImageSlots, StereoBatch)
