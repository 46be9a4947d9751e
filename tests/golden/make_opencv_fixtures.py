#!/usr/bin/env python3
"""Pins the CPU oracle against the real OpenCV — WHERE OpenCV EXISTS. The build container has neither OpenCV nor
Eigen, so the oracle's restatements of cv::pyrDown, cv::Sobel, cv::calcOpticalFlowPyrLK, cv::remap and
cv::ORB::detect are "parity unpinned" here. This script runs the third-party functions the reference calls
(core/visual_odometry/feature_tracker.cpp:29,60,69,108,117,186; stereo_vo.cpp:551-552; camera.cpp:166-183, :300-336;
feature_extractor.cpp:48-56, :241) with the reference's arguments on the small committed images of
tests/golden/klt_small.npz and writes their outputs to tests/golden/opencv_fixtures.npz:

    python tests/golden/make_opencv_fixtures.py          # needs `import cv2` (OpenCV 4.x)

tests/test_opencv_pin.py then compares the oracle with that file (or with a live cv2), and is SKIPPED when neither is
there. Committing the produced .npz on a machine that has OpenCV turns "unpinned" into "pinned" for these five
functions. Nothing here reads /root/reference.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "opencv_fixtures.npz")


def inputs():
    """Deterministic inputs derived from the committed klt_small.npz (images, points) — no generator that could drift."""
    k = np.load(os.path.join(HERE, "klt_small.npz"))
    img0, img1, pts0 = k["img0"], k["img1"], k["pts0"]
    h, w = img0.shape
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    # a smooth warp that leaves the image on two sides, with exact ties of the 1/32 quantisation on a few pixels
    map_u = (xx + np.float32(3.0) * np.sin(yy / np.float32(23.0)) + np.float32(1.25)).astype(np.float32)
    map_v = (yy * np.float32(1.01) - np.float32(2.0) + np.float32(0.015625) * (xx % 4)).astype(np.float32)
    return dict(img0=img0, img1=img1, pts0=pts0, map_u=map_u, map_v=map_v)


def run_opencv(d):
    import cv2
    out = {"cv_version": np.array(cv2.__version__)}
    out["pyr_down"] = cv2.pyrDown(d["img0"])
    out["sobel_x"] = cv2.Sobel(d["img0"], cv2.CV_32F, 1, 0, ksize=3, scale=1.0, delta=0.0, borderType=cv2.BORDER_DEFAULT)
    out["sobel_y"] = cv2.Sobel(d["img0"], cv2.CV_32F, 0, 1, ksize=3, scale=1.0, delta=0.0, borderType=cv2.BORDER_DEFAULT)
    crit = (cv2.TERM_CRITERIA_COUNT + cv2.TERM_CRITERIA_EPS, 30, 0.01)
    for name, kw in (("lk_default", dict(flags=0, minEigThreshold=1e-4)),):
        p1, st, err = cv2.calcOpticalFlowPyrLK(d["img0"], d["img1"], d["pts0"].reshape(-1, 1, 2), None, winSize=(21, 21),
                                               maxLevel=3, criteria=crit, **kw)
        out[name + "_pts"], out[name + "_status"], out[name + "_err"] = p1.reshape(-1, 2), st.reshape(-1), err.reshape(-1)
    # `{}` criteria / minEigThreshold of the *WithPrior calls: TermCriteria() -> (COUNT+EPS, 30, 0.01) by OpenCV's own
    # defaulting inside calcOpticalFlowPyrLK, minEig 0; initial flow = the default result perturbed
    init = (out["lk_default_pts"] + np.float32(0.75)).astype(np.float32)
    p1, st, err = cv2.calcOpticalFlowPyrLK(d["img0"], d["img1"], d["pts0"].reshape(-1, 1, 2), init.reshape(-1, 1, 2).copy(),
                                           winSize=(21, 21), maxLevel=3, criteria=(0, 0, 0.0),
                                           flags=cv2.OPTFLOW_USE_INITIAL_FLOW, minEigThreshold=0.0)
    out["lk_prior_init"], out["lk_prior_pts"] = init, p1.reshape(-1, 2)
    out["lk_prior_status"], out["lk_prior_err"] = st.reshape(-1), err.reshape(-1)
    # Camera::undistortImage: convertTo(CV_32FC1), remap(INTER_LINEAR, BORDER_CONSTANT 0), driver's convertTo(CV_8UC1)
    src = d["img0"].astype(np.float32)
    rm = cv2.remap(src, d["map_u"], d["map_v"], cv2.INTER_LINEAR, borderMode=cv2.BORDER_CONSTANT, borderValue=0)
    out["remap_u8"] = np.clip(np.rint(rm), 0, 255).astype(np.uint8)  # cv::Mat::convertTo rounds half to even: np.rint too
    # FeatureExtractor::initParams (feature_extractor.cpp:48-56) + detect (:241)
    orb = cv2.ORB_create(nfeatures=10000, scaleFactor=1.2, nlevels=8, edgeThreshold=31, firstLevel=0, WTA_K=2,
                         scoreType=cv2.ORB_HARRIS_SCORE, patchSize=31, fastThreshold=15)
    kps = orb.detect(d["img0"], None)
    out["orb_xy"] = np.array([k.pt for k in kps], np.float32).reshape(-1, 2)
    out["orb_response"] = np.array([k.response for k in kps], np.float32)
    out["orb_octave"] = np.array([k.octave for k in kps], np.int32)
    return out


def main():
    try:
        import cv2  # noqa: F401
    except ImportError:
        sys.exit("make_opencv_fixtures.py: `import cv2` failed — run this where OpenCV 4.x is installed")
    d = inputs()
    out = run_opencv(d)
    np.savez_compressed(OUT, **out)
    print(f"wrote {OUT} (OpenCV {out['cv_version']}): {len(out['orb_xy'])} ORB keypoints, {int(out['lk_default_status'].sum())} "
          f"of {len(out['lk_default_status'])} points tracked")


if __name__ == "__main__":
    main()
