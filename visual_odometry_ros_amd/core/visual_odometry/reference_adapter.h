// reference_adapter.h — the reference's own class surface for the hot path, implemented on libvo_hip.so:
//   FeatureTracker    core/visual_odometry/feature_tracker.h:44-104   (all seven methods)
//   MotionEstimator   core/visual_odometry/motion_estimator.h:107-147 (pose-only BA pair, the epipolar distances,
//                                                                      setThres1p / setThres5p)
//   FeatureExtractor  core/visual_odometry/feature_extractor.h:136-160 (initParams, the weight-bin calls,
//                                                                      extractORBwithBinning_fast, descriptorDistance)
// with the reference's exact signatures on the reference's types (cv::Mat, cv::Point2f, Eigen::Matrix4f,
// std::vector<bool>, CameraConstPtr). It is meant to be compiled INSIDE the reference tree (include root = the
// reference repository root, core/CMakeLists.txt:15) in place of the three reference headers, so that
// stereo_vo.cpp / mono_vo.cpp and the ROS nodes compile and link unchanged against libvo_hip.so.
//
// Conversions made here and nowhere else:
//   * Eigen::Matrix4f / Matrix3f are column-major, the C ABI is row-major: to_row_major / from_row_major
//     transpose element by element (T(i,j) <-> a[4*i+j]).
//   * std::vector<bool> is bit-packed, the C ABI takes one byte per mask entry (done by the vo:: classes).
//   * cv::Point2f / Eigen::Vector3f have the layouts of vo::Pixel / vo::Point (static_asserts below); vectors are
//     copied (8 / 12 bytes per element), not reinterpret_cast.
// A context (HIP stream, pyramid slots, point buffers) belongs to ONE adapter object and is created at the first
// call from the sizes that call brings (image size, point count, pyramid depth); it is re-created larger if a later
// call needs more. Nothing is process-global: two VO instances in one process do not share device state.
//
// The image container has neither Eigen nor OpenCV, so this header is type-checked by tests/test_reference_adapter.py
// against minimal stand-in headers (tests/typecheck_stubs/, declared as such, used for nothing else); the first
// build against the real libraries happens on the integrator's machine.
#ifndef VO_AMD_REFERENCE_ADAPTER_H_
#define VO_AMD_REFERENCE_ADAPTER_H_

#include <cstdint>
#include <cstring>
#include <functional>
#include <memory>
#include <stdexcept>
#include <vector>

#include "eigen3/Eigen/Dense"
#include "opencv4/opencv2/core.hpp"

#include "core/defines/define_type.h"     // the reference's aliases: Pixel, Point, PixelVec, MaskVec, PoseSE3, ...
#include "core/visual_odometry/camera.h"  // Camera::fx() fy() cx() cy() Kinv()

#include "feature_extractor.h"
#include "feature_tracker.h"
#include "motion_estimator.h"
#include "mono_vo.h"
#include "mono_vo_config.h"
#include "stereo_vo.h"
#include "stereo_vo_config.h"

namespace vo_adapter {

static_assert(sizeof(cv::Point2f) == sizeof(vo::Pixel), "cv::Point2f must be two packed floats");
static_assert(sizeof(Eigen::Vector3f) == sizeof(vo::Point), "Eigen::Vector3f must be three packed floats");

// ---- layout conversions ---------------------------------------------------------------------------
inline vo::PoseSE3 to_row_major(const Eigen::Matrix4f &T) {
  vo::PoseSE3 o;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) o[(size_t)(i * 4 + j)] = T(i, j);
  return o;
}
inline void from_row_major(const vo::PoseSE3 &a, Eigen::Matrix4f &T) {
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) T(i, j) = a[(size_t)(i * 4 + j)];
}
inline vo::Rot3 to_row_major(const Eigen::Matrix3f &R) {
  vo::Rot3 o;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) o[(size_t)(i * 3 + j)] = R(i, j);
  return o;
}
inline void from_row_major(const vo::Rot3 &a, Eigen::Matrix3f &R) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) R(i, j) = a[(size_t)(i * 3 + j)];
}
inline vo::PixelVec to_vo(const PixelVec &v) {
  vo::PixelVec o(v.size());
  if (!v.empty()) std::memcpy(static_cast<void *>(o.data()), static_cast<const void *>(v.data()), sizeof(vo::Pixel) * v.size());
  return o;
}
inline void from_vo(const vo::PixelVec &v, PixelVec &o) {
  o.resize(v.size());
  if (!v.empty()) std::memcpy(static_cast<void *>(o.data()), static_cast<const void *>(v.data()), sizeof(vo::Pixel) * v.size());
}
inline vo::PointVec to_vo(const PointVec &v) {
  vo::PointVec o(v.size());
  for (size_t i = 0; i < v.size(); ++i) o[i] = vo::Point(v[i](0), v[i](1), v[i](2));
  return o;
}
// Image identity for the slot cache of the classes below (vo::FeatureTracker::bind): (buffer address, size, the frame
// stamp the caller set with setFrameStamp). With a stamp, the operator calls of one frame that take the same cv::Mat share
// one upload and one pyramid; without (stamp 0) every call uploads and rebuilds, as cv::calcOpticalFlowPyrLK does.
inline std::uint64_t image_id(const cv::Mat &m, std::uint64_t stamp) {
  if (!stamp) return 0;
  std::uint64_t h = 1469598103934665603ull;
  const std::uint64_t w[5] = {(std::uint64_t)(std::uintptr_t)m.data, (std::uint64_t)m.cols, (std::uint64_t)m.rows,
                              (std::uint64_t)m.step, stamp};
  for (std::uint64_t v : w)
    for (int b = 0; b < 8; ++b) {
      h ^= (v >> (8 * b)) & 0xffu;
      h *= 1099511628211ull;
    }
  return h ? h : 1;
}
inline vo::Image view(const cv::Mat &m, std::uint64_t stamp = 0) {
  if (m.type() != CV_8UC1) throw std::runtime_error("libvo_hip adapter: CV_8UC1 image expected");
  return vo::Image(m.data, m.cols, m.rows, (int)m.step, image_id(m, stamp));
}
inline vo::Camera intrinsics(CameraConstPtr &cam) { return vo::Camera{cam->fx(), cam->fy(), cam->cx(), cam->cy()}; }

// ---- one device context per adapter object, sized from what the calls bring --------------------------
class LazyContext {
 public:
  // returns true when the context was (re)created: objects built on the old one must be rebuilt
  bool ensure(int width, int height, int n_points, int max_level, int n_slots) {
    if (ctx_ && width <= w_ && height <= h_ && n_points <= n_ && max_level <= lvl_ && n_slots <= slots_) return false;
    w_ = width > w_ ? width : w_;
    h_ = height > h_ ? height : h_;
    if (n_points > n_) n_ = n_points + n_points / 2 + 256;  // head room: the track set of the next frame is rarely equal
    lvl_ = max_level > lvl_ ? max_level : lvl_;
    slots_ = n_slots > slots_ ? n_slots : slots_;
    ctx_ = std::make_shared<vo::Context>(device_, w_, h_, n_, slots_, lvl_);
    return true;
  }
  const vo::ContextPtr &get() const { return ctx_; }
  void setDevice(int device) { device_ = device; }

 private:
  vo::ContextPtr ctx_;
  int device_ = 0, w_ = 0, h_ = 0, n_ = 0, lvl_ = 0, slots_ = 0;
};

}  // namespace vo_adapter

// ===== FeatureTracker (core/visual_odometry/feature_tracker.h) =========================================
class FeatureTracker {
 public:
  FeatureTracker() {}
  ~FeatureTracker() {}

  // NOT in the reference: one call per frame (e.g. the frame counter + 1, at the top of trackStereoImages / trackImage)
  // lets the tracker calls of that frame share uploads and pyramids — 3 instead of 8 per stereo frame. 0 switches it off.
  void setFrameStamp(std::uint64_t stamp) { stamp_ = stamp; }

  void track(const cv::Mat &img0, const cv::Mat &img1, const PixelVec &pts0, int window_size, int max_pyr_lvl,
             float thres_err, PixelVec &pts_track, MaskVec &mask_valid) {
    vo::FeatureTracker &t = impl(img0, pts0.size(), max_pyr_lvl);
    vo::PixelVec out;  // cleared and refilled by the callee (feature_tracker.cpp:23-24)
    t.track(vo_adapter::view(img0, stamp_), vo_adapter::view(img1, stamp_), vo_adapter::to_vo(pts0), window_size, max_pyr_lvl, thres_err,
            out, mask_valid);
    vo_adapter::from_vo(out, pts_track);
  }
  void trackBidirection(const cv::Mat &img0, const cv::Mat &img1, const PixelVec &pts0, int window_size,
                        int max_pyr_lvl, float thres_err, float thres_bidirection, PixelVec &pts_track,
                        MaskVec &mask_valid) {
    vo::FeatureTracker &t = impl(img0, pts0.size(), max_pyr_lvl);
    vo::PixelVec out;
    t.trackBidirection(vo_adapter::view(img0, stamp_), vo_adapter::view(img1, stamp_), vo_adapter::to_vo(pts0), window_size, max_pyr_lvl,
                       thres_err, thres_bidirection, out, mask_valid);
    vo_adapter::from_vo(out, pts_track);
  }
  void trackBidirectionWithPrior(const cv::Mat &img0, const cv::Mat &img1, const PixelVec &pts0, int window_size,
                                 int max_pyr_lvl, float thres_err, float thres_bidirection, PixelVec &pts_track,
                                 MaskVec &mask_valid) {
    vo::FeatureTracker &t = impl(img0, pts0.size(), max_pyr_lvl);
    vo::PixelVec io = vo_adapter::to_vo(pts_track);  // carries the prior in (feature_tracker.cpp:88-169)
    t.trackBidirectionWithPrior(vo_adapter::view(img0, stamp_), vo_adapter::view(img1, stamp_), vo_adapter::to_vo(pts0), window_size,
                                max_pyr_lvl, thres_err, thres_bidirection, io, mask_valid);
    vo_adapter::from_vo(io, pts_track);
  }
  void trackWithPrior(const cv::Mat &img0, const cv::Mat &img1, const PixelVec &pts0, int window_size,
                      int max_pyr_lvl, float thres_err, PixelVec &pts_track, MaskVec &mask_valid) {
    vo::FeatureTracker &t = impl(img0, pts0.size(), max_pyr_lvl);
    vo::PixelVec io = vo_adapter::to_vo(pts_track);
    t.trackWithPrior(vo_adapter::view(img0, stamp_), vo_adapter::view(img1, stamp_), vo_adapter::to_vo(pts0), window_size, max_pyr_lvl,
                     thres_err, io, mask_valid);
    vo_adapter::from_vo(io, pts_track);
  }
  void calcPrior(const PixelVec &pts0, const PointVec &Xw, const PoseSE3 &Tw1, const Eigen::Matrix3f &K,
                 PixelVec &pts1_prior) {
    // (a recreated context invalidates the cached tracker, which is bound to the old one)
    if (lazy_.ensure(64, 64, (int)(pts0.size() > Xw.size() ? pts0.size() : Xw.size()), 1, 4)) impl_.reset();
    vo::FeatureTracker t(lazy_.get());
    vo::PixelVec out;
    t.calcPrior(vo_adapter::to_vo(pts0), vo_adapter::to_vo(Xw), vo_adapter::to_row_major(Tw1), vo_adapter::to_row_major(K),
                out);
    vo_adapter::from_vo(out, pts1_prior);
  }
  // du0 / dv0 (cv::Sobel of img0, stereo_vo.cpp:549-552) are recomputed on the device from img0; the arguments are
  // accepted for source compatibility and not read.
  void trackWithScale(const cv::Mat &img0, const cv::Mat &du0, const cv::Mat &dv0, const cv::Mat &img1,
                      const PixelVec &pts0, const std::vector<float> &scale_est, PixelVec &pts_track,
                      MaskVec &mask_valid) {
    // the derivative images must at least be what the drivers pass: cv::Sobel of img0, i.e. img0's size
    if (du0.rows != img0.rows || du0.cols != img0.cols || dv0.rows != img0.rows || dv0.cols != img0.cols)
      throw std::runtime_error("trackWithScale: du0 / dv0 are not derivative images of img0 (size mismatch)");
    vo::FeatureTracker &t = impl(img0, pts0.size(), 0);
    vo::PixelVec io = vo_adapter::to_vo(pts_track);
    t.trackWithScale(vo_adapter::view(img0, stamp_), vo_adapter::view(img1, stamp_), vo_adapter::to_vo(pts0), scale_est, io, mask_valid);
    vo_adapter::from_vo(io, pts_track);
  }

 private:
  vo::FeatureTracker &impl(const cv::Mat &img, size_t n, int max_pyr_lvl) {
    if (lazy_.ensure(img.cols, img.rows, (int)n, max_pyr_lvl, 4) || !impl_)
      impl_.reset(new vo::FeatureTracker(lazy_.get()));
    return *impl_;
  }
  vo_adapter::LazyContext lazy_;
  std::unique_ptr<vo::FeatureTracker> impl_;
  std::uint64_t stamp_ = 0;
};

// ===== MotionEstimator (core/visual_odometry/motion_estimator.h) — the pose-only BA and epipolar members =====
// (calcPose5PointsAlgorithm / calcPosePnPAlgorithm / findInliers1PointHistogram / the local-BA drivers are OpenCV
// calib3d and landmark-graph code outside the hot path: keep the reference's own definitions for those.)
class MotionEstimator {
 public:
  MotionEstimator(bool is_stereo_mode = false, const PoseSE3 &T_lr = PoseSE3::Identity())
      : is_stereo_mode_(is_stereo_mode), T_lr_(T_lr) {}
  ~MotionEstimator() {}

  bool poseOnlyBundleAdjustment(const PointVec &X, const PixelVec &pts1, CameraConstPtr &cam,
                                const int &thres_reproj_outlier, Rot3 &R01_true, Pos3 &t01_true, MaskVec &mask_inlier) {
    vo::MotionEstimator &m = impl(X.size());
    vo::Rot3 R = vo_adapter::to_row_major(R01_true);
    vo::Pos3 t = {t01_true(0), t01_true(1), t01_true(2)};
    const bool ok = m.poseOnlyBundleAdjustment(vo_adapter::to_vo(X), vo_adapter::to_vo(pts1), vo_adapter::intrinsics(cam),
                                               thres_reproj_outlier, R, t, mask_inlier);
    if (ok) {  // on NaN the reference leaves R01 / t01 alone (motion_estimator.cpp:846-857)
      vo_adapter::from_row_major(R, R01_true);
      for (int i = 0; i < 3; ++i) t01_true(i) = t[(size_t)i];
    }
    return ok;
  }
  bool poseOnlyBundleAdjustment_Stereo(const PointVec &X, const PixelVec &pts_l1, const PixelVec &pts_r1,
                                       CameraConstPtr &cam_left, CameraConstPtr &cam_right, const PoseSE3 &T_lr,
                                       float thres_reproj_outlier, PoseSE3 &T01, MaskVec &mask_inlier) {
    vo::MotionEstimator &m = impl(X.size());
    vo::PoseSE3 T = vo_adapter::to_row_major(T01);
    const bool ok = m.poseOnlyBundleAdjustment_Stereo(vo_adapter::to_vo(X), vo_adapter::to_vo(pts_l1),
                                                      vo_adapter::to_vo(pts_r1), vo_adapter::intrinsics(cam_left),
                                                      vo_adapter::intrinsics(cam_right), vo_adapter::to_row_major(T_lr),
                                                      thres_reproj_outlier, T, mask_inlier);
    if (ok) vo_adapter::from_row_major(T, T01);  // NaN: T01 untouched (motion_estimator.cpp:1070-1084)
    return ok;
  }

  // motion_estimator.cpp:538-570: F10 = Kinv^T [t10]x R10 Kinv, evaluated by Eigen exactly as the reference writes it
  void calcSampsonDistance(const PixelVec &pts0, const PixelVec &pts1, CameraConstPtr &cam, const Rot3 &R10,
                           const Pos3 &t10, std::vector<float> &sampson_dist) {
    calcSampsonDistance(pts0, pts1, fundamental(cam, R10, t10), sampson_dist);
  }
  void calcSampsonDistance(const PixelVec &pts0, const PixelVec &pts1, const Mat33 &F10, std::vector<float> &sampson_dist) {
    impl(pts0.size()).calcSampsonDistance(vo_adapter::to_vo(pts0), vo_adapter::to_vo(pts1), vo_adapter::to_row_major(F10),
                                          sampson_dist);
  }
  float calcSampsonDistance(const Pixel &pt0, const Pixel &pt1, const Mat33 &F10) {
    std::vector<float> d;
    calcSampsonDistance(PixelVec(1, pt0), PixelVec(1, pt1), F10, d);
    return d[0];
  }
  void calcSymmetricEpipolarDistance(const PixelVec &pts0, const PixelVec &pts1, CameraConstPtr &cam, const Rot3 &R10,
                                     const Pos3 &t10, std::vector<float> &sym_epi_dist) {
    impl(pts0.size()).calcSymmetricEpipolarDistance(vo_adapter::to_vo(pts0), vo_adapter::to_vo(pts1),
                                                    vo_adapter::to_row_major(fundamental(cam, R10, t10)), sym_epi_dist);
  }
  void setThres1p(float thres_1p) { thres_1p_ = thres_1p; }
  void setThres5p(float thres_5p) { thres_5p_ = thres_5p; }

 private:
  static Mat33 fundamental(CameraConstPtr &cam, const Rot3 &R10, const Pos3 &t10) {
    Mat33 S;  // mapping::skew(t10)
    S(0, 0) = 0.0f;     S(0, 1) = -t10(2);  S(0, 2) = t10(1);
    S(1, 0) = t10(2);   S(1, 1) = 0.0f;     S(1, 2) = -t10(0);
    S(2, 0) = -t10(1);  S(2, 1) = t10(0);   S(2, 2) = 0.0f;
    Mat33 E10, F10;
    E10 = S * R10;
    F10 = cam->Kinv().transpose() * E10 * cam->Kinv();
    return F10;
  }
  vo::MotionEstimator &impl(size_t n) {
    if (lazy_.ensure(64, 64, (int)n, 1, 2) || !impl_)
      impl_.reset(new vo::MotionEstimator(lazy_.get(), is_stereo_mode_, vo_adapter::to_row_major(T_lr_)));
    return *impl_;
  }
  bool is_stereo_mode_;
  PoseSE3 T_lr_;
  float thres_1p_ = 0.f, thres_5p_ = 0.f;
  vo_adapter::LazyContext lazy_;
  std::unique_ptr<vo::MotionEstimator> impl_;
};

// ===== FeatureExtractor (core/visual_odometry/feature_extractor.h) ======================================
// cv::ORB::detect + the per-bucket arg-max run on the device; extractAndComputeORB / extractORBwithBinning (the
// non-"_fast" variants, descriptors) are not on the frame path and are not provided.
class FeatureExtractor {
 public:
  FeatureExtractor() {}
  ~FeatureExtractor() {}

  void initParams(int n_cols, int n_rows, int n_bins_u, int n_bins_v, int THRES_FAST, int radius) {
    lazy_.ensure(n_cols, n_rows, n_bins_u * n_bins_v + 64, 1, 2);
    impl_.reset(new vo::FeatureExtractor(lazy_.get()));
    impl_->initParams(n_cols, n_rows, n_bins_u, n_bins_v, THRES_FAST, radius);
    impl_->setOrbParams(THRES_FAST);  // setMaxFeatures(10000) ... setFastThreshold(THRES_FAST), feature_extractor.cpp:48-56
    flag_nonmax_ = true;              // :37
  }
  void updateWeightBin(const PixelVec &pts) { need().updateWeightBin(vo_adapter::to_vo(pts)); }
  void resetWeightBin() { need().resetWeightBin(); }
  void suppressCenterBins() { need().suppressCenterBins(); }
  void setNonmaxSuppression(bool flag_on) { flag_nonmax_ = flag_on; }
  void extractORBwithBinning_fast(const cv::Mat &img, PixelVec &pts_extracted, bool flag_nonmax) {
    // the reference reads the member flag_nonmax_ here, not the argument (feature_extractor.cpp:244); initParams
    // sets it to true and both drivers pass true
    (void)flag_nonmax;
    if (!flag_nonmax_) throw std::runtime_error("libvo_hip adapter: extractORBwithBinning_fast without non-max bucketing is not provided");
    vo::FeatureExtractor &e = need();
    const vo::Image im = vo_adapter::view(img);
    lazy_.get()->check(vo_set_image(lazy_.get()->get(), 0, im.data, im.width, im.height, im.stride));
    vo::PixelVec out;
    e.extractORBwithBinning_fast(0, out);
    vo_adapter::from_vo(out, pts_extracted);
  }
  int descriptorDistance(const cv::Mat &a, const cv::Mat &b) { return need().descriptorDistance(a.data, b.data); }

 private:
  vo::FeatureExtractor &need() {
    if (!impl_) throw std::runtime_error("FeatureExtractor: initParams was not called");
    return *impl_;
  }
  bool flag_nonmax_ = true;
  vo_adapter::LazyContext lazy_;
  std::unique_ptr<vo::FeatureExtractor> impl_;
};


// ===== StereoVO (core/visual_odometry/stereo_vo/stereo_vo.h:233-249) =====================================================
// trackStereoImages(const cv::Mat&, const cv::Mat&, const double&) and getStatistics() with the reference's signatures;
// the whole frame — track set carried from frame to frame, new landmarks, keyframes, local BA — runs in libvo_hip.so
// with the track set on the device (vo::StereoVO, stereo_vo.h next to this file). The reference's constructor
// (mode, YAML directory) loads a cv::FileStorage: that stays the caller's, the parameters arrive as vo::StereoVOParams
// (the YAML's numbers). getDebugImage() returns an empty image (no GUI on this path).
class StereoVO {
 public:
  struct AlgorithmStatistics {  // the members the ROS nodes read (ros*/visual_odometry/stereo_vo_ros*.cpp)
    struct FrameStatistics {
      PoseSE3 Twc, Tcw, dT_01, dT_10;
      PointVec mappoints;
    };
    using LandmarkStatistics = vo::StereoVO::AlgorithmStatistics::LandmarkStatistics;
    using ExecutionStatistics = vo::StereoVO::AlgorithmStatistics::ExecutionStatistics;
    struct KeyframeStatistics {  // stereo_vo_ros2.cpp:141-166: keyframe trajectory and map-point cloud
      PoseSE3 Twc;
      PointVec mappoints;
    };
    std::vector<LandmarkStatistics> stats_landmark;
    std::vector<FrameStatistics> stats_frame;
    std::vector<KeyframeStatistics> stats_keyframe;
    std::vector<ExecutionStatistics> stats_execution;
  };

  // the reference's constructor (stereo_vo.h:233, stereo_vo.cpp:6-57): mode "rosbag" + the path of a config/stereo/*.yaml
  // file; stats_keyframe is rewritten at every keyframe, as there
  StereoVO(std::string mode, std::string directory_intrinsic) : StereoVO(with_statistics(vo::stereoVOParamsForMode(mode, directory_intrinsic))) {}
  // (vo::StereoVOParams::keyframe_statistics = true gives the reference's behaviour: stats_keyframe rewritten at every keyframe)
  explicit StereoVO(const vo::StereoVOParams &p, int device = 0)
      : ctx_(std::make_shared<vo::Context>(device, p.width, p.height,
                                           2 * p.feature_extractor.n_bins_u * p.feature_extractor.n_bins_v + 1024, 5,
                                           p.feature_tracker.max_level)),
        impl_(ctx_, p) {}
  ~StereoVO() noexcept(false) {}

  void trackStereoImages(const cv::Mat &img_left, const cv::Mat &img_right, const double &timestamp) {
    impl_.trackStereoImages(vo_adapter::view(img_left), vo_adapter::view(img_right), timestamp);
    const auto &s = impl_.getStatistics();
    AlgorithmStatistics::FrameStatistics f;
    vo_adapter::from_row_major(s.stats_frame.back().Twc, f.Twc);
    vo_adapter::from_row_major(s.stats_frame.back().Tcw, f.Tcw);
    vo_adapter::from_row_major(s.stats_frame.back().dT_01, f.dT_01);
    vo_adapter::from_row_major(s.stats_frame.back().dT_10, f.dT_10);
    stat_.stats_frame.push_back(f);
    stat_.stats_landmark.push_back(s.stats_landmark.back());
    stat_.stats_execution.push_back(s.stats_execution.back());
    if (impl_.lastFrameInfo().is_keyframe && !s.stats_keyframe.empty()) {  // (filled by the implementation when asked for)
      stat_.stats_keyframe.resize(s.stats_keyframe.size());
      for (size_t j = 0; j < s.stats_keyframe.size(); ++j) {
        vo_adapter::from_row_major(s.stats_keyframe[j].Twc, stat_.stats_keyframe[j].Twc);
        stat_.stats_keyframe[j].mappoints.resize(s.stats_keyframe[j].mappoints.size());
        for (size_t i = 0; i < s.stats_keyframe[j].mappoints.size(); ++i)
          stat_.stats_keyframe[j].mappoints[i] = Point(s.stats_keyframe[j].mappoints[i].x, s.stats_keyframe[j].mappoints[i].y,
                                                       s.stats_keyframe[j].mappoints[i].z);
      }
    }
  }
  const AlgorithmStatistics &getStatistics() const { return stat_; }
  const cv::Mat &getDebugImage() { return img_debug_; }
  vo::StereoVO &device() { return impl_; }

 private:
  static vo::StereoVOParams with_statistics(vo::StereoVOParams p) {
    p.keyframe_statistics = true;
    return p;
  }
  vo::ContextPtr ctx_;
  vo::StereoVO impl_;
  AlgorithmStatistics stat_;
  cv::Mat img_debug_;
};


// ===== MonoVO (core/visual_odometry/mono_vo/mono_vo.h:235-243, :267) =====================================================
// MonoVO(mode, directory_intrinsic), trackImage(const cv::Mat&, const double&), getStatistics(), getDebugImage() with the
// reference's signatures; the loop runs in libvo_hip.so (vo::MonoVO, mono_vo.h next to this file). The 5-point pose is the
// one piece the library does not contain (OpenCV calib3d): bind MonoVO::setFivePointSolver to the reference's
// own MotionEstimator::calcPose5PointsAlgorithm (motion_estimator.cpp:21-203 + findCorrectRT, kept from the reference tree)
// before the first trackImage — INTEGRATION.md shows the three lines. getDebugImage() returns an empty image.
class MonoVO {
 public:
  struct AlgorithmStatistics {  // the members the ROS node reads (ros1/visual_odometry/mono_vo_ros1.cpp:123-190)
    struct FrameStatistics {
      PoseSE3 Twc, Tcw, dT_01, dT_10;
      PointVec mappoints;
    };
    using LandmarkStatistics = vo::MonoVO::AlgorithmStatistics::LandmarkStatistics;
    using ExecutionStatistics = vo::MonoVO::AlgorithmStatistics::ExecutionStatistics;
    struct KeyframeStatistics {
      PoseSE3 Twc;
      PointVec mappoints;
    };
    std::vector<LandmarkStatistics> stats_landmark;
    std::vector<FrameStatistics> stats_frame;
    std::vector<KeyframeStatistics> stats_keyframe;
    std::vector<ExecutionStatistics> stats_execution;
  };
  // calcPose5PointsAlgorithm(pts0, pts1, cam, R10, t10, X0, mask) on the reference's types (X0 is unused by trackImage)
  using FivePointSolver = std::function<bool(const PixelVec &pts0, const PixelVec &pts1, const Eigen::Matrix3f &K, Rot3 &R10, Pos3 &t10,
                                             MaskVec &mask_inlier)>;

  MonoVO(std::string mode, std::string directory_intrinsic) : MonoVO(with_statistics(vo::monoVOParamsForMode(mode, directory_intrinsic))) {}
  explicit MonoVO(const vo::MonoVOParams &p, int device = 0)
      : ctx_(std::make_shared<vo::Context>(device, p.width, p.height,
                                           2 * p.feature_extractor.n_bins_u * p.feature_extractor.n_bins_v + 1024, 3,
                                           p.feature_tracker.max_level)),
        impl_(ctx_, p, [this](const vo::PixelVec &a, const vo::PixelVec &b, const float K[4], float R10[9], float t10[3],
                              std::vector<std::uint8_t> &mask) { return this->five_point(a, b, K, R10, t10, mask); }) {}
  ~MonoVO() noexcept(false) {}

  void setFivePointSolver(FivePointSolver f) { solver_ = std::move(f); }

  void trackImage(const cv::Mat &img, const double &timestamp) {
    impl_.trackImage(vo_adapter::view(img), timestamp);
    const auto &s = impl_.getStatistics();
    AlgorithmStatistics::FrameStatistics f;
    vo_adapter::from_row_major(s.stats_frame.back().Twc, f.Twc);
    vo_adapter::from_row_major(s.stats_frame.back().Tcw, f.Tcw);
    vo_adapter::from_row_major(s.stats_frame.back().dT_01, f.dT_01);
    vo_adapter::from_row_major(s.stats_frame.back().dT_10, f.dT_10);
    stat_.stats_frame.push_back(f);
    stat_.stats_landmark.push_back(s.stats_landmark.back());
    stat_.stats_execution.push_back(s.stats_execution.back());
    if (impl_.lastFrameInfo().is_keyframe && !s.stats_keyframe.empty()) {
      stat_.stats_keyframe.resize(s.stats_keyframe.size());
      for (size_t j = 0; j < s.stats_keyframe.size(); ++j) {
        vo_adapter::from_row_major(s.stats_keyframe[j].Twc, stat_.stats_keyframe[j].Twc);
        stat_.stats_keyframe[j].mappoints.resize(s.stats_keyframe[j].mappoints.size());
        for (size_t i = 0; i < s.stats_keyframe[j].mappoints.size(); ++i)
          stat_.stats_keyframe[j].mappoints[i] = Point(s.stats_keyframe[j].mappoints[i].x, s.stats_keyframe[j].mappoints[i].y,
                                                       s.stats_keyframe[j].mappoints[i].z);
      }
    }
  }
  const AlgorithmStatistics &getStatistics() const { return stat_; }
  const cv::Mat &getDebugImage() { return img_debug_; }
  vo::MonoVO &device() { return impl_; }

 private:
  static vo::MonoVOParams with_statistics(vo::MonoVOParams p) {
    p.keyframe_statistics = true;
    return p;
  }
  bool five_point(const vo::PixelVec &a, const vo::PixelVec &b, const float K[4], float R10[9], float t10[3], std::vector<std::uint8_t> &mask) {
    if (!solver_) throw std::runtime_error("MonoVO: setFivePointSolver was not called (calcPose5PointsAlgorithm is the caller's)");
    PixelVec p0, p1;
    vo_adapter::from_vo(a, p0);
    vo_adapter::from_vo(b, p1);
    Eigen::Matrix3f Km;  // (element-wise: the type-check stand-in for Eigen has no comma initialiser)
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) Km(i, j) = 0.0f;
    Km(0, 0) = K[0];
    Km(0, 2) = K[2];
    Km(1, 1) = K[1];
    Km(1, 2) = K[3];
    Km(2, 2) = 1.0f;
    Rot3 R;
    Pos3 t;
    MaskVec m(a.size(), true);
    if (!solver_(p0, p1, Km, R, t, m)) return false;
    for (int i = 0; i < 3; ++i) {
      for (int j = 0; j < 3; ++j) R10[i * 3 + j] = R(i, j);
      t10[i] = t(i);
    }
    for (size_t i = 0; i < mask.size() && i < m.size(); ++i) mask[i] = m[i] ? 1 : 0;
    return true;
  }
  vo::ContextPtr ctx_;
  vo::MonoVO impl_;
  FivePointSolver solver_;
  AlgorithmStatistics stat_;
  cv::Mat img_debug_;
};

// ===== Landmark::setPatch (core/visual_odometry/landmark.h:67-86) =========================================================
// The ROS 1 nodes call it once before they construct the VO object (ros1/visual_odometry/stereo_vo_ros1.cpp:41,
// mono_vo_ros1.cpp:49). In the reference the pattern it fills is read at one place only — the first observation of a
// landmark copies it into a local vector that is dropped again (landmark.cpp:88-98) — so it has no effect on any result;
// the call is accepted and the pattern kept, for source compatibility.
class Landmark {
 public:
  inline static PixelVec patt_ = PixelVec();
  static void setPatch(int half_win_sz) {
    const int win_sz = 2 * half_win_sz + 1;
    patt_.clear();
    for (int v = 0; v < win_sz; ++v)
      for (int u = !(v & 0x01); u < win_sz; u += 2) patt_.push_back(Pixel((float)(u - half_win_sz), (float)(v - half_win_sz)));
  }
};

#endif  // VO_AMD_REFERENCE_ADAPTER_H_
