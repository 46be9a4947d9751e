// reference_adapter.h — the reference's own FeatureTracker / MotionEstimator class
// surface (cv::Mat, cv::Point2f, Eigen::Matrix4f, std::vector<bool>; signatures of
// core/visual_odometry/feature_tracker.h:44-104 and motion_estimator.h:117-120),
// implemented on libvo_hip.so. It is compiled ONLY where OpenCV 4 and Eigen 3
// headers exist (they do not in the build container, so this file is type-checked
// nowhere here). Dropping this header in place of the
// reference's two headers lets stereo_vo.cpp / mono_vo.cpp and the ROS nodes
// link unchanged against libvo_hip.so.
#ifndef VO_AMD_REFERENCE_ADAPTER_H_
#define VO_AMD_REFERENCE_ADAPTER_H_

#if defined(__has_include)
#if __has_include("opencv4/opencv2/core.hpp") && __has_include("eigen3/Eigen/Dense")
#define VO_AMD_HAVE_REFERENCE_TYPES 1
#endif
#endif

#ifdef VO_AMD_HAVE_REFERENCE_TYPES
#include <memory>
#include <stdexcept>
#include <vector>

#include "eigen3/Eigen/Dense"
#include "opencv4/opencv2/core.hpp"

#include "feature_tracker.h"
#include "motion_estimator.h"

// reference aliases (core/defines/define_type.h:15-64)
using Pixel = cv::Point2f;
using Point = Eigen::Vector3f;
using PixelVec = std::vector<Pixel>;
using PointVec = std::vector<Point>;
using MaskVec = std::vector<bool>;
using PoseSE3 = Eigen::Matrix4f;
using Rot3 = Eigen::Matrix3f;
using Pos3 = Eigen::Vector3f;

namespace vo_adapter {
static_assert(sizeof(cv::Point2f) == sizeof(vo::Pixel), "layout");
static_assert(sizeof(Eigen::Vector3f) == sizeof(vo::Point), "layout");
inline vo::Image view(const cv::Mat &m) {
  if (m.type() != CV_8UC1) throw std::runtime_error("libvo_hip adapter: CV_8UC1 image expected");
  return vo::Image(m.data, m.cols, m.rows, (int)m.step, 0);
}
inline vo::PixelVec &as_vo(PixelVec &v) { return reinterpret_cast<vo::PixelVec &>(v); }
inline const vo::PixelVec &as_vo(const PixelVec &v) { return reinterpret_cast<const vo::PixelVec &>(v); }
inline const vo::PointVec &as_vo(const PointVec &v) { return reinterpret_cast<const vo::PointVec &>(v); }
// Eigen is column-major: transpose into the row-major C-ABI layout and back
inline vo::PoseSE3 row_major(const PoseSE3 &T) {
  vo::PoseSE3 o;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) o[i * 4 + j] = T(i, j);
  return o;
}
inline void from_row_major(const vo::PoseSE3 &a, PoseSE3 &T) {
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) T(i, j) = a[i * 4 + j];
}
inline std::shared_ptr<vo::Context> &shared_context() {
  static std::shared_ptr<vo::Context> c;
  return c;
}
}  // namespace vo_adapter

class FeatureTracker {
 public:
  FeatureTracker() : impl_(ctx()) {}
  void track(const cv::Mat &img0, const cv::Mat &img1, const PixelVec &pts0, int window_size, int max_pyr_lvl,
             float thres_err, PixelVec &pts_track, MaskVec &mask_valid) {
    using namespace vo_adapter;
    pts_track.resize(pts0.size());
    impl_.track(view(img0), view(img1), as_vo(pts0), window_size, max_pyr_lvl, thres_err, as_vo(pts_track),
                mask_valid);
  }
  void trackBidirection(const cv::Mat &img0, const cv::Mat &img1, const PixelVec &pts0, int window_size,
                        int max_pyr_lvl, float thres_err, float thres_bidirection, PixelVec &pts_track,
                        MaskVec &mask_valid) {
    using namespace vo_adapter;
    pts_track.resize(pts0.size());
    impl_.trackBidirection(view(img0), view(img1), as_vo(pts0), window_size, max_pyr_lvl, thres_err,
                           thres_bidirection, as_vo(pts_track), mask_valid);
  }
  void trackBidirectionWithPrior(const cv::Mat &img0, const cv::Mat &img1, const PixelVec &pts0, int window_size,
                                 int max_pyr_lvl, float thres_err, float thres_bidirection, PixelVec &pts_track,
                                 MaskVec &mask_valid) {
    using namespace vo_adapter;
    impl_.trackBidirectionWithPrior(view(img0), view(img1), as_vo(pts0), window_size, max_pyr_lvl, thres_err,
                                    thres_bidirection, as_vo(pts_track), mask_valid);
  }
  void trackWithPrior(const cv::Mat &img0, const cv::Mat &img1, const PixelVec &pts0, int window_size,
                      int max_pyr_lvl, float thres_err, PixelVec &pts_track, MaskVec &mask_valid) {
    using namespace vo_adapter;
    impl_.trackWithPrior(view(img0), view(img1), as_vo(pts0), window_size, max_pyr_lvl, thres_err,
                         as_vo(pts_track), mask_valid);
  }
  // dI0u / dI0v (cv::Sobel of img0) are recomputed on the device from img0 and ignored here.
  void trackWithScale(const cv::Mat &img0, const cv::Mat & /*dI0u*/, const cv::Mat & /*dI0v*/, const cv::Mat &img1,
                      const PixelVec &pts0, const std::vector<float> &scale_est, PixelVec &pts_track,
                      MaskVec &mask_valid) {
    using namespace vo_adapter;
    impl_.trackWithScale(view(img0), view(img1), as_vo(pts0), scale_est, as_vo(pts_track), mask_valid);
  }

 private:
  static vo::ContextPtr ctx() {
    auto &c = vo_adapter::shared_context();
    if (!c) c = std::make_shared<vo::Context>(0, 4096, 2304, 16384, 6, 8);
    return c;
  }
  vo::FeatureTracker impl_;
};
#endif  // VO_AMD_HAVE_REFERENCE_TYPES
#endif
