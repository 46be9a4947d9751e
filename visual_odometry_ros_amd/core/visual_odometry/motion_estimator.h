// motion_estimator.h — host-side MotionEstimator (pose-only BA part) with the
// reference's method names and behaviour (core/visual_odometry/motion_estimator.h:107,117-120),
// forwarding to libvo_hip.so. Returns the reference's bool ("pose usable"); size
// mismatches and a stereo call on a mono estimator throw like the reference
// (motion_estimator.cpp:669-670, 866-867, 872-873).
#ifndef VO_AMD_MOTION_ESTIMATOR_H_
#define VO_AMD_MOTION_ESTIMATOR_H_

#include <cstdint>
#include <stdexcept>
#include <vector>

#include "../defines/define_type.h"
#include "vo_context.h"

namespace vo {

class MotionEstimator {
 public:
  explicit MotionEstimator(ContextPtr ctx, bool is_stereo_mode = false,
                           const PoseSE3 &T_left2right = PoseSE3{1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1})
      : ctx_(std::move(ctx)), is_stereo_mode_(is_stereo_mode), T_left2right_(T_left2right) {}

  // motion_estimator.cpp:665-861
  bool poseOnlyBundleAdjustment(const PointVec &X, const PixelVec &pts1, const Camera &cam,
                                const int &thres_reproj_outlier, Rot3 &R01_true, Pos3 &t01_true,
                                MaskVec &mask_inlier, int variant = VO_GN_VARIANT_CORE) {
    if (X.size() != pts1.size())
      throw std::runtime_error("In 'poseOnlyBundleAdjustment()': X.size() != pts1.size().");
    const int n = (int)X.size();
    mask_inlier.resize(n);
    std::vector<std::uint8_t> m(n + 1);
    const float K[4] = {cam.fx, cam.fy, cam.cx, cam.cy};
    const int rc = ctx_->check(vo_gn_pose_mono(ctx_->get(), n ? &X.data()->x : zero_, n ? &pts1.data()->x : zero_,
                                               n, K, thres_reproj_outlier, R01_true.data(), t01_true.data(),
                                               m.data(), variant, &last_info_));
    for (int i = 0; i < n; ++i) mask_inlier[i] = m[i] != 0;
    return rc == 1;
  }

  // motion_estimator.cpp:863-1088
  bool poseOnlyBundleAdjustment_Stereo(const PointVec &X, const PixelVec &pts_l1, const PixelVec &pts_r1,
                                       const Camera &cam_left, const Camera &cam_right, const PoseSE3 &T_lr,
                                       float thres_reproj_outlier, PoseSE3 &T01, MaskVec &mask_inlier) {
    if (!is_stereo_mode_)
      throw std::runtime_error("In 'poseOnlyBundleAdjustment_Stereo()', is_stereo_mode_ == false");
    if (X.size() != pts_l1.size() || X.size() != pts_r1.size())
      throw std::runtime_error(
          "In 'poseOnlyStereoBundleAdjustment()': X.size() != pts_l1.size() || X.size() != pts_r1.size().");
    const int n = (int)X.size();
    mask_inlier.assign(n, true);
    std::vector<std::uint8_t> m(n + 1);
    const float Kl[4] = {cam_left.fx, cam_left.fy, cam_left.cx, cam_left.cy};
    const float Kr[4] = {cam_right.fx, cam_right.fy, cam_right.cx, cam_right.cy};
    const int rc = ctx_->check(vo_gn_pose_stereo(
        ctx_->get(), n ? &X.data()->x : zero_, n ? &pts_l1.data()->x : zero_, n ? &pts_r1.data()->x : zero_, n, Kl,
        Kr, T_lr.data(), thres_reproj_outlier, T01.data(), m.data(), &last_info_));
    for (int i = 0; i < n; ++i) mask_inlier[i] = m[i] != 0;
    return rc == 1;
  }

  // motion_estimator.cpp:572-599 (F10 overload; F10 row-major here, see reference_adapter.h for Eigen types)
  void calcSampsonDistance(const PixelVec &pts0, const PixelVec &pts1, const Rot3 &F10, std::vector<float> &sampson_dist) {
    if (pts0.size() != pts1.size())
      throw std::runtime_error("Error in 'fineInliers1PointHistogram()': pts0.size() != pts1.size()");
    sampson_dist.resize(pts0.size());
    if (pts0.empty()) return;
    ctx_->check(vo_sampson_distance(ctx_->get(), &pts0.data()->x, &pts1.data()->x, (int)pts0.size(), F10.data(),
                                    sampson_dist.data()));
  }
  // motion_estimator.cpp:621-653 with F10 = Kinv^T [t10]x R10 Kinv computed by the caller (:636-638)
  void calcSymmetricEpipolarDistance(const PixelVec &pts0, const PixelVec &pts1, const Rot3 &F10,
                                     std::vector<float> &sym_epi_dist) {
    if (pts0.size() != pts1.size())
      throw std::runtime_error("In 'calcSymmetricEpipolarDistance()', pts0.size() != pts1.size()");
    sym_epi_dist.resize(pts0.size());
    if (pts0.empty()) return;
    ctx_->check(vo_symmetric_epipolar_distance(ctx_->get(), &pts0.data()->x, &pts1.data()->x, (int)pts0.size(),
                                               F10.data(), sym_epi_dist.data()));
  }

  void setThres1p(float thres_1p) { thres_1p_ = thres_1p; }  // motion_estimator.cpp:655-658
  void setThres5p(float thres_5p) { thres_5p_ = thres_5p; }  // :660-663

  const vo_gn_info &lastInfo() const { return last_info_; }

 private:
  ContextPtr ctx_;
  bool is_stereo_mode_;
  PoseSE3 T_left2right_;
  vo_gn_info last_info_{};
  float thres_1p_ = 0.f, thres_5p_ = 0.f;  // consumed by the host-side 1-point / 5-point RANSAC (out of scope)
  float zero_[4] = {0, 0, 0, 0};
};

}  // namespace vo
#endif
