// sparse_bundle_adjustment.h — SparseBundleAdjustmentSolver
// (core/visual_odometry/ba_solver/sparse_bundle_adjustment.h:42-158) on the device. The reference solver pulls
// its problem out of a SparseBAParameters object (sparse_ba_parameters.h: maps from FramePtr / LandmarkPtr);
// here the same content is a plain struct of arrays that the caller's graph code fills — see
// include/vo_hip.h, vo_sba_solve, for the meaning of every field.
#ifndef VO_AMD_SPARSE_BUNDLE_ADJUSTMENT_H_
#define VO_AMD_SPARSE_BUNDLE_ADJUSTMENT_H_

#include <array>
#include <cstdint>
#include <stdexcept>
#include <vector>

#include "vo_context.h"

namespace vo {

// what SparseBAParameters holds after setPosesAndPoints (sparse_ba_parameters.h:283-440)
struct SparseBAProblem {
  std::vector<std::array<double, 16>> T_jw;  // getPose(kf): row-major, reference frame, scaled translations
  std::vector<std::int32_t> opt_index;       // getOptPoseIndex(kf), -1 = isFixFrame(kf)
  std::vector<std::array<double, 3>> X;      // getLandmarkBA(i).X
  std::vector<std::int32_t> obs_ptr;         // landmark i: observations obs_ptr[i] .. obs_ptr[i+1]-1 (kfs_seen order)
  std::vector<std::int32_t> obs_frame;       // index into T_jw (the left frame for a right-image observation)
  std::vector<std::uint8_t> obs_right;       // kf->isRightImage()
  std::vector<std::array<double, 2>> obs_px; // pts_on_kfs
};

class SparseBundleAdjustmentSolver {
 public:
  SparseBundleAdjustmentSolver(ContextPtr ctx, bool is_stereo) : ctx_(std::move(ctx)), is_stereo_mode_(is_stereo) {}
  // sparse_bundle_adjustment.cpp:56-83 (same throws)
  void setCamera(const std::array<double, 4> &K) {
    if (is_stereo_mode_)
      throw std::runtime_error("In 'SparseBundleAdjustmentSolver::setCamera()': Before call this function, 'is_stereo' should be set to 'false'.");
    Kl_ = Kr_ = K;
  }
  // T_lr: SparseBAParameters::getStereoPose() (scaled), row-major
  void setStereoCameras(const std::array<double, 4> &K0, const std::array<double, 4> &K1, const std::array<double, 16> &T_lr) {
    if (!is_stereo_mode_)
      throw std::runtime_error("In 'SparseBundleAdjustmentSolver::setStereoCameras()': Before call this function, 'is_stereo' should be set to 'true'.");
    Kl_ = K0;
    Kr_ = K1;
    T_lr_ = T_lr;
  }
  void setHuberThreshold(double thres_huber) { THRES_HUBER_ = thres_huber; }  // :52-54
  // :150-643. Poses and points of `p` are updated in place; avg_err (optional) gets the per-iteration pixel error.
  // Returns flag_success; throws where the reference throws (NaN).
  bool solveForFiniteIterations(int MAX_ITER, SparseBAProblem &p, std::vector<double> *avg_err = nullptr) {
    if (p.opt_index.size() != p.T_jw.size() || p.obs_ptr.size() != p.X.size() + 1 ||
        p.obs_frame.size() != p.obs_px.size() || p.obs_right.size() != p.obs_px.size())
      throw std::runtime_error("SparseBAProblem: array lengths are inconsistent");
    vo_sba_problem prm{};
    prm.n_frames = (int)p.T_jw.size();
    prm.n_points = (int)p.X.size();
    prm.n_obs = (int)p.obs_px.size();
    prm.n_opt = 0;
    for (std::int32_t j : p.opt_index) prm.n_opt = j + 1 > prm.n_opt ? j + 1 : prm.n_opt;
    prm.stereo = is_stereo_mode_ ? 1 : 0;
    prm.max_iter = MAX_ITER;
    for (int k = 0; k < 4; ++k) {
      prm.Kl[k] = Kl_[k];
      prm.Kr[k] = Kr_[k];
    }
    for (int k = 0; k < 16; ++k) prm.T_lr[k] = T_lr_[k];
    prm.thres_huber = THRES_HUBER_;
    std::vector<double> err((size_t)(MAX_ITER > 0 ? MAX_ITER : 1), 0.0);
    const int rc = ctx_->check(vo_sba_solve(ctx_->get(), &prm, p.T_jw.empty() ? nullptr : p.T_jw[0].data(),
                                            p.opt_index.data(), p.X.empty() ? nullptr : p.X[0].data(), p.obs_ptr.data(),
                                            p.obs_frame.data(), p.obs_right.data(),
                                            p.obs_px.empty() ? nullptr : p.obs_px[0].data(), err.data()));
    if (avg_err) avg_err->assign(err.begin(), err.begin() + (MAX_ITER > 0 ? MAX_ITER : 0));
    return rc == 1;
  }

 private:
  ContextPtr ctx_;
  bool is_stereo_mode_;
  std::array<double, 4> Kl_{}, Kr_{};
  std::array<double, 16> T_lr_{1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  double THRES_HUBER_ = 0.0;
};

}  // namespace vo
#endif
