// frame_pipeline.h — host-side drivers of the device-chained frame operators: what the reference's
// StereoVO::trackStereoImages (core/visual_odometry/stereo_vo/stereo_vo.cpp:483-711, steps [3]-[7] and
// the tracking part of [10]) and MonoVO::trackImage (core/visual_odometry/mono_vo/mono_vo.cpp:739-963)
// do between "the new image arrived" and "the landmark graph is updated", as one asynchronous call each.
// The landmark / frame / keyframe bookkeeping around it stays the caller's (it is pointer-chasing host
// code); these classes take and return the flat arrays that code already builds (lmtrack.pts0, the
// 3-D points, the motion prior) in the reference's types.
#ifndef VO_AMD_FRAME_PIPELINE_H_
#define VO_AMD_FRAME_PIPELINE_H_

#include <cstdint>
#include <stdexcept>
#include <vector>

#include "../defines/define_type.h"
#include "vo_context.h"

namespace vo {

// ---- stereo ---------------------------------------------------------------------------------------
struct StereoFrameResult {
  PixelVec pts_l1, pts_r1;             // tracked left / right pixels, index space of the input
  std::vector<std::uint8_t> stage;     // gates passed: 1 [4] l0->l1, 2 [4-1] refinement, 3 [5] l1->r1, 4 [7] BA inlier
  PoseSE3 dT_pc;                       // motion estimate (the prior when the BA failed)
  PixelVec pts_new_r;                  // step [10]: right pixels of the new-point candidates
  MaskVec mask_new;
  PixelVec pts_new;                    // enqueueClosed only: their left pixels (the bucketed keypoints, bins ascending)
  vo_frame_counts counts;
  vo_gn_info gn;
  bool pose_ok;                        // poseOnlyBundleAdjustment_Stereo's return value
};

class StereoFramePipeline {
 public:
  // strict_border: reproduce trackWithScale's never-reset tap state (feature_tracker.cpp:236-504)
  StereoFramePipeline(ContextPtr ctx, const vo_stereo_params &prm, bool strict_border = true)
      : ctx_(std::move(ctx)), prm_(prm) {
    if (ctx_->n_slots() < 3) throw std::runtime_error("StereoFramePipeline needs a context with >= 3 image slots");
    ctx_->check(vo_stereo_frame_set_strict_border(ctx_->get(), strict_border ? 1 : 0));
  }
  // 0 masked taps, 1..4 the reference's never-reset tap state (vo_hip.h: vo_stereo_frame_set_strict_border; 4 lets the
  // operator choose per frame where the replay of the border-touching features runs — same results)
  void setStrictBorderMode(int mode) { ctx_->check(vo_stereo_frame_set_strict_border(ctx_->get(), mode)); }
  // First frame: only the left image is kept (stereo_vo.cpp:212-330 initialises from it).
  void setFirstImage(const Image &left) {
    ctx_->check(vo_set_image(ctx_->get(), kPrev, left.data, left.width, left.height, left.stride));
  }
  // A new stereo pair; the previous current-left stays on the device as the new previous-left.
  void pushStereoPair(const Image &left, const Image &right) {
    if (has_cur_) ctx_->check(vo_swap_slots(ctx_->get(), kPrev, kLeft));
    ctx_->check(vo_set_image(ctx_->get(), kLeft, left.data, left.width, left.height, left.stride));
    ctx_->check(vo_set_image(ctx_->get(), kRight, right.data, right.width, right.height, right.stride));
    has_cur_ = true;
  }
  // pts_l0 / pts_r0: lmtrack pixels in the previous pair; Xp: landmarks in the previous left camera frame
  // (stereo_vo.cpp:483-494); dT_pc_prior: motion prior; pts_new: bucketed candidates of step [10];
  // triangulated[i] = lms[i]->isTriangulated() (stereo_vo.cpp:490, :599) — empty = every landmark is.
  void enqueue(const PixelVec &pts_l0, const PixelVec &pts_r0, const PointVec &Xp, const PoseSE3 &dT_pc_prior,
               const PixelVec &pts_new, const MaskVec &triangulated = MaskVec()) {
    if (pts_l0.size() != pts_r0.size() || pts_l0.size() != Xp.size())
      throw std::runtime_error("pts_l0 / pts_r0 / Xp differ in length");
    if (!triangulated.empty() && triangulated.size() != pts_l0.size())
      throw std::runtime_error("triangulated.size() != pts_l0.size()");
    n_ = (int)pts_l0.size();
    n_new_ = (int)pts_new.size();
    flags_.assign(triangulated.size(), 0);
    for (std::size_t i = 0; i < triangulated.size(); ++i) flags_[i] = triangulated[i] ? VO_LM_TRIANGULATED : 0;
    ctx_->check(vo_stereo_frame_enqueue(ctx_->get(), &prm_, kPrev, kLeft, kRight, n_ ? &pts_l0.data()->x : zero_,
                                        n_ ? &pts_r0.data()->x : zero_, n_ ? &Xp.data()->x : zero_,
                                        flags_.empty() ? nullptr : flags_.data(), n_, dT_pc_prior.data(),
                                        n_new_ ? &pts_new.data()->x : zero_, n_new_, 0));
  }
  // Step [10] closed on the device (stereo_vo.cpp:691-711; vo_hip.h: vo_stereo_frame_enqueue_closed). The best keypoint
  // of every bin of the CURRENT left image goes into `table` (0 / 1) — from the image alone, so this can be called as
  // soon as the pair is pushed, e.g. while the previous frame is still running;
  void enqueueCandidates(const vo_bin_params &bins, int table) {
    ctx_->check(vo_new_point_candidates_enqueue(ctx_->get(), kLeft, &bins, table));
  }
  // the frame then tracks every bin's candidate and reports the ones of the bins lmtrack_final leaves empty
  // (result().pts_new = their left pixels, bins ascending).
  void enqueueClosed(const PixelVec &pts_l0, const PixelVec &pts_r0, const PointVec &Xp, const PoseSE3 &dT_pc_prior,
                     const vo_bin_params &bins, int table, const MaskVec &triangulated = MaskVec()) {
    if (pts_l0.size() != pts_r0.size() || pts_l0.size() != Xp.size())
      throw std::runtime_error("pts_l0 / pts_r0 / Xp differ in length");
    if (!triangulated.empty() && triangulated.size() != pts_l0.size())
      throw std::runtime_error("triangulated.size() != pts_l0.size()");
    n_ = (int)pts_l0.size();
    n_new_ = bins.n_bins_u * bins.n_bins_v;  // capacity; the result says how many were emitted
    closed_ = true;
    flags_.assign(triangulated.size(), 0);
    for (std::size_t i = 0; i < triangulated.size(); ++i) flags_[i] = triangulated[i] ? VO_LM_TRIANGULATED : 0;
    ctx_->check(vo_stereo_frame_enqueue_closed(ctx_->get(), &prm_, kPrev, kLeft, kRight, n_ ? &pts_l0.data()->x : zero_,
                                               n_ ? &pts_r0.data()->x : zero_, n_ ? &Xp.data()->x : zero_,
                                               flags_.empty() ? nullptr : flags_.data(), n_, dT_pc_prior.data(), &bins,
                                               table, 0));
  }
  StereoFrameResult result() {
    StereoFrameResult r;
    r.pts_l1.resize(n_);
    r.pts_r1.resize(n_);
    r.stage.resize(n_);
    r.pts_new_r.resize(n_new_);
    std::vector<std::uint8_t> m(n_new_ + 1);
    ctx_->check(vo_stereo_frame_result(ctx_->get(), n_ ? &r.pts_l1.data()->x : nullptr, n_ ? &r.pts_r1.data()->x : nullptr,
                                       n_ ? r.stage.data() : nullptr, r.dT_pc.data(),
                                       n_new_ ? &r.pts_new_r.data()->x : nullptr, m.data(), &r.counts, &r.gn));
    int nn = n_new_;
    if (closed_) {  // the candidates are the device's: their number and left pixels come with the result
      closed_ = false;
      r.pts_new.resize(n_new_);
      ctx_->check(vo_stereo_frame_new_points(ctx_->get(), n_new_ ? &r.pts_new.data()->x : nullptr, &nn));
      r.pts_new.resize(nn);
      r.pts_new_r.resize(nn);
    }
    r.mask_new.assign(nn, false);
    for (int i = 0; i < nn; ++i) r.mask_new[i] = m[i] != 0;
    r.pose_ok = !r.gn.is_nan;
    return r;
  }

 private:
  enum { kPrev = 0, kLeft = 1, kRight = 2 };
  ContextPtr ctx_;
  vo_stereo_params prm_;
  int n_ = 0, n_new_ = 0;
  bool has_cur_ = false, closed_ = false;
  float zero_[3] = {0.f, 0.f, 0.f};
  std::vector<std::uint8_t> flags_;
};

// ---- mono -----------------------------------------------------------------------------------------
struct MonoFrameResult {
  PixelVec pts1;                    // refined pixel (stage >= 2), else the forward KLT result
  std::vector<float> scale;         // patch scale handed to trackWithScale
  std::vector<std::uint8_t> stage;  // 1 tracked, 2 refined, 3 motion inlier / not in the BA set, 4 Sampson gate
  PoseSE3 dT01;
  vo_mono_counts counts;
  vo_gn_info gn;
  bool need_five_point;             // mono_vo.cpp:905: the caller runs calcPose5PointsAlgorithm
  // closed new-point step (enqueueClosed): pixels in the current image, back-tracked pixels in the previous one, masks
  PixelVec pts1_new, pts0_new;
  MaskVec mask_new;
};

class MonoFramePipeline {
 public:
  enum : std::uint8_t { kBundled = 1, kPoseBA = 2 };  // flags: lm->isBundled(), member of the BA class (:800-826)
  MonoFramePipeline(ContextPtr ctx, const vo_mono_params &prm, bool strict_border = true)
      : ctx_(std::move(ctx)), prm_(prm) {
    if (ctx_->n_slots() < 2) throw std::runtime_error("MonoFramePipeline needs a context with >= 2 image slots");
    ctx_->check(vo_stereo_frame_set_strict_border(ctx_->get(), strict_border ? 1 : 0));
  }
  // The first image fills the "previous" slot; from the second on the old current image becomes the previous one.
  void pushImage(const Image &img) {
    if (n_images_ >= 2) ctx_->check(vo_swap_slots(ctx_->get(), kPrev, kCur));
    ctx_->check(vo_set_image(ctx_->get(), n_images_ == 0 ? kPrev : kCur, img.data, img.width, img.height, img.stride));
    ++n_images_;
  }
  void enqueue(const PixelVec &pts0, const PointVec &Xw, const std::vector<std::uint8_t> &flags, const PoseSE3 &Tcw_prev,
               const PoseSE3 &Tcw_prior, const PoseSE3 &dT01_prior) {
    if (pts0.size() != Xw.size() || pts0.size() != flags.size())
      throw std::runtime_error("pts0 / Xw / flags differ in length");
    n_ = (int)pts0.size();
    ctx_->check(vo_mono_frame_enqueue(ctx_->get(), &prm_, kPrev, kCur, n_ ? &pts0.data()->x : zero_,
                                      n_ ? &Xw.data()->x : zero_, n_ ? flags.data() : zero_u8_, n_, Tcw_prev.data(),
                                      Tcw_prior.data(), dT01_prior.data(), 0));
  }
  // The new-point step closed on the device (mono_vo.cpp:977-1001; vo_hip.h: vo_mono_frame_enqueue_closed): the best
  // keypoint of every bin of the CURRENT image goes into `table` (from the image alone, any time after pushImage),
  void enqueueCandidates(const vo_bin_params &bins, int table) {
    ctx_->check(vo_new_point_candidates_enqueue(ctx_->get(), kCur, &bins, table));
  }
  // the frame back-tracks every bin's candidate and reports the ones of the bins lmtrack_final leaves empty.
  void enqueueClosed(const PixelVec &pts0, const PointVec &Xw, const std::vector<std::uint8_t> &flags,
                     const PoseSE3 &Tcw_prev, const PoseSE3 &Tcw_prior, const PoseSE3 &dT01_prior,
                     const vo_bin_params &bins, int table) {
    if (pts0.size() != Xw.size() || pts0.size() != flags.size())
      throw std::runtime_error("pts0 / Xw / flags differ in length");
    n_ = (int)pts0.size();
    n_bins_ = bins.n_bins_u * bins.n_bins_v;
    ctx_->check(vo_mono_frame_enqueue_closed(ctx_->get(), &prm_, kPrev, kCur, n_ ? &pts0.data()->x : zero_,
                                             n_ ? &Xw.data()->x : zero_, n_ ? flags.data() : zero_u8_, n_, Tcw_prev.data(),
                                             Tcw_prior.data(), dT01_prior.data(), &bins, table, 0));
  }
  MonoFrameResult result() {
    MonoFrameResult r;
    r.pts1.resize(n_);
    r.scale.resize(n_);
    r.stage.resize(n_);
    ctx_->check(vo_mono_frame_result(ctx_->get(), n_ ? &r.pts1.data()->x : nullptr, n_ ? r.scale.data() : nullptr,
                                     n_ ? r.stage.data() : nullptr, r.dT01.data(), &r.counts, &r.gn));
    r.need_five_point = r.counts.need_five_point != 0;
    if (n_bins_ > 0) {
      int nn = 0;
      r.pts1_new.resize(n_bins_);
      r.pts0_new.resize(n_bins_);
      std::vector<std::uint8_t> m(n_bins_);
      ctx_->check(vo_mono_frame_new_points(ctx_->get(), &r.pts1_new.data()->x, &r.pts0_new.data()->x, m.data(), &nn));
      r.pts1_new.resize(nn);
      r.pts0_new.resize(nn);
      r.mask_new.assign(nn, false);
      for (int i = 0; i < nn; ++i) r.mask_new[i] = m[i] != 0;
      n_bins_ = 0;
    }
    return r;
  }

 private:
  enum { kPrev = 0, kCur = 1 };
  ContextPtr ctx_;
  vo_mono_params prm_;
  int n_ = 0, n_bins_ = 0;
  int n_images_ = 0;
  float zero_[3] = {0.f, 0.f, 0.f};
  std::uint8_t zero_u8_[1] = {0};
};

}  // namespace vo
#endif
