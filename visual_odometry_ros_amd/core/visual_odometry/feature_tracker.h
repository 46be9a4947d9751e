// feature_tracker.h — host-side FeatureTracker with the reference's method names,
// argument order and error behaviour (core/visual_odometry/feature_tracker.h:44-104),
// on dependency-free types, forwarding to the C ABI of libvo_hip.so.
//   reference `throw std::runtime_error`  -> std::runtime_error (same message)
//   in/out semantics of pts_track / mask_valid are the reference's
//   (mask_valid.resize(n, true) keeps pre-set entries, feature_tracker.cpp:20,49,98,178,305).
#ifndef VO_AMD_FEATURE_TRACKER_H_
#define VO_AMD_FEATURE_TRACKER_H_

#include <cstdint>
#include <stdexcept>
#include <vector>

#include "../defines/define_type.h"
#include "vo_context.h"

namespace vo {

class FeatureTracker {
 public:
  explicit FeatureTracker(ContextPtr ctx) : ctx_(std::move(ctx)), slot_ids_(ctx_->n_slots(), 0) {}

  // feature_tracker.cpp:13-37
  void track(const Image &img0, const Image &img1, const PixelVec &pts0, int window_size, int max_pyr_lvl,
             float thres_err, PixelVec &pts_track, MaskVec &mask_valid) {
    const int n = (int)pts0.size();
    mask_valid.resize(n, true);
    pts_track.assign(n, Pixel());
    std::vector<std::uint8_t> m = to_u8(mask_valid);
    const int s0 = bind(img0, -1), s1 = bind(img1, s0);
    ctx_->check(vo_track(ctx_->get(), s0, s1, xy(pts0), n, window_size, max_pyr_lvl, thres_err, xy(pts_track),
                         m.data()));
    from_u8(m, mask_valid);
  }
  // feature_tracker.cpp:39-86
  void trackBidirection(const Image &img0, const Image &img1, const PixelVec &pts0, int window_size,
                        int max_pyr_lvl, float thres_err, float thres_bidirection, PixelVec &pts_track,
                        MaskVec &mask_valid) {
    const int n = (int)pts0.size();
    mask_valid.resize(n, true);
    pts_track.assign(n, Pixel());
    std::vector<std::uint8_t> m = to_u8(mask_valid);
    const int s0 = bind(img0, -1), s1 = bind(img1, s0);
    ctx_->check(vo_track_bidirection(ctx_->get(), s0, s1, xy(pts0), n, window_size, max_pyr_lvl, thres_err,
                                     thres_bidirection, xy(pts_track), m.data()));
    from_u8(m, mask_valid);
  }
  // feature_tracker.cpp:88-169 (pts_track carries the prior in)
  void trackBidirectionWithPrior(const Image &img0, const Image &img1, const PixelVec &pts0, int window_size,
                                 int max_pyr_lvl, float thres_err, float thres_bidirection, PixelVec &pts_track,
                                 MaskVec &mask_valid) {
    const int n = (int)pts0.size();
    if ((int)pts_track.size() != n) throw std::runtime_error("pts_track.size() != pts0.size()");
    mask_valid.resize(n, true);
    std::vector<std::uint8_t> m = to_u8(mask_valid);
    const int s0 = bind(img0, -1), s1 = bind(img1, s0);
    ctx_->check(vo_track_bidirection_with_prior(ctx_->get(), s0, s1, xy(pts0), n, window_size, max_pyr_lvl,
                                                thres_err, thres_bidirection, xy(pts_track), m.data()));
    from_u8(m, mask_valid);
  }
  // feature_tracker.cpp:171-206
  void trackWithPrior(const Image &img0, const Image &img1, const PixelVec &pts0, int window_size,
                      int max_pyr_lvl, float thres_err, PixelVec &pts_track, MaskVec &mask_valid) {
    const int n = (int)pts0.size();
    if ((int)pts_track.size() != n) throw std::runtime_error("pts_track.size() != pts0.size()");
    mask_valid.resize(n, true);
    std::vector<std::uint8_t> m = to_u8(mask_valid);
    const int s0 = bind(img0, -1), s1 = bind(img1, s0);
    ctx_->check(vo_track_with_prior(ctx_->get(), s0, s1, xy(pts0), n, window_size, max_pyr_lvl, thres_err,
                                    xy(pts_track), m.data()));
    from_u8(m, mask_valid);
  }
  // feature_tracker.cpp:208-234 (Tw1 row-major 4x4, K row-major 3x3)
  void calcPrior(const PixelVec &pts0, const PointVec &Xw, const PoseSE3 &Tw1, const Rot3 &K,
                 PixelVec &pts1_prior) {
    pts1_prior.resize(pts0.size());
    ctx_->check(vo_calc_prior(ctx_->get(), xy(pts0), (int)pts0.size(), &Xw.data()->x, (int)Xw.size(), Tw1.data(),
                              K.data(), xy(pts1_prior)));
  }
  // feature_tracker.cpp:236-504. The reference takes dI0u / dI0v (cv::Sobel of img0) as inputs; they
  // are recomputed on the device from img0, so they are not parameters here (the adapter ignores them).
  void trackWithScale(const Image &img0, const Image &img1, const PixelVec &pts0,
                      const std::vector<float> &scale_est, PixelVec &pts_track, MaskVec &mask_valid) {
    if (pts_track.size() != pts0.size()) throw std::runtime_error("pts_track.size() != pts0.size()");
    const int n = (int)pts0.size();
    mask_valid.resize(n, true);
    std::vector<std::uint8_t> m = to_u8(mask_valid);
    const int s0 = bind(img0, -1), s1 = bind(img1, s0);
    ctx_->check(vo_track_with_scale(ctx_->get(), s0, s1, xy(pts0), scale_est.data(), n, xy(pts_track), m.data(),
                                    /*strict_border=*/1));
    from_u8(m, mask_valid);
  }

 private:
  static const float *xy(const PixelVec &v) { return v.empty() ? nullptr_f() : &v.data()->x; }
  static float *xy(PixelVec &v) { return v.empty() ? nullptr_f() : &v.data()->x; }
  static float *nullptr_f() {
    static float dummy[2] = {0, 0};
    return dummy;
  }
  static std::vector<std::uint8_t> to_u8(const MaskVec &m) {
    std::vector<std::uint8_t> o(m.size() + 1);
    for (size_t i = 0; i < m.size(); ++i) o[i] = m[i] ? 1 : 0;
    return o;
  }
  static void from_u8(const std::vector<std::uint8_t> &u, MaskVec &m) {
    for (size_t i = 0; i < m.size(); ++i) m[i] = u[i] != 0;
  }
  // Upload `img` into a slot (or reuse the slot that already holds image id `img.id`).
  int bind(const Image &img, int avoid) {
    if (img.id) {
      for (int s = 0; s < (int)slot_ids_.size(); ++s)
        if (slot_ids_[s] == img.id) return s;
    }
    int s = next_slot_;
    if (s == avoid) s = (s + 1) % (int)slot_ids_.size();
    next_slot_ = (s + 1) % (int)slot_ids_.size();
    ctx_->check(vo_set_image(ctx_->get(), s, img.data, img.width, img.height, img.stride));
    slot_ids_[s] = img.id;
    return s;
  }
  ContextPtr ctx_;
  std::vector<std::uint64_t> slot_ids_;
  int next_slot_ = 0;
};

}  // namespace vo
#endif
