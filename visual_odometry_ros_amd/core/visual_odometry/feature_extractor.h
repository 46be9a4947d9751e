// feature_extractor.h — FeatureExtractor::descriptorDistance
// (core/visual_odometry/feature_extractor.cpp:338-357) for descriptor sets, and the bucketing around
// cv::ORB::detect (feature_extractor.h:58-135, feature_extractor.cpp:241-277).
#ifndef VO_AMD_FEATURE_EXTRACTOR_H_
#define VO_AMD_FEATURE_EXTRACTOR_H_

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <stdexcept>
#include <vector>

#include "../defines/define_type.h"
#include "vo_context.h"

namespace vo {

class FeatureExtractor {
 public:
  explicit FeatureExtractor(ContextPtr ctx) : ctx_(std::move(ctx)) {}
  // single pair, same contract as the reference (two 32-byte ORB descriptors -> [0,256])
  int descriptorDistance(const std::uint8_t *a, const std::uint8_t *b) {
    std::uint16_t d = 0;
    ctx_->check(vo_orb_hamming(ctx_->get(), a, 1, b, 1, &d));
    return (int)d;
  }
  // ---- bucketing around cv::ORB::detect (the detector itself stays OpenCV's, on the host) ----
  // feature_extractor.cpp:26-69 / WeightBin::init (feature_extractor.h:90-118)
  void initParams(int n_cols, int n_rows, int n_bins_u, int n_bins_v, int /*THRES_FAST*/ = 15, int /*radius*/ = 5) {
    n_bins_u_ = n_bins_u;
    n_bins_v_ = n_bins_v;
    u_step_ = (int)std::floor((float)n_cols / (float)n_bins_u);
    v_step_ = (int)std::floor((float)n_rows / (float)n_bins_v);
    inv_u_step_ = 1.0f / (float)u_step_;
    inv_v_step_ = 1.0f / (float)v_step_;
    weight_.assign((size_t)n_bins_u * n_bins_v, 1);
  }
  // extractor_orb_ settings of initParams (:48-56); THRES_FAST of the call above
  void setOrbParams(int fast_threshold, int nfeatures = 10000, double scale_factor = 1.2, int n_levels = 8,
                    int edge_threshold = 31) {
    orb_ = vo_orb_params{nfeatures, scale_factor, n_levels, edge_threshold, fast_threshold};
  }
  // extractor_orb_->detect(img, fts) (:241) on the image in `slot`
  void detect(int slot, PixelVec &kp, std::vector<float> &response, std::vector<std::int32_t> &octave,
              int max_kp = 60000) {
    kp.assign((size_t)max_kp, Pixel{0.f, 0.f});
    response.assign((size_t)max_kp, 0.f);
    octave.assign((size_t)max_kp, 0);
    int n = 0;
    ctx_->check(vo_orb_detect(ctx_->get(), slot, &orb_, &kp.data()->x, response.data(), octave.data(), max_kp, &n));
    kp.resize((size_t)n);
    response.resize((size_t)n);
    octave.resize((size_t)n);
  }
  // extractORBwithBinning_fast (:211-318) with flag_nonmax_ (set by initParams, :37): detection and the per-bin
  // arg-max chained on the device. The image is the one held by `slot`.
  void extractORBwithBinning_fast(int slot, PixelVec &pts_extracted) {
    pts_extracted.assign(weight_.size() + 1, Pixel{0.f, 0.f});
    int m = 0;
    ctx_->check(vo_extract_orb_with_binning(ctx_->get(), slot, &orb_, inv_u_step_, inv_v_step_, n_bins_u_, n_bins_v_,
                                            weight_.data(), &pts_extracted.data()->x, &m, nullptr));
    pts_extracted.resize((size_t)m);
  }
  void resetWeightBin() { std::fill(weight_.begin(), weight_.end(), 1); }  // :71-74
  void suppressCenterBins() {                                              // :76-92
    const int u_cent = (int)(n_bins_u_ * 0.5), v_cent = (int)(n_bins_v_ * 0.5);
    const int wu = (int)(0.15f * n_bins_u_), wv = (int)(0.30f * n_bins_v_), wv2 = (int)(0.15f * n_bins_v_);
    for (int w = -wv; w <= wv; ++w) {
      const int v_idx = n_bins_u_ * (w + v_cent - wv2);
      for (int u = -wu; u <= wu; ++u) weight_[(size_t)(v_idx + u + u_cent)] = 0;
    }
  }
  void updateWeightBin(const PixelVec &fts) {  // :94-98
    ctx_->check(vo_weight_bin_update(ctx_->get(), fts.empty() ? zero_ : &fts.data()->x, (int)fts.size(), u_step_, v_step_,
                                     n_bins_u_, n_bins_v_, weight_.data()));
  }
  // the flag_nonmax_ branch of extractORBwithBinning_fast (:241-277) on the detector's keypoints
  void bucketKeypoints(const PixelVec &kp, const std::vector<float> &response, PixelVec &pts_extracted) {
    if (kp.size() != response.size()) throw std::runtime_error("keypoint positions / responses differ in length");
    pts_extracted.assign(weight_.size(), Pixel{0.f, 0.f});
    int m = 0;
    ctx_->check(vo_bucket_argmax(ctx_->get(), kp.empty() ? zero_ : &kp.data()->x, response.empty() ? zero_ : response.data(),
                                 (int)kp.size(), inv_u_step_, inv_v_step_, n_bins_u_, n_bins_v_, weight_.data(),
                                 &pts_extracted.data()->x, nullptr, &m));
    pts_extracted.resize((size_t)m);
  }
  const std::vector<std::int32_t> &weightBin() const { return weight_; }

  // all pairs: dist[i*nb + j]
  void descriptorDistance(const std::uint8_t *a, int na, const std::uint8_t *b, int nb,
                          std::vector<std::uint16_t> &dist) {
    dist.assign((size_t)na * nb, 0);
    if (na && nb) ctx_->check(vo_orb_hamming(ctx_->get(), a, na, b, nb, dist.data()));
  }

 private:
  ContextPtr ctx_;
  int n_bins_u_ = 0, n_bins_v_ = 0, u_step_ = 1, v_step_ = 1;
  float inv_u_step_ = 1.f, inv_v_step_ = 1.f;
  std::vector<std::int32_t> weight_;
  vo_orb_params orb_{10000, 1.2, 8, 31, 15};
  float zero_[2] = {0.f, 0.f};
};

}  // namespace vo
#endif
