// mono_vo.h — MonoVO (core/visual_odometry/mono_vo/mono_vo.h:45-272) on the C ABI (include/vo_hip.h: vo_mvo_*).
//
// Same public surface as the reference's class: trackImage(img, timestamp), getStatistics(); the statistics structs carry
// the reference's field names. What trackImage does — the track set carried from frame to frame with every landmark's
// first observation, age and parallax, the initialisation, new landmarks, keyframes, reconstruction, the mono local bundle
// adjustment — runs inside libvo_hip.so with the track set on the device (csrc/mono_vo.hip). One piece stays with the
// caller: MotionEstimator::calcPose5PointsAlgorithm (motion_estimator.cpp:21-203, OpenCV calib3d — out of scope per
// SURVEY §2), handed over as `five_point`; it is called for the second image and whenever the pose-only BA gives no pose.
#ifndef VO_AMD_MONO_VO_H_
#define VO_AMD_MONO_VO_H_

#include <chrono>
#include <cmath>
#include <cstring>
#include <functional>
#include <stdexcept>
#include <string>
#include <vector>

#include "../defines/define_type.h"
#include "trajectory_io.h"
#include "vo_context.h"

namespace vo {

struct MonoVOParams {
  int width = 752, height = 480;
  float K[4] = {458.654f, 457.296f, 367.215f, 248.375f};
  // MonoVO::AlgorithmParameters (mono_vo.h:57-110), defaults of the reference
  struct FeatureTrackerParameters {
    float thres_error = 125.0f, thres_bidirection = 1.0f, thres_sampson = 10.0f;
    int window_size = 15, max_level = 6;
  } feature_tracker;
  struct FeatureExtractorParameters {
    int n_features = 100, n_bins_u = 16, n_bins_v = 8;
    float thres_fastscore = 25.0f, radius = 15.0f;
  } feature_extractor;
  struct MotionEstimatorParameters {
    float thres_1p_error = 10.0f, thres_5p_error = 2.0f, thres_poseba_error = 5.0f;
  } motion_estimator;
  struct KeyframeUpdateParameters {
    float thres_translation = 1.0f, thres_rotation = 3.0f /* degrees */, thres_overlap_ratio = 0.7f;
    int n_max_keyframes_in_window = 9;
  } keyframe_update;
  struct MapUpdateParameters {
    float thres_parallax = 1.0f;  // DEGREES as in the YAML (the reference multiplies by D2R, mono_vo.cpp:219)
  } map_update;
  bool flagDoUndistortion = false;  // mono_vo.cpp:509-513: D = k1, k2, p1, p2, k3
  float D[5] = {0, 0, 0, 0, 0};
  // not in the reference
  int strict_border = 4;  // vo_stereo_frame_set_strict_border (4: the replay next to the frame kernel when it pays; same results as 1)
  bool local_ba = true;
  bool keyframe_statistics = false;  // stats_keyframe rewritten at every keyframe (RECORD_KEYFRAME_STAT, mono_vo.cpp:1130-1155)
};

class MonoVO {
 public:
  // calcPose5PointsAlgorithm(pts0, pts1, cam, R10, t10, X0, mask): true on success
  using FivePoint = std::function<bool(const PixelVec &pts0, const PixelVec &pts1, const float K[4], float R10[9], float t10[3],
                                       std::vector<std::uint8_t> &mask)>;
  struct AlgorithmStatistics {
    struct LandmarkStatistics {
      int n_initial = 0, n_pass_bidirection = 0, n_pass_1p = 0, n_pass_5p = 0, n_new = 0, n_final = 0;
      int max_age = 0, min_age = 0;
      float avg_age = 0.0f;
      int n_ok_parallax = 0;
      float min_parallax = 0.0f, max_parallax = 0.0f, avg_parallax = 0.0f;
    };
    struct FrameStatistics {
      PoseSE3 Twc, Tcw, dT_01, dT_10;
    };
    struct ExecutionStatistics {
      float time_track = 0.0f, time_1p = 0.0f, time_5p = 0.0f, time_localba = 0.0f, time_new = 0.0f, time_total = 0.0f;  // [ms]
    };
    struct KeyframeStatistics {
      PoseSE3 Twc;
      PointVec mappoints;
    };
    std::vector<KeyframeStatistics> stats_keyframe;
    std::vector<LandmarkStatistics> stats_landmark;
    std::vector<FrameStatistics> stats_frame;
    std::vector<ExecutionStatistics> stats_execution;
  };

  MonoVO(ContextPtr ctx, const MonoVOParams &p, FivePoint five_point) : ctx_(std::move(ctx)), prm_(p), hook_(std::move(five_point)) {
    if (!hook_) throw std::runtime_error("MonoVO: the 5-point pose hook is missing");
    vo_mvo_params q;
    std::memset(&q, 0, sizeof(q));
    q.frame.width = p.width;
    q.frame.height = p.height;
    q.frame.win = p.feature_tracker.window_size;
    q.frame.max_level = p.feature_tracker.max_level;
    q.frame.thres_err = p.feature_tracker.thres_error;
    q.frame.thres_bidirection = p.feature_tracker.thres_bidirection;
    q.frame.thres_poseba = (int)p.motion_estimator.thres_poseba_error;  // (an int parameter in the reference, motion_estimator.h:107)
    q.frame.thres_sampson = p.feature_tracker.thres_sampson;
    for (int k = 0; k < 4; ++k) q.frame.K[k] = p.K[k];
    if (p.flagDoUndistortion) {
      ctx_->check(vo_rectify_init_mono(ctx_->get(), 0, p.width, p.height, p.K, p.D));
      q.rectify = 1;
    }
    q.bins.n_bins_u = p.feature_extractor.n_bins_u;
    q.bins.n_bins_v = p.feature_extractor.n_bins_v;
    q.bins.u_step = (int)std::floor((float)p.width / (float)p.feature_extractor.n_bins_u);
    q.bins.v_step = (int)std::floor((float)p.height / (float)p.feature_extractor.n_bins_v);
    q.bins.inv_u_step = 1.0f / (float)q.bins.u_step;
    q.bins.inv_v_step = 1.0f / (float)q.bins.v_step;
    q.bins.orb.nfeatures = 10000;
    q.bins.orb.scale_factor = 1.2;
    q.bins.orb.n_levels = 8;
    q.bins.orb.edge_threshold = 31;
    q.bins.orb.fast_threshold = (int)p.feature_extractor.thres_fastscore;
    q.kf_overlap_ratio = p.keyframe_update.thres_overlap_ratio;
    q.kf_rotation_deg = p.keyframe_update.thres_rotation;
    q.kf_translation = p.keyframe_update.thres_translation;
    q.kf_window = p.keyframe_update.n_max_keyframes_in_window;
    q.thres_parallax_deg = p.map_update.thres_parallax;
    q.strict_border = p.strict_border;
    q.local_ba = p.local_ba ? 1 : 0;
    q.five_point = &MonoVO::trampoline;
    q.five_point_user = this;
    ctx_->check(vo_mvo_create(ctx_->get(), &q, &mvo_));
  }
  ~MonoVO() {
    if (mvo_) vo_mvo_destroy(mvo_);
  }
  MonoVO(const MonoVO &) = delete;
  MonoVO &operator=(const MonoVO &) = delete;

  // MonoVO::trackImage (mono_vo.cpp:496-1194). Throws std::runtime_error where the reference throws.
  void trackImage(const Image &img, const double &timestamp) {
    if (img.width != prm_.width || img.height != prm_.height) throw std::runtime_error("MonoVO: image size differs from the camera model");
    const auto t0 = std::chrono::steady_clock::now();
    vo_mvo_frame_info info;
    ctx_->check(vo_mvo_track(mvo_, img.data, img.stride, 0, timestamp, &info));
    push_statistics(info, std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count());
  }
  // NOT in the reference: the next image handed over early (see vo_mvo_prefetch)
  void enqueueImage(const Image &img, const double &timestamp) {
    t_enq_ = std::chrono::steady_clock::now();
    ctx_->check(vo_mvo_enqueue(mvo_, img.data, img.stride, 0, timestamp));
  }
  void prefetchImage(const Image &img) { ctx_->check(vo_mvo_prefetch(mvo_, img.data, img.stride, 0)); }
  void resultImage() {
    vo_mvo_frame_info info;
    ctx_->check(vo_mvo_result(mvo_, &info));
    push_statistics(info, std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_enq_).count());
  }

  const AlgorithmStatistics &getStatistics() const { return stat_; }
  void refreshKeyframeStatistics() {
    int nk = 0;
    ctx_->check(vo_mvo_keyframe_count(mvo_, &nk));
    stat_.stats_keyframe.resize((size_t)nk);
    if (nk == 0) return;
    std::vector<float> T((size_t)nk * 16);
    std::vector<std::int32_t> cnt((size_t)nk);
    std::size_t total = 0;
    ctx_->check(vo_mvo_get_keyframes(mvo_, T.data(), cnt.data(), nullptr, 0, &total));
    std::vector<Point> all(total);
    if (total) ctx_->check(vo_mvo_get_keyframes(mvo_, nullptr, nullptr, reinterpret_cast<float *>(all.data()), total, &total));
    std::size_t off = 0;
    for (int j = 0; j < nk; ++j) {
      AlgorithmStatistics::KeyframeStatistics &k = stat_.stats_keyframe[(size_t)j];
      for (int q = 0; q < 16; ++q) k.Twc[(size_t)q] = T[(size_t)j * 16 + (size_t)q];
      k.mappoints.assign(all.begin() + (std::ptrdiff_t)off, all.begin() + (std::ptrdiff_t)(off + (std::size_t)cnt[(size_t)j]));
      off += (std::size_t)cnt[(size_t)j];
    }
  }
  const vo_mvo_frame_info &lastFrameInfo() const { return last_; }

 private:
  static int trampoline(void *user, const float *p0, const float *p1, int n, const float K[4], float R10[9], float t10[3], std::uint8_t *mask) {
    MonoVO *self = static_cast<MonoVO *>(user);
    try {  // (no exception crosses the C ABI: a throwing hook fails the call, as the reference's throw does)
      PixelVec a((size_t)n), b((size_t)n);
      for (int i = 0; i < n; ++i) {
        a[(size_t)i] = Pixel{p0[2 * i], p0[2 * i + 1]};
        b[(size_t)i] = Pixel{p1[2 * i], p1[2 * i + 1]};
      }
      std::vector<std::uint8_t> m((size_t)n, 1);
      if (!self->hook_(a, b, K, R10, t10, m)) return 0;
      for (int i = 0; i < n; ++i) mask[i] = m[(size_t)i] ? 1 : 0;
      return 1;
    } catch (...) {
      return 0;
    }
  }
  static PoseSE3 inverse_se3(const PoseSE3 &T) {  // geometry::inverseSE3_f
    PoseSE3 o{};
    for (int i = 0; i < 3; ++i) {
      for (int j = 0; j < 3; ++j) o[(size_t)(i * 4 + j)] = T[(size_t)(j * 4 + i)];
      o[(size_t)(i * 4 + 3)] = ((-T[(size_t)(0 * 4 + i)]) * T[3] + (-T[(size_t)(1 * 4 + i)]) * T[7]) + (-T[(size_t)(2 * 4 + i)]) * T[11];
    }
    o[15] = 1.0f;
    return o;
  }
  void push_statistics(const vo_mvo_frame_info &info, float ms) {
    last_ = info;
    AlgorithmStatistics::FrameStatistics f;
    for (int k = 0; k < 16; ++k) {
      f.Twc[(size_t)k] = info.T_wc[k];
      f.dT_01[(size_t)k] = info.dT01[k];
    }
    f.Tcw = inverse_se3(f.Twc);
    f.dT_10 = inverse_se3(f.dT_01);
    stat_.stats_frame.push_back(f);
    // (the reference pushes a default-constructed LandmarkStatistics / ExecutionStatistics every frame, mono_vo.cpp:502-505,
    // :1180-1187; the counts the frame does produce are filled in here)
    AlgorithmStatistics::LandmarkStatistics l;
    l.n_initial = info.n_tracks_in;
    l.n_pass_bidirection = info.counts.n_klt;
    l.n_pass_1p = info.counts.n_motion;
    l.n_pass_5p = info.counts.n_final;
    l.n_new = info.n_new;
    l.n_final = info.n_tracks_out;
    stat_.stats_landmark.push_back(l);
    AlgorithmStatistics::ExecutionStatistics e;
    e.time_total = e.time_track = ms;
    stat_.stats_execution.push_back(e);
    if (prm_.keyframe_statistics && info.is_keyframe) refreshKeyframeStatistics();
  }

  ContextPtr ctx_;
  MonoVOParams prm_;
  FivePoint hook_;
  vo_mvo *mvo_ = nullptr;
  AlgorithmStatistics stat_;
  vo_mvo_frame_info last_{};
  std::chrono::steady_clock::time_point t_enq_;
};

}  // namespace vo
#endif
