// stereo_vo.h — StereoVO (core/visual_odometry/stereo_vo/stereo_vo.h:45-273) on the C ABI (include/vo_hip.h: vo_svo_*).
//
// Same public surface as the reference's class: trackStereoImages(left, right, timestamp), getStatistics(); the
// statistics structs carry the reference's field names. The constructor takes numbers instead of (mode, YAML directory):
// loading the YAML (cv::FileStorage, stereo_vo.cpp:118-392) stays with the caller. What trackStereoImages does — the
// track set carried from frame to frame, new landmarks, keyframes, reconstruction, local bundle adjustment — runs inside
// libvo_hip.so with the track set on the device (csrc/stereo_vo.hip).
// SURVEY F9 / F12 behind switches: the reference's destructor writes /home/kch/frame_poses.txt and throws when it cannot
// (here: only when trajectory_path is set, never throwing from the destructor); the ROS1 node reads
// stats_execution.back() / stats_landmark.back(), which the reference never pushes (here: pushed every frame).
#ifndef VO_AMD_STEREO_VO_H_
#define VO_AMD_STEREO_VO_H_

#include <chrono>
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#include "../defines/define_type.h"
#include "trajectory_io.h"
#include "vo_context.h"

namespace vo {

struct StereoVOParams {
  // Camera.* / T_lr of the YAML (rectified pair)
  int width = 1241, height = 376;
  float Kl[4] = {718.856f, 718.856f, 607.1928f, 185.2157f}, Kr[4] = {718.856f, 718.856f, 607.1928f, 185.2157f};
  PoseSE3 T_lr = {1, 0, 0, 0.5371657189f, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  // StereoVO::AlgorithmParameters (stereo_vo.h:57-103), defaults of the reference
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
    float thres_alive_ratio = 0.7f;
    float thres_trans = 1.0f;     // metres
    float thres_rotation = 3.0f;  // DEGREES as in the YAML (the reference multiplies by D2R in its constructor)
    int n_max_keyframes_in_window = 9;
  } keyframe_update;
  // system_flags_.flagDoUndistortion (stereo_vo.cpp:239, :414-427): Kl / Kr / T_lr above are then the RAW cameras with the
  // distortion below (k1, k2, p1, p2, k3); the constructor makes the rectification maps on the context
  // (StereoCamera::initStereoCameraToRectify) and the loop runs on getRectifiedCamera / getRectifiedStereoPoseLeft2Right
  bool flagDoUndistortion = false;
  float Dl[5] = {0, 0, 0, 0, 0}, Dr[5] = {0, 0, 0, 0, 0};
  // not in the reference
  // stats_keyframe: the reference rewrites EVERY keyframe's pose and map points at every keyframe (stereo_vo.cpp:805-821,
  // a cost that grows with the length of the run). true: the same here (one gather + copy per keyframe so far, at every
  // keyframe); false: stats_keyframe stays empty and refreshKeyframeStatistics() fills it when the caller wants it
  bool keyframe_statistics = false;
  int strict_border = 4;        // vo_stereo_frame_set_strict_border
  bool local_ba = true;         // localBundleAdjustmentSparseSolver_Stereo at keyframes (the reference always does)
  std::string trajectory_path;  // F9: non-empty = write the frame poses there on destruction (the reference's format)
};

class StereoVO {
 public:
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
    struct KeyframeStatistics {  // what the ROS 2 node publishes as trajectory and map points (stereo_vo_ros2.cpp:141-166)
      PoseSE3 Twc;
      PointVec mappoints;
    };
    std::vector<KeyframeStatistics> stats_keyframe;
    std::vector<LandmarkStatistics> stats_landmark;
    std::vector<FrameStatistics> stats_frame;
    std::vector<ExecutionStatistics> stats_execution;
  };

  StereoVO(ContextPtr ctx, const StereoVOParams &p) : ctx_(std::move(ctx)), prm_(p) {
    vo_svo_params q;
    std::memset(&q, 0, sizeof(q));
    q.frame.width = p.width;
    q.frame.height = p.height;
    q.frame.win = p.feature_tracker.window_size;
    q.frame.max_level = p.feature_tracker.max_level;
    q.frame.thres_err = p.feature_tracker.thres_error;
    q.frame.thres_bidirection = p.feature_tracker.thres_bidirection;
    q.frame.thres_poseba = p.motion_estimator.thres_poseba_error;
    q.frame.thres_sampson = p.feature_tracker.thres_sampson;
    for (int k = 0; k < 4; ++k) {
      q.frame.Kl[k] = p.Kl[k];
      q.frame.Kr[k] = p.Kr[k];
    }
    for (int k = 0; k < 16; ++k) q.frame.T_lr[k] = p.T_lr[(size_t)k];
    if (p.flagDoUndistortion) {
      float K_rect[4], T_lr_rect[16];
      ctx_->check(vo_rectify_init_stereo(ctx_->get(), p.width, p.height, p.Kl, p.Dl, p.Kr, p.Dr, q.frame.T_lr, K_rect, T_lr_rect, nullptr));
      for (int k = 0; k < 4; ++k) q.frame.Kl[k] = q.frame.Kr[k] = K_rect[k];
      for (int k = 0; k < 16; ++k) q.frame.T_lr[k] = T_lr_rect[k];
      q.rectify = 1;
    }
    // FeatureExtractor::initParams -> WeightBin::init (feature_extractor.h:90-118, feature_extractor.cpp:48-56)
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
    q.kf_overlap_ratio = p.keyframe_update.thres_alive_ratio;
    q.kf_rotation_deg = p.keyframe_update.thres_rotation;
    q.kf_translation = p.keyframe_update.thres_trans;
    q.kf_window = p.keyframe_update.n_max_keyframes_in_window;
    q.strict_border = p.strict_border;
    q.local_ba = p.local_ba ? 1 : 0;
    ctx_->check(vo_svo_create(ctx_->get(), &q, &svo_));
  }
  ~StereoVO() {
    if (!prm_.trajectory_path.empty()) {
      try {
        std::vector<int> ids;
        std::vector<PoseSE3> poses;
        for (size_t k = 0; k < stat_.stats_frame.size(); ++k) {
          ids.push_back(frame_ids_[k]);
          poses.push_back(stat_.stats_frame[k].Twc);
        }
        writeTrajectory(prm_.trajectory_path, ids, poses);
      } catch (...) {  // (the reference's destructor is noexcept(false) and throws: not reproduced)
      }
    }
    if (svo_) vo_svo_destroy(svo_);
  }
  StereoVO(const StereoVO &) = delete;
  StereoVO &operator=(const StereoVO &) = delete;

  // StereoVO::trackStereoImages (stereo_vo.cpp:392-989). Throws std::runtime_error where the reference throws.
  void trackStereoImages(const Image &img_left, const Image &img_right, const double &timestamp) {
    if (img_left.width != prm_.width || img_left.height != prm_.height || img_right.width != prm_.width ||
        img_right.height != prm_.height || img_left.stride != img_right.stride)
      throw std::runtime_error("StereoVO: image size differs from the camera model");
    const auto t0 = std::chrono::steady_clock::now();
    vo_svo_frame_info info;
    ctx_->check(vo_svo_track(svo_, img_left.data, img_right.data, img_left.stride, 0, timestamp, &info));
    push_statistics(info, std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count());
  }
  // NOT in the reference: a caller that already holds the next pair (a recorded sequence) may hand it over before
  // trackStereoImages of the current one returns its result — see vo_svo_prefetch. The buffers must stay untouched
  // until that pair has been tracked.
  void enqueueStereoImages(const Image &l, const Image &r, const double &timestamp) {
    t_enq_ = std::chrono::steady_clock::now();
    ctx_->check(vo_svo_enqueue(svo_, l.data, r.data, l.stride, 0, timestamp));
  }
  void prefetchStereoImages(const Image &l, const Image &r) { ctx_->check(vo_svo_prefetch(svo_, l.data, r.data, l.stride, 0)); }
  void resultStereoImages() {
    vo_svo_frame_info info;
    ctx_->check(vo_svo_result(svo_, &info));
    push_statistics(info, std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_enq_).count());
  }

  // A recorded sequence (host images, all of one stride) through the three calls above, the loop inside the library
  // (vo_svo_run): every pair is handed over one frame early. One statistics record per frame, as trackStereoImages leaves it.
  void trackSequence(const std::vector<Image> &left, const std::vector<Image> &right) {
    const size_t n = left.size() < right.size() ? left.size() : right.size();
    if (n == 0) return;
    std::vector<const void *> L(n), R(n);
    for (size_t k = 0; k < n; ++k) {
      L[k] = left[k].data;
      R[k] = right[k].data;
    }
    std::vector<vo_svo_frame_info> &infos = sequence_infos_;
    infos.assign(n, vo_svo_frame_info());
    std::vector<double> stamps(n);
    const auto t0 = std::chrono::steady_clock::now();
    ctx_->check(vo_svo_run(svo_, L.data(), R.data(), (int)n, left[0].stride, 0, 0, (int)n, infos.data(), stamps.data()));
    const float total = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    for (size_t k = 0; k < n; ++k)
      push_statistics(infos[k], k == 0 ? total - (float)(1e3 * (stamps[n - 1] - stamps[0])) : (float)(1e3 * (stamps[k] - stamps[k - 1])));
  }

  const std::vector<vo_svo_frame_info> &sequenceInfos() const { return sequence_infos_; }  // (of the last trackSequence)

  const AlgorithmStatistics &getStatistics() const { return stat_; }
  // stats_keyframe as of now: every keyframe's current pose and the current 3-D points of its related landmarks
  void refreshKeyframeStatistics() {
    int nk = 0;
    ctx_->check(vo_svo_keyframe_count(svo_, &nk));
    stat_.stats_keyframe.resize((size_t)nk);
    if (nk == 0) return;
    std::vector<float> T((size_t)nk * 16);
    std::vector<std::int32_t> cnt((size_t)nk);
    std::size_t total = 0;
    ctx_->check(vo_svo_get_keyframes(svo_, T.data(), cnt.data(), nullptr, 0, &total));
    std::vector<Point> all(total);  // one gather + one copy for all keyframes
    if (total) ctx_->check(vo_svo_get_keyframes(svo_, nullptr, nullptr, reinterpret_cast<float *>(all.data()), total, &total));
    std::size_t off = 0;
    for (int j = 0; j < nk; ++j) {
      AlgorithmStatistics::KeyframeStatistics &k = stat_.stats_keyframe[(size_t)j];
      for (int q = 0; q < 16; ++q) k.Twc[(size_t)q] = T[(size_t)j * 16 + (size_t)q];
      k.mappoints.assign(all.begin() + (std::ptrdiff_t)off, all.begin() + (std::ptrdiff_t)(off + (std::size_t)cnt[(size_t)j]));
      off += (std::size_t)cnt[(size_t)j];
    }
  }
  const vo_svo_frame_info &lastFrameInfo() const { return last_; }
  // stframe_prev_'s tracked pixels and landmarks (ids, flags: VO_LM_*), for inspection
  void getTracks(std::vector<std::int32_t> &ids, PixelVec &pts_l, PixelVec &pts_r, PointVec &Xw, std::vector<std::uint8_t> &flags) {
    int n = 0;
    ctx_->check(vo_svo_get_tracks(svo_, nullptr, nullptr, nullptr, nullptr, nullptr, 0, &n));
    ids.resize((size_t)n);
    pts_l.resize((size_t)n);
    pts_r.resize((size_t)n);
    Xw.resize((size_t)n);
    flags.resize((size_t)n);
    if (n)
      ctx_->check(vo_svo_get_tracks(svo_, ids.data(), reinterpret_cast<float *>(pts_l.data()), reinterpret_cast<float *>(pts_r.data()),
                                    reinterpret_cast<float *>(Xw.data()), flags.data(), n, &n));
  }

 private:
  static PoseSE3 inverse_se3(const PoseSE3 &T) {  // geometry::inverseSE3_f
    PoseSE3 o{};
    for (int i = 0; i < 3; ++i) {
      for (int j = 0; j < 3; ++j) o[(size_t)(i * 4 + j)] = T[(size_t)(j * 4 + i)];
      o[(size_t)(i * 4 + 3)] = ((-T[(size_t)(0 * 4 + i)]) * T[3] + (-T[(size_t)(1 * 4 + i)]) * T[7]) + (-T[(size_t)(2 * 4 + i)]) * T[11];
    }
    o[15] = 1.0f;
    return o;
  }
  void push_statistics(const vo_svo_frame_info &info, float ms) {
    last_ = info;
    AlgorithmStatistics::FrameStatistics f;
    for (int k = 0; k < 16; ++k) {
      f.Twc[(size_t)k] = info.T_wc[k];
      f.dT_01[(size_t)k] = info.dT[k];
    }
    f.Tcw = inverse_se3(f.Twc);
    f.dT_10 = inverse_se3(f.dT_01);
    stat_.stats_frame.push_back(f);  // stereo_vo.cpp:979-980
    frame_ids_.push_back(info.frame_id);
    AlgorithmStatistics::LandmarkStatistics l;  // F12: the ROS1 node reads .back() of these two
    l.n_initial = info.n_tracks_in;
    l.n_pass_bidirection = info.counts.n_l1r1;
    l.n_pass_1p = l.n_pass_5p = info.counts.n_inlier;
    l.n_new = info.n_new;
    l.n_final = info.n_tracks_out;
    stat_.stats_landmark.push_back(l);
    AlgorithmStatistics::ExecutionStatistics e;
    e.time_total = e.time_track = ms;
    stat_.stats_execution.push_back(e);
    if (prm_.keyframe_statistics && info.is_keyframe) refreshKeyframeStatistics();
  }

  ContextPtr ctx_;
  StereoVOParams prm_;
  vo_svo *svo_ = nullptr;
  AlgorithmStatistics stat_;
  std::vector<int> frame_ids_;
  vo_svo_frame_info last_{};
  std::chrono::steady_clock::time_point t_enq_;
  std::vector<vo_svo_frame_info> sequence_infos_;
};

}  // namespace vo
#endif
