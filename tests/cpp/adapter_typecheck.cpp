// adapter_typecheck.cpp — instantiates every method of the reference-typed adapter (reference_adapter.h) with the
// reference's exact argument types, so that `g++ -fsyntax-only` type-checks the whole surface, and — when built and
// run — checks the layout conversions the adapter owns WITHOUT touching the GPU (the `use_*` functions are compiled,
// never called).
#include <cstdio>
#include <cstring>

#include "visual_odometry_ros_amd/core/visual_odometry/reference_adapter.h"

// ---- signatures: exactly what stereo_vo.cpp / mono_vo.cpp write (feature_tracker.h:44-104, motion_estimator.h:107-147,
// feature_extractor.h:150-160)
void use_feature_tracker(FeatureTracker &t, const cv::Mat &I0, const cv::Mat &I1, const cv::Mat &du0, const cv::Mat &dv0,
                         const PixelVec &pts0, PixelVec &pts_track, MaskVec &mask, const PointVec &Xw, const PoseSE3 &Tw1,
                         const Eigen::Matrix3f &K, const std::vector<float> &scale) {
  t.track(I0, I1, pts0, 21, 6, 80.0f, pts_track, mask);
  t.trackBidirection(I0, I1, pts0, 21, 6, 80.0f, 0.5f, pts_track, mask);
  t.trackBidirectionWithPrior(I0, I1, pts0, 21, 6, 80.0f, 0.5f, pts_track, mask);
  t.trackWithPrior(I0, I1, pts0, 21, 6, 80.0f, pts_track, mask);
  t.calcPrior(pts0, Xw, Tw1, K, pts_track);
  t.trackWithScale(I0, du0, dv0, I1, pts0, scale, pts_track, mask);
}
bool use_motion_estimator(const PointVec &X, const PixelVec &pl, const PixelVec &pr, CameraConstPtr &cl, CameraConstPtr &cr,
                          const PoseSE3 &T_lr, PoseSE3 &T01, Rot3 &R01, Pos3 &t01, MaskVec &mask, std::vector<float> &d) {
  MotionEstimator mono;
  MotionEstimator stereo(true, T_lr);
  const int thres = 3;
  bool ok = mono.poseOnlyBundleAdjustment(X, pl, cl, thres, R01, t01, mask);
  ok = stereo.poseOnlyBundleAdjustment_Stereo(X, pl, pr, cl, cr, T_lr, 3.0f, T01, mask) && ok;
  Mat33 F10;
  mono.calcSampsonDistance(pl, pr, cl, R01, t01, d);
  mono.calcSampsonDistance(pl, pr, F10, d);
  const float one = mono.calcSampsonDistance(pl[0], pr[0], F10);
  mono.calcSymmetricEpipolarDistance(pl, pr, cl, R01, t01, d);
  mono.setThres1p(1.0f);
  mono.setThres5p(1.5f);
  return ok && one >= 0.f;
}
int use_feature_extractor(const cv::Mat &img, const cv::Mat &da, const cv::Mat &db, const PixelVec &pts, PixelVec &out) {
  FeatureExtractor e;
  e.initParams(1241, 376, 24, 12, 15, 5);
  e.updateWeightBin(pts);
  e.resetWeightBin();
  e.suppressCenterBins();
  e.setNonmaxSuppression(true);
  e.extractORBwithBinning_fast(img, out, true);
  return e.descriptorDistance(da, db);
}

// stereo_vo.h:233-249: the signatures the ROS nodes call (ros2/visual_odometry/stereo_vo_ros2.cpp:104-126)
float use_stereo_vo(const cv::Mat &img_left, const cv::Mat &img_right, const double &timestamp) {
  vo::StereoVOParams p;  // (the YAML's numbers; cv::FileStorage loading stays the caller's)
  StereoVO svo(p);
  svo.trackStereoImages(img_left, img_right, timestamp);
  const StereoVO::AlgorithmStatistics &st = svo.getStatistics();
  const PoseSE3 &Twc = st.stats_frame.back().Twc;
  const cv::Mat &dbg = svo.getDebugImage();
  (void)dbg;
  return Twc(0, 3) + st.stats_execution.back().time_total + (float)st.stats_landmark.back().n_final;
}
// mono_vo.h:235-243, :267: what ros1/visual_odometry/mono_vo_ros1.cpp calls (:49 Landmark::setPatch, :60 the constructor
// from (mode, YAML path), :123 trackImage, :126-190 the statistics, :199 getDebugImage) — plus the one binding the
// integration adds: the reference's own calcPose5PointsAlgorithm as the 5-point solver
float use_mono_vo(const cv::Mat &img, const double &timestamp) {
  Landmark::setPatch(7);
  MonoVO mvo("rosbag", "config/mono/mono0.yaml");
  mvo.setFivePointSolver([](const PixelVec &pts0, const PixelVec &pts1, const Eigen::Matrix3f &K, Rot3 &R10, Pos3 &t10, MaskVec &mask) {
    (void)pts0, (void)pts1, (void)K, (void)R10, (void)t10, (void)mask;  // motion_estimator_->calcPose5PointsAlgorithm(pts0, pts1, cam, R10, t10, X0, mask)
    return true;
  });
  mvo.trackImage(img, timestamp);
  const MonoVO::AlgorithmStatistics &st = mvo.getStatistics();
  const PoseSE3 &Twc = st.stats_frame.back().Twc;
  const cv::Mat &dbg = mvo.getDebugImage();
  (void)dbg;
  float acc = Twc(0, 3) + st.stats_execution.back().time_total + (float)st.stats_landmark.back().n_final + st.stats_landmark.back().avg_parallax;
  for (const auto &kf : st.stats_keyframe) acc += kf.Twc(2, 3) + (kf.mappoints.empty() ? 0.0f : kf.mappoints[0](2));
  return acc;
}
// the ROS 2 node as it stands (ros2/visual_odometry/stereo_vo_ros2.cpp:18-20, :104-166): constructed from (mode, YAML path),
// publishes the last pose, the keyframes' trajectory and their map points
float use_stereo_vo_like_the_ros2_node(const cv::Mat &img_left, const cv::Mat &img_right, const double &timestamp) {
  StereoVO svo(std::string("rosbag"), std::string("/tmp/config/stereo/exp_stereo2.yaml"));
  svo.trackStereoImages(img_left, img_right, timestamp);
  const StereoVO::AlgorithmStatistics &stat = svo.getStatistics();
  float s = stat.stats_frame.back().Twc(2, 3);
  for (size_t j = 0; j < stat.stats_keyframe.size(); ++j) {
    PoseSE3 Twc = stat.stats_keyframe[j].Twc;
    s += Twc(0, 3);
    for (const auto &x : stat.stats_keyframe[j].mappoints) s += x(2);
  }
  return s;
}

// ---- run-time part (CPU only): the conversions ----
static int fails = 0;
#define EXPECT(c)                                               \
  do {                                                          \
    if (!(c)) {                                                 \
      std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); \
      ++fails;                                                  \
    }                                                           \
  } while (0)

int main() {
  // a NON-symmetric 4x4: T(i,j) = 10 i + j + 1
  PoseSE3 T;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) T(i, j) = 10.0f * i + j + 1.0f;
  EXPECT(T.data()[1] == T(1, 0));  // the Eigen-typed side is column-major
  const vo::PoseSE3 a = vo_adapter::to_row_major(T);
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) EXPECT(a[(size_t)(4 * i + j)] == 10.0f * i + j + 1.0f);  // row-major on the C-ABI side
  EXPECT(a[1] == T(0, 1) && a[4] == T(1, 0) && a[1] != a[4]);
  PoseSE3 B;
  vo_adapter::from_row_major(a, B);
  EXPECT(std::memcmp(B.data(), T.data(), sizeof(float) * 16) == 0);
  // a blind memcpy of the column-major storage would have produced the transpose
  vo::PoseSE3 wrong;
  std::memcpy(wrong.data(), T.data(), sizeof(float) * 16);
  EXPECT(wrong[1] == T(1, 0) && wrong != a);
  Rot3 R;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) R(i, j) = 7.0f * i - 2.0f * j + 0.5f;
  const vo::Rot3 r = vo_adapter::to_row_major(R);
  EXPECT(r[1] == R(0, 1) && r[3] == R(1, 0));
  Rot3 R2;
  vo_adapter::from_row_major(r, R2);
  EXPECT(std::memcmp(R2.data(), R.data(), sizeof(float) * 9) == 0);
  // pixel / point vectors keep element order and values
  PixelVec p = {Pixel(1.5f, 2.5f), Pixel(-3.0f, 4.0f)};
  const vo::PixelVec q = vo_adapter::to_vo(p);
  EXPECT(q.size() == 2 && q[1].x == -3.0f && q[1].y == 4.0f);
  PixelVec p2;
  vo_adapter::from_vo(q, p2);
  EXPECT(p2.size() == 2 && p2[0].x == 1.5f && p2[0].y == 2.5f);
  PointVec X = {Point(1.0f, 2.0f, 3.0f)};
  const vo::PointVec Xv = vo_adapter::to_vo(X);
  EXPECT(Xv[0].x == 1.0f && Xv[0].y == 2.0f && Xv[0].z == 3.0f);
  // image view: stride and size come from the cv::Mat; a float image is refused like the trackers' 8-bit contract
  unsigned char buf[6 * 8] = {0};
  const cv::Mat img(6, 5, CV_8UC1, buf, 8);
  const vo::Image v = vo_adapter::view(img);
  EXPECT(v.width == 5 && v.height == 6 && v.stride == 8 && v.data == buf);
  bool threw = false;
  try {
    (void)vo_adapter::view(cv::Mat(6, 5, CV_32FC1, buf, 8));
  } catch (const std::runtime_error &) {
    threw = true;
  }
  EXPECT(threw);
  // the context is created lazily and only grows
  vo_adapter::LazyContext lc;
  EXPECT(!lc.get());
  // F10 = Kinv^T [t]x R Kinv through the camera overload's helper is exercised on the GPU box (tests/cpp/...);
  // here: intrinsics come out in the C-ABI order fx, fy, cx, cy
#ifndef VO_TYPECHECK_REAL_REFERENCE_HEADERS  // (the stand-in Camera has a 4-value constructor, the reference's has initParams)
  CameraConstPtr cam = std::make_shared<Camera>(718.856f, 718.0f, 607.19f, 185.2f);
  const vo::Camera k = vo_adapter::intrinsics(cam);
  EXPECT(k.fx == 718.856f && k.fy == 718.0f && k.cx == 607.19f && k.cy == 185.2f);
#endif
  std::printf(fails ? "adapter conversions: %d FAILED\n" : "adapter conversions ok\n", fails);
  return fails ? 1 : 0;
}
