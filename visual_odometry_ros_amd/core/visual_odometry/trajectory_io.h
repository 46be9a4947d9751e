// trajectory_io.h — the reference's on-disk trajectory format and the per-frame execution statistics its ROS nodes
// publish.
//   * StereoVO::~StereoVO / MonoVO::~MonoVO (core/visual_odometry/stereo_vo/stereo_vo.cpp:55-115, mono_vo.cpp:64-125)
//     dump one line per frame: `<frame id> r00 r01 r02 tx r10 r11 r12 ty r20 r21 r22 tz`, 12 floats of the left
//     camera's T_wc in fixed notation with precision 4 (`of.precision(4); of.setf(std::ios_base::fixed, ...)`), one
//     space between fields, no trailing space, to a hard-coded path under /home/kch (and THROW when that path cannot be
//     opened). writeTrajectory() writes the same bytes to a path of the caller's choice.
//   * AlgorithmStatistics::ExecutionStatistics (stereo_vo.h:161-172): the ROS1 nodes read stats_execution.back()
//     (ros1/visual_odometry/stereo_vo_ros1.cpp:111-115) although StereoVO never pushes one (undefined
//     behaviour on an empty vector). A drop-in pushes one per frame; FrameTimer fills it from the wall clock around
//     enqueue / result.
#ifndef VO_AMD_TRAJECTORY_IO_H_
#define VO_AMD_TRAJECTORY_IO_H_

#include <chrono>
#include <fstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "../defines/define_type.h"

namespace vo {

// (row-major 4x4 poses: element (r, c) = T[4 r + c])
inline void writeTrajectory(const std::string &path, const std::vector<int> &frame_ids, const std::vector<PoseSE3> &T_wc) {
  if (frame_ids.size() != T_wc.size()) throw std::runtime_error("writeTrajectory: ids and poses differ in length");
  std::ofstream of(path, std::ios::trunc);
  if (!of.is_open()) throw std::runtime_error("file_dir cannot be opened!");  // stereo_vo.cpp:85
  of.precision(4);
  of.setf(std::ios_base::fixed, std::ios_base::floatfield);
  for (size_t j = 0; j < T_wc.size(); ++j) {
    of << frame_ids[j];
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 4; ++c) of << " " << T_wc[j][(size_t)(4 * r + c)];
    of << std::endl;
  }
}

struct ExecutionStatistics {  // stereo_vo.h:161-172, all in milliseconds
  float time_track = 0.f;     // the tracking + motion-estimation part of the frame (here: enqueue .. result)
  float time_1p = 0.f;        // 1-point RANSAC: not on the stereo path
  float time_5p = 0.f;        // 5-point fallback: the caller's (mono), reported by the caller
  float time_localba = 0.f;   // local BA: the caller's (keyframes only)
  float time_new = 0.f;       // new-point extraction: inside the frame when step [10] is closed on the device
  float time_total = 0.f;
};

class FrameTimer {
 public:
  void begin() { t0_ = clock::now(); }
  void afterTrack() { t1_ = clock::now(); }
  ExecutionStatistics end() {
    const auto t2 = clock::now();
    ExecutionStatistics s;
    s.time_track = ms(t0_, t1_);
    s.time_total = ms(t0_, t2);
    return s;
  }

 private:
  using clock = std::chrono::steady_clock;
  static float ms(clock::time_point a, clock::time_point b) {
    return std::chrono::duration<float, std::milli>(b - a).count();
  }
  clock::time_point t0_, t1_;
};

}  // namespace vo
#endif
