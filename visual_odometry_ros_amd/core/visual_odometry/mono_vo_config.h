// mono_vo_config.h — the reference's config/mono/*.yaml files into vo::MonoVOParams: the keys
// MonoVO::loadCameraIntrinsicAndUserParameters reads (core/visual_odometry/mono_vo/mono_vo.cpp:130-224), with the YAML 1.0
// subset parser of stereo_vo_config.h. A key the file does not have reads as 0, as a missing cv::FileNode converts.
#ifndef VO_AMD_MONO_VO_CONFIG_H_
#define VO_AMD_MONO_VO_CONFIG_H_

#include "mono_vo.h"
#include "stereo_vo_config.h"

namespace vo {

inline MonoVOParams loadMonoVOParams(const std::string &path) {
  std::ifstream f(path);
  if (!f.is_open()) throw std::runtime_error("MonoVO - failed to open the configuration file: " + path);
  const config_detail::Parsed y = config_detail::parse(f);
  MonoVOParams p;
  p.width = (int)y.get("Camera.width");
  p.height = (int)y.get("Camera.height");
  p.K[0] = (float)y.get("Camera.fx");
  p.K[1] = (float)y.get("Camera.fy");
  p.K[2] = (float)y.get("Camera.cx");
  p.K[3] = (float)y.get("Camera.cy");
  p.D[0] = (float)y.get("Camera.k1");
  p.D[1] = (float)y.get("Camera.k2");
  p.D[2] = (float)y.get("Camera.p1");
  p.D[3] = (float)y.get("Camera.p2");
  p.D[4] = (float)y.get("Camera.k3");
  p.flagDoUndistortion = (int)y.get("flagDoUndistortion") != 0;
  p.feature_tracker.thres_error = (float)y.get("feature_tracker.thres_error");
  p.feature_tracker.thres_bidirection = (float)y.get("feature_tracker.thres_bidirection");
  p.feature_tracker.thres_sampson = (float)y.get("feature_tracker.thres_sampson");
  p.feature_tracker.window_size = (int)y.get("feature_tracker.window_size");
  p.feature_tracker.max_level = (int)y.get("feature_tracker.max_level");
  p.feature_extractor.n_features = (int)y.get("feature_extractor.n_features");
  p.feature_extractor.n_bins_u = (int)y.get("feature_extractor.n_bins_u");
  p.feature_extractor.n_bins_v = (int)y.get("feature_extractor.n_bins_v");
  p.feature_extractor.thres_fastscore = (float)y.get("feature_extractor.thres_fastscore");
  p.feature_extractor.radius = (float)y.get("feature_extractor.radius");
  p.motion_estimator.thres_1p_error = (float)y.get("motion_estimator.thres_1p_error");
  p.motion_estimator.thres_5p_error = (float)y.get("motion_estimator.thres_5p_error");
  p.motion_estimator.thres_poseba_error = (float)y.get("motion_estimator.thres_poseba_error");
  p.keyframe_update.thres_translation = (float)y.get("keyframe_update.thres_translation");
  p.keyframe_update.thres_rotation = (float)y.get("keyframe_update.thres_rotation");
  p.keyframe_update.thres_overlap_ratio = (float)y.get("keyframe_update.thres_overlap_ratio");
  p.keyframe_update.n_max_keyframes_in_window = (int)y.get("keyframe_update.n_max_keyframes_in_window");
  p.map_update.thres_parallax = (float)y.get("map_update.thres_parallax");
  return p;
}

// MonoVO(mode, directory_intrinsic) of the reference (mono_vo.cpp:15-60): "rosbag" loads the file, "dataset" throws
inline MonoVOParams monoVOParamsForMode(const std::string &mode, const std::string &directory_intrinsic) {
  if (mode == "dataset") throw std::runtime_error("MonoVO - 'dataset' mode is not supported now...");
  if (mode != "rosbag") throw std::runtime_error("MonoVO - unknown mode.");
  return loadMonoVOParams(directory_intrinsic);
}

}  // namespace vo
#endif
