// stereo_vo_config.h — the reference's configuration files (config/stereo/*.yaml) into vo::StereoVOParams.
// StereoVO::loadStereoCameraIntrinsicAndUserParameters (core/visual_odometry/stereo_vo/stereo_vo.cpp:118-280) reads them
// through cv::FileStorage — OpenCV's YAML 1.0 dialect. The files use a small part of it: a `%YAML:1.0` line, comments,
// flat `a.b.c: number` entries and one `!!opencv-matrix` node (rows, cols, dt, data: [...]). This header parses exactly that
// with the standard library, so that the (mode, YAML path) constructor of the reference keeps working without OpenCV in
// the way. A key the file does not have reads as 0, as a missing cv::FileNode converts.
#ifndef VO_AMD_STEREO_VO_CONFIG_H_
#define VO_AMD_STEREO_VO_CONFIG_H_

#include <cstdlib>
#include <fstream>
#include <map>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "stereo_vo.h"

namespace vo {
namespace config_detail {
inline std::string strip(const std::string &s) {
  size_t a = s.find_first_not_of(" \t\r\n"), b = s.find_last_not_of(" \t\r\n");
  return a == std::string::npos ? std::string() : s.substr(a, b - a + 1);
}
inline std::string uncomment(const std::string &s) {
  const size_t h = s.find('#');
  return h == std::string::npos ? s : s.substr(0, h);
}
struct Parsed {
  std::map<std::string, double> num;
  std::map<std::string, std::vector<double>> mat;  // data of the !!opencv-matrix nodes, row-major as written
  double get(const std::string &k) const {
    auto it = num.find(k);
    return it == num.end() ? 0.0 : it->second;
  }
};
inline Parsed parse(std::istream &in) {
  Parsed out;
  std::string line, mat_key, data;
  bool in_data = false;
  while (std::getline(in, line)) {
    if (line.rfind("%YAML", 0) == 0 || line.rfind("---", 0) == 0) continue;
    std::string s = uncomment(line);
    if (in_data) {  // a data: [ ... ] list that runs over several lines
      data += " " + s;
      if (s.find(']') != std::string::npos) in_data = false;
    } else {
      const std::string t = strip(s);
      if (t.empty()) continue;
      const size_t c = t.find(':');
      if (c == std::string::npos) continue;
      const std::string key = strip(t.substr(0, c)), val = strip(t.substr(c + 1));
      const bool nested = s[0] == ' ' || s[0] == '\t';
      if (!nested) {
        mat_key.clear();
        if (val.rfind("!!opencv-matrix", 0) == 0)
          mat_key = key;
        else if (!val.empty())
          out.num[key] = std::strtod(val.c_str(), nullptr);
        continue;
      }
      if (mat_key.empty() || key != "data") continue;  // rows / cols / dt: the 4x4 float layout is fixed by the reader
      data = val;
      in_data = val.find(']') == std::string::npos;
    }
    if (!in_data && !mat_key.empty() && !data.empty()) {
      std::vector<double> v;
      std::string body = data;
      for (char &ch : body)
        if (ch == '[' || ch == ']' || ch == ',') ch = ' ';
      std::istringstream is(body);
      double x;
      while (is >> x) v.push_back(x);
      out.mat[mat_key] = v;
      data.clear();
    }
  }
  return out;
}
}  // namespace config_detail

// the numbers of one config/stereo/*.yaml file. Cameras: K = fx, fy, cx, cy; D = k1, k2, p1, p2, k3 (stereo_vo.cpp:147-163).
inline StereoVOParams loadStereoVOParams(const std::string &path) {
  std::ifstream f(path);
  if (!f.is_open()) throw std::runtime_error("StereoVO - failed to open the configuration file: " + path);  // stereo_vo.cpp:121-124
  const config_detail::Parsed y = config_detail::parse(f);
  StereoVOParams p;
  p.width = (int)y.get("Camera.left.width");
  p.height = (int)y.get("Camera.left.height");
  const char *side[2] = {"left", "right"};
  for (int s = 0; s < 2; ++s) {
    const std::string b = std::string("Camera.") + side[s] + ".";
    float *K = s ? p.Kr : p.Kl, *D = s ? p.Dr : p.Dl;
    K[0] = (float)y.get(b + "fx");
    K[1] = (float)y.get(b + "fy");
    K[2] = (float)y.get(b + "cx");
    K[3] = (float)y.get(b + "cy");
    D[0] = (float)y.get(b + "k1");
    D[1] = (float)y.get(b + "k2");
    D[2] = (float)y.get(b + "p1");
    D[3] = (float)y.get(b + "p2");
    D[4] = (float)y.get(b + "k3");
  }
  auto it = y.mat.find("T_lr");
  if (it != y.mat.end() && it->second.size() == 16)
    for (int k = 0; k < 16; ++k) p.T_lr[(size_t)k] = (float)it->second[(size_t)k];
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
  p.keyframe_update.thres_alive_ratio = (float)y.get("keyframe_update.thres_alive_ratio");
  p.keyframe_update.thres_trans = (float)y.get("keyframe_update.thres_trans");
  p.keyframe_update.thres_rotation = (float)y.get("keyframe_update.thres_rotation");
  p.keyframe_update.n_max_keyframes_in_window = (int)y.get("keyframe_update.n_max_keyframes_in_window");
  return p;
}

// StereoVO(mode, directory_intrinsic) of the reference (stereo_vo.cpp:6-57): "rosbag" loads the file, "dataset" throws
inline StereoVOParams stereoVOParamsForMode(const std::string &mode, const std::string &directory_intrinsic) {
  if (mode == "dataset") throw std::runtime_error("StereoVO - 'dataset' mode is not supported now...");
  if (mode != "rosbag") throw std::runtime_error("StereoVO - unknown mode...");
  return loadStereoVOParams(directory_intrinsic);
}

}  // namespace vo
#endif
