// config_demo.cpp — vo::loadStereoVOParams on a config file in the reference's format: prints what it read, one
// `key value` per line (tests/test_config.py compares with the Python reader).
#include <cstdio>

#include "visual_odometry_ros_amd/core/visual_odometry/stereo_vo_config.h"

int main(int argc, char **argv) {
  if (argc < 2) return 1;
  try {
    const vo::StereoVOParams p = argc > 2 ? vo::stereoVOParamsForMode(argv[2], argv[1]) : vo::loadStereoVOParams(argv[1]);
    printf("flagDoUndistortion %d\nwidth %d\nheight %d\n", p.flagDoUndistortion ? 1 : 0, p.width, p.height);
    for (int k = 0; k < 4; ++k) printf("Kl%d %.9g\nKr%d %.9g\n", k, p.Kl[k], k, p.Kr[k]);
    for (int k = 0; k < 5; ++k) printf("Dl%d %.9g\nDr%d %.9g\n", k, p.Dl[k], k, p.Dr[k]);
    for (int k = 0; k < 16; ++k) printf("T%d %.9g\n", k, p.T_lr[(size_t)k]);
    printf("thres_error %.9g\nthres_bidirection %.9g\nthres_sampson %.9g\nwindow_size %d\nmax_level %d\n", p.feature_tracker.thres_error,
           p.feature_tracker.thres_bidirection, p.feature_tracker.thres_sampson, p.feature_tracker.window_size, p.feature_tracker.max_level);
    printf("n_features %d\nn_bins_u %d\nn_bins_v %d\nthres_fastscore %.9g\nradius %.9g\n", p.feature_extractor.n_features,
           p.feature_extractor.n_bins_u, p.feature_extractor.n_bins_v, p.feature_extractor.thres_fastscore, p.feature_extractor.radius);
    printf("thres_1p_error %.9g\nthres_5p_error %.9g\nthres_poseba_error %.9g\n", p.motion_estimator.thres_1p_error,
           p.motion_estimator.thres_5p_error, p.motion_estimator.thres_poseba_error);
    printf("thres_alive_ratio %.9g\nthres_trans %.9g\nthres_rotation %.9g\nn_max_keyframes_in_window %d\n", p.keyframe_update.thres_alive_ratio,
           p.keyframe_update.thres_trans, p.keyframe_update.thres_rotation, p.keyframe_update.n_max_keyframes_in_window);
  } catch (const std::exception &e) {
    printf("error %s\n", e.what());
    return 2;
  }
  return 0;
}
