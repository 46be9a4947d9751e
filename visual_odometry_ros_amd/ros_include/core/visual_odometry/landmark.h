// core/visual_odometry/landmark.h — Landmark::setPatch for the ROS 1 nodes (stereo_vo_ros1.cpp:41, mono_vo_ros1.cpp:49);
// the landmarks themselves live on the device (reference_adapter.h).
#ifndef VO_AMD_FORWARD_LANDMARK_H_
#define VO_AMD_FORWARD_LANDMARK_H_
#include "visual_odometry_ros_amd/core/visual_odometry/reference_adapter.h"
#endif
