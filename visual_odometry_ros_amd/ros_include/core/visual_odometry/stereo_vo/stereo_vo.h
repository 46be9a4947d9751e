// core/visual_odometry/stereo_vo/stereo_vo.h — the include path the reference's ROS nodes use
// (ros1/visual_odometry/stereo_vo_ros1.h:31, ros2/visual_odometry/stereo_vo_ros2.h:28). Put THIS directory tree
// (visual_odometry_ros_amd/ros_include) in front of the reference root on the nodes' include path: the nodes then get the
// libvo_hip-backed StereoVO (reference_adapter.h) instead of the reference's class, while everything else they include
// (core/defines/define_type.h, core/util/timer.h, geometry_library.h, signal_handler_linux.h) still resolves to the
// reference tree. INTEGRATION.md has the CMake lines.
#ifndef VO_AMD_FORWARD_STEREO_VO_H_
#define VO_AMD_FORWARD_STEREO_VO_H_
#include "visual_odometry_ros_amd/core/visual_odometry/reference_adapter.h"
#endif
