// core/visual_odometry/mono_vo/mono_vo.h — the include path the reference's ROS 1 node uses
// (ros1/visual_odometry/mono_vo_ros1.h:32): the libvo_hip-backed MonoVO of reference_adapter.h. See stereo_vo/stereo_vo.h.
#ifndef VO_AMD_FORWARD_MONO_VO_H_
#define VO_AMD_FORWARD_MONO_VO_H_
#include "visual_odometry_ros_amd/core/visual_odometry/reference_adapter.h"
#endif
