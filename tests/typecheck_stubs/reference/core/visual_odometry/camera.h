// TYPE-CHECK STAND-IN for the reference's core/visual_odometry/camera.h: the accessors the adapter calls
// (camera.h:100-130), nothing else (see tests/typecheck_stubs/README.md).
#ifndef VO_TYPECHECK_STUB_CAMERA_
#define VO_TYPECHECK_STUB_CAMERA_
#include "core/defines/define_type.h"
class Camera {
 public:
  Camera(float fx, float fy, float cx, float cy) : fx_(fx), fy_(fy), cx_(cx), cy_(cy) {
    Kinv_(0, 0) = 1.0f / fx;
    Kinv_(1, 1) = 1.0f / fy;
    Kinv_(0, 2) = -cx / fx;
    Kinv_(1, 2) = -cy / fy;
    Kinv_(2, 2) = 1.0f;
  }
  float fx() const { return fx_; }  // by value, as the reference's (camera.h:91-103)
  float fy() const { return fy_; }
  float cx() const { return cx_; }
  float cy() const { return cy_; }
  const Eigen::Matrix3f Kinv() const { return Kinv_; }

 private:
  float fx_, fy_, cx_, cy_;
  Eigen::Matrix3f Kinv_;
};
#endif
