// TYPE-CHECK STAND-IN for the reference's core/defines/define_type.h (aliases of :15-64 that the adapter uses;
// see tests/typecheck_stubs/README.md). Where /root/reference exists the real file is checked as well.
#ifndef VO_TYPECHECK_STUB_DEFINE_TYPE_
#define VO_TYPECHECK_STUB_DEFINE_TYPE_
#include <memory>
#include <vector>

#include <eigen3/Eigen/Dense>

#include "opencv4/opencv2/core.hpp"
class Camera;
using Pixel = cv::Point2f;
using Point = Eigen::Vector3f;
using Mask = bool;
using CameraPtr = std::shared_ptr<Camera>;
using CameraConstPtr = const CameraPtr;
using PixelVec = std::vector<Pixel>;
using PointVec = std::vector<Point>;
using MaskVec = std::vector<Mask>;
using Pos3 = Eigen::Vector3f;
using Rot3 = Eigen::Matrix3f;
using PoseSE3 = Eigen::Matrix4f;
using Mat33 = Eigen::Matrix3f;
#endif
