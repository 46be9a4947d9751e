// define_type.h — dependency-free counterparts of the reference's type aliases
// (core/defines/define_type.h:15-64). Layouts are identical to the reference's
// cv::Point2f (2 floats), Eigen::Vector3f (3 floats, no padding) and
// std::vector<bool>; PoseSE3 here is ROW-MAJOR float[16] (Eigen::Matrix4f is
// column-major — the OpenCV/Eigen adapter transposes explicitly).
#ifndef VO_AMD_DEFINE_TYPE_H_
#define VO_AMD_DEFINE_TYPE_H_

#include <array>
#include <cstdint>
#include <vector>

namespace vo {

struct Pixel {
  float x, y;
  Pixel() : x(0), y(0) {}
  Pixel(float x_, float y_) : x(x_), y(y_) {}
};
struct Point {
  float x, y, z;
  Point() : x(0), y(0), z(0) {}
  Point(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {}
};
static_assert(sizeof(Pixel) == 8 && sizeof(Point) == 12, "AoS layouts must match cv::Point2f / Eigen::Vector3f");

using Mask = bool;
using PixelVec = std::vector<Pixel>;
using PointVec = std::vector<Point>;
using MaskVec = std::vector<Mask>;  // bit-packed, as in the reference
using PoseSE3 = std::array<float, 16>;  // row-major 4x4
using Rot3 = std::array<float, 9>;      // row-major 3x3
using Pos3 = std::array<float, 3>;

// A CV_8UC1 image view (what the reference passes as const cv::Mat&). `id` lets
// a caller tell the tracker that the same image is passed again (0 = unknown:
// the pyramid is rebuilt, exactly what cv::calcOpticalFlowPyrLK does per call).
struct Image {
  const std::uint8_t *data;
  int width, height, stride;
  std::uint64_t id;
  Image(const std::uint8_t *d, int w, int h, int s, std::uint64_t id_ = 0)
      : data(d), width(w), height(h), stride(s), id(id_) {}
};

struct Camera {  // the accessors MotionEstimator uses (core/visual_odometry/camera.h fx(), fy(), cx(), cy())
  float fx, fy, cx, cy;
};

}  // namespace vo
#endif
