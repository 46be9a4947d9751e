// TYPE-CHECK STAND-IN, not OpenCV (see tests/typecheck_stubs/README.md).
#ifndef VO_TYPECHECK_STUB_OPENCV_CORE_
#define VO_TYPECHECK_STUB_OPENCV_CORE_
#include <cstddef>
#define CV_8UC1 0
#define CV_32FC1 5
namespace cv {
struct Point2f {
  float x, y;
  Point2f() : x(0), y(0) {}
  Point2f(float x_, float y_) : x(x_), y(y_) {}
};
class Mat {
 public:
  Mat() {}
  Mat(int rows_, int cols_, int type_, void *data_, std::size_t step_)
      : data(static_cast<unsigned char *>(data_)), rows(rows_), cols(cols_), step(step_), type_(type_) {}
  int type() const { return type_; }
  bool empty() const { return data == nullptr; }
  unsigned char *data = nullptr;
  int rows = 0, cols = 0;
  std::size_t step = 0;

 private:
  int type_ = CV_8UC1;
};
}  // namespace cv
#endif
