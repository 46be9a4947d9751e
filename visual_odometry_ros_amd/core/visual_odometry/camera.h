// camera.h — the undistortion / rectification side of the reference's Camera and StereoCamera
// (core/visual_odometry/camera.h:22-195, camera.cpp:56-90, :166-183, :300-336, :364-546) on top of the C
// ABI. Maps are generated and kept on the device; "the undistorted image" is delivered straight into an
// image slot's pyramid (the float intermediate of the reference and the driver's convertTo(CV_8UC1),
// stereo_vo.cpp:420-421 / mono_vo.cpp:512, are fused away).
#ifndef VO_AMD_CAMERA_H_
#define VO_AMD_CAMERA_H_

#include <array>
#include <stdexcept>

#include "../defines/define_type.h"
#include "vo_context.h"

namespace vo {

using Distortion = std::array<float, 5>;  // k1, k2, p1, p2, k3 (cvD order, camera.h:54)

// Camera with its distortion model (the reference's `Camera`; vo::Camera in define_type.h is the pinhole POD).
class DistortedCamera {
 public:
  explicit DistortedCamera(ContextPtr ctx, int cam = 0) : ctx_(std::move(ctx)), cam_(cam) {}
  // Camera::initParams (camera.cpp:17-53): stores the model and generates the image-undistortion map
  void initParams(int n_cols, int n_rows, const Camera &K, const Distortion &D) {
    n_cols_ = n_cols;
    n_rows_ = n_rows;
    K_ = K;
    D_ = D;
    const float k[4] = {K.fx, K.fy, K.cx, K.cy};
    ctx_->check(vo_rectify_init_mono(ctx_->get(), cam_, n_cols, n_rows, k, D.data()));
  }
  // Camera::undistortImage (camera.cpp:166-183) into image slot `slot`
  void undistortImage(const Image &raw, int slot) {
    if (!raw.data || raw.width != n_cols_ || raw.height != n_rows_)
      throw std::runtime_error("undistort image: provided image has not the same size as the camera model!\n");
    ctx_->check(vo_set_image_rectified(ctx_->get(), slot, raw.data, raw.width, raw.height, raw.stride, cam_));
  }
  int cols() const { return n_cols_; }
  int rows() const { return n_rows_; }
  float fx() const { return K_.fx; }
  float fy() const { return K_.fy; }
  float cx() const { return K_.cx; }
  float cy() const { return K_.cy; }

 private:
  ContextPtr ctx_;
  int cam_, n_cols_ = 0, n_rows_ = 0;
  Camera K_{};
  Distortion D_{};
};

class StereoCamera {
 public:
  explicit StereoCamera(ContextPtr ctx) : ctx_(std::move(ctx)) {}
  void initParams(int n_cols, int n_rows, const Camera &Kl, const Distortion &Dl, const Camera &Kr,
                  const Distortion &Dr) {
    n_cols_ = n_cols;
    n_rows_ = n_rows;
    Kl_ = Kl;
    Kr_ = Kr;
    Dl_ = Dl;
    Dr_ = Dr;
  }
  void setStereoPoseLeft2Right(const PoseSE3 &T_lr) { T_lr_ = T_lr; }  // camera.cpp:276-280
  // camera.cpp:282-286 -> generateStereoImagesUndistortAndRectifyMaps (:364-546)
  void initStereoCameraToRectify() {
    const float kl[4] = {Kl_.fx, Kl_.fy, Kl_.cx, Kl_.cy}, kr[4] = {Kr_.fx, Kr_.fy, Kr_.cx, Kr_.cy};
    float k[4];
    ctx_->check(vo_rectify_init_stereo(ctx_->get(), n_cols_, n_rows_, kl, Dl_.data(), kr, Dr_.data(), T_lr_.data(), k,
                                       T_lr_rect_.data(), T_rl_rect_.data()));
    cam_rect_ = Camera{k[0], k[1], k[2], k[3]};
    is_initialized_to_stereo_rectify_ = true;
  }
  // camera.cpp:300-336; the rectified pair goes to image slots slot_l / slot_r
  void rectifyStereoImages(const Image &img_left, const Image &img_right, int slot_l, int slot_r) {
    if (!is_initialized_to_stereo_rectify_)
      throw std::runtime_error("In 'rectifyStereoImages()', is_initialized_to_stereo_rectify_ == false");
    for (const Image *im : {&img_left, &img_right})
      if (!im->data || im->width != n_cols_ || im->height != n_rows_)
        throw std::runtime_error(
            "In 'rectifyStereoImages()': provided image has not the same size as the camera model!\n");
    ctx_->check(vo_set_image_rectified(ctx_->get(), slot_l, img_left.data, img_left.width, img_left.height,
                                       img_left.stride, 0));
    ctx_->check(vo_set_image_rectified(ctx_->get(), slot_r, img_right.data, img_right.width, img_right.height,
                                       img_right.stride, 1));
  }
  const Camera &getRectifiedCamera() const {
    need_init("getRectifiedCamera()");
    return cam_rect_;
  }
  const PoseSE3 &getRectifiedStereoPoseLeft2Right() const {
    need_init("getRectifiedStereoPoseLeft2Right()");
    return T_lr_rect_;
  }
  const PoseSE3 &getRectifiedStereoPoseRight2Left() const {
    need_init("getRectifiedStereoPoseRight2Left()");
    return T_rl_rect_;
  }
  const PoseSE3 &getStereoPoseLeft2Right() const { return T_lr_; }

 private:
  void need_init(const char *where) const {
    if (!is_initialized_to_stereo_rectify_)
      throw std::runtime_error(std::string("In '") + where + "', is_initialized_to_stereo_rectify_ == false");
  }
  ContextPtr ctx_;
  int n_cols_ = 0, n_rows_ = 0;
  Camera Kl_{}, Kr_{}, cam_rect_{};
  Distortion Dl_{}, Dr_{};
  PoseSE3 T_lr_{1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1}, T_lr_rect_{}, T_rl_rect_{};
  bool is_initialized_to_stereo_rectify_ = false;
};

}  // namespace vo
#endif
