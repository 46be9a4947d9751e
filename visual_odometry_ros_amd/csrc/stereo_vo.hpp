// stereo_vo.hpp — state of the StereoVO driver (stereo_vo.hip, stereo_vo_lba.hip)
#pragma once
#include <vector>

#include "svo_device.hpp"
#include "vo_internal.hpp"

// Keyframe-centric storage of what the reference keeps per landmark (getObservationsOnKeyframes / getRelatedKeyframePtr):
// a keyframe holds its related landmarks' ids and both pixels. The ids of a track set are ASCENDING (survivors keep their
// order, new landmarks get larger ids and are appended), so the landmark population of a window and every landmark's
// observation list come out of one merge over the window's keyframes — no per-landmark containers.
struct SvoKeyframe {
  int serial, frame_id;
  float T_wc[16];
  std::vector<int32_t> ids;   // (filled when the local BA is on)
  std::vector<float> pl, pr;  // [n][2]
};

struct vo_svo {
  vo_ctx *c = nullptr;
  vo_svo_params prm;
  int cap = 0;
  SvoTrackSet ts[2] = {};
  int cur = 0, n = 0;
  uint8_t *d_accept = nullptr;
  int *d_acc_bin = nullptr;
  SvoHdr *d_hdr = nullptr, *h_hdr = nullptr;
  int seq = 0;
  SvoCam cam;
  float T_rl[16];
  float T_wp[16], dT01[16];  // pose of the previous left frame; its getPoseDiff01()
  float kf_rot = 0.f;
  bool first = true, pending = false, pending_first = false, prefetched = false;
  const void *pre_l = nullptr, *pre_r = nullptr;
  int slot[5];
  int tab_cur = 0, tab_next = 0;
  int frame_id = 0, first_n_cand = 0;
  // keyframes
  std::vector<SvoKeyframe> keyframes;  // the window (stereo_kfs_list_)
  int n_keyframes = 0, n_kf_lms = 0;
  // landmarks that were seen on a keyframe, by id (local BA on): lm->get3DPoint(), isTriangulated() (bit 0), !isAlive() (bit 1)
  std::vector<float> lmX;
  std::vector<uint8_t> lmS;
  // pinned staging of the keyframe's track set (device -> host) and of what the BA changed (host -> device)
  int32_t *h_ids = nullptr;
  float *h_pl = nullptr, *h_pr = nullptr, *h_Xw = nullptr;
  uint8_t *h_fl = nullptr;
  // the BA problem, rebuilt at every keyframe into the same buffers
  std::vector<double> ba_X, ba_px, ba_T;
  std::vector<int32_t> ba_obs_ptr, ba_obs_frame, ba_used, ba_opt;
  std::vector<uint8_t> ba_obs_right;
};

void svo_mul44(const float A[16], const float B[16], float C[16]);
void svo_inv_se3(const float T[16], float Ti[16]);
