// stereo_vo.hpp — state of the StereoVO driver (stereo_vo.hip, stereo_vo_lba.hip)
#pragma once
#include <vector>

#include "svo_device.hpp"
#include "vo_internal.hpp"

// Keyframe-centric storage of what the reference keeps per landmark (getObservationsOnKeyframes / getRelatedKeyframePtr):
// a keyframe holds its related landmarks' ids and both pixels. The ids of a track set are ASCENDING and dense (survivors
// keep their order, new landmarks get the next ids and are appended), so every container is a direct-address array.
struct SvoKeyframe {
  int serial, frame_id;
  float T_wc[16];
  // the keyframe's related landmarks live in slot `ring` of the device-side keyframe ring (ids, pixels)
  int ring = 0, n = 0, id_min = 0;  // entries, smallest (= first) landmark id
  int global = 0;                   // index among all keyframes of the stream (stats_keyframe)
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
  // landmark table, keyframe ring, window scratch and the solver's arena, all on the device (stereo_vo_lba.hip)
  struct vo_svo_lba *lba = nullptr;
  int mono = 0;  // the keyframe storage serves a MonoVO (mono_vo.hip): one observation per keyframe entry, bundled flags
  // all_stkeyframes_ (stats_keyframe): every keyframe's current pose (host) and where its related landmarks' ids are kept
  // on the device (a pool that only grows)
  struct SvoKfAll {
    float T_wc[16];
    int n;
    const int32_t *d_ids;
  };
  std::vector<SvoKfAll> kf_all;
  // VO_SVO_TRACE=1: where the host's time goes per frame (per StereoVO; averages on stderr every 200 frames)
  struct {
    double t_ret = 0, acc[5] = {0, 0, 0, 0, 0};  // caller between result and enqueue, enqueue, prefetch, wait in result, rest of result
    int n = 0, n_all = 0;                        // ordinary frames, all steady-state frames
  } ht;
#ifdef GN_STAMP
  double gn_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  int gn_nacc = 0;
#endif
};

int vo_svo_lba_init(vo_svo *s);  // stereo_vo_lba.hip: landmark table, keyframe ring, window scratch, solver arena — all at construction
int vo_svo_local_ba(vo_svo *s, vo_svo_frame_info *info, int id_min);
void vo_svo_lba_free(vo_svo *s);
size_t vo_svo_lba_bytes(const vo_svo *s);
int vo_svo_lba_update_points(vo_svo *s, const SvoTrackSet &ts, int n);  // a landmark got its 3-D point outside a keyframe

void svo_mul44(const float A[16], const float B[16], float C[16]);
void svo_inv_se3(const float T[16], float Ti[16]);
