// frame_state.hpp — device / pinned buffers shared by the frame operators (frame_pipeline.hip: stereo,
// frame_mono.hip: mono). Allocated once per context by vo_frame_init (capacity cfg.max_points).
#pragma once
#include "mvo_device.hpp"
#include "svo_device.hpp"
#include "vo_internal.hpp"

// header of the packed result block (device and pinned-host copies share the layout)
struct vo_frame_hdr {
  int cnt[8];  // [0]=nA [1]=nB [2]=nC [3]=features replayed by the strict-border pass [4]=size of the pose-only BA set
               // [5]=new-point candidates emitted by the closed step [10]
  vo_gn_dev_info gn;
  int flags;
  int seq;  // host copy only: the frame's sequence number, written LAST by the BA launch (what vo_*_frame_result polls)
  float dT[16];
};

struct vo_frame_state {
  int cap;
  // inputs (device copies when the caller passes host pointers)
  float *in_l0, *in_r0, *in_X, *in_new;
  uint8_t *in_flags;
  // scratch in full index space
  float *F_scale;
  int32_t *F_orig;
  // compacted sets
  float *A_pl0, *A_pl1, *A_pr1, *A_X, *A_scale, *A_ref, *A_lastpu;
  uint8_t *A_touched, *A_cls;
  int32_t *A_orig;
  float *B_pl1, *B_pr1, *B_X;
  int32_t *B_orig;
  float *C_pl1, *C_pr1, *C_X;
  int32_t *C_orig;
  uint8_t *m1, *m2, *m3, *mG;
  uint8_t *st1, *st2, *st3;
  float *e1, *e2, *e3;
  float *new_back;
  float *bin_r;     // closed step [10]: per-bin forward result / trackBidirection mask of the frame kernel
  uint8_t *bin_m;
  int *ctl;  // fused path: error flags + replay control words (zero between frames)
  // device-side hand-shakes between the frame kernel (main stream) and the concurrent strict-border replay (its own
  // stream), never zeroed — the host hands each frame the cumulative value to wait for:
  //   sync[0] += 1 per feature past pass 1 (frame kernel)   sync[1] += 1 per replay workgroup that has finished
  int *sync;
  int sync_p1_target, sync_done_target;
  int last_replayed;  // features the previous frame's replay handled (strict-border mode 4 chooses by it)
  int conc_grid;      // workgroups of the concurrent replay's pool for the frame in flight
  // packed result block
  uint8_t *res_dev, *res_host;
  size_t res_cap;
  // views into res_dev for the frame in flight
  vo_frame_hdr *hdr;
  uint8_t *stage, *mNew;
  float *F_pl1, *F_pr1, *new_r;
  size_t off_stage, off_mnew, off_pl1, off_pr1, off_newr, off_newl, res_bytes;
  int n, n_new;
  int closed;       // the frame in flight takes its candidates from a bin table; n_new is then the number of bins
  const struct vo_cand_table *table;
  bool pending;
  int seq;          // sequence number of the frame in flight (fused stereo path: the result is awaited by polling res_host)
  bool seq_poll;
  bool known_done;  // the caller has seen a later launch of the same stream finish: the result call need not wait
  hipEvent_t ev_done;  // recorded after the packed D2H: result() waits for this, not for the stream
  // what the frame in flight was enqueued with (device pointers): vo_stereo_frame_result re-issues the frame with the
  // stream-ordered replay when the device-side join of the concurrent arrangements timed out
  struct {
    vo_stereo_params prm;
    int slot_l0, slot_l1, slot_r1;
    const float *l0, *r0, *X, *pts_new;
    const uint8_t *fl;
    int n, n_new;
    float dT_prior[16];
    int has_bins, table, has_world;
    vo_bin_params bins;
    float T_pw[16], T_cw_prior[16];
    VoAdvArgs adv;
  } again;
  // the same for the mono frame (vo_mono_frame_result)
  struct {
    vo_mono_params prm;
    int slot0, slot1, n, has_bins, table;
    const float *pts0, *Xw;
    const uint8_t *flags;
    float Tcw_prev[16], Tcw_prior[16], dT01_prior[16];
    vo_bin_params bins;
    int flag_mode;
  } again_mono;
  // MonoVO: what the NEXT mono enqueue hands to the BA launch so that its epilogue builds the next track set
  // (vo_mono_frame_set_advance, consumed by that enqueue)
  MvoAdvArgs mvo_adv;
  int mvo_adv_on;
  int mono_flag_mode;  // the NEXT mono enqueue's flag bytes are track-set flags (VO_LM_*): 1 BA class = triangulated, 2 = bundled
  // StereoVO: what the NEXT enqueue hands to the BA launch so that its epilogue builds the next track set
  // (vo_frame_set_advance, consumed by that enqueue); the DLT workers' cumulative completion count and its running target
  VoAdvArgs adv_next;
  // StereoVO's synchronous call (no pair handed over early): the NEXT enqueue starts the features' part of the frame
  // kernel at once, runs the keypoint detection of slot_l1 on the side stream next to it and tracks the candidates as a
  // launch of their own behind the detection (vo_frame_set_deferred_detection, consumed by that enqueue)
  int defer_detect;
  int mono_split;   // the mono frame in flight tracks its candidates as a launch of their own (a join that can time out)
  int *cand_done;   // cumulative count of finished workgroups of the candidates' own launches (device word, never zeroed)
  int cand_total;   // what it reads when every such launch so far has finished
  int *adv_done;
  int adv_total;
  int recovered;      // the last result was produced by such a re-issue
};

// strict-border mode 4 takes the concurrent replay for a frame when the previous one replayed at least this many features
// and the frame kernel is not many times the chip's resident wavefronts (frame_pipeline.hip, frame_mono.hip)
#define VO_CONC_MIN_REPLAYED 16
#define VO_CONC_MAX_WORKGROUPS 4096  // twice the frame kernel's resident wavefronts on 256 compute units

int vo_frame_init(vo_ctx *c);
