// vo_kernels.hpp — host-side launchers of the gfx950 kernels (one .hip file each).
// Every launcher only ENQUEUES on ctx->stream (no synchronisation, no allocation),
// so callers can chain them without returning to the host.
#pragma once
#include "vo_internal.hpp"

// gn_pose.hip
// frame mode of the GN launch (fused frame path): compaction prologue + result copy-out epilogue
struct vo_gn_frame {
  int n;                    // features in input index space
  const uint8_t *stage;
  const uint8_t *lm_flags;  // stereo: bit 0 = landmark triangulated (null: all are); BA set = stage 3 && triangulated
  const float *X, *pl1, *pr1;
  const float *T_pw;        // non-null: X holds WORLD points, the BA takes Xp = T_pw * X (stereo_vo.cpp:605)
  float *C_X, *C_pl1, *C_pr1;
  int32_t *C_orig;
  int *cnt;
  int *ctl;                 // null (general path): no control block to report / reset
  int ctl_words, nt_word;
  int *hdr_flags;
  const int *join_word;     // non-null: before anything else wait until *join_word has reached join_target (cumulative
  int join_target;          // count of the concurrent strict-border replay's finished workgroups; bounded wait)
  const void *res_dev;
  void *res_host;           // pinned, device-visible; null = no copy-out
  size_t res_bytes;
  int seq;                  // != 0: written to the host block's header (word seq_word) after everything else, system scope
  int seq_word;
  size_t res_late_bytes;    // leading part (header + stage bytes, multiple of 16) the GN launch itself still writes
  // mono frame (frame_mono.hip): the BA set is m1 && m2 && m3 instead of stage >= 3 (counts: m1, m1 && m2, the set),
  // and the epilogue is mono_gate_body(*mono_gate) — a MonoGateArgs, copied into the kernel arguments
  const uint8_t *m1, *m2, *m3;
  const struct MonoGateArgs *mono_gate;
  // closed step [10] (stereo): updateWeightBin(lmtrack_final.pts_l1) + the emission of the bucketed candidates of the
  // bins left empty, from the speculative per-bin results of the frame kernel
  int np_bins;              // > 0: enabled; n_bins_u * n_bins_v
  int np_bins_u, np_u_step, np_v_step;
  const uint8_t *np_has;    // [np_bins] table: the bin holds a keypoint
  const float *np_xy;       // [np_bins][2] table: its pixel
  const float *np_bin_r;    // [np_bins][2] frame kernel: forward result of the bin's candidate
  const uint8_t *np_bin_m;  // [np_bins]    frame kernel: trackBidirection mask of the bin's candidate
  float *np_out_l, *np_out_r;  // compacted, bins ascending (inside the result block)
  uint8_t *np_out_m;
  float *np_host_l, *np_host_r;  // the same places in the pinned host block (written entry by entry, not copied)
  uint8_t *np_host_m;
  const int *np_cand_done;  // VoNpArgs::cand_done / cand_target
  int np_cand_target;
  const struct VoAdvArgs *adv;   // StereoVO: the epilogue also builds the next track set (svo_device.hpp)
};
int vo_gn_enqueue(vo_ctx *c, bool stereo, bool mono_general_inverse, const float *dX, const float *dP1,
                  const float *dP2, int n, const int *d_n, const float Kl[4], const float Kr[4],
                  const float T_lr[16], float thres, int variant, const float T01_init[16],
                  float *d_Tout, uint8_t *d_mask, vo_gn_dev_info *d_info, bool write_init_on_nan = false,
                  uint8_t *d_stage = nullptr, const int32_t *d_orig = nullptr, int stage_val = 0,
                  float gate_thres = 0.f, const vo_gn_frame *frame = nullptr);

// pyramid.hip
int vo_pyr_levels_host(int w, int h, int win, int max_level);
int vo_pyramid_build(vo_ctx *c, int slot, const uint8_t *d_img, int w, int h, int stride);
int vo_pyramid_build_pair(vo_ctx *c, int slot_l, const uint8_t *d_l, int slot_r, const uint8_t *d_r, int w, int h,
                          int stride);

int vo_pyramid_build_rectified(vo_ctx *c, int slot, const uint8_t *d_img, int w, int h, int stride, int cam);
int vo_pyramid_build_pair_rectified(vo_ctx *c, int slot_l, const uint8_t *d_l, int slot_r, const uint8_t *d_r, int w,
                                    int h, int stride);

// rectify.hip
void vo_rectify_free(vo_ctx *c);

// sba.hip
void vo_sba_free(vo_ctx *c);

// orb_detect.hip
void vo_orb_free(vo_ctx *c);

// klt_track.hip
int vo_klt_enqueue(vo_ctx *c, int slot0, int slot1, const float *d_pts0, const float *d_pts1_init,
                   float *d_pts1, int n_max,
                   const int *d_n, int win, int max_level, int flags, int max_iter, double eps,
                   float min_eig, uint8_t *d_status, float *d_err);
int vo_klt_mask_enqueue(vo_ctx *c, int mode, int n_max, const int *d_n, int n_cols, int n_rows,
                        float thres_err, float thres_bidir, const float *pts0, const float *pts_track,
                        const float *pts_back, const uint8_t *st_f, const uint8_t *st_b,
                        const float *err_f, const float *err_b, const uint8_t *mask_in, uint8_t *mask);

// ic_refine.hip
int vo_ic_enqueue(vo_ctx *c, int slot0, int slot1, const float *d_pts0, const float *d_scale,
                  const float *d_prior, float *d_pts_track, const uint8_t *d_mask_in, uint8_t *d_mask,
                  uint8_t *d_touched, uint8_t *d_cls, float *d_last_pu, int n_max, const int *d_n,
                  int *d_flags = nullptr, bool with_records = false);
int vo_ic_strict_enqueue(vo_ctx *c, int slot0, int slot1, const float *d_pts0, const float *d_scale,
                         const float *d_prior, float *d_pts_track, uint8_t *d_mask, uint8_t *d_touched,
                         uint8_t *d_cls, float *d_last_pu, int n_max, const int *d_n, int *d_flags = nullptr,
                         bool sequential_only = false);

// the frame kernel's view of the IC state (ic_refine.hip); IcArgs lives in ic_device.hpp
struct IcArgs;
size_t vo_ic_ctl_bytes();
int vo_ic_ctl_nt_word();
int vo_ic_frame_args(vo_ctx *c, int slot0, int slot1, IcArgs *a, int *ctl, bool with_records);
void vo_ic_strict_launch(vo_ctx *c, const IcArgs &a);

// frame_fused.hip
struct vo_frame_fused_bufs {
  float *scale, *k1, *pr_prior, *pl1, *pr1, *ref, *lastpu;
  uint8_t *stage, *m2, *touched, *cls;
  float *new_r;    // new-point candidates: forward result / mask
  uint8_t *m_new;
  const uint8_t *cand_has;  // closed step [10]: candidate j is bin j of a table, present where cand_has[j] != 0
  int *ctl;        // control block (vo_ic_ctl_bytes): error flags + replay control words
  int *sync;       // [0] features past pass 1, [1] replay workgroups finished: cumulative over the frames
  int *sync_p1_target, *sync_done_target;  // host-side running totals (updated by the enqueue)
  int conc_grid;   // concurrent replay: workgroups of the pool
  int split_cands; // phase 0 launches the features only; the candidates follow as phase 2 (behind a detection still in flight)
  int *cand_done;  // phase 2: cumulative count of finished candidate workgroups (the BA launch joins on it), or null
  int *hdr_flags;  // where the frame's error flags are reported
  float *C_X, *C_pl1, *C_pr1;
  int32_t *C_orig;
  int *cnt;
};
int vo_frame_fused_supported(int win);
int vo_frame_fused_enqueue(vo_ctx *c, const vo_stereo_params *prm, int slot_l0, int slot_l1, int slot_r1,
                           const float *d_l0, const float *d_r0, const float *d_X, const uint8_t *d_flags, int n,
                           const float T_cp[16],
                           const float T_rl[16], const float *d_new, int n_new, const vo_frame_fused_bufs &b,
                           int phase, const float *T_pw = nullptr);

// misc_kernels.hip
int vo_hamming_enqueue(vo_ctx *c, const uint8_t *d_a, int na, const uint8_t *d_b, int nb, uint16_t *d_dist);
int vo_match_enqueue(vo_ctx *c, const uint8_t *d_a, int na, const uint8_t *d_b, int nb, int th_low, float ratio,
                     int32_t *d_best, uint16_t *d_bd, uint16_t *d_sd);
struct CompactArgsHost {
  const uint8_t *mask = nullptr, *alive = nullptr, *tracked = nullptr;
  const uint8_t *lm_flags = nullptr;  // drop where (lm_flags[i] & lm_reject) != 0
  int lm_reject = 0;
  int n = 0;
  const int *d_n = nullptr;
  int32_t *index_valid = nullptr;
  int *d_n_out = nullptr;
  const float *in2[4] = {nullptr, nullptr, nullptr, nullptr};
  float *out2[4] = {nullptr, nullptr, nullptr, nullptr};
  const float *in3 = nullptr;
  float *out3 = nullptr;
  const float *in1 = nullptr;
  float *out1 = nullptr;
  const int32_t *in_i = nullptr;
  int32_t *out_i = nullptr;
  uint8_t *stage = nullptr;
  int stage_val = 0;
  const float *sc_src = nullptr;
  float *sc_dst = nullptr;
  const float *gate_pts = nullptr;
  float gate_thres = 0.f;
  const uint8_t *klt_status = nullptr;
  const float *klt_err = nullptr;
  const float *klt_pts = nullptr;
  float klt_thres_err = 0.f;
  int klt_W = 0, klt_H = 0;
};
int vo_compact_enqueue(vo_ctx *c, const CompactArgsHost &h);
int vo_calc_prior_enqueue(vo_ctx *c, const float *d_pts0, int n_pts0, const float *d_Xw, int n,
                          const float T1w[16], const float K[9], float *d_out);
int vo_stereo_prior_enqueue(vo_ctx *c, const float *d_Xp, const float *d_pl0, const float *d_pr0,
                            const uint8_t *d_flags, int n,
                            const float T_cp[16], const float T_rl[16], const float Kl[4], const float Kr[4],
                            int W, int H, float *d_pl1, float *d_pr1, float *d_scale, int32_t *d_orig,
                            uint8_t *d_stage);

int vo_epi_distance_enqueue(vo_ctx *c, int mode, const float *d_pts0, const float *d_pts1, int n, const float F10[9],
                            float *d_dist);

int vo_weight_bin_update_enqueue(vo_ctx *c, const float *d_pts, int n, int u_step, int v_step, int n_bins_u,
                                 int n_bins_v, int32_t *d_weight);
int vo_bucket_argmax_enqueue(vo_ctx *c, const float *d_xy, const float *d_response, int n, float inv_u, float inv_v,
                             int n_bins_u, int n_bins_v, const int32_t *d_weight, unsigned long long *d_key,
                             float *d_pts_out, int32_t *d_idx_out, int *d_n_out, const int *d_n = nullptr);

int vo_bucket_table_enqueue(vo_ctx *c, const float *d_xy, const float *d_response, int n_max, float inv_u, float inv_v,
                            int n_bins_u, int n_bins_v, unsigned long long *d_key, float *d_tab_xy, uint8_t *d_tab_has,
                            const int *d_n);

// orb_detect.hip: per-bin candidate tables of the closed step [10] (vo_new_point_candidates_enqueue)
struct vo_cand_table {
  float *xy;        // [n_bins][2] best keypoint of the bin (level-0 pixels)
  uint8_t *has;     // [n_bins]    the bin holds a keypoint
  int n_bins;       // 0 = never filled
  hipEvent_t ready; // recorded on the side stream behind the table's kernels and the copy of its flags
  int *h_flags;     // pinned: the detector's capacity flags of this table's detection
  int dbg_filled;   // (VO_DBG_SKIP_DETECT only)
};
const vo_cand_table *vo_orb_cand_table(vo_ctx *c, int table);
// the table of an image that is not (yet) a pyramid slot: `dev_img` is the image itself (device memory, `stride` bytes per row), read by
// the tile kernels on the side stream with NO wait for anything — the caller orders the side stream behind whatever fills the
// image. Returns 1 (nothing enqueued) where the tile kernels do not apply: the caller detects from the slot as usual.
int vo_set_image_host_async(vo_ctx *c, int slot, const uint8_t *host, int width, int height, int stride);  // vo_capi.hip
int vo_new_point_candidates_enqueue_image(vo_ctx *c, const uint8_t *dev_img, int stride, int w, int h, const vo_bin_params *bp, int table);

// frame_pipeline.hip
void vo_frame_free(vo_ctx *c);
