// sba_device.hpp — what the kernels of the sparse local BA (sba.hip) work on, shared with the device-side problem
// builder of the StereoVO driver (stereo_vo_lba.hip): the caller fills an SbaDev inside one device arena and hands it to
// vo_sba_enqueue_iterations.
#pragma once
#include <stdint.h>

#include "vo_internal.hpp"

#define SBA_PG 8        // partial-sum workgroups per optimised pose
#define SBA_SG 8        // partial-sum workgroups per block of B C^-1 B^T
#define SBA_MAX_OPT 20  // reduced system up to 120 x 120 in LDS
#define SBA_LQ 8        // lanes per landmark in the point kernel (4 or 8)

struct SbaDev {
  int n_frames, n_opt, M, n_obs, n_slots, stereo, max_iter;
  // non-null: {M, n_obs, n_slots} live in device memory (a problem built on the device: the host never sees the counts
  // before the solve; launches are sized by upper bounds and the surplus lanes leave)
  const int *dyn;
  double Kl[4], Kr[4], R_rl[9], t_rl[3], thres_huber, lambda;
  double *T;
  const int *opt_index;
  double *X;
  const int *obs_ptr, *obs_frame;
  const uint8_t *obs_right;
  const double *obs_px;
  const int *slot_ptr, *slot_obs, *slot_j, *slot_bobs;
  // gather lists. List j spans [ptr[j], end ? end[j] : ptr[j + 1]): packed back to back by the host builder, at fixed
  // strides with explicit ends by the device builder
  const int *pose_obs_ptr, *pose_obs_end, *pose_obs, *pose_lm;  // per optimised pose: its observations and their landmarks
  const int *opt_frame;                                         // frame of optimised pose j
  const int *pose_slot_ptr, *pose_slot_end, *pose_slot, *slot_lm;
  const int *pair_ptr, *pair_end, *pair_a, *pair_b;
  double *Cinvb, *b;
  double *err_part;  // squared-error sum of each workgroup of the point kernel (n_err of them)
  int n_err;
  double *Bs, *BCs, *BCb;  // per slot: B_ji, B_ji C_i^-1 (6x3 each), (B_ji C_i^-1) b_i (6)
  double *Apart;  // n_opt * SBA_PG * 48 (36 A, 6 a, 6 BCinv_b)
  double *S;      // n_opt * n_opt * SBA_SG * 36 (partial sums; blocks below the diagonal are never used)
  double *x;      // n_opt * 6
  double *G;      // reduced system, (6 n_opt)^2 lower triangle + 6 n_opt right-hand side
  double *avg_err;
  int *flags;     // [0] error bits (1: pose NaN, 2: error NaN), [1..3] phase ticks of the solve kernel, [4..15] stamps
  // non-null (the StereoVO / MonoVO local BA, stereo_vo_lba.hip): what the host needs of the solve — res_words words at
  // res_src: poses | errors | flags | counts, all final behind the last iteration's solve — goes to pinned host memory from
  // THAT kernel, the sequence number last (word res_seq_word): the host has it while the final point update and the
  // write-back launches still run (vo_sba_delivers_result says whether the solve kernel in use does this)
  uint32_t *res_host;
  const uint32_t *res_src;
  int res_words, res_seq_word;
  uint32_t res_seq;
};
bool vo_sba_delivers_result(const vo_ctx *c, const SbaDev &d);

// bytes of the solver's own work areas for at most M landmarks, ns slots, No optimised poses (256-byte aligned pieces),
// and their placement inside an arena starting at `base` + `off`: fills the work pointers of d, returns the new offset
size_t vo_sba_place_work(SbaDev *d, uint8_t *base, size_t off, size_t M, size_t ns, int No, int max_iter, int n_err);
// enqueue max_iter iterations + the final point update on c->stream (no synchronisation); n_err = workgroups of the
// point kernel = ceil(M_upper_bound / (64 / SBA_LQ))
int vo_sba_enqueue_iterations(vo_ctx *c, const SbaDev &d, int max_iter);
