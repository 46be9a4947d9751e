// vo_kernels.hpp — host-side launchers of the gfx950 kernels (one .hip file each).
// Every launcher only ENQUEUES on ctx->stream (no synchronisation, no allocation),
// so callers can chain them and capture them.
#pragma once
#include "vo_internal.hpp"

// gn_pose.hip
int vo_gn_enqueue(vo_ctx *c, bool stereo, bool mono_general_inverse, const float *dX, const float *dP1,
                  const float *dP2, int n, const int *d_n, const float Kl[4], const float Kr[4],
                  const float T_lr[16], float thres, int variant, const float T01_init[16],
                  float *d_Tout, uint8_t *d_mask, vo_gn_dev_info *d_info);

// pyramid.hip
int vo_pyr_levels_host(int w, int h, int win, int max_level);
int vo_pyramid_build(vo_ctx *c, int slot, const uint8_t *d_img, int w, int h, int stride);

// klt_track.hip
int vo_klt_enqueue(vo_ctx *c, int slot0, int slot1, const float *d_pts0, float *d_pts1, int n_max,
                   const int *d_n, int win, int max_level, int flags, int max_iter, double eps,
                   float min_eig, uint8_t *d_status, float *d_err);
int vo_klt_mask_enqueue(vo_ctx *c, int mode, int n_max, const int *d_n, int n_cols, int n_rows,
                        float thres_err, float thres_bidir, const float *pts0, const float *pts_track,
                        const float *pts_back, const uint8_t *st_f, const uint8_t *st_b,
                        const float *err_f, const float *err_b, uint8_t *mask);

// frame_pipeline.hip
void vo_frame_free(vo_ctx *c);
