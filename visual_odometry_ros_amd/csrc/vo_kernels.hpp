// vo_kernels.hpp — host-side launchers of the gfx950 kernels (one .hip file each).
#pragma once
#include "vo_internal.hpp"

// gn_pose.hip
int vo_gn_enqueue(vo_ctx *c, bool stereo, bool mono_general_inverse, const float *dX, const float *dP1,
                  const float *dP2, int n, const int *d_n, const float Kl[4], const float Kr[4],
                  const float T_lr[16], float thres, int variant, const float T01_init[16],
                  float *d_Tout, uint8_t *d_mask, vo_gn_dev_info *d_info);

// frame_pipeline.hip
void vo_frame_free(vo_ctx *c);
