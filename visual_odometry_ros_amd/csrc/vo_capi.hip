// vo_capi.hip — extern "C" surface of libvo_hip.so (declared in include/vo_hip.h):
// context / buffers, host<->device staging, and the per-operator entry points.
// No CPU fallback anywhere: without a usable gfx950 device every compute entry
// point returns VO_ERR_NO_DEVICE.
#include "vo_internal.hpp"
#include "vo_kernels.hpp"

#include <stdlib.h>

static thread_local char g_err[256] = "";

extern "C" int vo_abi_version(void) { return VO_HIP_ABI_VERSION; }

extern "C" int vo_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

extern "C" const char *vo_last_error(const vo_ctx *ctx) { return ctx ? ctx->err : g_err; }

extern "C" void *vo_stream(vo_ctx *ctx) { return ctx ? (void *)ctx->stream : nullptr; }

extern "C" int vo_synchronize(vo_ctx *ctx) {
  if (!ctx) return VO_ERR_INVALID;
  VO_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return VO_OK;
}

template <typename T>
static hipError_t dalloc(vo_ctx *c, T **p, size_t n) {
  return vo_dev_malloc(c, (void **)p, n * sizeof(T));
}

static size_t pyramid_bytes(int w, int h, int max_level, vo_pyramid *P) {
  size_t off = 0;
  for (int l = 0; l <= max_level && l < VO_MAX_LEVELS; ++l) {
    int stride = ((w + 2 * VO_PAD) + 63) & ~63;
    if (P) {
      P->lv[l].w = w;
      P->lv[l].h = h;
      P->lv[l].stride = stride;
      P->lv[l].base = (uint8_t *)off;  // offset for now
    }
    off += (size_t)stride * (size_t)(h + 2 * VO_PAD);
    off = (off + 255) & ~(size_t)255;
    w = (w + 1) / 2;
    h = (h + 1) / 2;
  }
  return off;
}

extern "C" int vo_create(const vo_config *cfg, vo_ctx **out) {
  if (!cfg || !out) return VO_ERR_INVALID;
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    snprintf(g_err, sizeof(g_err), "no HIP device visible (libvo_hip has no CPU fallback)");
    return VO_ERR_NO_DEVICE;
  }
  if (cfg->device < 0 || cfg->device >= ndev || cfg->max_points <= 0 || cfg->n_slots <= 0 ||
      cfg->max_width <= 0 || cfg->max_height <= 0 || cfg->max_level < 0 ||
      cfg->max_level >= VO_MAX_LEVELS) {
    snprintf(g_err, sizeof(g_err), "vo_create: invalid config");
    return VO_ERR_INVALID;
  }
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, cfg->device) != hipSuccess) return VO_ERR_HIP;
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    snprintf(g_err, sizeof(g_err), "device %d is %s; libvo_hip is built for gfx950 only", cfg->device,
             prop.gcnArchName);
    return VO_ERR_NO_DEVICE;
  }
  vo_ctx *c = (vo_ctx *)calloc(1, sizeof(vo_ctx));
  c->cfg = *cfg;
  c->device = cfg->device;
  *out = c;  // so that the caller can read the error and destroy
  VO_CHECK_HIP(c, hipSetDevice(c->device));
  VO_CHECK_HIP(c, hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  c->stream_main = c->stream;
  VO_CHECK_HIP(c, hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking));
  {
    // the concurrent strict-border replay lives here: its workgroups must find room NEXT TO the frame kernel's, so its
    // queue is served first when a wavefront slot frees up
    int least = 0, greatest = 0;
    VO_CHECK_HIP(c, hipDeviceGetStreamPriorityRange(&least, &greatest));
    VO_CHECK_HIP(c, hipStreamCreateWithPriority(&c->stream3, hipStreamNonBlocking, greatest));
  }
  VO_CHECK_HIP(c, hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
  VO_CHECK_HIP(c, hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming));
  VO_CHECK_HIP(c, hipEventCreateWithFlags(&c->ev_pyr, hipEventDisableTiming));
  const size_t N = (size_t)cfg->max_points;
  c->slots = (vo_pyramid *)calloc((size_t)cfg->n_slots, sizeof(vo_pyramid));
  for (int s = 0; s < cfg->n_slots; ++s) {
    vo_pyramid *P = &c->slots[s];
    P->bytes = pyramid_bytes(cfg->max_width, cfg->max_height, cfg->max_level, nullptr);
    VO_CHECK_HIP(c, vo_dev_malloc(c, (void **)&P->mem, P->bytes));
    P->n_levels = 0;
    VO_CHECK_HIP(c, hipEventCreateWithFlags(&P->ready, hipEventDisableTiming));
    P->seen[0] = P->seen[1] = 1;  // nothing built yet: nothing to wait for
  }
  VO_CHECK_HIP(c, dalloc(c, &c->d_pts0, 2 * N));
  VO_CHECK_HIP(c, dalloc(c, &c->d_pts1, 2 * N));
  VO_CHECK_HIP(c, dalloc(c, &c->d_pts2, 2 * N));
  VO_CHECK_HIP(c, dalloc(c, &c->d_pts3, 2 * N));
  VO_CHECK_HIP(c, dalloc(c, &c->d_err, N));
  VO_CHECK_HIP(c, dalloc(c, &c->d_err2, N));
  VO_CHECK_HIP(c, dalloc(c, &c->d_scale, N));
  VO_CHECK_HIP(c, dalloc(c, &c->d_X, 3 * N));
  VO_CHECK_HIP(c, dalloc(c, &c->d_X2, 3 * N));
  VO_CHECK_HIP(c, dalloc(c, &c->d_status, N));
  VO_CHECK_HIP(c, dalloc(c, &c->d_status2, N));
  VO_CHECK_HIP(c, dalloc(c, &c->d_mask, N));
  VO_CHECK_HIP(c, dalloc(c, &c->d_mask2, N));
  VO_CHECK_HIP(c, dalloc(c, &c->d_idx, N));
  VO_CHECK_HIP(c, dalloc(c, &c->d_count, 16));
  VO_CHECK_HIP(c, dalloc(c, &c->d_mat, 256));
  VO_CHECK_HIP(c, dalloc(c, &c->d_gninfo, 4));
  VO_CHECK_HIP(c, dalloc(c, &c->d_flags, 16));
  VO_CHECK_HIP(c, hipMemsetAsync(c->d_flags, 0, 16 * sizeof(int), c->stream));
  c->h_stage_bytes = (size_t)cfg->max_width * cfg->max_height + 64 * N + 4096;
  VO_CHECK_HIP(c, vo_host_malloc(c, (void **)&c->h_stage, c->h_stage_bytes, hipHostMallocDefault));
  VO_CHECK_HIP(c, vo_dev_malloc(c, (void **)&c->d_img_stage, (size_t)cfg->max_width * cfg->max_height));
  VO_CHECK_HIP(c, hipStreamSynchronize(c->stream));
  return VO_OK;
}

extern "C" void vo_destroy(vo_ctx *c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->stream3) (void)hipStreamSynchronize(c->stream3);
  if (c->stream2) (void)hipStreamSynchronize(c->stream2);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  vo_frame_free(c);
  vo_rectify_free(c);
  vo_sba_free(c);
  vo_orb_free(c);
  if (c->prof) {
    for (int i = 0; i < c->prof_cap; ++i) {
      (void)hipEventDestroy(c->prof[i].a);
      (void)hipEventDestroy(c->prof[i].b);
    }
    free(c->prof);
  }
  if (c->slots) {
    for (int s = 0; s < c->cfg.n_slots; ++s) {
      (void)hipFree(c->slots[s].mem);
      if (c->slots[s].stage) (void)hipFree(c->slots[s].stage);
      if (c->slots[s].ready) (void)hipEventDestroy(c->slots[s].ready);
    }
    free(c->slots);
  }
  void *bufs[] = {c->d_pts0, c->d_pts1, c->d_pts2, c->d_pts3, c->d_err, c->d_err2, c->d_scale, c->d_X, c->d_X2,
                  c->d_status, c->d_status2, c->d_mask, c->d_mask2, c->d_idx, c->d_count, c->d_mat,
                  c->d_gninfo, c->d_flags, c->d_img_stage, c->d_desc_a, c->d_desc_b, c->d_dist, c->ic_rec};
  for (void *b : bufs)
    if (b) (void)hipFree(b);
  if (c->h_stage) (void)hipHostFree(c->h_stage);
  if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
  if (c->ev_join) (void)hipEventDestroy(c->ev_join);
  if (c->ev_pyr) (void)hipEventDestroy(c->ev_pyr);
  if (c->stream3) (void)hipStreamDestroy(c->stream3);
  if (c->stream2) (void)hipStreamDestroy(c->stream2);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  free(c);
}

// ---- profiling ---------------------------------------------------------------
extern "C" int vo_profile_enable(vo_ctx *c, int max_records) {
  if (!c || max_records <= 0) return VO_ERR_INVALID;
  if (c->prof) return vo_profile_reset(c);
  c->prof = (vo_prof_rec *)calloc((size_t)max_records, sizeof(vo_prof_rec));
  for (int i = 0; i < max_records; ++i) {
    VO_CHECK_HIP(c, hipEventCreate(&c->prof[i].a));
    VO_CHECK_HIP(c, hipEventCreate(&c->prof[i].b));
  }
  c->prof_cap = max_records;
  c->prof_n = 0;
  return VO_OK;
}
extern "C" int vo_profile_reset(vo_ctx *c) {
  if (!c) return VO_ERR_INVALID;
  c->prof_n = 0;
  return VO_OK;
}
extern "C" int vo_profile_set_classes(vo_ctx *c, unsigned mask) {
  if (!c) return VO_ERR_INVALID;
  c->prof_mask = mask;
  return VO_OK;
}
extern "C" int vo_profile_get(vo_ctx *c, int cls, int *launches, double *total_ms) {
  if (!c || !launches || !total_ms) return VO_ERR_INVALID;
  *launches = 0;
  *total_ms = 0.0;
  for (int i = 0; i < c->prof_n; ++i) {
    if (c->prof[i].cls != cls) continue;
    float ms = 0.f;
    VO_CHECK_HIP(c, hipEventElapsedTime(&ms, c->prof[i].a, c->prof[i].b));
    *total_ms += ms;
    ++*launches;
  }
  return VO_OK;
}

// ---- helpers -------------------------------------------------------------------
static int check_n(vo_ctx *c, int n) {
  if (n < 0) VO_FAIL(c, VO_ERR_INVALID, "negative point count");
  if (n > c->cfg.max_points) VO_FAIL(c, VO_ERR_CAPACITY, "n=%d exceeds vo_config.max_points=%d", n, c->cfg.max_points);
  return VO_OK;
}
#define H2D(dst, src, bytes) VO_CHECK_HIP(c, hipMemcpyAsync((dst), (src), (bytes), hipMemcpyHostToDevice, c->stream))
#define D2H(dst, src, bytes) VO_CHECK_HIP(c, hipMemcpyAsync((dst), (src), (bytes), hipMemcpyDeviceToHost, c->stream))
#define SYNC() VO_CHECK_HIP(c, hipStreamSynchronize(c->stream))

// ---- MotionEstimator -----------------------------------------------------------
extern "C" int vo_gn_pose_stereo(vo_ctx *c, const float *X, const float *pts_l1, const float *pts_r1, int n,
                                 const float Kl[4], const float Kr[4], const float T_lr[16],
                                 float thres, float T01[16], uint8_t *mask_inlier, vo_gn_info *info) {
  if (!c || !X || !pts_l1 || !pts_r1 || !Kl || !Kr || !T_lr || !T01 || !mask_inlier) return VO_ERR_INVALID;
  int rc = check_n(c, n);
  if (rc) return rc;
  VO_CHECK_HIP(c, hipSetDevice(c->device));
  if (n > 0) {
    H2D(c->d_X, X, sizeof(float) * 3 * (size_t)n);
    H2D(c->d_pts0, pts_l1, sizeof(float) * 2 * (size_t)n);
    H2D(c->d_pts1, pts_r1, sizeof(float) * 2 * (size_t)n);
  }
  rc = vo_gn_enqueue(c, true, false, c->d_X, c->d_pts0, c->d_pts1, n, nullptr, Kl, Kr, T_lr, thres, 0, T01,
                     c->d_mat, c->d_mask, c->d_gninfo);
  if (rc) return rc;
  vo_gn_dev_info gi;
  float Tout[16];
  D2H(&gi, c->d_gninfo, sizeof(gi));
  D2H(Tout, c->d_mat, sizeof(Tout));
  if (n > 0) D2H(mask_inlier, c->d_mask, (size_t)n);
  SYNC();
  if (info) {
    info->iterations = gi.iterations;
    info->err = gi.err;
    info->delta_err = gi.delta_err;
    info->delta_norm = gi.delta_norm;
    info->cnt_invalid = gi.cnt_invalid;
    info->is_nan = gi.is_nan;
  }
  if (gi.is_nan) return 0;
  memcpy(T01, Tout, sizeof(Tout));
  return 1;
}

extern "C" int vo_gn_pose_mono(vo_ctx *c, const float *X, const float *pts1, int n, const float K[4],
                               int thres_reproj_outlier, float R01[9], float t01[3], uint8_t *mask_inlier,
                               int variant, vo_gn_info *info) {
  if (!c || !X || !pts1 || !K || !R01 || !t01 || !mask_inlier) return VO_ERR_INVALID;
  int rc = check_n(c, n);
  if (rc) return rc;
  VO_CHECK_HIP(c, hipSetDevice(c->device));
  if (n > 0) {
    H2D(c->d_X, X, sizeof(float) * 3 * (size_t)n);
    H2D(c->d_pts0, pts1, sizeof(float) * 2 * (size_t)n);
  }
  float T01[16] = {R01[0], R01[1], R01[2], t01[0], R01[3], R01[4], R01[5], t01[1],
                   R01[6], R01[7], R01[8], t01[2], 0, 0, 0, 1};
  rc = vo_gn_enqueue(c, false, true, c->d_X, c->d_pts0, nullptr, n, nullptr, K, K, nullptr,
                     (float)thres_reproj_outlier, variant, T01, c->d_mat, c->d_mask, c->d_gninfo);
  if (rc) return rc;
  vo_gn_dev_info gi;
  float Tout[16];
  D2H(&gi, c->d_gninfo, sizeof(gi));
  D2H(Tout, c->d_mat, sizeof(Tout));
  if (n > 0) D2H(mask_inlier, c->d_mask, (size_t)n);
  SYNC();
  if (info) {
    info->iterations = gi.iterations;
    info->err = gi.err;
    info->delta_err = gi.delta_err;
    info->delta_norm = gi.delta_norm;
    info->cnt_invalid = gi.cnt_invalid;
    info->is_nan = gi.is_nan;
  }
  if (gi.is_nan) return 0;
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) R01[i * 3 + j] = Tout[i * 4 + j];
    t01[i] = Tout[i * 4 + 3];
  }
  return 1;
}

// ---- images & pyramids ---------------------------------------------------------
extern "C" int vo_pyramid_levels(int width, int height, int win, int max_level) {
  return vo_pyr_levels_host(width, height, win, max_level);
}

extern "C" int vo_set_image_device(vo_ctx *c, int slot, const void *dev, int width, int height, int stride) {
  if (!c || !dev) return VO_ERR_INVALID;
  VO_CHECK_HIP(c, hipSetDevice(c->device));
  return vo_pyramid_build(c, slot, (const uint8_t *)dev, width, height, stride);
}

extern "C" int vo_set_image(vo_ctx *c, int slot, const uint8_t *host, int width, int height, int stride) {
  if (!c || !host) return VO_ERR_INVALID;
  if (width <= 0 || height <= 0 || width > c->cfg.max_width || height > c->cfg.max_height)
    VO_FAIL(c, VO_ERR_CAPACITY, "image %dx%d exceeds vo_config %dx%d", width, height, c->cfg.max_width,
            c->cfg.max_height);
  VO_CHECK_HIP(c, hipSetDevice(c->device));
  vo_ingest_scope ingest(c);
  // the previous use of the staging buffers must have drained
  SYNC();
  for (int y = 0; y < height; ++y) memcpy(c->h_stage + (size_t)y * width, host + (size_t)y * stride, (size_t)width);
  H2D(c->d_img_stage, c->h_stage, (size_t)width * height);
  int rc = vo_pyramid_build(c, slot, c->d_img_stage, width, height, width);
  if (rc) return rc;
  SYNC();
  return VO_OK;
}

extern "C" int vo_set_stereo_pair_device(vo_ctx *c, int slot_l, const void *dev_l, int slot_r, const void *dev_r,
                                         int width, int height, int stride) {
  if (!c || !dev_l || !dev_r) return VO_ERR_INVALID;
  VO_CHECK_HIP(c, hipSetDevice(c->device));
  return vo_pyramid_build_pair(c, slot_l, (const uint8_t *)dev_l, slot_r, (const uint8_t *)dev_r, width, height,
                               stride);
}

extern "C" int vo_set_ingest_side_stream(vo_ctx *c, int on) {
  if (!c) return VO_ERR_INVALID;
  c->ingest_side = on ? 1 : 0;
  return VO_OK;
}

// Host images without a host synchronisation: H2D of both images into the slots' staging planes and the pyramid
// chain, all on the ingest stream; the frame that reads the slots waits for the slots' events on the device.
extern "C" int vo_set_stereo_pair_host_async(vo_ctx *c, int slot_l, const uint8_t *host_l, int slot_r,
                                             const uint8_t *host_r, int width, int height, int stride) {
  if (!c || !host_l || !host_r) return VO_ERR_INVALID;
  if (slot_l < 0 || slot_l >= c->cfg.n_slots || slot_r < 0 || slot_r >= c->cfg.n_slots || slot_l == slot_r)
    VO_FAIL(c, VO_ERR_INVALID, "slot out of range");
  if (width <= 0 || height <= 0 || width > c->cfg.max_width || height > c->cfg.max_height || stride < width)
    VO_FAIL(c, VO_ERR_CAPACITY, "image %dx%d exceeds vo_config %dx%d", width, height, c->cfg.max_width,
            c->cfg.max_height);
  VO_CHECK_HIP(c, hipSetDevice(c->device));
  vo_ingest_scope ingest(c);
  hipStream_t s = c->stream;
  const int slots[2] = {slot_l, slot_r};
  const uint8_t *src[2] = {host_l, host_r};
  for (int i = 0; i < 2; ++i) {
    vo_pyramid &P = c->slots[slots[i]];
    if (!P.stage) VO_CHECK_HIP(c, vo_dev_malloc(c, (void **)&P.stage, (size_t)c->cfg.max_width * c->cfg.max_height));
    if (stride == width)  // one linear copy (a 2-D copy is issued row by row: milliseconds instead of microseconds)
      VO_CHECK_HIP(c, hipMemcpyAsync(P.stage, src[i], (size_t)width * height, hipMemcpyHostToDevice, s));
    else
      VO_CHECK_HIP(c, hipMemcpy2DAsync(P.stage, (size_t)width, src[i], (size_t)stride, (size_t)width, (size_t)height,
                                       hipMemcpyHostToDevice, s));
    if (i == 0 && c->early_bins && s != c->stream2) {
      // StereoVO's synchronous call: the left image is on its way — its keypoint detection follows it on the side stream (one
      // event), under the right image's upload and the pair's pyramids
      VO_CHECK_HIP(c, hipEventRecord(c->ev_fork, s));
      VO_CHECK_HIP(c, hipStreamWaitEvent(c->stream2, c->ev_fork, 0));
      const int rc = vo_new_point_candidates_enqueue_image(c, P.stage, width, width, height, c->early_bins, c->early_table);
      if (rc < 0) return rc;
      c->early_issued = rc == VO_OK ? 1 : 0;
    }
  }
  return vo_pyramid_build_pair(c, slot_l, c->slots[slot_l].stage, slot_r, c->slots[slot_r].stage, width, height, width);
}

// One host image without a host synchronisation (MonoVO's ingestion; vo_set_image waits for the device twice): H2D into the
// slot's staging plane and the pyramid chain on the ingest stream; consumers wait for the slot's event on the device.
int vo_set_image_host_async(vo_ctx *c, int slot, const uint8_t *host, int width, int height, int stride) {
  if (!c || !host) return VO_ERR_INVALID;
  if (slot < 0 || slot >= c->cfg.n_slots) VO_FAIL(c, VO_ERR_INVALID, "slot out of range");
  if (width <= 0 || height <= 0 || width > c->cfg.max_width || height > c->cfg.max_height || stride < width)
    VO_FAIL(c, VO_ERR_CAPACITY, "image %dx%d exceeds vo_config %dx%d", width, height, c->cfg.max_width, c->cfg.max_height);
  VO_CHECK_HIP(c, hipSetDevice(c->device));
  vo_ingest_scope ingest(c);
  hipStream_t s = c->stream;
  vo_pyramid &P = c->slots[slot];
  if (!P.stage) VO_CHECK_HIP(c, vo_dev_malloc(c, (void **)&P.stage, (size_t)c->cfg.max_width * c->cfg.max_height));
  if (stride == width)
    VO_CHECK_HIP(c, hipMemcpyAsync(P.stage, host, (size_t)width * height, hipMemcpyHostToDevice, s));
  else
    VO_CHECK_HIP(c, hipMemcpy2DAsync(P.stage, (size_t)width, host, (size_t)stride, (size_t)width, (size_t)height, hipMemcpyHostToDevice, s));
  if (c->early_bins && s != c->stream2) {  // (MonoVO's synchronous call: the detection follows the upload on the side stream)
    VO_CHECK_HIP(c, hipEventRecord(c->ev_fork, s));
    VO_CHECK_HIP(c, hipStreamWaitEvent(c->stream2, c->ev_fork, 0));
    const int rc = vo_new_point_candidates_enqueue_image(c, P.stage, width, width, height, c->early_bins, c->early_table);
    if (rc < 0) return rc;
    c->early_issued = rc == VO_OK ? 1 : 0;
  }
  return vo_pyramid_build(c, slot, P.stage, width, height, width);
}

extern "C" int vo_set_pyramid_window_hint(vo_ctx *c, int win) {
  if (!c || win < 0) return VO_ERR_INVALID;
  c->pyr_win_hint = win;
  return VO_OK;
}

extern "C" int vo_swap_slots(vo_ctx *c, int a, int b) {
  if (!c || a < 0 || b < 0 || a >= c->cfg.n_slots || b >= c->cfg.n_slots) return VO_ERR_INVALID;
  // the "slot is read by the frame in flight" guard of the pyramid build goes by slot index: a swap under a frame in
  // flight would let a side-stream rebuild overwrite memory that frame still reads
  if (c->frame_slots_busy)
    for (int k = 0; k < 3; ++k)
      if (c->frame_slot[k] == a || c->frame_slot[k] == b)
        VO_FAIL(c, VO_ERR_INVALID, "slot %d is read by the frame in flight: collect its result before swapping", c->frame_slot[k]);
  vo_pyramid t = c->slots[a];
  c->slots[a] = c->slots[b];
  c->slots[b] = t;
  return VO_OK;
}

extern "C" int vo_get_level(vo_ctx *c, int slot, int level, uint8_t *host, int *width, int *height) {
  if (!c || !host || slot < 0 || slot >= c->cfg.n_slots) return VO_ERR_INVALID;
  const vo_pyramid &P = c->slots[slot];
  if (level < 0 || level >= P.n_levels) VO_FAIL(c, VO_ERR_INVALID, "level %d not built (%d levels)", level, P.n_levels);
  const vo_level &L = P.lv[level];
  VO_CHECK_HIP(c, hipSetDevice(c->device));
  if (vo_slot_acquire(c, slot) < 0) return VO_ERR_HIP;
  VO_CHECK_HIP(c, hipMemcpy2DAsync(host, (size_t)L.w, L.origin(), (size_t)L.stride, (size_t)L.w, (size_t)L.h,
                                   hipMemcpyDeviceToHost, c->stream));
  SYNC();
  if (width) *width = L.w;
  if (height) *height = L.h;
  return VO_OK;
}

// ---- pyramidal LK ----------------------------------------------------------------
extern "C" int vo_klt_track(vo_ctx *c, int slot0, int slot1, const float *pts0, float *pts1, int n, int win,
                            int max_level, int flags, int max_iter, double eps, float min_eig_thr,
                            uint8_t *status, float *err) {
  if (!c || !pts0 || !pts1 || !status || !err) return VO_ERR_INVALID;
  int rc = check_n(c, n);
  if (rc) return rc;
  if (n == 0) return 0;
  VO_CHECK_HIP(c, hipSetDevice(c->device));
  H2D(c->d_pts0, pts0, sizeof(float) * 2 * (size_t)n);
  if (flags & VO_KLT_USE_INITIAL_FLOW)
    H2D(c->d_pts1, pts1, sizeof(float) * 2 * (size_t)n);
  else
    VO_CHECK_HIP(c, hipMemsetAsync(c->d_pts1, 0, sizeof(float) * 2 * (size_t)n, c->stream));
  rc = vo_klt_enqueue(c, slot0, slot1, c->d_pts0, nullptr, c->d_pts1, n, nullptr, win, max_level, flags, max_iter, eps,
                      min_eig_thr, c->d_status, c->d_err);
  if (rc < 0) return rc;
  D2H(pts1, c->d_pts1, sizeof(float) * 2 * (size_t)n);
  D2H(status, c->d_status, (size_t)n);
  D2H(err, c->d_err, sizeof(float) * (size_t)n);
  SYNC();
  return rc;
}

// shared body of the four FeatureTracker wrappers.
//  mode 0 track, 1 trackWithPrior, 2 trackBidirection, 3 trackBidirectionWithPrior
static int track_common(vo_ctx *c, int mode, int slot0, int slot1, const float *pts0, int n, int win,
                        int max_level, float thres_err, float thres_bidir, float *pts_track,
                        uint8_t *mask_valid) {
  if (!c || !pts0 || !pts_track || !mask_valid) return VO_ERR_INVALID;
  int rc = check_n(c, n);
  if (rc) return rc;
  if (n == 0) return VO_OK;
  VO_CHECK_HIP(c, hipSetDevice(c->device));
  const vo_pyramid &P0 = c->slots[slot0 >= 0 && slot0 < c->cfg.n_slots ? slot0 : 0];
  const size_t pb = sizeof(float) * 2 * (size_t)n;
  H2D(c->d_pts0, pts0, pb);
  H2D(c->d_mask, mask_valid, (size_t)n);
  const bool prior = (mode == 1 || mode == 3);
  if (prior)
    H2D(c->d_pts1, pts_track, pb);
  else
    VO_CHECK_HIP(c, hipMemsetAsync(c->d_pts1, 0, pb, c->stream));
  // forward: track()/trackBidirection() use the OpenCV defaults (30, 0.01, minEig 1e-4);
  // the *WithPrior variants pass `{}` criteria and `{}` minEigThreshold (= 0)
  rc = vo_klt_enqueue(c, slot0, slot1, c->d_pts0, nullptr, c->d_pts1, n, nullptr, win, max_level,
                      prior ? VO_KLT_USE_INITIAL_FLOW : 0, 30, 0.01, prior ? 0.f : 1e-4f, c->d_status, c->d_err);
  if (rc < 0) return rc;
  if (mode >= 2) {
    // backward: pts0_backward starts as a copy of pts0; trackBidirection uses maxLevel-1
    VO_CHECK_HIP(c, hipMemcpyAsync(c->d_pts2, c->d_pts0, pb, hipMemcpyDeviceToDevice, c->stream));
    const int ml = (mode == 2) ? max_level - 1 : max_level;
    rc = vo_klt_enqueue(c, slot1, slot0, c->d_pts1, nullptr, c->d_pts2, n, nullptr, win, ml, VO_KLT_USE_INITIAL_FLOW, 0,
                        0., 0.f, c->d_status2, c->d_err2);
    if (rc < 0) return rc;
  }
  rc = vo_klt_mask_enqueue(c, mode == 0 ? 0 : (mode == 1 ? 1 : mode), n, nullptr, P0.w, P0.h, thres_err,
                           thres_bidir, c->d_pts0, c->d_pts1, c->d_pts2, c->d_status, c->d_status2, c->d_err,
                           c->d_err2, c->d_mask, c->d_mask);
  if (rc < 0) return rc;
  D2H(pts_track, c->d_pts1, pb);
  D2H(mask_valid, c->d_mask, (size_t)n);
  SYNC();
  return VO_OK;
}

extern "C" int vo_track(vo_ctx *c, int slot0, int slot1, const float *pts0, int n, int win, int max_level,
                        float thres_err, float *pts_track, uint8_t *mask_valid) {
  return track_common(c, 0, slot0, slot1, pts0, n, win, max_level, thres_err, 0.f, pts_track, mask_valid);
}
extern "C" int vo_track_with_prior(vo_ctx *c, int slot0, int slot1, const float *pts0, int n, int win,
                                   int max_level, float thres_err, float *pts_track, uint8_t *mask_valid) {
  return track_common(c, 1, slot0, slot1, pts0, n, win, max_level, thres_err, 0.f, pts_track, mask_valid);
}
extern "C" int vo_track_bidirection(vo_ctx *c, int slot0, int slot1, const float *pts0, int n, int win,
                                    int max_level, float thres_err, float thres_bidirection, float *pts_track,
                                    uint8_t *mask_valid) {
  return track_common(c, 2, slot0, slot1, pts0, n, win, max_level, thres_err, thres_bidirection, pts_track,
                      mask_valid);
}
extern "C" int vo_track_bidirection_with_prior(vo_ctx *c, int slot0, int slot1, const float *pts0, int n, int win,
                                               int max_level, float thres_err, float thres_bidirection,
                                               float *pts_track, uint8_t *mask_valid) {
  return track_common(c, 3, slot0, slot1, pts0, n, win, max_level, thres_err, thres_bidirection, pts_track,
                      mask_valid);
}
