/* sanitize_main.c — TEST INFRASTRUCTURE ONLY (see vo_oracle.h). Drives every entry point of the CPU oracle on small
 * seeded inputs; built by `make -C oracle sanitize` with -fsanitize=address,undefined -fno-sanitize-recover=all, so
 * that an out-of-bounds access, a misaligned load, a signed overflow or an invalid shift in the restatement ends the
 * run with a non-zero status (tests/test_oracle_sanitize.py). CPU only: GPU sanitizers are not available on this pool.
 * Sizes are chosen so that border paths run: features next to every image edge, windows larger than the top
 * pyramid level, empty sets. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "vo_oracle.h"

static unsigned lcg_state = 12345u;
static unsigned lcg(void) {
  lcg_state = lcg_state * 1664525u + 1013904223u;
  return lcg_state >> 8;
}
static float frand(void) { return (float)(lcg() & 0xffff) / 65536.0f; }

/* smooth texture + noise, shifted by (dx, dy): something KLT and the IC refinement can lock on */
static void make_image(uint8_t *img, int w, int h, float dx, float dy) {
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x) {
      const float u = x - dx, v = y - dy;
      float s = 128.f + 50.f * sinf(0.21f * u) * cosf(0.17f * v) + 40.f * sinf(0.05f * u + 0.09f * v) +
                20.f * sinf(0.6f * u) * sinf(0.45f * v);
      s = s < 0 ? 0 : (s > 255 ? 255 : s);
      img[y * w + x] = (uint8_t)s;
    }
}

#define CHECK(c)                                                  \
  do {                                                            \
    if (!(c)) {                                                   \
      fprintf(stderr, "sanitize_main: check failed: %s\n", #c);  \
      return 1;                                                   \
    }                                                             \
  } while (0)

int main(void) {
  enum { W = 160, H = 96, N = 60 };
  uint8_t *I0 = malloc(W * H), *I1 = malloc(W * H), *I2 = malloc(W * H);
  make_image(I0, W, H, 0.f, 0.f);
  make_image(I1, W, H, 1.6f, -0.8f);
  make_image(I2, W, H, 5.1f, -0.8f);
  float pts0[2 * N], pts1[2 * N], prior[2 * N], err[N], scale[N];
  uint8_t status[N], mask[N], touched[N];
  for (int i = 0; i < N; ++i) {
    /* a ring of points hugging the border (2 px .. 12 px from it) and some interior ones */
    const int k = i % 4;
    float x = 2.f + frand() * (W - 4.f), y = 2.f + frand() * (H - 4.f);
    if (i < 40) {
      if (k == 0) y = 2.f + frand() * 10.f;
      if (k == 1) y = H - 3.f - frand() * 10.f;
      if (k == 2) x = 2.f + frand() * 10.f;
      if (k == 3) x = W - 3.f - frand() * 10.f;
    }
    pts0[2 * i] = x;
    pts0[2 * i + 1] = y;
    prior[2 * i] = x + 1.2f;
    prior[2 * i + 1] = y - 0.5f;
    scale[i] = 0.9f + 0.2f * frand();
  }
  /* pyramids, derivative images */
  for (int win = 7; win <= 31; win += 8) {
    const int L = vo_ref_pyramid_levels(W, H, win, 6);
    CHECK(L >= 0 && L <= 6);
  }
  {
    int lw, lh;
    vo_ref_level_size(W, H, 1, &lw, &lh);
    uint8_t *d = malloc((size_t)lw * lh);
    vo_ref_pyr_down(I0, W, H, W, d, lw);
    int16_t *dxy = malloc(sizeof(int16_t) * 2 * W * H);
    vo_ref_scharr(I0, W, H, W, dxy);
    float *du = malloc(sizeof(float) * W * H), *dv = malloc(sizeof(float) * W * H);
    vo_ref_sobel3(I0, W, H, W, du, dv);
    free(d);
    free(dxy);
    free(du);
    free(dv);
  }
  /* PyrLK in every flag combination, several windows, 1 and 3 threads */
  for (int win = 7; win <= 31; win += 6)
    for (int fl = 0; fl <= 4; fl += 4) {
      memcpy(pts1, prior, sizeof(pts1));
      CHECK(vo_ref_calc_optical_flow_pyr_lk(I0, I1, W, H, W, pts0, pts1, N, win, 4, fl, 30, 0.01, fl ? 0.f : 1e-4f, status,
                                            err, 1 + 2 * (win & 1)) >= 0);
    }
  CHECK(vo_ref_calc_optical_flow_pyr_lk(I0, I1, W, H, W, pts0, pts1, 0, 21, 4, 0, 30, 0.01, 1e-4f, status, err, 1) >= 0);
  /* tracker front-ends */
  memset(mask, 1, sizeof(mask));
  CHECK(vo_ref_track(I0, I1, W, H, W, pts0, N, 21, 3, 80.f, pts1, mask, 2) >= 0);
  memset(mask, 1, sizeof(mask));
  CHECK(vo_ref_track_bidirection(I0, I1, W, H, W, pts0, N, 15, 3, 80.f, 0.5f, pts1, mask, 2) >= 0);
  memset(mask, 1, sizeof(mask));
  memcpy(pts1, prior, sizeof(pts1));
  CHECK(vo_ref_track_bidirection_with_prior(I0, I1, W, H, W, pts0, N, 15, 3, 80.f, 1.0f, pts1, mask, 2) >= 0);
  memset(mask, 1, sizeof(mask));
  memcpy(pts1, prior, sizeof(pts1));
  CHECK(vo_ref_track_with_prior(I0, I1, W, H, W, pts0, N, 21, 3, 80.f, pts1, mask, 2) >= 0);
  /* IC refinement: all border modes x summation orders (the REFERENCE mode carries state across points) */
  for (int bm = 0; bm <= 1; ++bm)
    for (int sm = 0; sm <= 1; ++sm) {
      float tr[2 * N];
      memcpy(tr, pts1, sizeof(tr));
      memset(mask, 1, sizeof(mask));
      CHECK(vo_ref_track_with_scale(I0, I1, W, H, W, pts0, scale, N, tr, mask, bm, sm, touched) >= 0);
    }
  /* GN, LDLT, se3 */
  {
    enum { M = 200 };
    float X[3 * M], pl[2 * M], pr[2 * M];
    const float K[4] = {300.f, 300.f, 80.f, 48.f};
    float T_lr[16] = {1, 0, 0, 0.5f, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    for (int i = 0; i < M; ++i) {
      X[3 * i] = -4.f + 8.f * frand();
      X[3 * i + 1] = -2.f + 4.f * frand();
      X[3 * i + 2] = 4.f + 20.f * frand();
      pl[2 * i] = K[0] * X[3 * i] / X[3 * i + 2] + K[2] + frand();
      pl[2 * i + 1] = K[1] * X[3 * i + 1] / X[3 * i + 2] + K[3] + frand();
      pr[2 * i] = K[0] * (X[3 * i] - 0.5f) / X[3 * i + 2] + K[2] + frand();
      pr[2 * i + 1] = pl[2 * i + 1];
    }
    uint8_t inl[M];
    vo_ref_gn_info info;
    for (int sm = 0; sm <= 1; ++sm) {
      float T01[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
      CHECK(vo_ref_gn_pose_stereo(X, pl, pr, M, K, K, T_lr, 3.f, T01, inl, sm, 512, &info) >= 0);
      for (int var = 0; var <= 1; ++var) {
        float R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, t[3] = {0, 0, 0};
        CHECK(vo_ref_gn_pose_mono(X, pl, M, K, 3, R, t, inl, var, sm, 512, &info) >= 0);
      }
    }
    float T01[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    CHECK(vo_ref_gn_pose_stereo(X, pl, pr, 0, K, K, T_lr, 3.f, T01, inl, 0, 0, &info) >= 0); /* empty BA set */
    const float xi[6] = {0.1f, -0.2f, 0.3f, 0.01f, 0.02f, -0.03f}, xi0[6] = {1, 2, 3, 0, 0, 0};
    float T[16], Ti[16];
    vo_ref_se3_exp(xi, T);
    vo_ref_se3_exp(xi0, T);
    vo_ref_inverse_se3(T, Ti);
    vo_ref_inverse4x4(T, Ti);
    /* the stereo and mono frames (priors, all four tracker calls, IC, GN, gates, new points, flag bytes) */
    vo_ref_stereo_params sp;
    memset(&sp, 0, sizeof(sp));
    sp.width = W;
    sp.height = H;
    sp.win = 21;
    sp.max_level = 3;
    sp.thres_err = 80.f;
    sp.thres_bidirection = 0.5f;
    sp.thres_poseba = 3.f;
    sp.thres_sampson = 60.f;
    memcpy(sp.Kl, K, sizeof(K));
    memcpy(sp.Kr, K, sizeof(K));
    memcpy(sp.T_lr, T_lr, sizeof(T_lr));
    float Xp[3 * N], pl1[2 * N], pr1[2 * N], newr[2 * 8], dT[16];
    uint8_t stage[N], mnew[8], fl[N];
    for (int i = 0; i < N; ++i) {
      Xp[3 * i + 2] = 6.f + 10.f * frand();
      Xp[3 * i] = (pts0[2 * i] - K[2]) / K[0] * Xp[3 * i + 2];
      Xp[3 * i + 1] = (pts0[2 * i + 1] - K[3]) / K[1] * Xp[3 * i + 2];
      pr1[2 * i] = pts0[2 * i] - K[0] * 0.5f / Xp[3 * i + 2];
      pr1[2 * i + 1] = pts0[2 * i + 1];
      fl[i] = (uint8_t)((i % 3 != 0) | ((i % 17 == 0) << 1));
    }
    const float dTp[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0.1f, 0, 0, 0, 1};
    vo_ref_frame_counts fc;
    for (int bm = 0; bm <= 1; ++bm)
      for (int withfl = 0; withfl <= 1; ++withfl) {
        float pr1c[2 * N];
        memcpy(pr1c, pr1, sizeof(pr1c));
        const int rc = vo_ref_stereo_frame(&sp, I0, I1, I2, W, pts0, Xp, withfl ? fl : NULL, N, dTp, pts0, 8, bm, 512, bm, 2, pl1,
                                           pr1c, stage, dT, newr, mnew, &fc);
        CHECK(rc == 0 || rc == -6);
      }
    CHECK(vo_ref_stereo_frame(&sp, I0, I1, I2, W, pts0, Xp, NULL, 0, dTp, pts0, 0, 0, 0, 0, 1, pl1, pr1, stage, dT, newr, mnew,
                              &fc) == 0);
    vo_ref_mono_params mp;
    memset(&mp, 0, sizeof(mp));
    mp.width = W;
    mp.height = H;
    mp.win = 15;
    mp.max_level = 3;
    mp.thres_err = 20.f;
    mp.thres_bidirection = 1.f;
    mp.thres_poseba = 5;
    mp.thres_sampson = 1.f;
    memcpy(mp.K, K, sizeof(K));
    const float I4[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    float sc[N];
    vo_ref_mono_counts mc;
    for (int i = 0; i < N; ++i) fl[i] = (uint8_t)((i & 3) | ((i % 13 == 0) << 2));
    CHECK(vo_ref_mono_frame(&mp, I0, I1, W, pts0, Xp, fl, N, I4, I4, I4, 0, 0, 0, 2, pl1, sc, stage, dT, &mc) >= 0);
    CHECK(vo_ref_mono_frame(&mp, I0, I1, W, pts0, Xp, fl, N, I4, I4, I4, 1, 512, 1, 2, pl1, sc, stage, dT, &mc) >= 0);
    /* epipolar distances, priors */
    float F[9], d[N];
    const float R10[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, t10[3] = {0.1f, 0.02f, 1.f};
    vo_ref_fundamental_from_pose(K, R10, t10, F);
    vo_ref_sampson_distance(pts0, pts1, N, F, d);
    vo_ref_symmetric_epipolar_distance(pts0, pts1, N, F, d);
    const float K9[9] = {K[0], 0, K[2], 0, K[1], K[3], 0, 0, 1};
    vo_ref_calc_prior(pts0, N, Xp, N - 7, I4, K9, pl1);
  }
  /* Hamming, compaction, bucketing */
  {
    enum { NA = 37, NB = 53 };
    uint8_t a[32 * NA], b[32 * NB];
    for (int i = 0; i < 32 * NA; ++i) a[i] = (uint8_t)lcg();
    for (int i = 0; i < 32 * NB; ++i) b[i] = (uint8_t)lcg();
    uint16_t dist[NA * NB], bd[NA], sd[NA];
    int32_t bi[NA];
    vo_ref_hamming_matrix(a, NA, b, NB, dist);
    vo_ref_hamming_match(a, NA, b, NB, 100, 0.8f, bi, bd, sd);
    CHECK(vo_ref_descriptor_distance(a, a) == 0);
    uint8_t m[N], al[N], tk[N], tko[N];
    int32_t idx[N];
    for (int i = 0; i < N; ++i) {
      m[i] = lcg() & 1;
      al[i] = (lcg() & 7) != 0;
      tk[i] = (lcg() & 7) != 0;
    }
    CHECK(vo_ref_compact_indices(m, al, tk, N, idx, tko) <= N);
    CHECK(vo_ref_compact_indices(m, NULL, NULL, 0, idx, NULL) == 0);
    int us, vs;
    float ius, ivs;
    vo_ref_weight_bin_init(W, H, 8, 4, &us, &vs, &ius, &ivs);
    int32_t wgt[32], oi[32];
    vo_ref_weight_bin_update(pts0, N, us, vs, 8, 4, wgt);
    float resp[N], out[2 * 32];
    for (int i = 0; i < N; ++i) resp[i] = frand();
    CHECK(vo_ref_bucket_argmax(pts1, resp, N, ius, ivs, 8, 4, wgt, out, oi) <= 32);
  }
  /* keypoint detection (8 levels at 1.2: the top levels are a few dozen pixels wide) and rectification */
  {
    enum { MAXKP = 4096 };
    float *xy = malloc(sizeof(float) * 2 * MAXKP), *rs = malloc(sizeof(float) * MAXKP);
    int32_t *oc = malloc(sizeof(int32_t) * MAXKP);
    uint8_t *lv = malloc((size_t)W * H * 8);
    for (int nf = 0; nf <= 10000; nf += 5000)
      CHECK(vo_ref_orb_detect(I0, W, H, W, nf ? nf : 50, 1.2, 8, 31, 15, xy, rs, oc, MAXKP, lv) >= 0);
    CHECK(vo_ref_orb_detect(I0, W, H, W, 10000, 1.2, 8, 31, 15, xy, rs, oc, 3, NULL) >= -1); /* list too small */
    uint8_t *sc = malloc((size_t)W * H);
    vo_ref_fast_score_image(I1, W, H, W, 15, sc);
    uint8_t *rz = malloc(133 * 80);
    vo_ref_resize_linear_exact_u8(I0, W, H, W, rz, 133, 80);
    const float K[4] = {150.f, 151.f, 80.f, 48.f}, Kr[4] = {152.f, 149.f, 79.f, 49.f};
    const float Dl[5] = {-0.2f, 0.05f, 0.001f, -0.001f, 0.f}, Dr[5] = {-0.19f, 0.04f, -0.001f, 0.001f, 0.f};
    const float T_lr[16] = {0.9998f, 0.01f, -0.017f, 0.3f, -0.01f, 0.99995f, 0.f, 0.002f, 0.017f, 0.0002f, 0.99985f, -0.001f, 0, 0, 0, 1};
    float *mu = malloc(sizeof(float) * W * H), *mv = malloc(sizeof(float) * W * H);
    float *mu2 = malloc(sizeof(float) * W * H), *mv2 = malloc(sizeof(float) * W * H);
    vo_ref_image_undistort_maps(W, H, K, Dl, mu, mv);
    float Kc[4], Tc[16];
    vo_ref_stereo_rectify_maps(W, H, K, Dl, Kr, Dr, T_lr, mu, mv, mu2, mv2, Kc, Tc);
    mu[5] = NAN; /* cv::remap's corner cases: NaN, far outside, straddling the border */
    mu[6] = 1e9f;
    mu[7] = -0.5f;
    mv[8] = H - 0.5f;
    uint8_t *dst = malloc((size_t)W * H);
    vo_ref_remap_linear_u8(I0, W, H, W, mu, mv, W, H, dst);
    free(xy); free(rs); free(oc); free(lv); free(sc); free(rz); free(mu); free(mv); free(mu2); free(mv2); free(dst);
  }
  /* local BA: 4 keyframes (2 optimised), 30 landmarks, mono and stereo */
  for (int stereo = 0; stereo <= 1; ++stereo) {
    enum { NF = 4, NP = 30 };
    vo_ref_sba_dims d;
    memset(&d, 0, sizeof(d));
    d.n_frames = NF;
    d.n_opt = 2;
    d.n_points = NP;
    d.stereo = stereo;
    d.max_iter = 3;
    const double K[4] = {300., 300., 80., 48.};
    memcpy(d.Kl, K, sizeof(K));
    memcpy(d.Kr, K, sizeof(K));
    const double Tlr[16] = {1, 0, 0, 0.05, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    memcpy(d.T_lr, Tlr, sizeof(Tlr));
    d.thres_huber = 0.5;
    double T[NF * 16], X[NP * 3];
    int opt[NF] = {-1, -1, 0, 1};
    for (int f = 0; f < NF; ++f) {
      const double I4[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
      memcpy(T + 16 * f, I4, sizeof(I4));
      T[16 * f + 11] = -0.08 * f; /* T_jw: camera moves forward */
    }
    int ptr[NP + 1], nobs = 0;
    int *of = malloc(sizeof(int) * NP * NF * 2);
    uint8_t *orr = malloc(NP * NF * 2);
    double *px = malloc(sizeof(double) * 2 * NP * NF * 2);
    for (int i = 0; i < NP; ++i) {
      X[3 * i] = -0.4 + 0.8 * frand();
      X[3 * i + 1] = -0.2 + 0.4 * frand();
      X[3 * i + 2] = 0.6 + 2.0 * frand();
      ptr[i] = nobs;
      for (int f = 0; f < NF; ++f) {
        if ((i + f) % 5 == 0) continue;
        for (int r = 0; r <= stereo; ++r) {
          const double z = X[3 * i + 2] + T[16 * f + 11], x = X[3 * i] - (r ? 0.05 : 0.0);
          of[nobs] = f;
          orr[nobs] = (uint8_t)r;
          px[2 * nobs] = K[0] * x / z + K[2] + 0.3 * frand();
          px[2 * nobs + 1] = K[1] * X[3 * i + 1] / z + K[3] + 0.3 * frand();
          ++nobs;
        }
      }
    }
    ptr[NP] = nobs;
    d.n_obs = nobs;
    double avg[3];
    CHECK(vo_ref_sba_solve(&d, T, opt, X, ptr, of, orr, px, avg) >= 0);
    free(of);
    free(orr);
    free(px);
  }
  free(I0);
  free(I1);
  free(I2);
  printf("oracle sanitize run ok\n");
  { /* oracle_vo.c: DLT through the restated JacobiSVD (degenerate inputs included), keyframe reconstruction, pose products */
    const float K[4] = {300.f, 300.f, 80.f, 48.f};
    float T_rl[16] = {1, 0, 0, -0.5f, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    float pl[2 * 8], pr[2 * 8], Xw[3 * 8], Xl[3 * 8];
    uint8_t mk[8], acc[8], set[8];
    for (int i = 0; i < 8; ++i) {
      pl[2 * i] = 20.f + 15.f * i;
      pl[2 * i + 1] = 10.f + 9.f * i;
      pr[2 * i] = pl[2 * i] - (i == 3 ? 0.f : (i == 5 ? -4.f : 2.5f + i)); /* zero and negative disparity too */
      pr[2 * i + 1] = pl[2 * i + 1] + 0.1f * i;
      mk[i] = (uint8_t)(i != 6);
      Xw[3 * i] = Xw[3 * i + 1] = Xw[3 * i + 2] = 0.f;
    }
    pl[0] = NAN; /* a non-finite DLT matrix: Eigen reports InvalidInput */
    (void)vo_ref_new_landmark_accept(pl, pr, mk, 8, T_rl, K, K, acc, Xl);
    (void)vo_ref_keyframe_reconstruct(pl, pr, 8, T_rl, K, K, T_rl, Xw, set);
    (void)vo_ref_keyframe_reconstruct(pl, pr, 8, T_rl, K, K, NULL, Xw, set);
    float A[16], V[16], sv[4];
    memset(A, 0, sizeof(A));
    (void)vo_ref_jacobi_svd4(A, V, sv); /* the zero matrix */
    vo_ref_mul44(T_rl, T_rl, A);
    (void)vo_ref_jacobi_svd4(A, V, sv);
  }
  return 0;
}
