// batch.hip — batch-of-sequences mode on ONE device: S independent image streams, each with its own context (HIP streams,
// pyramid slots, per-stream landmark / frame id counters — SURVEY F11), its own StereoVO and its own host thread.
// BASELINE's batch mode proper is one stream per GPU (bench.py --gpus N); a single sequential stream is latency-bound and
// leaves most of an MI355X idle, so a recorded-dataset server runs several per device. The streams share nothing: no
// collective, no lock (the library keeps no mutable global state on this path).
#include <atomic>
#include <chrono>
#include <stdlib.h>
#include <thread>
#include <vector>

#include "stereo_vo.hpp"

struct vo_batch {
  int device = 0, n = 0, strict_border = 0;  // strict_border: the replay arrangement the streams run with
  std::vector<vo_ctx *> ctx;
  std::vector<vo_svo *> svo;
  char err[512] = {0};
};

extern "C" void vo_batch_destroy(vo_batch *b) {
  if (!b) return;
  for (vo_svo *s : b->svo)
    if (s) vo_svo_destroy(s);
  for (vo_ctx *c : b->ctx)
    if (c) vo_destroy(c);
  delete b;
}

extern "C" const char *vo_batch_last_error(const vo_batch *b) { return b ? b->err : "null batch"; }

extern "C" int vo_batch_strict_border(const vo_batch *b) { return b ? b->strict_border : VO_ERR_INVALID; }

extern "C" int vo_batch_create(const vo_config *cfg, const vo_svo_params *prm, int n_streams, vo_batch **out) {
  if (!cfg || !prm || !out || n_streams <= 0 || n_streams > 64) return VO_ERR_INVALID;
  *out = nullptr;
  vo_batch *b = new vo_batch();
  b->device = cfg->device;
  b->n = n_streams;
  b->ctx.assign(n_streams, nullptr);
  b->svo.assign(n_streams, nullptr);
  // The replay arrangements that run NEXT TO the frame kernel (strict-border 3, 4, 5) rely on two queues of one context
  // making progress together. With several contexts on the device the runtime multiplexes their HIP streams onto a few
  // hardware queues, the device-side joins time out (0.1 s each) and the frames are re-issued: correct, but slow. More
  // than one stream: the stream-ordered replay (same results by construction).
  vo_svo_params q = *prm;
  if (n_streams > 1 && q.strict_border >= 3) q.strict_border = 1;
  b->strict_border = q.strict_border;
  for (int s = 0; s < n_streams; ++s) {
    int rc = vo_create(cfg, &b->ctx[s]);
    if (rc == VO_OK) rc = vo_svo_create(b->ctx[s], &q, &b->svo[s]);
    if (rc != VO_OK) {
      vo_batch_destroy(b);
      return rc;
    }
  }
  *out = b;
  return VO_OK;
}

// Every stream s tracks its n_frames pairs left[s * n_frames + k], right[...] (device pointers when on_device) in order,
// handing pair k+1 over while frame k is in flight. The first `warmup` frames of every stream are outside the timed region
// (all threads meet at a barrier behind them). T_wc (may be NULL): [n_streams][n_frames][16]; last_ids (may be NULL):
// [n_streams][ids_cap] the ids of every stream's final track set, n_ids[s] their number; seconds[s]: wall time of stream
// s's timed frames; *wall: first start to last end over all streams.
extern "C" int vo_batch_debug_set(vo_batch *b, int key, int value) {
  if (!b) return VO_ERR_INVALID;
  for (int s = 0; s < b->n; ++s) {
    const int rc = vo_debug_set(b->ctx[s], key, value);
    if (rc < 0) return rc;
  }
  return VO_OK;
}

extern "C" int vo_batch_run(vo_batch *b, const void *const *left, const void *const *right, int n_frames, int stride,
                            int on_device, int warmup, float *T_wc, int32_t *last_ids, int ids_cap, int *n_ids,
                            double *seconds, double *wall) {
  if (!b || !left || !right || n_frames <= 0 || warmup < 0 || warmup >= n_frames) return VO_ERR_INVALID;
  const int S = b->n;
  std::vector<int> rcs(S, VO_OK);
  std::vector<double> t_begin(S, 0.0), t_end(S, 0.0);
  std::atomic<int> arrived{0};
  auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  auto body = [&](int s) {
    vo_svo *svo = b->svo[s];
    const void *const *L = left + (size_t)s * n_frames, *const *R = right + (size_t)s * n_frames;
    vo_svo_frame_info info;
    for (int k = 0; k < n_frames; ++k) {
      if (k == warmup) {  // every stream has finished its untimed frames: start together
        arrived.fetch_add(1);
        while (arrived.load() < S) std::this_thread::yield();
        t_begin[s] = now();
      }
      int rc = vo_svo_enqueue(svo, L[k], R[k], stride, on_device, 0.1 * k);
      if (rc >= 0 && k + 1 < n_frames) rc = vo_svo_prefetch(svo, L[k + 1], R[k + 1], stride, on_device);
      if (rc >= 0) rc = vo_svo_result(svo, &info);
      if (rc < 0) {
        rcs[s] = rc;
        if (k < warmup) {  // (do not leave the others waiting at the barrier)
          arrived.fetch_add(1);
        }
        return;
      }
      if (T_wc) memcpy(T_wc + ((size_t)s * n_frames + k) * 16, info.T_wc, sizeof(float) * 16);
    }
    t_end[s] = now();
    if (last_ids && n_ids) {
      int n = 0;
      int rc = vo_svo_get_tracks(svo, nullptr, nullptr, nullptr, nullptr, nullptr, 0, &n);
      if (rc >= 0 && n <= ids_cap) rc = vo_svo_get_tracks(svo, last_ids + (size_t)s * ids_cap, nullptr, nullptr, nullptr, nullptr, ids_cap, &n);
      n_ids[s] = n;
      if (rc < 0) rcs[s] = rc;
    }
  };
  std::vector<std::thread> th;
  for (int s = 0; s < S; ++s) th.emplace_back(body, s);
  for (auto &t : th) t.join();
  double t0 = 1e300, t1 = 0.0;
  for (int s = 0; s < S; ++s) {
    if (rcs[s] < 0) {
      snprintf(b->err, sizeof(b->err), "stream %d: %s", s, vo_last_error(b->ctx[s]));
      return rcs[s];
    }
    if (seconds) seconds[s] = t_end[s] - t_begin[s];
    t0 = t_begin[s] < t0 ? t_begin[s] : t0;
    t1 = t_end[s] > t1 ? t_end[s] : t1;
  }
  if (wall) *wall = t1 - t0;
  return VO_OK;
}
