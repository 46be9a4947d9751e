// graphprobe.hip — diagnostic: cost of N dependent tiny launches on a stream vs one hipGraphLaunch
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>
__global__ void tiny(int *p, int k) { if (threadIdx.x == 0 && blockIdx.x == 0) p[k & 15] += k; }
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  int *d; hipMalloc(&d, 64); hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  hipEvent_t ev; hipEventCreateWithFlags(&ev, hipEventDisableTiming);
  const int N = 12, REP = 300;
  for (int r = 0; r < 20; ++r) { for (int k = 0; k < N; ++k) hipLaunchKernelGGL(tiny, dim3(64), dim3(64), 0, s, d, k); hipStreamSynchronize(s); }
  double t0 = now(), tenq = 0;
  for (int r = 0; r < REP; ++r) {
    double a = now();
    for (int k = 0; k < N; ++k) hipLaunchKernelGGL(tiny, dim3(64), dim3(64), 0, s, d, k);
    hipEventRecord(ev, s);
    tenq += now() - a;
    hipEventSynchronize(ev);
  }
  double t1 = now();
  printf("stream: %d launches: %.1f us per batch (host enqueue %.1f us)\n", N, (t1 - t0) / REP, tenq / REP);
  hipGraph_t g; hipGraphExec_t ge;
  hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
  for (int k = 0; k < N; ++k) hipLaunchKernelGGL(tiny, dim3(64), dim3(64), 0, s, d, k);
  hipStreamEndCapture(s, &g);
  hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  for (int r = 0; r < 20; ++r) { hipGraphLaunch(ge, s); hipStreamSynchronize(s); }
  t0 = now(); tenq = 0;
  for (int r = 0; r < REP; ++r) {
    double a = now();
    hipGraphLaunch(ge, s);
    hipEventRecord(ev, s);
    tenq += now() - a;
    hipEventSynchronize(ev);
  }
  t1 = now();
  printf("graph : %d nodes   : %.1f us per batch (host enqueue %.1f us)\n", N, (t1 - t0) / REP, tenq / REP);
  // single launch round trip
  t0 = now();
  for (int r = 0; r < REP; ++r) { hipLaunchKernelGGL(tiny, dim3(64), dim3(64), 0, s, d, 1); hipEventRecord(ev, s); hipEventSynchronize(ev); }
  t1 = now();
  printf("single launch + event sync round trip: %.1f us\n", (t1 - t0) / REP);
  t0 = now();
  for (int r = 0; r < REP; ++r) { hipLaunchKernelGGL(tiny, dim3(64), dim3(64), 0, s, d, 1); hipStreamSynchronize(s); }
  t1 = now();
  printf("single launch + stream sync round trip: %.1f us\n", (t1 - t0) / REP);
  return 0;
}
