// round 4: what a dependency between two HIP streams costs on this platform, against a dependent launch on ONE stream.
// hipcc --offload-arch=gfx950 -O2 -o tools/calib_xstream tools/calib_xstream.hip && tools/calib_xstream
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void k_tiny(int *p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1; }
int main() {
  int *d; CK(hipMalloc(&d, 64)); CK(hipMemset(d, 0, 64));
  hipStream_t s1, s2; CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  const int N = 500;
  hipEvent_t ev[2 * N]; for (auto &e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  for (int rep = 0; rep < 3; rep++) {
    CK(hipDeviceSynchronize());
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < 2 * N; i++) k_tiny<<<1, 64, 0, s1>>>(d);
    CK(hipStreamSynchronize(s1));
    auto t1 = std::chrono::steady_clock::now();
    for (int i = 0; i < N; i++) {
      k_tiny<<<1, 64, 0, s1>>>(d); CK(hipEventRecord(ev[2 * i], s1)); CK(hipStreamWaitEvent(s2, ev[2 * i], 0));
      k_tiny<<<1, 64, 0, s2>>>(d); CK(hipEventRecord(ev[2 * i + 1], s2)); CK(hipStreamWaitEvent(s1, ev[2 * i + 1], 0));
    }
    CK(hipStreamSynchronize(s1)); CK(hipStreamSynchronize(s2));
    auto t2 = std::chrono::steady_clock::now();
    const double same = std::chrono::duration<double, std::micro>(t1 - t0).count() / (2 * N);
    const double cross = std::chrono::duration<double, std::micro>(t2 - t1).count() / (2 * N);
    printf("dependent launch on one stream: %.2f us per launch; alternating between two streams through events: %.2f us per launch (+%.2f us per hand-over)\n", same, cross, cross - same);
  }
  return 0;
}
