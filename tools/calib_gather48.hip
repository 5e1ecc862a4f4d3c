// Lab: rate of the sweep's record gathers for 64-byte records (4 x global_load_lds_dwordx4 per 64 records, one quad per
// record) against 48-byte records (3 instructions per 64 records, piece g = 64 r + lane -> record g / 3), table resident in L2
// (2 MB) or beyond (64 MB); indices random or in runs of 8 consecutive records (what a cell-ordered neighbour list looks like).
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/calib_gather48 tools/calib_gather48.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <random>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int RECB>
__global__ __launch_bounds__(256) void k_gather(const char *__restrict__ table, const int *__restrict__ idx, long long ntrips, int *out) {
  extern __shared__ char lds[];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  char *tile = lds + wv * 2 * 4096;
  long long trip = (long long)blockIdx.x * 4 + wv;
  int acc = 0;
  constexpr int NI = RECB / 16;   // instructions per trip: 4 (64 B) or 3 (48 B)
  int slot[NI], piece[NI];
#pragma unroll
  for (int r = 0; r < NI; r++) {
    const int g = 64 * r + lane;
    slot[r] = RECB == 64 ? (g >> 2) : ((g * 171) >> 9);
    piece[r] = g - slot[r] * NI;
  }
  int buf = 0;
  for (; trip < ntrips; trip += (long long)gridDim.x * 4) {
    const int mine = idx[trip * 64 + lane];
    char *t = tile + buf * 4096;
#pragma unroll
    for (int r = 0; r < NI; r++) {
      const int rec = __shfl(mine, slot[r], 64);
      const char *g = table + (size_t)rec * RECB + piece[r] * 16;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                       (__attribute__((address_space(3))) void *)(t + r * 1024), 16, 0, 0);
    }
    __builtin_amdgcn_s_waitcnt(0);
    acc += *(volatile int *)(t + lane * RECB);
    buf ^= 1;
  }
  if (acc == 0x12345) *out = acc;
}

int main() {
  int *out; CK(hipMalloc(&out, 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  std::mt19937_64 rng(1);
  for (long long nrec : {32768ll, 1ll << 20}) {          // 2 MB / 64 MB of 64-byte records
    for (int runs : {1, 8}) {
      const long long nidx = 64ll << 20;                   // 64 M gathered records per launch
      std::vector<int> h(nidx);
      for (long long k = 0; k < nidx; k += runs) {
        const int b = (int)(rng() % (unsigned long long)(nrec - runs));
        for (int u = 0; u < runs && k + u < nidx; u++) h[k + u] = b + u;
      }
      int *idx; CK(hipMalloc(&idx, nidx * 4)); CK(hipMemcpy(idx, h.data(), nidx * 4, hipMemcpyHostToDevice));
      char *t64, *t48; CK(hipMalloc(&t64, nrec * 64)); CK(hipMalloc(&t48, nrec * 48 + 64));
      CK(hipMemset(t64, 1, nrec * 64)); CK(hipMemset(t48, 1, nrec * 48 + 64));
      const long long ntrips = nidx / 64;
      for (int which = 0; which < 2; which++) {
        float best = 1e30f;
        for (int rep = 0; rep < 3; rep++) {
          CK(hipEventRecord(e0));
          if (which == 0) k_gather<64><<<2560, 256, 4 * 2 * 4096>>>(t64, idx, ntrips, out);
          else k_gather<48><<<2560, 256, 4 * 2 * 4096>>>(t48, idx, ntrips, out);
          CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
          float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best;
        }
        printf("table %5.1f MB  runs of %d  record %2d B: %7.3f ms  %6.1f G records/s  %6.2f TB/s of records\n",
               nrec * (which ? 48 : 64) / 1e6, runs, which ? 48 : 64, best, nidx / best / 1e6, nidx * (which ? 48.0 : 64.0) / best / 1e9);
      }
      CK(hipFree(idx)); CK(hipFree(t64)); CK(hipFree(t48));
    }
  }
  return 0;
}
