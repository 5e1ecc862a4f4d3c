// Calibration of rocprofv3's FETCH_SIZE on gfx950 for the access shapes of the dipole sweep (MI355X_MICROARCH.md, HBM:
// "other access widths are uncalibrated: calibrate on a known byte count in your own access pattern").
//   k_stream   : 16 B per lane, coalesced (the sweep's index stream)                      -- known bytes = N * 16
//   k_gather64 : 64-byte records fetched by quads of lanes with global_load_lds_dwordx4,
//                every record of the table exactly once, in a random order (the sweep's gathers, all misses)
//   k_gather64p: the same, but neighbouring quads of one instruction fetch the two halves of one 128-byte line
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/calib_fetch tools/calib_fetch.hip ; run under rocprofv3 --pmc FETCH_SIZE
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <random>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ void k_stream(const int4 *__restrict__ src, long long n, int *out) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  int acc = 0;
  for (; i < n; i += (long long)gridDim.x * blockDim.x) { const int4 v = src[i]; acc += v.x ^ v.y ^ v.z ^ v.w; }
  if (acc == 0x12345) *out = acc;
}

// one wave per 64 records per trip: lane 4q+k fetches piece k of record idx[base + q (+16 r)] -- four instructions per trip
__global__ __launch_bounds__(256) void k_gather64(const char *__restrict__ table, const int *__restrict__ idx, long long ntrips, int *out) {
  extern __shared__ char lds[];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  char *tile = lds + wv * 4096;
  long long trip = (long long)blockIdx.x * 4 + wv;
  int acc = 0;
  for (; trip < ntrips; trip += (long long)gridDim.x * 4) {
    const int *ip = idx + trip * 64;
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int rec = ip[16 * r + (lane >> 2)];
      const char *g = table + (size_t)rec * 64 + (lane & 3) * 16;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                       (__attribute__((address_space(3))) void *)(tile + r * 1024), 16, 0, 0);
    }
    __builtin_amdgcn_s_waitcnt(0);
    acc += *(volatile int *)(tile + lane * 64);
  }
  if (acc == 0x12345) *out = acc;
}

int main(int argc, char **argv) {
  const long long nrec = 4ll << 20;   // 4 M records of 64 B = 256 MiB table: every line is a miss in L2, most in the Infinity Cache
  char *table; int *idx, *idxp, *out; int4 *stream;
  CK(hipMalloc(&table, nrec * 64)); CK(hipMemset(table, 1, nrec * 64));
  CK(hipMalloc(&stream, nrec * 64)); CK(hipMemset(stream, 1, nrec * 64));
  CK(hipMalloc(&out, 4));
  std::vector<int> perm(nrec); std::iota(perm.begin(), perm.end(), 0);
  std::mt19937_64 rng(1); std::shuffle(perm.begin(), perm.end(), rng);
  CK(hipMalloc(&idx, nrec * 4)); CK(hipMemcpy(idx, perm.data(), nrec * 4, hipMemcpyHostToDevice));
  // paired: random order of 128-byte lines, the two records of a line in neighbouring slots (neighbouring quads of one instruction)
  std::vector<int> lines(nrec / 2); std::iota(lines.begin(), lines.end(), 0); std::shuffle(lines.begin(), lines.end(), rng);
  std::vector<int> pp(nrec);
  for (long long k = 0; k < nrec / 2; k++) { pp[2 * k] = 2 * lines[k]; pp[2 * k + 1] = 2 * lines[k] + 1; }
  CK(hipMalloc(&idxp, nrec * 4)); CK(hipMemcpy(idxp, pp.data(), nrec * 4, hipMemcpyHostToDevice));
  const long long ntrips = nrec / 64;
  for (int rep = 0; rep < 2; rep++) {
    k_stream<<<4096, 256>>>(stream, nrec * 4, out);                       // nrec * 64 bytes
    k_gather64<<<4096, 256, 4 * 4096>>>(table, idx, ntrips, out);         // nrec * 64 bytes (+ nrec * 4 of indices)
    k_gather64<<<4096, 256, 4 * 4096>>>(table, idxp, ntrips, out);        // the same bytes, line halves side by side
  }
  CK(hipDeviceSynchronize());
  printf("known bytes per kernel: stream %lld, gather %lld (+ %lld index bytes)\n", nrec * 64, nrec * 64, nrec * 4);
  return 0;
}
