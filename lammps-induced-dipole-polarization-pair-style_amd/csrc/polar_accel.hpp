// polar_accel.hpp -- `polar_accel m` (extension keyword): Anderson mixing on the sweep map.  Part of the hand-written HIP
// kernels (gfx950 / CDNA4, wave64) of the lj/cut/coul/long/polarization hot path; see polar_kernels.hpp.
#pragma once

#include "polar_common.hpp"
#include "polar_solver.hpp"

namespace polar {

// ------------------------------------------------------------------------------------------
// `polar_accel m` (extension keyword): Anderson mixing of depth m on the sweep map.  G(x) = one colour-phase Gauss-Seidel
// sweep over the dipoles x of the own rows (PS.cpp:1158-1180), r_k = G(x_k) - x_k -- the quantity the stop rule measures
// (PS.cpp:1194-1210), so the rule itself is untouched.  After sweep k (records hold g_k = G(x_k)):
//     dF_k = r_k - r_{k-1},  dG_k = g_k - g_{k-1}   (the last m of them kept, ring of slots)
//     gamma = argmin | r_k - sum_j gamma_j dF_j |   (normal equations, m <= 8, solved by one thread in FP64)
//     x_{k+1} = g_k - sum_j gamma_j dG_j            (written into the records: the next sweep starts from it)
// Vectors are stored by component, [3][pitch] with pitch = rows rounded up to 256, rows in launch order (desc[r].x = atom).
// k_accel_diff writes per-workgroup partial dot products (no atomics: the same sums run to run); k_accel_solve folds them,
// extends the Gram matrix by the new row, solves and leaves gamma; k_accel_mix applies it.  All three return at once when
// the solver has stopped (scal->done): the dipoles returned are G(x_k) of the last sweep, as without the keyword.
#define POLAR_ACCEL_MAXM 8
struct AccelState {
  double gram[POLAR_ACCEL_MAXM * POLAR_ACCEL_MAXM];  // dF_i . dF_j by slot
  double gamma[POLAR_ACCEL_MAXM];
  int count, head, sweeps, pad;                       // differences stored, slot of the newest, sweeps seen
};
static __global__ void k_accel_init(int nrows, long long pitch, const int2 *__restrict__ desc, const AtomRec *__restrict__ rec,
                                    double *__restrict__ x, AccelState *st) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r == 0) {
    st->count = 0; st->head = -1; st->sweeps = 0;
    for (int k = 0; k < POLAR_ACCEL_MAXM * POLAR_ACCEL_MAXM; k++) st->gram[k] = 0.0;
    for (int k = 0; k < POLAR_ACCEL_MAXM; k++) st->gamma[k] = 0.0;
  }
  if (r >= nrows) return;
  const AtomRec &a = rec[desc[r].x];
  x[r] = a.mx; x[pitch + r] = a.my; x[2 * pitch + r] = a.mz;
}
// partial[(2 * M) * block + q]: q < M: dF_q . dF_head, q >= M: dF_{q-M} . r_k (all slots; unused ones are zero vectors)
template <int M>
static __global__ __launch_bounds__(256) void k_accel_diff(int nrows, long long pitch, const int2 *__restrict__ desc, const AtomRec *__restrict__ rec,
                                                    const Scal *scal, const AccelState *st, const double *__restrict__ x, double *__restrict__ fprev,
                                                    double *__restrict__ gprev, double *__restrict__ dF, double *__restrict__ dG,
                                                    double *__restrict__ partial, int ring) {
  if (scal->done) return;
  __shared__ double red[4][2 * M];
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  const bool have_prev = st->sweeps > 0;
  const int head = (st->head + 1) % ring;   // the slot this sweep's differences go to (ring of `ring` <= M slots = the depth)
  double acc[2 * M];
#pragma unroll
  for (int q = 0; q < 2 * M; q++) acc[q] = 0.0;
  if (r < nrows) {
    const AtomRec &a = rec[desc[r].x];
    const double g[3] = {a.mx, a.my, a.mz};
#pragma unroll
    for (int c = 0; c < 3; c++) {
      const long long e = c * pitch + r;
      const double f = g[c] - x[e];
      double dfh = 0.0;
      if (have_prev) {
        dfh = f - fprev[e];
        dF[(long long)head * 3 * pitch + e] = dfh;
        dG[(long long)head * 3 * pitch + e] = g[c] - gprev[e];
      }
      fprev[e] = f; gprev[e] = g[c];
      if (have_prev) {
#pragma unroll
        for (int q = 0; q < M; q++) {
          if (q >= ring) break;
          const double dq = q == head ? dfh : dF[(long long)q * 3 * pitch + e];
          acc[q] += dq * dfh;
          acc[M + q] += dq * f;
        }
      }
    }
  }
#pragma unroll
  for (int q = 0; q < 2 * M; q++) {
    const double v = wave_sum(acc[q]);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][q] = v;
  }
  __syncthreads();
  if (threadIdx.x < 2 * M) partial[(size_t)(2 * M) * blockIdx.x + threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}
// one workgroup: fold the partial sums (optionally the all-reduced ones arrive in `global`: [2 M] doubles), new Gram row, solve.
// The m x m system (Gram matrix of the differences + Tikhonov term: symmetric positive definite, no pivoting needed) is
// eliminated by ONE WAVE with row i in the registers of lane i -- statically indexed, exchanged by shuffles: a one-thread
// version with a local array cost 80 us per sweep (dynamic indexing = scratch memory).
// The work of k_accel_solve, for a workgroup of any size >= 256 threads: EVERY thread of the workgroup calls it (barriers inside),
// threads 0..255 do the work.  Not called once the solver has stopped.
template <int M>
__device__ __forceinline__ void accel_solve_body(int tid, int nblocks, const double *__restrict__ partial, AccelState *st,
                                                 double *__restrict__ local_out, const double *__restrict__ global, int ring) {
  __shared__ double stripe[16][2 * M];
  __shared__ double sums[2 * M];
  static_assert(2 * M == 16, "the fold below deals 256 threads as 16 stripes x 16 values");
  const int q = tid & 15, sp = (tid >> 4) & 15;
  if (!global && tid < 256) {
    double v = 0.0;
    for (int b = sp; b < nblocks; b += 16) v += partial[(size_t)(2 * M) * b + q];
    stripe[sp][q] = v;
  }
  __syncthreads();
  if (tid < 2 * M) {
    double v = 0.0;
    if (global) v = global[tid];
    else for (int k = 0; k < 16; k++) v += stripe[k][tid];
    sums[tid] = v;
    if (local_out && !global) local_out[tid] = v;
  }
  __syncthreads();
  if (local_out && !global) return;   // multi-GPU: this launch only exports the local sums; a second one, after the all-reduce, solves
  if (tid >= 64) return;
  const int lane = tid;
  const bool have_prev = st->sweeps > 0;
  __builtin_amdgcn_wave_barrier();
  if (!have_prev) { if (lane == 0) st->sweeps += 1; return; }   // first sweep: nothing to mix yet (x_1 = g_0)
  const int head = (st->head + 1) % ring;
  const int n = st->count < ring ? st->count + 1 : ring;
  // Gram matrix with the new row / column (kept in memory for the sweeps to come)
  if (lane < ring) { st->gram[head * M + lane] = sums[lane]; st->gram[lane * M + head] = sums[lane]; }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  // window: the n newest slots; lane i < n owns row i (slot of row i: head - i around the ring)
  const int my_slot = (head - lane + 2 * ring) % ring;
  double a[M], rhs = 0.0;
  double tr = 0.0;
  for (int k = 0; k < n; k++) { const int sk = (head - k + 2 * ring) % ring; tr += st->gram[sk * M + sk]; }
#pragma unroll
  for (int j = 0; j < M; j++) {
    const int sj = (head - j + 2 * ring) % ring;
    a[j] = (lane < n && j < n) ? st->gram[my_slot * M + sj] : (lane == j ? 1.0 : 0.0);
    if (lane == j && lane < n) a[j] += 1e-14 * tr + 1e-300;   // Tikhonov: the differences become collinear as the iteration converges
  }
  if (lane < n) rhs = sums[M + my_slot];
#pragma unroll
  for (int c = 0; c < M; c++) {       // elimination (rows >= n are identity rows: nothing happens to them)
    const double piv = __shfl(a[c], c, 64), rc = __shfl(rhs, c, 64);
    const double m = (lane > c && lane < M) ? a[c] / piv : 0.0;
#pragma unroll
    for (int j = 0; j < M; j++) { const double pj = __shfl(a[j], c, 64); if (j >= c) a[j] -= m * pj; }
    rhs -= m * rc;
  }
  double xs[M];
#pragma unroll
  for (int i = M - 1; i >= 0; i--) {  // back substitution: lane i solves for unknown i, everybody hears it
    double v = rhs;
#pragma unroll
    for (int j = i + 1; j < M; j++) v -= a[j] * xs[j];
    xs[i] = __shfl(v / a[i], i, 64);
  }
  bool ok = true;
#pragma unroll
  for (int i = 0; i < M; i++) ok = ok && (xs[i] - xs[i] == 0.0);
  if (lane < M) st->gamma[lane] = 0.0;
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int i = 0; i < M; i++) if (ok && i < n && lane == i) st->gamma[my_slot] = xs[i];
  if (lane == 0) {
    st->sweeps += 1;
    st->head = head;
    st->count = ok ? n : 0;           // not finite: restart the history; this step is a plain sweep
  }
}
template <int M>
static __global__ __launch_bounds__(256) void k_accel_solve(int nblocks, const double *__restrict__ partial, const Scal *scal, AccelState *st,
                                                     double *__restrict__ local_out, const double *__restrict__ global, int ring) {
  if (scal->done) return;
  accel_solve_body<M>(threadIdx.x, nblocks, partial, st, local_out, global, ring);
}
// Round 5: the single-workgroup launches around the mixing folded together (each dependent launch costs ~5 us; a rank's sweep at 8
// GPUs is ~110 us).
// k_accel_export: multi-GPU, before the all-reduce -- this rank's sum (dmu)^2 (what k_fold_change does; `fold`) into dst[0] and
// its 2 M dot products (k_accel_solve's export mode) into dst[1 ..] in ONE launch.
template <int M>
static __global__ __launch_bounds__(POLAR_NSLOT) void k_accel_export(Scal *scal, double *__restrict__ slots, double *__restrict__ dst, int fold,
                                                             int nblocks, const double *__restrict__ partial, AccelState *st, int ring) {
  __shared__ double red[POLAR_NSLOT / 64];
  if (fold) {
    double v = slots[(size_t)threadIdx.x * POLAR_SLOT_STRIDE + SL_CHANGE];
    slots[(size_t)threadIdx.x * POLAR_SLOT_STRIDE + SL_CHANGE] = 0.0;
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
      double sum = 0.0;
      for (int k = 0; k < POLAR_NSLOT / 64; k++) sum += red[k];
      scal->change = sum;
      dst[0] = sum;
    }
  }
  if (scal->done) return;   // (uniform: set by an earlier launch)
  accel_solve_body<M>(threadIdx.x, nblocks, partial, st, dst + 1, nullptr, ring);
}
// k_solver_accel: the end-of-sweep decision (k_solver_step; `decide`; multi-GPU: on the all-reduced change ared[0]) and, unless
// it ended the solve, the mixing coefficients (k_accel_solve; multi-GPU: from the all-reduced dot products ared[1 ..]) in ONE launch.
template <int M>
static __global__ __launch_bounds__(POLAR_NSLOT) void k_solver_accel(Scal *scal, double *__restrict__ slots, int nlocal, int iterations_max, double precision,
                                                             const double *__restrict__ ared, int decide, int nblocks,
                                                             const double *__restrict__ partial, AccelState *st, int ring) {
  if (scal->done) return;
  __shared__ double red[POLAR_NSLOT / 64];
  __shared__ int stop;
  if (threadIdx.x == 0) stop = 0;
  if (decide) {
    double v = slots[(size_t)threadIdx.x * POLAR_SLOT_STRIDE + SL_CHANGE];
    slots[(size_t)threadIdx.x * POLAR_SLOT_STRIDE + SL_CHANGE] = 0.0;
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
      double sum = 0.0;
      for (int k = 0; k < POLAR_NSLOT / 64; k++) sum += red[k];
      solver_decide(scal, sum, ared, nlocal, 0, iterations_max, precision, 0, 1);
      stop = scal->done;
    }
  }
  __syncthreads();
  if (stop) return;
  accel_solve_body<M>(threadIdx.x, nblocks, partial, st, nullptr, ared ? ared + 1 : nullptr, ring);
}
template <int M>
static __global__ __launch_bounds__(256) void k_accel_mix(int nrows, long long pitch, const int2 *__restrict__ desc, AtomRec *__restrict__ rec,
                                                   const Scal *scal, const AccelState *st, double *__restrict__ x,
                                                   const double *__restrict__ gprev, const double *__restrict__ dG) {
  if (scal->done) return;
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= nrows) return;
  double gam[M];
#pragma unroll
  for (int q = 0; q < M; q++) gam[q] = st->gamma[q];
  double xn[3];
#pragma unroll
  for (int c = 0; c < 3; c++) {
    const long long e = c * pitch + r;
    double v = gprev[e];
#pragma unroll
    for (int q = 0; q < M; q++) if (gam[q] != 0.0) v -= gam[q] * dG[(long long)q * 3 * pitch + e];
    xn[c] = v;
    x[e] = v;
  }
  AtomRec &a = rec[desc[r].x];
  a.mx = xn[0]; a.my = xn[1]; a.mz = xn[2];
}

}  // namespace polar
