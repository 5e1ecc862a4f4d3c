// polar_exact.hpp -- exact (all-pairs) mode of the dipole solver (a6/a7): the exact-order Gauss-Seidel.  Matrix-free form (one wave
// walks the recurrence of 64 atoms: k_gs_block_seq / k_gs_block_push) and the form on the HBM-resident packed tensor, where a
// block's recurrence is replaced by triangular inverses formed once per step (k_build_T6, k_gs_blockinv, k_gs_expand,
// k_gs_gemm, k_gs_blk).  Part of the hand-written HIP kernels (gfx950 / CDNA4, wave64) of the
// lj/cut/coul/long/polarization hot path; see polar_kernels.hpp for the mapping and the index spaces.
#pragma once

#include "polar_common.hpp"
#include "polar_solver.hpp"   // tensor_scalars, Scal, slots

namespace polar {

// ------------------------------------------------------------------------------------------
// a7  sequential (ranked) Gauss-Seidel, exact-order, blocked for the GPU.
// F_j = -sum_k T_jk mu_k is kept current for every atom.  For a block of 64 consecutive atoms of
// the ranked order:
//   k_gs_block_seq  (ONE wave): in order k = 0..63   mu_k <- alpha_k (E_k + F_k), then every
//                   other lane of the block applies  F_l -= T_lk dmu_k  (wave broadcast) --
//                   exactly the reference's "use the newest mu" recurrence (PS.cpp:1158-1180)
//   k_gs_block_push (all rows outside the block): F_j -= sum_k T_jk dmu_k
// so atoms later in the order see the new dipoles, atoms earlier keep a current field for the
// next sweep.  Arithmetic differs from the reference only in summation order.
template <int DAMP>
static __global__ __launch_bounds__(64) void k_gs_block_seq(int nlocal, int b0, const int *__restrict__ order,
                                                     AtomRec *__restrict__ rec, Box box, double pd,
                                                     const double *__restrict__ ef, double *__restrict__ F,
                                                     double *__restrict__ dmu_blk, const Scal *scal,
                                                     double *__restrict__ slots) {
  if (scal->done) return;
  const int lane = threadIdx.x;
  const int cnt = min(64, nlocal - b0);
  const bool act = lane < cnt;
  const int i = act ? order[b0 + lane] : 0;
  AtomRec r = rec[i];
  double Fx = act ? F[3 * i] : 0, Fy = act ? F[3 * i + 1] : 0, Fz = act ? F[3 * i + 2] : 0;
  const double Ex = act ? ef[3 * i] : 0, Ey = act ? ef[3 * i + 1] : 0, Ez = act ? ef[3 * i + 2] : 0;
  const double mx0 = r.mx, my0 = r.my, mz0 = r.mz;
  double dsq = 0.0;
  for (int k = 0; k < cnt; k++) {
    const double nx = r.a * (Ex + Fx), ny = r.a * (Ey + Fy), nz = r.a * (Ez + Fz);
    const double ddx = nx - r.mx, ddy = ny - r.my, ddz = nz - r.mz;
    const double bx = __shfl(r.x, k, 64), by = __shfl(r.y, k, 64), bz = __shfl(r.z, k, 64);
    const double bdx = __shfl(ddx, k, 64), bdy = __shfl(ddy, k, 64), bdz = __shfl(ddz, k, 64);
    if (lane == k) {
      r.mx = nx; r.my = ny; r.mz = nz;
    } else if (act && (bdx != 0.0 || bdy != 0.0 || bdz != 0.0)) {
      double dx, dy, dz;
      min_image_del(box, r.x, r.y, r.z, bx, by, bz, dx, dy, dz);
      const double r2 = dx * dx + dy * dy + dz * dz;
      double s3, s5;
      tensor_scalars<DAMP>(r2, pd, s3, s5);
      const double md = bdx * dx + bdy * dy + bdz * dz;
      const double c = s5 * md;
      Fx -= s3 * bdx - c * dx; Fy -= s3 * bdy - c * dy; Fz -= s3 * bdz - c * dz;
    }
  }
  if (act) {
    const double tx = r.mx - mx0, ty = r.my - my0, tz = r.mz - mz0;
    dsq = tx * tx + ty * ty + tz * tz;
    rec[i].mx = r.mx; rec[i].my = r.my; rec[i].mz = r.mz;
    F[3 * i] = Fx; F[3 * i + 1] = Fy; F[3 * i + 2] = Fz;
    dmu_blk[3 * lane] = tx; dmu_blk[3 * lane + 1] = ty; dmu_blk[3 * lane + 2] = tz;
  }
  dsq = wave_sum(dsq);
  if (lane == 0 && dsq != 0.0) atomicAdd(slots + (size_t)((b0 >> 6) & (POLAR_NSLOT - 1)) * POLAR_SLOT_STRIDE + SL_CHANGE, dsq);
}

template <int DAMP>
static __global__ __launch_bounds__(POLAR_BLOCK) void k_gs_block_push(int nlocal, int b0, const int *__restrict__ order,
                                                               const int *__restrict__ pos_in_order,
                                                               const AtomRec *__restrict__ rec, Box box, double pd,
                                                               const double *__restrict__ dmu_blk,
                                                               double *__restrict__ F, const Scal *scal) {
  if (scal->done) return;
  const int lane = threadIdx.x & 63;
  const int j = blockIdx.x * POLAR_ROWS_PER_BLOCK + (threadIdx.x >> 6);
  if (j >= nlocal) return;
  const int pj = pos_in_order[j];
  if (pj >= b0 && pj < b0 + 64) return;  // rows of the block were updated by k_gs_block_seq
  const AtomRec rj = rec[j];
  if (rj.a == 0.0) return;  // mu_j stays 0: its field is never read
  const int cnt = min(64, nlocal - b0);
  double fx = 0, fy = 0, fz = 0;
  if (lane < cnt) {
    const double bdx = dmu_blk[3 * lane], bdy = dmu_blk[3 * lane + 1], bdz = dmu_blk[3 * lane + 2];
    if (bdx != 0.0 || bdy != 0.0 || bdz != 0.0) {
      const AtomRec rk = rec[order[b0 + lane]];
      double dx, dy, dz;
      min_image_del(box, rj.x, rj.y, rj.z, rk.x, rk.y, rk.z, dx, dy, dz);
      const double r2 = dx * dx + dy * dy + dz * dz;
      double s3, s5;
      tensor_scalars<DAMP>(r2, pd, s3, s5);
      const double md = bdx * dx + bdy * dy + bdz * dz;
      const double c = s5 * md;
      fx = -(s3 * bdx - c * dx); fy = -(s3 * bdy - c * dy); fz = -(s3 * bdz - c * dz);
    }
  }
  fx = wave_sum(fx); fy = wave_sum(fy); fz = wave_sum(fz);
  if (lane == 0) { F[3 * j] += fx; F[3 * j + 1] += fy; F[3 * j + 2] += fz; }
}

// ------------------------------------------------------------------------------------------
// Exact mode with the tensor held in HBM, as the reference does (build_dipole_field_matrix,
// PS.cpp:1243-1316) but packed: T6[i][j] = {Txx,Txy,Txz,Tyy,Tyz,Tzz}, 48 N^2 bytes (the reference's
// dense matrix is 72 N^2).  Used by the exact-order Gauss-Seidel when it fits: the sequential chain
// then has no exp / rsqrt / minimum image in it, only 9 FMAs per step.  Atoms are in RANKED order
// here (s space = sweep order), so a block of 64 consecutive steps reads contiguous tensor rows.
// Layout (round 4): component-major rows, T6[(i * 6 + c) * np + j], np = n rounded up to 64 -- the 64 lanes of a wave that
// walk 64 COLUMNS of a row read 512 contiguous bytes per component.  (Round 1-3 stored the six components of a pair together:
// a wave's load then touched 64 x 48 = 3 KB, and the one wave of the recurrence spent 8 of its 13 us per block waiting for its
// 384 such loads: config 0, 283 us per iteration.)
template <int DAMP>
static __global__ __launch_bounds__(POLAR_BLOCK) void k_build_T6(int n, long long np, const AtomRec *__restrict__ rec, Box box, double pd,
                                                          double *__restrict__ T6) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * POLAR_ROWS_PER_BLOCK + (threadIdx.x >> 6);
  if (i >= n) return;
  const AtomRec ri = rec[i];
  for (int j = lane; j < n; j += 64) {
    double t[6] = {0, 0, 0, 0, 0, 0};
    if (j != i) {
      const AtomRec rj = rec[j];
      double dx, dy, dz;
      min_image_del(box, ri.x, ri.y, ri.z, rj.x, rj.y, rj.z, dx, dy, dz);
      double s3, s5;
      tensor_scalars<DAMP>(dx * dx + dy * dy + dz * dz, pd, s3, s5);
      t[0] = s3 - s5 * dx * dx; t[1] = -s5 * dx * dy; t[2] = -s5 * dx * dz;
      t[3] = s3 - s5 * dy * dy; t[4] = -s5 * dy * dz; t[5] = s3 - s5 * dz * dz;
    }
#pragma unroll
    for (int c = 0; c < 6; c++) T6[((size_t)i * 6 + c) * np + j] = t[c];
  }
}

// F_i = - sum_j T_ij mu_j  (dense mat-vec; initial running field of the Gauss-Seidel)
static __global__ __launch_bounds__(POLAR_BLOCK) void k_dense_field(int n, long long np, const double *__restrict__ T6,
                                                             const AtomRec *__restrict__ rec, double *__restrict__ F) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * POLAR_ROWS_PER_BLOCK + (threadIdx.x >> 6);
  if (i >= n) return;
  double fx = 0, fy = 0, fz = 0;
  for (int j = lane; j < n; j += 64) {
    double t[6];
#pragma unroll
    for (int c = 0; c < 6; c++) t[c] = T6[((size_t)i * 6 + c) * np + j];
    const double mx = rec[j].mx, my = rec[j].my, mz = rec[j].mz;
    fx -= t[0] * mx + t[1] * my + t[2] * mz;
    fy -= t[1] * mx + t[3] * my + t[4] * mz;
    fz -= t[2] * mx + t[4] * my + t[5] * mz;
  }
  fx = wave_sum(fx); fy = wave_sum(fy); fz = wave_sum(fz);
  if (lane == 0) { F[3 * i] = fx; F[3 * i + 1] = fy; F[3 * i + 2] = fz; }
}

// ------------------------------------------------------------------------------------------
// The exact-order sweep WITHOUT a sequential recurrence (round 4).  Take B consecutive atoms of the sweep order (a "block",
// R = 3 B scalars).  Inside it the reference's loop (PS.cpp:1158-1180) is
//     d_k = a_k (E_k + F_k - sum_{j<k} T_kj d_j) - mu_k,
// the unit lower triangular system (I + A L) d = c with c_k = a_k (E_k + F_k) - mu_k, F the field of the dipoles as they stand
// when the block starts, L the block's strictly lower tensor blocks, A = diag(a).  The tensor does not change during a solve:
//     G = (I + A L)^-1                    once per step, per block;
//     N = G A T[block][block before it]   once per step, per block (the block before it in the sweep, cyclically);
// and with cb = a (E + F) - mu taken BEFORE the change d' of the block before it has reached F,
//     d = G cb - N d'
// is two mat-vecs whose operands all exist when the launch starts: ONE launch per block, no dependent chain inside it, the
// rows of d spread over 3 B / 4 workgroups.  The same numbers as the recurrence up to rounding (other summation order).
// Rounds 1-4 walked the recurrence with one wave: 160 ns per step whatever was done to it (profiles/r04_config0_chain.txt).
#define POLAR_GS64 192   // scalars of 64 atoms
// G of 64 atoms: one wave per 16 columns, the columns in LDS.  Rows three at a time (atom k): row k of G = e_k - a_k sum_{j<k}
// T_kj (rows of j); the four quarter-waves take j = 0, 1, 2, 3 mod 4.  The tensor rows arrive four atoms at a time: the next
// four rows (6 x 4 coalesced requests) are fetched into the other LDS buffer while the current four are worked on -- one
// memory latency per four steps instead of one per step, and that one mostly hidden.  `ld`, `stride`, `per`: G of 64-atom
// piece p lands at G + (p / per) * stride + (p % per) * 192 * (ld + 1) -- the diagonal pieces of a larger block's G.
#define POLAR_GSB_COLS 16
#define POLAR_GSB_ROWS 4
static __global__ __launch_bounds__(64) void k_gs_blockinv(int n, long long np, const double *__restrict__ T6,
                                                    const AtomRec *__restrict__ rec, double *__restrict__ G, int ld, long long stride, int per) {
  __shared__ double col[POLAR_GS64][POLAR_GSB_COLS];
  __shared__ double trow[2][POLAR_GSB_ROWS][6][64];
  constexpr int PARTS = POLAR_GS64 / POLAR_GSB_COLS, RB = POLAR_GSB_ROWS;
  const int piece = blockIdx.x / PARTS, b0 = piece * 64, cnt = min(64, n - b0);
  const int lane = threadIdx.x, t = lane & (POLAR_GSB_COLS - 1), quarter = lane / POLAR_GSB_COLS, q = (blockIdx.x % PARTS) * POLAR_GSB_COLS + t;
  double *out = G + (size_t)(piece / per) * stride + (size_t)(piece % per) * POLAR_GS64 * ((size_t)ld + 1) + q;
  if (cnt <= 0) {   // a piece past the end of the system: identity
    for (int r = quarter; r < POLAR_GS64; r += 4) out[(size_t)r * ld] = r == q ? 1.0 : 0.0;
    return;
  }
  const int j0 = ((blockIdx.x % PARTS) * POLAR_GSB_COLS) / 3;   // rows above a column's own atom are zero: start at the workgroup's first
  const int jl = lane < cnt ? lane : 0;                          // (columns past the end are never used)
  double nx[RB][6];
#define POLAR_GSB_FETCH(K0)                                                                            \
  _Pragma("unroll") for (int u = 0; u < RB; u++) {                                                     \
    const int kr = (K0) + u < cnt ? (K0) + u : 0;                                                      \
    _Pragma("unroll") for (int c = 0; c < 6; c++) nx[u][c] = T6[((size_t)(b0 + kr) * 6 + c) * np + b0 + jl]; \
  }
#define POLAR_GSB_STAGE(BUF)                                                                           \
  _Pragma("unroll") for (int u = 0; u < RB; u++)                                                       \
    _Pragma("unroll") for (int c = 0; c < 6; c++) trow[BUF][u][c][lane] = nx[u][c];
  POLAR_GSB_FETCH(0);
  POLAR_GSB_STAGE(0);
  for (int k0 = 0; k0 < 64; k0 += RB) {
    const int buf = (k0 / RB) & 1;
    if (k0 + RB < 64) { POLAR_GSB_FETCH(k0 + RB); }   // in flight while the four steps below run
    for (int u = 0; u < RB; u++) {
      const int k = k0 + u;
      __syncthreads();   // (rows of col written by the step before; at u = 0 also the staged tensor rows)
      double m0 = q == 3 * k ? 1.0 : 0.0, m1 = q == 3 * k + 1 ? 1.0 : 0.0, m2 = q == 3 * k + 2 ? 1.0 : 0.0;
      const double a = k < cnt ? rec[b0 + k].a : 0.0;                       // (uniform)
      if (a != 0.0 && j0 < k) {
        const double(*tr)[64] = trow[buf][u];
        double s0 = 0, s1 = 0, s2 = 0;
#pragma unroll 4
        for (int j = j0 + quarter; j < k; j += 4) {
          const double t0 = tr[0][j], t1 = tr[1][j], t2 = tr[2][j], t3 = tr[3][j], t4 = tr[4][j], t5 = tr[5][j];
          const double p0 = col[3 * j][t], p1 = col[3 * j + 1][t], p2 = col[3 * j + 2][t];
          s0 += t0 * p0 + t1 * p1 + t2 * p2;
          s1 += t1 * p0 + t3 * p1 + t4 * p2;
          s2 += t2 * p0 + t4 * p1 + t5 * p2;
        }
        s0 += __shfl_xor(s0, 16, 64); s1 += __shfl_xor(s1, 16, 64); s2 += __shfl_xor(s2, 16, 64);
        s0 += __shfl_xor(s0, 32, 64); s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64);
        m0 -= a * s0; m1 -= a * s1; m2 -= a * s2;
      }
      if (quarter == 0) { col[3 * k][t] = m0; col[3 * k + 1][t] = m1; col[3 * k + 2][t] = m2; }
    }
    if (k0 + RB < 64) { POLAR_GSB_STAGE(buf ^ 1); }   // (the other buffer: its last readers passed the barriers of this group's steps)
  }
#undef POLAR_GSB_FETCH
#undef POLAR_GSB_STAGE
  __syncthreads();
  for (int r = quarter; r < POLAR_GS64; r += 4) out[(size_t)r * ld] = col[r][t];
}

// A T[rows][cols] written out densely (row-major, leading dimension ld, batch member z = blockIdx.z at out + z * stride):
// entry (3 k + x, 3 j + y) = a_k T_kj^xy for `m` atoms from r0(z) and `m` atoms from c0(z); zero past the end of the system.
// half = 0: rows = block z (m = B atoms), columns = the block before it in the sweep (cyclically): the operand of N.
// half > 0: z = (block, pair): rows = the second m atoms of the pair's 2 m atoms, columns = its first m: the off-diagonal piece
// that joins two triangles G1, G2 of m atoms into one of 2 m.
static __global__ __launch_bounds__(256) void k_gs_expand(int n, long long np, const double *__restrict__ T6, const AtomRec *__restrict__ rec,
                                                   int B, int nblocks, int half, int m, double *__restrict__ out, int ld, long long stride) {
  const int z = blockIdx.z;
  int r0, c0;
  if (half == 0) { r0 = z * B; c0 = ((z + nblocks - 1) % nblocks) * B; }
  else { const int per = B / (2 * m); c0 = (z / per) * B + (z % per) * 2 * m; r0 = c0 + m; }
  const int j = blockIdx.x * 64 + (threadIdx.x & 63), k = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (j >= m || k >= m) return;
  double t[6] = {0, 0, 0, 0, 0, 0};
  if (r0 + k < n && c0 + j < n) {
    const double a = rec[r0 + k].a;
#pragma unroll
    for (int c = 0; c < 6; c++) t[c] = a * T6[((size_t)(r0 + k) * 6 + c) * np + c0 + j];
  }
  double *o = out + (size_t)z * stride + (size_t)(3 * k) * ld + 3 * j;
  o[0] = t[0]; o[1] = t[1]; o[2] = t[2];
  o[ld] = t[1]; o[ld + 1] = t[3]; o[ld + 2] = t[4];
  o[2 * (size_t)ld] = t[2]; o[2 * (size_t)ld + 1] = t[4]; o[2 * (size_t)ld + 2] = t[5];
}

// C = alpha A B, row-major, batched over blockIdx.z (GsBatch: where a member sits); m, n, k multiples of 64.  LOWER: A is
// lower triangular (the k range of a row tile ends with the tile).  The one GEMM-shaped piece of the path, on the matrix cores:
// v_mfma_f64_16x16x4_f64, 64 x 64 of C per workgroup, 32 x 32 per wave = 2 x 2 MFMA tiles, operands staged through LDS k-major
// (lane l feeds A[row l & 15][k = l >> 4] and B[k = l >> 4][col l & 15]; a result register holds C[row (l >> 4) + 4 reg][col l & 15]).
// It runs once per step on a few 192..768-square matrices (G's joins and N = G A T).
struct GsBatch { long long outer, inner; int per; };   // batch member z sits at (z / per) * outer + (z % per) * inner
typedef double gs_d4 __attribute__((ext_vector_type(4)));
template <bool LOWER>
static __global__ __launch_bounds__(256) void k_gs_gemm(int k, double alpha, const double *__restrict__ A, long long lda, GsBatch bA,
                                                 const double *__restrict__ Bm, long long ldb, GsBatch bB,
                                                 double *__restrict__ C, long long ldc, GsBatch bC) {
  constexpr int KC = 32;   // k per LDS stage
  __shared__ double As[KC][65];
  __shared__ double Bs[KC][64];
  const int z = blockIdx.z;
  A += (size_t)(z / bA.per) * bA.outer + (size_t)(z % bA.per) * bA.inner;
  Bm += (size_t)(z / bB.per) * bB.outer + (size_t)(z % bB.per) * bB.inner;
  C += (size_t)(z / bC.per) * bC.outer + (size_t)(z % bC.per) * bC.inner;
  const int row0 = blockIdx.y * 64, col0 = blockIdx.x * 64, tid = threadIdx.x;
  const int lane = tid & 63, w = tid >> 6, wr = (w >> 1) * 32, wc = (w & 1) * 32, l15 = lane & 15, l4 = lane >> 4;
  gs_d4 acc[2][2] = {};
  const int kend = LOWER ? min(k, row0 + 64) : k;   // (a multiple of 64)
  for (int k0 = 0; k0 < kend; k0 += KC) {
#pragma unroll
    for (int e = tid; e < 64 * KC; e += 256) {
      As[e % KC][e / KC] = A[(size_t)(row0 + e / KC) * lda + k0 + e % KC];
      Bs[e >> 6][e & 63] = Bm[(size_t)(k0 + (e >> 6)) * ldb + col0 + (e & 63)];
    }
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < KC / 4; ks++) {
      const int kk = 4 * ks + l4;
      const double a0 = As[kk][wr + l15], a1 = As[kk][wr + 16 + l15];
      const double b0 = Bs[kk][wc + l15], b1 = Bs[kk][wc + 16 + l15];
      acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
    }
    __syncthreads();
  }
#pragma unroll
  for (int rb = 0; rb < 2; rb++)
#pragma unroll
    for (int cb = 0; cb < 2; cb++)
#pragma unroll
      for (int r = 0; r < 4; r++)
        C[(size_t)(row0 + wr + 16 * rb + l4 + 4 * r) * ldc + col0 + wc + 16 * cb + l15] = alpha * acc[rb][cb][r];
}

// cb = a (E + F) - mu for every atom, from the field the solve starts with
static __global__ __launch_bounds__(256) void k_gs_cb_init(int n, const AtomRec *__restrict__ rec, const double *__restrict__ ef,
                                                    const double *__restrict__ F, double *__restrict__ cb) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const AtomRec r = rec[i];
  cb[3 * i] = r.a * (ef[3 * i] + F[3 * i]) - r.mx;
  cb[3 * i + 1] = r.a * (ef[3 * i + 1] + F[3 * i + 1]) - r.my;
  cb[3 * i + 2] = r.a * (ef[3 * i + 2] + F[3 * i + 2]) - r.mz;
}

// ONE launch per block of B atoms starting at b0 (block index sb); p0 = start of the block before it in the sweep (< 0: the
// first launch of a solve), next0 = start of the block after it.  Workgroups of POLAR_GS_WAVES waves (W).
//   workgroups 0 .. 3B/W - 1: one wave per row r of the block: d_r = G[r][0..r] . cb - N[r][:] . d', the new dipole component;
//   the others: one wave per atom j of the system: F_j -= T[j][block p0] d' (every row, the rows of block b0 included -- cb
//   holds what the first group needs of them), and for the atoms of the NEXT block cb_j = a_j (E_j + F_j) - mu_j.
// Nothing in a launch depends on anything else in it.  (Needs at least two blocks: with one, cb of the block would be written
// and read in the same launch.)
#define POLAR_GS_WAVES 4   // waves per workgroup of k_gs_blk
// the end-of-sweep logic on a sweep's last launch: cnt = a zeroed counter (nullptr: no tail), npart = entries of dsq_part
struct GsTail { int *cnt; int npart, nlocal, fixed_iteration, iterations_max; double precision; };
template <int B>
static __global__ __launch_bounds__(64 * POLAR_GS_WAVES) void k_gs_blk(int n, long long np, int b0, int p0, int next0, const double *__restrict__ T6,
                                                 const double *__restrict__ G, const double *__restrict__ Nm, AtomRec *rec,
                                                 const double *__restrict__ ef, double *F, double *cb, const double *__restrict__ dmu_prev,
                                                 double *__restrict__ dmu_out, const Scal *scal, double *dsq_part, GsTail tail) {
  if (scal->done) return;
  constexpr int R = 3 * B, WV = POLAR_GS_WAVES, DW = R / WV;
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (blockIdx.x < DW) {
    __shared__ double dsh[WV];
    const int r = blockIdx.x * WV + w, k = r / 3, x = r - 3 * k, i = b0 + k;
    double acc = 0.0;
    if (i < n) {
      const size_t base = (size_t)(b0 / B) * R * R + (size_t)r * R;
      const double *g = G + base, *nn = Nm + base;
      const double *c = cb + 3 * (size_t)b0;
#pragma unroll
      for (int u = 0; u < R / 64; u++) {
        const int q = u * 64 + lane;
        if (q < 3 * k + 3) acc = fma(g[q], c[q], acc);       // (G is lower triangular with 3 x 3 identities on its diagonal)
        if (p0 >= 0) acc = fma(-nn[q], dmu_prev[q], acc);
      }
    }
    acc = wave_sum(acc);
    if (lane == 0) {
      dsh[w] = acc;
      dmu_out[r] = acc;
      if (i < n) { double *m = x == 0 ? &rec[i].mx : x == 1 ? &rec[i].my : &rec[i].mz; *m += acc; }
    }
    __syncthreads();
    __shared__ int lastwg;
    if (threadIdx.x == 0) {
      double dsq = 0.0;
#pragma unroll
      for (int v = 0; v < WV; v++) dsq += dsh[v] * dsh[v];
      // one entry per (block, workgroup), rewritten every sweep and added up in a fixed order, so the sweep's sum |dmu|^2 -- and
      // with it the iteration count at a knife edge -- is the same run to run
      double *mine = dsq_part + (size_t)(b0 / B) * DW + blockIdx.x;
      if (!tail.cnt) *mine = dsq;
      else {
        // the sweep's LAST launch: the workgroups count themselves out and the last one applies the end-of-sweep logic.  The entry
        // goes out as an atomic (performed at the coherence point: a plain store could still sit in this XCD's L2 when a workgroup
        // on another XCD adds the entries up), and the count is an agent-scope RELEASE: the HSA memory model orders the entry
        // before the count for workgroups on other XCDs only then (a workgroup-scope fence happened to be enough with
        // this compiler).  One thread of 3B/4 workgroups on one launch per sweep pays for it.
        atomicExch((unsigned long long *)mine, (unsigned long long)__double_as_longlong(dsq));
        const int last = __hip_atomic_fetch_add(tail.cnt, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT) == DW - 1;
        if (last) atomicExch(tail.cnt, 0);
        lastwg = last;
      }
    }
    if (!tail.cnt) return;
    __syncthreads();
    if (!lastwg) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // pairs with the counting workgroups' release
    double v = 0.0;   // thread t adds entries t, t + 256, ...; the wave sums, then the waves in order: a fixed association
    for (int k2 = threadIdx.x; k2 < tail.npart; k2 += 64 * WV) v += __hip_atomic_load(dsq_part + k2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    v = wave_sum(v);
    __syncthreads();
    if (lane == 0) dsh[w] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
      double sum = 0.0;
#pragma unroll
      for (int q2 = 0; q2 < WV; q2++) sum += dsh[q2];
      solver_decide(const_cast<Scal *>(scal), sum, nullptr, tail.nlocal, tail.fixed_iteration, tail.iterations_max, tail.precision, 0, 1);
    }
    return;
  }
  const int j = (blockIdx.x - DW) * WV + w;
  if (p0 < 0 || j >= n) return;
  const AtomRec rj = rec[j];
  const bool nextrow = j >= next0 && j < next0 + B;
  if (rj.a == 0.0) {  // its field is never read; c = -mu (whatever dipole it was handed goes away in its first step)
    if (nextrow && lane == 0) { cb[3 * j] = -rj.mx; cb[3 * j + 1] = -rj.my; cb[3 * j + 2] = -rj.mz; }
    return;
  }
  const int pcnt = min(B, n - p0);
  double fx = 0, fy = 0, fz = 0;
#pragma unroll
  for (int u = 0; u < B / 64; u++) {
    const int cidx = u * 64 + lane;
    if (cidx < pcnt) {
      double t[6];
#pragma unroll
      for (int c = 0; c < 6; c++) t[c] = T6[((size_t)j * 6 + c) * np + p0 + cidx];
      const double bdx = dmu_prev[3 * cidx], bdy = dmu_prev[3 * cidx + 1], bdz = dmu_prev[3 * cidx + 2];
      fx -= t[0] * bdx + t[1] * bdy + t[2] * bdz;
      fy -= t[1] * bdx + t[3] * bdy + t[4] * bdz;
      fz -= t[2] * bdx + t[4] * bdy + t[5] * bdz;
    }
  }
  fx = wave_sum(fx); fy = wave_sum(fy); fz = wave_sum(fz);
  if (lane == 0) {
    const double Fx = F[3 * j] + fx, Fy = F[3 * j + 1] + fy, Fz = F[3 * j + 2] + fz;
    F[3 * j] = Fx; F[3 * j + 1] = Fy; F[3 * j + 2] = Fz;
    if (nextrow) {
      cb[3 * j] = rj.a * (ef[3 * j] + Fx) - rj.mx;
      cb[3 * j + 1] = rj.a * (ef[3 * j + 1] + Fy) - rj.my;
      cb[3 * j + 2] = rj.a * (ef[3 * j + 2] + Fz) - rj.mz;
    }
  }
}

}  // namespace polar
