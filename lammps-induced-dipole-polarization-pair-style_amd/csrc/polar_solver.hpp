// polar_solver.hpp -- the dipole solver (a6/a7): matrix-free field kernels, the list-mode sweep (k_field_lp), device-resident loop control.
// (Exact mode's exact-order Gauss-Seidel: polar_exact.hpp; `polar_accel`: polar_accel.hpp.)
// Part of the hand-written HIP kernels (gfx950 / CDNA4, wave64) of the lj/cut/coul/long/polarization
// hot path; see polar_kernels.hpp for the mapping and the index spaces.
#pragma once

#include "polar_common.hpp"

namespace polar {

// ------------------------------------------------------------------------------------------
// a6 + a7  the dipole-field sweep (matrix-free): for row i
//     ef_ind_i = - sum_j T_ij mu_j ,   mu_new_i = alpha_i (E_static_i + ef_ind_i)
// (PS.cpp:1158-1180 with the tensor of PS.cpp:1273-1306 recomputed per pair).
// Epilogues:
//   EP_JACOBI : read rec[cur], write rec[1-cur] (reference "polar_gs no")
//   EP_INPLACE: write mu into the same buffer (colour-phase Gauss-Seidel; rows of one colour do
//               not read each other's NEW values by construction of the phases)
//   EP_FIELD  : store ef_ind only (initial field of the blocked sequential Gauss-Seidel)
enum { EP_JACOBI = 0, EP_INPLACE = 1, EP_FIELD = 2 };

template <bool ALLPAIRS, int DAMP, int EP>
static __global__ __launch_bounds__(1024) void k_field(int nrows, const int *__restrict__ rows, int nlocal,
                                                       AtomRec *__restrict__ recA, AtomRec *__restrict__ recB, Box box,
                                                       RowList ddl,
                                                       const int *__restrict__ dd_j,
                                                       const double2 *__restrict__ dd_s, double ddcutsq, double pd,
                                                       const double *__restrict__ ef, double *__restrict__ Fout,
                                                       const Scal *scal, double *__restrict__ slots POLAR_LAB_PARAM) {
  if (scal->done) return;  // device-resident loop control: finished solves turn later launches into no-ops
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);  // blockDim.x/64 rows per workgroup
  if (row >= nrows) return;
  const int i = rows ? rows[row] : row;
  const int cur = scal->cur;
  const AtomRec *__restrict__ src = (EP == EP_JACOBI && cur) ? recB : recA;
  AtomRec *__restrict__ dst = (EP == EP_JACOBI) ? (cur ? recA : recB) : recA;
  const AtomRec ri = src[i];
  double fx = 0, fy = 0, fz = 0;
  if (ri.a != 0.0 || EP == EP_FIELD) {
    long long beg = 0, end = nlocal;
    if (!ALLPAIRS) row_range(ddl, i, beg, end);
    if (ALLPAIRS) {
      for (long long p = beg + lane; p < end; p += 64) {
        const int j = (int)p;
        if (j == i) continue;
        const AtomRec rj = src[j];
        double dx, dy, dz;
        min_image_del(box, ri.x, ri.y, ri.z, rj.x, rj.y, rj.z, dx, dy, dz);
        const double r2 = dx * dx + dy * dy + dz * dz;
        double s3, s5;
        tensor_scalars<DAMP>(r2, pd, s3, s5);
        const double md = rj.mx * dx + rj.my * dy + rj.mz * dz;
        const double c = s5 * md;
        fx -= s3 * rj.mx - c * dx;
        fy -= s3 * rj.my - c * dy;
        fz -= s3 * rj.mz - c * dz;
      }
    } else {
#ifdef POLAR_LAB
      // list mode (lab: the register-staged lane-per-pair sweep).  The damped tensor scalars (s3, s5) were cached per pair by k_dd_scalars, so a
      // sweep streams 20 B per pair (int32 j + two doubles) and gathers one 64-byte record.
      //   gather : scattered 16-byte loads cost one L1 (TCP) transaction per LANE, so the records of
      //            a trip's 64 pairs are fetched QUAD-cooperatively -- lane k of quad q loads piece k
      //            of record (r*16+q): 4 load instructions, each quad one coalesced 64-byte access;
      //   LDS    : the pieces are written to a per-wave staging tile (80-byte pitch: conflict-free
      //            b128 reads) and every lane reads back ITS pair's record: a wave-local transpose,
      //            no workgroup barrier (rows have different trip counts);
      //   math   : lane-per-pair, 64 pairs per VALU instruction.
      if (ablate & 1) end = beg;  // lab: no pair loop at all
      extern __shared__ double2 stage_all[];
      double2 *stage = stage_all + (size_t)(threadIdx.x >> 6) * (64 * 5);
      const int q4 = lane >> 2, k = lane & 3;
      // Three trips in flight (software pipeline): while trip t is transposed and computed, the
      // records of trip t+1 are being gathered and the index/scalar stream of trip t+2 is being read,
      // so a row pays its memory latencies once instead of twice per 64 pairs.
#define POLAR_LOAD_STREAM(BASE, JM, SC)                                   \
  {                                                                      \
    const long long p_ = (BASE) + lane;                                  \
    const bool ok_ = p_ < end;                                           \
    JM = (ok_ && !(ablate & 8)) ? dd_j[p_] : i;                          \
    SC = (ok_ && !(ablate & 4)) ? dd_s[p_] : make_double2(0.0, 0.0);     \
  }
#define POLAR_GATHER(JM, P0, P1, P2, P3)                                                   \
  {                                                                                        \
    int j0_, j1_, j2_, j3_;                                                                 \
    if (ablate & 128) { j0_ = JM; j1_ = JM ^ 1; j2_ = JM ^ 2; j3_ = JM ^ 3; } /* lab: no bpermute */ \
    else { j0_ = __shfl(JM, q4, 64); j1_ = __shfl(JM, 16 + q4, 64);                         \
           j2_ = __shfl(JM, 32 + q4, 64); j3_ = __shfl(JM, 48 + q4, 64); }                  \
    if (ablate & 2) j0_ = j1_ = j2_ = j3_ = i;                                             \
    P0 = reinterpret_cast<const double2 *>(src + j0_)[k];                                   \
    P1 = reinterpret_cast<const double2 *>(src + j1_)[k];                                   \
    P2 = reinterpret_cast<const double2 *>(src + j2_)[k];                                   \
    P3 = reinterpret_cast<const double2 *>(src + j3_)[k];                                   \
  }
      int jm0 = i, jm1 = i, jm2 = i;
      double2 sc0 = make_double2(0.0, 0.0), sc1 = sc0, sc2 = sc0;
      double2 pa0 = sc0, pa1 = sc0, pa2 = sc0, pa3 = sc0, pb0 = sc0, pb1 = sc0, pb2 = sc0, pb3 = sc0;
      if (beg < end) {
        POLAR_LOAD_STREAM(beg, jm0, sc0);
        POLAR_LOAD_STREAM(beg + 64, jm1, sc1);
        POLAR_GATHER(jm0, pa0, pa1, pa2, pa3);
      }
      for (long long base = beg; base < end; base += 64) {
        POLAR_LOAD_STREAM(base + 128, jm2, sc2);   // trip t+2 (predicated off past the row's end)
        POLAR_GATHER(jm1, pb0, pb1, pb2, pb3);     // trip t+1
        double2 a, b, c2;
        if (ablate & 64) {  // lab: no LDS transpose (wrong numbers, timing only)
          a = pa0; b = pa1; c2 = make_double2(pa2.x + pa3.x, pa2.y + pa3.y);
        } else {
        stage[(q4)*5 + k] = pa0; stage[(16 + q4) * 5 + k] = pa1;  // trip t
        stage[(32 + q4) * 5 + k] = pa2; stage[(48 + q4) * 5 + k] = pa3;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        a = stage[lane * 5]; b = stage[lane * 5 + 1]; c2 = stage[lane * 5 + 2];
        __builtin_amdgcn_wave_barrier();  // the tile is rewritten by the next trip
        }
        double dx, dy, dz;
        min_image_rint(box, ri.x, ri.y, ri.z, a.x, b.x, c2.x, dx, dy, dz);
        const double md = a.y * dx + b.y * dy + c2.y * dz;
        const double c = sc0.y * md;
        fx -= sc0.x * a.y - c * dx;
        fy -= sc0.x * b.y - c * dy;
        fz -= sc0.x * c2.y - c * dz;
        jm1 = jm2; sc0 = sc1; sc1 = sc2;
        pa0 = pb0; pa1 = pb1; pa2 = pb2; pa3 = pb3;
      }
#undef POLAR_LOAD_STREAM
#undef POLAR_GATHER
      fx = wave_sum(fx); fy = wave_sum(fy); fz = wave_sum(fz);
#endif
    }
    if (ALLPAIRS) { fx = wave_sum(fx); fy = wave_sum(fy); fz = wave_sum(fz); }
  }
  if (lane == 0) {
    if (EP == EP_FIELD) {
      Fout[3 * i] = fx; Fout[3 * i + 1] = fy; Fout[3 * i + 2] = fz;
    } else {
      const double a = ri.a;
      const double mx = a * (ef[3 * i] + fx), my = a * (ef[3 * i + 1] + fy), mz = a * (ef[3 * i + 2] + fz);
      const double ddx = mx - ri.mx, ddy = my - ri.my, ddz = mz - ri.mz;
      dst[i].mx = mx; dst[i].my = my; dst[i].mz = mz;
      const double c = ddx * ddx + ddy * ddy + ddz * ddz;
      if (c != 0.0 && !POLAR_ABL(16)) atomicAdd(slot_ptr(slots, SL_CHANGE), c);
    }
  }
}

#ifdef POLAR_LAB
// ------------------------------------------------------------------------------------------
// List-mode sweep, component-per-lane form (round 1's production kernel; lab build).
// The lane-per-pair kernel (k_field) gives every LANE one pair, so the 64-byte records fetched quad-wise have to
// be transposed through LDS and the indices shuffled to the quads: ~180 of the ~200 VALU slots of a
// 64-pair trip were bookkeeping, and the kernel was VALU-issue bound on it.  Here the quad that
// fetches a record also does its arithmetic: lane k of a quad owns COMPONENT k of the pair
//     d_k = x_ik - x_jk (wrapped),  dot = sum_k mu_jk d_k (quad DPP),  E_k -= s3 mu_jk - s5 dot d_k
// so nothing is transposed, no LDS is used, and the three field components are three lanes of one
// accumulator.  A gather instruction covers 16 pairs (one 64-byte access per quad for the record
// pieces {x_k, mu_k}).  Lane 3 of each quad rides along on component z (its results are unused).
// Rows are padded to whole 64-pair trips by k_dd_scalars (j = i, s = 0), so a trip needs no masks.
// SMODE 0: stream the cached (s3,s5) (20 B/pair); 1 / 2: stream the cached r^2 (12 B/pair) and
// rebuild (s3,s5) with exponential / no damping -- lane L for ITS pair, before the quad hand-round.
template <int EP, int SMODE>
static __global__ __launch_bounds__(1024) void k_field_quad(int nrows, const int *__restrict__ rows,
                                                            AtomRec *__restrict__ recA, AtomRec *__restrict__ recB,
                                                            Box box, RowList ddl, const int *__restrict__ dd_j,
                                                            const double2 *__restrict__ dd_s,
                                                            const double *__restrict__ dd_r2, double pd, ExpCoef K,
                                                            const double *__restrict__ ef, const Scal *scal,
                                                            double *__restrict__ slots, int ablate) {
  if (scal->done) return;
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int rpb = blockDim.x >> 6;  // rows (waves) per workgroup
  const int lb = xcd_block(blockIdx.x, (nrows + rpb - 1) / rpb);
  if (lb < 0) return;
  const int row = lb * rpb + wv;
  if (row >= nrows) return;
  const int i = __builtin_amdgcn_readfirstlane(rows ? rows[row] : row);
  const int cur = __builtin_amdgcn_readfirstlane(scal->cur);  // wave-uniform: the record base stays in SGPRs
  const AtomRec *__restrict__ src = (EP == EP_JACOBI && cur) ? recB : recA;
  AtomRec *__restrict__ dst = (EP == EP_JACOBI) ? (cur ? recA : recB) : recA;
  const int k = lane & 3, kk = k < 3 ? k : 2;
  const double *ri = reinterpret_cast<const double *>(src + i);
  const double xi = ri[2 * kk], mi = ri[2 * kk + 1], ai = ri[7];
  const double efk = ef[3 * i + kk];  // needed only in the epilogue: loaded here so that its latency is hidden
  long long c = ddl.cnt[i];
  if (c > ddl.pitch) c = ddl.pitch;
  if (ai == 0.0) c = 0;
  int T = __builtin_amdgcn_readfirstlane((int)((c + 63) >> 6));
  if (ablate & 1) T = 0;             // lab switches (POLAR_ABLATE): timing only, wrong numbers
  if ((ablate & 8) && T > 1) T = 1;
  const double prd = box.periodic[kk] ? box.prd[kk] : 0.0, inv = box.inv[kk];
  // gather address = scalar base + 32-bit lane offset (j * 64 + piece * 16; j < 2^26): one vector
  // instruction per gather instead of a 64-bit shift and a 64-bit add
  const char *srcc = reinterpret_cast<const char *>(src);
  const unsigned piece = (unsigned)kk * 16u;
  // stream: lane L reads pair L of the trip (ONE coalesced instruction each for j and (s3,s5): the
  // vector-memory address unit spends ~16 cycles per wave instruction however little it fetches, and
  // it is the unit this kernel saturates).  Pair 4q+r of a trip belongs to quad q, step r, so the
  // quad already holds its four pairs' stream values and hands them round with quad_perm DPP moves.
  const int *pj = dd_j + (size_t)i * ddl.pitch + lane;
  const double2 *ps = dd_s + (size_t)i * ddl.pitch + lane;
  const double *pr = dd_r2 + (size_t)i * ddl.pitch + lane;
  double acc = 0.0;
  // One trip per iteration; only the NEXT trip's indices are prefetched.  A deeper software pipeline
  // (gathers one trip ahead) was measured and bought nothing: with <= 64 VGPRs eight waves per SIMD
  // hide the latencies, and the kernel sits on the stream bandwidth and the VALU rate instead.
  int jn = pj[0];  // the pitch keeps this in bounds even for an empty row
  for (int t = 0; t < T; t++) {
    const int jv = (ablate & 2) ? i : jn;
    double2 P[4];
#define POLAR_QGATHER(R)                                                                        \
  {                                                                                            \
    const unsigned j_ = (unsigned)__builtin_amdgcn_update_dpp(0, jv, (R) * 0x55, 0xF, 0xF, true); \
    P[R] = *reinterpret_cast<const double2 *>(srcc + ((j_ << 6) + piece));                      \
  }
    POLAR_QGATHER(0) POLAR_QGATHER(1) POLAR_QGATHER(2) POLAR_QGATHER(3)
#undef POLAR_QGATHER
    double2 Sv = make_double2(0.0, 0.0);
    double r2v = 0.0;
    if (SMODE == 0) Sv = (ablate & 4) ? make_double2(1e-3, 1e-4) : ps[64 * t];
    else if (SMODE <= 2) r2v = (ablate & 4) ? 30.0 : pr[64 * t];
    if (t + 1 < T) jn = pj[64 * (t + 1)];  // wave-uniform: the next trip's indices travel during the math
    double D[4];
#define POLAR_QDEL(R)                                  \
  {                                                   \
    double d = xi - P[R].x;                            \
    D[R] = fma(-prd, rint(d * inv), d);                \
  }
    POLAR_QDEL(0) POLAR_QDEL(1) POLAR_QDEL(2) POLAR_QDEL(3)
#undef POLAR_QDEL
    if (SMODE >= 3) {
      // no per-pair stream value at all (4 B/pair; chosen when the stream would not stay in the 256 MB
      // Infinity Cache, see build_lists) -- r^2 from the quad's three
      // component lanes (lane 3 rides on z, so [1,2,0,0] / [2,0,1,1] give ALL four lanes the sum), and
      // lane r of the quad keeps the r^2 of step r: its own pair, as in the cached forms
      double r2s[4];
#pragma unroll
      for (int R = 0; R < 4; R++) {
        const double q = D[R] * D[R];
        r2s[R] = q + dpp_full<0x09>(q) + dpp_full<0x52>(q);  // quad_perm [1,2,0,0], [2,0,1,1]
      }
      r2v = k == 0 ? r2s[0] : (k == 1 ? r2s[1] : (k == 2 ? r2s[2] : r2s[3]));
      r2v = r2v > 0.0 ? r2v : 1e60;  // padding entries (the atom itself): an inert pair
    }
    if (SMODE != 0) tensor_scalars_k<(SMODE == 1 || SMODE == 3) ? 0 : 1>(r2v, pd, K, Sv.x, Sv.y);  // lane L: pair L of the trip
#define POLAR_QSTEP(R)                                                                          \
  {                                                                                            \
    const double s3_ = dpp_full<(R) * 0x55>(Sv.x), s5_ = dpp_full<(R) * 0x55>(Sv.y);             \
    const double d = D[R];                                                                      \
    const double m = P[R].y * d;                                                                \
    /* dot over the quad's three component lanes (lane 3 gets a don't-care) */                  \
    const double dot = m + dpp_full<0xC9>(m) + dpp_full<0xD2>(m); /* quad_perm [1,2,0,3], [2,0,1,3] */ \
    const double cc = s5_ * dot;                                                                \
    acc = fma(-s3_, P[R].y, acc);                                                               \
    acc = fma(cc, d, acc);                                                                      \
  }
    POLAR_QSTEP(0) POLAR_QSTEP(1) POLAR_QSTEP(2) POLAR_QSTEP(3)
#undef POLAR_QSTEP
  }
  // sum the 16 quads: rotate-adds inside the 16-lane rows, then across the four rows
  acc += dpp_full<0x124>(acc);  // row_ror:4
  acc += dpp_full<0x128>(acc);  // row_ror:8
  acc += __shfl_xor(acc, 16, 64);
  acc += __shfl_xor(acc, 32, 64);
  const double mu_new = ai * (efk + acc);
  const double dm = mu_new - mi;
  double chg = (k < 3) ? dm * dm : 0.0;
  chg = chg + dpp_full<0xC9>(chg) + dpp_full<0xD2>(chg);
  if (lane < 3) reinterpret_cast<double *>(dst + i)[2 * lane + 1] = mu_new;
  if (lane == 0 && chg != 0.0) atomicAdd(slot_ptr(slots, SL_CHANGE), chg);
}

#endif  // POLAR_LAB

// ------------------------------------------------------------------------------------------
// List-mode sweep, lane-per-pair form with the gather done by LDS-DMA (`global_load_lds_dwordx4`).
// A 64-pair trip needs the 64-byte records of 64 neighbours.  Four DMA instructions fetch them
// quad-cooperatively (lane 4q+k of instruction r loads one 16-byte piece of the record of pair 4q+r:
// one 64-byte access per quad) straight into a per-wave LDS tile -- no destination registers, no
// ds_write, no shuffles; the index of pair 4q+r sits in lane 4q+r of the coalesced index load and
// reaches its quad with one quad_perm DPP move.  Every lane then reads ITS pair's three pieces
// {x,mu_x} {y,mu_y} {z,mu_z} with three ds_read_b128 and does the whole pair: 64 pairs per vector
// instruction, no quad hand-round, no idle lane, r^2 from the lane's own displacement (so nothing but
// the 4-byte index is streamed per pair: SURVEY 8(d)'s algorithmic stream).
// LDS image of a trip (4 KB): block r (1 KB) = DMA instruction r, written lane-linearly (lane L -> 16-byte
// slot L).  Lane 4q+k of instruction r fetches piece k^r, so the reader (pair 4q+r) finds piece p in slot
// 4q + (p^r) of block r: the four lanes of a quad read four different slots of four different blocks and
// the ds_read_b128 lane groups ({0-3,12-15,20-27}, ...) cover all 16 slots of the 256-byte bank row --
// conflict-free.  Two tiles per wave: the DMA of trip t+1 lands while trip t is computed.
// index stream of this kernel: byte offsets (j << 6) into the record table; rows padded to whole trips with the
// offset of the DUMMY record (index n: zero dipole, so its pairs contribute exactly nothing)
template <int R>
__device__ __forceinline__ void lp_gather(const char *srcc, int joff, unsigned piece, char *tile) {
  // quad_perm broadcast of lane R's offset, folded into the add (v_add_u32_dpp): one vector instruction per gather
  const unsigned o_ = (unsigned)__builtin_amdgcn_update_dpp(0, joff, R * 0x55, 0xF, 0xF, true) + piece;
  const char *g = srcc + o_;  // scalar base + 32-bit lane offset
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                   (__attribute__((address_space(3))) void *)(tile + R * 1024), 16, 0, 0);
}
// 1/sqrt(x), x > 0 finite: v_rsq_f64 (about 2^-23 relative) and one second-order (Newton) correction: 2^-45 = 3e-14
// relative, four instructions (the third-order form bought 2^-69 for a fifth: the sweep is bound by its instruction count)
__device__ __forceinline__ double rsqrt_pos(double x) {
  const double y = __builtin_amdgcn_rsq(x);
  const double e = fma(-(x * y), y, 1.0);
  return fma(y * e, 0.5, y);
}
// exp(x), -700 <= x <= 0 after the clamp: round-to-nearest of x*log2(e) through the 1.5*2^52 shift (the integer lands
// in the low word), Cody-Waite remainder, Taylor polynomial of degree 10 (|r| <= ln2/2: truncation 2.2e-13 relative --
// the damping term e^(-ar) p(ar) it feeds is at most 0.3 of the tensor scalar, three orders inside the parity tolerance),
// and the power of two added straight into the exponent field (the result stays normal: n >= -1010).
__device__ __forceinline__ double exp_neg_fast(double x, const ExpCoef &K) {
  x = fmax(x, -700.0);
  const double shift = 6755399441055744.0;
  const double ns = fma(x, K.log2e, shift);
  const double n = ns - shift;
  double r = fma(-n, K.ln2hi, x);
  r = fma(-n, K.ln2lo, r);
  double p = K.c[10];
#pragma unroll
  for (int k = 9; k >= 0; k--) p = fma(p, r, K.c[k]);
  return __hiloint2double(__double2hiint(p) + (__double2loint(ns) << 20), __double2loint(p));
}
template <int DAMP>
__device__ __forceinline__ void tensor_scalars_lp(double r2, double pd, const ExpCoef &K, double &s3, double &s5) {
  const double rinv = rsqrt_pos(r2);
  const double rinv2 = rinv * rinv;
  const double r3 = rinv * rinv2;
  const double r5 = r3 * rinv2;
  if (DAMP == 0) {  // exponential (Thole-like) damping, PS.cpp:1287-1297
    const double ar = pd * (r2 * rinv);
    const double e = exp_neg_fast(-ar, K);
    const double ar2 = ar * ar;
    const double p2 = fma(ar2, 0.5, 1.0 + ar);
    const double p3 = fma(ar2 * ar, 1.0 / 6.0, p2);
    s3 = fma(-e, p2, 1.0) * r3;
    s5 = fma(-e, p3, 1.0) * (3.0 * r5);
  } else {
    s3 = r3;
    s5 = 3.0 * r5;
  }
}
// Row epilogue: the three field sums of the 64 lanes, the new dipole, sum (dmu)^2.  The three wave reductions are
// folded into one butterfly: after the xor-1 step a lane keeps x (even) or y (odd), after the xor-2 step lanes
// 4m+{0,1,2,3} hold {x, y, z, z}; four more steps finish all three at once (29 instead of 54 vector instructions).
struct LpSelf { double mu_old, alpha, ef; };  // lanes 0..2: component `lane` of the row atom's dipole, its alpha, its E_static
// requested when the row STARTS: the epilogue below then finds them in registers instead of paying a memory round trip
// per row after the last trip (rows are only ~8 trips long)
__device__ __forceinline__ LpSelf lp_self(int lane, const AtomRec *self, const double *efi) {
  LpSelf s{0.0, 0.0, 0.0};
  if (lane < 3) {
    const double *r = reinterpret_cast<const double *>(self);
    s.mu_old = r[2 * lane + 1]; s.alpha = r[7]; s.ef = efi[lane];
  }
  return s;
}
// DET (`deterministic yes`): the new dipole and its (dmu)^2 go to the launch row's slot of `pend` instead: no row of the
// launch then reads a dipole another wave of the same launch may or may not have written yet, and the sum of the changes is
// formed in a fixed order (k_lp_commit) instead of by atomics
template <bool DET>
__device__ __forceinline__ void lp_finish(double ax, double ay, double az, int lane, const LpSelf &self, AtomRec *out,
                                          double *slots, double omega, double *pend) {
  const bool odd = lane & 1, hi = lane & 2;
  const double keep1 = odd ? ay : ax, give1 = odd ? ax : ay;
  double v = keep1 + dpp_full<0xB1>(give1);  // quad_perm [1,0,3,2]: even lanes x(l)+x(l+1), odd lanes y(l-1)+y(l)
  double w = az + dpp_full<0xB1>(az);        // z pairs
  const double keep2 = hi ? w : v, give2 = hi ? v : w;
  v = keep2 + dpp_full<0x4E>(give2);         // quad_perm [2,3,0,1]: lanes 4m+{0,1}: x, y of the quad; 4m+{2,3}: z of the quad
  v += dpp_full<0x124>(v);                   // row_ror:4
  v += dpp_full<0x128>(v);                   // row_ror:8: every lane holds its component's 16-lane row total
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  // lanes 0, 1, 2 hold E_x, E_y, E_z of the row: mu_new = alpha (E_static + E_ind), PS.cpp:1170-1180
  double d2 = 0.0;
  if (lane < 3) {
    // omega = 1: the reference's update (PS.cpp:1170-1180); `polar_sor` (extension) over-relaxes it
    const double mu_new = fma(omega, self.alpha * (self.ef + v) - self.mu_old, self.mu_old);
    if (DET) pend[lane] = mu_new;
    else reinterpret_cast<double *>(out)[2 * lane + 1] = mu_new;
    const double d = mu_new - self.mu_old;
    d2 = d * d;
  }
  d2 += dpp_full<0xB1>(d2);
  d2 += dpp_full<0x4E>(d2);
  if (DET) { if (lane == 0) pend[3] = d2; }
  else if (lane == 0 && d2 != 0.0) atomicAdd(slot_ptr(slots, SL_CHANGE), d2);
}
// DET: fold a launch's pending dipoles into the record table; every workgroup leaves the sum of its 256 rows' (dmu)^2 --
// wave butterflies, then the four waves in order: the same association every run -- in its own slot of `part`, which the
// end-of-sweep kernel adds up in slot order
static __global__ __launch_bounds__(256) void k_lp_commit(int nrows, long long row0, const int2 *__restrict__ desc,
                                                   const double *__restrict__ pend, AtomRec *recA, AtomRec *recB, int jacobi,
                                                   const Scal *scal, double *__restrict__ part) {
  if (scal->done) return;
  __shared__ double red[4];
  AtomRec *dst = jacobi ? (scal->cur ? recA : recB) : recA;
  const int r = blockIdx.x * 256 + threadIdx.x;
  double acc = 0.0;
  if (r < nrows) {
    const int i = desc[r].x;
    const double *p = pend + 4 * (size_t)(row0 + r);
    dst[i].mx = p[0]; dst[i].my = p[1]; dst[i].mz = p[2];
    acc = p[3];
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = ((red[0] + red[1]) + red[2]) + red[3];
}
#define POLAR_LP_TILE 4096
// Index stream layout of this kernel ("chunked"): a row's entries are stored in chunks of 4 trips (256 entries);
// entry `e` of the row (trip e>>6, lane e&63) lives at  (e>>8)*256 + (e&63)*4 + ((e>>6)&3), so ONE 16-byte load
// per lane brings a lane's indices of four trips.  The first two chunks are requested when the row starts: a
// typical row (8 trips) has its whole index stream in flight at once instead of one HBM latency per trip
// (measured: with a 4-byte load per trip the loop took 110 us per sweep at 135k atoms with neither gathers nor
// arithmetic in it).  The pitch of the dd rows is a multiple of 256 in this mode.
// (lp_slot: polar_common.hpp)

// one 64-pair trip: this lane's record out of tile `cur`, the DMA of the next trip (index jnext) into `nxt`, the pair arithmetic
// WRAP: 0 the row has no pair across a periodic face, 1 orthogonal minimum image, 2 tilted box (list mode in a
// triclinic cell: whole lattice vectors c, b, a taken off in that order, as min_image_rint)
template <int WRAP, int DAMP, int NT>
__device__ __forceinline__ void lp_trip(const char *rd0, const char *rd1, const char *rd2, int cur, int nxt, bool more,
                                        const char *srcc, int jnext, unsigned g0, unsigned g1, unsigned g2, unsigned g3,
                                        char *tile0, const AtomRec &ri, double px, double py, double pz, const Box &box,
                                        double pd, const ExpCoef &K, double &ax, double &ay, double &az POLAR_LAB_PARAM) {
  // the compiler waits vmcnt(0) here: the DMA of this trip (issued one trip ago)
  const double2 A = *reinterpret_cast<const double2 *>(rd0 + cur);
  const double2 B = *reinterpret_cast<const double2 *>(rd1 + cur);
  const double2 C = *reinterpret_cast<const double2 *>(rd2 + cur);
  if (more && !POLAR_ABL(8)) {  // wave-uniform
    if (NT == 1) __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): the reads above are done before the tile is rewritten
    char *nt = tile0 + nxt;
    const int jg = POLAR_ABL(4) ? 0 : jnext;  // lab: every gather hits record 0
    lp_gather<0>(srcc, jg, g0, nt); lp_gather<1>(srcc, jg, g1, nt);
    lp_gather<2>(srcc, jg, g2, nt); lp_gather<3>(srcc, jg, g3, nt);
  }
  if (POLAR_ABL(16)) { ax += A.x + B.y + C.x; return; }  // lab: no pair arithmetic
  double dx = ri.x - A.x, dy = ri.y - B.x, dz = ri.z - C.x;
  if (WRAP == 1) {
    dx = fma(-px, rint(dx * box.inv[0]), dx);
    dy = fma(-py, rint(dy * box.inv[1]), dy);
    dz = fma(-pz, rint(dz * box.inv[2]), dz);
  } else if (WRAP == 2) {
    const double nz = pz != 0.0 ? rint(dz * box.inv[2]) : 0.0;
    dz = fma(-pz, nz, dz); dy = fma(-box.yz, nz, dy); dx = fma(-box.xz, nz, dx);
    const double ny = py != 0.0 ? rint(dy * box.inv[1]) : 0.0;
    dy = fma(-py, ny, dy); dx = fma(-box.xy, ny, dx);
    dx = fma(-px, rint(dx * box.inv[0]), dx);
  }
  const double r2 = fmax(fma(dx, dx, fma(dy, dy, dz * dz)), 1e-12);  // the dummy record may coincide with the row atom
  double s3, s5;
  tensor_scalars_lp<DAMP>(r2, pd, K, s3, s5);
  const double dot = fma(A.y, dx, fma(B.y, dy, C.y * dz));
  const double cc = s5 * dot;
  ax = fma(cc, dx, fma(-s3, A.y, ax));
  ay = fma(cc, dy, fma(-s3, B.y, ay));
  az = fma(cc, dz, fma(-s3, C.y, az));
}
__device__ __forceinline__ void lp_first_gather(const char *srcc, int joff, int lane, char *tile0) {
  const int k = lane & 3;
  lp_gather<0>(srcc, joff, (unsigned)(k * 16), tile0); lp_gather<1>(srcc, joff, (unsigned)((k ^ 1) * 16), tile0);
  lp_gather<2>(srcc, joff, (unsigned)((k ^ 2) * 16), tile0); lp_gather<3>(srcc, joff, (unsigned)((k ^ 3) * 16), tile0);
}
template <int WRAP, int DAMP, int NT>
__device__ __forceinline__ void lp_row(int T, const int4 *pc, const int4 &Ja0, const int4 &Jb0, const char *srcc, char *tile0, int lane, const AtomRec &ri,
                                       const Box &box, double pd, const ExpCoef &K, double &ax, double &ay, double &az
                                       POLAR_LAB_PARAM) {
  const double px = box.periodic[0] ? box.prd[0] : 0.0, py = box.periodic[1] ? box.prd[1] : 0.0,
               pz = box.periodic[2] ? box.prd[2] : 0.0;
  const int k = lane & 3, q = lane >> 2;
  const unsigned g0 = (unsigned)(k * 16), g1 = (unsigned)((k ^ 1) * 16), g2 = (unsigned)((k ^ 2) * 16), g3 = (unsigned)((k ^ 3) * 16);
  // this lane is pair 4q+k of a trip: its record is in block k, piece p in slot 4q + (p^k)
  const char *rd0 = tile0 + k * 1024 + (4 * q + k) * 16;
  const char *rd1 = tile0 + k * 1024 + (4 * q + (k ^ 1)) * 16;
  const char *rd2 = tile0 + k * 1024 + (4 * q + (k ^ 2)) * 16;
  if (T <= 0) return;
  const int C = (T + 3) >> 2;  // chunks of four trips; lane L's int4 of chunk c is pc[64 c]
  int4 Ja = Ja0, Jb = Jb0, Jc = Jb0;  // the first two chunks were requested before the row descriptor arrived;
  // the gathers of trip 0 are already in flight (lp_first_gather, issued before the row atom's own data was requested)
  const int other = NT == 1 ? 0 : POLAR_LP_TILE;
#define POLAR_LP_TRIP(CUR, NXT, TT, JNEXT)                                                                              \
  lp_trip<WRAP, DAMP, NT>(rd0, rd1, rd2, CUR, NXT, t0 + (TT) + 1 < T, srcc, JNEXT, g0, g1, g2, g3, tile0, ri, px, py, pz, \
                          box, pd, K, ax, ay, az POLAR_LAB_PASS)
  for (int c = 0; c < C; c++) {
    const int t0 = 4 * c;
    if (c + 2 < C) Jc = pc[64 * (c + 2)];  // rows longer than 8 trips: two chunks ahead
    POLAR_LP_TRIP(0, other, 0, Ja.y);
    if (t0 + 1 >= T) break;
    POLAR_LP_TRIP(other, 0, 1, Ja.z);
    if (t0 + 2 >= T) break;
    POLAR_LP_TRIP(0, other, 2, Ja.w);
    if (t0 + 3 >= T) break;
    POLAR_LP_TRIP(other, 0, 3, Jb.x);
    Ja = Jb; Jb = Jc;
  }
#undef POLAR_LP_TRIP
}
// row descriptors of a step: {row atom (s index), trips | wrap flag << 30}; one coalesced 8-byte load tells a wave
// everything it needs to start its index stream, its record fetch and its field fetch at once
static __global__ void k_lp_desc(int nrows, const int *__restrict__ rows, RowList ddl, const int *__restrict__ dd_wrap,
                          int2 *__restrict__ desc) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= nrows) return;
  const int i = rows ? rows[r] : r;
  long long c = ddl.cnt[i];
  if (c > ddl.pitch) c = ddl.pitch;
  desc[r] = make_int2(i, (int)((c + 63) >> 6) | (dd_wrap[i] ? 0x40000000 : 0));
}
template <int EP, int DAMP, int NT, bool DET>
static __global__ __launch_bounds__(1024) void k_field_lp(int nrows, long long row0, const int2 *__restrict__ desc, AtomRec *recA,
                                                   AtomRec *recB, Box box, long long pitch,
                                                   const int *__restrict__ dd_j, double pd, ExpCoef K,
                                                   const double *__restrict__ ef, const Scal *scal,
                                                   double *__restrict__ slots, double omega, double *pend POLAR_LAB_PARAM) {
  extern __shared__ __attribute__((aligned(16))) char lp_lds[];
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int rpb = blockDim.x >> 6;
  const int lb = xcd_block(blockIdx.x, (nrows + rpb - 1) / rpb);
  if (lb < 0) return;
  const int row = lb * rpb + wv;
  if (row >= nrows) return;
  // A row starts with a chain of dependent round trips (descriptor -> row atom's record -> first gathers).  Everything
  // that needs only the launch row is requested at once: the loop state, the descriptor, and the first two chunks of
  // the index stream -- dd rows are stored in launch order (k_nl_build, dd_slot), row row0 + row of this launch.
  const int4 *pc = reinterpret_cast<const int4 *>(dd_j + (size_t)(row0 + row) * pitch) + lane;
  const int2 de = desc[row];
  const int4 Ja0 = pc[0], Jb0 = pc[64];
  const int done = scal->done, curv = scal->cur;
  if (done) return;
  const int i = __builtin_amdgcn_readfirstlane(de.x);
  int T = __builtin_amdgcn_readfirstlane(de.y & 0xFFFF);
  const int wrapped = __builtin_amdgcn_readfirstlane(de.y >> 30) | POLAR_ABL(2);
  const int cur = __builtin_amdgcn_readfirstlane(curv);
  const AtomRec *src = (EP == EP_JACOBI && cur) ? recB : recA;
  AtomRec *dst = (EP == EP_JACOBI) ? (cur ? recA : recB) : recA;
  if (POLAR_ABL(1)) T = 0;  // lab switches (POLAR_ABLATE, lab build only): timing only, wrong numbers
  const char *srcc = reinterpret_cast<const char *>(src);
  char *tile0 = lp_lds + (size_t)wv * (NT * POLAR_LP_TILE);
  if (T > 0) lp_first_gather(srcc, Ja0.x, lane, tile0);  // second round trip: the first gathers ...
  AtomRec ri;  // ... together with the row atom's position (wave-uniform, parked in scalar registers) and epilogue data
  {
    const double *r = reinterpret_cast<const double *>(src + i);
    ri.x = wave_uniform(r[0]); ri.y = wave_uniform(r[2]); ri.z = wave_uniform(r[4]);
  }
  const LpSelf self = lp_self(lane, src + i, ef + 3 * (size_t)i);
  double ax = 0.0, ay = 0.0, az = 0.0;
  // rows whose list holds no pair across a periodic face (flag written by k_nl_build) skip the minimum-image wrap
  if (!wrapped) lp_row<0, DAMP, NT>(T, pc, Ja0, Jb0, srcc, tile0, lane, ri, box, pd, K, ax, ay, az POLAR_LAB_PASS);
  else if (!box.triclinic) lp_row<1, DAMP, NT>(T, pc, Ja0, Jb0, srcc, tile0, lane, ri, box, pd, K, ax, ay, az POLAR_LAB_PASS);
  else lp_row<2, DAMP, NT>(T, pc, Ja0, Jb0, srcc, tile0, lane, ri, box, pd, K, ax, ay, az POLAR_LAB_PASS);
  lp_finish<DET>(ax, ay, az, lane, self, dst + i, slots, omega, DET ? pend + 4 * (size_t)(row0 + row) : nullptr);
}

#ifdef POLAR_LAB
// ------------------------------------------------------------------------------------------
// k_field_lpr: the same row sweep, `R` consecutive launch rows per wave, the next row's start hidden behind the current
// row's trips.  A row of k_field_lp begins with two dependent round trips (descriptor + index chunks, then the row atom's
// record + the first gathers) and ends with a reduction: with rows only ~8 trips long a wave computes for about half of its
// life.  Here trip 0 of a row also requests the next row's descriptor and first two index chunks, trip 1 its record and
// static field, and the LAST trip issues the next row's first gathers into the tile it would have used for "the next trip" --
// across a row boundary the wave sees one uninterrupted stream of trips, and the epilogue of row r runs while the gathers of
// row r+1 are in flight.  Tiles alternate per trip, so after a row with an odd number of trips the two tile roles swap.
// LAB ONLY -- it loses: 246 / 250 / 261 / 291 us per sweep at R = 2 / 3 / 4 / 6 against 229.5 for k_field_lp at 135k atoms
// (profiles/r03_lab_lpr_rows_per_wave.txt): 115 registers (four waves per SIMD instead of six) and a longer drain cost more
// than the hidden row starts return -- the sweep is not waiting on its row prologues.
struct LpNext { int2 de; int4 Ja, Jb; double x, y, z; LpSelf self; };
__device__ __forceinline__ void lpr_gather4(const char *srcc, int joff, unsigned g0, unsigned g1, unsigned g2, unsigned g3, char *tile) {
  lp_gather<0>(srcc, joff, g0, tile); lp_gather<1>(srcc, joff, g1, tile);
  lp_gather<2>(srcc, joff, g2, tile); lp_gather<3>(srcc, joff, g3, tile);
}
template <int WRAP, int DAMP>
__device__ __forceinline__ void lp_pair_math(const double2 &A, const double2 &B, const double2 &C, double rix, double riy, double riz,
                                             double px, double py, double pz, const Box &box, double pd, const ExpCoef &K,
                                             double &ax, double &ay, double &az) {
  double dx = rix - A.x, dy = riy - B.x, dz = riz - C.x;
  if (WRAP == 1) {
    dx = fma(-px, rint(dx * box.inv[0]), dx);
    dy = fma(-py, rint(dy * box.inv[1]), dy);
    dz = fma(-pz, rint(dz * box.inv[2]), dz);
  } else if (WRAP == 2) {
    const double nz = pz != 0.0 ? rint(dz * box.inv[2]) : 0.0;
    dz = fma(-pz, nz, dz); dy = fma(-box.yz, nz, dy); dx = fma(-box.xz, nz, dx);
    const double ny = py != 0.0 ? rint(dy * box.inv[1]) : 0.0;
    dy = fma(-py, ny, dy); dx = fma(-box.xy, ny, dx);
    dx = fma(-px, rint(dx * box.inv[0]), dx);
  }
  const double r2 = fmax(fma(dx, dx, fma(dy, dy, dz * dz)), 1e-12);  // the dummy record may coincide with the row atom
  double s3, s5;
  tensor_scalars_lp<DAMP>(r2, pd, K, s3, s5);
  const double dot = fma(A.y, dx, fma(B.y, dy, C.y * dz));
  const double cc = s5 * dot;
  ax = fma(cc, dx, fma(-s3, A.y, ax));
  ay = fma(cc, dy, fma(-s3, B.y, ay));
  az = fma(cc, dz, fma(-s3, C.y, az));
}
// next row, stage 1 (needs only the launch row number) and stage 2 (needs the descriptor)
__device__ __forceinline__ void lpr_stage1(LpNext &N, const int2 *descN, const int4 *pcN) {
  N.de = *descN; N.Ja = pcN[0]; N.Jb = pcN[64];
}
__device__ __forceinline__ void lpr_stage2(LpNext &N, int lane, const AtomRec *src, const double *ef) {
  const int iN = __builtin_amdgcn_readfirstlane(N.de.x);
  const double *r = reinterpret_cast<const double *>(src + iN);
  N.x = r[0]; N.y = r[2]; N.z = r[4];
  N.self = lp_self(lane, src + iN, ef + 3 * (size_t)iN);
}
template <int WRAP, int DAMP>
__device__ __forceinline__ void lpr_row(int T, const int4 *pc, const int4 &Ja0, const int4 &Jb0, const char *srcc, char *lds, int lane,
                                        double rix, double riy, double riz, const Box &box, double pd, const ExpCoef &K,
                                        int oa0, int oa1, int oa2, int ob0, int ob1, int ob2, int ta, int tb, bool has_next,
                                        const int2 *descN, const int4 *pcN, const AtomRec *src, const double *ef, LpNext &N,
                                        bool &gatheredN, double &ax, double &ay, double &az) {
  const double px = box.periodic[0] ? box.prd[0] : 0.0, py = box.periodic[1] ? box.prd[1] : 0.0,
               pz = box.periodic[2] ? box.prd[2] : 0.0;
  const int k = lane & 3;
  const unsigned g0 = (unsigned)(k * 16), g1 = (unsigned)((k ^ 1) * 16), g2 = (unsigned)((k ^ 2) * 16), g3 = (unsigned)((k ^ 3) * 16);
  if (T <= 0) return;
  const int C = (T + 3) >> 2;
  int4 Ja = Ja0, Jb = Jb0, Jc = Jb0;
#define POLAR_LPR_TRIP(U, JNEXT)                                                                                        \
  {                                                                                                                     \
    const int tt = t0 + (U);                                                                                            \
    const double2 A = *reinterpret_cast<const double2 *>(lds + (((U) & 1) ? ob0 : oa0));                                \
    const double2 B = *reinterpret_cast<const double2 *>(lds + (((U) & 1) ? ob1 : oa1));                                \
    const double2 Cc = *reinterpret_cast<const double2 *>(lds + (((U) & 1) ? ob2 : oa2));                               \
    char *nt_ = lds + (((U) & 1) ? ta : tb);                                                                            \
    if (tt + 1 < T) lpr_gather4(srcc, JNEXT, g0, g1, g2, g3, nt_);                                                      \
    else if (has_next && tt >= 1) {  /* the last trip: the next row's first records (its stage 1 left with trip 0) */   \
      if (__builtin_amdgcn_readfirstlane(N.de.y & 0xFFFF) > 0) lpr_gather4(srcc, N.Ja.x, g0, g1, g2, g3, nt_);          \
      gatheredN = true;                                                                                                 \
    }                                                                                                                   \
    if (has_next && tt == 0) lpr_stage1(N, descN, pcN);                                                                 \
    if (has_next && tt == 1) lpr_stage2(N, lane, src, ef);                                                              \
    lp_pair_math<WRAP, DAMP>(A, B, Cc, rix, riy, riz, px, py, pz, box, pd, K, ax, ay, az);                              \
  }
  for (int c = 0; c < C; c++) {
    const int t0 = 4 * c;
    if (c + 2 < C) Jc = pc[64 * (c + 2)];  // rows longer than 8 trips: two chunks ahead
    POLAR_LPR_TRIP(0, Ja.y);
    if (t0 + 1 >= T) break;
    POLAR_LPR_TRIP(1, Ja.z);
    if (t0 + 2 >= T) break;
    POLAR_LPR_TRIP(2, Ja.w);
    if (t0 + 3 >= T) break;
    POLAR_LPR_TRIP(3, Jb.x);
    Ja = Jb; Jb = Jc;
  }
#undef POLAR_LPR_TRIP
}
template <int EP, int DAMP, bool DET>
static __global__ __launch_bounds__(256) void k_field_lpr(int nrows, long long row0, const int2 *__restrict__ desc, AtomRec *recA,
                                                   AtomRec *recB, Box box, long long pitch, const int *__restrict__ dd_j, double pd,
                                                   ExpCoef K, const double *__restrict__ ef, const Scal *scal,
                                                   double *__restrict__ slots, double omega, double *pend, int R) {
  extern __shared__ __attribute__((aligned(16))) char lp_lds[];
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int rpb = blockDim.x >> 6;
  const int per = rpb * R;
  const int lb = xcd_block(blockIdx.x, (nrows + per - 1) / per);
  if (lb < 0) return;
  int row = (lb * rpb + wv) * R;  // R consecutive launch rows
  if (row >= nrows) return;
  const int last = min(row + R, nrows);
  const int4 *pc = reinterpret_cast<const int4 *>(dd_j + (size_t)(row0 + row) * pitch) + lane;
  const int2 de0 = desc[row];
  int4 Ja = pc[0], Jb = pc[64];
  const int done = scal->done, curv = scal->cur;
  if (done) return;
  int i = __builtin_amdgcn_readfirstlane(de0.x);
  int T = __builtin_amdgcn_readfirstlane(de0.y & 0xFFFF);
  int wrapped = __builtin_amdgcn_readfirstlane(de0.y >> 30);
  const int cur = __builtin_amdgcn_readfirstlane(curv);
  const AtomRec *src = (EP == EP_JACOBI && cur) ? recB : recA;
  AtomRec *dst = (EP == EP_JACOBI) ? (cur ? recA : recB) : recA;
  const char *srcc = reinterpret_cast<const char *>(src);
  const int k = lane & 3, q = lane >> 2;
  int ta = wv * (2 * POLAR_LP_TILE), tb = ta + POLAR_LP_TILE;
  // this lane is pair 4q+k of a trip: its record is in block k, piece p in slot 4q + (p^k)
  int oa0 = ta + k * 1024 + (4 * q + k) * 16, oa1 = ta + k * 1024 + (4 * q + (k ^ 1)) * 16, oa2 = ta + k * 1024 + (4 * q + (k ^ 2)) * 16;
  int ob0 = oa0 + POLAR_LP_TILE, ob1 = oa1 + POLAR_LP_TILE, ob2 = oa2 + POLAR_LP_TILE;
  if (T > 0) lp_first_gather(srcc, Ja.x, lane, lp_lds + ta);
  double rix, riy, riz;
  {
    const double *r = reinterpret_cast<const double *>(src + i);
    rix = wave_uniform(r[0]); riy = wave_uniform(r[2]); riz = wave_uniform(r[4]);
  }
  LpSelf self = lp_self(lane, src + i, ef + 3 * (size_t)i);
  const int unit = lane & 3;
  const unsigned g0 = (unsigned)(unit * 16), g1 = (unsigned)((unit ^ 1) * 16), g2 = (unsigned)((unit ^ 2) * 16), g3 = (unsigned)((unit ^ 3) * 16);
  for (;;) {
    const bool has_next = row + 1 < last;
    LpNext N;
    N.de = make_int2(0, 0); N.Ja = Ja; N.Jb = Jb; N.x = N.y = N.z = 0.0; N.self = self;
    bool gatheredN = false;
    const int2 *descN = desc + row + 1;
    const int4 *pcN = pc + (pitch >> 2);
    double ax = 0.0, ay = 0.0, az = 0.0;
    if (!wrapped) lpr_row<0, DAMP>(T, pc, Ja, Jb, srcc, lp_lds, lane, rix, riy, riz, box, pd, K, oa0, oa1, oa2, ob0, ob1, ob2, ta, tb, has_next, descN, pcN, src, ef, N, gatheredN, ax, ay, az);
    else if (!box.triclinic) lpr_row<1, DAMP>(T, pc, Ja, Jb, srcc, lp_lds, lane, rix, riy, riz, box, pd, K, oa0, oa1, oa2, ob0, ob1, ob2, ta, tb, has_next, descN, pcN, src, ef, N, gatheredN, ax, ay, az);
    else lpr_row<2, DAMP>(T, pc, Ja, Jb, srcc, lp_lds, lane, rix, riy, riz, box, pd, K, oa0, oa1, oa2, ob0, ob1, ob2, ta, tb, has_next, descN, pcN, src, ef, N, gatheredN, ax, ay, az);
    if (has_next) {  // rows shorter than two trips leave stages of the next row's start to do
      if (T < 1) lpr_stage1(N, descN, pcN);
      if (T < 2) lpr_stage2(N, lane, src, ef);
      if (!gatheredN && __builtin_amdgcn_readfirstlane(N.de.y & 0xFFFF) > 0) lpr_gather4(srcc, N.Ja.x, g0, g1, g2, g3, lp_lds + ((T & 1) ? tb : ta));
    }
    lp_finish<DET>(ax, ay, az, lane, self, dst + i, slots, omega, DET ? pend + 4 * (size_t)(row0 + row) : nullptr);
    if (!has_next) break;
    if (T & 1) {  // the next row's trip 0 reads the tile this row's last trip filled
      int t_;
      t_ = ta; ta = tb; tb = t_;
      t_ = oa0; oa0 = ob0; ob0 = t_; t_ = oa1; oa1 = ob1; ob1 = t_; t_ = oa2; oa2 = ob2; ob2 = t_;
    }
    row++;
    pc = pcN;
    i = __builtin_amdgcn_readfirstlane(N.de.x);
    T = __builtin_amdgcn_readfirstlane(N.de.y & 0xFFFF);
    wrapped = __builtin_amdgcn_readfirstlane(N.de.y >> 30);
    Ja = N.Ja; Jb = N.Jb;
    rix = wave_uniform(N.x); riy = wave_uniform(N.y); riz = wave_uniform(N.z);
    self = N.self;
  }
}

// ------------------------------------------------------------------------------------------
// k_field_lp2 (lab): PAIRED ROWS -- one wave sweeps two rows of one colour phase that share a cell over the union of their
// neighbours (polar_lists.hpp, k_dd_units).  Same trip structure as k_field_lp; a gathered record is used for both rows, an
// entry's two low bits say which of the rows it belongs to (the other one's tensor scalars are multiplied by zero).
// LAB ONLY -- no gain: 49.9 us per launch against 50.4 with 0.64 x the gathers and half the waves, and the per-step union
// build (0.5 ms) on top (profiles/r03_lab_paired_rows.txt).  Forcing five waves per SIMD (95 registers, 4 spilled) changes nothing.
template <int WRAP, int DAMP>
__device__ __forceinline__ void lp2_trip(const char *rd0, const char *rd1, const char *rd2, int cur, int nxt, bool more, const char *srcc,
                                         int jcur, int jnext, unsigned g0, unsigned g1, unsigned g2, unsigned g3, char *tile0,
                                         double ax_, double ay_, double az_, double bx_, double by_, double bz_, bool twoB,
                                         double px, double py, double pz, const Box &box, double pd, const ExpCoef &K,
                                         double &aax, double &aay, double &aaz, double &bax, double &bay, double &baz) {
  const double2 A = *reinterpret_cast<const double2 *>(rd0 + cur);
  const double2 B = *reinterpret_cast<const double2 *>(rd1 + cur);
  const double2 C = *reinterpret_cast<const double2 *>(rd2 + cur);
  if (more) {  // wave-uniform
    char *nt = tile0 + nxt;
    const int jg = jnext & ~63;
    lp_gather<0>(srcc, jg, g0, nt); lp_gather<1>(srcc, jg, g1, nt);
    lp_gather<2>(srcc, jg, g2, nt); lp_gather<3>(srcc, jg, g3, nt);
  }
  const double mA = (double)(jcur & 1), mB = (double)((jcur >> 1) & 1);
  {
    double dx = ax_ - A.x, dy = ay_ - B.x, dz = az_ - C.x;
    if (WRAP == 1) {
      dx = fma(-px, rint(dx * box.inv[0]), dx); dy = fma(-py, rint(dy * box.inv[1]), dy); dz = fma(-pz, rint(dz * box.inv[2]), dz);
    } else if (WRAP == 2) {
      const double nz = pz != 0.0 ? rint(dz * box.inv[2]) : 0.0;
      dz = fma(-pz, nz, dz); dy = fma(-box.yz, nz, dy); dx = fma(-box.xz, nz, dx);
      const double ny = py != 0.0 ? rint(dy * box.inv[1]) : 0.0;
      dy = fma(-py, ny, dy); dx = fma(-box.xy, ny, dx);
      dx = fma(-px, rint(dx * box.inv[0]), dx);
    }
    const double r2 = fmax(fma(dx, dx, fma(dy, dy, dz * dz)), 1e-12);
    double s3, s5;
    tensor_scalars_lp<DAMP>(r2, pd, K, s3, s5);
    s3 *= mA; s5 *= mA;
    const double dot = fma(A.y, dx, fma(B.y, dy, C.y * dz));
    const double cc = s5 * dot;
    aax = fma(cc, dx, fma(-s3, A.y, aax)); aay = fma(cc, dy, fma(-s3, B.y, aay)); aaz = fma(cc, dz, fma(-s3, C.y, aaz));
  }
  if (twoB) {  // wave-uniform: units with one row skip the second half
    double dx = bx_ - A.x, dy = by_ - B.x, dz = bz_ - C.x;
    if (WRAP == 1) {
      dx = fma(-px, rint(dx * box.inv[0]), dx); dy = fma(-py, rint(dy * box.inv[1]), dy); dz = fma(-pz, rint(dz * box.inv[2]), dz);
    } else if (WRAP == 2) {
      const double nz = pz != 0.0 ? rint(dz * box.inv[2]) : 0.0;
      dz = fma(-pz, nz, dz); dy = fma(-box.yz, nz, dy); dx = fma(-box.xz, nz, dx);
      const double ny = py != 0.0 ? rint(dy * box.inv[1]) : 0.0;
      dy = fma(-py, ny, dy); dx = fma(-box.xy, ny, dx);
      dx = fma(-px, rint(dx * box.inv[0]), dx);
    }
    const double r2 = fmax(fma(dx, dx, fma(dy, dy, dz * dz)), 1e-12);
    double s3, s5;
    tensor_scalars_lp<DAMP>(r2, pd, K, s3, s5);
    s3 *= mB; s5 *= mB;
    const double dot = fma(A.y, dx, fma(B.y, dy, C.y * dz));
    const double cc = s5 * dot;
    bax = fma(cc, dx, fma(-s3, A.y, bax)); bay = fma(cc, dy, fma(-s3, B.y, bay)); baz = fma(cc, dz, fma(-s3, C.y, baz));
  }
}
template <int WRAP, int DAMP>
__device__ __forceinline__ void lp2_row(int T, const int4 *pc, const int4 &Ja0, const int4 &Jb0, const char *srcc, char *tile0, int lane,
                                        double ax_, double ay_, double az_, double bx_, double by_, double bz_, bool twoB,
                                        const Box &box, double pd, const ExpCoef &K,
                                        double &aax, double &aay, double &aaz, double &bax, double &bay, double &baz) {
  const double px = box.periodic[0] ? box.prd[0] : 0.0, py = box.periodic[1] ? box.prd[1] : 0.0,
               pz = box.periodic[2] ? box.prd[2] : 0.0;
  const int k = lane & 3, q = lane >> 2;
  const unsigned g0 = (unsigned)(k * 16), g1 = (unsigned)((k ^ 1) * 16), g2 = (unsigned)((k ^ 2) * 16), g3 = (unsigned)((k ^ 3) * 16);
  const char *rd0 = tile0 + k * 1024 + (4 * q + k) * 16;
  const char *rd1 = tile0 + k * 1024 + (4 * q + (k ^ 1)) * 16;
  const char *rd2 = tile0 + k * 1024 + (4 * q + (k ^ 2)) * 16;
  if (T <= 0) return;
  const int C = (T + 3) >> 2;
  int4 Ja = Ja0, Jb = Jb0, Jc = Jb0;
#define POLAR_LP2_TRIP(CUR, NXT, TT, JCUR, JNEXT)                                                                         \
  lp2_trip<WRAP, DAMP>(rd0, rd1, rd2, CUR, NXT, t0 + (TT) + 1 < T, srcc, JCUR, JNEXT, g0, g1, g2, g3, tile0, ax_, ay_, az_, \
                       bx_, by_, bz_, twoB, px, py, pz, box, pd, K, aax, aay, aaz, bax, bay, baz)
  for (int c = 0; c < C; c++) {
    const int t0 = 4 * c;
    if (c + 2 < C) Jc = pc[64 * (c + 2)];
    POLAR_LP2_TRIP(0, POLAR_LP_TILE, 0, Ja.x, Ja.y);
    if (t0 + 1 >= T) break;
    POLAR_LP2_TRIP(POLAR_LP_TILE, 0, 1, Ja.y, Ja.z);
    if (t0 + 2 >= T) break;
    POLAR_LP2_TRIP(0, POLAR_LP_TILE, 2, Ja.z, Ja.w);
    if (t0 + 3 >= T) break;
    POLAR_LP2_TRIP(POLAR_LP_TILE, 0, 3, Ja.w, Jb.x);
    Ja = Jb; Jb = Jc;
  }
#undef POLAR_LP2_TRIP
}
template <int EP, int DAMP>
static __global__ __launch_bounds__(256) void k_field_lp2(int nunits, long long unit0, const int4 *__restrict__ udesc, AtomRec *recA, AtomRec *recB,
                                                   Box box, long long upitch, const int *__restrict__ udd_j, double pd, ExpCoef K,
                                                   const double *__restrict__ ef, const Scal *scal, double *__restrict__ slots, double omega) {
  extern __shared__ __attribute__((aligned(16))) char lp_lds[];
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int rpb = blockDim.x >> 6;
  const int lb = xcd_block(blockIdx.x, (nunits + rpb - 1) / rpb);
  if (lb < 0) return;
  const int u = lb * rpb + wv;
  if (u >= nunits) return;
  const int4 *pc = reinterpret_cast<const int4 *>(udd_j + (size_t)(unit0 + u) * upitch) + lane;
  const int4 de = udesc[unit0 + u];
  const int4 Ja0 = pc[0], Jb0 = pc[64];
  const int done = scal->done, curv = scal->cur;
  if (done) return;
  const int iA = __builtin_amdgcn_readfirstlane(de.x), iB = __builtin_amdgcn_readfirstlane(de.y);
  const int T = __builtin_amdgcn_readfirstlane(de.z & 0xFFFF);
  const int wrapped = __builtin_amdgcn_readfirstlane(de.z >> 30);
  const int cur = __builtin_amdgcn_readfirstlane(curv);
  const AtomRec *src = (EP == EP_JACOBI && cur) ? recB : recA;
  AtomRec *dst = (EP == EP_JACOBI) ? (cur ? recA : recB) : recA;
  const char *srcc = reinterpret_cast<const char *>(src);
  char *tile0 = lp_lds + (size_t)wv * (2 * POLAR_LP_TILE);
  if (T > 0) lp_first_gather(srcc, Ja0.x & ~63, lane, tile0);
  const bool twoB = iB >= 0;
  const int iBs = twoB ? iB : iA;
  double ax_, ay_, az_, bx_, by_, bz_;
  {
    const double *r = reinterpret_cast<const double *>(src + iA);
    ax_ = wave_uniform(r[0]); ay_ = wave_uniform(r[2]); az_ = wave_uniform(r[4]);
    const double *rb = reinterpret_cast<const double *>(src + iBs);
    bx_ = wave_uniform(rb[0]); by_ = wave_uniform(rb[2]); bz_ = wave_uniform(rb[4]);
  }
  const LpSelf selfA = lp_self(lane, src + iA, ef + 3 * (size_t)iA);
  const LpSelf selfB = lp_self(lane, src + iBs, ef + 3 * (size_t)iBs);
  double aax = 0.0, aay = 0.0, aaz = 0.0, bax = 0.0, bay = 0.0, baz = 0.0;
  if (!wrapped) lp2_row<0, DAMP>(T, pc, Ja0, Jb0, srcc, tile0, lane, ax_, ay_, az_, bx_, by_, bz_, twoB, box, pd, K, aax, aay, aaz, bax, bay, baz);
  else if (!box.triclinic) lp2_row<1, DAMP>(T, pc, Ja0, Jb0, srcc, tile0, lane, ax_, ay_, az_, bx_, by_, bz_, twoB, box, pd, K, aax, aay, aaz, bax, bay, baz);
  else lp2_row<2, DAMP>(T, pc, Ja0, Jb0, srcc, tile0, lane, ax_, ay_, az_, bx_, by_, bz_, twoB, box, pd, K, aax, aay, aaz, bax, bay, baz);
  lp_finish<false>(aax, aay, aaz, lane, selfA, dst + iA, slots, omega, nullptr);
  if (twoB) lp_finish<false>(bax, bay, baz, lane, selfB, dst + iB, slots, omega, nullptr);
}

// ------------------------------------------------------------------------------------------
// Cluster sweep: one wave = one CLUSTER of up to four rows (polar_lists.hpp, k_cl_build) against the union of
// their neighbours.  Per 64-neighbour trip the gather, the index stream and the LDS reads are paid once and the
// pair arithmetic M times (a pair outside the dd cutoff of a member is switched off through its r^2), so the bytes
// that go through the gather path per computed pair drop by M x (list efficiency).
// The members of a cluster are closer than the colour distance, so they must not be updated Jacobi-style against
// each other (colour-phase GS diverges for couplings closer than ~1.2 A): the list holds no member, the wave adds the
// in-cluster fields itself and updates the members ONE AFTER THE OTHER with the newest dipoles -- exactly the
// reference's sequential rule (PS.cpp:1158-1180) restricted to the cluster.  Across clusters of one colour the
// phase is Jacobi-like as before (clusters of a colour are farther apart than the colour distance).
template <int DAMP>
__device__ __forceinline__ void cl_pair(double xm, double ym, double zm, const double2 &A, const double2 &B, const double2 &C,
                                        bool wrap, double px, double py, double pz, const Box &box, double ddcutsq, double pd,
                                        const ExpCoef &K, double &ax, double &ay, double &az) {
  double dx = xm - A.x, dy = ym - B.x, dz = zm - C.x;
  if (wrap) {
    dx = fma(-px, rint(dx * box.inv[0]), dx);
    dy = fma(-py, rint(dy * box.inv[1]), dy);
    dz = fma(-pz, rint(dz * box.inv[2]), dz);
  }
  double r2 = fma(dx, dx, fma(dy, dy, dz * dz));
  r2 = r2 < ddcutsq ? r2 : 1e30;  // outside this member's cutoff: s3 ~ 1e-45, s5 ~ 1e-75 -- below every ulp of the sums
  r2 = fmax(r2, 1e-12);           // the dummy record may coincide with a member
  double s3, s5;
  tensor_scalars_lp<DAMP>(r2, pd, K, s3, s5);
  const double dot = fma(A.y, dx, fma(B.y, dy, C.y * dz));
  const double cc = s5 * dot;
  ax = fma(cc, dx, fma(-s3, A.y, ax));
  ay = fma(cc, dy, fma(-s3, B.y, ay));
  az = fma(cc, dz, fma(-s3, C.y, az));
}
// the three wave sums of one member: lanes 0, 1, 2 (and every lane = its lane&3 class; class 3 = z) get x, y, z
__device__ __forceinline__ double cl_reduce3(double ax, double ay, double az, int lane) {
  const bool odd = lane & 1, hi = lane & 2;
  const double keep1 = odd ? ay : ax, give1 = odd ? ax : ay;
  double v = keep1 + dpp_full<0xB1>(give1);
  double w = az + dpp_full<0xB1>(az);
  const double keep2 = hi ? w : v, give2 = hi ? v : w;
  v = keep2 + dpp_full<0x4E>(give2);
  v += dpp_full<0x124>(v);
  v += dpp_full<0x128>(v);
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}
__device__ __forceinline__ double lane_value(double v, int k) {  // k compile-time
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), k);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), k);
  return __hiloint2double(hi, lo);
}
template <bool WRAP, int DAMP, int NT>
__device__ __forceinline__ void cl_rows(int T, int M, const int4 *pc, const char *srcc, char *tile0, int lane,
                                        const double (&xm)[4], const double (&ym)[4], const double (&zm)[4], const Box &box,
                                        double ddcutsq, double pd, const ExpCoef &K, double (&ax)[4], double (&ay)[4],
                                        double (&az)[4]) {
  const double px = box.periodic[0] ? box.prd[0] : 0.0, py = box.periodic[1] ? box.prd[1] : 0.0,
               pz = box.periodic[2] ? box.prd[2] : 0.0;
  const int k = lane & 3, q = lane >> 2;
  const unsigned g0 = (unsigned)(k * 16), g1 = (unsigned)((k ^ 1) * 16), g2 = (unsigned)((k ^ 2) * 16), g3 = (unsigned)((k ^ 3) * 16);
  const char *rd0 = tile0 + k * 1024 + (4 * q + k) * 16;
  const char *rd1 = tile0 + k * 1024 + (4 * q + (k ^ 1)) * 16;
  const char *rd2 = tile0 + k * 1024 + (4 * q + (k ^ 2)) * 16;
  if (T <= 0) return;
  const int C = (T + 3) >> 2;
  int4 Ja = pc[0], Jb = make_int4(0, 0, 0, 0), Jc = Jb;
  if (C > 1) Jb = pc[64];
  lp_gather<0>(srcc, Ja.x, g0, tile0); lp_gather<1>(srcc, Ja.x, g1, tile0);
  lp_gather<2>(srcc, Ja.x, g2, tile0); lp_gather<3>(srcc, Ja.x, g3, tile0);
  const int other = NT == 1 ? 0 : POLAR_LP_TILE;
#define POLAR_CL_TRIP(CUR, NXT, TT, JNEXT)                                                                 \
  {                                                                                                        \
    const double2 A = *reinterpret_cast<const double2 *>(rd0 + (CUR));                                     \
    const double2 B = *reinterpret_cast<const double2 *>(rd1 + (CUR));                                     \
    const double2 Cc = *reinterpret_cast<const double2 *>(rd2 + (CUR));                                    \
    if (t0 + (TT) + 1 < T) {                                                                               \
      if (NT == 1) __builtin_amdgcn_s_waitcnt(0xc07f);                                                     \
      char *nt_ = tile0 + (NXT);                                                                           \
      lp_gather<0>(srcc, JNEXT, g0, nt_); lp_gather<1>(srcc, JNEXT, g1, nt_);                              \
      lp_gather<2>(srcc, JNEXT, g2, nt_); lp_gather<3>(srcc, JNEXT, g3, nt_);                              \
    }                                                                                                      \
    cl_pair<DAMP>(xm[0], ym[0], zm[0], A, B, Cc, WRAP, px, py, pz, box, ddcutsq, pd, K, ax[0], ay[0], az[0]); \
    if (M > 1) cl_pair<DAMP>(xm[1], ym[1], zm[1], A, B, Cc, WRAP, px, py, pz, box, ddcutsq, pd, K, ax[1], ay[1], az[1]); \
    if (M > 2) cl_pair<DAMP>(xm[2], ym[2], zm[2], A, B, Cc, WRAP, px, py, pz, box, ddcutsq, pd, K, ax[2], ay[2], az[2]); \
    if (M > 3) cl_pair<DAMP>(xm[3], ym[3], zm[3], A, B, Cc, WRAP, px, py, pz, box, ddcutsq, pd, K, ax[3], ay[3], az[3]); \
  }
  for (int c = 0; c < C; c++) {
    const int t0 = 4 * c;
    if (c + 2 < C) Jc = pc[64 * (c + 2)];
    POLAR_CL_TRIP(0, other, 0, Ja.y);
    if (t0 + 1 >= T) break;
    POLAR_CL_TRIP(other, 0, 1, Ja.z);
    if (t0 + 2 >= T) break;
    POLAR_CL_TRIP(0, other, 2, Ja.w);
    if (t0 + 3 >= T) break;
    POLAR_CL_TRIP(other, 0, 3, Jb.x);
    Ja = Jb; Jb = Jc;
  }
#undef POLAR_CL_TRIP
}
// descriptors of a step: members (s-space) come from the cluster table; {trips | wrap << 30}
static __global__ void k_cl_desc(int ncl, const int *__restrict__ cnt, long long pitch, const int *__restrict__ wrapf,
                          int *__restrict__ tw) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= ncl) return;
  long long n = cnt[c];
  if (n > pitch) n = pitch;
  tw[c] = (int)((n + 63) >> 6) | (wrapf[c] ? 0x40000000 : 0);
}
template <int EP, int DAMP, int NT>
static __global__ __launch_bounds__(256) void k_field_cl(int ncl, int first, const int4 *__restrict__ members,
                                                  const int *__restrict__ tw, AtomRec *recA, AtomRec *recB, Box box,
                                                  long long pitch, const int *__restrict__ dd_j, double ddcutsq,
                                                  double pd, ExpCoef K, const double *__restrict__ ef, const Scal *scal,
                                                  double *__restrict__ slots, int ablate) {
  extern __shared__ __attribute__((aligned(16))) char lp_lds[];
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int rpb = blockDim.x >> 6;
  const int lb = xcd_block(blockIdx.x, (ncl + rpb - 1) / rpb);
  if (lb < 0) return;
  const int r_ = lb * rpb + wv;
  if (r_ >= ncl) return;
  const int c = first + r_;  // cluster index: the list row and the member table entry
  const int4 mem = members[c];
  const int twv = tw[c];
  const int done = scal->done, curv = scal->cur;
  if (done) return;
  const int mi[4] = {__builtin_amdgcn_readfirstlane(mem.x), __builtin_amdgcn_readfirstlane(mem.y),
                     __builtin_amdgcn_readfirstlane(mem.z), __builtin_amdgcn_readfirstlane(mem.w)};
  const int M = (mi[1] >= 0) + (mi[2] >= 0) + (mi[3] >= 0) + 1;  // members are packed to the front
  int T = __builtin_amdgcn_readfirstlane(twv & 0xFFFF);
  const int wrapped = __builtin_amdgcn_readfirstlane(twv >> 30) | (ablate & 2);
  const int cur = __builtin_amdgcn_readfirstlane(curv);
  const AtomRec *src = (EP == EP_JACOBI && cur) ? recB : recA;
  AtomRec *dst = (EP == EP_JACOBI) ? (cur ? recA : recB) : recA;
  double xm[4], ym[4], zm[4];
#pragma unroll
  for (int m = 0; m < 4; m++) {
    const double *r = reinterpret_cast<const double *>(src + (mi[m] >= 0 ? mi[m] : mi[0]));
    xm[m] = wave_uniform(r[0]); ym[m] = wave_uniform(r[2]); zm[m] = wave_uniform(r[4]);
  }
  if (ablate & 1) T = 0;
  const char *srcc = reinterpret_cast<const char *>(src);
  char *tile0 = lp_lds + (size_t)wv * (NT * POLAR_LP_TILE);
  const int4 *pc = reinterpret_cast<const int4 *>(dd_j + (size_t)c * pitch) + lane;
  double ax[4] = {0, 0, 0, 0}, ay[4] = {0, 0, 0, 0}, az[4] = {0, 0, 0, 0};
  if (wrapped) cl_rows<true, DAMP, NT>(T, M, pc, srcc, tile0, lane, xm, ym, zm, box, ddcutsq, pd, K, ax, ay, az);
  else cl_rows<false, DAMP, NT>(T, M, pc, srcc, tile0, lane, xm, ym, zm, box, ddcutsq, pd, K, ax, ay, az);
  // ---- in-cluster part: external fields -> uniform values, then the members one after the other ----
  const double px = box.periodic[0] ? box.prd[0] : 0.0, py = box.periodic[1] ? box.prd[1] : 0.0,
               pz = box.periodic[2] ? box.prd[2] : 0.0;
  double Ex[4], Ey[4], Ez[4], mux[4], muy[4], muz[4], al[4];
#pragma unroll
  for (int m = 0; m < 4; m++) {
    Ex[m] = Ey[m] = Ez[m] = mux[m] = muy[m] = muz[m] = al[m] = 0.0;
    if (m < M) {
      const double v = cl_reduce3(ax[m], ay[m], az[m], lane);
      const double *r = reinterpret_cast<const double *>(src + mi[m]);
      const double *e = ef + 3 * (size_t)mi[m];
      Ex[m] = e[0] + lane_value(v, 0); Ey[m] = e[1] + lane_value(v, 1); Ez[m] = e[2] + lane_value(v, 2);
      mux[m] = r[1]; muy[m] = r[3]; muz[m] = r[5]; al[m] = r[7];
    }
  }
  // pair tensors of the cluster (symmetric): p = (a,b), a < b
  double s3p[6], s5p[6], dxp[6], dyp[6], dzp[6];
  {
    const int pa[6] = {0, 0, 1, 0, 1, 2}, pb[6] = {1, 2, 2, 3, 3, 3};
#pragma unroll
    for (int p = 0; p < 6; p++) {
      s3p[p] = s5p[p] = dxp[p] = dyp[p] = dzp[p] = 0.0;
      if (pb[p] < M) {
        double dx = xm[pa[p]] - xm[pb[p]], dy = ym[pa[p]] - ym[pb[p]], dz = zm[pa[p]] - zm[pb[p]];
        dx = fma(-px, rint(dx * box.inv[0]), dx);
        dy = fma(-py, rint(dy * box.inv[1]), dy);
        dz = fma(-pz, rint(dz * box.inv[2]), dz);
        const double r2 = fma(dx, dx, fma(dy, dy, dz * dz));
        // the dipole-dipole cutoff applies inside a cluster too (members are ~2 A apart: always inside);
        // coincident atoms follow the reference's r = 0 rule (PS.cpp:1285-1286: no coupling through DBL_MAX r3/r5 is
        // NOT reproduced: such pairs are dropped here)
        if (r2 < ddcutsq && r2 > 0.0) tensor_scalars_lp<DAMP>(r2, pd, K, s3p[p], s5p[p]);
        dxp[p] = dx; dyp[p] = dy; dzp[p] = dz;
      }
    }
  }
  double nx[4], ny[4], nz[4];
  double chg = 0.0;
#pragma unroll
  for (int m = 0; m < 4; m++) {
    nx[m] = mux[m]; ny[m] = muy[m]; nz[m] = muz[m];
  }
#pragma unroll
  for (int m = 0; m < 4; m++) {
    if (m < M) {
      double ex = Ex[m], ey = Ey[m], ez = Ez[m];
#pragma unroll
      for (int o = 0; o < 4; o++) {
        if (o != m && o < M) {
          const int a = o < m ? o : m, b = o < m ? m : o;
          const int p = a == 0 ? (b == 1 ? 0 : (b == 2 ? 1 : 3)) : (a == 1 ? (b == 2 ? 2 : 4) : 5);
          // Gauss-Seidel: members earlier in the cluster order already carry their new dipoles; Jacobi: all old
          const double ox = (EP == EP_JACOBI || o > m) ? mux[o] : nx[o];
          const double oy = (EP == EP_JACOBI || o > m) ? muy[o] : ny[o];
          const double oz = (EP == EP_JACOBI || o > m) ? muz[o] : nz[o];
          const double dot = ox * dxp[p] + oy * dyp[p] + oz * dzp[p];
          const double cc = s5p[p] * dot;
          ex = fma(cc, dxp[p], fma(-s3p[p], ox, ex));
          ey = fma(cc, dyp[p], fma(-s3p[p], oy, ey));
          ez = fma(cc, dzp[p], fma(-s3p[p], oz, ez));
        }
      }
      nx[m] = al[m] * ex; ny[m] = al[m] * ey; nz[m] = al[m] * ez;
      const double ddx = nx[m] - mux[m], ddy = ny[m] - muy[m], ddz = nz[m] - muz[m];
      chg += ddx * ddx + ddy * ddy + ddz * ddz;
      if (lane == 0) { dst[mi[m]].mx = nx[m]; dst[mi[m]].my = ny[m]; dst[mi[m]].mz = nz[m]; }
    }
  }
  if (lane == 0 && chg != 0.0) atomicAdd(slot_ptr(slots, SL_CHANGE), chg);
}

// ------------------------------------------------------------------------------------------
// The same sweep with the gathers kept D trips ahead (software pipeline inside the wave).  With compiler-issued
// LDS-DMA every read of a tile waits vmcnt(0), so a wave alternates "wait for its gathers" and "compute", and the
// waves of a CU fall into step: the address unit and the vector ALU were each ~55-60 % busy, one after the other
// (profiles/r02_lp3_*).  Here the DMA instructions are inline assembly (M0 = LDS address of the block, set in the
// same statement) and the waits are counted by hand: before trip t is read, the 4 (D-1) gather instructions of the
// trips t+1 .. t+D-1 may still be in flight (trip t+D is requested right after the read).  Four tiles per wave (16 KB): trip t lives in tile t & 3.
// Compiler-issued loads in the loop (the index chunk of rows longer than 8 trips) only make the hand counts
// conservative: vmcnt retires in order, and a count that ignores younger operations waits for more, never for less.
__device__ __forceinline__ void lpa_dma(const char *srcc, unsigned voff, unsigned lds_addr) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(voff), "s"(lds_addr), "s"(srcc)
               : "memory");
}
template <int R>
__device__ __forceinline__ void lpa_gather(const char *srcc, int joff, unsigned piece, unsigned tile_addr) {
  const unsigned o_ = (unsigned)__builtin_amdgcn_update_dpp(0, joff, R * 0x55, 0xF, 0xF, true) + piece;
  lpa_dma(srcc, o_, tile_addr + R * 1024);
}
__device__ __forceinline__ void lpa_gather4(const char *srcc, int joff, unsigned g0, unsigned g1, unsigned g2, unsigned g3,
                                            unsigned tile_addr) {
  lpa_gather<0>(srcc, joff, g0, tile_addr); lpa_gather<1>(srcc, joff, g1, tile_addr);
  lpa_gather<2>(srcc, joff, g2, tile_addr); lpa_gather<3>(srcc, joff, g3, tile_addr);
}
template <int N>
__device__ __forceinline__ void lpa_wait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <bool WRAP, int DAMP, int D>
__device__ __forceinline__ void lpa_row(int T, const int4 *pc, const char *srcc, char *tile0, int lane, const AtomRec &ri,
                                        const Box &box, double pd, const ExpCoef &K, double &ax, double &ay, double &az) {
  const double px = box.periodic[0] ? box.prd[0] : 0.0, py = box.periodic[1] ? box.prd[1] : 0.0,
               pz = box.periodic[2] ? box.prd[2] : 0.0;
  const int k = lane & 3, q = lane >> 2;
  const unsigned g0 = (unsigned)(k * 16), g1 = (unsigned)((k ^ 1) * 16), g2 = (unsigned)((k ^ 2) * 16), g3 = (unsigned)((k ^ 3) * 16);
  const char *rd0 = tile0 + k * 1024 + (4 * q + k) * 16;
  const char *rd1 = tile0 + k * 1024 + (4 * q + (k ^ 1)) * 16;
  const char *rd2 = tile0 + k * 1024 + (4 * q + (k ^ 2)) * 16;
  if (T <= 0) return;
  const unsigned ta = (unsigned)(size_t)(__attribute__((address_space(3))) char *)tile0;  // LDS byte address of tile 0
  // trip TT of a chunk: tile TT; JAHEAD = index of trip t+D; the wait leaves the gathers of the trips ahead in flight
#define POLAR_LPA_TRIP(TT, JAHEAD)                                                                       \
  {                                                                                                      \
    const int t = t0 + (TT);                                                                             \
    const int ahead = Tn - 1 - t;  /* trips after this one (wave-uniform) */                             \
    /* younger than the gathers of trip t: those of the trips t+1 .. t+D-1 (trip t+D is requested below) */ \
    if (D > 2 && ahead >= 2) lpa_wait<8>();                                                              \
    else if (ahead >= 1) lpa_wait<4>();                                                                  \
    else lpa_wait<0>();                                                                                  \
    const double2 A = *reinterpret_cast<const double2 *>(rd0 + (TT) * POLAR_LP_TILE);                    \
    const double2 B = *reinterpret_cast<const double2 *>(rd1 + (TT) * POLAR_LP_TILE);                    \
    const double2 Cc = *reinterpret_cast<const double2 *>(rd2 + (TT) * POLAR_LP_TILE);                   \
    if (t + D < Tn) lpa_gather4(srcc, JAHEAD, g0, g1, g2, g3, ta + (((TT) + D) & 3) * POLAR_LP_TILE);    \
    double dx = ri.x - A.x, dy = ri.y - B.x, dz = ri.z - Cc.x;                                           \
    if (WRAP) {                                                                                          \
      dx = fma(-px, rint(dx * box.inv[0]), dx);                                                          \
      dy = fma(-py, rint(dy * box.inv[1]), dy);                                                          \
      dz = fma(-pz, rint(dz * box.inv[2]), dz);                                                          \
    }                                                                                                    \
    const double r2 = fmax(fma(dx, dx, fma(dy, dy, dz * dz)), 1e-12);                                    \
    double s3, s5;                                                                                       \
    tensor_scalars_lp<DAMP>(r2, pd, K, s3, s5);                                                          \
    const double dot = fma(A.y, dx, fma(B.y, dy, Cc.y * dz));                                            \
    const double cc = s5 * dot;                                                                          \
    ax = fma(cc, dx, fma(-s3, A.y, ax));                                                                 \
    ay = fma(cc, dy, fma(-s3, B.y, ay));                                                                 \
    az = fma(cc, dz, fma(-s3, Cc.y, az));                                                                \
  }
  // A row is walked in stretches of up to 12 trips (three index chunks, all requested before the stretch starts, so
  // the loop below holds no compiler-issued load and nothing but the hand-counted waits); rows longer than 768 pairs
  // restart the pipeline once per stretch.
  for (int base = 0; base < T; base += 12) {
    const int Tn = (T - base) < 12 ? (T - base) : 12;
    const int4 *p = pc + 64 * (base >> 2);
    int4 Ja = p[0], Jb = make_int4(0, 0, 0, 0), Jc = Jb;
    if (Tn > 4) Jb = p[64];
    if (Tn > 8) Jc = p[128];
    lpa_gather4(srcc, Ja.x, g0, g1, g2, g3, ta);  // the gathers of trips 0 .. D-1
    if (Tn > 1) lpa_gather4(srcc, Ja.y, g0, g1, g2, g3, ta + POLAR_LP_TILE);
    if (D > 2 && Tn > 2) lpa_gather4(srcc, Ja.z, g0, g1, g2, g3, ta + 2 * POLAR_LP_TILE);
#pragma unroll
    for (int c = 0; c < 3; c++) {
      const int t0 = 4 * c;
      if (t0 >= Tn) break;
      if (D == 2) {
        POLAR_LPA_TRIP(0, Ja.z);
        if (t0 + 1 >= Tn) break;
        POLAR_LPA_TRIP(1, Ja.w);
        if (t0 + 2 >= Tn) break;
        POLAR_LPA_TRIP(2, Jb.x);
        if (t0 + 3 >= Tn) break;
        POLAR_LPA_TRIP(3, Jb.y);
      } else {
        POLAR_LPA_TRIP(0, Ja.w);
        if (t0 + 1 >= Tn) break;
        POLAR_LPA_TRIP(1, Jb.x);
        if (t0 + 2 >= Tn) break;
        POLAR_LPA_TRIP(2, Jb.y);
        if (t0 + 3 >= Tn) break;
        POLAR_LPA_TRIP(3, Jb.z);
      }
      Ja = Jb; Jb = Jc;
    }
  }
#undef POLAR_LPA_TRIP
}
template <int EP, int DAMP, int D>
static __global__ __launch_bounds__(256) void k_field_lpa(int nrows, long long row0, const int2 *__restrict__ desc, AtomRec *recA, AtomRec *recB,
                                                   Box box, long long pitch, const int *__restrict__ dd_j, double pd,
                                                   ExpCoef K, const double *__restrict__ ef, const Scal *scal,
                                                   double *__restrict__ slots, int ablate) {
  extern __shared__ __attribute__((aligned(16))) char lp_lds[];
  if (scal->done) return;
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int rpb = blockDim.x >> 6;
  const int lb = xcd_block(blockIdx.x, (nrows + rpb - 1) / rpb);
  if (lb < 0) return;
  const int row = lb * rpb + wv;
  if (row >= nrows) return;
  const int2 de = desc[row];
  const int i = __builtin_amdgcn_readfirstlane(de.x);
  int T = __builtin_amdgcn_readfirstlane(de.y & 0xFFFF);
  const int wrapped = __builtin_amdgcn_readfirstlane(de.y >> 30) | (ablate & 2);
  const int cur = __builtin_amdgcn_readfirstlane(scal->cur);
  const AtomRec *src = (EP == EP_JACOBI && cur) ? recB : recA;
  AtomRec *dst = (EP == EP_JACOBI) ? (cur ? recA : recB) : recA;
  const AtomRec ri = uniform_rec(src[i]);
  if (ablate & 1) T = 0;
  const double efx = ef[3 * i], efy = ef[3 * i + 1], efz = ef[3 * i + 2];
  const char *srcc = reinterpret_cast<const char *>(src);
  char *tile0 = lp_lds + (size_t)wv * (4 * POLAR_LP_TILE);
  const int4 *pc = reinterpret_cast<const int4 *>(dd_j + (size_t)(row0 + row) * pitch) + lane;  // dd rows are in launch order
  double ax = 0.0, ay = 0.0, az = 0.0;
  if (wrapped) lpa_row<true, DAMP, D>(T, pc, srcc, tile0, lane, ri, box, pd, K, ax, ay, az);
  else lpa_row<false, DAMP, D>(T, pc, srcc, tile0, lane, ri, box, pd, K, ax, ay, az);
  ax = wave_sum(ax); ay = wave_sum(ay); az = wave_sum(az);
  if (lane == 0) {
    const double mx = ri.a * (efx + ax), my = ri.a * (efy + ay), mz = ri.a * (efz + az);
    const double ddx = mx - ri.mx, ddy = my - ri.my, ddz = mz - ri.mz;
    dst[i].mx = mx; dst[i].my = my; dst[i].mz = mz;
    const double chg = ddx * ddx + ddy * ddy + ddz * ddz;
    if (chg != 0.0) atomicAdd(slot_ptr(slots, SL_CHANGE), chg);
  }
}

// a6 for the list path: the damped tensor scalars of every listed pair, once per step
// (the sparse, matrix-free-storage analog of build_dipole_field_matrix, PS.cpp:1273-1306).
template <int DAMP>
static __global__ __launch_bounds__(POLAR_BLOCK) void k_dd_scalars(const int *__restrict__ rows, int nrows, const AtomRec *__restrict__ rec,
                                                            Box box,
                                                            RowList ddl,
                                                            int *__restrict__ dd_j, double pd,
                                                            double2 *__restrict__ dd_s, double *__restrict__ dd_r2) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * POLAR_ROWS_PER_BLOCK + (threadIdx.x >> 6);
  if (row >= nrows) return;
  const int i = rows ? rows[row] : row;
  const double xi = rec[i].x, yi = rec[i].y, zi = rec[i].z;
  long long beg, end;
  row_range(ddl, i, beg, end);
  if (dd_r2 || dd_s) {
    for (long long p = beg + lane; p < end; p += 64) {
      const int j = dd_j[p];
      double dx, dy, dz;
      min_image_rint(box, xi, yi, zi, rec[j].x, rec[j].y, rec[j].z, dx, dy, dz);
      double s3, s5;
      const double r2 = dx * dx + dy * dy + dz * dz;
      if (dd_r2) dd_r2[p] = r2;
      else {
        tensor_scalars<DAMP>(r2, pd, s3, s5);
        dd_s[p] = make_double2(s3, s5);
      }
    }
  }
  // pad the row to whole 64-pair trips with inert entries (the atom itself, zero tensor): the
  // component-per-lane sweep then runs without lane masks.  The pitch is a multiple of 64.
  const long long pad_end = beg + (((end - beg) + 63) & ~63ll);
  for (long long p = end + lane; p < pad_end; p += 64) {
    dd_j[p] = i;
    if (dd_r2) dd_r2[p] = 1e60;  // s3 ~ 1e-90, and d = 0 kills the s5 term: contributes nothing
    else if (dd_s) dd_s[p] = make_double2(0.0, 0.0);
  }
}

#endif  // POLAR_LAB

// The end-of-sweep logic (PS.cpp:1193-1236) by ONE thread, given the sweep's sum |dmu|^2: shared by k_solver_step and by the tail
// of exact mode's last launch of a sweep (k_gs_blk).  `count` > 1: the logic of several sweeps at once (fixed-iteration
// Gauss-Seidel takes no decision between sweeps, so the host launches k_solver_step only before and after the last one).
__device__ __forceinline__ void solver_decide(Scal *scal, double sum, const double *global_change, int nlocal, int fixed_iteration,
                                              int iterations_max, double precision, int jacobi, int count) {
  scal->change = sum;  // this handle's own sum (exported to the all-reduce in multi-GPU runs)
  // multi-GPU: the all-reduced sum over ranks arrives through global_change (device memory)
  const double change = (global_change ? *global_change : sum) / ((double)nlocal * 3.0);
  scal->last_change = change;
  for (int c = 0; c < count; c++) {
    scal->sweeps += 1;
    int keep = 1;
    if (!fixed_iteration) keep = change > precision * precision;
    else if (scal->iterations >= iterations_max) { scal->done = 1; return; }  // returns BEFORE the copy
    if (jacobi) scal->cur ^= 1;  // "mu = mu_new"
    scal->iterations += 1;
    if (scal->iterations > iterations_max) { scal->status = 1; scal->done = 1; return; }
    if (!keep) { scal->done = 1; return; }
  }
}

// ------------------------------------------------------------------------------------------
// a7 loop control, one thread: the reference's end-of-sweep logic (PS.cpp:1193-1236) kept on the
// device so the host never has to look at ||dmu||^2 between sweeps.
static __global__ __launch_bounds__(POLAR_NSLOT) void k_solver_step(Scal *scal, double *__restrict__ slots, int nlocal,
                                                           int fixed_iteration, int iterations_max, double precision,
                                                           int jacobi, const double *__restrict__ global_change,
                                                           int count, const double *__restrict__ part, int npart) {
  if (scal->done) return;
  __shared__ double red[POLAR_NSLOT / 64];
  double v = slots[(size_t)threadIdx.x * POLAR_SLOT_STRIDE + SL_CHANGE];
  slots[(size_t)threadIdx.x * POLAR_SLOT_STRIDE + SL_CHANGE] = 0.0;
  // `deterministic yes`: the sweep's changes sit in `part` (k_lp_commit), one value per 256 rows in launch order; thread t adds
  // slots t, t + 1024, ...: a fixed association (the atomic slots above stay zero in that mode)
  for (int k = threadIdx.x; k < npart; k += POLAR_NSLOT) v += part[k];
  v = wave_sum(v);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x != 0) return;
  double sum = 0.0;
  for (int k = 0; k < POLAR_NSLOT / 64; k++) sum += red[k];
  solver_decide(scal, sum, global_change, nlocal, fixed_iteration, iterations_max, precision, jacobi, count);
}

// fold the change slots into scal->change without touching the loop state (multi-GPU export)
static __global__ __launch_bounds__(POLAR_NSLOT) void k_fold_change(Scal *scal, double *__restrict__ slots, double *dst,
                                                           const double *__restrict__ part, int npart) {
  __shared__ double red[POLAR_NSLOT / 64];
  double v = slots[(size_t)threadIdx.x * POLAR_SLOT_STRIDE + SL_CHANGE];
  slots[(size_t)threadIdx.x * POLAR_SLOT_STRIDE + SL_CHANGE] = 0.0;
  for (int k = threadIdx.x; k < npart; k += POLAR_NSLOT) v += part[k];
  v = wave_sum(v);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x != 0) return;
  double sum = 0.0;
  for (int k = 0; k < POLAR_NSLOT / 64; k++) sum += red[k];
  scal->change = sum;
  *dst = sum;
}

// fold energies / virial / rmin slots into the Scal block (run before the host reads it)
static __global__ __launch_bounds__(POLAR_NSLOT) void k_fold_scal(Scal *scal, double *__restrict__ slots, int rmin_only) {
  // one wave per accumulator field (wave 0: rmin; waves 1..11: energies and virial), all at once
  const int t = threadIdx.x, w = t >> 6, lane = t & 63;
  if (w == 0) {
    double r = 1.0e300;
    for (int k = lane; k < POLAR_NSLOT; k += 64) {
      const unsigned long long b = ((unsigned long long *)slots)[(size_t)k * POLAR_SLOT_STRIDE + SL_RMIN];
      r = fmin(r, __longlong_as_double((long long)b));
    }
    r = wave_min(r);
    if (lane == 0) scal->rmin_bits = (unsigned long long)__double_as_longlong(r);
    return;
  }
  if (rmin_only) return;
  const int f = SL_EVDWL + (w - 1);
  if (f > SL_V5) return;
  double v = 0.0;
  for (int k = lane; k < POLAR_NSLOT; k += 64) v += slots[(size_t)k * POLAR_SLOT_STRIDE + f];
  v = wave_sum(v);
  if (lane == 0) {
    double *dst = f == SL_EVDWL ? &scal->eng_vdwl : f == SL_ECOUL ? &scal->eng_coul : f == SL_USELF ? &scal->u_self
                : f == SL_UEF ? &scal->u_ef : f == SL_UDD ? &scal->u_dd : &scal->virial[f - SL_V0];
    *dst = v;
  }
}
// start-of-step reset of the slot accumulators and (thread 0) of the Scal block: one launch
static __global__ void k_zero_slots(double *__restrict__ slots, Scal *s) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t == 0 && s) {
    s->eng_vdwl = s->eng_coul = s->u_self = s->u_ef = s->u_dd = 0.0;
    for (int k = 0; k < 6; k++) s->virial[k] = 0.0;
    s->change = 0.0; s->last_change = 0.0; s->pad = 0; s->det_change = 0.0;
    s->rmin_bits = (unsigned long long)__double_as_longlong(1000.0);
    s->iterations = 0; s->done = 0; s->status = 0; s->cur = 0; s->sweeps = 0;
  }
  if (t >= POLAR_NSLOT) return;
  for (int f = 0; f < POLAR_SLOT_STRIDE; f++) slots[(size_t)t * POLAR_SLOT_STRIDE + f] = 0.0;
  ((unsigned long long *)slots)[(size_t)t * POLAR_SLOT_STRIDE + SL_RMIN] = (unsigned long long)__double_as_longlong(1000.0);
}

// `debug yes`: u_polar = -1/2 sum_i E_static,i . mu_i after a sweep (PS.cpp:1182-1191 prints it per iteration).
// The dipoles are read where the sweep keeps them: component k of record i = double 2k + 1 at base + i * stride.
static __global__ __launch_bounds__(1024) void k_debug_upolar(int n, const Scal *scal, const char *recA, const char *recB, int stride,
                                                       const double *__restrict__ ef, double *__restrict__ trace, int slot,
                                                       int jacobi_next) {
  if (scal->done) return;  // a launch past the end of a finished solve (the host looks at the state every 4 sweeps)
  __shared__ double red[16];
  // Jacobi: the sweep that just ran wrote the OTHER buffer (the copy "mu = mu_new" happens in k_solver_step)
  const char *base = (scal->cur ^ jacobi_next) ? recB : recA;
  double v = 0.0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const double *r = reinterpret_cast<const double *>(base + (size_t)i * stride);
    v += ef[3 * i] * r[1] + ef[3 * i + 1] * r[3] + ef[3 * i + 2] * r[5];
  }
  v = wave_sum(v);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    double sum = 0.0;
    for (int k = 0; k < 16; k++) sum += red[k];
    trace[slot] = -0.5 * sum;
  }
}

// divergence fallback mu = alpha * E (no gamma), PS.cpp:1227-1235
static __global__ void k_fallback(int n, const Scal *scal, AtomRec *__restrict__ recA, AtomRec *__restrict__ recB,
                           const double *__restrict__ ef) {
  if (!scal->status) return;
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  AtomRec *r = scal->cur ? recB : recA;
  const double a = r[i].a;
  r[i].mx = a * ef[3 * i]; r[i].my = a * ef[3 * i + 1]; r[i].mz = a * ef[3 * i + 2];
}

// copy the final dipoles and the static field out (records are in s order, outputs in orig order)
static __global__ void k_unpack(int n, const int *__restrict__ perm, const Scal *scal, const AtomRec *__restrict__ recA,
                         const AtomRec *__restrict__ recB, const double *__restrict__ ef_s, double *__restrict__ mu,
                         double *__restrict__ ef) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const AtomRec *r = scal->cur ? recB : recA;
  const int o = perm ? perm[i] : i;
  mu[3 * o] = r[i].mx; mu[3 * o + 1] = r[i].my; mu[3 * o + 2] = r[i].mz;
  ef[3 * o] = ef_s[3 * i]; ef[3 * o + 1] = ef_s[3 * i + 1]; ef[3 * o + 2] = ef_s[3 * i + 2];
}

}  // namespace polar
