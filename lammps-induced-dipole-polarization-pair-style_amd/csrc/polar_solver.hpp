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
#include "lab/sweep_register_staged_body.hpp"
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
#include "lab/sweep_quad.hpp"
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
#include "lab/sweep_lpr_lpa_lp2_cluster.hpp"
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
