// polar_common.hpp -- types, slot accumulators, wave/DPP helpers, minimum-image rules, tensor scalars, record fetch.
// Part of the hand-written HIP kernels (gfx950 / CDNA4, wave64) of the lj/cut/coul/long/polarization
// hot path; see polar_kernels.hpp for the mapping and the index spaces.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

// LAB BUILD (-DPOLAR_LAB, libpolar_mi355x_lab.so): the sweep kernels that were built, measured and did not become the default
// (component-per-lane, register-staged, cluster rows, tile sweep, variants of the row sweep), the `ablate` timing switches
// and the environment knobs that select them.  The product library is compiled without it: its kernels carry no lab
// argument and it reads no lab environment variable (tests/test_gpu_parity.py::test_product_library_ignores_lab_switches).
#ifdef POLAR_LAB
#define POLAR_LAB_PARAM , int ablate
#define POLAR_LAB_PASS , ablate
#define POLAR_ABL(bit) (ablate & (bit))
#else
#define POLAR_LAB_PARAM
#define POLAR_LAB_PASS
#define POLAR_ABL(bit) 0
#endif

namespace polar {


// 64-byte atom record: one L2 line-half per gathered neighbor.
struct __attribute__((aligned(64))) AtomRec {
  // component-interleaved: 16-byte piece k (k = 0,1,2) holds (position_k, dipole_k); piece 3 = (q, alpha).
  // A QUAD of lanes reads one whole record with one coalesced 64-byte access (see k_field, list mode).
  double x, mx;
  double y, my;
  double z, mz;
  double q, a;
};

struct Box {
  double prd[3], half[3], inv[3];
  double xy, xz, yz;  // triclinic tilt factors (domain.cpp:1258-1305); zero for orthogonal boxes
  int periodic[3];
  int triclinic;
};

// Device-resident solver/accumulator block (one per handle).
struct Scal {
  double eng_vdwl, eng_coul, u_self, u_ef, u_dd;
  double virial[6];
  double change;           // sum (mu_new - mu_old)^2 of the running sweep
  double last_change;      // change / (3N) of the last finished sweep
  unsigned long long rmin_bits;  // double bits of rmin (positive doubles order like uint64)
  int iterations, done, status, cur, sweeps, pad;
  double det_change;       // `deterministic yes`: sum (dmu)^2 of the running sweep, added launch by launch in a fixed order
};

// Contended accumulators (energies, virial, sum dmu^2, rmin) are spread over NSLOT cache lines:
// every wave adds into the line picked by its workgroup id, a single-workgroup kernel folds the
// lines.  (One shared address costs ~12 ns per atomic on MI355X: 36k rows -> 0.4 ms per launch.)
#define POLAR_NSLOT 1024
#define POLAR_SLOT_STRIDE 16
enum { SL_EVDWL = 0, SL_ECOUL, SL_USELF, SL_UEF, SL_UDD, SL_V0, SL_V1, SL_V2, SL_V3, SL_V4, SL_V5, SL_CHANGE, SL_RMIN };
__device__ __forceinline__ double *slot_ptr(double *slots, int field) {
  return slots + (size_t)(blockIdx.x & (POLAR_NSLOT - 1)) * POLAR_SLOT_STRIDE + field;
}

#define POLAR_WAVE 64
#define POLAR_BLOCK 256
#define POLAR_ROWS_PER_BLOCK (POLAR_BLOCK / POLAR_WAVE)

// Wave-wide reductions through DPP (data-parallel primitives: no LDS crossbar round trips).
// quad_perm xor1, xor2 -> row_half_mirror -> row_mirror give every lane its 16-lane row total;
// row_bcast15 / row_bcast31 (GFX9/CDNA) carry row totals into the following rows, so lane 63 ends
// with the wave total, which readlane broadcasts.  ~18 short VALU ops per double instead of 12
// dependent ds_bpermute round trips.
template <int CTRL, int ROWMASK>
__device__ __forceinline__ double dpp_get(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROWMASK, 0xF, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROWMASK, 0xF, false);
  return __hiloint2double(hi, lo);
}
// XCD-aware row placement.  Consecutive workgroup ids are dealt round-robin to the 8 XCDs, each with
// its own 4 MB L2.  Rows are in cell order (spatially sorted), so giving XCD x the x-th contiguous
// eighth of the launch's workgroups keeps each L2's gather working set to one slab of the box plus
// its cutoff halo instead of the whole record table (8.6 MB at 135k atoms).  The grid is
// 8 * ceil(nblocks / 8) workgroups; returns -1 for the padding workgroups.
__device__ __forceinline__ int xcd_block(int b, int nblocks) {
  const int chunk = (nblocks + 7) >> 3;
  const int lb = (b & 7) * chunk + (b >> 3);
  return lb < nblocks ? lb : -1;
}

// full-mask permutations (every lane has a source): bound_ctrl lets the compiler skip the
// zero-initialisation of the destination that dpp_get needs for its masked rows
template <int CTRL>
__device__ __forceinline__ double dpp_full(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
// a value every lane of the wave holds identically: park it in scalar registers (the VALU can read
// one SGPR pair per instruction), which keeps wave-uniform row data out of the vector register file
__device__ __forceinline__ double wave_uniform(double v) {
  const int lo = __builtin_amdgcn_readfirstlane(__double2loint(v));
  const int hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ AtomRec uniform_rec(const AtomRec &r) {
  AtomRec u;
  u.x = wave_uniform(r.x); u.mx = wave_uniform(r.mx); u.y = wave_uniform(r.y); u.my = wave_uniform(r.my);
  u.z = wave_uniform(r.z); u.mz = wave_uniform(r.mz); u.q = wave_uniform(r.q); u.a = wave_uniform(r.a);
  return u;
}
__device__ __forceinline__ double lane63(double v) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum(double v) {
  v += dpp_full<0xB1>(v);   // quad_perm [1,0,3,2]
  v += dpp_full<0x4E>(v);   // quad_perm [2,3,0,1]
  v += dpp_full<0x141>(v);  // row_half_mirror
  v += dpp_full<0x140>(v);  // row_mirror: all 16 lanes of a row hold the row total
  v += dpp_get<0x142, 0xA>(v);  // row_bcast15 into rows 1 and 3
  v += dpp_get<0x143, 0xC>(v);  // row_bcast31 into rows 2 and 3
  return lane63(v);
}
__device__ __forceinline__ double wave_min(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmin(v, __shfl_xor(v, off, 64));
  return v;
}

// Domain::closest_image wrap of one component d = xj - xi (domain.cpp:1231-1257): the reference's
// add/subtract sequence, not a rint() wrap, so that pairs at exactly L/2 pick the same image.
__device__ __forceinline__ double wrap_ci(double d, double L, double h, int periodic) {
  if (periodic) {
    if (d < 0.0) {
      while (d < 0.0) d += L;
      if (d > h) d -= L;
    } else {
      while (d > 0.0) d -= L;
      if (d < -h) d += L;
    }
  }
  return d;
}

// del = x_i - closest_image(x_j)
__device__ __forceinline__ void min_image_del(const Box &b, double xi, double yi, double zi, double xj, double yj,
                                              double zj, double &dx, double &dy, double &dz) {
  if (!b.triclinic) {
    dx = -wrap_ci(xj - xi, b.prd[0], b.half[0], b.periodic[0]);
    dy = -wrap_ci(yj - yi, b.prd[1], b.half[1], b.periodic[1]);
    dz = -wrap_ci(zj - zi, b.prd[2], b.half[2], b.periodic[2]);
    return;
  }
  // triclinic branch of Domain::closest_image (domain.cpp:1258-1305): z first (carrying yz, xz into
  // y and x), then y (carrying xy into x), then x -- same add/subtract sequence as the reference
  double ex = xj - xi, ey = yj - yi, ez = zj - zi;
  if (b.periodic[2]) {
    if (ez < 0.0) {
      while (ez < 0.0) { ez += b.prd[2]; ey += b.yz; ex += b.xz; }
      if (ez > b.half[2]) { ez -= b.prd[2]; ey -= b.yz; ex -= b.xz; }
    } else {
      while (ez > 0.0) { ez -= b.prd[2]; ey -= b.yz; ex -= b.xz; }
      if (ez < -b.half[2]) { ez += b.prd[2]; ey += b.yz; ex += b.xz; }
    }
  }
  if (b.periodic[1]) {
    if (ey < 0.0) {
      while (ey < 0.0) { ey += b.prd[1]; ex += b.xy; }
      if (ey > b.half[1]) { ey -= b.prd[1]; ex -= b.xy; }
    } else {
      while (ey > 0.0) { ey -= b.prd[1]; ex -= b.xy; }
      if (ey < -b.half[1]) { ey += b.prd[1]; ex += b.xy; }
    }
  }
  ex = wrap_ci(ex, b.prd[0], b.half[0], b.periodic[0]);
  dx = -ex; dy = -ey; dz = -ez;
}

// quad (4-lane) exchange through DPP quad_perm -- no LDS crossbar
__device__ __forceinline__ double quad_xor(double v, const int which) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  if (which == 1) {  // lanes 0<->1, 2<->3   quad_perm [1,0,3,2]
    lo = __builtin_amdgcn_update_dpp(0, lo, 0xB1, 0xF, 0xF, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, 0xB1, 0xF, 0xF, true);
  } else {           // lanes 0<->2, 1<->3   quad_perm [2,3,0,1]
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x4E, 0xF, 0xF, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x4E, 0xF, 0xF, true);
  }
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double quad_sum(double v) {
  v += quad_xor(v, 1);
  v += quad_xor(v, 2);
  return v;
}

// Pitched row lists (list mode): row i owns the slots [i*pitch, i*pitch + cnt[i]).  A fixed pitch lets
// k_nl_build emit both lists in ONE pass (no count pass, no prefix scans, no host sync for the
// totals); a row that would overflow sets a flag and the step is redone with a larger pitch.
struct RowList {
  const int *cnt;
  long long pitch;
};
__device__ __forceinline__ void row_range(const RowList &L, int i, long long &beg, long long &end) {
  beg = (long long)i * L.pitch;
  const long long c = L.cnt[i];
  end = beg + (c < L.pitch ? c : L.pitch);
}

// Branch-free minimum image for the LIST kernels (dd_cutoff extension): d - L*rint(d/L).
// Equals closest_image except for pairs at exactly L/2, which lie outside every cutoff there
// (the list path requires L >= 2*cutoff).  The all-pairs (reference-exact) kernels keep wrap_ci.
__device__ __forceinline__ void min_image_rint(const Box &b, double xi, double yi, double zi, double xj, double yj,
                                               double zj, double &dx, double &dy, double &dz) {
  // One straight-line form for orthogonal and tilted boxes: whole lattice vectors c = (xz, yz, zprd), b = (xy, yprd, 0),
  // a = (xprd, 0, 0) are taken off in that order -- the order of Domain::closest_image's triclinic branch
  // (domain.cpp:1258-1305), equal to it for every pair inside a cutoff <= half the perpendicular box widths (what the
  // list mode requires).  With zero tilt factors the extra FMAs add exact zeros: bit-identical to the per-component
  // d - L*rint(d/L).  (A branch on b.triclinic with an early return cost 24 bytes of scratch per lane in four kernels.)
  double ex = xi - xj, ey = yi - yj, ez = zi - zj;
  const double nz = b.periodic[2] ? rint(ez * b.inv[2]) : 0.0;
  ez = fma(-b.prd[2], nz, ez); ey = fma(-b.yz, nz, ey); ex = fma(-b.xz, nz, ex);
  const double ny = b.periodic[1] ? rint(ey * b.inv[1]) : 0.0;
  ey = fma(-b.prd[1], ny, ey); ex = fma(-b.xy, ny, ex);
  const double nx = b.periodic[0] ? rint(ex * b.inv[0]) : 0.0;
  ex = fma(-b.prd[0], nx, ex);
  dx = ex; dy = ey; dz = ez;
}
// The same for the list build, with the box shape known at compile time (orthogonal boxes drop the three tilt FMAs) and
// "a lattice vector was taken off" reported from the rounded multiples themselves (the build marks rows whose pairs all
// lie inside the box: their sweeps skip the minimum image).
template <bool TRI>
__device__ __forceinline__ bool min_image_rint_w(const Box &b, double xi, double yi, double zi, double xj, double yj,
                                                 double zj, double &dx, double &dy, double &dz) {
  double ex = xi - xj, ey = yi - yj, ez = zi - zj;
  const double nz = b.periodic[2] ? rint(ez * b.inv[2]) : 0.0;
  ez = fma(-b.prd[2], nz, ez);
  if (TRI) { ey = fma(-b.yz, nz, ey); ex = fma(-b.xz, nz, ex); }
  const double ny = b.periodic[1] ? rint(ey * b.inv[1]) : 0.0;
  ey = fma(-b.prd[1], ny, ey);
  if (TRI) ex = fma(-b.xy, ny, ex);
  const double nx = b.periodic[0] ? rint(ex * b.inv[0]) : 0.0;
  ex = fma(-b.prd[0], nx, ex);
  dx = ex; dy = ey; dz = ez;
  return nz != 0.0 || ny != 0.0 || nx != 0.0;
}
// fractional ("lamda") coordinates of a point, src/domain.cpp x2lamda: orthogonal boxes divide by the box lengths,
// tilted boxes back-substitute through the triangular cell matrix
__host__ __device__ __forceinline__ void frac_coords(const Box &b, const double lo[3], double x, double y, double z, double fr[3]) {
  fr[2] = (z - lo[2]) * b.inv[2];
  if (b.triclinic) {
    fr[1] = ((y - lo[1]) - b.yz * fr[2] * 1.0) * b.inv[1];
    fr[0] = ((x - lo[0]) - b.xy * fr[1] - b.xz * fr[2]) * b.inv[0];
  } else {
    fr[1] = (y - lo[1]) * b.inv[1];
    fr[0] = (x - lo[0]) * b.inv[0];
  }
}
template <bool EXACT>
__device__ __forceinline__ void pair_del(const Box &b, double xi, double yi, double zi, double xj, double yj,
                                         double zj, double &dx, double &dy, double &dz) {
  if (EXACT) min_image_del(b, xi, yi, zi, xj, yj, zj, dx, dy, dz);
  else min_image_rint(b, xi, yi, zi, xj, yj, zj, dx, dy, dz);
}

// chunked order of the lp sweep's index stream (see k_field_lp): entry e of a row -> slot.  Chunks of 256 entries = 4 trips;
// inside a chunk the slot is 4 * lane + trip.  `qm` (quad-major, the default): entry p of a trip goes to lane 4 * (p % 16) +
// p / 16, so that gather instruction r of the sweep (quad q fetches the record of lane 4q + r) works on the 16 CONSECUTIVE
// list entries 16r .. 16r + 15 -- consecutive entries are mostly consecutive records of one cell, i.e. the two halves of
// one 128-byte line and neighbouring lines, requested by neighbouring quads of ONE instruction.
__host__ __device__ __forceinline__ long long lp_slot(long long e, int qm = 1) {
  const long long p = e & 63;
  const long long lane = qm ? (((p & 15) << 2) | (p >> 4)) : p;
  return ((e >> 8) << 8) + (lane << 2) + ((e >> 6) & 3);
}

#define POLAR_NL_SAMEMOL 0x40000000
#define POLAR_NL_MASK 0x3FFFFFFF

// Wave-cooperative gather of 64 atom records for the lane-per-pair row kernels (list mode).  A
// scattered load costs the vector-memory address unit one 64-byte line per LANE; here lane k of quad
// q loads piece k of the records of lanes q, 16+q, 32+q, 48+q (4 instructions, one line per quad
// each), a per-wave LDS tile (80-byte pitch: conflict-free b128 reads) transposes them, and every
// lane gets ITS record back.  All 64 lanes call it (idle lanes pass any valid index).
struct RecQuad { double2 a, b, c, d; };  // {x,mx} {y,my} {z,mz} {q,alpha}
__device__ __forceinline__ RecQuad fetch_records(const AtomRec *__restrict__ rec, int j, double2 *stage, int lane) {
  const int q4 = lane >> 2;
  // scalar base + 32-bit lane offset (index * 64 + piece * 16; fewer than 2^26 records)
  const char *base = reinterpret_cast<const char *>(rec);
  const unsigned piece = (unsigned)(lane & 3) * 16u, k = lane & 3;
  const unsigned j0 = __shfl(j, q4, 64), j1 = __shfl(j, 16 + q4, 64), j2 = __shfl(j, 32 + q4, 64), j3 = __shfl(j, 48 + q4, 64);
  const double2 p0 = *reinterpret_cast<const double2 *>(base + ((j0 << 6) + piece));
  const double2 p1 = *reinterpret_cast<const double2 *>(base + ((j1 << 6) + piece));
  const double2 p2 = *reinterpret_cast<const double2 *>(base + ((j2 << 6) + piece));
  const double2 p3 = *reinterpret_cast<const double2 *>(base + ((j3 << 6) + piece));
  stage[q4 * 5 + k] = p0; stage[(16 + q4) * 5 + k] = p1;
  stage[(32 + q4) * 5 + k] = p2; stage[(48 + q4) * 5 + k] = p3;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  RecQuad r;
  r.a = stage[lane * 5]; r.b = stage[lane * 5 + 1]; r.c = stage[lane * 5 + 2]; r.d = stage[lane * 5 + 3];
  __builtin_amdgcn_wave_barrier();  // the tile is rewritten by the next trip
  return r;
}

// The same for kernels that need only position and charge of the neighbour (static field): 32-byte records {x, y, z, q}
// (s order), two lanes per record -- lane L of instruction r loads piece L & 1 of the record of lane 32 r + L / 2: half the
// vector-memory instructions, half the bytes.  LDS tile with a 48-byte pitch (stride 12 banks: conflict-free b128 reads).
struct XQ { double2 a, b; };  // {x, y} {z, q}
__device__ __forceinline__ XQ fetch_xq(const double4 *__restrict__ xq, int j, double2 *stage, int lane) {
  const int h2 = lane >> 1;
  const char *base = reinterpret_cast<const char *>(xq);
  const unsigned piece = (unsigned)(lane & 1) * 16u, k = lane & 1;
  const unsigned j0 = __shfl(j, h2, 64), j1 = __shfl(j, 32 + h2, 64);
  const double2 p0 = *reinterpret_cast<const double2 *>(base + ((j0 << 5) + piece));
  const double2 p1 = *reinterpret_cast<const double2 *>(base + ((j1 << 5) + piece));
  stage[h2 * 3 + k] = p0; stage[(32 + h2) * 3 + k] = p1;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  XQ r;
  r.a = stage[lane * 3]; r.b = stage[lane * 3 + 1];
  __builtin_amdgcn_wave_barrier();  // the tile is rewritten by the next trip
  return r;
}

// exp(x) for x <= 0 with its constants in SCALAR registers.  The library routine materialises a dozen
// FP64 constants with v_mov pairs on every call (25 of the 173 vector instructions of a sweep trip);
// here they arrive as a kernel argument (kernel arguments live in SGPRs, and an FP64 VALU instruction
// reads one SGPR pair for free).  Cody-Waite reduction x = n ln2 + r, |r| <= ln2/2, Taylor polynomial
// of degree 13 in Horner form (truncation 4e-18), ldexp.  x is clamped at -745 (the padding pairs of the
// sweep carry r^2 = 1e60).
struct ExpCoef {
  double log2e, ln2hi, ln2lo;
  double c[14];  // 1/k!
};
inline ExpCoef make_expcoef() {
  ExpCoef K;
  K.log2e = 1.4426950408889634074; K.ln2hi = 6.93147180369123816490e-01; K.ln2lo = 1.90821492927058770002e-10;
  const double c[14] = {1.0, 1.0, 0.5, 0.16666666666666666, 0.041666666666666664, 0.008333333333333333, 0.001388888888888889, 0.0001984126984126984, 2.48015873015873e-05, 2.7557319223985893e-06, 2.755731922398589e-07, 2.505210838544172e-08, 2.08767569878681e-09, 1.6059043836821613e-10};
  for (int k = 0; k < 14; k++) K.c[k] = c[k];
  return K;
}
__device__ __forceinline__ double exp_neg(double x, const ExpCoef &K) {
  x = fmax(x, -745.0);
  const double n = rint(x * K.log2e);
  double r = fma(-n, K.ln2hi, x);
  r = fma(-n, K.ln2lo, r);
  double p = K.c[13];
#pragma unroll
  for (int k = 12; k >= 0; k--) p = fma(p, r, K.c[k]);
  return ldexp(p, (int)n);
}

// Dipole field tensor scalars of build_dipole_field_matrix (PS.cpp:1284-1306):
//   T_pq = delta_pq * s3 - d_p d_q * s5,  s3 = damp1 / r^3,  s5 = 3 damp2 / r^5
template <int DAMP>
__device__ __forceinline__ void tensor_scalars(double r2, double pd, double &s3, double &s5) {
  double rinv = rsqrt(r2);
  double r = r2 * rinv;
  double rinv2 = rinv * rinv;
  double r3 = rinv * rinv2;
  double r5 = r3 * rinv2;
  if (DAMP == 0) {  // exponential (Thole-like) damping
    double ar = pd * r;
    double e = exp(-ar);
    double p2 = 1.0 + ar + 0.5 * ar * ar;
    double p3 = p2 + ar * ar * ar * (1.0 / 6.0);
    s3 = (1.0 - e * p2) * r3;
    s5 = 3.0 * (1.0 - e * p3) * r5;
  } else {
    s3 = r3;
    s5 = 3.0 * r5;
  }
}
// the same with exp_neg (constants in scalar registers): the sweep kernel's form
template <int DAMP>
__device__ __forceinline__ void tensor_scalars_k(double r2, double pd, const ExpCoef &K, double &s3, double &s5) {
  double rinv = rsqrt(r2);
  double r = r2 * rinv;
  double rinv2 = rinv * rinv;
  double r3 = rinv * rinv2;
  double r5 = r3 * rinv2;
  if (DAMP == 0) {  // exponential (Thole-like) damping
    double ar = pd * r;
    double e = exp_neg(-ar, K);
    double p2 = 1.0 + ar + 0.5 * ar * ar;
    double p3 = p2 + ar * ar * ar * (1.0 / 6.0);
    s3 = (1.0 - e * p2) * r3;
    s5 = 3.0 * (1.0 - e * p3) * r5;
  } else {
    s3 = r3;
    s5 = 3.0 * r5;
  }
}

}  // namespace polar
