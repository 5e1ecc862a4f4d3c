// polar_kernels.hpp -- hand-written HIP kernels (gfx950 / CDNA4, wave64) for the
// lj/cut/coul/long/polarization hot path.  FP64 throughout (the reference is FP64).
//
// Mapping used by every pair kernel: ONE WAVEFRONT PER ATOM ROW.  A 256-thread workgroup holds
// four rows; the 64 lanes stride the row's neighbor stream (coalesced 4-byte index loads), gather
// the 64-byte atom records of their j's (one cache line each), and the three field / force
// components are reduced across the wave with DPP/permute shuffles (no LDS, no atomics on i).
// No MFMA anywhere: this is a sparse neighbor stencil, not a dense contraction.
//
// Index spaces: "orig" = LAMMPS' atom index (inputs x/q/alpha/mol, outputs f/mu/ef_static);
// "s" = the library's internal order.  In list (dd_cutoff) mode s is CELL ORDER (perm[s] = orig,
// inv[orig] = s): the atoms of a cell are contiguous, so a row's neighbor stream -- emitted cell by
// cell -- makes the 64 lanes of a wave gather CONSECUTIVE 64-byte records, and the rows of one
// workgroup (neighbouring atoms) re-read the same records out of L1.  In exact (all-pairs) mode
// s == orig (perm == nullptr).
//
// Reference line numbers ("PS.cpp") are into
// /root/reference/src/pair_lj_cut_coul_long_polarization.cpp.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

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
  dx = xi - xj; dy = yi - yj; dz = zi - zj;
  if (b.periodic[0]) dx = fma(-b.prd[0], rint(dx * b.inv[0]), dx);
  if (b.periodic[1]) dy = fma(-b.prd[1], rint(dy * b.inv[1]), dy);
  if (b.periodic[2]) dz = fma(-b.prd[2], rint(dz * b.inv[2]), dz);
}
template <bool EXACT>
__device__ __forceinline__ void pair_del(const Box &b, double xi, double yi, double zi, double xj, double yj,
                                         double zj, double &dx, double &dy, double &dz) {
  if (EXACT) min_image_del(b, xi, yi, zi, xj, yj, zj, dx, dy, dz);
  else min_image_rint(b, xi, yi, zi, xj, yj, zj, dx, dy, dz);
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
  const int q4 = lane >> 2, k = lane & 3;
  const char *base = reinterpret_cast<const char *>(rec) + k * 16;
  const unsigned j0 = __shfl(j, q4, 64), j1 = __shfl(j, 16 + q4, 64), j2 = __shfl(j, 32 + q4, 64), j3 = __shfl(j, 48 + q4, 64);
  const double2 p0 = *reinterpret_cast<const double2 *>(base + ((size_t)j0 << 6));
  const double2 p1 = *reinterpret_cast<const double2 *>(base + ((size_t)j1 << 6));
  const double2 p2 = *reinterpret_cast<const double2 *>(base + ((size_t)j2 << 6));
  const double2 p3 = *reinterpret_cast<const double2 *>(base + ((size_t)j3 << 6));
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

// ------------------------------------------------------------------------------------------
// pack: x/q/alpha (+ initial mu) -> 64-byte records (both Jacobi buffers)
__global__ void k_pack(int n, const int *__restrict__ perm, const double *__restrict__ x, const double *__restrict__ q,
                       const double *__restrict__ alpha, const int *__restrict__ mol, const double *__restrict__ mu0,
                       AtomRec *__restrict__ r0, AtomRec *__restrict__ r1, int *__restrict__ mol_s,
                       double4 *__restrict__ pos4) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int o = perm ? perm[i] : i;
  AtomRec r;
  r.x = x[3 * o]; r.y = x[3 * o + 1]; r.z = x[3 * o + 2]; r.q = q[o];
  r.mx = mu0 ? mu0[3 * o] : 0.0; r.my = mu0 ? mu0[3 * o + 1] : 0.0; r.mz = mu0 ? mu0[3 * o + 2] : 0.0;
  r.a = alpha[o];
  r0[i] = r;
  r1[i] = r;
  mol_s[i] = mol[o];
  // 32-byte {x, y, z, (molecule id, alpha != 0)} for the list build
  if (pos4) pos4[i] = make_double4(r.x, r.y, r.z, __hiloint2double(mol[o], r.a != 0.0 ? 1 : 0));
}

// ------------------------------------------------------------------------------------------
// a2  rank metric, PS.cpp:192-227.  Pass 1: rmin; pass 2: rank_metric.
// ALLPAIRS: raw (non-minimum-image) distances to locals AND ghosts, exactly as the reference.
// list mode (extension): minimum-image distances over the library's full list.
template <bool ALLPAIRS, int PASS>
__global__ __launch_bounds__(POLAR_BLOCK) void k_rank(int nlocal, int ntotal, const double *__restrict__ x,
                                                      const double *__restrict__ alpha, const int *__restrict__ mol,
                                                      Box box, RowList nl,
                                                      const int *__restrict__ nl_j,
                                                      const AtomRec *__restrict__ rec,
                                                      const int *__restrict__ mol_s, Scal *scal,
                                                      double *__restrict__ slots,
                                                      double *__restrict__ rank_metric) {
  // ALLPAIRS: orig space (x/alpha/mol incl. ghosts).  List mode: s space (records, mol_s).
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * POLAR_ROWS_PER_BLOCK + (threadIdx.x >> 6);
  if (i >= nlocal) return;
  const double xi = ALLPAIRS ? x[3 * i] : rec[i].x, yi = ALLPAIRS ? x[3 * i + 1] : rec[i].y,
               zi = ALLPAIRS ? x[3 * i + 2] : rec[i].z, ai = ALLPAIRS ? alpha[i] : rec[i].a;
  const int mi = ALLPAIRS ? mol[i] : mol_s[i];
  double rmin = (PASS == 1) ? 1000.0 : __longlong_as_double((long long)scal->rmin_bits);
  double acc = 0.0;
  long long beg = 0, end = ntotal;
  if (!ALLPAIRS) row_range(nl, i, beg, end);
  for (long long base = beg; base < end; base += 64) {
    const long long p = base + lane;
    bool hit = false;
    double term = 0.0;
    if (p < end) {
      const int j = ALLPAIRS ? (int)p : (nl_j[p] & POLAR_NL_MASK);
      if (j != i) {
        double dx, dy, dz, aj;
        int mj;
        if (ALLPAIRS) {
          dx = xi - x[3 * j]; dy = yi - x[3 * j + 1]; dz = zi - x[3 * j + 2];
          aj = alpha[j]; mj = mol[j];
        } else {
          const AtomRec rj = rec[j];
          min_image_rint(box, xi, yi, zi, rj.x, rj.y, rj.z, dx, dy, dz);
          aj = rj.a; mj = mol_s[j];
        }
        const double r = sqrt(dx * dx + dy * dy + dz * dz);
        const bool molok = (mi != mj) || mi == 0;
        if (PASS == 1) {
          if (ai > 0 && aj > 0 && molok) rmin = fmin(rmin, r);
        } else if (rmin * 1.5 > r && molok) {
          hit = true;
          term = ai * aj;
        }
      }
    }
    if (PASS == 2) {
      // add the (few) qualifying terms in ascending j, like the reference's serial loop, so that
      // ties in rank_metric -- and with them the ranked sweep order -- come out bit-identical
      unsigned long long m = __ballot(hit);
      while (m) {
        const int b = __ffsll((long long)m) - 1;
        acc += __shfl(term, b, 64);
        m &= m - 1;
      }
    }
  }
  if (PASS == 1) {
    rmin = wave_min(rmin);
    if (lane == 0)
      atomicMin((unsigned long long *)slot_ptr(slots, SL_RMIN), (unsigned long long)__double_as_longlong(rmin));
  } else {
    if (lane == 0) rank_metric[i] = acc;  // identical in every lane
  }
}

// ------------------------------------------------------------------------------------------
// a3  LJ + real-space Ewald Coulomb over the LAMMPS half list, PS.cpp:232-321.
// One wave per listed atom i; -F is deposited on j (local or ghost) with FP64 atomics so that
// ghost forces come back exactly as LAMMPS' reverse_comm expects.
struct LJCoulParams {
  int ntypes, newton_pair, nlocal;
  int full_list;  // 1: LAMMPS full list (each pair in both rows): force on i only, tallies halved
  int ncoultablebits, ncoulmask, ncoulshiftbits;
  int ablate;     // lab switch (POLAR_ABLATE & 32: no deposit on j)
  double tabinnersq, cut_coulsq, g_ewald, qqrd2e;
  double special_lj[4], special_coul[4];
  const double *ljpack;   // [(ntypes+1)^2][8] = cutsq, cut_ljsq, lj1, lj2, lj3, lj4, offset, pad
  const double *ctab;     // [ntable][8]      = r, dr, f, df, e, de, c, dc  (one 64-byte line per bin)
};

// per-atom pack for the half-list loop: 32-byte {x,y,z,q} + type, locals AND ghosts, orig order
__global__ void k_pack_lj(int nall, const double *__restrict__ x, const double *__restrict__ q, double4 *__restrict__ xq) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nall) xq[i] = make_double4(x[3 * i], x[3 * i + 1], x[3 * i + 2], q[i]);
}

// Symmetrised copy of LAMMPS' half list, built on the device when the list is uploaded: every pair
// (i,j) of the half list appears in the row of i AND in the row of j (ghost atoms get rows too), so
// the force loop needs no atomics on j -- the three scattered FP64 atomics per pair were 85 % of
// the kernel.  Each row then accumulates the full force on its atom; pair tallies count 1/2 per row.
__global__ __launch_bounds__(POLAR_BLOCK) void k_sym_count(int inum, const int *__restrict__ ilist,
                                                           const int *__restrict__ numneigh,
                                                           const long long *__restrict__ first,
                                                           const int *__restrict__ neigh, int *__restrict__ cnt) {
  const int lane = threadIdx.x & 63;
  const int ii = blockIdx.x * POLAR_ROWS_PER_BLOCK + (threadIdx.x >> 6);
  if (ii >= inum) return;
  const int i = ilist[ii];
  const int *jl = neigh + first[i];
  const int jn = numneigh[i];
  for (int jj = lane; jj < jn; jj += 64) atomicAdd(&cnt[jl[jj] & 0x3FFFFFFF], 1);
  if (lane == 0) atomicAdd(&cnt[i], jn);
}
__global__ __launch_bounds__(POLAR_BLOCK) void k_sym_fill(int inum, const int *__restrict__ ilist,
                                                          const int *__restrict__ numneigh,
                                                          const long long *__restrict__ first,
                                                          const int *__restrict__ neigh,
                                                          const long long *__restrict__ sfirst, int *__restrict__ fill,
                                                          int *__restrict__ sj) {
  const int lane = threadIdx.x & 63;
  const int ii = blockIdx.x * POLAR_ROWS_PER_BLOCK + (threadIdx.x >> 6);
  if (ii >= inum) return;
  const int i = ilist[ii];
  const int *jl = neigh + first[i];
  const int jn = numneigh[i];
  // own row: one slot range per wave, entries in list order
  long long base = 0;
  if (lane == 0) base = sfirst[i] + atomicAdd(&fill[i], jn);
  base = __shfl(base, 0, 64);
  for (int jj = lane; jj < jn; jj += 64) {
    const int e = jl[jj];
    const int j = e & 0x3FFFFFFF;
    sj[base + jj] = e;                                                 // j with its special bits
    sj[sfirst[j] + atomicAdd(&fill[j], 1)] = i | (e & 0xC0000000);     // reverse entry, same bits
  }
}

// One wave per listed atom i.  Per pair: 1 coalesced index load, two 16-byte gathers of {x,y,z,q},
// one 4-byte gather of the type, the type-pair parameters out of LDS, and the Coulomb bin as one
// 64-byte line -- the loop is bound by L1 transactions and by the three FP64 atomics that deposit
// -F on j (LAMMPS' newton-on contract: ghosts are folded back by reverse_comm).
template <bool EFLAG, bool VPAIR>
__global__ __launch_bounds__(POLAR_BLOCK) void k_ljcoul(LJCoulParams P, int inum, const int *__restrict__ ilist,
                                                        const int *__restrict__ numneigh,
                                                        const long long *__restrict__ first,
                                                        const int *__restrict__ neigh,
                                                        const double4 *__restrict__ xq, const int *__restrict__ type,
                                                        double *__restrict__ f, double *__restrict__ slots,
                                                        double *__restrict__ eatom, double *__restrict__ vatom,
                                                        int vglobal) {
  const double EWALD_F = 1.12837917, EWALD_P = 0.3275911, A1 = 0.254829592, A2 = -0.284496736, A3 = 1.421413741,
               A4 = -1.453152027, A5 = 1.061405429;  // PS.cpp:43-49
  extern __shared__ double lj_lds[];
  const int w = P.ntypes + 1;
  for (int t = threadIdx.x; t < w * w * 8; t += blockDim.x) lj_lds[t] = P.ljpack[t];
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int ii = blockIdx.x * POLAR_ROWS_PER_BLOCK + (threadIdx.x >> 6);
  if (ii >= inum) return;
  const int i = ilist ? ilist[ii] : ii;
  const double4 pi = xq[i];
  const double qtmp = pi.w, xtmp = pi.x, ytmp = pi.y, ztmp = pi.z;
  const int itype = type[i];
  const int *jlist = neigh + first[i];
  const int jnum = numneigh ? numneigh[i] : (int)(first[i + 1] - first[i]);
  if (jnum == 0) return;
  double fx = 0, fy = 0, fz = 0, ev = 0, ec = 0;
  double v0 = 0, v1 = 0, v2 = 0, v3 = 0, v4 = 0, v5 = 0;
  for (int jj = lane; jj < jnum; jj += 64) {
    int j = jlist[jj];
    const int sb = (j >> 30) & 3;  // sbmask, src/pair.h:241
    const double factor_lj = P.special_lj[sb], factor_coul = P.special_coul[sb];
    j &= 0x3FFFFFFF;  // NEIGHMASK
    const double4 pj = xq[j];
    const double delx = xtmp - pj.x, dely = ytmp - pj.y, delz = ztmp - pj.z;
    const double rsq = delx * delx + dely * dely + delz * delz;
    const double *lj = lj_lds + (itype * w + type[j]) * 8;
    if (rsq < lj[0]) {
      const double r2inv = 1.0 / rsq;
      const double qiqj = qtmp * pj.w;
      double forcecoul = 0.0, forcelj = 0.0, prefactor = 0.0, erfc_ = 0.0, fraction = 0.0, r6inv = 0.0;
      double2 tab_e = make_double2(0.0, 0.0);
      bool direct = true;
      if (rsq < P.cut_coulsq) {
        direct = (!P.ncoultablebits) || (rsq <= P.tabinnersq);
        if (direct) {
          const double r = sqrt(rsq), grij = P.g_ewald * r, expm2 = exp(-grij * grij);
          const double t = 1.0 / (1.0 + EWALD_P * grij);
          erfc_ = t * (A1 + t * (A2 + t * (A3 + t * (A4 + t * A5)))) * expm2;
          prefactor = P.qqrd2e * qiqj / r;
          forcecoul = prefactor * (erfc_ + EWALD_F * grij * expm2);
          if (factor_coul < 1.0) forcecoul -= (1.0 - factor_coul) * prefactor;
        } else {
          const float rsqf = (float)rsq;  // union_int_float_t lookup, PS.cpp:268-272
          const int itable = (__float_as_int(rsqf) & P.ncoulmask) >> P.ncoulshiftbits;
          const double2 *bin = reinterpret_cast<const double2 *>(P.ctab + (size_t)itable * 8);
          const double2 rdr = bin[0], fdf = bin[1];
          if (EFLAG) tab_e = bin[2];
          fraction = ((double)rsqf - rdr.x) * rdr.y;
          forcecoul = qiqj * (fdf.x + fraction * fdf.y);
          if (factor_coul < 1.0) {
            const double2 cdc = bin[3];
            prefactor = qiqj * (cdc.x + fraction * cdc.y);
            forcecoul -= (1.0 - factor_coul) * prefactor;
          }
        }
      }
      if (rsq < lj[1]) {
        r6inv = r2inv * r2inv * r2inv;
        forcelj = r6inv * (lj[2] * r6inv - lj[3]);
      }
      const double fpair = (forcecoul + factor_lj * forcelj) * r2inv;
      fx += delx * fpair; fy += dely * fpair; fz += delz * fpair;
      if (!P.full_list && (P.newton_pair || j < P.nlocal) && !(P.ablate & 32)) {
        atomicAdd(&f[3 * j], -delx * fpair);
        atomicAdd(&f[3 * j + 1], -dely * fpair);
        atomicAdd(&f[3 * j + 2], -delz * fpair);
      }
      double wgt = 1.0;  // ev_tally, src/pair.cpp:854-950
      if (P.full_list) wgt = 0.5;  // ev_tally_full, src/pair.cpp:957-995
      else if (!P.newton_pair) wgt = 0.5 * ((i < P.nlocal) + (j < P.nlocal));
      if (EFLAG) {
        if (rsq < P.cut_coulsq) {
          double ecoul;
          if (direct) ecoul = prefactor * erfc_;
          else ecoul = qiqj * (tab_e.x + fraction * tab_e.y);
          if (factor_coul < 1.0) ecoul -= (1.0 - factor_coul) * prefactor;
          ec += wgt * ecoul;
        }
        if (rsq < lj[1]) ev += wgt * factor_lj * (r6inv * (lj[4] * r6inv - lj[5]) - lj[6]);
      }
      if (VPAIR) {
        v0 += wgt * delx * delx * fpair; v1 += wgt * dely * dely * fpair; v2 += wgt * delz * delz * fpair;
        v3 += wgt * delx * dely * fpair; v4 += wgt * delx * delz * fpair; v5 += wgt * dely * delz * fpair;
      }
    }
  }
  fx = wave_sum(fx); fy = wave_sum(fy); fz = wave_sum(fz);
  if (lane == 0) {
    atomicAdd(&f[3 * i], fx); atomicAdd(&f[3 * i + 1], fy); atomicAdd(&f[3 * i + 2], fz);
  }
  if (EFLAG) {
    ev = wave_sum(ev); ec = wave_sum(ec);
    if (lane == 0) {
      atomicAdd(slot_ptr(slots, SL_EVDWL), ev); atomicAdd(slot_ptr(slots, SL_ECOUL), ec);
      // per-atom energy, src/pair.cpp:881-885: every pair of a full row carries weight 1/2, so the
      // row total IS eatom[i] (one wave per row: plain store-add, no atomics)
      if (eatom) eatom[i] += ev + ec;
    }
  }
  if (VPAIR) {
    v0 = wave_sum(v0); v1 = wave_sum(v1); v2 = wave_sum(v2); v3 = wave_sum(v3); v4 = wave_sum(v4); v5 = wave_sum(v5);
    if (lane == 0) {
      if (vglobal) {
        atomicAdd(slot_ptr(slots, SL_V0), v0); atomicAdd(slot_ptr(slots, SL_V1), v1); atomicAdd(slot_ptr(slots, SL_V2), v2);
        atomicAdd(slot_ptr(slots, SL_V3), v3); atomicAdd(slot_ptr(slots, SL_V4), v4); atomicAdd(slot_ptr(slots, SL_V5), v5);
      }
      if (vatom) {  // src/pair.cpp:925-942
        double *va = vatom + 6 * (size_t)i;
        va[0] += v0; va[1] += v1; va[2] += v2; va[3] += v3; va[4] += v4; va[5] += v5;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// a4 + a5  static field (shifted-force Coulomb, PS.cpp:324-361), unit scale and initial guess
// (PS.cpp:363-386).  Full-row evaluation: E_i = sum_j ef_temp * q_j * del_ij, which is the
// reference's i<j scatter seen from row i (del is antisymmetric under the image rule).
template <bool ALLPAIRS>
__global__ __launch_bounds__(POLAR_BLOCK) void k_static_field(const int *__restrict__ rows, int nrows, int nlocal,
                                                              const AtomRec *__restrict__ rec,
                                                              const int *__restrict__ mol, Box box,
                                                              RowList nl,
                                                              const int *__restrict__ nl_j, double cut_coulsq,
                                                              double e2s, double gamma, int use_previous,
                                                              double *__restrict__ ef, AtomRec *__restrict__ rec0,
                                                              AtomRec *__restrict__ rec1) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * POLAR_ROWS_PER_BLOCK + (threadIdx.x >> 6);
  if (row >= nrows) return;
  const int i = rows ? rows[row] : row;
  const AtomRec ri = rec[i];
  const int mi = mol[i];
  const double f_shift = -1.0 / cut_coulsq;
  double ex = 0, ey = 0, ez = 0;
  long long beg = 0, end = nlocal;
  if (!ALLPAIRS) row_range(nl, i, beg, end);
  if (ALLPAIRS) {
    for (long long p = beg + lane; p < end; p += 64) {
      const int j = (int)p;
      if (j == i) continue;
      const AtomRec rj = rec[j];
      double dx, dy, dz;
      pair_del<ALLPAIRS>(box, ri.x, ri.y, ri.z, rj.x, rj.y, rj.z, dx, dy, dz);
      const double rsq = dx * dx + dy * dy + dz * dz;
      if (rsq <= cut_coulsq && ((mi != mol[j]) || mi == 0)) {  // note <=, PS.cpp:342
        const double rinv = rsqrt(rsq);
        const double ef_temp = (rinv * rinv + f_shift) * rinv * rj.q;
        ex += ef_temp * dx; ey += ef_temp * dy; ez += ef_temp * dz;
      }
    }
  } else {
    __shared__ double2 s_stage[POLAR_ROWS_PER_BLOCK][64 * 5];
    double2 *stage = s_stage[threadIdx.x >> 6];
    for (long long base = beg; base < end; base += 64) {  // wave-uniform trip count: the fetch is cooperative
      const long long p = base + lane;
      const bool valid = p < end;
      const int e = valid ? nl_j[p] : i;
      const int j = e & POLAR_NL_MASK;
      const RecQuad rj = fetch_records(rec, j, stage, lane);
      double dx, dy, dz;
      pair_del<ALLPAIRS>(box, ri.x, ri.y, ri.z, rj.a.x, rj.b.x, rj.c.x, dx, dy, dz);
      const double rsq = dx * dx + dy * dy + dz * dz;
      if (valid && j != i && rsq <= cut_coulsq && !(e & POLAR_NL_SAMEMOL)) {  // note <=, PS.cpp:342
        const double rinv = rsqrt(rsq);
        const double ef_temp = (rinv * rinv + f_shift) * rinv * rj.d.x;
        ex += ef_temp * dx; ey += ef_temp * dy; ez += ef_temp * dz;
      }
    }
  }
  ex = wave_sum(ex); ey = wave_sum(ey); ez = wave_sum(ez);
  if (lane == 0) {
    ex *= e2s; ey *= e2s; ez *= e2s;
    ef[3 * i] = ex; ef[3 * i + 1] = ey; ef[3 * i + 2] = ez;
    if (!use_previous) {  // mu = gamma * alpha * E
      const double a = ri.a;
      double mx = a * ex, my = a * ey, mz = a * ez;
      mx *= gamma; my *= gamma; mz *= gamma;
      rec0[i].mx = mx; rec0[i].my = my; rec0[i].mz = mz;
      rec1[i].mx = mx; rec1[i].my = my; rec1[i].mz = mz;
    }
  }
}

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
__global__ __launch_bounds__(1024) void k_field(int nrows, const int *__restrict__ rows, int nlocal,
                                                       AtomRec *__restrict__ recA, AtomRec *__restrict__ recB, Box box,
                                                       RowList ddl,
                                                       const int *__restrict__ dd_j,
                                                       const double2 *__restrict__ dd_s, double ddcutsq, double pd,
                                                       const double *__restrict__ ef, double *__restrict__ Fout,
                                                       const Scal *scal, double *__restrict__ slots, int ablate) {
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
      // list mode.  The damped tensor scalars (s3, s5) were cached per pair by k_dd_scalars, so a
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
      if (c != 0.0 && !(ablate & 16)) atomicAdd(slot_ptr(slots, SL_CHANGE), c);
    }
  }
}

// ------------------------------------------------------------------------------------------
// List-mode sweep, production form: ONE WAVE STREAMS SEVERAL ROWS.
// A row has only ~6 trips of 64 pairs, so a per-row software pipeline spends most of its life
// filling and draining (measured: memory time was not overlapped at all).  Here a wave owns `rpw`
// consecutive rows of the launch and runs ONE continuous 3-stage pipeline across them:
//   slot g:  stream(j, s3/s5) of trip g+2  |  quad-cooperative record gather of trip g+1  |
//            LDS transpose + lane-per-pair math of trip g
// Row boundaries only reset the accumulators (wave reduction + epilogue by lane 0).  The body is
// unrolled six times so the three stream sets and two gather sets rotate by name (no copies, no
// wait right behind an issue).  Per-row data (index, count, x_i, mu_i, alpha_i, E_i) sits in a
// small per-wave LDS table filled by lanes 0..rpw-1.
#define POLAR_RPW 8
template <int EP>
__global__ __launch_bounds__(POLAR_BLOCK) void k_field_rows(int nrows, const int *__restrict__ rows, int rpw,
                                                            AtomRec *__restrict__ recA, AtomRec *__restrict__ recB,
                                                            Box box, RowList ddl, const int *__restrict__ dd_j,
                                                            const double2 *__restrict__ dd_s,
                                                            const double *__restrict__ ef, const Scal *scal,
                                                            double *__restrict__ slots) {
  if (scal->done) return;
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform: keep the control flow scalar
  const int r0 = (blockIdx.x * POLAR_ROWS_PER_BLOCK + wv) * rpw;
  if (r0 >= nrows) return;
  const int nr = min(rpw, nrows - r0);
  const int cur = scal->cur;
  const AtomRec *__restrict__ src = (EP == EP_JACOBI && cur) ? recB : recA;
  AtomRec *__restrict__ dst = (EP == EP_JACOBI) ? (cur ? recA : recB) : recA;

  __shared__ double s_info[POLAR_ROWS_PER_BLOCK][POLAR_RPW][10];  // x,y,z, mx,my,mz, a, Ex,Ey,Ez
  __shared__ int s_idx[POLAR_ROWS_PER_BLOCK][POLAR_RPW][2];        // atom index, pair count
  __shared__ double2 s_stage[POLAR_ROWS_PER_BLOCK][64 * 5];
  double (*info)[10] = s_info[wv];
  int (*idx)[2] = s_idx[wv];
  double2 *stage = s_stage[wv];

  double chg = 0.0;  // sum |dmu|^2 of this wave's rows (lane 0)
  if (lane < nr) {
    const int i = rows ? rows[r0 + lane] : r0 + lane;
    const AtomRec r = src[i];
    const double e0 = ef[3 * i], e1 = ef[3 * i + 1], e2 = ef[3 * i + 2];
    long long c = ddl.cnt[i];
    if (c > ddl.pitch) c = ddl.pitch;
    if (r.a == 0.0) c = 0;
    info[lane][0] = r.x; info[lane][1] = r.y; info[lane][2] = r.z;
    info[lane][3] = r.mx; info[lane][4] = r.my; info[lane][5] = r.mz;
    info[lane][6] = r.a; info[lane][7] = e0; info[lane][8] = e1; info[lane][9] = e2;
    idx[lane][0] = i; idx[lane][1] = (int)c;
    if (c == 0) {  // no listed neighbor: mu_new = alpha * E right away
      const double mx = r.a * e0, my = r.a * e1, mz = r.a * e2;
      dst[i].mx = mx; dst[i].my = my; dst[i].mz = mz;
      const double ax = mx - r.mx, ay = my - r.my, az = mz - r.mz;
      chg = ax * ax + ay * ay + az * az;
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  chg = wave_sum(chg);  // all in lane registers; lane 0 keeps the running total

  const int q4 = lane >> 2, k = lane & 3;
  const char *srcb = reinterpret_cast<const char *>(src) + k * 16;
  const bool allper = box.periodic[0] && box.periodic[1] && box.periodic[2];

  // stream cursor: row sr (within this wave), offset so inside it; skips empty rows
  int sr = 0, so = 0, s_i = 0, s_cnt = 0;
#define POLAR_CURSOR_NORMALISE()                                            \
  while (sr < nr) {                                                         \
    s_i = __builtin_amdgcn_readfirstlane(idx[sr][0]);                       \
    s_cnt = __builtin_amdgcn_readfirstlane(idx[sr][1]);                     \
    if (so < s_cnt) break;                                                  \
    sr++; so = 0;                                                           \
  }
  POLAR_CURSOR_NORMALISE();
  const int i_any = __builtin_amdgcn_readfirstlane(idx[0][0]);

// The loaded values are NOT touched here (a select on them would force a wait right behind the
// issue); lanes past the row's end are masked when the values are consumed (REM = valid lanes).
// The loads are issued unconditionally from a clamped address (no branch around a VMEM op: the
// compiler's s_waitcnt bookkeeping stays exact only in straight-line code).
#define POLAR_LOAD_STREAM(JM, SC, ROW, LAST, REM)                                          \
  {                                                                                        \
    const bool live_ = sr < nr;                                                            \
    REM = live_ ? s_cnt - so : 0;                                                          \
    const long long pc_ = (long long)(live_ ? s_i : i_any) * ddl.pitch + (lane < REM ? so + lane : 0); \
    JM = dd_j[pc_];                                                                        \
    SC = dd_s[pc_];                                                                        \
    ROW = live_ ? sr : -1;                                                                 \
    so += 64;                                                                              \
    LAST = live_ && so >= s_cnt;                                                           \
    if (LAST) { sr++; so = 0; POLAR_CURSOR_NORMALISE(); }                                  \
  }
#define POLAR_GATHER(JM, REM, P0, P1, P2, P3)                                              \
  {                                                                                        \
    const int jm_ = lane < REM ? JM : i_any; /* lanes past the end gather a harmless record */ \
    const unsigned j0_ = __shfl(jm_, q4, 64), j1_ = __shfl(jm_, 16 + q4, 64);               \
    const unsigned j2_ = __shfl(jm_, 32 + q4, 64), j3_ = __shfl(jm_, 48 + q4, 64);          \
    P0 = *reinterpret_cast<const double2 *>(srcb + ((size_t)j0_ << 6));                     \
    P1 = *reinterpret_cast<const double2 *>(srcb + ((size_t)j1_ << 6));                     \
    P2 = *reinterpret_cast<const double2 *>(srcb + ((size_t)j2_ << 6));                     \
    P3 = *reinterpret_cast<const double2 *>(srcb + ((size_t)j3_ << 6));                     \
  }
  int crow = -1;
  double xi = 0, yi = 0, zi = 0, fx = 0, fy = 0, fz = 0;
#define POLAR_COMPUTE(P0, P1, P2, P3, SCRAW, ROW, LAST, REM)                                 \
  {                                                                                        \
    if (ROW < 0) break;                                                                    \
    const double2 SC = lane < REM ? SCRAW : make_double2(0.0, 0.0);                        \
    if (ROW != crow) { crow = ROW; xi = info[crow][0]; yi = info[crow][1]; zi = info[crow][2]; } \
    stage[q4 * 5 + k] = P0; stage[(16 + q4) * 5 + k] = P1;                                  \
    stage[(32 + q4) * 5 + k] = P2; stage[(48 + q4) * 5 + k] = P3;                           \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");                                 \
    __builtin_amdgcn_wave_barrier();                                                       \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");                                 \
    const double2 a_ = stage[lane * 5], b_ = stage[lane * 5 + 1], c_ = stage[lane * 5 + 2]; \
    __builtin_amdgcn_wave_barrier();                                                       \
    double dx_ = xi - a_.x, dy_ = yi - b_.x, dz_ = zi - c_.x;                               \
    if (allper) {                                                                          \
      dx_ = fma(-box.prd[0], rint(dx_ * box.inv[0]), dx_);                                 \
      dy_ = fma(-box.prd[1], rint(dy_ * box.inv[1]), dy_);                                 \
      dz_ = fma(-box.prd[2], rint(dz_ * box.inv[2]), dz_);                                 \
    } else {                                                                               \
      if (box.periodic[0]) dx_ = fma(-box.prd[0], rint(dx_ * box.inv[0]), dx_);            \
      if (box.periodic[1]) dy_ = fma(-box.prd[1], rint(dy_ * box.inv[1]), dy_);            \
      if (box.periodic[2]) dz_ = fma(-box.prd[2], rint(dz_ * box.inv[2]), dz_);            \
    }                                                                                      \
    const double md_ = a_.y * dx_ + b_.y * dy_ + c_.y * dz_;                               \
    const double cc_ = SC.y * md_;                                                         \
    fx -= SC.x * a_.y - cc_ * dx_;                                                         \
    fy -= SC.x * b_.y - cc_ * dy_;                                                         \
    fz -= SC.x * c_.y - cc_ * dz_;                                                         \
    if (LAST) { /* end of the row: reduce, update the dipole, restart the accumulators */   \
      fx = wave_sum(fx); fy = wave_sum(fy); fz = wave_sum(fz);                             \
      if (lane == 0) {                                                                     \
        const double al_ = info[crow][6];                                                  \
        const double mx_ = al_ * (info[crow][7] + fx), my_ = al_ * (info[crow][8] + fy),    \
                     mz_ = al_ * (info[crow][9] + fz);                                     \
        const int ii_ = idx[crow][0];                                                      \
        dst[ii_].mx = mx_; dst[ii_].my = my_; dst[ii_].mz = mz_;                            \
        const double ax_ = mx_ - info[crow][3], ay_ = my_ - info[crow][4], az_ = mz_ - info[crow][5]; \
        chg += ax_ * ax_ + ay_ * ay_ + az_ * az_;                                          \
      }                                                                                    \
      fx = fy = fz = 0.0;                                                                  \
    }                                                                                      \
  }
#define POLAR_TRIP(SA, RA, LA, MA, JB, MB, JC, SC_, RC, LC, MC, GA0, GA1, GA2, GA3, GB0, GB1, GB2, GB3) \
  {                                                                                        \
    POLAR_LOAD_STREAM(JC, SC_, RC, LC, MC);                                                \
    POLAR_GATHER(JB, MB, GB0, GB1, GB2, GB3);                                              \
    POLAR_COMPUTE(GA0, GA1, GA2, GA3, SA, RA, LA, MA);                                     \
  }
  int j_0 = i_any, j_1 = i_any, j_2 = i_any, w_0 = -1, w_1 = -1, w_2 = -1, m_0 = 0, m_1 = 0, m_2 = 0;
  bool l_0 = false, l_1 = false, l_2 = false;
  const double2 z2 = make_double2(0.0, 0.0);
  double2 s_0 = z2, s_1 = z2, s_2 = z2;
  double2 ga0, ga1, ga2, ga3, gb0 = z2, gb1 = z2, gb2 = z2, gb3 = z2;
  POLAR_LOAD_STREAM(j_0, s_0, w_0, l_0, m_0);
  POLAR_LOAD_STREAM(j_1, s_1, w_1, l_1, m_1);
  POLAR_GATHER(j_0, m_0, ga0, ga1, ga2, ga3);
  for (;;) {
    POLAR_TRIP(s_0, w_0, l_0, m_0, j_1, m_1, j_2, s_2, w_2, l_2, m_2, ga0, ga1, ga2, ga3, gb0, gb1, gb2, gb3);
    POLAR_TRIP(s_1, w_1, l_1, m_1, j_2, m_2, j_0, s_0, w_0, l_0, m_0, gb0, gb1, gb2, gb3, ga0, ga1, ga2, ga3);
    POLAR_TRIP(s_2, w_2, l_2, m_2, j_0, m_0, j_1, s_1, w_1, l_1, m_1, ga0, ga1, ga2, ga3, gb0, gb1, gb2, gb3);
    POLAR_TRIP(s_0, w_0, l_0, m_0, j_1, m_1, j_2, s_2, w_2, l_2, m_2, gb0, gb1, gb2, gb3, ga0, ga1, ga2, ga3);
    POLAR_TRIP(s_1, w_1, l_1, m_1, j_2, m_2, j_0, s_0, w_0, l_0, m_0, ga0, ga1, ga2, ga3, gb0, gb1, gb2, gb3);
    POLAR_TRIP(s_2, w_2, l_2, m_2, j_0, m_0, j_1, s_1, w_1, l_1, m_1, gb0, gb1, gb2, gb3, ga0, ga1, ga2, ga3);
  }
#undef POLAR_TRIP
#undef POLAR_COMPUTE
#undef POLAR_GATHER
#undef POLAR_LOAD_STREAM
#undef POLAR_CURSOR_NORMALISE
  if (lane == 0 && chg != 0.0) atomicAdd(slot_ptr(slots, SL_CHANGE), chg);
}

// ------------------------------------------------------------------------------------------
// List-mode sweep, component-per-lane form (production).
// k_field / k_field_rows give every LANE one pair, so the 64-byte records fetched quad-wise have to
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
__global__ __launch_bounds__(1024) void k_field_quad(int nrows, const int *__restrict__ rows,
                                                            AtomRec *__restrict__ recA, AtomRec *__restrict__ recB,
                                                            Box box, RowList ddl, const int *__restrict__ dd_j,
                                                            const double2 *__restrict__ dd_s,
                                                            const double *__restrict__ dd_r2, double pd,
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
  const int cur = scal->cur;
  const AtomRec *__restrict__ src = (EP == EP_JACOBI && cur) ? recB : recA;
  AtomRec *__restrict__ dst = (EP == EP_JACOBI) ? (cur ? recA : recB) : recA;
  const int k = lane & 3, kk = k < 3 ? k : 2;
  const double *ri = reinterpret_cast<const double *>(src + i);
  const double xi = ri[2 * kk], mi = ri[2 * kk + 1], ai = ri[7];
  long long c = ddl.cnt[i];
  if (c > ddl.pitch) c = ddl.pitch;
  if (ai == 0.0) c = 0;
  int T = __builtin_amdgcn_readfirstlane((int)((c + 63) >> 6));
  if (ablate & 1) T = 0;             // lab switches (POLAR_ABLATE): timing only, wrong numbers
  if ((ablate & 8) && T > 1) T = 1;
  const double prd = box.periodic[kk] ? box.prd[kk] : 0.0, inv = box.inv[kk];
  const char *srcb = reinterpret_cast<const char *>(src) + kk * 16;
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
    P[R] = *reinterpret_cast<const double2 *>(srcb + ((size_t)j_ << 6));                        \
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
    if (SMODE != 0) tensor_scalars<(SMODE == 1 || SMODE == 3) ? 0 : 1>(r2v, pd, Sv.x, Sv.y);  // lane L: pair L of the trip
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
  const double mu_new = ai * (ef[3 * i + kk] + acc);
  const double dm = mu_new - mi;
  double chg = (k < 3) ? dm * dm : 0.0;
  chg = chg + dpp_full<0xC9>(chg) + dpp_full<0xD2>(chg);
  if (lane < 3) reinterpret_cast<double *>(dst + i)[2 * lane + 1] = mu_new;
  if (lane == 0 && chg != 0.0) atomicAdd(slot_ptr(slots, SL_CHANGE), chg);
}

// a6 for the list path: the damped tensor scalars of every listed pair, once per step
// (the sparse, matrix-free-storage analog of build_dipole_field_matrix, PS.cpp:1273-1306).
template <int DAMP>
__global__ __launch_bounds__(POLAR_BLOCK) void k_dd_scalars(const int *__restrict__ rows, int nrows, const AtomRec *__restrict__ rec,
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
  if (dd_r2 || dd_s)
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
  // pad the row to whole 64-pair trips with inert entries (the atom itself, zero tensor): the
  // component-per-lane sweep then runs without lane masks.  The pitch is a multiple of 64.
  const long long pad_end = beg + (((end - beg) + 63) & ~63ll);
  for (long long p = end + lane; p < pad_end; p += 64) {
    dd_j[p] = i;
    if (dd_r2) dd_r2[p] = 1e60;  // s3 ~ 1e-90, and d = 0 kills the s5 term: contributes nothing
    else if (dd_s) dd_s[p] = make_double2(0.0, 0.0);
  }
}

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
__global__ __launch_bounds__(64) void k_gs_block_seq(int nlocal, int b0, const int *__restrict__ order,
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
__global__ __launch_bounds__(POLAR_BLOCK) void k_gs_block_push(int nlocal, int b0, const int *__restrict__ order,
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
template <int DAMP>
__global__ __launch_bounds__(POLAR_BLOCK) void k_build_T6(int n, const AtomRec *__restrict__ rec, Box box, double pd,
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
    double *o = T6 + ((size_t)i * n + j) * 6;
#pragma unroll
    for (int c = 0; c < 6; c++) o[c] = t[c];
  }
}

// F_i = - sum_j T_ij mu_j  (dense mat-vec; initial running field of the Gauss-Seidel)
__global__ __launch_bounds__(POLAR_BLOCK) void k_dense_field(int n, const double *__restrict__ T6,
                                                             const AtomRec *__restrict__ rec, double *__restrict__ F) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * POLAR_ROWS_PER_BLOCK + (threadIdx.x >> 6);
  if (i >= n) return;
  double fx = 0, fy = 0, fz = 0;
  for (int j = lane; j < n; j += 64) {
    const double *t = T6 + ((size_t)i * n + j) * 6;
    const double mx = rec[j].mx, my = rec[j].my, mz = rec[j].mz;
    fx -= t[0] * mx + t[1] * my + t[2] * mz;
    fy -= t[1] * mx + t[3] * my + t[4] * mz;
    fz -= t[2] * mx + t[4] * my + t[5] * mz;
  }
  fx = wave_sum(fx); fy = wave_sum(fy); fz = wave_sum(fz);
  if (lane == 0) { F[3 * i] = fx; F[3 * i + 1] = fy; F[3 * i + 2] = fz; }
}

__device__ __forceinline__ double readlane_d(double v, int k) {  // k wave-uniform
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), k);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), k);
  return __hiloint2double(hi, lo);
}

// ONE wave: the sequential recurrence over the 64 atoms b0..b0+63 of the sweep order
// (PS.cpp:1158-1180).  Lane l owns atom b0+l; step k: mu_k <- alpha_k (E_k + F_k), and every lane
// folds T_{l,k} dmu_k into its running field.  Tensor rows are prefetched four steps ahead.
__global__ __launch_bounds__(64) void k_gs_seq_T6(int n, int b0, const double *__restrict__ T6,
                                                  AtomRec *__restrict__ rec, const double *__restrict__ ef,
                                                  double *__restrict__ F, double *__restrict__ dmu_blk,
                                                  const Scal *scal, double *__restrict__ slots) {
  if (scal->done) return;
  const int lane = threadIdx.x;
  const int cnt = min(64, n - b0);
  const bool act = lane < cnt;
  const int i = act ? b0 + lane : b0;
  const AtomRec r = rec[i];
  const double a = act ? r.a : 0.0;
  double mx = r.mx, my = r.my, mz = r.mz;
  const double mx0 = mx, my0 = my, mz0 = mz;
  double Fx = F[3 * i], Fy = F[3 * i + 1], Fz = F[3 * i + 2];
  const double Ex = ef[3 * i], Ey = ef[3 * i + 1], Ez = ef[3 * i + 2];
  // T_{k,l} = T_{l,k}: read row (b0+k), columns b0..b0+63 -> consecutive lanes, contiguous 3 KB
  const double *tcol = T6 + ((size_t)b0 * n + i) * 6;
  const size_t rowstride = (size_t)n * 6;
  double ta[4][6], tb[4][6];
#define POLAR_LOADT(BUF, K0)                                                     \
  _Pragma("unroll") for (int u = 0; u < 4; u++) {                                \
    const int kk = (K0) + u < cnt ? (K0) + u : cnt - 1;                          \
    const double *t_ = tcol + (size_t)kk * rowstride;                            \
    _Pragma("unroll") for (int c = 0; c < 6; c++) BUF[u][c] = t_[c];             \
  }
#define POLAR_STEPT(BUF, K0)                                                     \
  _Pragma("unroll") for (int u = 0; u < 4; u++) {                                \
    const int k = (K0) + u;                                                      \
    if (k < cnt) {                                                               \
      const double nx = a * (Ex + Fx), ny = a * (Ey + Fy), nz = a * (Ez + Fz);   \
      const double bdx = readlane_d(nx - mx, k), bdy = readlane_d(ny - my, k),   \
                   bdz = readlane_d(nz - mz, k);                                 \
      if (lane == k) { mx = nx; my = ny; mz = nz; }                              \
      /* the diagonal block T_kk is stored as zero: lane k leaves its own field alone */ \
      Fx -= BUF[u][0] * bdx + BUF[u][1] * bdy + BUF[u][2] * bdz;                 \
      Fy -= BUF[u][1] * bdx + BUF[u][3] * bdy + BUF[u][4] * bdz;                 \
      Fz -= BUF[u][2] * bdx + BUF[u][4] * bdy + BUF[u][5] * bdz;                 \
    }                                                                            \
  }
  POLAR_LOADT(ta, 0);
  for (int k0 = 0; k0 < cnt; k0 += 8) {
    POLAR_LOADT(tb, k0 + 4);
    POLAR_STEPT(ta, k0);
    POLAR_LOADT(ta, k0 + 8);
    POLAR_STEPT(tb, k0 + 4);
  }
#undef POLAR_LOADT
#undef POLAR_STEPT
  double dsq = 0.0;
  if (act) {
    const double tx = mx - mx0, ty = my - my0, tz = mz - mz0;
    dsq = tx * tx + ty * ty + tz * tz;
    rec[i].mx = mx; rec[i].my = my; rec[i].mz = mz;
    F[3 * i] = Fx; F[3 * i + 1] = Fy; F[3 * i + 2] = Fz;
    dmu_blk[3 * lane] = tx; dmu_blk[3 * lane + 1] = ty; dmu_blk[3 * lane + 2] = tz;
  }
  dsq = wave_sum(dsq);
  if (lane == 0 && dsq != 0.0) atomicAdd(slots + (size_t)((b0 >> 6) & (POLAR_NSLOT - 1)) * POLAR_SLOT_STRIDE + SL_CHANGE, dsq);
}

// rows outside the block receive the block's dipole changes: F_j -= sum_k T_{j,b0+k} dmu_k
__global__ __launch_bounds__(POLAR_BLOCK) void k_gs_push_T6(int n, int b0, const double *__restrict__ T6,
                                                            const AtomRec *__restrict__ rec,
                                                            const double *__restrict__ dmu_blk, double *__restrict__ F,
                                                            const Scal *scal) {
  if (scal->done) return;
  const int lane = threadIdx.x & 63;
  const int j = blockIdx.x * POLAR_ROWS_PER_BLOCK + (threadIdx.x >> 6);
  if (j >= n || (j >= b0 && j < b0 + 64)) return;
  if (rec[j].a == 0.0) return;  // mu_j stays 0: its field is never read
  const int cnt = min(64, n - b0);
  double fx = 0, fy = 0, fz = 0;
  if (lane < cnt) {
    const double *t = T6 + ((size_t)j * n + b0 + lane) * 6;
    const double bdx = dmu_blk[3 * lane], bdy = dmu_blk[3 * lane + 1], bdz = dmu_blk[3 * lane + 2];
    fx = -(t[0] * bdx + t[1] * bdy + t[2] * bdz);
    fy = -(t[1] * bdx + t[3] * bdy + t[4] * bdz);
    fz = -(t[2] * bdx + t[4] * bdy + t[5] * bdz);
  }
  fx = wave_sum(fx); fy = wave_sum(fy); fz = wave_sum(fz);
  if (lane == 0) { F[3 * j] += fx; F[3 * j + 1] += fy; F[3 * j + 2] += fz; }
}

// ------------------------------------------------------------------------------------------
// a7 loop control, one thread: the reference's end-of-sweep logic (PS.cpp:1193-1236) kept on the
// device so the host never has to look at ||dmu||^2 between sweeps.
__global__ __launch_bounds__(POLAR_NSLOT) void k_solver_step(Scal *scal, double *__restrict__ slots, int nlocal,
                                                           int fixed_iteration, int iterations_max, double precision,
                                                           int jacobi, const double *__restrict__ global_change,
                                                           int count) {
  if (scal->done) return;
  __shared__ double red[POLAR_NSLOT / 64];
  double v = slots[(size_t)threadIdx.x * POLAR_SLOT_STRIDE + SL_CHANGE];
  slots[(size_t)threadIdx.x * POLAR_SLOT_STRIDE + SL_CHANGE] = 0.0;
  v = wave_sum(v);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x != 0) return;
  double sum = 0.0;
  for (int k = 0; k < POLAR_NSLOT / 64; k++) sum += red[k];
  scal->change = sum;  // this handle's own sum (exported to the all-reduce in multi-GPU runs)
  // multi-GPU: the all-reduced sum over ranks arrives through global_change (device memory)
  const double change = (global_change ? *global_change : sum) / ((double)nlocal * 3.0);
  scal->last_change = change;
  // `count` > 1: the end-of-sweep logic of several sweeps at once (fixed-iteration Gauss-Seidel takes
  // no decision between sweeps, so the host launches this only before and after the last one)
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

// fold the change slots into scal->change without touching the loop state (multi-GPU export)
__global__ __launch_bounds__(POLAR_NSLOT) void k_fold_change(Scal *scal, double *__restrict__ slots, double *dst) {
  __shared__ double red[POLAR_NSLOT / 64];
  double v = slots[(size_t)threadIdx.x * POLAR_SLOT_STRIDE + SL_CHANGE];
  slots[(size_t)threadIdx.x * POLAR_SLOT_STRIDE + SL_CHANGE] = 0.0;
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
__global__ __launch_bounds__(POLAR_NSLOT) void k_fold_scal(Scal *scal, double *__restrict__ slots, int rmin_only) {
  __shared__ double red[POLAR_NSLOT / 64];
  const int t = threadIdx.x;
  {
    unsigned long long b = ((unsigned long long *)slots)[(size_t)t * POLAR_SLOT_STRIDE + SL_RMIN];
    double r = __longlong_as_double((long long)b);
    r = wave_min(r);
    if ((t & 63) == 0) red[t >> 6] = r;
    __syncthreads();
    if (t == 0) {
      double m = red[0];
      for (int k = 1; k < POLAR_NSLOT / 64; k++) m = fmin(m, red[k]);
      scal->rmin_bits = (unsigned long long)__double_as_longlong(m);
    }
    __syncthreads();
  }
  if (rmin_only) return;
  for (int f = SL_EVDWL; f <= SL_V5; f++) {
    double v = slots[(size_t)t * POLAR_SLOT_STRIDE + f];
    v = wave_sum(v);
    if ((t & 63) == 0) red[t >> 6] = v;
    __syncthreads();
    if (t == 0) {
      double sum = 0.0;
      for (int k = 0; k < POLAR_NSLOT / 64; k++) sum += red[k];
      double *dst = f == SL_EVDWL ? &scal->eng_vdwl : f == SL_ECOUL ? &scal->eng_coul : f == SL_USELF ? &scal->u_self
                  : f == SL_UEF ? &scal->u_ef : f == SL_UDD ? &scal->u_dd : &scal->virial[f - SL_V0];
      *dst = sum;
    }
    __syncthreads();
  }
}
__global__ void k_zero_slots(double *__restrict__ slots) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= POLAR_NSLOT) return;
  for (int f = 0; f < POLAR_SLOT_STRIDE; f++) slots[(size_t)t * POLAR_SLOT_STRIDE + f] = 0.0;
  ((unsigned long long *)slots)[(size_t)t * POLAR_SLOT_STRIDE + SL_RMIN] = (unsigned long long)__double_as_longlong(1000.0);
}

// divergence fallback mu = alpha * E (no gamma), PS.cpp:1227-1235
__global__ void k_fallback(int n, const Scal *scal, AtomRec *__restrict__ recA, AtomRec *__restrict__ recB,
                           const double *__restrict__ ef) {
  if (!scal->status) return;
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  AtomRec *r = scal->cur ? recB : recA;
  const double a = r[i].a;
  r[i].mx = a * ef[3 * i]; r[i].my = a * ef[3 * i + 1]; r[i].mz = a * ef[3 * i + 2];
}

// copy the final dipoles and the static field out (records are in s order, outputs in orig order)
__global__ void k_unpack(int n, const int *__restrict__ perm, const Scal *scal, const AtomRec *__restrict__ recA,
                         const AtomRec *__restrict__ recB, const double *__restrict__ ef_s, double *__restrict__ mu,
                         double *__restrict__ ef) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const AtomRec *r = scal->cur ? recB : recA;
  const int o = perm ? perm[i] : i;
  mu[3 * o] = r[i].mx; mu[3 * o + 1] = r[i].my; mu[3 * o + 2] = r[i].mz;
  ef[3 * o] = ef_s[3 * i]; ef[3 * o + 1] = ef_s[3 * i + 1]; ef[3 * o + 2] = ef_s[3 * i + 2];
}

// ------------------------------------------------------------------------------------------
// a8  polarization forces and energies, PS.cpp:406-641, evaluated per row (force on i from every j).
// The pair force is antisymmetric, so summing rows reproduces the reference's i<j scatter;
// pair energies are counted from both rows and halved.
template <bool ALLPAIRS, int DAMP, bool EFLAG, bool VPAIR>
__global__ __launch_bounds__(POLAR_BLOCK) void k_polar_force(const int *__restrict__ rows, int nrows, const int *__restrict__ perm,
                                                             int nlocal, const Scal *scal_in,
                                                             const AtomRec *__restrict__ recA,
                                                             const AtomRec *__restrict__ recB,
                                                             const int *__restrict__ mol, Box box,
                                                             RowList nl,
                                                             const int *__restrict__ nl_j, double cut_coulsq,
                                                             double ddcutsq, double pd, double e2s,
                                                             double *__restrict__ f, double *__restrict__ slots,
                                                             double *__restrict__ vatom, int vglobal) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * POLAR_ROWS_PER_BLOCK + (threadIdx.x >> 6);
  if (row >= nrows) return;
  const int i = rows ? rows[row] : row;
  const AtomRec *__restrict__ rec = scal_in->cur ? recB : recA;
  const AtomRec ri = rec[i];
  const int mi = mol[i];
  const double f_shift = -1.0 / cut_coulsq;
  double fx = 0, fy = 0, fz = 0, uef = 0, udd = 0;
  double v0 = 0, v1 = 0, v2 = 0, v3 = 0, v4 = 0, v5 = 0;
  long long beg = 0, end = nlocal;
  if (!ALLPAIRS) row_range(nl, i, beg, end);
  // (the cooperative record fetch of k_static_field was tried here too: this kernel is bound by its FP64
  //  arithmetic, not by the gathers, and got 7 % slower)
  for (long long p = beg + lane; p < end; p += 64) {
    const int e = ALLPAIRS ? (int)p : nl_j[p];
    const int j = ALLPAIRS ? e : (e & POLAR_NL_MASK);
    if (j == i) continue;
    const AtomRec rj = rec[j];
    const bool molok = ALLPAIRS ? ((mi != mol[j]) || mi == 0) : !(e & POLAR_NL_SAMEMOL);
    double dx, dy, dz;
    pair_del<ALLPAIRS>(box, ri.x, ri.y, ri.z, rj.x, rj.y, rj.z, dx, dy, dz);
    const double xsq = dx * dx, ysq = dy * dy, zsq = dz * dz;
    const double rsq = xsq + ysq + zsq;
    const double rinv = rsqrt(rsq);
    const double r2inv = rinv * rinv;
    const double r = rsq * rinv;
    const double r3inv = r2inv * rinv;
    double px = 0, py = 0, pz = 0;
    if (rsq < cut_coulsq && molok) {  // note <, PS.cpp:454
      // shifted-force charge-dipole tensor G_pq = delta_pq (r^-2 + f_shift) r^2 ... written as the
      // reference does: M_pp = (-2 p^2 + q^2 + s^2) r2inv + f_shift (q^2 + s^2), M_pq = -pq (3 r2inv + f_shift)
      const double mxx = (-2.0 * xsq + ysq + zsq) * r2inv + f_shift * (ysq + zsq);
      const double myy = (-2.0 * ysq + xsq + zsq) * r2inv + f_shift * (xsq + zsq);
      const double mzz = (-2.0 * zsq + xsq + ysq) * r2inv + f_shift * (xsq + ysq);
      const double k = -(3.0 * r2inv + f_shift);
      const double mxy = k * dx * dy, mxz = k * dx * dz, myz = k * dy * dz;
      const double ef_temp = (r2inv + f_shift) * rinv * e2s;
      if (ri.a != 0.0 && rj.q != 0.0) {  // dipole on i, charge on j
        const double cf = rj.q * e2s * r3inv;
        px += cf * (ri.mx * mxx + ri.my * mxy + ri.mz * mxz);
        py += cf * (ri.mx * mxy + ri.my * myy + ri.mz * myz);
        pz += cf * (ri.mx * mxz + ri.my * myz + ri.mz * mzz);
        if (EFLAG) uef -= ef_temp * rj.q * (ri.mx * dx + ri.my * dy + ri.mz * dz);
      }
      if (rj.a != 0.0 && ri.q != 0.0) {  // dipole on j, charge on i
        const double cf = ri.q * e2s * r3inv;
        px -= cf * (rj.mx * mxx + rj.my * mxy + rj.mz * mxz);
        py -= cf * (rj.mx * mxy + rj.my * myy + rj.mz * myz);
        pz -= cf * (rj.mx * mxz + rj.my * myz + rj.mz * mzz);
        if (EFLAG) uef += ef_temp * ri.q * (rj.mx * dx + rj.my * dy + rj.mz * dz);
      }
    }
    if (ri.a != 0.0 && rj.a != 0.0 && (ALLPAIRS || rsq < ddcutsq)) {  // dipole-dipole, PS.cpp:512-602
      const double r5inv = r3inv * r2inv, r7inv = r5inv * r2inv;
      const double pdotp = ri.mx * rj.mx + ri.my * rj.my + ri.mz * rj.mz;
      const double pidotr = ri.mx * dx + ri.my * dy + ri.mz * dz;
      const double pjdotr = rj.mx * dx + rj.my * dy + rj.mz * dz;
      double pre_r, pre2, pre3;
      if (DAMP == 0) {
        const double t1 = exp(-pd * r);
        const double t2 = 1.0 + pd * r + 0.5 * pd * pd * r * r;
        const double t3 = t2 + (1.0 / 6.0) * pd * pd * pd * r * r * r;
        const double g2 = 1.0 - t1 * t2, g3 = 1.0 - t1 * t3;
        const double pre1 = 3.0 * r5inv * pdotp * g2 - 15.0 * r7inv * pidotr * pjdotr * g3;
        pre2 = 3.0 * r5inv * pjdotr * g3;
        pre3 = 3.0 * r5inv * pidotr * g3;
        const double pre4 = -pdotp * r3inv * (-t1 * (pd * rinv + pd * pd) + t1 * pd * t2 * rinv);
        const double pre5 = 3.0 * pidotr * pjdotr * r5inv *
                            (-t1 * (pd * rinv + pd * pd + 0.5 * r * pd * pd * pd) + t1 * pd * t3 * rinv);
        pre_r = pre1 + pre4 + pre5;
        if (EFLAG) udd += r3inv * pdotp * g2 - 3.0 * r5inv * pidotr * pjdotr * g3;
      } else {
        pre_r = 3.0 * r5inv * pdotp - 15.0 * r7inv * pidotr * pjdotr;
        pre2 = 3.0 * r5inv * pjdotr;
        pre3 = 3.0 * r5inv * pidotr;
        if (EFLAG) udd += r3inv * pdotp - 3.0 * r5inv * pidotr * pjdotr;
      }
      px += pre_r * dx + pre2 * ri.mx + pre3 * rj.mx;
      py += pre_r * dy + pre2 * ri.my + pre3 * rj.my;
      pz += pre_r * dz + pre2 * ri.mz + pre3 * rj.mz;
    }
    fx += px; fy += py; fz += pz;
    if (VPAIR) {  // ev_tally_xyz, src/pair.cpp:1001-1075 (each pair seen from both rows -> 0.5)
      v0 += 0.5 * dx * px; v1 += 0.5 * dy * py; v2 += 0.5 * dz * pz;
      v3 += 0.5 * dx * py; v4 += 0.5 * dx * pz; v5 += 0.5 * dy * pz;
    }
  }
  fx = wave_sum(fx); fy = wave_sum(fy); fz = wave_sum(fz);
  if (lane == 0) {
    const int o = perm ? perm[i] : i;  // forces leave in LAMMPS' order
    atomicAdd(&f[3 * o], fx); atomicAdd(&f[3 * o + 1], fy); atomicAdd(&f[3 * o + 2], fz);
  }
  if (EFLAG) {
    uef = wave_sum(uef); udd = wave_sum(udd);
    if (lane == 0) {
      if (ri.a != 0.0) atomicAdd(slot_ptr(slots, SL_USELF), 0.5 * (ri.mx * ri.mx + ri.my * ri.my + ri.mz * ri.mz) / ri.a);
      atomicAdd(slot_ptr(slots, SL_UEF), 0.5 * uef);
      atomicAdd(slot_ptr(slots, SL_UDD), 0.5 * udd);
    }
  }
  if (VPAIR) {
    v0 = wave_sum(v0); v1 = wave_sum(v1); v2 = wave_sum(v2); v3 = wave_sum(v3); v4 = wave_sum(v4); v5 = wave_sum(v5);
    if (lane == 0) {
      if (vglobal) {
        atomicAdd(slot_ptr(slots, SL_V0), v0); atomicAdd(slot_ptr(slots, SL_V1), v1); atomicAdd(slot_ptr(slots, SL_V2), v2);
        atomicAdd(slot_ptr(slots, SL_V3), v3); atomicAdd(slot_ptr(slots, SL_V4), v4); atomicAdd(slot_ptr(slots, SL_V5), v5);
      }
      if (vatom) {  // per-atom part of ev_tally_xyz, src/pair.cpp:1065-1082 (the row total is vatom[i])
        double *va = vatom + 6 * (size_t)(perm ? perm[i] : i);
        va[0] += v0; va[1] += v1; va[2] += v2; va[3] += v3; va[4] += v4; va[5] += v5;
      }
    }
  }
}

__global__ void k_add_into(long long n, const double *__restrict__ src, double *__restrict__ dst) {
  long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (i < n) dst[i] += src[i];
}

// a10  virial_fdotr_compute, src/pair.cpp:1495-1540: sum over locals AND ghosts of f_i x_i
__global__ __launch_bounds__(POLAR_BLOCK) void k_virial_fdotr(int nall, const double *__restrict__ x,
                                                              const double *__restrict__ f, double *__restrict__ slots) {
  double v[6] = {0, 0, 0, 0, 0, 0};
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nall; i += gridDim.x * blockDim.x) {
    const double fx = f[3 * i], fy = f[3 * i + 1], fz = f[3 * i + 2];
    const double xx = x[3 * i], yy = x[3 * i + 1], zz = x[3 * i + 2];
    v[0] += fx * xx; v[1] += fy * yy; v[2] += fz * zz; v[3] += fy * xx; v[4] += fz * xx; v[5] += fz * yy;
  }
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int k = 0; k < 6; k++) {
    double s = wave_sum(v[k]);
    if (lane == 0 && s != 0.0) atomicAdd(slot_ptr(slots, SL_V0 + k), s);
  }
}

// ------------------------------------------------------------------------------------------
// Cutoff-mode lists (extension): cell binning + CSR full lists over LOCAL atoms, minimum image.
struct CellGrid {
  int nc[3];
  double lo[3], inv[3];  // cell index = floor((x - lo) * inv) wrapped
};

__device__ __forceinline__ int cell_of(const CellGrid &g, const Box &b, double x, double y, double z) {
  int c[3];
  const double p[3] = {x, y, z};
#pragma unroll
  for (int k = 0; k < 3; k++) {
    double fr = (p[k] - g.lo[k]) / b.prd[k];
    fr -= floor(fr);
    int ck = (int)(fr * g.nc[k]);
    c[k] = ck >= g.nc[k] ? g.nc[k] - 1 : ck;
  }
  return (c[2] * g.nc[1] + c[1]) * g.nc[0] + c[0];
}

__global__ void k_cell_count(int n, const double *__restrict__ x, CellGrid g, Box b, int *__restrict__ cell_id,
                             int *__restrict__ cell_cnt) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int c = cell_of(g, b, x[3 * i], x[3 * i + 1], x[3 * i + 2]);
  cell_id[i] = c;
  atomicAdd(&cell_cnt[c], 1);
}

// single-workgroup exclusive scan (n up to a few million; run once per list build)
template <typename T>
__global__ __launch_bounds__(1024) void k_exclusive_scan(long long n, const T *__restrict__ in,
                                                         long long *__restrict__ out) {
  __shared__ long long part[1024];
  const int t = threadIdx.x;
  const long long chunk = (n + 1023) / 1024;
  const long long a = t * chunk, bnd = (a + chunk < n) ? a + chunk : n;
  long long s = 0;
  for (long long k = a; k < bnd; k++) s += (long long)in[k];
  part[t] = s;
  __syncthreads();
  if (t == 0) {
    long long run = 0;
    for (int k = 0; k < 1024; k++) { long long v = part[k]; part[k] = run; run += v; }
    out[n] = run;
  }
  __syncthreads();
  long long run = part[t];
  for (long long k = a; k < bnd; k++) { out[k] = run; run += (long long)in[k]; }
}

// counting-sort fill: perm[s] = orig index of the atom stored at sorted position s, inv = inverse.
// (Order inside a cell follows the atomics, i.e. it only permutes floating-point summation order.)
__global__ void k_cell_fill(int n, const int *__restrict__ cell_id, const long long *__restrict__ cell_first,
                            int *__restrict__ fill, int *__restrict__ perm, int *__restrict__ inv) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int c = cell_id[i];
  const int s = (int)cell_first[c] + atomicAdd(&fill[c], 1);
  perm[s] = i;
  inv[i] = s;
}
__global__ void k_map_rows(int n, const int *__restrict__ inv, const int *__restrict__ in, int *__restrict__ out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = inv[in[i]];
}
__global__ void k_map_range(int lo, int n, const int *__restrict__ inv, int *__restrict__ out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = inv[lo + i];
}

// One wave per atom row; lanes stride the atoms of the <=27 distinct neighbor cells (contiguous s
// ranges); ballot + popcount compacts in order.  Single pass into the pitched lists:
//   nl : every j with rsq <= cutallsq                      (static field, forces, rank metric)
//   dd : alpha_i != 0, alpha_j != 0 and rsq < ddcutsq      (the dipole sweep stream)
// cnt[] receives the TRUE counts; writes stop at the pitch and *overflow is raised.
__global__ __launch_bounds__(POLAR_BLOCK) void k_nl_build(const int *__restrict__ rows, int nrows,
                                                          const double4 *__restrict__ pos4, Box box, CellGrid g,
                                                          const long long *__restrict__ cell_first, double cutallsq,
                                                          double ddcutsq, long long nl_pitch, long long dd_pitch,
                                                          int *__restrict__ nl_cnt, int *__restrict__ dd_cnt,
                                                          int *__restrict__ nl_j, int *__restrict__ dd_j,
                                                          int *__restrict__ overflow,
                                                          unsigned long long *__restrict__ dd_total) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * POLAR_ROWS_PER_BLOCK + (threadIdx.x >> 6);
  if (row >= nrows) return;
  const int i = rows ? rows[row] : row;  // s space: the atoms of cell c are the indices [cell_first[c], cell_first[c+1])
  const double4 ri = pos4[i];            // {x, y, z, (molecule, polarizable)}
  const int imol = __double2hiint(ri.w), ipol = __double2loint(ri.w);
  const int ci = cell_of(g, box, ri.x, ri.y, ri.z);
  const int c0 = ci % g.nc[0], c1 = (ci / g.nc[0]) % g.nc[1], c2 = ci / (g.nc[0] * g.nc[1]);
  const long long nl0 = (long long)i * nl_pitch, dd0 = (long long)i * dd_pitch;
  int ncount = 0, dcount = 0;
  // Cells have an edge >= cutoff/2, so the stencil reaches +-2 cells (125 cells hold 42 % fewer
  // candidates than 27 cells of edge >= cutoff).  Cells are stored x-fastest, so the 5 cells of a
  // stencil row are ONE contiguous run of atoms (two runs when the row wraps around the box): the
  // lanes stride runs of ~100 atoms instead of single small cells.  Dimensions with fewer than 5
  // cells visit every cell exactly once.
  const unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  const int n0 = g.nc[0], n1 = g.nc[1], n2 = g.nc[2];
  const int zlo = n2 >= 5 ? c2 - 2 : 0, zhi = n2 >= 5 ? c2 + 2 : n2 - 1;
  const int ylo = n1 >= 5 ? c1 - 2 : 0, yhi = n1 >= 5 ? c1 + 2 : n1 - 1;
  const int xlo = n0 >= 5 ? c0 - 2 : 0, xhi = n0 >= 5 ? c0 + 2 : n0 - 1;
  for (int zz = zlo; zz <= zhi; zz++) {
    int b2 = zz;
    if (b2 < 0 || b2 >= n2) { if (!box.periodic[2]) continue; b2 = (b2 + n2) % n2; }
    for (int yy = ylo; yy <= yhi; yy++) {
      int b1 = yy;
      if (b1 < 0 || b1 >= n1) { if (!box.periodic[1]) continue; b1 = (b1 + n1) % n1; }
      const long long rowbase = ((long long)b2 * n1 + b1) * n0;
      // the x-run [xlo, xhi] as at most three pieces: below 0 (wrapped), inside, above n0-1 (wrapped)
      for (int piece = 0; piece < 3; piece++) {
        int xa, xb;
        if (piece == 0) { if (xlo >= 0) continue; if (!box.periodic[0]) continue; xa = xlo + n0; xb = n0 - 1; }
        else if (piece == 1) { xa = xlo < 0 ? 0 : xlo; xb = xhi >= n0 ? n0 - 1 : xhi; }
        else { if (xhi < n0) continue; if (!box.periodic[0]) continue; xa = 0; xb = xhi - n0; }
        const long long a = cell_first[rowbase + xa], b = cell_first[rowbase + xb + 1];
        for (long long base = a; base < b; base += 64) {
          const long long p = base + lane;
          bool in_nl = false, in_dd = false;
          const int j = (int)p;
          int same = 0;
          if (p < b && j != i) {
            const double4 rj = pos4[j];  // consecutive lanes read consecutive 32-byte entries
            double ex, ey, ez;
            min_image_rint(box, ri.x, ri.y, ri.z, rj.x, rj.y, rj.z, ex, ey, ez);
            const double rsq = ex * ex + ey * ey + ez * ez;
            in_nl = rsq <= cutallsq;
            in_dd = ipol && __double2loint(rj.w) && (rsq < ddcutsq);
            same = (imol != 0 && imol == __double2hiint(rj.w)) ? POLAR_NL_SAMEMOL : 0;
          }
          const unsigned long long m_nl = __ballot(in_nl), m_dd = __ballot(in_dd);
          const int kn = ncount + __popcll(m_nl & below), kd = dcount + __popcll(m_dd & below);
          // bit 30 of an nl entry: "same non-zero molecule" -- the static field and the charge-dipole terms
          // skip such pairs (PS.cpp:342,454), so those kernels need no molecule gather
          if (in_nl && kn < nl_pitch) nl_j[nl0 + kn] = j | same;
          if (in_dd && kd < dd_pitch) dd_j[dd0 + kd] = j;
          ncount += __popcll(m_nl);
          dcount += __popcll(m_dd);
        }
      }
    }
  }
  if (lane == 0) {
    nl_cnt[i] = ncount; dd_cnt[i] = dcount;
    if (ncount > nl_pitch || dcount > dd_pitch) atomicMax(overflow, ncount > dcount ? ncount : dcount);
    if (dcount) atomicAdd(dd_total + (blockIdx.x & 63) * 16, (unsigned long long)(dcount < dd_pitch ? dcount : (int)dd_pitch));
  }
}

// ------------------------------------------------------------------------------------------
// Device-side neighbor build for a3: what Neighbor hands this style --
// src/neighbor.cpp + src/npair_half_bin_newton.cpp, with NPair::exclusion() (molecule/intra) and
// NPair::find_special() (src/npair.cpp) -- as a FULL list over locals + ghosts for the local rows.
// Ghosts are explicit periodic images, so the grid is a plain (non-periodic) binning of the
// bounding box of all atoms.
struct LJGrid {
  int nc[3];
  double lo[3], inv[3];  // cell = clamp(floor((x - lo) * inv))
};
__device__ __forceinline__ int lj_cell_of(const LJGrid &g, double x, double y, double z) {
  const double p[3] = {x, y, z};
  int c[3];
#pragma unroll
  for (int k = 0; k < 3; k++) {
    int ck = (int)floor((p[k] - g.lo[k]) * g.inv[k]);
    c[k] = ck < 0 ? 0 : (ck >= g.nc[k] ? g.nc[k] - 1 : ck);
  }
  return (c[2] * g.nc[1] + c[1]) * g.nc[0] + c[0];
}
__global__ void k_lj_cell_count(int nall, const double *__restrict__ x, LJGrid g, int *__restrict__ cell_id,
                                int *__restrict__ cell_cnt) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nall) return;
  const int c = lj_cell_of(g, x[3 * i], x[3 * i + 1], x[3 * i + 2]);
  cell_id[i] = c;
  atomicAdd(&cell_cnt[c], 1);
}
// s order = cell order: pos[s] = {x, y, z, (type, molecule)}, aux[s] = {atom index, tag}
__global__ void k_lj_cell_fill(int nall, const int *__restrict__ cell_id, const long long *__restrict__ cell_first,
                               int *__restrict__ fill, const double *__restrict__ x, const int *__restrict__ type,
                               const int *__restrict__ mol, const int *__restrict__ tag, double4 *__restrict__ pos,
                               int2 *__restrict__ aux) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nall) return;
  const int c = cell_id[i];
  const int s = (int)cell_first[c] + atomicAdd(&fill[c], 1);
  pos[s] = make_double4(x[3 * i], x[3 * i + 1], x[3 * i + 2], __hiloint2double(mol[i], type[i]));
  aux[s] = make_int2(i, tag ? tag[i] : i + 1);
}

// One wave per local row i; lanes stride the atoms of the +-2 stencil of half-cutoff cells
// (contiguous x-runs), ballot + popcount compacts in order into the pitched row.
// Pair rules, in LAMMPS' order (npair_half_bin_newton.cpp):
//   rsq <= cutneighsq[itype][jtype]; exclusion: same molecule with molecule/intra;
//   special: which = find_special(special[i], nspecial[i], tag[j]) mapped through special_flag
//            (0: drop the pair, 1: keep plain, 2: keep with `which` in bits 30-31), except that a
//            pair farther apart than half a periodic box length is an image and kept plain
//            (Domain::minimum_image_check).
__global__ __launch_bounds__(POLAR_BLOCK) void k_lj_nl_build(int nlocal, int ntypes, const double *__restrict__ x,
                                                             const int *__restrict__ type, const int *__restrict__ mol,
                                                             const double4 *__restrict__ pos, const int2 *__restrict__ aux,
                                                             LJGrid g, const long long *__restrict__ cell_first,
                                                             const double *__restrict__ cutneighsq, Box box,
                                                             int exclude_intra, const int *__restrict__ nspecial,
                                                             const int *__restrict__ special, int maxspecial, int sf1,
                                                             int sf2, int sf3, long long pitch, int *__restrict__ cnt,
                                                             int *__restrict__ out_j, int *__restrict__ overflow,
                                                             unsigned long long *__restrict__ total) {
  extern __shared__ double cn_lds[];
  const int w = ntypes + 1;
  for (int t = threadIdx.x; t < w * w; t += blockDim.x) cn_lds[t] = cutneighsq[t];
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * POLAR_ROWS_PER_BLOCK + (threadIdx.x >> 6);
  if (i >= nlocal) return;
  const double xi = x[3 * i], yi = x[3 * i + 1], zi = x[3 * i + 2];
  const int itype = type[i], imol = mol[i];
  const double *cn = cn_lds + itype * w;
  int n1 = 0, n2 = 0, n3 = 0;
  if (nspecial) { n1 = nspecial[3 * i]; n2 = nspecial[3 * i + 1]; n3 = nspecial[3 * i + 2]; }
  const int *sp = special ? special + (size_t)i * maxspecial : nullptr;
  const int ci = lj_cell_of(g, xi, yi, zi);
  const int n0 = g.nc[0], n1c = g.nc[1], n2c = g.nc[2];
  const int c0 = ci % n0, c1 = (ci / n0) % n1c, c2 = ci / (n0 * n1c);
  const unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  const long long row0 = (long long)i * pitch;
  int count = 0;
  for (int zz = max(c2 - 2, 0); zz <= min(c2 + 2, n2c - 1); zz++)
    for (int yy = max(c1 - 2, 0); yy <= min(c1 + 2, n1c - 1); yy++) {
      const long long rb = ((long long)zz * n1c + yy) * n0;
      const long long a = cell_first[rb + max(c0 - 2, 0)], b = cell_first[rb + min(c0 + 2, n0 - 1) + 1];
      for (long long base = a; base < b; base += 64) {
        const long long p = base + lane;
        bool keep = false;
        int entry = 0;
        if (p < b) {
          const double4 pj = pos[p];
          const int2 aj = aux[p];
          const int jtype = __double2loint(pj.w), jmol = __double2hiint(pj.w);
          const double dx = xi - pj.x, dy = yi - pj.y, dz = zi - pj.z;
          const double rsq = dx * dx + dy * dy + dz * dz;
          keep = aj.x != i && rsq <= cn[jtype];
          if (keep && exclude_intra && imol == jmol) keep = false;
          entry = aj.x;
          if (keep && n3 > 0) {
            int which = 0;
            for (int k = 0; k < n3; k++)
              if (sp[k] == aj.y) {
                const int cls = k < n1 ? 1 : (k < n2 ? 2 : 3);
                const int flag = cls == 1 ? sf1 : (cls == 2 ? sf2 : sf3);
                which = flag == 0 ? -1 : (flag == 1 ? 0 : cls);
                break;
              }
            if (which > 0) {  // minimum_image_check: a partner more than half a box away is an image
              if ((box.periodic[0] && fabs(dx) > box.half[0]) || (box.periodic[1] && fabs(dy) > box.half[1]) ||
                  (box.periodic[2] && fabs(dz) > box.half[2]))
                which = 0;
            }
            if (which < 0) keep = false;
            else entry |= which << 30;
          }
        }
        const unsigned long long m = __ballot(keep);
        const int k = count + __popcll(m & below);
        if (keep && k < pitch) out_j[row0 + k] = entry;
        count += __popcll(m);
      }
    }
  if (lane == 0) {
    cnt[i] = count < pitch ? count : (int)pitch;
    if (count > pitch) atomicMax(overflow, count);
    atomicAdd(total + (blockIdx.x & 63) * 16, (unsigned long long)count);
  }
}
__global__ void k_lj_rows(int nlocal, long long pitch, int *__restrict__ ilist, long long *__restrict__ first) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nlocal) return;
  ilist[i] = i;
  first[i] = (long long)i * pitch;
}

// multi-GPU plumbing: dipoles of a contiguous row range <-> packed [n][3] buffers
__global__ void k_mu_gather(long long lo, long long hi, const int *__restrict__ inv, const Scal *scal,
                            const AtomRec *__restrict__ recA, const AtomRec *__restrict__ recB,
                            double *__restrict__ dst) {
  long long i = lo + blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (i >= hi) return;
  const long long s = inv ? inv[i] : i;
  const AtomRec *r = scal->cur ? recB : recA;
  dst[3 * (i - lo)] = r[s].mx; dst[3 * (i - lo) + 1] = r[s].my; dst[3 * (i - lo) + 2] = r[s].mz;
}
__global__ void k_mu_scatter(long long lo, long long hi, const int *__restrict__ inv, const Scal *scal,
                             AtomRec *__restrict__ recA, AtomRec *__restrict__ recB, const double *__restrict__ src) {
  long long i = lo + blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (i >= hi) return;
  const long long s = inv ? inv[i] : i;
  AtomRec *r = scal->cur ? recB : recA;
  r[s].mx = src[3 * (i - lo)]; r[s].my = src[3 * (i - lo) + 1]; r[s].mz = src[3 * (i - lo) + 2];
}

// halo exchange by index list (orig ids; negative entries are padding and skipped)
__global__ void k_mu_gather_idx(long long n, const int *__restrict__ idx, const int *__restrict__ inv, const Scal *scal,
                                const AtomRec *__restrict__ recA, const AtomRec *__restrict__ recB,
                                double *__restrict__ dst) {
  long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (t >= n) return;
  const int o = idx[t];
  if (o < 0) return;
  const AtomRec *r = scal->cur ? recB : recA;
  const int s = inv ? inv[o] : o;
  dst[3 * t] = r[s].mx; dst[3 * t + 1] = r[s].my; dst[3 * t + 2] = r[s].mz;
}
__global__ void k_mu_scatter_idx(long long n, const int *__restrict__ idx, const int *__restrict__ inv, const Scal *scal,
                                 AtomRec *__restrict__ recA, AtomRec *__restrict__ recB, const double *__restrict__ src,
                                 int own_lo, int own_hi) {
  long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (t >= n) return;
  const int o = idx[t];
  if (o < 0 || (o >= own_lo && o < own_hi)) return;  // padding, or a row this handle owns itself
  AtomRec *r = scal->cur ? recB : recA;
  const int s = inv ? inv[o] : o;
  r[s].mx = src[3 * t]; r[s].my = src[3 * t + 1]; r[s].mz = src[3 * t + 2];
}

// small utilities
__global__ void k_zero_scal(Scal *s, int keep_solver) {
  s->eng_vdwl = s->eng_coul = s->u_self = s->u_ef = s->u_dd = 0.0;
  for (int k = 0; k < 6; k++) s->virial[k] = 0.0;
  s->change = 0.0; s->last_change = 0.0; s->pad = 0;
  s->rmin_bits = (unsigned long long)__double_as_longlong(1000.0);
  if (!keep_solver) { s->iterations = 0; s->done = 0; s->status = 0; s->cur = 0; s->sweeps = 0; }
}
__global__ void k_set_done(Scal *s, int done) { s->done = done; }

}  // namespace polar
