// polar_tiles.hpp -- list mode, the TILE sweep (a6 + a7, PS.cpp:1158-1180 with the tensor of PS.cpp:1273-1306).
// Part of the hand-written HIP kernels (gfx950 / CDNA4, wave64) of the lj/cut/coul/long/polarization
// hot path; see polar_kernels.hpp for the mapping and the index spaces.
//
// Why.  The row sweep (k_field_lp) gathers one 64-byte record per pair from L2 and uses it once: it runs at the
// chip's L2 gather rate (about 15 TB/s) whatever else is done to it.  Here a WORKGROUP owns a TILE = the polarizable
// atoms of one half-cutoff cell (about 15 rows).  It copies the records every row of the tile can see -- the UNION
// list, about 2.5 single-row lists, each entry already shifted into the tile's periodic image -- into LDS once
// (coalesced: the union is made of runs of consecutive records), and every row of the tile then gathers its
// partners out of LDS by 16-bit positions.  Per computed pair the L2 sees 1/6 of a record instead of one, the index
// stream is 2 bytes instead of 4, and no pair needs a minimum-image wrap.
//
// Gauss-Seidel order.  Launches = tile colours: cells of one parity class (2 x 2 x 2; a third class per dimension
// holds the last cell when a periodic dimension has an odd cell count) are never adjacent, so rows of two tiles of one
// launch are at least a cell edge (>= cutoff / 2) apart.  Inside a tile the rows are coloured greedily ON THE DEVICE,
// every step (rows closer than the colour distance get different sub-phases), and the workgroup walks the sub-phases
// with a barrier in between, writing each new dipole into its LDS copy and into the record table.  No host-side
// colouring, no rank metric, nothing to re-validate on reneighbor steps.
#pragma once

#include "polar_common.hpp"
#include "polar_solver.hpp"
#include "polar_lists.hpp"

namespace polar {

// 48-byte sweep record: what a pair needs from its partner (position, dipole), in the piece order of AtomRec
struct __attribute__((aligned(16))) SRec {
  double x, mx, y, my, z, mz;
};

#define POLAR_TILE_MAXSUB 11   // sub-phases (in-tile colours) a tile header can describe
#define POLAR_TILE_MAXROWS 256 // polarizable atoms of one cell the builder can colour
#define POLAR_TILE_RECMASK 0x03FFFFFF  // union entry = record index | image code << 26 (code = (sx+1) + 3 (sy+1) + 9 (sz+1))

struct __attribute__((aligned(64))) TileHdr {
  int r0;     // first record of the cell (s space); local row m is record r0 + m
  int nrows;  // rows of this tile (polarizable atoms this handle owns); 0: nothing to do
  int U;      // entries of the union list = staged records (the dummy sits at position U)
  int nsub;   // sub-phases
  int sub_off[POLAR_TILE_MAXSUB + 1];  // rows of sub-phase p: trow[r0 + sub_off[p] .. r0 + sub_off[p + 1])
};

// tiles of one launch: cells c_k = start_k + stride_k * i_k, i_k < count_k
struct TileLaunch {
  int start[3], stride[3], count[3], nc[3];
};

// row lists: entry e of a row (trip e >> 6, lane e & 63) as a 16-bit LDS position, stored in chunks of 8 trips so that
// ONE 16-byte load per lane brings that lane's partners of eight trips (a typical row is 8-9 trips long)
__host__ __device__ __forceinline__ long long tile_slot16(long long e) {
  const long long t = e >> 6, lane = e & 63;
  return ((t >> 3) << 9) + (lane << 3) + (t & 7);
}

// periodic image shift of a union entry (whole lattice vectors a, b, c; domain.cpp:1258-1305 order)
__device__ __forceinline__ void tile_shift(const Box &b, int code, double &sx, double &sy, double &sz) {
  const int s0 = code % 3 - 1, s1 = (code / 3) % 3 - 1, s2 = code / 9 - 1;
  sx = s0 * b.prd[0] + s1 * b.xy + s2 * b.xz;
  sy = s1 * b.prd[1] + s2 * b.yz;
  sz = s2 * b.prd[2];
}

// AtomRec (both buffers hold the same initial dipoles) -> sweep records, and the solved dipoles back
__global__ void k_srec_pack(int n, const AtomRec *__restrict__ r, SRec *__restrict__ s0, SRec *__restrict__ s1) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const AtomRec a = r[i];
  SRec s;
  s.x = a.x; s.mx = a.mx; s.y = a.y; s.my = a.my; s.z = a.z; s.mz = a.mz;
  s0[i] = s;
  if (s1) s1[i] = s;
}
__global__ void k_srec_unpack(int n, const Scal *scal, const SRec *__restrict__ s0, const SRec *__restrict__ s1,
                              AtomRec *__restrict__ r0, AtomRec *__restrict__ r1) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const SRec s = scal->cur ? s1[i] : s0[i];
  AtomRec *r = scal->cur ? r1 : r0;
  r[i].mx = s.mx; r[i].my = s.my; r[i].mz = s.mz;
}

// ------------------------------------------------------------------------------------------
// Builder: one workgroup per cell.  Pass 1 lists the union (polarizable atoms of the +-2 stencil whose shifted
// position lies within the cutoff of the bounding box of the tile's rows; two sweeps over the stencil -- count, then
// fill at prefix offsets -- keep the order deterministic), pass 2 gives every row its partners as positions in that
// list, pass 3 colours the rows of the tile, pass 4 writes the row table in sub-phase order.
//   flags[5] union entries needed (+1 for the dummy) when un_pitch is too small     flags[6] the same for a row list
//   flags[7] a tile the builder cannot describe (more than MAXROWS atoms or MAXSUB sub-phases)   flags[9] largest U
__global__ __launch_bounds__(256) void k_tile_build(CellGrid g, Box box, const double4 *__restrict__ pos4,
                                                    const long long *__restrict__ cell_first,
                                                    const int *__restrict__ npol, const int *__restrict__ perm,
                                                    int own_lo, int own_hi, double ddcutsq, double colordistsq,
                                                    int un_pitch, int *__restrict__ un_j, long long pitch16,
                                                    unsigned short *__restrict__ dd16, TileHdr *__restrict__ hdr,
                                                    int2 *__restrict__ trow, int *__restrict__ flags,
                                                    unsigned long long *__restrict__ dd_total) {
  extern __shared__ __attribute__((aligned(16))) char tb_lds[];
  double *ux = reinterpret_cast<double *>(tb_lds), *uy = ux + un_pitch, *uz = uy + un_pitch;
  int *cnt = reinterpret_cast<int *>(uz + un_pitch);  // [128] entries per stencil cell
  int *off = cnt + 128;                               // [128] their prefix
  int *rowT = off + 128;                              // [MAXROWS] trips of local row m, -1: not a row of this handle
  int *col = rowT + POLAR_TILE_MAXROWS;               // [MAXROWS] sub-phase
  double *bb = reinterpret_cast<double *>(col + POLAR_TILE_MAXROWS);  // [6] bounding box of the rows
  int *misc = reinterpret_cast<int *>(bb + 6);        // [4] U, rows
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, nwv = blockDim.x >> 6;
  const int c = blockIdx.x;
  const int n0 = g.nc[0], n1 = g.nc[1], n2 = g.nc[2];
  const int c0 = c % n0, c1 = (c / n0) % n1, c2 = c / (n0 * n1);
  const int r0 = (int)cell_first[c], P = npol[c];
  TileHdr *H = hdr + c;
  if (P > POLAR_TILE_MAXROWS) {
    if (tid == 0) { atomicMax(flags + 7, P); H->r0 = r0; H->nrows = 0; H->U = 0; H->nsub = 0; }
    return;
  }
  for (int m = tid; m < POLAR_TILE_MAXROWS; m += blockDim.x) {
    bool row = false;
    if (m < P) { const int o = perm[r0 + m]; row = o >= own_lo && o < own_hi; }
    rowT[m] = row ? 0 : -1;
    col[m] = -1;
  }
  __syncthreads();
  if (wv == 0) {  // bounding box of the rows
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    int rows = 0;
    for (int mb = 0; mb < P; mb += 64) {
      const int m = mb + lane;
      const bool is = m < P && rowT[m] >= 0;
      if (is) {
        const double4 p = pos4[r0 + m];
        lo[0] = fmin(lo[0], p.x); lo[1] = fmin(lo[1], p.y); lo[2] = fmin(lo[2], p.z);
        hi[0] = fmax(hi[0], p.x); hi[1] = fmax(hi[1], p.y); hi[2] = fmax(hi[2], p.z);
      }
      rows += __popcll(__ballot(is));
    }
#pragma unroll
    for (int k = 0; k < 3; k++) { lo[k] = wave_min(lo[k]); hi[k] = -wave_min(-hi[k]); }
    if (lane == 0) {
      for (int k = 0; k < 3; k++) { bb[k] = lo[k]; bb[3 + k] = hi[k]; }
      misc[1] = rows;
    }
  }
  __syncthreads();
  if (misc[1] == 0) {  // no row of this handle in the cell
    if (tid == 0) { H->r0 = r0; H->nrows = 0; H->U = 0; H->nsub = 0; }
    return;
  }
  const double b0 = bb[0], b1 = bb[1], b2 = bb[2], b3 = bb[3], b4 = bb[4], b5 = bb[5];
  const unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  // the two sweeps over the 125 stencil cells (FILL = 0: count, 1: write)
  auto stencil_pass = [&](const int fill) {
    for (int q = wv; q < 125; q += nwv) {
      int bq[3] = {c0 + q % 5 - 2, c1 + (q / 5) % 5 - 2, c2 + q / 25 - 2};
      int sh[3] = {0, 0, 0};
      const int nn[3] = {n0, n1, n2};
      bool ok = true;
#pragma unroll
      for (int k = 0; k < 3; k++) {
        if (bq[k] < 0) { ok = ok && box.periodic[k]; bq[k] += nn[k]; sh[k] = -1; }
        else if (bq[k] >= nn[k]) { ok = ok && box.periodic[k]; bq[k] -= nn[k]; sh[k] = 1; }
        ok = ok && bq[k] >= 0 && bq[k] < nn[k];  // (a dimension with fewer than 3 cells cannot be periodic in list mode)
      }
      int count = 0;
      if (ok) {
        const int cj = (bq[2] * n1 + bq[1]) * n0 + bq[0];
        const int a = (int)cell_first[cj], np = npol[cj];
        const double sx = sh[0] * box.prd[0] + sh[1] * box.xy + sh[2] * box.xz, sy = sh[1] * box.prd[1] + sh[2] * box.yz,
                     sz = sh[2] * box.prd[2];
        const int code = (sh[0] + 1) + 3 * (sh[1] + 1) + 9 * (sh[2] + 1);
        const int base0 = fill ? off[q] : 0;
        for (int b = 0; b < np; b += 64) {
          const int j = a + b + lane;
          bool in = false;
          double px = 0, py = 0, pz = 0;
          if (b + lane < np) {
            const double4 p = pos4[j];
            px = p.x + sx; py = p.y + sy; pz = p.z + sz;
            const double ex = fmax(fmax(b0 - px, px - b3), 0.0), ey = fmax(fmax(b1 - py, py - b4), 0.0),
                         ez = fmax(fmax(b2 - pz, pz - b5), 0.0);
            in = q == 62 || (ex * ex + ey * ey + ez * ez) < ddcutsq;  // the home cell is listed whole: row m sits at off[62] + m
          }
          const unsigned long long mk = __ballot(in);
          if (fill && in) {
            const int k = base0 + count + __popcll(mk & below);
            ux[k] = px; uy[k] = py; uz[k] = pz;
            un_j[(size_t)c * un_pitch + k] = j | (code << 26);
          }
          count += __popcll(mk);
        }
      }
      if (!fill && lane == 0) cnt[q] = count;
    }
  };
  stencil_pass(0);
  __syncthreads();
  if (wv == 0) {  // exclusive prefix of the 125 counts
    const int v0 = cnt[lane], v1 = lane + 64 < 125 ? cnt[lane + 64] : 0;
    int i0 = v0, i1 = v1;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int u0 = __shfl_up(i0, o, 64), u1 = __shfl_up(i1, o, 64);
      if (lane >= o) { i0 += u0; i1 += u1; }
    }
    const int t0 = __shfl(i0, 63, 64), t1 = __shfl(i1, 63, 64);
    off[lane] = i0 - v0;
    off[64 + lane] = t0 + i1 - v1;
    if (lane == 0) misc[0] = t0 + t1;
  }
  __syncthreads();
  const int U = misc[0];
  if (U + 1 > un_pitch) {
    if (tid == 0) { atomicMax(flags + 5, U + 1); H->r0 = r0; H->nrows = 0; H->U = 0; H->nsub = 0; }
    return;
  }
  stencil_pass(1);
  __syncthreads();
  const int selfbase = off[62];
  // pass 2: the rows' partner lists
  for (int m = wv; m < P; m += nwv) {
    if (rowT[m] < 0) continue;
    const int i = r0 + m, self = selfbase + m;
    const double xi = ux[self], yi = uy[self], zi = uz[self];
    unsigned short *row = dd16 + (size_t)i * pitch16;
    int count = 0;
    for (int b = 0; b < U; b += 64) {
      const int e = b + lane;
      bool in = false;
      if (e < U && e != self) {
        const double dx = xi - ux[e], dy = yi - uy[e], dz = zi - uz[e];
        in = (dx * dx + dy * dy + dz * dz) < ddcutsq;
      }
      const unsigned long long mk = __ballot(in);
      const int k = count + __popcll(mk & below);
      if (in && k < pitch16) row[tile_slot16(k)] = (unsigned short)e;
      count += __popcll(mk);
    }
    const int have = count < pitch16 ? count : (int)pitch16;
    const int padded = (have + 63) & ~63;
    for (int k = have + lane; k < padded; k += 64) row[tile_slot16(k)] = (unsigned short)U;  // the dummy: zero dipole
    if (lane == 0) {
      rowT[m] = padded >> 6;
      if (count > pitch16) atomicMax(flags + 6, count);
      if (have) atomicAdd(dd_total + (c & 63) * 16, (unsigned long long)have);
    }
  }
  __syncthreads();
  if (wv != 0) return;
  // pass 3: greedy colouring of the rows in cell order -- a row takes the lowest sub-phase no earlier row within the colour
  // distance holds (atoms that close must not be relaxed Jacobi-fashion against each other)
  int nsub = 0;
  for (int m = 0; m < P; m++) {
    if (rowT[m] < 0) continue;
    const int self = selfbase + m;
    const double xm = ux[self], ym = uy[self], zm = uz[self];
    unsigned used = 0u;
    for (int jb = 0; jb < m; jb += 64) {
      const int j = jb + lane;
      if (j < m && rowT[j] >= 0) {
        const double dx = xm - ux[selfbase + j], dy = ym - uy[selfbase + j], dz = zm - uz[selfbase + j];
        if ((dx * dx + dy * dy + dz * dz) < colordistsq) used |= 1u << col[j];
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) used |= (unsigned)__shfl_xor((int)used, o, 64);
    int cm = __ffs((int)~used) - 1;
    if (cm >= POLAR_TILE_MAXSUB) { if (lane == 0) atomicMax(flags + 7, cm + 1); cm = POLAR_TILE_MAXSUB - 1; }
    if (lane == 0) col[m] = cm;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    nsub = cm + 1 > nsub ? cm + 1 : nsub;
  }
  // pass 4: row table in sub-phase order
  int k = 0;
  for (int s = 0; s < nsub; s++) {
    if (lane == 0) H->sub_off[s] = k;
    for (int mb = 0; mb < P; mb += 64) {
      const int m = mb + lane;
      const bool is = m < P && rowT[m] >= 0 && col[m] == s;
      const unsigned long long mk = __ballot(is);
      if (is) trow[r0 + k + __popcll(mk & below)] = make_int2(m | (rowT[m] << 16), selfbase + m);
      k += __popcll(mk);
    }
  }
  if (lane == 0) {
    H->sub_off[nsub] = k;
    H->r0 = r0; H->nrows = k; H->U = U; H->nsub = nsub;
    atomicMax(flags + 9, U);
  }
}

// ------------------------------------------------------------------------------------------
// The sweep.  One workgroup per tile: stage the union records (shifted) into LDS, then the sub-phases.
//   EP_INPLACE  Gauss-Seidel: a row's new dipole goes into the tile's LDS copy and into the record table
//   EP_JACOBI   all rows of the tile against the staged (old) dipoles, results into the other record table
// DET (Gauss-Seidel, `deterministic yes`): the rows of a sub-phase commit their dipoles together after a barrier, and
// the record table is written through a pending array that k_tile_commit folds in after the launch -- no row ever reads
// a dipole that another wave of the same launch may or may not have written yet.
template <int DAMP>
__device__ __forceinline__ void tile_pair(double xi, double yi, double zi, const double2 &A, const double2 &B, const double2 &C,
                                          double pd, const ExpCoef &K, double &ax, double &ay, double &az) {
  const double dx = xi - A.x, dy = yi - B.x, dz = zi - C.x;
  const double r2 = fmax(fma(dx, dx, fma(dy, dy, dz * dz)), 1e-12);  // the dummy may coincide with the row atom
  double s3, s5;
  tensor_scalars_lp<DAMP>(r2, pd, K, s3, s5);
  const double dot = fma(A.y, dx, fma(B.y, dy, C.y * dz));
  const double cc = s5 * dot;
  ax = fma(cc, dx, fma(-s3, A.y, ax));
  ay = fma(cc, dy, fma(-s3, B.y, ay));
  az = fma(cc, dz, fma(-s3, C.y, az));
}
__device__ __forceinline__ void tile_read(const char *lds, unsigned pos, double2 &A, double2 &B, double2 &C) {
  const char *p = lds + pos * 48u;
  A = *reinterpret_cast<const double2 *>(p);
  B = *reinterpret_cast<const double2 *>(p + 16);
  C = *reinterpret_cast<const double2 *>(p + 32);
}

template <int EP, int DAMP, bool DET>
__global__ __launch_bounds__(256) void k_field_tile(TileLaunch L, const TileHdr *__restrict__ hdr, const int2 *__restrict__ trow,
                                                    const int *__restrict__ un_j, int un_pitch,
                                                    const unsigned short *__restrict__ dd16, long long pitch16, SRec *s0,
                                                    SRec *s1, double *pend, const AtomRec *__restrict__ rec,
                                                    const double *__restrict__ ef, Box box, double pd, ExpCoef K,
                                                    const Scal *scal, double *__restrict__ slots) {
  extern __shared__ __attribute__((aligned(16))) char tl_lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), nwv = blockDim.x >> 6;
  const int ntile = L.count[0] * L.count[1] * L.count[2];
  const int lb = xcd_block(blockIdx.x, ntile);
  if (lb < 0) return;
  const int i0 = lb % L.count[0], i1 = (lb / L.count[0]) % L.count[1], i2 = lb / (L.count[0] * L.count[1]);
  const int c = ((L.start[2] + L.stride[2] * i2) * L.nc[1] + (L.start[1] + L.stride[1] * i1)) * L.nc[0] + L.start[0] + L.stride[0] * i0;
  const TileHdr *H = hdr + c;
  const int nrows = H->nrows;
  if (nrows == 0) return;
  if (scal->done) return;
  const int cur = EP == EP_JACOBI ? scal->cur : 0;
  const SRec *src = (EP == EP_JACOBI && cur) ? s1 : s0;
  SRec *dst = (EP == EP_JACOBI) ? (cur ? s0 : s1) : s0;
  const int U = H->U, r0 = H->r0;
  // ---- stage the union: piece g = 3 e + p of entry e, 16 bytes each, lane-linear in LDS ----
  {
    const int *uj = un_j + (size_t)c * un_pitch;
    const char *sb = reinterpret_cast<const char *>(src);
    const int np = 3 * U, step = blockDim.x;
    for (int g0 = tid; g0 < np; g0 += 4 * step) {
      int ent[4];
      double2 v[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int gq = g0 + u * step;
        ent[u] = gq < np ? uj[gq / 3] : 0;
      }
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int gq = g0 + u * step, p = gq % 3;
        v[u] = make_double2(0.0, 0.0);
        if (gq < np) v[u] = *reinterpret_cast<const double2 *>(sb + (size_t)(ent[u] & POLAR_TILE_RECMASK) * 48 + p * 16);
      }
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int gq = g0 + u * step, p = gq % 3, code = ent[u] >> 26;
        if (gq < np) {
          if (code != 13) {
            double sx, sy, sz;
            tile_shift(box, code, sx, sy, sz);
            v[u].x += p == 0 ? sx : (p == 1 ? sy : sz);
          }
          *reinterpret_cast<double2 *>(tl_lds + (size_t)gq * 16) = v[u];
        }
      }
    }
    if (tid < 3) *reinterpret_cast<double2 *>(tl_lds + (size_t)U * 48 + tid * 16) = make_double2(0.0, 0.0);  // the dummy
  }
  __syncthreads();
  double chg = 0.0;
  const int nsub = EP == EP_JACOBI ? 1 : H->nsub;
  for (int sp = 0; sp < nsub; sp++) {
    const int kb = EP == EP_JACOBI ? 0 : H->sub_off[sp], ke = EP == EP_JACOBI ? nrows : H->sub_off[sp + 1];
    for (int k = kb + wv; k < ke; k += nwv) {
      const int2 tr = trow[r0 + k];
      const int m = __builtin_amdgcn_readfirstlane(tr.x & 0xFFFF), T = __builtin_amdgcn_readfirstlane(tr.x >> 16);
      const unsigned self = (unsigned)__builtin_amdgcn_readfirstlane(tr.y);
      const int i = r0 + m;
      const uint4 *pc = reinterpret_cast<const uint4 *>(dd16 + (size_t)i * pitch16) + lane;
      uint4 Ja = pc[0];
      // the row atom: position out of the staged copy (same in every lane -> scalar registers), epilogue inputs in lanes 0..2
      const char *sp_ = tl_lds + self * 48u;
      const double xi = wave_uniform(*reinterpret_cast<const double *>(sp_)),
                   yi = wave_uniform(*reinterpret_cast<const double *>(sp_ + 16)),
                   zi = wave_uniform(*reinterpret_cast<const double *>(sp_ + 32));
      double mu_old = 0.0, alpha = 0.0, efk = 0.0;
      if (lane < 3) {
        mu_old = *reinterpret_cast<const double *>(sp_ + lane * 16 + 8);
        alpha = rec[i].a;
        efk = ef[3 * (size_t)i + lane];
      }
      double ax = 0.0, ay = 0.0, az = 0.0;
      if (T > 0) {
        double2 A, B, C, An, Bn, Cn;
        tile_read(tl_lds, Ja.x & 0xFFFFu, A, B, C);
        const int NC = (T + 7) >> 3;
#define POLAR_TILE_TRIP(UU, NEXTPOS)                                        \
  {                                                                        \
    const bool more = t0 + (UU) + 1 < T;                                   \
    if (more) tile_read(tl_lds, (NEXTPOS), An, Bn, Cn);                    \
    tile_pair<DAMP>(xi, yi, zi, A, B, C, pd, K, ax, ay, az);               \
    if (!more) break;                                                      \
    A = An; B = Bn; C = Cn;                                                \
  }
        for (int cc = 0; cc < NC; cc++) {
          const int t0 = 8 * cc;
          uint4 Jn = Ja;
          if (cc + 1 < NC) Jn = pc[64 * (cc + 1)];
          POLAR_TILE_TRIP(0, Ja.x >> 16)
          POLAR_TILE_TRIP(1, Ja.y & 0xFFFFu)
          POLAR_TILE_TRIP(2, Ja.y >> 16)
          POLAR_TILE_TRIP(3, Ja.z & 0xFFFFu)
          POLAR_TILE_TRIP(4, Ja.z >> 16)
          POLAR_TILE_TRIP(5, Ja.w & 0xFFFFu)
          POLAR_TILE_TRIP(6, Ja.w >> 16)
          POLAR_TILE_TRIP(7, Jn.x & 0xFFFFu)
          Ja = Jn;
        }
#undef POLAR_TILE_TRIP
      }
      // the three wave sums in one butterfly (lp_finish): lanes 0, 1, 2 end with E_x, E_y, E_z of the row
      const double v = cl_reduce3(ax, ay, az, lane);
      if (lane < 3) {
        const double mu_new = alpha * (efk + v);  // PS.cpp:1170-1180
        const double d = mu_new - mu_old;
        chg = fma(d, d, chg);
        if (DET) {
          pend[3 * (size_t)i + lane] = mu_new;  // committed below (LDS copy) and by k_tile_commit (record table)
        } else {
          reinterpret_cast<double *>(dst + i)[2 * lane + 1] = mu_new;
          if (EP != EP_JACOBI) *reinterpret_cast<double *>(tl_lds + self * 48u + lane * 16 + 8) = mu_new;
        }
      }
    }
    if (EP != EP_JACOBI) {
      if (DET) {
        __syncthreads();  // every row of the sub-phase has read what it needed
        for (int k = kb + wv; k < ke; k += nwv) {  // this wave's rows again: their pending dipoles into the LDS copy
          const int2 tr = trow[r0 + k];
          if (lane < 3)
            *reinterpret_cast<double *>(tl_lds + (unsigned)tr.y * 48u + lane * 16 + 8) = pend[3 * (size_t)(r0 + (tr.x & 0xFFFF)) + lane];
        }
      }
      if (sp + 1 < nsub) __syncthreads();
    }
  }
  chg += dpp_full<0xB1>(chg);
  chg += dpp_full<0x4E>(chg);
  if (lane == 0 && chg != 0.0) atomicAdd(slot_ptr(slots, SL_CHANGE), chg);
}

// DET: fold the pending dipoles of a launch's rows into the record table (the launch itself only read the table)
__global__ void k_tile_commit(TileLaunch L, const TileHdr *__restrict__ hdr, const int2 *__restrict__ trow,
                              const double *__restrict__ pend, SRec *s0, const Scal *scal) {
  if (scal->done) return;
  const int ntile = L.count[0] * L.count[1] * L.count[2];
  const int lb = blockIdx.x;
  if (lb >= ntile) return;
  const int i0 = lb % L.count[0], i1 = (lb / L.count[0]) % L.count[1], i2 = lb / (L.count[0] * L.count[1]);
  const int c = ((L.start[2] + L.stride[2] * i2) * L.nc[1] + (L.start[1] + L.stride[1] * i1)) * L.nc[0] + L.start[0] + L.stride[0] * i0;
  const TileHdr *H = hdr + c;
  const int nrows = H->nrows, r0 = H->r0;
  for (int t = threadIdx.x; t < 3 * nrows; t += blockDim.x) {
    const int k = t / 3, comp = t - 3 * k;
    const int i = r0 + (trow[r0 + k].x & 0xFFFF);
    reinterpret_cast<double *>(s0 + i)[2 * comp + 1] = pend[3 * (size_t)i + comp];
  }
}

}  // namespace polar
