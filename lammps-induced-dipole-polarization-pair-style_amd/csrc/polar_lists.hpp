// polar_lists.hpp -- list mode: cell binning, pitched neighbor lists, device-side neighbor build for a3, dipole exchange and small utilities.
// Part of the hand-written HIP kernels (gfx950 / CDNA4, wave64) of the lj/cut/coul/long/polarization
// hot path; see polar_kernels.hpp for the mapping and the index spaces.
#pragma once

#include "polar_common.hpp"

namespace polar {

// ------------------------------------------------------------------------------------------
// Cutoff-mode lists (extension): cell binning + CSR full lists over LOCAL atoms, minimum image.
struct CellGrid {
  int trim;  // 1: per-atom trimming of the stencil in k_nl_build (POLAR_NL_TRIM)
  int nc[3];
  double lo[3], inv[3];  // cell index = floor((x - lo) * inv) wrapped
};

__device__ __forceinline__ int cell_of(const CellGrid &g, const Box &b, double x, double y, double z) {
  int c[3];
  double fr3[3];
  frac_coords(b, g.lo, x, y, z, fr3);
#pragma unroll
  for (int k = 0; k < 3; k++) {
    double fr = fr3[k];
    fr -= floor(fr);
    int ck = (int)(fr * g.nc[k]);
    c[k] = ck >= g.nc[k] ? g.nc[k] - 1 : ck;
  }
  return (c[2] * g.nc[1] + c[1]) * g.nc[0] + c[0];
}

static __global__ void k_cell_count(int n, const double *__restrict__ x, CellGrid g, Box b, int *__restrict__ cell_id,
                             int *__restrict__ cell_cnt) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int c = cell_of(g, b, x[3 * i], x[3 * i + 1], x[3 * i + 2]);
  cell_id[i] = c;
  atomicAdd(&cell_cnt[c], 1);
}

// Exclusive scan of n ints into n + 1 long longs (out[n] = the total), three small launches: every workgroup scans 4,096
// consecutive entries (four per thread: one 16-byte load each, coalesced) and leaves its total, one workgroup scans the totals,
// the offsets are added.  (Rounds 1-3 used ONE workgroup whose threads each walked n / 1024 consecutive entries: 7 us for the
// 6,400 cells of the headline box, 32 us for the 25,000 cells of configs[4], 1.03 ms for the 259,306 rows of an uploaded list's
// symmetrisation.)  launch_scan() below; `tot` = scratch for the workgroup totals (one per stream that scans).
#define POLAR_SCAN_ITEMS 4096
static __global__ __launch_bounds__(1024) void k_scan_block(long long n, const int *__restrict__ in, long long *__restrict__ out,
                                                     long long *__restrict__ btot) {
  __shared__ long long wtot[16];
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const long long base = (long long)blockIdx.x * POLAR_SCAN_ITEMS + 4 * t;
  int v[4] = {0, 0, 0, 0};
  if (base + 3 < n) { const int4 q = *(const int4 *)(in + base); v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w; }
  else for (int k = 0; k < 4; k++) if (base + k < n) v[k] = in[base + k];
  const long long s = (long long)v[0] + v[1] + v[2] + v[3];
  long long inc = s;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const long long up = __shfl_up(inc, off, 64);
    if (lane >= off) inc += up;
  }
  if (lane == 63) wtot[wv] = inc;
  __syncthreads();
  if (wv == 0) {
    long long x = lane < 16 ? wtot[lane] : 0, w = x;
#pragma unroll
    for (int off = 1; off < 16; off <<= 1) {
      const long long up = __shfl_up(w, off, 64);
      if (lane >= off) w += up;
    }
    if (lane < 16) wtot[lane] = w - x;  // exclusive offset of each wave
    if (lane == 15) { btot[blockIdx.x] = w; if (gridDim.x == 1) out[n] = w; }
  }
  __syncthreads();
  long long run = wtot[wv] + inc - s;
#pragma unroll
  for (int k = 0; k < 4; k++) { if (base + k < n) out[base + k] = run; run += v[k]; }
}
// the workgroup totals, in place: btot[b] <- sum of the totals before b; out[n] <- the grand total
static __global__ __launch_bounds__(1024) void k_scan_totals(int nb, long long *__restrict__ btot, long long *__restrict__ total) {
  __shared__ long long wtot[16];
  __shared__ long long carry;
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  if (t == 0) carry = 0;
  __syncthreads();
  for (int b0 = 0; b0 < nb; b0 += 1024) {   // (one trip up to 4 M entries)
    const long long s = b0 + t < nb ? btot[b0 + t] : 0;
    long long inc = s;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const long long up = __shfl_up(inc, off, 64);
      if (lane >= off) inc += up;
    }
    if (lane == 63) wtot[wv] = inc;
    __syncthreads();
    if (wv == 0) {
      long long x = lane < 16 ? wtot[lane] : 0, w = x;
#pragma unroll
      for (int off = 1; off < 16; off <<= 1) {
        const long long up = __shfl_up(w, off, 64);
        if (lane >= off) w += up;
      }
      if (lane < 16) wtot[lane] = w - x;
    }
    __syncthreads();
    const long long mine = carry + wtot[wv] + inc - s;
    if (b0 + t < nb) btot[b0 + t] = mine;
    __syncthreads();
    if (t == 1023) carry = mine + s;
    __syncthreads();
  }
  if (t == 0) *total = carry;
}
static __global__ __launch_bounds__(256) void k_scan_add(long long n, long long *__restrict__ out, const long long *__restrict__ boff) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x + POLAR_SCAN_ITEMS;   // (the first workgroup's offset is zero)
  if (i < n) out[i] += boff[i / POLAR_SCAN_ITEMS];
}

// counting-sort fill: perm[s] = orig index of the atom stored at sorted position s, inv = inverse.
// (Order inside a cell follows the atomics, i.e. it only permutes floating-point summation order.)
// Inside a cell the polarizable atoms come first (filled from the front), the others last (from the back): the dipole lists
// then hold long runs of CONSECUTIVE records -- gathers of neighbouring records merge into whole 128-byte lines and run
// 20 % faster than scattered ones (tools/calib_gather48.hip: 246 against 203 G records/s from an L2-resident table).
static __global__ void k_cell_fill(int n, const int *__restrict__ cell_id, const long long *__restrict__ cell_first,
                            int *__restrict__ fill, int *__restrict__ fill_back, const double *__restrict__ alpha,
                            int *__restrict__ perm, int *__restrict__ inv) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int c = cell_id[i];
  int s;
  if (!fill_back || alpha[i] != 0.0) s = (int)cell_first[c] + atomicAdd(&fill[c], 1);
  else s = (int)cell_first[c + 1] - 1 - atomicAdd(&fill_back[c], 1);
  perm[s] = i;
  inv[i] = s;
}
// k_cell_fill places the atoms of a cell in the order its atomics happened to run; here one wave per cell puts each of the
// two groups (polarizable atoms, the others) into ascending atom index, so that every list -- and with it every
// floating-point sum and the device colouring's tie-breaks -- has the same order run after run.  A lane holds one atom and
// counts the smaller indices of its group (rank sort, ~20 entries); groups of more than 64 fall to a serial insertion sort.
static __global__ __launch_bounds__(256) void k_cell_sort(long long ncell, const long long *__restrict__ cell_first, const int *__restrict__ npol,
                                                   int *__restrict__ perm, int *__restrict__ inv) {
  const int lane = threadIdx.x & 63;
  const long long c = blockIdx.x * (long long)(blockDim.x >> 6) + (threadIdx.x >> 6);
  if (c >= ncell) return;
  const int a = (int)cell_first[c], b = (int)cell_first[c + 1], m = npol ? a + npol[c] : b;
  for (int part = 0; part < 2; part++) {
    const int lo = part ? m : a, hi = part ? b : m;
    const int cnt = hi - lo;
    if (cnt <= 1) continue;
    if (cnt <= 64) {
      const int v = lane < cnt ? perm[lo + lane] : 0x7fffffff;
      int r = 0;
      for (int k = 0; k < cnt; k++) r += __shfl(v, k, 64) < v;
      __builtin_amdgcn_wave_barrier();   // all reads of the group above, all writes below
      if (lane < cnt) { perm[lo + r] = v; inv[v] = lo + r; }  // (atom indices are distinct: the ranks are a permutation)
    } else if (lane == 0) {
      for (int k = lo + 1; k < hi; k++) {
        const int v = perm[k];
        int j = k - 1;
        while (j >= lo && perm[j] > v) { perm[j + 1] = perm[j]; j--; }
        perm[j + 1] = v;
      }
      for (int k = lo; k < hi; k++) inv[perm[k]] = k;
    }
  }
}
static __global__ void k_map_rows(int n, const int *__restrict__ inv, const int *__restrict__ in, int *__restrict__ out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = inv[in[i]];
}
static __global__ void k_map_rows_pad(int n, const int *__restrict__ inv, const int *__restrict__ in, int *__restrict__ out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = in[i] >= 0 ? inv[in[i]] : -1;  // -1: padding of a cluster's member table
}
// dd_slot[rows[r]] = r: where k_nl_build stores the dd row of an atom (launch order of the lp sweep)
static __global__ void k_slot_from_rows(int n, const int *__restrict__ rows, int *__restrict__ slot) {
  int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r < n) slot[rows ? rows[r] : r] = r;
}
// colours of the atoms in this step's s order (colour re-validation): -1 = no colour
static __global__ void k_color_map(int n, const int *__restrict__ perm, const int *__restrict__ color_orig, int *__restrict__ color_s) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) color_s[i] = color_orig[perm[i]];
}
static __global__ void k_map_range(int lo, int n, const int *__restrict__ inv, int *__restrict__ out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = inv[lo + i];
}

// ------------------------------------------------------------------------------------------
// Colour phases of the list-mode Gauss-Seidel, built ON THE DEVICE (round 3; the host-side conflict graph + DSATUR of
// rounds 1-2 cost 70-80 ms at 135k atoms, whenever a colouring had to be rebuilt).  Atoms of one colour must be farther
// apart than the colour distance (they are relaxed Jacobi-fashion against each other inside a launch).
//   k_color_adj    conflict lists: the polarizable own atoms within the colour distance (3 x 3 x 3 cells), `apitch` per atom
//                  (an atom with more raises a flag and the lists are rebuilt wider), and the atom's degree
//   k_color_round  Jones-Plassmann: an uncoloured atom whose priority beats every uncoloured neighbour's takes the lowest
//                  colour none of its coloured neighbours holds.  Neighbours never decide in the same round, so no race.
//   k_color_fold   atoms of the highest class move to a lower colour their neighbours leave free (no two atoms of one class
//                  are adjacent, so all of them can move at once): classes dissolve from the top, 6 -> 4 on the MOF boxes
//   k_color_stats / k_color_relabel / k_color_cellcount / k_color_fill: phase order (ranked: by mean rank metric) and the
//                  rows of every phase in cell order
__device__ __forceinline__ unsigned color_hash(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}
static __global__ void k_color_adj(int n, const double4 *__restrict__ pos4, const int *__restrict__ perm, int own_lo, int own_hi,
                            Box box, CellGrid g, const long long *__restrict__ cell_first, const int *__restrict__ npol,
                            double colordistsq, int apitch, int *__restrict__ adj, int *__restrict__ deg,
                            unsigned long long *__restrict__ prio, int *__restrict__ color_s, int *__restrict__ flags,
                            int with_halo, int sw) {
  // `with_halo` (multi-GPU, one colouring consistent across the ranks): the polarizable atoms this handle holds for other
  // ranks' rows count as neighbours too.  They are never coloured here (deg = -1): their colour arrives from their owner
  // (-1 until then: an uncoloured neighbour constrains nothing, its owner will respect OUR colours when its turn comes).
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double4 ri = pos4[i];
  const int o = perm[i];
  const bool row = __double2loint(ri.w) && o >= own_lo && o < own_hi;
  color_s[i] = -1;
  if (!row) { deg[i] = -1; prio[i] = 0ull; return; }
  int cc[3];
  {
    double fr3[3];
    frac_coords(box, g.lo, ri.x, ri.y, ri.z, fr3);
#pragma unroll
    for (int k = 0; k < 3; k++) {
      double fr = fr3[k];
      fr -= floor(fr);
      int ck = (int)(fr * g.nc[k]);
      cc[k] = ck >= g.nc[k] ? g.nc[k] - 1 : ck;
    }
  }
  int d = 0;
  // `sw` cells to either side (1 where the colour distance fits into a cell, 2 for the small cells of short cutoffs)
  for (int dz = -sw; dz <= sw; dz++)
    for (int dy = -sw; dy <= sw; dy++)
      for (int dx = -sw; dx <= sw; dx++) {
        int b[3] = {cc[0] + dx, cc[1] + dy, cc[2] + dz};
        bool ok = true;
#pragma unroll
        for (int k = 0; k < 3; k++) {
          if (b[k] < 0 || b[k] >= g.nc[k]) {
            if (!box.periodic[k] || g.nc[k] < 2 * sw + 1) ok = false;  // (fewer cells than the stencil is wide: every cell is visited through the offsets inside the grid already)
            b[k] = (b[k] + 2 * g.nc[k]) % g.nc[k];
          }
        }
        if (!ok) continue;
        const long long cj = ((long long)b[2] * g.nc[1] + b[1]) * g.nc[0] + b[0];
        const int a = (int)cell_first[cj], e = a + npol[cj];  // the polarizable atoms of a cell come first
        for (int j = a; j < e; j++) {
          if (j == i) continue;
          const double4 rj = pos4[j];
          double ex, ey, ez;
          min_image_rint(box, ri.x, ri.y, ri.z, rj.x, rj.y, rj.z, ex, ey, ez);
          if (ex * ex + ey * ey + ez * ez < colordistsq) {
            const int oj = perm[j];
            if (!with_halo && (oj < own_lo || oj >= own_hi)) continue;
            if (d < apitch) adj[(size_t)i * apitch + d] = j;
            d++;
          }
        }
      }
  if (d > apitch) { atomicMax(flags, d); d = apitch; }
  deg[i] = d;
  // largest degree first; ties by a hash of the atom's index in the caller's order (the same atoms get the same priorities
  // step after step, whatever this step's cell order), then by that index
  prio[i] = ((unsigned long long)(d < 255 ? d : 255) << 56) | ((unsigned long long)(color_hash((unsigned)o) >> 8) << 32) | (unsigned)o;
}
// Sequential DSATUR, cell by cell: one wave colours the rows of ONE cell in saturation order (always the uncoloured row that
// sees the most colours; ties by degree, then by position in the cell), every decision made on up-to-date colours.  Cells of
// one launch are never adjacent (parity classes of the cell grid; a cell edge is at least half a cutoff, far beyond the colour
// distance), so no two waves of a launch ever colour neighbours: 8 launches walk the whole box.  The parallel rounds of
// Jones-Plassmann decide many neighbours-of-neighbours on stale saturation counts and end with 5 classes on the MOF boxes;
// this order finds the 4 the host-side DSATUR of rounds 1-2 found.  cells: c_k = start_k + stride_k * i_k, i_k < count_k.
static __global__ __launch_bounds__(64) void k_color_cells(int s0, int s1, int s2, int t0, int t1, int t2, int m0, int m1, int m2, int n0, int n1,
                                                    const long long *__restrict__ cell_first, const int *__restrict__ npol, int apitch,
                                                    const int *__restrict__ adj, const int *__restrict__ deg, int *__restrict__ color_s,
                                                    int *__restrict__ flags) {
  const int lane = threadIdx.x;
  const int b = blockIdx.x;
  const int i0 = b % m0, i1 = (b / m0) % m1, i2 = b / (m0 * m1);
  if (i2 >= m2) return;
  const long long c = ((long long)(s2 + t2 * i2) * n1 + (s1 + t1 * i1)) * n0 + (s0 + t0 * i0);
  const int a = (int)cell_first[c], np = npol[c];
  for (int base = 0; base < np; base += 64) {   // (cells with more than 64 rows: 64 at a time)
    const int i = a + base + lane;
    const bool row = base + lane < np && deg[i] >= 0;
    int mine = row ? color_s[i] : 0;            // >= 0: done (or not a row)
    // One pass over the conflict list: the colours the neighbours hold now, and which of them sit in this batch (bit = lane).
    // Neighbours outside the cell do not change during this launch (no adjacent cell is in it), those inside are tracked
    // through the broadcast of every pick: the loop below touches no memory but the one store of the picked colour.
    unsigned long long used = 0ull, inbatch = 0ull;
    int d = 0;
    if (row && mine < 0) {
      d = deg[i];
      for (int k = 0; k < d; k++) {
        const int j = adj[(size_t)i * apitch + k];
        const int cj = color_s[j];
        if (cj >= 0) used |= 1ull << cj;
        const int rel = j - (a + base);
        if (rel >= 0 && rel < 64) inbatch |= 1ull << rel;
      }
    }
    const int todo = __popcll(__ballot(row && mine < 0));
    for (int it = 0; it < todo; it++) {
      // key = (saturation, degree, first in the cell); 0 for lanes with nothing to colour
      const unsigned key = (row && mine < 0) ? (((unsigned)__popcll(used) + 1u) << 16) | ((unsigned)(d < 255 ? d : 255) << 8) | (unsigned)(63 - lane) : 0u;
      unsigned best = key;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) { const unsigned v = (unsigned)__shfl_xor((int)best, o, 64); best = v > best ? v : best; }
      const int wl = 63 - (int)(best & 63u);    // the picked lane (the lane number is part of the key: exactly one)
      int cm = 0;
      if (lane == wl) {
        cm = __ffsll((long long)~used) - 1;
        if (cm >= 64) { atomicMax(flags, 1000); cm = 63; }
        mine = cm;
        color_s[i] = cm;
      }
      cm = __shfl(cm, wl, 64);
      if ((inbatch >> wl) & 1ull) used |= 1ull << cm;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");   // a second batch of the same cell reads these colours from memory
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
}

// priorities of the uncoloured atoms for the next round: (colours already seen among the neighbours, degree, hash, index) --
// the saturation-first order of DSATUR, evaluated on a snapshot (its own launch) so that two neighbours never both think
// they go first
static __global__ void k_color_prio(int n, int apitch, const int *__restrict__ adj, const int *__restrict__ deg, const int *__restrict__ color_s,
                             const int *__restrict__ perm, unsigned long long *__restrict__ prio, int hashed) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int d = deg[i];
  if (d < 0 || color_s[i] >= 0) return;
  unsigned long long used = 0ull;
  for (int k = 0; k < d; k++) { const int cj = color_s[adj[(size_t)i * apitch + k]]; if (cj >= 0) used |= 1ull << cj; }
  const unsigned o = (unsigned)perm[i];
  // ties by the atom's index (`hashed` = 0): in a crystal built cell by cell the decisions then sweep through the structure in
  // ONE consistent order, as the sequential DSATUR's do, and find its 4 classes on the MOF boxes where hashed ties end with 5;
  // the price is rounds (a dependency chain per replica).  `hashed` = 1 breaks chains that grow too long.
  const unsigned long long tie = hashed ? ((unsigned long long)(color_hash(o) >> 16) << 32) | o : (unsigned long long)o;
  prio[i] = ((unsigned long long)__popcll(used) << 56) | ((unsigned long long)(d < 255 ? d : 255) << 48) | tie;
}
static __global__ void k_color_round(int n, int apitch, const int *__restrict__ adj, const int *__restrict__ deg,
                              const unsigned long long *__restrict__ prio, int *__restrict__ color_s, int *__restrict__ left) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int d = deg[i];
  if (d < 0 || color_s[i] >= 0) return;
  const unsigned long long pi = prio[i];
  unsigned long long used = 0ull;
  for (int k = 0; k < d; k++) {
    const int j = adj[(size_t)i * apitch + k];
    const int cj = color_s[j];
    if (cj >= 0) used |= 1ull << cj;
    else if (prio[j] > pi) { atomicAdd(left, 1); return; }  // a stronger neighbour decides first
  }
  color_s[i] = __ffsll((long long)~used) - 1;
}
static __global__ void k_color_fold(int n, int apitch, int top, const int *__restrict__ adj, const int *__restrict__ deg, int *__restrict__ color_s,
                             int *__restrict__ stay) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n || deg[i] < 0 || color_s[i] != top) return;
  unsigned long long used = 0ull;
  for (int k = 0; k < deg[i]; k++) { const int cj = color_s[adj[(size_t)i * apitch + k]]; if (cj >= 0) used |= 1ull << cj; }  // (no neighbour has colour `top`; -1: a halo atom its owner has not coloured yet)
  const int c = __ffsll((long long)~used) - 1;
  if (c < top) color_s[i] = c;
  else atomicAdd(stay, 1);
}
// Local repair of a small top class.  A row v stuck in the top class sees every lower colour among its neighbours; the ball of
// rows within `hops` conflict-graph steps of v is uncoloured and coloured again with the lower colours only, by exhaustive
// search (depth first, at most 64 rows, a bounded number of steps) against the fixed colours around the ball.  If the search
// fails nothing changes.  k_color_collect lists the rows of the top class (then sorted by index: k_sort_small), k_color_ball
// gives one wave to each; a wave waits while an earlier, still untried row of the list lies within `reach` (the balls write up
// to `hops` steps from their centres and read one step further), so the result does not depend on the order the waves run in.
// `state` 0 = untried, 1 = tried; `prev` = the states before this launch.
static __global__ void k_color_collect(int n, int top, const int *__restrict__ color_s, int cap, int *__restrict__ list, int *__restrict__ count) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n || color_s[i] != top) return;
  const int k = atomicAdd(count, 1);
  if (k < cap) list[k] = i;
}
static __global__ __launch_bounds__(256) void k_sort_small(const int *__restrict__ count, int cap, const int *__restrict__ in, int *__restrict__ out,
                                                    int *__restrict__ state_a, int *__restrict__ state_b) {
  const int m = *count;
  if (m > cap) return;   // (the host skips the repair as well)
  for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < m; k += gridDim.x * blockDim.x) {
    const int v = in[k];
    int r = 0;
    for (int q = 0; q < m; q++) r += in[q] < v;
    out[r] = v;
    state_a[k] = 0; state_b[k] = 0;
  }
}
static __global__ __launch_bounds__(64) void k_color_ball(const int *__restrict__ count, int cap, const int *__restrict__ list, const int *__restrict__ prev,
                                                   int *__restrict__ state, const double4 *__restrict__ pos4, Box box, double reachsq, int top, int hops,
                                                   int budget, int apitch, const int *__restrict__ adj, const int *__restrict__ deg, int *__restrict__ color_s) {
  __shared__ int ball[64];
  const int m_all = *count;
  const int w = blockIdx.x, lane = threadIdx.x;
  if (m_all > cap || w >= m_all) return;
  if (prev[w]) { if (lane == 0) state[w] = 1; return; }
  const int v = list[w];
  const double4 rv = pos4[v];
  bool blocked = false;
  for (int q = lane; q < w; q += 64) {
    if (prev[q]) continue;
    const double4 rq = pos4[list[q]];
    double dx, dy, dz;
    min_image_del(box, rv.x, rv.y, rv.z, rq.x, rq.y, rq.z, dx, dy, dz);
    blocked |= dx * dx + dy * dy + dz * dz < reachsq;
  }
  if (__ballot(blocked)) { if (lane == 0) state[w] = 0; return; }
  if (lane == 0) state[w] = 1;
  // the ball, breadth first
  if (lane == 0) ball[0] = v;
  int m = 1, level0 = 0;
  __syncthreads();
  for (int hop = 0; hop < hops; hop++) {
    const int level1 = m;
    for (int q = level0; q < level1 && m < 64; q++) {
      const int node = ball[q];
      const int d = deg[node];
      int j = lane < d ? adj[(size_t)node * apitch + lane] : -1;
      if (j >= 0 && deg[j] < 0) j = -1;   // another rank's row: its colour is fixed
      if (j >= 0) for (int k = 0; k < m; k++) if (ball[k] == j) { j = -1; break; }
      const unsigned long long fresh = __ballot(j >= 0);
      const int at = m + __popcll(fresh & ((1ull << lane) - 1ull));
      if (j >= 0 && at < 64) ball[at] = j;
      m = min(64, m + __popcll(fresh));
      __syncthreads();
    }
    level0 = level1;
  }
  // per ball row (lane = position in the ball): the colours its neighbours outside the ball hold, and its neighbours inside
  unsigned long long in = 0ull, allow = 0ull;
  if (lane < m) {
    const int node = ball[lane];
    unsigned long long used = 0ull;
    const int d = deg[node];
    for (int k = 0; k < d; k++) {
      const int j = adj[(size_t)node * apitch + k];
      int at = -1;
      for (int q = 0; q < m; q++) if (ball[q] == j) { at = q; break; }
      if (at >= 0) in |= 1ull << at;
      else { const int cj = color_s[j]; if (cj >= 0) used |= 1ull << cj; }
    }
    allow = ~used & ((1ull << top) - 1ull);
  }
  // Depth-first search, rows in ball order (the centre first, then by distance), the whole wave in step: lane q keeps the
  // colour of ball row q in a register, the colours the earlier in-ball neighbours of row k hold are collected with one
  // ballot per colour -- no memory in the loop (a one-lane version walking LDS arrays took about a microsecond per step).
  int mycol = -1;
  int k = 0;
  for (int step = 0; step < budget && k >= 0 && k < m; step++) {
    const unsigned long long ek = __shfl(in, k, 64) & ((1ull << k) - 1ull);   // earlier in-ball neighbours of row k (k < 64)
    const bool nb = (ek >> lane) & 1ull;
    unsigned long long ok = __shfl(allow, k, 64);
    for (int c = 0; c < top; c++)
      if (((ok >> c) & 1ull) && __ballot(nb && mycol == c)) ok &= ~(1ull << c);
    const int cur = __shfl(mycol, k, 64);
    if (cur >= 0) ok &= ~((2ull << cur) - 1ull);   // colours above the one tried last
    if (ok) {
      const int c = __ffsll((long long)ok) - 1;
      if (lane == k) mycol = c;
      k++;
      if (lane == k) mycol = -1;
    } else {
      if (lane == k) mycol = -1;
      k--;
    }
  }
  if (k >= m && lane < m) color_s[ball[lane]] = mycol;   // found: every row of the ball has a colour below `top`
}
// iterated greedy (Culberson): recolour greedily in an order that keeps every old class together -- never more colours than
// before, often fewer.  This kernel sets the stage: priorities = (rank of the atom's old class in the new order, hash, index),
// colours cleared; k_color_round then needs one round per old class (a class is an independent set: it decides at once).
static __global__ void k_color_regroup(int n, const int *__restrict__ deg, const int *__restrict__ class_rank, const int *__restrict__ perm,
                                int *__restrict__ color_s, unsigned long long *__restrict__ prio) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n || deg[i] < 0) return;
  const unsigned o = (unsigned)perm[i];
  prio[i] = ((unsigned long long)(63 - class_rank[color_s[i]]) << 56) | ((unsigned long long)(color_hash(o) >> 8) << 32) | o;
  color_s[i] = -1;
}
// per colour: rows and the sum of their rank metric (phase order of the ranked flavour)
static __global__ __launch_bounds__(256) void k_color_stats(int n, const int *__restrict__ color_s, const double *__restrict__ rank,
                                                     double *__restrict__ sums, const int *__restrict__ perm, int own_lo, int own_hi) {
  // (summed per workgroup in LDS first: a hundred thousand FP64 atomics on five addresses took 2 ms)
  __shared__ double part[128];
  if (threadIdx.x < 128) part[threadIdx.x] = 0.0;
  __syncthreads();
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  int c = i < n ? color_s[i] : -1;
  if (c >= 0) { const int o = perm[i]; if (o < own_lo || o >= own_hi) c = -1; }   // (a halo atom carries its owner's colour: not a row here)
  if (c >= 0) {
    atomicAdd(&part[2 * c], 1.0);
    atomicAdd(&part[2 * c + 1], rank ? rank[i] : 1.0);
  }
  __syncthreads();
  if (threadIdx.x < 128 && part[threadIdx.x] != 0.0) atomicAdd(sums + threadIdx.x, part[threadIdx.x]);
}
static __global__ void k_color_relabel(int n, const int *__restrict__ relabel, const int *__restrict__ perm, int *__restrict__ color_s,
                                int *__restrict__ color_orig) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int c = color_s[i] >= 0 ? relabel[color_s[i]] : -1;
  color_s[i] = c;
  color_orig[perm[i]] = c;
}
// rows of every phase, cells in order, atoms of a cell in order: counts per (class, cell), one scan, fill.  Rows = the
// polarizable atoms this handle owns (halo atoms carry their owners' colours and are skipped).  Without `bflag` a class is a
// colour; with `sub` (a sub-class 0 .. nsub-1 per atom, by original index) class nsub * c + k holds colour c's rows of
// sub-class k: multi-GPU: 0 = boundary rows ("a peer receives this row's dipole"), 1 = interior rows, so that a phase can
// send its boundary rows off while the interior rows are still swept.
__device__ __forceinline__ int color_class(int j, const int *__restrict__ color_s, const int *__restrict__ perm, int own_lo, int own_hi,
                                           const int *__restrict__ sub, int nsub) {
  const int c = color_s[j];
  if (c < 0) return -1;
  const int o = perm[j];
  if (o < own_lo || o >= own_hi) return -1;
  return sub ? nsub * c + sub[o] : c;
}
static __global__ void k_color_cellcount(long long ncell, int nclass, const long long *__restrict__ cell_first, const int *__restrict__ npol,
                                  const int *__restrict__ color_s, int *__restrict__ cnt, const int *__restrict__ perm, int own_lo, int own_hi,
                                  const int *__restrict__ sub, int nsub) {
  const long long c = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (c >= ncell) return;
  const int a = (int)cell_first[c], e = a + npol[c];
  for (int q = 0; q < nclass; q++) {
    int k = 0;
    for (int j = a; j < e; j++) k += color_class(j, color_s, perm, own_lo, own_hi, sub, nsub) == q;
    cnt[(size_t)q * ncell + c] = k;
  }
}
static __global__ void k_color_fill(long long ncell, int nclass, const long long *__restrict__ cell_first, const int *__restrict__ npol,
                             const int *__restrict__ color_s, const int *__restrict__ perm, const long long *__restrict__ off,
                             int *__restrict__ rows_orig, int own_lo, int own_hi, const int *__restrict__ sub, int nsub) {
  const long long c = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (c >= ncell) return;
  const int a = (int)cell_first[c], e = a + npol[c];
  for (int q = 0; q < nclass; q++) {
    long long w = off[(size_t)q * ncell + c];
    for (int j = a; j < e; j++) if (color_class(j, color_s, perm, own_lo, own_hi, sub, nsub) == q) rows_orig[w++] = perm[j];
  }
}

// One wave per atom row; lanes stride the atoms of the <=27 distinct neighbor cells (contiguous s
// ranges); ballot + popcount compacts in order.  Single pass into the pitched lists:
//   nl : every j with rsq <= cutallsq                      (static field, forces, rank metric)
//   dd : alpha_i != 0, alpha_j != 0 and rsq < ddcutsq      (the dipole sweep stream)
// cnt[] receives the TRUE counts; writes stop at the pitch and *overflow is raised.
template <bool TRI, bool RECHECK>
static __global__ __launch_bounds__(POLAR_BLOCK) void k_nl_build(const int *__restrict__ rows, int nrows,
                                                          const double4 *__restrict__ pos4, Box box, CellGrid g,
                                                          const long long *__restrict__ cell_first, double cutallsq,
                                                          double ddcutsq, long long nl_pitch, long long dd_pitch,
                                                          int *__restrict__ nl_cnt, int *__restrict__ dd_cnt,
                                                          int *__restrict__ nl_j, int *__restrict__ dd_j,
                                                          double *__restrict__ dd_r2, int pad_dd,
                                                          int dd_shift_qm, int dd_pad_index, int *__restrict__ dd_wrap,
                                                          const int *__restrict__ color_s, double colordistsq,
                                                          int *__restrict__ color_conflict,
                                                          const int *__restrict__ dd_slot,
                                                          int *__restrict__ overflow,
                                                          unsigned long long *__restrict__ dd_total) {
  const int dd_shift = dd_shift_qm & 255, dd_qm = dd_shift_qm >> 8;  // record shift of the lp stream | quad-major slot order
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * POLAR_ROWS_PER_BLOCK + (threadIdx.x >> 6);
  if (row >= nrows) return;
  const int i = rows ? rows[row] : row;  // s space: the atoms of cell c are the indices [cell_first[c], cell_first[c+1])
  const double4 ri = pos4[i];            // {x, y, z, (molecule, polarizable)}
  const int imol = __double2hiint(ri.w), ipol = __double2loint(ri.w);
  // colour re-validation on reneighbor steps (color_s != NULL): the colouring of the previous list stays in use unless
  // two polarizable atoms of one colour have come closer than the colour distance
  const int icol = (RECHECK && color_s && ipol) ? color_s[i] : -2;
  bool clash = RECHECK && color_s && ipol && icol < 0;  // a polarizable atom without a colour
  // home cell and the position inside it in cell units (same arithmetic as cell_of)
  int cc[3];
  double uu[3], edge[3];
  {
    double fr3[3];
    frac_coords(box, g.lo, ri.x, ri.y, ri.z, fr3);
#pragma unroll
    for (int k = 0; k < 3; k++) {
      double fr = fr3[k];
      fr -= floor(fr);
      const double t = fr * g.nc[k];
      int ck = (int)t;
      ck = ck >= g.nc[k] ? g.nc[k] - 1 : ck;
      cc[k] = __builtin_amdgcn_readfirstlane(ck);  // the row atom is the same in every lane
      uu[k] = wave_uniform(t - ck); edge[k] = wave_uniform(box.prd[k] / g.nc[k]);  // (trimming: orthogonal boxes only)
    }
  }
  const int c0 = cc[0], c1 = cc[1], c2 = cc[2];
  // lp sweep: the dd row of atom i is stored where the sweep will walk it (dd_slot[i] = its row in launch order), so that a
  // sweep wave can request its index stream before it knows which atom it works on; rows without a slot hold no dd pair
  const int slot_i = dd_slot ? dd_slot[i] : i;
  const long long nl0 = (long long)i * nl_pitch, dd0 = (long long)(slot_i >= 0 ? slot_i : 0) * dd_pitch;
  int ncount = 0, dcount = 0;
  bool wrap_lane = false;  // this lane saw a dd pair of the row that reaches across a periodic face (lp sweep: rows without skip the wrap)
  // Cells have an edge >= cutoff/2, so the stencil reaches +-2 cells (125 cells hold 42 % fewer
  // candidates than 27 cells of edge >= cutoff).  Cells are stored x-fastest, so the 5 cells of a
  // stencil row are ONE contiguous run of atoms (two runs when the row wraps around the box): the
  // lanes stride runs of ~100 atoms instead of single small cells.  Dimensions with fewer than 5
  // cells visit every cell exactly once.
  // The stencil is trimmed per atom: a (y,z) row of cells whose nearest point is beyond the cutoff is
  // skipped, and its x-run is cut to the cells a sphere of the remaining radius can reach.
  const unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  const int n0 = g.nc[0], n1 = g.nc[1], n2 = g.nc[2];
  const int zlo = n2 >= 5 ? c2 - 2 : 0, zcnt = n2 >= 5 ? 5 : n2;
  const int ylo = n1 >= 5 ? c1 - 2 : 0, ycnt = n1 >= 5 ? 5 : n1;
  const double reach = sqrt(cutallsq > ddcutsq ? cutallsq : ddcutsq) + 1e-6;  // margin: atoms sit on cell faces
  const double reach2 = wave_uniform(reach * reach);
  // Lane L works out stencil row L (<= 25 rows of cells): skipped or not, its x-range, and the atom ranges
  // of its one or two contiguous runs (two when the row wraps around the box) -- all cell_first loads of a
  // row atom are issued at once instead of one dependent pair per stencil row.  The wave then walks the
  // rows uniformly and reads the ranges with v_readlane.
  const int nsr = zcnt * ycnt;
  int ra0 = 0, rb0 = 0, ra1 = 0, rb1 = 0;  // run 0: [ra0, rb0), run 1: [ra1, rb1) (s indices)
  if (lane < nsr) {
    const int zz = zlo + lane / ycnt, yy = ylo + lane % ycnt;
    int b2 = zz, b1 = yy;
    bool ok = true;
    if (b2 < 0 || b2 >= n2) { if (!box.periodic[2]) ok = false; b2 = (b2 + n2) % n2; }
    if (b1 < 0 || b1 >= n1) { if (!box.periodic[1]) ok = false; b1 = (b1 + n1) % n1; }
    double dzmin = 0.0, dymin = 0.0;
    if (n2 >= 5) { const int d = zz - c2; dzmin = d > 0 ? (d - uu[2]) * edge[2] : (d < 0 ? (uu[2] - (d + 1)) * edge[2] : 0.0); }
    if (n1 >= 5) { const int d = yy - c1; dymin = d > 0 ? (d - uu[1]) * edge[1] : (d < 0 ? (uu[1] - (d + 1)) * edge[1] : 0.0); }
    const double rem2 = reach2 - dzmin * dzmin - dymin * dymin;
    if (g.trim && rem2 < 0.0) ok = false;  // the whole row of cells is out of reach
    int xlo = n0 >= 5 ? c0 - 2 : 0, xhi = n0 >= 5 ? c0 + 2 : n0 - 1;
    if (ok && g.trim && n0 >= 5) {
      const double xr = sqrt(rem2) / edge[0];
      int lo_off = (int)floor(uu[0] - xr), hi_off = (int)floor(uu[0] + xr);
      lo_off = lo_off < -2 ? -2 : lo_off; hi_off = hi_off > 2 ? 2 : hi_off;
      xlo = c0 + lo_off; xhi = c0 + hi_off;
    }
    if (ok) {
      const long long rowbase = ((long long)b2 * n1 + b1) * n0;
      // inside piece, then the wrapped piece (below 0 or above n0-1; never both: the range spans <= 5 <= n0 cells)
      const int xa = xlo < 0 ? 0 : xlo, xb = xhi >= n0 ? n0 - 1 : xhi;
      ra0 = (int)cell_first[rowbase + xa]; rb0 = (int)cell_first[rowbase + xb + 1];
      if (box.periodic[0]) {
        if (xlo < 0) { ra1 = (int)cell_first[rowbase + xlo + n0]; rb1 = (int)cell_first[rowbase + n0]; }
        else if (xhi >= n0) { ra1 = (int)cell_first[rowbase]; rb1 = (int)cell_first[rowbase + xhi - n0 + 1]; }
      }
    }
  }
  for (int sr = 0; sr < nsr; sr++) {
#pragma unroll
    for (int piece = 0; piece < 2; piece++) {
      const int a = __builtin_amdgcn_readlane(piece ? ra1 : ra0, sr), b = __builtin_amdgcn_readlane(piece ? rb1 : rb0, sr);
      for (int base = a; base < b; base += 64) {
        const int p = base + lane;
        bool in_nl = false, in_dd = false;
        const int j = p;
        int same = 0;
        double rsq = 0.0;
        if (p < b && j != i) {
          const double4 rj = pos4[j];  // consecutive lanes read consecutive 32-byte entries
          double ex, ey, ez;
          const bool shifted = min_image_rint_w<TRI>(box, ri.x, ri.y, ri.z, rj.x, rj.y, rj.z, ex, ey, ez);
          rsq = ex * ex + ey * ey + ez * ez;
          in_nl = rsq <= cutallsq;
          in_dd = ipol && slot_i >= 0 && __double2loint(rj.w) && (rsq < ddcutsq);
          if (RECHECK && icol >= 0 && rsq < colordistsq && __double2loint(rj.w) && color_s[j] == icol) clash = true;
          wrap_lane |= in_dd && shifted;
          same = (imol != 0 && imol == __double2hiint(rj.w)) ? POLAR_NL_SAMEMOL : 0;
        }
        const unsigned long long m_nl = __ballot(in_nl), m_dd = __ballot(in_dd);
        const int kn = ncount + __popcll(m_nl & below), kd = dcount + __popcll(m_dd & below);
        // bit 30 of an nl entry: "same non-zero molecule" -- the static field and the charge-dipole terms
        // skip such pairs (PS.cpp:342,454), so those kernels need no molecule gather
        if (in_nl && kn < nl_pitch) nl_j[nl0 + kn] = j | same;
        if (in_dd && kd < dd_pitch) {
          // lp sweep (dd_shift 6): byte offset of the 64-byte record, stored in the chunked order (lp_slot)
          dd_j[dd0 + (dd_shift ? lp_slot(kd, dd_qm) : (long long)kd)] = j << dd_shift;
          if (dd_r2) dd_r2[dd0 + kd] = rsq;  // the sweep's per-pair stream value (same positions, same image rule)
        }
        ncount += __popcll(m_nl);
        dcount += __popcll(m_dd);
      }
    }
  }
  if (pad_dd) {  // pad the dd row to whole 64-pair trips with inert entries (the component-per-lane sweep has no lane masks)
    const int have = dcount < dd_pitch ? dcount : (int)dd_pitch;
    const int padded = (have + 63) & ~63;
    for (int k = have + lane; k < padded; k += 64) {
      dd_j[dd0 + (dd_shift ? lp_slot(k, dd_qm) : (long long)k)] = (dd_pad_index >= 0 ? dd_pad_index : i) << dd_shift;
      if (dd_r2) dd_r2[dd0 + k] = 1e60;  // s3 ~ 1e-90, and d = 0 kills the s5 term: contributes nothing
    }
  }
  if (RECHECK && color_s && __ballot(clash) != 0ull && lane == 0) atomicOr(color_conflict, 1);
  const unsigned long long anywrap = __ballot(wrap_lane);  // all lanes are back together here
  if (lane == 0) {
    nl_cnt[i] = ncount; dd_cnt[i] = dcount;
    if (dd_wrap) dd_wrap[i] = anywrap != 0ull;
    if (ncount > nl_pitch || dcount > dd_pitch) atomicMax(overflow, ncount > dcount ? ncount : dcount);
    if (dcount) atomicAdd(dd_total + (blockIdx.x & 63) * 16, (unsigned long long)(dcount < dd_pitch ? dcount : (int)dd_pitch));
  }
}

#ifdef POLAR_LAB
#include "lab/lists_paired_rows.hpp"
#endif  // POLAR_LAB (paired rows)

#ifdef POLAR_LAB
#include "lab/lists_cluster_rows.hpp"
#endif  // POLAR_LAB

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
static __global__ void k_lj_cell_count(int nall, const double *__restrict__ x, LJGrid g, int *__restrict__ cell_id,
                                int *__restrict__ cell_cnt) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nall) return;
  const int c = lj_cell_of(g, x[3 * i], x[3 * i + 1], x[3 * i + 2]);
  cell_id[i] = c;
  atomicAdd(&cell_cnt[c], 1);
}
// s order = cell order: pos[s] = {x, y, z, (type, molecule)}, aux[s] = {atom index, tag}
static __global__ void k_lj_cell_fill(int nall, const int *__restrict__ cell_id, const long long *__restrict__ cell_first,
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
static __global__ __launch_bounds__(POLAR_BLOCK) void k_lj_nl_build(int row_lo, int nrows, int ntypes, const double *__restrict__ x,
                                                             const int *__restrict__ type, const int *__restrict__ mol,
                                                             const double4 *__restrict__ pos, const int2 *__restrict__ aux,
                                                             LJGrid g, const long long *__restrict__ cell_first,
                                                             const double *__restrict__ cutneighsq, Box box,
                                                             int exclude_intra, const int *__restrict__ nspecial,
                                                             const int *__restrict__ special, int maxspecial, int sf1,
                                                             int sf2, int sf3, long long pitch, int *__restrict__ cnt,
                                                             int *__restrict__ out_j, int *__restrict__ overflow,
                                                             unsigned long long *__restrict__ total, int typed) {
  extern __shared__ double cn_lds[];
  const int w = ntypes + 1;
  for (int t = threadIdx.x; t < w * w; t += blockDim.x) cn_lds[t] = cutneighsq[t];
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int r_ = blockIdx.x * POLAR_ROWS_PER_BLOCK + (threadIdx.x >> 6);
  if (r_ >= nrows) return;
  const int i = row_lo + r_;  // rows [row_lo, row_lo + nrows): the atoms this handle owns (all locals unless sharded)
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
          entry = typed ? (aj.x | (jtype << 24)) : aj.x;  // typed lists: the partner's type in bits 24-29 (k_ljcoul)
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
static __global__ void k_lj_rows(int row_lo, int nrows, long long pitch, int *__restrict__ ilist, long long *__restrict__ first) {
  int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= nrows) return;
  ilist[r] = row_lo + r;
  first[row_lo + r] = (long long)(row_lo + r) * pitch;
}

// multi-GPU plumbing: dipoles of a contiguous row range <-> packed [n][3] buffers.  The dipoles live in the record table
// the sweep works on: 64-byte AtomRecs, or the 48-byte sweep records of the tile sweep -- component k of record s is
// double 2k + 1 of the record at base + s * stride (MuView).
struct MuView {
  char *a, *b;  // the two buffers (Jacobi ping-pong; scal->cur picks)
  int stride;
};
__device__ __forceinline__ double *mu_of(const MuView &v, const Scal *scal, long long s) {
  return reinterpret_cast<double *>((scal->cur ? v.b : v.a) + s * v.stride);
}
static __global__ void k_mu_gather(long long lo, long long hi, const int *__restrict__ inv, const Scal *scal, MuView v,
                            double *__restrict__ dst) {
  long long i = lo + blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (i >= hi) return;
  const double *r = mu_of(v, scal, inv ? inv[i] : i);
  dst[3 * (i - lo)] = r[1]; dst[3 * (i - lo) + 1] = r[3]; dst[3 * (i - lo) + 2] = r[5];
}
static __global__ void k_mu_scatter(long long lo, long long hi, const int *__restrict__ inv, const Scal *scal, MuView v,
                             const double *__restrict__ src) {
  long long i = lo + blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (i >= hi) return;
  double *r = mu_of(v, scal, inv ? inv[i] : i);
  r[1] = src[3 * (i - lo)]; r[3] = src[3 * (i - lo) + 1]; r[5] = src[3 * (i - lo) + 2];
}

// halo exchange by index list (orig ids; negative entries are padding and skipped)
static __global__ void k_mu_gather_idx(long long n, const int *__restrict__ idx, const int *__restrict__ inv, const Scal *scal,
                                MuView v, double *__restrict__ dst) {
  long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (t >= n) return;
  const int o = idx[t];
  if (o < 0) return;
  const double *r = mu_of(v, scal, inv ? inv[o] : o);
  dst[3 * t] = r[1]; dst[3 * t + 1] = r[3]; dst[3 * t + 2] = r[5];
}
// The end-of-sweep logic and the pack of the halo dipoles in ONE launch (round 4; the multi-GPU driver's one-exchange-per-sweep
// schedule, Gauss-Seidel): workgroup 0 is k_solver_step, the others k_mu_gather_idx.  The two do not touch each other's data
// (in-place sweeps never flip `cur`), and a dependent launch costs ~5 us whatever it does: 38 sweeps x 1 launch.
static __global__ __launch_bounds__(POLAR_NSLOT) void k_solver_step_gather(Scal *scal, double *__restrict__ slots, int nlocal, int fixed_iteration,
                                                                  int iterations_max, double precision, const double *__restrict__ global_change,
                                                                  const double *__restrict__ part, int npart, long long n,
                                                                  const int *__restrict__ idx, const int *__restrict__ inv, MuView v,
                                                                  double *__restrict__ dst) {
  if (blockIdx.x == 0) {
    if (scal->done) return;
    __shared__ double red[POLAR_NSLOT / 64];
    double c = slots[(size_t)threadIdx.x * POLAR_SLOT_STRIDE + SL_CHANGE];
    slots[(size_t)threadIdx.x * POLAR_SLOT_STRIDE + SL_CHANGE] = 0.0;
    for (int k = threadIdx.x; k < npart; k += POLAR_NSLOT) c += part[k];
    c = wave_sum(c);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x != 0) return;
    double sum = 0.0;
    for (int k = 0; k < POLAR_NSLOT / 64; k++) sum += red[k];
    solver_decide(scal, sum, global_change, nlocal, fixed_iteration, iterations_max, precision, 0, 1);
    return;
  }
  const long long t = (blockIdx.x - 1) * (long long)blockDim.x + threadIdx.x;
  if (t >= n) return;
  const int o = idx[t];
  if (o < 0) return;
  const double *r = mu_of(v, scal, inv ? inv[o] : o);
  dst[3 * t] = r[1]; dst[3 * t + 1] = r[3]; dst[3 * t + 2] = r[5];
}
static __global__ void k_mu_scatter_idx(long long n, const int *__restrict__ idx, const int *__restrict__ inv, const Scal *scal,
                                 MuView v, const double *__restrict__ src, int own_lo, int own_hi) {
  long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (t >= n) return;
  const int o = idx[t];
  if (o < 0 || (o >= own_lo && o < own_hi)) return;  // padding, or a row this handle owns itself
  double *r = mu_of(v, scal, inv ? inv[o] : o);
  r[1] = src[3 * t]; r[3] = src[3 * t + 1]; r[5] = src[3 * t + 2];
}

// the same plumbing for the colours of a distributed colouring (int per atom, s space: color_s) and for positions
// (orig space: x[nall][3]); idx = handle-local atom indices
// (colours travel as doubles: one data type on the wire)
static __global__ void k_color_gather_idx(long long n, const int *__restrict__ idx, const int *__restrict__ inv, const int *__restrict__ color_s,
                                          double *__restrict__ dst) {
  long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (t >= n) return;
  dst[t] = (double)color_s[inv ? inv[idx[t]] : idx[t]];
}
static __global__ void k_color_scatter_idx(long long n, const int *__restrict__ idx, const int *__restrict__ inv, int *__restrict__ color_s,
                                           const double *__restrict__ src) {
  long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (t >= n) return;
  color_s[inv ? inv[idx[t]] : idx[t]] = (int)src[t];
}
static __global__ void k_vec3_gather_idx(long long n, const int *__restrict__ idx, const double *__restrict__ x, double *__restrict__ dst) {
  long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (t >= n) return;
  const size_t o = (size_t)idx[t];
  dst[3 * t] = x[3 * o]; dst[3 * t + 1] = x[3 * o + 1]; dst[3 * t + 2] = x[3 * o + 2];
}
static __global__ void k_vec3_scatter_idx(long long n, const int *__restrict__ idx, double *__restrict__ x, const double *__restrict__ src) {
  long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (t >= n) return;
  const size_t o = (size_t)idx[t];
  x[3 * o] = src[3 * t]; x[3 * o + 1] = src[3 * t + 1]; x[3 * o + 2] = src[3 * t + 2];
}
// periodic images: atom first + g sits at its owner's position plus a whole number of box vectors
static __global__ void k_ghost_images(long long n, long long first, const int *__restrict__ owner, const double *__restrict__ shift,
                                      double *__restrict__ x) {
  long long g = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (g >= n) return;
  const size_t o = (size_t)owner[g], a = (size_t)(first + g);
  x[3 * a] = x[3 * o] + shift[3 * g]; x[3 * a + 1] = x[3 * o + 1] + shift[3 * g + 1]; x[3 * a + 2] = x[3 * o + 2] + shift[3 * g + 2];
}

// small utilities
// several small buffers zeroed by ONE launch (every launch costs ~4.4 us on this platform whatever it does)
struct ZeroJobs {
  unsigned int *p[6];
  unsigned long long nwords[6];  // 4-byte words
  int n;
};
static __global__ void k_zero_many(ZeroJobs jobs) {
  const unsigned long long gid = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x;
  const unsigned long long stride = gridDim.x * (unsigned long long)blockDim.x;
  for (int j = 0; j < jobs.n; j++)
    for (unsigned long long w = gid; w < jobs.nwords[j]; w += stride) jobs.p[j][w] = 0u;
}
static __global__ void k_zero_scal(Scal *s, int keep_solver) {
  s->eng_vdwl = s->eng_coul = s->u_self = s->u_ef = s->u_dd = 0.0;
  for (int k = 0; k < 6; k++) s->virial[k] = 0.0;
  s->change = 0.0; s->last_change = 0.0; s->pad = 0; s->det_change = 0.0;
  s->rmin_bits = (unsigned long long)__double_as_longlong(1000.0);
  if (!keep_solver) { s->iterations = 0; s->done = 0; s->status = 0; s->cur = 0; s->sweeps = 0; }
}
static __global__ void k_set_done(Scal *s, int done) { s->done = done; }

}  // namespace polar
