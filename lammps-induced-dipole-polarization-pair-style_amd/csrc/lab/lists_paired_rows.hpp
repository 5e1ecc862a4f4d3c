// lab/lists_paired_rows.hpp -- LAB BUILD ONLY (-DPOLAR_LAB, libpolar_mi355x_lab.so): code that was built, measured and did not become the
// product path.  Included by polar_lists.hpp inside `#ifdef POLAR_LAB`; the product library never sees it.
// No include guard: it is a fragment of polar_lists.hpp, textually in that file's scope.
// ------------------------------------------------------------------------------------------
// PAIRED ROWS (k_field_lp2, lab): two rows of ONE colour phase that sit in the same cell are swept by one wave over the
// UNION of their neighbours.  Rows of one phase are relaxed Jacobi-fashion against each other anyway (one launch), so nothing
// changes in the iteration; what changes is that a gathered record serves two rows (the spheres of two atoms a few A apart
// overlap by ~75 %): gathers, index stream and LDS reads per pair fall to ~0.64, row starts and ends halve.
//   k_unit_flag   rows of a phase are stored cell by cell: row r leads a unit when it is at an even place of its cell's run
//   (scan)        unit numbers
//   k_unit_fill   unit u = {row atom A, row atom B or -1}
//   k_dd_units    the union list of every unit: entries (j << 6) | inA | inB << 1 in the chunked layout of the lp stream,
//                 padded to whole trips with the dummy record; descriptor {A, B, trips | wrap << 30}
struct PhaseOff { int n; int off[66]; };
__device__ __forceinline__ int phase_of(const PhaseOff &P, int r) {
  int q = 0;
  while (q + 1 < P.n && r >= P.off[q + 1]) q++;
  return q;
}
static __global__ void k_unit_flag(int ntot, const int *__restrict__ rows, const int *__restrict__ perm, const int *__restrict__ cell_id,
                            PhaseOff P, int *__restrict__ lead) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= ntot) return;
  const int q = phase_of(P, r);
  const int lo = P.off[q];
  const int c = cell_id[perm[rows ? rows[r] : r]];
  int k = 0;  // place of row r inside the run of its cell (runs are short: the rows of one colour in one half-cutoff cell)
  for (int b = r - 1; b >= lo && cell_id[perm[rows ? rows[b] : b]] == c; b--) k++;
  lead[r] = (k & 1) ? 0 : 1;
}
static __global__ void k_unit_fill(int ntot, const int *__restrict__ rows, const int *__restrict__ perm, const int *__restrict__ cell_id,
                            PhaseOff P, const int *__restrict__ lead, const long long *__restrict__ upos, int2 *__restrict__ unit) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= ntot || !lead[r]) return;
  const int q = phase_of(P, r);
  const int hi = P.off[q + 1];
  const int iA = rows ? rows[r] : r;
  int iB = -1;
  if (r + 1 < hi) {
    const int cand = rows ? rows[r + 1] : r + 1;
    if (cell_id[perm[cand]] == cell_id[perm[iA]]) iB = cand;
  }
  unit[upos[r]] = make_int2(iA, iB);
}
template <bool TRI>
static __global__ __launch_bounds__(POLAR_BLOCK) void k_dd_units(int nunits, const int2 *__restrict__ unit, const double4 *__restrict__ pos4,
                                                          Box box, CellGrid g, const long long *__restrict__ cell_first, double ddcutsq,
                                                          long long upitch, int *__restrict__ udd_j, int4 *__restrict__ udesc, int qm,
                                                          int dummy, int *__restrict__ overflow, unsigned long long *__restrict__ dd_total) {
  const int lane = threadIdx.x & 63;
  const int u = blockIdx.x * POLAR_ROWS_PER_BLOCK + (threadIdx.x >> 6);
  if (u >= nunits) return;
  const int2 un = unit[u];
  const int iA = un.x, iB = un.y;
  const double4 rA = pos4[iA];
  const double4 rB = pos4[iB >= 0 ? iB : iA];
  // the home cell (both atoms are in it): the +-2 stencil around it reaches a cutoff beyond every point of the cell
  int cc[3];
  {
    double fr3[3];
    frac_coords(box, g.lo, rA.x, rA.y, rA.z, fr3);
#pragma unroll
    for (int k = 0; k < 3; k++) {
      double fr = fr3[k];
      fr -= floor(fr);
      int ck = (int)(fr * g.nc[k]);
      ck = ck >= g.nc[k] ? g.nc[k] - 1 : ck;
      cc[k] = __builtin_amdgcn_readfirstlane(ck);
    }
  }
  const int c0 = cc[0], c1 = cc[1], c2 = cc[2];
  const long long d0 = (long long)u * upitch;
  const unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  const int n0 = g.nc[0], n1 = g.nc[1], n2 = g.nc[2];
  const int zlo = n2 >= 5 ? c2 - 2 : 0, zcnt = n2 >= 5 ? 5 : n2;
  const int ylo = n1 >= 5 ? c1 - 2 : 0, ycnt = n1 >= 5 ? 5 : n1;
  const int nsr = zcnt * ycnt;
  int ra0 = 0, rb0 = 0, ra1 = 0, rb1 = 0;
  if (lane < nsr) {
    const int zz = zlo + lane / ycnt, yy = ylo + lane % ycnt;
    int b2 = zz, b1 = yy;
    bool ok = true;
    if (b2 < 0 || b2 >= n2) { if (!box.periodic[2]) ok = false; b2 = (b2 + n2) % n2; }
    if (b1 < 0 || b1 >= n1) { if (!box.periodic[1]) ok = false; b1 = (b1 + n1) % n1; }
    const int xlo = n0 >= 5 ? c0 - 2 : 0, xhi = n0 >= 5 ? c0 + 2 : n0 - 1;
    if (ok) {
      const long long rowbase = ((long long)b2 * n1 + b1) * n0;
      const int xa = xlo < 0 ? 0 : xlo, xb = xhi >= n0 ? n0 - 1 : xhi;
      ra0 = (int)cell_first[rowbase + xa]; rb0 = (int)cell_first[rowbase + xb + 1];
      if (box.periodic[0]) {
        if (xlo < 0) { ra1 = (int)cell_first[rowbase + xlo + n0]; rb1 = (int)cell_first[rowbase + n0]; }
        else if (xhi >= n0) { ra1 = (int)cell_first[rowbase]; rb1 = (int)cell_first[rowbase + xhi - n0 + 1]; }
      }
    }
  }
  int count = 0, useful = 0;
  bool wrap_lane = false;
  for (int sr = 0; sr < nsr; sr++) {
#pragma unroll
    for (int piece = 0; piece < 2; piece++) {
      const int a = __builtin_amdgcn_readlane(piece ? ra1 : ra0, sr), b = __builtin_amdgcn_readlane(piece ? rb1 : rb0, sr);
      for (int base = a; base < b; base += 64) {
        const int j = base + lane;
        bool inA = false, inB = false;
        if (j < b) {
          const double4 rj = pos4[j];
          if (__double2loint(rj.w)) {   // polarizable
            double ex, ey, ez;
            const bool shA = min_image_rint_w<TRI>(box, rA.x, rA.y, rA.z, rj.x, rj.y, rj.z, ex, ey, ez);
            inA = j != iA && ex * ex + ey * ey + ez * ez < ddcutsq;
            bool shB = false;
            if (iB >= 0) {
              shB = min_image_rint_w<TRI>(box, rB.x, rB.y, rB.z, rj.x, rj.y, rj.z, ex, ey, ez);
              inB = j != iB && ex * ex + ey * ey + ez * ez < ddcutsq;
            }
            wrap_lane |= (inA && shA) || (inB && shB);
          }
        }
        const bool in = inA || inB;
        const unsigned long long m = __ballot(in);
        const int k = count + __popcll(m & below);
        if (in && k < upitch) udd_j[d0 + lp_slot(k, qm)] = (j << 6) | (inA ? 1 : 0) | (inB ? 2 : 0);
        count += __popcll(m);
        useful += __popcll(__ballot(inA)) + __popcll(__ballot(inB));
      }
    }
  }
  const int have = count < upitch ? count : (int)upitch;
  const int padded = (have + 63) & ~63;
  for (int k = have + lane; k < padded; k += 64) udd_j[d0 + lp_slot(k, qm)] = dummy << 6;
  const unsigned long long anywrap = __ballot(wrap_lane);
  if (lane == 0) {
    udesc[u] = make_int4(iA, iB, (padded >> 6) | (anywrap ? 0x40000000 : 0), count);
    if (count > upitch) atomicMax(overflow, count);
    if (useful) atomicAdd(dd_total + (blockIdx.x & 63) * 16, (unsigned long long)useful);
  }
}
