// lab/lists_cluster_rows.hpp -- LAB BUILD ONLY (-DPOLAR_LAB, libpolar_mi355x_lab.so): code that was built, measured and did not become the
// product path.  Included by polar_lists.hpp inside `#ifdef POLAR_LAB`; the product library never sees it.
// No include guard: it is a fragment of polar_lists.hpp, textually in that file's scope.
// ------------------------------------------------------------------------------------------
// Cluster rows of the dipole sweep (k_field_cl).  The sweep is bound by the gather of the neighbour records
// (about 60 GB/s per CU through L1 misses, however many waves or gathers are in flight: profiles/r02_lab_*), so
// rows are grouped: a CLUSTER is up to four polarizable atoms within ~2 A of each other (bonded neighbours, the
// sites of one sorbate molecule), one wave sweeps them together over the UNION of their dd neighbours, and every
// gathered record serves all members.  This kernel builds that union list: one wave per cluster, the same
// half-cutoff cell stencil as k_nl_build widened by the cluster radius (up to +-3 cells), entries = byte offsets of
// the records in the chunked order of lp_slot(), rows padded to whole trips with the dummy record.
//   cnt[c]   entries of the union list          wrapf[c]  a listed pair reaches across a periodic face
//   total    directed (member, neighbour) pairs inside the dd cutoff = what a per-atom list would hold
struct ClusterRows {
  const int4 *members;  // [ncl] s-space indices, -1 padded; member 0 always valid
  int ncl;
};
static __global__ __launch_bounds__(POLAR_BLOCK) void k_cl_build(ClusterRows cl, const double4 *__restrict__ pos4, Box box,
                                                          CellGrid g, const long long *__restrict__ cell_first,
                                                          double ddcutsq, long long pitch, int *__restrict__ cnt,
                                                          int *__restrict__ dd_j, int pad_index,
                                                          int *__restrict__ wrapf, int *__restrict__ overflow,
                                                          unsigned long long *__restrict__ total) {
  const int lane = threadIdx.x & 63;
  const int c = blockIdx.x * POLAR_ROWS_PER_BLOCK + (threadIdx.x >> 6);
  if (c >= cl.ncl) return;
  const int4 mem = cl.members[c];
  const int m0 = __builtin_amdgcn_readfirstlane(mem.x), m1 = __builtin_amdgcn_readfirstlane(mem.y),
            m2 = __builtin_amdgcn_readfirstlane(mem.z), m3 = __builtin_amdgcn_readfirstlane(mem.w);
  double mx[4], my[4], mz[4];
  const int mi[4] = {m0, m1, m2, m3};
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const double4 r = pos4[mi[k] >= 0 ? mi[k] : m0];
    mx[k] = wave_uniform(r.x); my[k] = wave_uniform(r.y); mz[k] = wave_uniform(r.z);
  }
  // bounding sphere around member 0
  double rmax2 = 0.0;
#pragma unroll
  for (int k = 1; k < 4; k++) {
    double ex, ey, ez;
    min_image_rint(box, mx[0], my[0], mz[0], mx[k], my[k], mz[k], ex, ey, ez);
    const double d2 = ex * ex + ey * ey + ez * ez;
    rmax2 = (mi[k] >= 0 && d2 > rmax2) ? d2 : rmax2;
  }
  const double reach = wave_uniform(sqrt(ddcutsq) + sqrt(rmax2) + 1e-6);
  const double reach2 = reach * reach;
  int cc[3], W[3];
  double uu[3], edge[3];
  {
    double fr3[3];
    frac_coords(box, g.lo, mx[0], my[0], mz[0], fr3);  // the arithmetic of cell_of (orthogonal boxes only in cluster mode)
#pragma unroll
    for (int k = 0; k < 3; k++) {
      double fr = fr3[k];
      fr -= floor(fr);
      const double t = fr * g.nc[k];
      int ck = (int)t;
      ck = ck >= g.nc[k] ? g.nc[k] - 1 : ck;
      cc[k] = __builtin_amdgcn_readfirstlane(ck);
      uu[k] = wave_uniform(t - ck); edge[k] = wave_uniform(box.prd[k] / g.nc[k]);
      int w = (int)ceil(reach / edge[k]);
      W[k] = __builtin_amdgcn_readfirstlane(w < 1 ? 1 : (w > 3 ? 3 : w));  // the host keeps cluster radii within one cell edge
    }
  }
  const int c0 = cc[0], c1 = cc[1], c2 = cc[2];
  const long long row0 = (long long)c * pitch;
  const unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  const int n0 = g.nc[0], n1 = g.nc[1], n2 = g.nc[2];
  const bool f2 = n2 >= 2 * W[2] + 1, f1 = n1 >= 2 * W[1] + 1, f0 = n0 >= 2 * W[0] + 1;
  const int zlo = f2 ? c2 - W[2] : 0, zcnt = f2 ? 2 * W[2] + 1 : n2;
  const int ylo = f1 ? c1 - W[1] : 0, ycnt = f1 ? 2 * W[1] + 1 : n1;
  const int nsr = zcnt * ycnt;  // <= 49 stencil rows of cells: lane L prepares row L
  int ra0 = 0, rb0 = 0, ra1 = 0, rb1 = 0;
  if (lane < nsr) {
    const int zz = zlo + lane / ycnt, yy = ylo + lane % ycnt;
    int b2 = zz, b1 = yy;
    bool ok = true;
    if (b2 < 0 || b2 >= n2) { if (!box.periodic[2]) ok = false; b2 = (b2 + n2) % n2; }
    if (b1 < 0 || b1 >= n1) { if (!box.periodic[1]) ok = false; b1 = (b1 + n1) % n1; }
    double dzmin = 0.0, dymin = 0.0;
    if (f2) { const int d = zz - c2; dzmin = d > 0 ? (d - uu[2]) * edge[2] : (d < 0 ? (uu[2] - (d + 1)) * edge[2] : 0.0); }
    if (f1) { const int d = yy - c1; dymin = d > 0 ? (d - uu[1]) * edge[1] : (d < 0 ? (uu[1] - (d + 1)) * edge[1] : 0.0); }
    const double rem2 = reach2 - dzmin * dzmin - dymin * dymin;
    if (rem2 < 0.0) ok = false;
    int xlo = f0 ? c0 - W[0] : 0, xhi = f0 ? c0 + W[0] : n0 - 1;
    if (ok && f0) {
      const double xr = sqrt(rem2) / edge[0];
      int lo_off = (int)floor(uu[0] - xr), hi_off = (int)floor(uu[0] + xr);
      lo_off = lo_off < -W[0] ? -W[0] : lo_off; hi_off = hi_off > W[0] ? W[0] : hi_off;
      xlo = c0 + lo_off; xhi = c0 + hi_off;
    }
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
  int count = 0;
  unsigned long long anywrap = 0ull, npairs = 0ull;
  for (int sr = 0; sr < nsr; sr++) {
#pragma unroll
    for (int piece = 0; piece < 2; piece++) {
      const int a = __builtin_amdgcn_readlane(piece ? ra1 : ra0, sr), b = __builtin_amdgcn_readlane(piece ? rb1 : rb0, sr);
      for (int base = a; base < b; base += 64) {
        const int j = base + lane;
        bool in = false, wr = false;
        int hits = 0;
        if (j < b && j != m0 && j != m1 && j != m2 && j != m3) {
          const double4 rj = pos4[j];
          if (__double2loint(rj.w)) {  // polarizable
#pragma unroll
            for (int k = 0; k < 4; k++) {
              double ex, ey, ez;
              min_image_rint(box, mx[k], my[k], mz[k], rj.x, rj.y, rj.z, ex, ey, ez);
              const bool hit = mi[k] >= 0 && (ex * ex + ey * ey + ez * ez) < ddcutsq;
              hits += hit;
              wr |= hit && (ex != mx[k] - rj.x || ey != my[k] - rj.y || ez != mz[k] - rj.z);
            }
            in = hits > 0;
          }
        }
        const unsigned long long m_in = __ballot(in);
        anywrap |= __ballot(wr);
        const int kd = count + __popcll(m_in & below);
        if (in && kd < pitch) dd_j[row0 + lp_slot(kd)] = j << 6;
        count += __popcll(m_in);
        npairs += (unsigned long long)hits;
      }
    }
  }
  const int have = count < pitch ? count : (int)pitch;
  const int padded = (have + 63) & ~63;
  for (int k = have + lane; k < padded; k += 64) dd_j[row0 + lp_slot(k)] = pad_index << 6;
  // directed pairs of this cluster (lane-private counts summed over the wave)
  double np = (double)npairs;
  np = wave_sum(np);
  if (lane == 0) {
    cnt[c] = count;
    wrapf[c] = anywrap != 0ull;
    if (count > pitch) atomicMax(overflow, count);
    if (np > 0.0) atomicAdd(total + (blockIdx.x & 63) * 16, (unsigned long long)np);
  }
}

