// lab/tile_kernels.hpp -- LAB BUILD ONLY (-DPOLAR_LAB, libpolar_mi355x_lab.so): code that was built, measured and did not become the
// product path.  Included by polar_tiles.hpp inside `#ifdef POLAR_LAB`; the product library never sees it.
// No include guard: it is a fragment of polar_tiles.hpp, textually in that file's scope.
// AtomRec (both buffers hold the same initial dipoles) -> sweep records, and the solved dipoles back
static __global__ void k_srec_pack(int n, const AtomRec *__restrict__ r, SRec *__restrict__ s0, SRec *__restrict__ s1) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const AtomRec a = r[i];
  SRec s;
  s.x = a.x; s.mx = a.mx; s.y = a.y; s.my = a.my; s.z = a.z; s.mz = a.mz;
  s0[i] = s;
  if (s1) s1[i] = s;
}
static __global__ void k_srec_unpack(int n, const Scal *scal, const SRec *__restrict__ s0, const SRec *__restrict__ s1,
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
static __global__ __launch_bounds__(256) void k_tile_build(CellGrid g, Box box, const double4 *__restrict__ pos4,
                                                    const long long *__restrict__ cell_first,
                                                    const int *__restrict__ npol, const int *__restrict__ perm,
                                                    int own_lo, int own_hi, double ddcutsq, double colordistsq, int subcap,
                                                    int sw0, int sw1, int sw2, int un_pitch, int *__restrict__ un_j, long long pitch16,
                                                    unsigned short *__restrict__ dd16, TileHdr *__restrict__ hdr,
                                                    TileRowEnt *__restrict__ trow, const AtomRec *__restrict__ rec,
                                                    int *__restrict__ flags, unsigned long long *__restrict__ dd_total) {
  extern __shared__ __attribute__((aligned(16))) char tb_lds[];
  double *ux = reinterpret_cast<double *>(tb_lds), *uy = ux + un_pitch, *uz = uy + un_pitch;
  int *cnt = reinterpret_cast<int *>(uz + un_pitch);  // [128] candidates per stencil cell; later: accepted per (round, wave)
  int *off = cnt + 128;                               // [128] prefix of the candidates
  int *sca = off + 128;                               // [128] first record of a stencil cell's polarizable run
  int *scc = sca + 128;                               // [128] its periodic-image code
  int *rowT = scc + 128;                              // [MAXROWS] trips of local row m, -1: not a row of this handle
  int *col = rowT + POLAR_TILE_MAXROWS;               // [MAXROWS] sub-phase
  double *bb = reinterpret_cast<double *>(col + POLAR_TILE_MAXROWS);  // [4 waves][6] partial, then [6] bounding box of the rows
  int *misc = reinterpret_cast<int *>(bb + 24);       // [12] U, rows, shifted entries, selfbase, rows per wave [4..7], candidates [8]
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
  const unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  // stencil: +-sw_k cells in dimension k (2 for cells of half a cutoff, 1 for cells of a whole one), at most 5 x 5 x 5
  const int nx = 2 * sw0 + 1, ny = 2 * sw1 + 1, nq = nx * ny * (2 * sw2 + 1), qh = sw0 + nx * (sw1 + ny * sw2);
  // ---- round 1 of memory requests, all at once: the stencil cells (thread q: cell q), and the tile's own atoms
  //      (thread m: is atom m a row of this handle, where is it).  A workgroup that walked the stencil cell by cell paid two
  //      dependent memory latencies per cell and pass: 250 of them, a millisecond per step at 135k atoms.
  {
    int a = 0, np = 0, code = 13;
    if (tid < nq) {
      int bq[3] = {c0 + tid % nx - sw0, c1 + (tid / nx) % ny - sw1, c2 + tid / (nx * ny) - sw2};
      int sh[3] = {0, 0, 0};
      const int nn[3] = {n0, n1, n2};
      bool ok = true;
#pragma unroll
      for (int k = 0; k < 3; k++) {
        if (bq[k] < 0) { ok = ok && box.periodic[k]; bq[k] += nn[k]; sh[k] = -1; }
        else if (bq[k] >= nn[k]) { ok = ok && box.periodic[k]; bq[k] -= nn[k]; sh[k] = 1; }
        ok = ok && bq[k] >= 0 && bq[k] < nn[k];  // (a dimension with fewer than 3 cells cannot be periodic in list mode)
      }
      if (ok) {
        const int cj = (bq[2] * n1 + bq[1]) * n0 + bq[0];
        a = (int)cell_first[cj]; np = npol[cj];
        code = (sh[0] + 1) + 3 * (sh[1] + 1) + 9 * (sh[2] + 1);
      }
    }
    bool row = false;
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    if (tid < P) {
      const int o = perm[r0 + tid];
      row = o >= own_lo && o < own_hi;
      if (row) {
        const double4 p = pos4[r0 + tid];
        lo[0] = hi[0] = p.x; lo[1] = hi[1] = p.y; lo[2] = hi[2] = p.z;
      }
    }
    if (tid < 128) { cnt[tid] = np; sca[tid] = a; scc[tid] = code; }
    if (tid < POLAR_TILE_MAXROWS) { rowT[tid] = row ? 0 : -1; col[tid] = -1; }
#pragma unroll
    for (int k = 0; k < 3; k++) { lo[k] = wave_min(lo[k]); hi[k] = -wave_min(-hi[k]); }
    const int rows = __popcll(__ballot(row));
    if (lane == 0) {
      for (int k = 0; k < 3; k++) { bb[6 * wv + k] = lo[k]; bb[6 * wv + 3 + k] = hi[k]; }
      misc[4 + wv] = rows;
    }
  }
  __syncthreads();
  const int nrow_tile = misc[4] + misc[5] + misc[6] + misc[7];
  if (nrow_tile == 0) {  // no row of this handle in the cell
    if (tid == 0) { H->r0 = r0; H->nrows = 0; H->U = 0; H->nsub = 0; }
    return;
  }
  double b0 = bb[0], b1 = bb[1], b2 = bb[2], b3 = bb[3], b4 = bb[4], b5 = bb[5];
#pragma unroll
  for (int w = 1; w < 4; w++) {
    b0 = fmin(b0, bb[6 * w]); b1 = fmin(b1, bb[6 * w + 1]); b2 = fmin(b2, bb[6 * w + 2]);
    b3 = fmax(b3, bb[6 * w + 3]); b4 = fmax(b4, bb[6 * w + 4]); b5 = fmax(b5, bb[6 * w + 5]);
  }
  if (wv == 0) {  // exclusive prefix of the 125 candidate counts
    const int v0 = cnt[lane], v1 = lane + 64 < nq ? cnt[lane + 64] : 0;
    int i0 = v0, i1 = v1;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int u0 = __shfl_up(i0, o, 64), u1 = __shfl_up(i1, o, 64);
      if (lane >= o) { i0 += u0; i1 += u1; }
    }
    const int t0 = __shfl(i0, 63, 64), t1 = __shfl(i1, 63, 64);
    off[lane] = i0 - v0;
    off[64 + lane] = t0 + i1 - v1;  // (entries nq .. 127: the total)
    if (lane == 0) { misc[8] = t0 + t1; misc[0] = 0; misc[1] = nrow_tile; misc[2] = 0; misc[3] = 0; }
  }
  __syncthreads();
  // ---- round 2: the candidates, flattened (candidate idx of the concatenated runs -> its cell by bisection of the prefix),
  //      NBC per thread with all their position loads in flight at once; accepted = within the cutoff of the rows' bounding
  //      box (the home cell is taken whole: row m then sits at selfbase + m).  Ordered compaction: counts per (round, wave),
  //      one prefix, then every thread writes its accepted candidates -- the union keeps candidate order, run to run.
  const int C = misc[8];
  constexpr int NBC = 8;
  for (int base = 0; base < C; base += NBC * 256) {
    int ej[NBC];
    double4 pp[NBC];
#pragma unroll
    for (int u = 0; u < NBC; u++) {
      const int idx = base + u * 256 + tid;
      int q = 0;
      if (idx < C) {
#pragma unroll
        for (int stp = 64; stp > 0; stp >>= 1) if (q + stp < nq && off[q + stp] <= idx) q += stp;  // last q with off[q] <= idx
      }
      const int j = idx < C ? sca[q] + idx - off[q] : 0;
      ej[u] = idx < C ? (j | (scc[q] << 26)) : -1;
      pp[u] = pos4[j];
    }
    unsigned inmask = 0u;
#pragma unroll
    for (int u = 0; u < NBC; u++) {
      bool in = false;
      if (ej[u] >= 0) {
        const int code = ej[u] >> 26;
        double sx, sy, sz;
        tile_shift(box, code, sx, sy, sz);
        pp[u].x += sx; pp[u].y += sy; pp[u].z += sz;
        const double ex = fmax(fmax(b0 - pp[u].x, pp[u].x - b3), 0.0), ey = fmax(fmax(b1 - pp[u].y, pp[u].y - b4), 0.0),
                     ez = fmax(fmax(b2 - pp[u].z, pp[u].z - b5), 0.0);
        const int idx = base + u * 256 + tid;
        const bool home = idx >= off[qh] && idx < off[qh + 1];
        in = home || (ex * ex + ey * ey + ez * ez) < ddcutsq;
      }
      const unsigned long long mk = __ballot(in);
      if (in) inmask |= 1u << u;
      if (lane == 0) cnt[u * 4 + wv] = __popcll(mk);
    }
    __syncthreads();
    if (wv == 0) {  // prefix over (round, wave) in candidate order, on top of what earlier chunks accepted
      const int v = lane < NBC * 4 ? cnt[lane] : 0;
      int inc = v;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int up = __shfl_up(inc, o, 64);
        if (lane >= o) inc += up;
      }
      const int ubase = misc[0];
      if (lane < NBC * 4) cnt[32 + lane] = ubase + inc - v;
      if (lane == 63) misc[0] = ubase + inc;
    }
    __syncthreads();
    if (misc[0] + 1 > un_pitch) break;  // (uniform) does not fit: reported below
#pragma unroll
    for (int u = 0; u < NBC; u++) {
      const bool in = (inmask >> u) & 1u;
      const unsigned long long mk = __ballot(in);
      if (in) {
        const int k = cnt[32 + u * 4 + wv] + __popcll(mk & below);
        ux[k] = pp[u].x; uy[k] = pp[u].y; uz[k] = pp[u].z;
        un_j[(size_t)c * un_pitch + k] = ej[u];
        if ((ej[u] >> 26) != 13) misc[2] = 1;  // (every writer stores the same value)
        if (base + u * 256 + tid == off[qh]) misc[3] = k;  // first atom of the home cell
      }
    }
    __syncthreads();
  }
  const int U = misc[0];
  if (U + 1 > un_pitch) {
    if (tid == 0) { atomicMax(flags + 5, U + 1); H->r0 = r0; H->nrows = 0; H->U = 0; H->nsub = 0; }
    return;
  }
  const int selfbase = misc[3];
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
  int *filled = cnt;  // rows per sub-phase so far (the stencil counts are no longer needed)
  const int cap = subcap * ((misc[1] + 8 * subcap - 1) / (8 * subcap));  // rows per sub-phase: at most 8 sub-phases' worth
  if (lane < POLAR_TILE_MAXSUB) filled[lane] = 0;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
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
    // ... and no sub-phase takes more rows than the sweep's workgroup has waves (times a whole number for crowded cells):
    // every wave then has the same number of rows per sub-phase -- rows of a tile are about equally long -- and the waves
    // reach the barrier together
    if (lane < POLAR_TILE_MAXSUB && filled[lane] >= cap) used |= 1u << lane;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) used |= (unsigned)__shfl_xor((int)used, o, 64);
    int cm = __ffs((int)~used) - 1;
    if (cm >= POLAR_TILE_MAXSUB) { if (lane == 0) atomicMax(flags + 7, cm + 1); cm = POLAR_TILE_MAXSUB - 1; }
    if (lane == 0) { col[m] = cm; filled[cm] += 1; }
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
      if (is) {
        TileRowEnt e;
        e.mT = m | (rowT[m] << 16); e.self = selfbase + m; e.alpha = rec[r0 + m].a;
        trow[r0 + k + __popcll(mk & below)] = e;
      }
      k += __popcll(mk);
    }
  }
  if (lane == 0) {
    H->sub_off[nsub] = k;
    H->r0 = r0; H->nrows = k; H->U = U; H->nsub = nsub | (misc[2] ? 0x100 : 0);
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


// workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for the wave's global loads and stores
// (vmcnt(0)): here that would expose, at every sub-phase, the latency of the index stream requested for the NEXT row.
// Other waves read this wave's dipole out of LDS, never out of memory, inside a launch.
__device__ __forceinline__ void tile_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// What a wave asks for one row ahead: the first two chunks of the row's index stream (16 trips) and the row's E_static.
// The loads are inline assembly and the waits are counted by hand: hipcc's own waits lose count at a loop header and
// would drain the queue -- the next row's prefetch included -- before the first trip of every row.
typedef unsigned tile_u4 __attribute__((ext_vector_type(4)));
struct TileRow {
  tile_u4 Ja, Jb;
  double ef, alpha;
  const tile_u4 *pc;
  int i, T;
  unsigned self;
};
__device__ __forceinline__ tile_u4 tile_ld128(const void *p) {
  tile_u4 v;
  asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
  return v;
}
__device__ __forceinline__ double tile_ld64(const void *p) {
  double v;
  asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
  return v;
}
// THREE vector-memory instructions, always: tile_settle<3> below leaves exactly these in flight
__device__ __forceinline__ TileRow tile_prefetch(const char *lds, int r0, int k, const unsigned short *dd16, long long pitch16,
                                                 const double *ef, int lane) {
  TileRow R;
  const TileRowEnt tr = *reinterpret_cast<const TileRowEnt *>(lds + POLAR_TILE_LDS_ROWS + 16 * k);  // same address in every lane
  const int m = __builtin_amdgcn_readfirstlane(tr.mT & 0xFFFF);
  R.T = __builtin_amdgcn_readfirstlane(tr.mT >> 16);
  R.self = (unsigned)__builtin_amdgcn_readfirstlane(tr.self);
  R.alpha = tr.alpha;
  R.i = r0 + m;
  R.pc = reinterpret_cast<const tile_u4 *>(dd16 + (size_t)R.i * pitch16) + lane;
  R.Ja = tile_ld128(R.pc);
  R.Jb = tile_ld128(R.pc + 64);  // (rows of one chunk: the next row's first chunk, or the slack behind the table)
  R.ef = tile_ld64(ef + 3 * (size_t)R.i + (lane < 3 ? lane : 2));
  return R;
}
// everything but the N youngest vector-memory operations of this wave has completed; the row's prefetched values are
// tied to the wait so that no use of them can be scheduled above it
template <int N>
__device__ __forceinline__ void tile_settle(TileRow &R) {
  asm volatile("s_waitcnt vmcnt(%3)" : "+v"(R.Ja), "+v"(R.Jb), "+v"(R.ef) : "n"(N) : "memory");
}

// The trips of one row, two at a time: TWO independent dependency chains per step (a four-wave workgroup's LDS footprint leaves
// a SIMD two waves, so the latency of the ~45-deep FP64 chain of a pair has to be covered inside the wave), and two register
// sets used alternately: a step computes out of one while the records of the next step land in the other.
template <int DAMP>
__device__ __forceinline__ void tile_row_pairs2(const TileRow &R, const char *recs, double xi, double yi, double zi, double pd,
                                                const ExpCoef &K, double &ax, double &ay, double &az) {
  const int T = R.T;
  if (T <= 0) return;
  double bx = 0.0, by = 0.0, bz = 0.0;
  double2 PA0, PB0, PC0, PA1, PB1, PC1, QA0, QB0, QC0, QA1, QB1, QC1;
  tile_read(recs, R.Ja.x & 0xFFFFu, PA0, PB0, PC0);
  if (T > 1) tile_read(recs, R.Ja.x >> 16, PA1, PB1, PC1);
  const int NC = (T + 7) >> 3;
  tile_u4 J = R.Ja, Jn = R.Jb;
#define POLAR_TILE_STEP(QQ, XA0, XB0, XC0, XA1, XB1, XC1, YA0, YB0, YC0, YA1, YB1, YC1, NEXT0, NEXT1) \
  {                                                                                                   \
    const int t = t0 + 2 * (QQ);                                                                      \
    if (t + 2 < T) tile_read(recs, (NEXT0), YA0, YB0, YC0);                                           \
    if (t + 3 < T) tile_read(recs, (NEXT1), YA1, YB1, YC1);                                           \
    if (t + 1 < T) {                                                                                  \
      tile_pair<DAMP>(xi, yi, zi, XA0, XB0, XC0, pd, K, ax, ay, az);                                  \
      tile_pair<DAMP>(xi, yi, zi, XA1, XB1, XC1, pd, K, bx, by, bz);                                  \
    } else {                                                                                          \
      tile_pair<DAMP>(xi, yi, zi, XA0, XB0, XC0, pd, K, ax, ay, az);                                  \
    }                                                                                                 \
    if (t + 2 >= T) break;                                                                            \
  }
  for (int cc = 0; cc < NC; cc++) {
    const int t0 = 8 * cc;
    tile_u4 Jf = Jn;  // chunk cc + 2 (rows longer than 16 trips only): requested here, needed two chunks on
    if (cc + 2 < NC) Jf = tile_ld128(R.pc + 64 * (cc + 2));
    POLAR_TILE_STEP(0, PA0, PB0, PC0, PA1, PB1, PC1, QA0, QB0, QC0, QA1, QB1, QC1, J.y & 0xFFFFu, J.y >> 16)
    POLAR_TILE_STEP(1, QA0, QB0, QC0, QA1, QB1, QC1, PA0, PB0, PC0, PA1, PB1, PC1, J.z & 0xFFFFu, J.z >> 16)
    POLAR_TILE_STEP(2, PA0, PB0, PC0, PA1, PB1, PC1, QA0, QB0, QC0, QA1, QB1, QC1, J.w & 0xFFFFu, J.w >> 16)
    if (cc + 2 < NC) asm volatile("s_waitcnt vmcnt(0)" : "+v"(Jf) : : "memory");  // (long rows: the chunk requested above)
    POLAR_TILE_STEP(3, QA0, QB0, QC0, QA1, QB1, QC1, PA0, PB0, PC0, PA1, PB1, PC1, Jn.x & 0xFFFFu, Jn.x >> 16)
    J = Jn; Jn = Jf;
  }
#undef POLAR_TILE_STEP
  ax += bx; ay += by; az += bz;
}
// the same, one trip per step (eight-wave workgroups: four waves per SIMD cover the chain's latency, registers are scarce)
template <int DAMP>
__device__ __forceinline__ void tile_row_pairs1(const TileRow &R, const char *recs, double xi, double yi, double zi, double pd,
                                                const ExpCoef &K, double &ax, double &ay, double &az) {
  const int T = R.T;
  if (T <= 0) return;
  double2 A0, B0, C0, A1, B1, C1;
  tile_read(recs, R.Ja.x & 0xFFFFu, A0, B0, C0);
  const int NC = (T + 7) >> 3;
  tile_u4 J = R.Ja, Jn = R.Jb;
#define POLAR_TILE_TRIP(UU, PA, PB, PC, QA, QB, QC, NEXTPOS)               \
  {                                                                        \
    const bool more = t0 + (UU) + 1 < T;                                   \
    if (more) tile_read(recs, (NEXTPOS), QA, QB, QC);                      \
    tile_pair<DAMP>(xi, yi, zi, PA, PB, PC, pd, K, ax, ay, az);            \
    if (!more) break;                                                      \
  }
  for (int cc = 0; cc < NC; cc++) {
    const int t0 = 8 * cc;
    tile_u4 Jf = Jn;
    if (cc + 2 < NC) Jf = tile_ld128(R.pc + 64 * (cc + 2));
    POLAR_TILE_TRIP(0, A0, B0, C0, A1, B1, C1, J.x >> 16)
    POLAR_TILE_TRIP(1, A1, B1, C1, A0, B0, C0, J.y & 0xFFFFu)
    POLAR_TILE_TRIP(2, A0, B0, C0, A1, B1, C1, J.y >> 16)
    POLAR_TILE_TRIP(3, A1, B1, C1, A0, B0, C0, J.z & 0xFFFFu)
    POLAR_TILE_TRIP(4, A0, B0, C0, A1, B1, C1, J.z >> 16)
    POLAR_TILE_TRIP(5, A1, B1, C1, A0, B0, C0, J.w & 0xFFFFu)
    POLAR_TILE_TRIP(6, A0, B0, C0, A1, B1, C1, J.w >> 16)
    if (cc + 2 < NC) asm volatile("s_waitcnt vmcnt(0)" : "+v"(Jf) : : "memory");
    POLAR_TILE_TRIP(7, A1, B1, C1, A0, B0, C0, Jn.x & 0xFFFFu)
    J = Jn; Jn = Jf;
  }
#undef POLAR_TILE_TRIP
}

// NW = waves of the workgroup (4: two workgroups per CU leave a SIMD two waves; 8: four)
template <int EP, int DAMP, bool DET, int NW>
static __global__ __launch_bounds__(64 * NW, NW == 8 ? 4 : 2) void k_field_tile(TileLaunch L, const TileHdr *__restrict__ hdr, const TileRowEnt *__restrict__ trow,
                                                    const int *__restrict__ un_j, int un_pitch,
                                                    const unsigned short *__restrict__ dd16, long long pitch16, SRec *s0,
                                                    SRec *s1, double *pend, const double *__restrict__ ef, Box box, double pd,
                                                    ExpCoef K, const Scal *scal, double *__restrict__ slots, int nrec) {
  extern __shared__ __attribute__((aligned(16))) char tl_lds[];
  char *const recs = tl_lds + POLAR_TILE_LDS_REC;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), nwv = NW;
  const int ntile = L.count[0] * L.count[1] * L.count[2];
  const int lb = xcd_block(blockIdx.x, ntile);
  if (lb < 0) return;
  const int i0 = lb % L.count[0], i1 = (lb / L.count[0]) % L.count[1], i2 = lb / (L.count[0] * L.count[1]);
  const int c = ((L.start[2] + L.stride[2] * i2) * L.nc[1] + (L.start[1] + L.stride[1] * i1)) * L.nc[0] + L.start[0] + L.stride[0] * i0;
  // ---- prologue: THREE memory latencies for everything a tile needs before its first trip.
  //   1: the tile header (scalar loads) and, not waiting for it, this thread's entry words of the union list;
  //   2: the row table (into LDS: every later per-row decision then comes out of LDS, not out of a chain of dependent memory
  //      reads) and this wave's first row;
  //   3: the union records as LDS-DMA -- piece g = 3 e + p of entry e (16 bytes: {position_p, dipole_p}) goes to byte 16 g of
  //      the record area, no destination registers, lane-linear landing = piece order -- and the first row's index stream.
  // The DMA is inline assembly: beside a compiler-issued global_load_lds hipcc drains the whole queue (vmcnt(0)) before every
  // use of an ordinary load's result, i.e. once per piece.  Entries of another periodic image get their lattice vector added
  // in place afterwards by the thread that fetched them (tiles at a periodic face only).
  const int *uj = un_j + (size_t)c * un_pitch;
  const int step = 64 * NW;
  constexpr int NB = 16;  // pieces per thread and round: 4,096 (8,192) pieces = 1,365 (2,730) records per round of 256 (512) threads
  int ent[NB];
#pragma unroll
  for (int u = 0; u < NB; u++) {
    const int e = (u * step + tid) / 3;
    ent[u] = uj[e < un_pitch ? e : un_pitch - 1];
  }
  const TileHdr *H = hdr + c;
  const int nrows = H->nrows;
  if (nrows == 0) return;
  if (scal->done) return;
  const int cur = EP == EP_JACOBI ? scal->cur : 0;
  const SRec *src = (EP == EP_JACOBI && cur) ? s1 : s0;
  SRec *dst = (EP == EP_JACOBI) ? (cur ? s0 : s1) : s0;
  const int U = H->U, r0 = H->r0, hsub = H->nsub;
  const int nsub = EP == EP_JACOBI ? 1 : (hsub & 0xFF);
  const int end0 = EP == EP_JACOBI ? nrows : H->sub_off[1];  // rows of the first sub-phase: wave w starts with row w of it
  if (tid <= POLAR_TILE_MAXSUB) reinterpret_cast<int *>(tl_lds)[4 + tid] = H->sub_off[tid];
  for (int k = tid; k < nrows; k += step)
    *reinterpret_cast<TileRowEnt *>(tl_lds + POLAR_TILE_LDS_ROWS + 16 * k) = trow[r0 + k];
  TileRow R{};
  const bool early = wv < end0;  // (wave-uniform) this wave's first row is row wv: ask for it beside the records
  TileRowEnt tr0{};
  if (early) tr0 = trow[r0 + wv];
  {
    const char *sb = reinterpret_cast<const char *>(src);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char *)recs;
    const int np = 3 * U;
    for (int gb = 0; gb < np; gb += NB * step) {
      if (gb > 0) {  // (unions beyond one round: the next entry words)
#pragma unroll
        for (int u = 0; u < NB; u++) {
          const int gq = gb + u * step + tid;
          ent[u] = uj[(gq < np ? gq : 0) / 3];
        }
      }
      // (the entry words pass through an empty asm statement: the compiler then waits for them HERE, and places none of
      //  its counted waits -- which do not know of the DMAs -- between the DMA instructions below)
#pragma unroll
      for (int u = 0; u < NB; u++) asm volatile("" : "+v"(ent[u]));
#pragma unroll
      for (int u = 0; u < NB; u++) {
        const int gw = __builtin_amdgcn_readfirstlane(gb + u * step + (tid & ~63));  // first piece of this wave's instruction
        const int gq = gw + lane, p = gq % 3;
        if (gw < np) {  // wave-uniform; lanes past the end fetch a valid piece into the slack behind the dummy
          // (the entry words were requested before U was known: beyond the list they are whatever the table held)
          // Round 3's recorded abort (gpurun_out/r3_tile2_tests.log): a version of this line without the `gq < np` guard let the
          // lanes past a tile's list turn whatever the table held behind it -- words of an earlier, larger step, or of a fresh
          // hipMalloc -- into record numbers of up to 2^26: a 3 GB offset, a global read far outside the record table.  The guard
          // keeps stale words out; the clamp to the table (nrec = the dummy record, the last one) makes ANY word harmless.
          unsigned recno = gq < np ? (unsigned)(ent[u] & POLAR_TILE_RECMASK) : 0u;
          recno = recno < (unsigned)nrec ? recno : (unsigned)nrec;
          const unsigned voff = recno * 48u + (unsigned)p * 16u;
          lpa_dma(sb, voff, lds0 + (unsigned)gw * 16u);
        }
      }
      if (gb == 0 && early) {  // the first row's index stream and E_static travel with the records
        asm volatile("" : "+v"(tr0.mT), "+v"(tr0.self), "+v"(tr0.alpha));
        R.T = __builtin_amdgcn_readfirstlane(tr0.mT >> 16);
        R.self = (unsigned)__builtin_amdgcn_readfirstlane(tr0.self);
        R.alpha = tr0.alpha;
        R.i = r0 + __builtin_amdgcn_readfirstlane(tr0.mT & 0xFFFF);
        R.pc = reinterpret_cast<const tile_u4 *>(dd16 + (size_t)R.i * pitch16) + lane;
        R.Ja = tile_ld128(R.pc);
        R.Jb = tile_ld128(R.pc + 64);
        R.ef = tile_ld64(ef + 3 * (size_t)R.i + (lane < 3 ? lane : 2));
      }
      if (hsub >> 8) {  // some entry of this tile is another periodic image: its lattice vector goes onto the position words.
        // Every thread patches the pieces it fetched itself (their entry words are still in registers; a loop that read them
        // again cost a tile fifteen dependent memory round trips), once its own DMAs have landed.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int u = 0; u < NB; u++) {
          const int gq = gb + u * step + tid, code = ent[u] >> 26;
          if (gq < np && code != 13) {
            const int p = gq % 3;
            const int s0_ = code % 3 - 1, s1_ = (code / 3) % 3 - 1, s2_ = code / 9 - 1;
            const double sh = p == 0 ? s0_ * box.prd[0] + s1_ * box.xy + s2_ * box.xz : (p == 1 ? s1_ * box.prd[1] + s2_ * box.yz : s2_ * box.prd[2]);
            *reinterpret_cast<double *>(recs + (size_t)gq * 16) += sh;
          }
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(R.Ja), "+v"(R.Jb), "+v"(R.ef) : : "memory");
    __syncthreads();
    if (tid < 3) *reinterpret_cast<double2 *>(recs + (size_t)U * 48 + tid * 16) = make_double2(0.0, 0.0);  // the dummy (overhang of the last DMA block lands here first)
  }
  __syncthreads();
  // ---- the rows.  Wave w takes rows w, w + nwv, ... of every sub-phase, in sub-phase order; while it works on a row it
  //      has the NEXT row's index stream and E_static in flight (a row is ~9 trips = ~1 us: one memory latency).
  const int *soff = reinterpret_cast<const int *>(tl_lds) + 4;  // sub_off[0 .. nsub] (LDS copy)
  auto first_row = [&](int sp) { return EP == EP_JACOBI ? wv : soff[sp] + wv; };
  auto end_row = [&](int sp) { return EP == EP_JACOBI ? nrows : soff[sp + 1]; };
  auto settle = [&](int &sp, int &k) {  // (sp, k) -> the first existing row at or after it in this wave's sequence
    while (sp < nsub && k >= end_row(sp)) { sp++; if (sp < nsub) k = first_row(sp); }
  };
  // leaving sub-phase `from` for `to` (or for the end): the barriers in between; DET commits this wave's rows of `from`
  // into the LDS copy between two barriers (every row of the sub-phase has read what it needed; then everybody sees it)
  auto cross = [&](int from, int to) {
    if (EP == EP_JACOBI) return;
    for (int s = from; s < to; s++) {
      if (DET) {
        tile_barrier();
        if (s == from)
          for (int k = first_row(s); k < end_row(s); k += nwv) {
            const TileRowEnt tr = *reinterpret_cast<const TileRowEnt *>(tl_lds + POLAR_TILE_LDS_ROWS + 16 * k);
            if (lane < 3)
              *reinterpret_cast<double *>(recs + (unsigned)tr.self * 48u + lane * 16 + 8) = pend[3 * (size_t)(r0 + (tr.mT & 0xFFFF)) + lane];
          }
      }
      tile_barrier();
    }
  };
  double chg = 0.0;
  int sp = 0, k = first_row(0);
  settle(sp, k);
  int at = 0;  // sub-phase whose start this wave has reached (barriers passed)
  if (sp < nsub && !early) { R = tile_prefetch(tl_lds, r0, k, dd16, pitch16, ef, lane); tile_settle<0>(R); }  // (a wave whose first row is not row wv of sub-phase 0)
  while (sp < nsub) {
    cross(at, sp);
    at = sp;
    int sp2 = sp, k2 = k + nwv;
    settle(sp2, k2);
    const bool has_next = sp2 < nsub;
    TileRow N{};
    if (has_next) N = tile_prefetch(tl_lds, r0, k2, dd16, pitch16, ef, lane);  // in flight while this row is computed
    // the row atom: position out of the staged copy (same in every lane -> scalar registers), old dipole in lanes 0..2
    const char *sp_ = recs + R.self * 48u;
    const double xi = wave_uniform(*reinterpret_cast<const double *>(sp_)),
                 yi = wave_uniform(*reinterpret_cast<const double *>(sp_ + 16)),
                 zi = wave_uniform(*reinterpret_cast<const double *>(sp_ + 32));
    const double mu_old = *reinterpret_cast<const double *>(sp_ + (lane < 3 ? lane : 2) * 16 + 8);
    double ax = 0.0, ay = 0.0, az = 0.0;
    if (NW == 4) tile_row_pairs2<DAMP>(R, recs, xi, yi, zi, pd, K, ax, ay, az);
    else tile_row_pairs1<DAMP>(R, recs, xi, yi, zi, pd, K, ax, ay, az);
    // the three wave sums in one butterfly (lp_finish): lanes 0, 1, 2 end with E_x, E_y, E_z of the row
    const double v = cl_reduce3(ax, ay, az, lane);
    if (lane < 3) {
      const double mu_new = R.alpha * (R.ef + v);  // PS.cpp:1170-1180
      const double d = mu_new - mu_old;
      chg = fma(d, d, chg);
      if (DET) {
        pend[3 * (size_t)R.i + lane] = mu_new;  // committed in cross() (LDS copy) and by k_tile_commit (record table)
      } else {
        reinterpret_cast<double *>(dst + R.i)[2 * lane + 1] = mu_new;
        if (EP != EP_JACOBI) *reinterpret_cast<double *>(recs + R.self * 48u + lane * 16 + 8) = mu_new;
      }
    }
    if (has_next) tile_settle<1>(N);  // everything but the dipole store just issued: the next row's prefetch has landed
    R = N; sp = sp2; k = k2;             // (before these registers are copied: a copy of a register a load is still to write is stale)
  }
  cross(at, nsub - (DET ? 0 : 1));  // the barriers the other waves still wait at (DET: the last sub-phase's pair too)
  chg += dpp_full<0xB1>(chg);
  chg += dpp_full<0x4E>(chg);
  if (lane == 0 && chg != 0.0) atomicAdd(slot_ptr(slots, SL_CHANGE), chg);
}

// DET: fold the pending dipoles of a launch's rows into the record table (the launch itself only read the table)
static __global__ void k_tile_commit(TileLaunch L, const TileHdr *__restrict__ hdr, const TileRowEnt *__restrict__ trow,
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
    const int i = r0 + (trow[r0 + k].mT & 0xFFFF);
    reinterpret_cast<double *>(s0 + i)[2 * comp + 1] = pend[3 * (size_t)i + comp];
  }
}

