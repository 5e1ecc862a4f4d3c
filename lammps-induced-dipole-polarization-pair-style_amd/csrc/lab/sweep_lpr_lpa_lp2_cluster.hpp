// lab/sweep_lpr_lpa_lp2_cluster.hpp -- LAB BUILD ONLY (-DPOLAR_LAB, libpolar_mi355x_lab.so): code that was built, measured and did not become the
// product path.  Included by polar_solver.hpp inside `#ifdef POLAR_LAB`; the product library never sees it.
// No include guard: it is a fragment of polar_solver.hpp, textually in that file's scope.
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

