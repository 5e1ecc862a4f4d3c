// lab/sweep_quad.hpp -- LAB BUILD ONLY (-DPOLAR_LAB, libpolar_mi355x_lab.so): code that was built, measured and did not become the
// product path.  Included by polar_solver.hpp inside `#ifdef POLAR_LAB`; the product library never sees it.
// No include guard: it is a fragment of polar_solver.hpp, textually in that file's scope.
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

