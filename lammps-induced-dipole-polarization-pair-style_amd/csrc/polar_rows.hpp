// polar_rows.hpp -- per-row kernels of the step: pack, rank metric (a2), LJ + Ewald-real (a3), static field (a4/a5), polarization forces (a8), fdotr virial (a10).
// Part of the hand-written HIP kernels (gfx950 / CDNA4, wave64) of the lj/cut/coul/long/polarization
// hot path; see polar_kernels.hpp for the mapping and the index spaces.
#pragma once

#include "polar_common.hpp"

namespace polar {

// ------------------------------------------------------------------------------------------
// pack: x/q/alpha (+ initial mu) -> 64-byte records (both Jacobi buffers)
static __global__ void k_pack(int n, const int *__restrict__ perm, const double *__restrict__ x, const double *__restrict__ q,
                       const double *__restrict__ alpha, const int *__restrict__ mol, const double *__restrict__ mu0,
                       AtomRec *__restrict__ r0, AtomRec *__restrict__ r1, int *__restrict__ mol_s,
                       double4 *__restrict__ pos4, double4 *__restrict__ xq_s, int wrap, Box box, double lo0, double lo1,
                       double lo2) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i == n) {  // the DUMMY record the lp sweep pads its rows with: zero dipole (contributes nothing), zero
    // polarizability, a finite position (that of atom 0: the sweep floors r^2, so a coincidence is harmless)
    AtomRec d;
    d.x = n > 0 ? x[0] : 0.0; d.y = n > 0 ? x[1] : 0.0; d.z = n > 0 ? x[2] : 0.0;
    d.mx = d.my = d.mz = d.q = d.a = 0.0;
    r0[n] = d; r1[n] = d;
  }
  if (i >= n) return;
  const int o = perm ? perm[i] : i;
  AtomRec r;
  r.x = x[3 * o]; r.y = x[3 * o + 1]; r.z = x[3 * o + 2]; r.q = q[o];
  if (wrap) {
    // tile sweep: the records hold the image INSIDE the box (LAMMPS lets atoms drift out of it between reneighborings);
    // the builder then finds every partner's image from cell offsets alone.  Whole lattice vectors c, b, a come off.
    const double lo[3] = {lo0, lo1, lo2};
    double fr[3];
    frac_coords(box, lo, r.x, r.y, r.z, fr);
    const double n2 = box.periodic[2] ? floor(fr[2]) : 0.0, n1 = box.periodic[1] ? floor(fr[1]) : 0.0,
                 n0 = box.periodic[0] ? floor(fr[0]) : 0.0;
    r.z -= n2 * box.prd[2]; r.y -= n2 * box.yz + n1 * box.prd[1]; r.x -= n2 * box.xz + n1 * box.xy + n0 * box.prd[0];
  }
  r.mx = mu0 ? mu0[3 * o] : 0.0; r.my = mu0 ? mu0[3 * o + 1] : 0.0; r.mz = mu0 ? mu0[3 * o + 2] : 0.0;
  r.a = alpha[o];
  r0[i] = r;
  r1[i] = r;
  mol_s[i] = mol[o];
  // 32-byte {x, y, z, (molecule id, alpha != 0)} for the list build
  if (pos4) pos4[i] = make_double4(r.x, r.y, r.z, __hiloint2double(mol[o], r.a != 0.0 ? 1 : 0));
  if (xq_s) xq_s[i] = make_double4(r.x, r.y, r.z, r.q);  // 32-byte {x, y, z, q} for the static-field rows
}

// ------------------------------------------------------------------------------------------
// a2  rank metric, PS.cpp:192-227.  Pass 1: rmin; pass 2: rank_metric.
// ALLPAIRS: raw (non-minimum-image) distances to locals AND ghosts, exactly as the reference.
// list mode (extension): minimum-image distances over the library's full list.
template <bool ALLPAIRS, int PASS>
static __global__ __launch_bounds__(POLAR_BLOCK) void k_rank(int nlocal, int ntotal, const double *__restrict__ x,
                                                      const double *__restrict__ alpha, const int *__restrict__ mol,
                                                      Box box, RowList nl,
                                                      const int *__restrict__ nl_j,
                                                      const AtomRec *__restrict__ rec,
                                                      const int *__restrict__ mol_s, Scal *scal,
                                                      double *__restrict__ slots,
                                                      double *__restrict__ rank_metric, int nchunk) {
  // ALLPAIRS: orig space (x/alpha/mol incl. ghosts).  List mode: s space (records, mol_s).
  // nchunk > 1 (PASS 1 of the all-pairs form only: a minimum does not care about order): the partners of an atom are split
  // over nchunk waves -- one wave per atom leaves an exact-mode system of a thousand atoms with one wave per SIMD
  const int lane = threadIdx.x & 63;
  const int gw = blockIdx.x * POLAR_ROWS_PER_BLOCK + (threadIdx.x >> 6);
  if (gw >= nlocal * nchunk) return;
  const int i = gw % nlocal, chunk = gw / nlocal;
  const double xi = ALLPAIRS ? x[3 * i] : rec[i].x, yi = ALLPAIRS ? x[3 * i + 1] : rec[i].y,
               zi = ALLPAIRS ? x[3 * i + 2] : rec[i].z, ai = ALLPAIRS ? alpha[i] : rec[i].a;
  const int mi = ALLPAIRS ? mol[i] : mol_s[i];
  double rmin = (PASS == 1) ? 1000.0 : __longlong_as_double((long long)scal->rmin_bits);
  double rmin2 = 1.0e6;                                    // PASS 1: smallest r^2 seen (1000^2: the reference's start value)
  const double far2 = (rmin * 1.5) * (rmin * 1.5) * 1.000001;   // PASS 2: r^2 beyond this cannot satisfy rmin * 1.5 > r
  double acc = 0.0;
  long long beg = 0, end = ntotal;
  if (!ALLPAIRS) row_range(nl, i, beg, end);
  else if (nchunk > 1) { const long long per = ((ntotal + nchunk - 1) / nchunk + 255) / 256 * 256; beg = chunk * per; end = beg + per < ntotal ? beg + per : ntotal; }
  // one partner per lane and trip; the all-pairs form requests four trips' worth of partners at once (one wave walks ~10^4
  // partners: a memory latency per trip was 97 + 69 us per step at 1349 atoms) and then takes them in the same order
  constexpr int U = ALLPAIRS ? 4 : 1;
  for (long long base = beg; base < end; base += 64 * U) {
    double px[U], py[U], pz[U], pa[U];
    int pj[U], pm[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const long long p = base + 64 * u + lane;
      pj[u] = i;   // (a lane past the end looks at the atom itself: skipped below)
      if (p < end) pj[u] = ALLPAIRS ? (int)p : (nl_j[p] & POLAR_NL_MASK);
      if (ALLPAIRS) {
        px[u] = x[3 * pj[u]]; py[u] = x[3 * pj[u] + 1]; pz[u] = x[3 * pj[u] + 2];
        pa[u] = alpha[pj[u]]; pm[u] = mol[pj[u]];
      } else {
        const AtomRec rj = rec[pj[u]];
        px[u] = rj.x; py[u] = rj.y; pz[u] = rj.z; pa[u] = rj.a; pm[u] = mol_s[pj[u]];
      }
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      bool hit = false;
      double term = 0.0;
      if (pj[u] != i) {
        double dx, dy, dz;
        if (ALLPAIRS) { dx = xi - px[u]; dy = yi - py[u]; dz = zi - pz[u]; }
        else min_image_rint(box, xi, yi, zi, px[u], py[u], pz[u], dx, dy, dz);
        const double rsq = dx * dx + dy * dy + dz * dz;
        const bool molok = (mi != pm[u]) || mi == 0;
        // (the square root only where it decides: it is monotone and correctly rounded, so the smallest r is the root of the
        //  smallest r^2, and a pair well outside 1.5 rmin cannot pass the reference's test `rmin * 1.5 > r`)
        if (PASS == 1) {
          if (ai > 0 && pa[u] > 0 && molok) rmin2 = fmin(rmin2, rsq);
        } else if (rsq < far2 && molok && rmin * 1.5 > sqrt(rsq)) {
          hit = true;
          term = ai * pa[u];
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
  }
  if (PASS == 1) {
    rmin = wave_min(fmin(rmin, sqrt(rmin2)));
    if (lane == 0)
      atomicMin((unsigned long long *)slot_ptr(slots, SL_RMIN), (unsigned long long)__double_as_longlong(rmin));
  } else {
    if (lane == 0) rank_metric[i] = acc;  // identical in every lane
  }
}

// The ranked sweep order on the device: position of atom i = how many atoms come before it in a STABLE descending sort by
// rank_metric (== the reference's bubble sort, PS.cpp:1130-1143): those with a larger metric, and those with the same metric
// and a smaller index.  One wave per atom, n^2 comparisons (exact mode: a few thousand atoms); order[pos] = i, pos[i].
static __global__ __launch_bounds__(POLAR_BLOCK) void k_rank_order(int n, const double *__restrict__ rk, int *__restrict__ order, int *__restrict__ pos) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * POLAR_ROWS_PER_BLOCK + (threadIdx.x >> 6);
  if (i >= n) return;
  const double ri = rk[i];
  int before = 0;
  for (int j0 = 0; j0 < n; j0 += 64) {
    const int j = j0 + lane;
    const double rj = j < n ? rk[j] : 0.0;
    before += __popcll(__ballot(j < n && (rj > ri || (rj == ri && j < i))));
  }
  if (lane == 0) { pos[i] = before; order[before] = i; }
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
  int typed_list; // list entries carry the partner's type in bits 24-29 (index in bits 0-23)
  int tab_lowmask;     // (1 << ncoulshiftbits) - 1: the mantissa bits of (float)rsq below a bin's resolution
  int i_special[2];    // bins whose drtable is not the regular 2^k (the last bin before the cutoff, the wrap of the index), -1: none
  double dr_special[2];
  double tabinnersq, cut_coulsq, g_ewald, qqrd2e;
  double special_lj[4], special_coul[4];
  const double *ljpack;   // [(ntypes+1)^2][8] = cutsq, cut_ljsq, lj1, lj2, lj3, lj4, offset, pad
  const double *ctab;     // [ntable][8]      = r, dr, f, df, e, de, c, dc  (one 64-byte line per bin)
};

// per-atom pack for the half-list loop: 32-byte {x,y,z,q} + type, locals AND ghosts, orig order
static __global__ void k_pack_lj(int nall, const double *__restrict__ x, const double *__restrict__ q, double4 *__restrict__ xq) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nall) xq[i] = make_double4(x[3 * i], x[3 * i + 1], x[3 * i + 2], q[i]);
}

// Symmetrised copy of LAMMPS' half list, built on the device when the list is uploaded: every pair
// (i,j) of the half list appears in the row of i AND in the row of j (ghost atoms get rows too), so
// the force loop needs no atomics on j -- the three scattered FP64 atomics per pair were 85 % of
// the kernel.  Each row then accumulates the full force on its atom; pair tallies count 1/2 per row.
static __global__ __launch_bounds__(POLAR_BLOCK) void k_sym_count(int inum, const int *__restrict__ ilist,
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
static __global__ __launch_bounds__(POLAR_BLOCK) void k_sym_fill(int inum, const int *__restrict__ ilist,
                                                          const int *__restrict__ numneigh,
                                                          const long long *__restrict__ first,
                                                          const int *__restrict__ neigh,
                                                          const long long *__restrict__ sfirst, int *__restrict__ fill,
                                                          int *__restrict__ sj, const int *__restrict__ type) {
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
    // `type` != NULL (fewer than 2^24 atoms, fewer than 64 types): the partner's type rides in bits 24-29 of the entry, which
    // saves the force loop one scattered 4-byte gather per pair (the loop is bound by its vector-memory instructions)
    sj[base + jj] = type ? (e | (type[j] << 24)) : e;                                          // j with its special bits
    sj[sfirst[j] + atomicAdd(&fill[j], 1)] = i | (e & 0xC0000000) | (type ? type[i] << 24 : 0);  // reverse entry, same bits
  }
}

// One wave per listed atom i.  Per pair: 1 coalesced index load, two 16-byte gathers of {x,y,z,q},
// one 4-byte gather of the type, the type-pair parameters out of LDS, and the Coulomb bin as one
// 64-byte line -- the loop is bound by L1 transactions and by the three FP64 atomics that deposit
// -F on j (LAMMPS' newton-on contract: ghosts are folded back by reverse_comm).
// The work of ONE row (one wave).  TAB: where a pair finds its Coulomb bin (PS.cpp:268-285 reads rtable / drtable / ftable /
// dftable / etable / detable / ctable / dctable):
//   0  every table from memory, one 64-byte line per bin (three scattered 16-byte loads per pair);
//   1  {f, df, e, de} of every bin from LDS (`ctab_lds`, 32 B per bin: 128 KB at the 12 table bits LAMMPS defaults to), {r, dr}
//      from memory;
//   2  {f, df, e, de} from LDS and r, dr REBUILT from the bits of (float)rsq: r = the float with its low `shift` mantissa bits
//      cleared, dr = 2^(23 - shift - exponent) -- what Pair::init_tables stores for every regular bin (checked bin by bin on the
//      host when the tables arrive, upload_coul; the at most two irregular bins -- the last one before the cutoff, the wrap of
//      the index -- carry their dr in P.dr_special).  No table load is left per pair, only the (rare) special-bond correction.
// The vector-memory unit was 80 % busy in the TAB 0 kernel, on 64 distinct lines per gather instruction and five such
// instructions per trip: that, not the arithmetic and not the locality of the {x,y,z,q} gathers (profiles/r05_lab_ljcoul.txt),
// bounded it.  Sums of a wave over the rows it walks stay in registers (esum / vsum) until the caller adds them to a slot.
template <bool EFLAG, bool VPAIR, int TAB>
__device__ __forceinline__ void ljcoul_row(const LJCoulParams &P, int i, int jnum, const int *__restrict__ jlist, const double4 pi,
                                           const double4 *__restrict__ xq, const int *__restrict__ type, const double *lj_lds,
                                           const double4 *ctab_lds, int lane, double *__restrict__ f, double *__restrict__ eatom,
                                           double *__restrict__ vatom, double &ev_sum, double &ec_sum, double (&vsum)[6]) {
  const double EWALD_F = 1.12837917, EWALD_P = 0.3275911, A1 = 0.254829592, A2 = -0.284496736, A3 = 1.421413741,
               A4 = -1.453152027, A5 = 1.061405429;  // PS.cpp:43-49
  const int w = P.ntypes + 1;
  // row data is the same in every lane: scalar registers (see wave_uniform)
  const double qtmp = wave_uniform(pi.w), xtmp = wave_uniform(pi.x), ytmp = wave_uniform(pi.y), ztmp = wave_uniform(pi.z);
  const int itype = __builtin_amdgcn_readfirstlane(type[i]);
  double fx = 0, fy = 0, fz = 0, ev = 0, ec = 0;
  double v0 = 0, v1 = 0, v2 = 0, v3 = 0, v4 = 0, v5 = 0;
  // software prefetch, two trips deep: the index of trip t+2 and the {x,y,z,q} of trip t+1 travel while
  // trip t is computed (per trip the chain index -> position -> Coulomb bin is three dependent loads)
  int e_next = jlist[lane < jnum ? lane : 0];
  const int jmask = P.typed_list ? 0x00FFFFFF : 0x3FFFFFFF;
  double4 p_next = xq[e_next & jmask];
  int e_next2 = jlist[lane + 64 < jnum ? lane + 64 : 0];
  for (int jj = lane; jj < jnum; jj += 64) {
    int j = e_next;
    const double4 pj = p_next;
    e_next = e_next2;
    p_next = xq[e_next & jmask];
    e_next2 = jlist[jj + 128 < jnum ? jj + 128 : 0];
    const int sb = (j >> 30) & 3;  // sbmask, src/pair.h:241
    const double factor_lj = P.special_lj[sb], factor_coul = P.special_coul[sb];
    const int jt_bits = (j >> 24) & 63;  // typed lists (P.typed_list): the partner's type
    j &= P.typed_list ? 0x00FFFFFF : 0x3FFFFFFF;  // NEIGHMASK
    const double delx = xtmp - pj.x, dely = ytmp - pj.y, delz = ztmp - pj.z;
    const double rsq = delx * delx + dely * dely + delz * delz;
    const double *lj = lj_lds + (itype * w + (P.typed_list ? jt_bits : type[j])) * 8;
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
          const int rbits = __float_as_int(rsqf);
          const int itable = (rbits & P.ncoulmask) >> P.ncoulshiftbits;
          const double2 *bin = reinterpret_cast<const double2 *>(P.ctab + (size_t)itable * 8);
          double2 rdr, fdf;
          if (TAB == 2) {
            rdr.x = (double)__int_as_float(rbits & ~P.tab_lowmask);
            rdr.y = __hiloint2double((1173 - P.ncoulshiftbits - ((rbits >> 23) & 0xFF)) << 20, 0);   // 2^(23 - shift - (e - 127))
            if (itable == P.i_special[0]) rdr.y = P.dr_special[0];
            if (itable == P.i_special[1]) rdr.y = P.dr_special[1];
          } else rdr = bin[0];
          if (TAB == 0) {
            fdf = bin[1];
            if (EFLAG) tab_e = bin[2];
          } else {
            const double4 t4 = ctab_lds[itable];
            fdf = make_double2(t4.x, t4.y);
            if (EFLAG) tab_e = make_double2(t4.z, t4.w);
          }
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
    ev_sum += ev; ec_sum += ec;
    // per-atom energy, src/pair.cpp:881-885: every pair of a full row carries weight 1/2, so the
    // row total IS eatom[i] (one wave per row: plain store-add, no atomics)
    if (lane == 0 && eatom) eatom[i] += ev + ec;
  }
  if (VPAIR) {
    v0 = wave_sum(v0); v1 = wave_sum(v1); v2 = wave_sum(v2); v3 = wave_sum(v3); v4 = wave_sum(v4); v5 = wave_sum(v5);
    vsum[0] += v0; vsum[1] += v1; vsum[2] += v2; vsum[3] += v3; vsum[4] += v4; vsum[5] += v5;
    if (lane == 0 && vatom) {  // src/pair.cpp:925-942
      double *va = vatom + 6 * (size_t)i;
      va[0] += v0; va[1] += v1; va[2] += v2; va[3] += v3; va[4] += v4; va[5] += v5;
    }
  }
}
// a wave's sums over the rows it walked -> one accumulator slot (`slot`: any number, spread over the NSLOT lines)
template <bool EFLAG, bool VPAIR>
__device__ __forceinline__ void ljcoul_tally(double *__restrict__ slots, int slot, int lane, int vglobal, double ev, double ec, const double (&v)[6]) {
  if (lane != 0) return;
  double *s = slots + (size_t)(slot & (POLAR_NSLOT - 1)) * POLAR_SLOT_STRIDE;
  if (EFLAG) { atomicAdd(s + SL_EVDWL, ev); atomicAdd(s + SL_ECOUL, ec); }
  if (VPAIR && vglobal) {
    atomicAdd(s + SL_V0, v[0]); atomicAdd(s + SL_V1, v[1]); atomicAdd(s + SL_V2, v[2]);
    atomicAdd(s + SL_V3, v[3]); atomicAdd(s + SL_V4, v[4]); atomicAdd(s + SL_V5, v[5]);
  }
}

// one wave per listed row, Coulomb bins from memory (TAB 0): tables beyond the LDS, `pair_modify table 0`, small systems
template <bool EFLAG, bool VPAIR>
static __global__ __launch_bounds__(POLAR_BLOCK) void k_ljcoul(LJCoulParams P, int inum, const int *__restrict__ ilist,
                                                        const int *__restrict__ numneigh,
                                                        const long long *__restrict__ first,
                                                        const int *__restrict__ neigh,
                                                        const double4 *__restrict__ xq, const int *__restrict__ type,
                                                        double *__restrict__ f, double *__restrict__ slots,
                                                        double *__restrict__ eatom, double *__restrict__ vatom,
                                                        int vglobal) {
  extern __shared__ double lj_lds[];
  const int w = P.ntypes + 1;
  for (int t = threadIdx.x; t < w * w * 8; t += blockDim.x) lj_lds[t] = P.ljpack[t];
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int ii = blockIdx.x * POLAR_ROWS_PER_BLOCK + (threadIdx.x >> 6);
  if (ii >= inum) return;
  const int i = ilist ? ilist[ii] : ii;
  const int jnum = numneigh ? numneigh[i] : (int)(first[i + 1] - first[i]);
  if (jnum == 0) return;
  double ev = 0.0, ec = 0.0, v[6] = {0, 0, 0, 0, 0, 0};
  ljcoul_row<EFLAG, VPAIR, 0>(P, i, jnum, neigh + first[i], xq[i], xq, type, lj_lds, nullptr, lane, f, eatom, vatom, ev, ec, v);
  ljcoul_tally<EFLAG, VPAIR>(slots, blockIdx.x, lane, vglobal, ev, ec, v);
}

// PERSISTENT form with the Coulomb bins in LDS (TAB 1 / 2): one 1024-thread workgroup per CU loads {f, df, e, de} of every bin
// (32 B x 2^ncoultablebits = 128 KB at 12 bits) behind the LJ table ONCE, then its 16 waves walk rows ii = first, first + stride, ...
// A workgroup holds most of its CU's 160 KB of LDS while it runs: a3 shares the CU with kernels that need little LDS (cell
// sort, list build, static field) and finishes before the sweeps start.
#define POLAR_LJ_PERS_THREADS 1024
template <bool EFLAG, bool VPAIR, int TAB>
static __global__ __launch_bounds__(POLAR_LJ_PERS_THREADS) void k_ljcoul_pers(LJCoulParams P, int inum, const int *__restrict__ ilist,
                                                        const int *__restrict__ numneigh,
                                                        const long long *__restrict__ first,
                                                        const int *__restrict__ neigh,
                                                        const double4 *__restrict__ xq, const int *__restrict__ type,
                                                        double *__restrict__ f, double *__restrict__ slots,
                                                        double *__restrict__ eatom, double *__restrict__ vatom,
                                                        int vglobal) {
  extern __shared__ double lj_lds[];
  const int w = P.ntypes + 1, nlj = (w * w * 8 + 3) & ~3;    // (the bins start 32-byte aligned)
  double4 *ctab_lds = reinterpret_cast<double4 *>(lj_lds + nlj);
  for (int t = threadIdx.x; t < w * w * 8; t += blockDim.x) lj_lds[t] = P.ljpack[t];
  const int ntable = 1 << P.ncoultablebits;
  for (int b = threadIdx.x; b < ntable; b += blockDim.x) {
    const double2 *bin = reinterpret_cast<const double2 *>(P.ctab + (size_t)b * 8);
    const double2 fdf = bin[1], ede = bin[2];
    ctab_lds[b] = make_double4(fdf.x, fdf.y, ede.x, ede.y);
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nwv = blockDim.x >> 6;
  double ev = 0.0, ec = 0.0;
  // interleaved over the workgroups: rows ii, ii + 1, ... of neighbouring atoms run at the same time on different CUs, and every
  // workgroup sees the same mix of row lengths
  for (int ii = blockIdx.x + gridDim.x * wv; ii < inum; ii += gridDim.x * nwv) {
    const int i = ilist ? ilist[ii] : ii;
    const int jnum = numneigh ? numneigh[i] : (int)(first[i + 1] - first[i]);
    if (jnum == 0) continue;
    double v[6] = {0, 0, 0, 0, 0, 0};
    ljcoul_row<EFLAG, VPAIR, TAB>(P, i, jnum, neigh + first[i], xq[i], xq, type, lj_lds, ctab_lds, lane, f, eatom, vatom, ev, ec, v);
    // (the pairwise virial of a row goes out with the row: six running sums more would not fit the 128 registers of a
    //  1024-thread workgroup; the energies of all rows of the wave go out once, below)
    if (VPAIR) ljcoul_tally<false, true>(slots, ii, lane, vglobal, 0.0, 0.0, v);
  }
  const double none[6] = {0, 0, 0, 0, 0, 0};
  ljcoul_tally<EFLAG, false>(slots, blockIdx.x * nwv + wv, lane, 0, ev, ec, none);
}

// ------------------------------------------------------------------------------------------
// a4 + a5  static field (shifted-force Coulomb, PS.cpp:324-361), unit scale and initial guess
// (PS.cpp:363-386).  Full-row evaluation: E_i = sum_j ef_temp * q_j * del_ij, which is the
// reference's i<j scatter seen from row i (del is antisymmetric under the image rule).
template <bool ALLPAIRS>
static __global__ __launch_bounds__(POLAR_BLOCK) void k_static_field(const int *__restrict__ rows, int nrows, int nlocal,
                                                              const AtomRec *__restrict__ rec,
                                                              const int *__restrict__ mol, Box box,
                                                              RowList nl,
                                                              const int *__restrict__ nl_j, double cut_coulsq,
                                                              double e2s, double gamma, int use_previous,
                                                              double *__restrict__ ef, AtomRec *__restrict__ rec0,
                                                              AtomRec *__restrict__ rec1, const double4 *__restrict__ xq_s) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * POLAR_ROWS_PER_BLOCK + (threadIdx.x >> 6);
  if (row >= nrows) return;
  const int i = rows ? rows[row] : row;
  const AtomRec ri = uniform_rec(rec[i]);  // row data is the same in every lane: scalar registers
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
      double xj, yj, zj, qj;
      if (xq_s) { const XQ t = fetch_xq(xq_s, j, stage, lane); xj = t.a.x; yj = t.a.y; zj = t.b.x; qj = t.b.y; }
      else { const RecQuad rj = fetch_records(rec, j, stage, lane); xj = rj.a.x; yj = rj.b.x; zj = rj.c.x; qj = rj.d.x; }
      double dx, dy, dz;
      pair_del<ALLPAIRS>(box, ri.x, ri.y, ri.z, xj, yj, zj, dx, dy, dz);
      const double rsq = dx * dx + dy * dy + dz * dz;
      if (valid && j != i && rsq <= cut_coulsq && !(e & POLAR_NL_SAMEMOL)) {  // note <=, PS.cpp:342
        const double rinv = rsqrt(rsq);
        const double ef_temp = (rinv * rinv + f_shift) * rinv * qj;
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
// a8  polarization forces and energies, PS.cpp:406-641, evaluated per row (force on i from every j).
// The pair force is antisymmetric, so summing rows reproduces the reference's i<j scatter;
// pair energies are counted from both rows and halved.
template <bool ALLPAIRS, int DAMP, bool EFLAG, bool VPAIR>
static __global__ __launch_bounds__(POLAR_BLOCK) void k_polar_force(const int *__restrict__ rows, int nrows, const int *__restrict__ perm,
                                                             int nlocal, const Scal *scal_in,
                                                             const AtomRec *__restrict__ recA,
                                                             const AtomRec *__restrict__ recB,
                                                             const int *__restrict__ mol, Box box,
                                                             RowList nl,
                                                             const int *__restrict__ nl_j, double cut_coulsq,
                                                             double ddcutsq, double pd, double e2s,
                                                             double *__restrict__ f, double *__restrict__ slots,
                                                             double *__restrict__ vatom, int vglobal, ExpCoef K,
                                                             double *__restrict__ dbg6) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * POLAR_ROWS_PER_BLOCK + (threadIdx.x >> 6);
  if (row >= nrows) return;
  const int i = rows ? rows[row] : row;
  const AtomRec *__restrict__ rec = __builtin_amdgcn_readfirstlane(scal_in->cur) ? recB : recA;  // scalar base
  const AtomRec ri = uniform_rec(rec[i]);  // row data is the same in every lane: scalar registers
  const int mi = mol[i];
  const double f_shift = -1.0 / cut_coulsq;
  double fx = 0, fy = 0, fz = 0, uef = 0, udd = 0;
  double v0 = 0, v1 = 0, v2 = 0, v3 = 0, v4 = 0, v5 = 0;
  // `debug yes` (PS.cpp:542-556, 612-626, 637-638): the row of the caller's atom 0 also keeps the dipole-dipole part of its force
  const bool isdbg = dbg6 && (perm ? perm[i] : i) == 0;   // wave-uniform
  double ddx = 0, ddy = 0, ddz = 0;
  long long beg = 0, end = nlocal;
  if (!ALLPAIRS) row_range(nl, i, beg, end);
  // (the cooperative record fetch of k_static_field was tried here too: this kernel is bound by its FP64
  //  arithmetic, not by the gathers, and got 7 % slower)
  for (long long p = beg + lane; p < end; p += 64) {
    const int e = ALLPAIRS ? (int)p : nl_j[p];
    const int j = ALLPAIRS ? e : (e & POLAR_NL_MASK);
    if (j == i) continue;
    const AtomRec rj = *reinterpret_cast<const AtomRec *>(reinterpret_cast<const char *>(rec) + ((unsigned)j << 6));
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
        const double t1 = exp_neg(-pd * r, K);
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
      const double qx = pre_r * dx + pre2 * ri.mx + pre3 * rj.mx, qy = pre_r * dy + pre2 * ri.my + pre3 * rj.my,
                   qz = pre_r * dz + pre2 * ri.mz + pre3 * rj.mz;
      px += qx; py += qy; pz += qz;
      if (isdbg) { ddx += qx; ddy += qy; ddz += qz; }
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
  if (isdbg) {
    ddx = wave_sum(ddx); ddy = wave_sum(ddy); ddz = wave_sum(ddz);
    if (lane == 0) { dbg6[0] = fx; dbg6[1] = fy; dbg6[2] = fz; dbg6[3] = ddx; dbg6[4] = ddy; dbg6[5] = ddz; }
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

static __global__ void k_add_into(long long n, const double *__restrict__ src, double *__restrict__ dst) {
  long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (i < n) dst[i] += src[i];
}

// a10  virial_fdotr_compute, src/pair.cpp:1495-1540: sum over locals AND ghosts of f_i x_i
static __global__ __launch_bounds__(POLAR_BLOCK) void k_virial_fdotr(int nall, const double *__restrict__ x,
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

}  // namespace polar
