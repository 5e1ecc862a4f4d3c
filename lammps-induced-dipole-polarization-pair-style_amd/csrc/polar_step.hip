// polar_step.hip -- one PairLJCutCoulLongPolarization::compute (PS.cpp:125-645) on one MI355X: cell order and the pitched
// lists, a3 on its side stream, static field, the solve (DipoleSolverIterative, PS.cpp:1113-1238) with its sweep launchers,
// forces and tallies.  All arithmetic of the hot path runs in the kernels of polar_kernels.hpp.
#include "polar_handle.hpp"

// exclusive scan of n ints (out: n + 1 long longs), polar_lists.hpp
void launch_scan(long long n, const int *in, long long *out, DBuf<long long> &tot, hipStream_t s) {
  const int nb = (int)std::max<long long>(1, (n + POLAR_SCAN_ITEMS - 1) / POLAR_SCAN_ITEMS);
  tot.ensure((size_t)nb + 1);
  k_scan_block<<<nb, 1024, 0, s>>>(n, in, out, tot.p);
  if (nb > 1) {
    k_scan_totals<<<1, 1024, 0, s>>>(nb, tot.p, out + n);
    k_scan_add<<<nblk(n - POLAR_SCAN_ITEMS, 256), 256, 0, s>>>(n, out, tot.p);
  }
}

void build_cells(polar_handle *h) {
  const polar_settings &st = h->ph.st;
  const int n = h->nlocal;
  const double cutall = std::max(st.cut_coul, st.dd_cutoff);
  double width[3];
  box_widths(h->box, width);
  for (int k = 0; k < 3; k++)
    if (h->box.periodic[k] && width[k] < 2.0 * cutall * (1.0 - 1e-12))
      throw InputError("dd_cutoff mode needs box lengths >= 2*max(cut_coul,dd_cutoff); use exact mode (dd_cutoff 0)");
  CellGrid &g = h->grid;
  g.trim = 1;
#ifdef POLAR_LAB
  if (const char *e = getenv("POLAR_NL_TRIM")) g.trim = atoi(e);
#endif
  if (h->box.triclinic) g.trim = 0;  // the per-atom stencil trimming measures orthogonal distances
  long long ncell = 1;
  for (int k = 0; k < 3; k++) {
    g.nc[k] = std::max(1, (int)std::floor(width[k] / (0.5 * cutall)));  // cell height >= cutoff/2: +-2 stencil
    // tile sweep: the launches are the parity classes of the cells; an even count in a periodic dimension needs no third
    // class for the seam (cells grow by at most 1/6)
    if (h->sweep_kernel == 4 && k == 0 && h->tile_wide) g.nc[k] = std::max(1, (int)std::floor(width[k] / cutall));
    if (h->sweep_kernel == 4 && h->box.periodic[k] && g.nc[k] >= 7 && (g.nc[k] & 1)) g.nc[k] -= 1;
    h->tile_sw[k] = std::max(1, std::min(2, (int)std::ceil(st.dd_cutoff / (width[k] / g.nc[k]) - 1e-9)));
    g.lo[k] = h->boxlo[k];
    g.inv[k] = g.nc[k] / h->box.prd[k];
    ncell *= g.nc[k];
  }
  h->ncell = ncell;
  if (h->sweep_kernel == 4) {  // tile colours: per dimension the even cells, the odd cells and -- odd count, periodic -- the last cell
    struct Cls { int start, stride, count; };
    std::vector<Cls> cls[3];
    for (int k = 0; k < 3; k++) {
      const int nc = g.nc[k];
      const bool seam = h->box.periodic[k] && (nc & 1) && nc > 1;
      const int lim = seam ? nc - 1 : nc;
      if ((lim + 1) / 2 > 0) cls[k].push_back(Cls{0, 2, (lim + 1) / 2});
      if (lim / 2 > 0) cls[k].push_back(Cls{1, 2, lim / 2});
      if (seam) cls[k].push_back(Cls{nc - 1, 1, 1});
    }
    h->tile_launches.clear();
    for (const Cls &cz : cls[2]) for (const Cls &cy : cls[1]) for (const Cls &cx : cls[0]) {
      TileLaunch L;
      L.start[0] = cx.start; L.stride[0] = cx.stride; L.count[0] = cx.count;
      L.start[1] = cy.start; L.stride[1] = cy.stride; L.count[1] = cy.count;
      L.start[2] = cz.start; L.stride[2] = cz.stride; L.count[2] = cz.count;
      for (int k = 0; k < 3; k++) L.nc[k] = g.nc[k];
      h->tile_launches.push_back(L);
    }
    for (int k = 0; k < 3; k++) { h->tile_all.start[k] = 0; h->tile_all.stride[k] = 1; h->tile_all.count[k] = g.nc[k]; h->tile_all.nc[k] = g.nc[k]; }
  }
  hipStream_t s = h->stream;
  h->d_cell_id.ensure(n); h->d_cell_cnt.ensure(ncell + 1); h->d_cell_fill.ensure(2 * (ncell + 1));
  h->d_cell_first.ensure(ncell + 2); h->d_perm.ensure(n + 1); h->d_inv.ensure(n + 1);
  zero_many(s, {{h->d_cell_cnt.p, (size_t)(ncell + 1) * sizeof(int)}, {h->d_cell_fill.p, 2 * (size_t)(ncell + 1) * sizeof(int)}});
  k_cell_count<<<nblk(n, 256), 256, 0, s>>>(n, h->d_x.p, g, h->box, h->d_cell_id.p, h->d_cell_cnt.p);
  launch_scan(ncell, h->d_cell_cnt.p, h->d_cell_first.p, h->d_scan_a, s);
  k_cell_fill<<<nblk(n, 256), 256, 0, s>>>(n, h->d_cell_id.p, h->d_cell_first.p, h->d_cell_fill.p,
                                           (h->pol_first || h->sweep_kernel == 4) ? h->d_cell_fill.p + ncell + 1 : nullptr, h->d_alpha.p, h->d_perm.p, h->d_inv.p);
  // the order inside a cell follows the atomics of k_cell_fill: put it into atom order -- always, not only for `deterministic
  // yes` (reproducible sums): the device colouring breaks its ties by position in the cell, and a colouring that changed from
  // run to run would make unconverged (`fixed_iteration`) results differ at 1e-6 instead of the 1e-9 of the in-place race
  k_cell_sort<<<nblk(ncell, 4), 256, 0, s>>>(ncell, h->d_cell_first.p, (h->pol_first || h->sweep_kernel == 4) ? h->d_cell_fill.p : nullptr, h->d_perm.p, h->d_inv.p);
  h->sorted = true;
  if (sharded(h)) {
    h->d_ownrows.ensure(own_n(h) + 1);
    k_map_range<<<nblk(own_n(h), 256), 256, 0, s>>>(own_lo(h), own_n(h), h->d_inv.p, h->d_ownrows.p);
  }
}

// lp sweep: where k_nl_build stores every atom's dd row = its row in launch order (colour phases back to back for GS,
// own rows for Jacobi).  While no colouring exists yet (first step, or after a clash) the rows are laid out in atom
// order and the lists are built once more after the colouring (solve / polar_step_begin).
void compute_slots(polar_handle *h) {
  const polar_settings &st = h->ph.st;
  const bool gs = st.polar_gs || st.polar_gs_ranked;
  const int n = h->nlocal;
  hipStream_t s = h->stream;
  h->d_slot.ensure(n + 1);
  HIPCHECK(hipMemsetAsync(h->d_slot.p, 0xFF, (size_t)(n + 1) * sizeof(int), s));
  if (gs && h->colors_valid) {
    const int tot = h->color_off.empty() ? 0 : h->color_off.back();
    if (tot > 0) k_slot_from_rows<<<nblk(tot, 256), 256, 0, s>>>(tot, h->d_rows.p, h->d_slot.p);
    h->slots_by_color = true;
    h->slots_epoch = h->color_epoch;
  } else {
    k_slot_from_rows<<<nblk(own_n(h), 256), 256, 0, s>>>(own_n(h), own_rows(h), h->d_slot.p);
    h->slots_by_color = false;
  }
}

void build_lists(polar_handle *h) {
  const polar_settings &st = h->ph.st;
  const int n = h->nlocal;
  const double cutall = std::max(st.cut_coul, st.dd_cutoff);
  hipStream_t s = h->stream;
  const CellGrid &g = h->grid;
  if (h->nl_pitch == 0 && getenv("POLAR_INIT_PITCH")) h->nl_pitch = h->dd_pitch = ((std::max(64, atoi(getenv("POLAR_INIT_PITCH"))) + 63) / 64) * 64;  // tests: force the overflow path
  if (h->nl_pitch == 0) {  // first build: 1.5x the mean sphere population, rounded to 64
    // density over the OCCUPIED part of the box (a shard handle holds one slab plus its halo, not the
    // whole box): count the non-empty cells of a coarse host grid with edge ~cutoff
    double vol = h->box.prd[0] * h->box.prd[1] * h->box.prd[2];
    {
      int nc[3];
      long long tot = 1;
      for (int k = 0; k < 3; k++) { nc[k] = std::max(1, std::min(64, (int)(h->box.prd[k] / cutall))); tot *= nc[k]; }
      std::vector<char> occ((size_t)tot, 0);
      for (int a = 0; a < n; a++) {
        long long c = 0, mul = 1;
        for (int k = 0; k < 3; k++) {
          double fr = (h->hx[3 * (size_t)a + k] - h->grid.lo[k]) / h->box.prd[k];
          fr -= std::floor(fr);
          c += mul * std::min(nc[k] - 1, (int)(fr * nc[k]));
          mul *= nc[k];
        }
        occ[(size_t)c] = 1;
      }
      long long filled = 0;
      for (char v : occ) filled += v;
      if (filled > 0) vol *= (double)filled / (double)tot;
    }
    h->dens = n / vol;
    double mean = n / vol * 4.18879020478639 * cutall * cutall * cutall;
    h->nl_pitch = h->dd_pitch = (((long long)(1.5 * mean) + 64) / 64 + 1) * 64;
  }
  if (h->sweep_kernel == 2) h->dd_pitch = ((h->dd_pitch + 255) / 256) * 256;  // whole 4-trip chunks (k_field_lp)
  // the lp / cluster index streams hold byte offsets j << 6 in 32-bit words: the record table must stay below 2^31 bytes
  if ((h->sweep_kernel == 2 || h->sweep_kernel == 3) && ((long long)n + 1) * (long long)sizeof(AtomRec) >= (1ll << 31))
    throw InputError("more than 2^25 atoms on one handle: the 32-bit record offsets of the sweep's index stream would wrap (shard the system)");
  h->d_nl_cnt.ensure(n + 1); h->d_dd_cnt.ensure(n + 1);
  h->d_nl_j.ensure((size_t)n * h->nl_pitch + 64);
  if (h->sweep_kernel < 3) h->d_dd_j.ensure((size_t)n * h->dd_pitch + 1024);  // slack: k_field_lp requests two index chunks per row up front
  else h->d_dd_j.ensure(64);
  // What the sweep streams per pair.  Measured (tools/exp_nocache.sh): while index + r^2 of all pairs
  // (12 B/pair) stay resident in the 256 MB Infinity Cache between sweeps the cached r^2 wins (36k atoms,
  // 160 MB: 99 vs 107 us/sweep); beyond that the stream comes from HBM every sweep and rebuilding r^2 from
  // the gathered positions wins (135k atoms, 595 MB: 320 vs 350 us/sweep).
  int mode = h->cache_r2;
  if (h->sweep_kernel == 1) mode = 0;
  if (h->sweep_kernel == 2) mode = 3;  // lane-per-pair sweep: only the index (as a byte offset) is streamed
  if (h->sweep_kernel >= 3) mode = 4;  // cluster sweep: the dd lists are the clusters' union lists (build_cluster_lists); tile sweep: build_tiles
  if (mode < 0) {
    const double est_pairs = h->dd_pairs > 0 ? (double)h->dd_pairs : 0.35 * (double)own_n(h) * (double)h->dd_pitch;
    mode = (12.0 * est_pairs < 200.0e6) ? 1 : 2;
  }
  h->stream_mode = mode;
  const bool r2c = mode == 1;
  if (mode == 1) h->d_dd_r2.ensure((size_t)n * h->dd_pitch + 64);
  else if (mode == 0) h->d_dd_s.ensure((size_t)n * h->dd_pitch + 64);
  double *r2p = r2c ? h->d_dd_r2.p : nullptr;
  double2 *sp = mode == 0 ? h->d_dd_s.p : nullptr;
  bool fuse = mode != 0 && mode != 4;  // r^2 and padding written by k_nl_build (no k_dd_scalars pass)
#ifdef POLAR_LAB
  if (getenv("POLAR_NO_FUSE_R2")) fuse = false;
#endif
  const double cutallsq = cutall * cutall, ddsq = mode == 4 ? -1.0 : st.dd_cutoff * st.dd_cutoff;
  const int nr = own_n(h);
  const int *rows = own_rows(h);
  zero_many(s, {{h->d_nl_cnt.p, (size_t)(n + 1) * sizeof(int)}, {h->d_dd_cnt.p, (size_t)(n + 1) * sizeof(int)},
                {h->d_overflow.p, 16 * sizeof(int)}, {h->d_ddtot.p, 64 * 16 * sizeof(unsigned long long)}});
  const bool lp = mode == 3;
  if (lp) h->d_dd_wrap.ensure(n + 1);
  const bool recheck = h->colors_valid && h->colors_recheck && h->sweep_kernel < 3 && (int)h->h_color.size() == n;
  if (h->colors_recheck && !recheck) { h->colors_valid = false; h->colors_recheck = false; }
  if (recheck) {
    h->d_color_s.ensure(n + 1);
    k_color_map<<<nblk(n, 256), 256, 0, s>>>(n, h->d_perm.p, h->d_color_orig.p, h->d_color_s.p);
  }
#define NLB(TRI, RC)                                                                                                          \
  k_nl_build<TRI, RC><<<nblk(nr, POLAR_ROWS_PER_BLOCK), POLAR_BLOCK, 0, s>>>(                                                 \
      rows, nr, h->d_pos4.p, h->box, g, h->d_cell_first.p, cutallsq, ddsq, h->nl_pitch, h->dd_pitch, h->d_nl_cnt.p,            \
      h->d_dd_cnt.p, h->d_nl_j.p, h->d_dd_j.p, fuse ? r2p : nullptr, fuse ? 1 : 0, lp ? (6 | (h->lp_quad_major << 8)) : 0,    \
      lp ? n : -1, lp ? h->d_dd_wrap.p : nullptr, recheck ? h->d_color_s.p : nullptr, h->color_keep * h->color_keep,           \
      h->d_overflow.p + 8, lp ? h->d_slot.p : nullptr, h->d_overflow.p, h->d_ddtot.p)
  // the box shape and "colours are being re-validated" are compile-time: the kernel is bound by its vector instructions
  if (h->box.triclinic) { if (recheck) NLB(true, true); else NLB(true, false); }
  else                  { if (recheck) NLB(false, true); else NLB(false, false); }
#undef NLB
  const RowList ddl{h->d_dd_cnt.p, h->dd_pitch};
  if (fuse || mode == 4) {
    // modes 1 and 2: the list build wrote r^2 (mode 1) and the padding itself; mode 4: no per-atom dd rows at all
  }
#ifdef POLAR_LAB
  else if (st.damping_type == POLAR_DAMP_EXPONENTIAL)
    k_dd_scalars<0><<<nblk(nr, POLAR_ROWS_PER_BLOCK), POLAR_BLOCK, 0, s>>>(rows, nr, h->d_rec0.p, h->box, ddl, h->d_dd_j.p, st.polar_damp, sp, r2p);
  else
    k_dd_scalars<1><<<nblk(nr, POLAR_ROWS_PER_BLOCK), POLAR_BLOCK, 0, s>>>(rows, nr, h->d_rec0.p, h->box, ddl, h->d_dd_j.p, st.polar_damp, sp, r2p);
#endif
  // overflow flag and dd total come back with the end-of-step read (no sync here)
  HIPCHECK(hipMemcpyAsync(h->h_flags, h->d_overflow.p, sizeof(int), hipMemcpyDeviceToHost, s));
  HIPCHECK(hipMemcpyAsync(h->h_ddtot, h->d_ddtot.p, 64 * 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
}
template <bool AP>
static void launch_rank(polar_handle *h, int pass) {
  const int n = h->nlocal, ntot = h->nlocal + h->nghost;
  dim3 grid(nblk(n, POLAR_ROWS_PER_BLOCK)), block(POLAR_BLOCK);
  if (pass == 2) k_fold_scal<<<1, POLAR_NSLOT, 0, h->stream>>>(h->d_scal.p, h->d_slots.p, 1);
  // pass 1 of the all-pairs form: the partners of an atom over several waves while there are few atoms (a minimum: any order)
  const int nchunk = (AP && pass == 1) ? std::max(1, std::min(16, 8192 / std::max(n, 1))) : 1;
  if (pass == 1)
    k_rank<AP, 1><<<nblk((long long)n * nchunk, POLAR_ROWS_PER_BLOCK), block, 0, h->stream>>>(n, ntot, h->d_x.p, h->d_alpha.p, h->d_mol.p, h->box, RowList{h->d_nl_cnt.p, h->nl_pitch},
                                                 h->d_nl_j.p, h->d_rec0.p, h->d_mol_s.p, h->d_scal.p, h->d_slots.p, h->d_rank.p, nchunk);
  else
    k_rank<AP, 2><<<grid, block, 0, h->stream>>>(n, ntot, h->d_x.p, h->d_alpha.p, h->d_mol.p, h->box, RowList{h->d_nl_cnt.p, h->nl_pitch},
                                                 h->d_nl_j.p, h->d_rec0.p, h->d_mol_s.p, h->d_scal.p, h->d_slots.p, h->d_rank.p, 1);
}

void launch_rank_pass(polar_handle *h, bool allpairs, int pass) {
  if (allpairs) launch_rank<true>(h, pass); else launch_rank<false>(h, pass);
}

template <bool AP, int DAMP, int EP>
void launch_field(polar_handle *h, int nrows, const int *rows) {
  const polar_settings &st = h->ph.st;
  if (nrows <= 0) return;
  const int fb = h->field_block;  // threads per workgroup = 64 x rows that share one L1
  const size_t lds = AP ? 0 : (size_t)(fb / 64) * 64 * 5 * sizeof(double2);  // per-wave staging tiles (list mode)
  k_field<AP, DAMP, EP><<<nblk(nrows, fb / 64), fb, lds, h->stream>>>(
      nrows, rows, h->nlocal, h->d_rec0.p, h->d_rec1.p, h->box, RowList{h->d_dd_cnt.p, h->dd_pitch}, h->d_dd_j.p, h->d_dd_s.p,
      st.dd_cutoff * st.dd_cutoff, st.polar_damp, h->d_ef_s.p, h->d_F.p, h->d_scal.p, h->d_slots.p
#ifdef POLAR_LAB
      , h->ablate
#endif
      );
}
template <int EP>
void launch_field_dyn(polar_handle *h, bool ap, int nrows, const int *rows) {
  const bool expd = h->ph.st.damping_type == POLAR_DAMP_EXPONENTIAL;
  if (ap) { if (expd) launch_field<true, 0, EP>(h, nrows, rows); else launch_field<true, 1, EP>(h, nrows, rows); }
#ifdef POLAR_LAB
  else    { if (expd) launch_field<false, 0, EP>(h, nrows, rows); else launch_field<false, 1, EP>(h, nrows, rows); }
#else
  else throw std::logic_error("k_field's list branch exists in the lab build only");
#endif
}

#ifdef POLAR_LAB
#include "lab/launch_quad.inc"
#else
template <int EP> inline void launch_field_quad(polar_handle *, int, const int *) { throw std::logic_error("lab build only"); }
#endif  // POLAR_LAB

// list-mode sweep, lane-per-pair with LDS-DMA gathers (k_field_lp), one wave per row
// row descriptors of this step for k_field_lp: the colour phases back to back (GS), or the own rows (Jacobi)
void prepare_lp(polar_handle *h) {
  const polar_settings &st = h->ph.st;
  const bool gs = st.polar_gs || st.polar_gs_ranked;
  const int tot = gs ? (h->color_off.empty() ? 0 : h->color_off.back()) : own_n(h);
  h->d_lpdesc.ensure((size_t)tot + 1);
  h->lp_npart = 0;
  if (deterministic(h)) {  // one partial sum per 256 rows of every launch; launches are 256-aligned in the array
    h->d_lp_pend.ensure(4 * (size_t)tot + 4);
    const int nlaunch = gs ? (int)h->color_off.size() - 1 : 1;
    h->lp_npart = tot / 256 + nlaunch + 1;
    h->d_lp_part.ensure((size_t)h->lp_npart + 1);
    HIPCHECK(hipMemsetAsync(h->d_lp_part.p, 0, (size_t)h->lp_npart * sizeof(double), h->stream));
  }
  if (tot > 0)
    k_lp_desc<<<nblk(tot, 256), 256, 0, h->stream>>>(tot, gs ? h->d_rows.p : own_rows(h), RowList{h->d_dd_cnt.p, h->dd_pitch},
                                                     h->d_dd_wrap.p, h->d_lpdesc.p);
}
template <int EP>
void launch_field_lp(polar_handle *h, int nrows, const int2 *desc) {
  const long long row0 = desc - h->d_lpdesc.p;  // launch row of the first descriptor = its dd row
  if (nrows <= 0) return;
  const polar_settings &st = h->ph.st;
  const int qb = h->quad_block;
  const bool det = deterministic(h);
  const int nt = det ? 2 : h->lp_tiles;   // (the one-tile instance is a lab variant of the in-place sweep only: the LDS request must match the instance launched)
  size_t lds = (size_t)(qb / 64) * nt * POLAR_LP_TILE;
  if (h->lp_wg_per_cu > 0) lds = std::max(lds, std::min((size_t)64 * 1024, (size_t)160 * 1024 / h->lp_wg_per_cu));  // lab: cap the residency
  // the launch requests dynamic LDS without raising the kernel's limit: beyond 64 KB (workgroups of more than 512 threads
  // with two tiles per wave) it would fail, and the failure would only surface at the next read of the loop state
  if (lds > (size_t)64 * 1024) throw InputError("k_field_lp: workgroup size x tiles needs more than 64 KB of LDS (POLAR_QUAD_BLOCK <= 512 with two tiles)");
  const double omega = EP == EP_INPLACE ? st.polar_sor : 1.0;
#ifdef POLAR_LAB
#define POLAR_LAB_ARG , h->ablate
#else
#define POLAR_LAB_ARG
#endif
#define FL(D, NT, DT) k_field_lp<EP, D, NT, DT><<<nblk_xcd(nrows, qb / 64), qb, lds, h->stream>>>(                   \
      nrows, row0, desc, h->d_rec0.p, h->d_rec1.p, h->box, h->dd_pitch, h->d_dd_j.p,                               \
      st.polar_damp, make_expcoef(), h->d_ef_s.p, h->d_scal.p, h->d_slots.p, omega, h->d_lp_pend.p POLAR_LAB_ARG)
  const bool expd = st.damping_type == POLAR_DAMP_EXPONENTIAL;
#ifdef POLAR_LAB
#include "lab/launch_lpa.inc"
#endif
#ifdef POLAR_LAB
#include "lab/launch_lpr.inc"
#endif
  if (det) {
    if (expd) FL(0, 2, true); else FL(1, 2, true);
    // the launch only read the record table: its rows' new dipoles and the sum of their changes are folded in now
    // (launch `l` of the sweep starts at row0: its partial sums start at slot row0 / 256 + l -- disjoint for every launch)
    int launch_no = 0;
    if (EP == EP_INPLACE) while (launch_no + 1 < (int)h->color_off.size() && h->color_off[launch_no] < row0) launch_no++;
    k_lp_commit<<<nblk(nrows, 256), 256, 0, h->stream>>>(nrows, row0, desc, h->d_lp_pend.p, h->d_rec0.p, h->d_rec1.p, EP == EP_JACOBI ? 1 : 0,
                                                      h->d_scal.p, h->d_lp_part.p + row0 / 256 + launch_no);
  } else {
    if (expd) FL(0, 2, false); else FL(1, 2, false);
  }
#undef FL
#undef POLAR_LAB_ARG
}

#ifdef POLAR_LAB
#include "lab/launch_cluster_tile.inc"
#else
template <int EP> inline void launch_field_cl(polar_handle *, int, int) { throw std::logic_error("lab build only"); }
inline size_t tile_lds_bytes(int) { return 0; }
inline int tile_lds_cap() { return 0; }
inline void build_tiles(polar_handle *) { throw std::logic_error("lab build only"); }
template <int EP> inline void launch_field_tile(polar_handle *, const TileLaunch &) { throw std::logic_error("lab build only"); }
#endif  // POLAR_LAB

// one sweep over the rows this handle owns (Jacobi, or the colour phases)
#ifdef POLAR_LAB
#include "lab/build_units_paired_rows.inc"
#endif

void sweep_once(polar_handle *h, bool ap) {
  const polar_settings &st = h->ph.st;
  const bool gs = st.polar_gs || st.polar_gs_ranked;
  if (!ap && h->sweep_kernel == 4) {
    if (!gs) { launch_field_tile<EP_JACOBI>(h, h->tile_all); return; }
    const int ncol = (int)h->tile_launches.size();
    const int c0 = h->part_n > 1 ? ncol * h->part_k / h->part_n : 0, c1 = h->part_n > 1 ? ncol * (h->part_k + 1) / h->part_n : ncol;
    for (int c = c0; c < c1; c++) launch_field_tile<EP_INPLACE>(h, h->tile_launches[c]);  // (polar_step_sweep_part: a window of the tile colours)
    return;
  }
  if (!ap && h->sweep_kernel == 3) {
    const int ncol = (int)h->color_off.size() - 1;
    for (int c = 0; c < ncol; c++) {
      if (gs) launch_field_cl<EP_INPLACE>(h, h->color_off[c + 1] - h->color_off[c], h->color_off[c]);
      else launch_field_cl<EP_JACOBI>(h, h->color_off[c + 1] - h->color_off[c], h->color_off[c]);
    }
    return;
  }
#ifdef POLAR_LAB
  if (!ap && h->sweep_kernel == 2 && h->lp_pairs && !deterministic(h) && h->part_n <= 1) {
    if (!gs) { launch_field_lp2<EP_JACOBI>(h, 0); return; }
    for (int c = 0; c + 1 < (int)h->unit_off.size(); c++) launch_field_lp2<EP_INPLACE>(h, c);
    return;
  }
#endif
  if (!ap && h->sweep_kernel == 2) {
    if (!gs) { launch_field_lp<EP_JACOBI>(h, own_n(h), h->d_lpdesc.p); return; }
    const int ncol = (int)h->color_off.size() - 1;
    const int c0 = h->part_n > 1 ? ncol * h->part_k / h->part_n : 0, c1 = h->part_n > 1 ? ncol * (h->part_k + 1) / h->part_n : ncol;
    for (int c = c0; c < c1; c++)  // (polar_step_sweep_part: a window of the colour phases)
      launch_field_lp<EP_INPLACE>(h, h->color_off[c + 1] - h->color_off[c], h->d_lpdesc.p + h->color_off[c]);
    return;
  }
  if (!ap && h->sweep_kernel == 0) {
    if (!gs) { launch_field_quad<EP_JACOBI>(h, own_n(h), own_rows(h)); return; }
    const int ncol = (int)h->color_off.size() - 1;
    for (int c = 0; c < ncol; c++)
      launch_field_quad<EP_INPLACE>(h, h->color_off[c + 1] - h->color_off[c], h->d_rows.p + h->color_off[c]);
    return;
  }
  if (!gs) {
    launch_field_dyn<EP_JACOBI>(h, ap, own_n(h), own_rows(h));
    return;
  }
  const int ncol = (int)h->color_off.size() - 1;
  for (int c = 0; c < ncol; c++) {
    const int cnt = h->color_off[c + 1] - h->color_off[c];
    if (cnt <= 0) continue;
    launch_field_dyn<EP_INPLACE>(h, false, cnt, h->d_rows.p + h->color_off[c]);
  }
}

// one colour phase of the list-mode Gauss-Seidel (row sweep): part 0 = all its rows, 1 = its boundary rows (rows whose
// dipoles a peer of a multi-GPU run receives: they come first in the phase), 2 = its interior rows.  `deterministic yes`
// keeps a phase in one launch (its partial sums are laid out per launch): part 1 = everything, part 2 = nothing.
void sweep_phase(polar_handle *h, int c, int part) {
  const int ncol = (int)h->color_off.size() - 1;
  if (c < 0 || c >= ncol) throw std::logic_error("sweep_phase: no such colour");
  const int a = h->color_off[c], b = h->color_off[c + 1];
  const int m = (h->color_nsub == 2 && !deterministic(h)) ? h->color_sub[2 * c + 1] : b;
  const int lo = part == 2 ? m : a, hi = part == 1 ? m : b;
  launch_field_lp<EP_INPLACE>(h, hi - lo, h->d_lpdesc.p + lo);
}
template <bool AP, int DAMP>
void launch_force(polar_handle *h, int eflag, int vglobal, double *vatom, double *fdst) {
  const bool vpair = vglobal || vatom;
  const polar_settings &st = h->ph.st;
  dim3 grid(nblk(own_n(h), POLAR_ROWS_PER_BLOCK)), block(POLAR_BLOCK);
  const double ccs = st.cut_coul * st.cut_coul, dds = st.dd_cutoff * st.dd_cutoff, e2s = std::sqrt(h->P.qqrd2e);
  double *dbg6 = nullptr;   // `debug yes`: polarization force on the caller's atom 0 and its dipole-dipole part (PS.cpp:637-638)
  if (st.debug) {
    h->d_dbgf.ensure(8);
    HIPCHECK(hipMemsetAsync(h->d_dbgf.p, 0, 8 * sizeof(double), h->stream));
    dbg6 = h->d_dbgf.p;
  }
#define LF(E, V)                                                                                                    \
  k_polar_force<AP, DAMP, E, V><<<grid, block, 0, h->stream>>>(own_rows(h), own_n(h), h->sorted ? h->d_perm.p : nullptr, h->nlocal, h->d_scal.p, h->d_rec0.p, h->d_rec1.p,  \
                                                               h->d_mol_s.p, h->box, RowList{h->d_nl_cnt.p, h->nl_pitch}, h->d_nl_j.p,  \
                                                               ccs, dds, st.polar_damp, e2s, fdst, h->d_slots.p, vatom, vglobal, make_expcoef(),  \
                                                               dbg6)
  if (eflag) { if (vpair) LF(true, true); else LF(true, false); }
  else       { if (vpair) LF(false, true); else LF(false, false); }
#undef LF
}

void read_scal(polar_handle *h) {
  HIPCHECK(hipGetLastError());  // a failed kernel launch must not go unnoticed
  HIPCHECK(hipMemcpyAsync(h->h_scal, h->d_scal.p, sizeof(Scal), hipMemcpyDeviceToHost, h->stream));
  HIPCHECK(hipStreamSynchronize(h->stream));
}
#ifdef POLAR_LAB
#include "lab/build_cluster_lists.inc"
#else
void build_cluster_lists(polar_handle *) { throw std::logic_error("lab build only"); }
#endif  // POLAR_LAB
// `debug yes` (PS.cpp:1182-1191): u_polar after sweep `sw`, kept on the device until polar_get_debug_trace
void debug_trace(polar_handle *h, int sw, bool /*jacobi*/) {
  if (!h->ph.st.debug) return;
  h->d_trace.ensure((size_t)h->ph.st.iterations_max + 8);
  if (sw > h->ph.st.iterations_max + 1) return;
  const MuView mv = mu_view(h);
  k_debug_upolar<<<1, 1024, 0, h->stream>>>(h->nlocal, h->d_scal.p, mv.a, mv.b, mv.stride, h->d_ef_s.p, h->d_trace.p, sw, 0);  // Jacobi: like the reference, the value is formed BEFORE "mu = mu_new" (jacobi unused)
  h->ntrace = sw + 1;
}

// `polar_accel m` (extension keyword): Anderson mixing of depth m on the sweep map of the list-mode Gauss-Seidel
// (kernels and formulas: polar_accel.hpp, k_accel_*).  accel_begin: buffers, x_0 = the initial guess of the own rows; returns
// whether the keyword applies to this solve.  accel_step: after a sweep (and its end-of-sweep decision) -- differences and
// their dot products, the m x m solve, the mixed iterate into the records.  `global_dots`: multi-GPU, the all-reduced dot
// products in device memory ([2 * POLAR_ACCEL_MAXM] doubles); the local ones are then exported first (accel_export).
bool accel_begin(polar_handle *h, bool ap) {
  const polar_settings &st = h->ph.st;
  const bool gs = st.polar_gs || st.polar_gs_ranked;
  if (st.polar_accel <= 0) return false;
  if (ap || !gs || h->sweep_kernel != 2) throw InputError("polar_accel needs the Gauss-Seidel sweep of list mode (dd_cutoff > 0 with polar_gs or polar_gs_ranked)");
  if (deterministic(h)) throw InputError("polar_accel and `deterministic yes` exclude each other");
  const int tot = h->color_off.empty() ? 0 : h->color_off.back();
  const long long pitch = ((long long)tot + 255) / 256 * 256;
  const int M = POLAR_ACCEL_MAXM;
  h->acc_pitch = pitch; h->acc_rows = tot;
  const bool fresh = h->d_acc_dF.cap < (size_t)M * 3 * pitch + 1;
  h->d_acc_x.ensure(3 * pitch + 1); h->d_acc_f.ensure(3 * pitch + 1); h->d_acc_g.ensure(3 * pitch + 1);
  h->d_acc_dF.ensure((size_t)M * 3 * pitch + 1); h->d_acc_dG.ensure((size_t)M * 3 * pitch + 1);
  h->d_acc_part.ensure((size_t)2 * M * nblk(tot, 256) + 2 * M); h->d_acc_state.ensure(1);
  if (fresh) {  // (slots outside the window are multiplied by zero, never skipped: no NaN may sit there)
    HIPCHECK(hipMemsetAsync(h->d_acc_dF.p, 0, h->d_acc_dF.cap * sizeof(double), h->stream));
    HIPCHECK(hipMemsetAsync(h->d_acc_dG.p, 0, h->d_acc_dG.cap * sizeof(double), h->stream));
  }
  k_accel_init<<<nblk(tot, 256), 256, 0, h->stream>>>(tot, pitch, h->d_lpdesc.p, h->d_rec0.p, h->d_acc_x.p, h->d_acc_state.p);
  return true;
}
void accel_export(polar_handle *h, double *dev_local_dots) {
  const int tot = h->acc_rows, nb = nblk(tot, 256), ring = std::min(h->ph.st.polar_accel, POLAR_ACCEL_MAXM);
  k_accel_diff<POLAR_ACCEL_MAXM><<<nb, 256, 0, h->stream>>>(tot, h->acc_pitch, h->d_lpdesc.p, h->d_rec0.p, h->d_scal.p, h->d_acc_state.p, h->d_acc_x.p,
                                                           h->d_acc_f.p, h->d_acc_g.p, h->d_acc_dF.p, h->d_acc_dG.p, h->d_acc_part.p, ring);
  k_accel_solve<POLAR_ACCEL_MAXM><<<1, 256, 0, h->stream>>>(nb, h->d_acc_part.p, h->d_scal.p, h->d_acc_state.p, dev_local_dots, nullptr, ring);
}
// Round 5, precision mode: the single-workgroup launches around the mixing folded together (polar_accel.hpp).
// accel_export_fused: differences + partial dots, then this rank's sum (dmu)^2 into ared[0] and its dot products into ared[1 ..]
// (multi-GPU, before the all-reduce).  accel_decide_mix: the end-of-sweep decision -- on ared[0] if given -- and the mixing
// coefficients in one launch, then the mix; `diff_first`: single handle, where nothing has formed the differences yet.
void accel_export_fused(polar_handle *h, double *ared) {
  const int tot = h->acc_rows, nb = nblk(tot, 256), ring = std::min(h->ph.st.polar_accel, POLAR_ACCEL_MAXM);
  k_accel_diff<POLAR_ACCEL_MAXM><<<nb, 256, 0, h->stream>>>(tot, h->acc_pitch, h->d_lpdesc.p, h->d_rec0.p, h->d_scal.p, h->d_acc_state.p, h->d_acc_x.p,
                                                           h->d_acc_f.p, h->d_acc_g.p, h->d_acc_dF.p, h->d_acc_dG.p, h->d_acc_part.p, ring);
  k_accel_export<POLAR_ACCEL_MAXM><<<1, POLAR_NSLOT, 0, h->stream>>>(h->d_scal.p, h->d_slots.p, ared, 1, nb, h->d_acc_part.p, h->d_acc_state.p, ring);
}
void accel_decide_mix(polar_handle *h, const double *ared, bool diff_first) {
  const polar_settings &st = h->ph.st;
  const int tot = h->acc_rows, nb = nblk(tot, 256), ring = std::min(st.polar_accel, POLAR_ACCEL_MAXM);
  if (diff_first)
    k_accel_diff<POLAR_ACCEL_MAXM><<<nb, 256, 0, h->stream>>>(tot, h->acc_pitch, h->d_lpdesc.p, h->d_rec0.p, h->d_scal.p, h->d_acc_state.p, h->d_acc_x.p,
                                                             h->d_acc_f.p, h->d_acc_g.p, h->d_acc_dF.p, h->d_acc_dG.p, h->d_acc_part.p, ring);
  k_solver_accel<POLAR_ACCEL_MAXM><<<1, POLAR_NSLOT, 0, h->stream>>>(h->d_scal.p, h->d_slots.p, norm_count(h), st.iterations_max, st.polar_precision, ared, 1, nb,
                                                                     h->d_acc_part.p, h->d_acc_state.p, ring);
  k_accel_mix<POLAR_ACCEL_MAXM><<<nb, 256, 0, h->stream>>>(tot, h->acc_pitch, h->d_lpdesc.p, h->d_rec0.p, h->d_scal.p, h->d_acc_state.p, h->d_acc_x.p,
                                                          h->d_acc_g.p, h->d_acc_dG.p);
}
void accel_step(polar_handle *h, const double *global_dots) {
  const int tot = h->acc_rows, nb = nblk(tot, 256), ring = std::min(h->ph.st.polar_accel, POLAR_ACCEL_MAXM);
  if (!global_dots)
    k_accel_diff<POLAR_ACCEL_MAXM><<<nb, 256, 0, h->stream>>>(tot, h->acc_pitch, h->d_lpdesc.p, h->d_rec0.p, h->d_scal.p, h->d_acc_state.p, h->d_acc_x.p,
                                                             h->d_acc_f.p, h->d_acc_g.p, h->d_acc_dF.p, h->d_acc_dG.p, h->d_acc_part.p, ring);
  k_accel_solve<POLAR_ACCEL_MAXM><<<1, 256, 0, h->stream>>>(nb, h->d_acc_part.p, h->d_scal.p, h->d_acc_state.p, nullptr, global_dots, ring);
  k_accel_mix<POLAR_ACCEL_MAXM><<<nb, 256, 0, h->stream>>>(tot, h->acc_pitch, h->d_lpdesc.p, h->d_rec0.p, h->d_scal.p, h->d_acc_state.p, h->d_acc_x.p,
                                                          h->d_acc_g.p, h->d_acc_dG.p);
}

void solve(polar_handle *h, bool ap, polar_result *out) {
  const polar_settings &st = h->ph.st;
  const int n = h->nlocal;
  hipStream_t s = h->stream;
  const bool gs = st.polar_gs || st.polar_gs_ranked;
  const bool expd = st.damping_type == POLAR_DAMP_EXPONENTIAL;
  const int max_sweeps = st.iterations_max + 1;
  const int check_every = 4;
  out->ncolors = 0;
  if (st.polar_accel > 0 && (ap || !gs)) throw InputError("polar_accel needs the Gauss-Seidel sweep of list mode (dd_cutoff > 0 with polar_gs or polar_gs_ranked)");

  if (!gs || !ap) {  // Jacobi (reference "polar_gs no") or colour-phase Gauss-Seidel over the dd list
    const bool clm = !ap && h->sweep_kernel == 3, tile = !ap && h->sweep_kernel == 4;  // tile sweep: no host-side colours at all
    if (!ap && !tile) resolve_colors(h);
    if (!tile && (gs || clm) && !h->colors_valid) { ensure_colors(h); if (clm) build_cluster_lists(h); else map_color_rows(h); }
    if (!ap && gs && h->sweep_kernel == 2 && !slots_current(h)) { compute_slots(h); build_lists(h); }  // rows into launch order
    if (gs) out->ncolors = tile ? (int)h->tile_launches.size() : (int)h->color_off.size() - 1;
    if (!ap && h->sweep_kernel == 2) prepare_lp(h);
#ifdef POLAR_LAB
    if (!ap && h->sweep_kernel == 2 && h->lp_pairs && !deterministic(h)) build_units(h);
#endif
    // fixed-iteration GS takes no decision between sweeps: its end-of-sweep logic is applied in two
    // launches (all sweeps but the last, then the last one, whose sum |dmu|^2 is the one reported)
    const bool lazy = st.fixed_iteration && gs;
    const bool accel = accel_begin(h, ap);   // `polar_accel m`: Anderson mixing between the sweeps
    for (int sw = 0; sw < max_sweeps; sw++) {
      sweep_once(h, ap);
      debug_trace(h, sw, !gs);
      if (accel && lazy && sw < max_sweeps - 1) accel_step(h, nullptr);   // (no decision to wait for; nothing after the last sweep)
      if (lazy && sw < max_sweeps - 2) continue;
      const int count = (lazy && sw == max_sweeps - 2) ? max_sweeps - 1 : 1;
      if (accel && !lazy) accel_decide_mix(h, nullptr, true);   // differences, [decision + coefficients], mix: the mix is a no-op once the stop rule has fired (the result is G(x_k) of the last sweep)
      else k_solver_step<<<1, POLAR_NSLOT, 0, s>>>(h->d_scal.p, h->d_slots.p, norm_count(h), st.fixed_iteration, st.iterations_max, st.polar_precision, gs ? 0 : 1, nullptr, count, det_part(h), det_npart(h));
      if (!st.fixed_iteration && look_at_state(h, sw, check_every)) {
        read_scal(h);
        if (h->h_scal->done) break;
      }
    }
  } else if (h->dense_gs) {  // exact-order blocked Gauss-Seidel on the HBM-resident tensor
    const long long np = ((long long)n + 63) / 64 * 64;   // row pitch of the component-major tensor
    h->d_T6.ensure((size_t)n * 6 * np + 64);
    if (expd) k_build_T6<0><<<nblk(n, POLAR_ROWS_PER_BLOCK), POLAR_BLOCK, 0, s>>>(n, np, h->d_rec0.p, h->box, st.polar_damp, h->d_T6.p);
    else      k_build_T6<1><<<nblk(n, POLAR_ROWS_PER_BLOCK), POLAR_BLOCK, 0, s>>>(n, np, h->d_rec0.p, h->box, st.polar_damp, h->d_T6.p);
    k_dense_field<<<nblk(n, POLAR_ROWS_PER_BLOCK), POLAR_BLOCK, 0, s>>>(n, np, h->d_T6.p, h->d_rec0.p, h->d_F.p);
    // ONE launch per block of B atoms, nothing sequential inside it: d = G cb - N d' (polar_exact.hpp, k_gs_blk).  G and N
    // are formed here, once per step.
    const int B = n >= 1024 ? 256 : n >= 384 ? 128 : 64, R = 3 * B, nb = (n + B - 1) / B;   // (at least two blocks: dense_gs needs n > 64)
    const size_t RR = (size_t)R * R;
    h->d_Minv.ensure(nb * RR); h->d_gsN.ensure(nb * RR); h->d_gsAT.ensure(nb * RR); h->d_cb.ensure(3 * (size_t)n + 3);
    h->d_dmu.ensure(2 * (size_t)R + 8);
    if (B > 64) HIPCHECK(hipMemsetAsync(h->d_Minv.p, 0, nb * RR * sizeof(double), s));   // (what lies right of the diagonal pieces)
    k_gs_blockinv<<<nb * (B / 64) * (POLAR_GS64 / POLAR_GSB_COLS), 64, 0, s>>>(n, np, h->d_T6.p, h->d_rec0.p, h->d_Minv.p, R, (long long)RR, B / 64);
    for (int m = 64; m < B; m *= 2) {
      // two triangles of m atoms side by side on G's diagonal become one of 2 m: G = [[G1, 0], [-G2 (A T21) G1, G2]];
      // Y = (A T21) G1, then the piece below G1 = -G2 Y.  All pairs of all blocks in one batch.
      const int per = B / (2 * m), nz = nb * per, q = 3 * m;
      const long long inner = (long long)2 * q * (R + 1), qq = (long long)q * q;
      double *AT21 = h->d_gsAT.p, *Y = h->d_gsN.p;   // (scratch here; filled for good below)
      const GsBatch dense{qq, 0, 1}, diag{(long long)RR, inner, per};
      k_gs_expand<<<dim3(m / 64, m / 4, nz), 256, 0, s>>>(n, np, h->d_T6.p, h->d_rec0.p, B, nb, 1, m, AT21, q, qq);
      k_gs_gemm<false><<<dim3(q / 64, q / 64, nz), 256, 0, s>>>(q, 1.0, AT21, q, dense, h->d_Minv.p, R, diag, Y, q, dense);
      k_gs_gemm<true><<<dim3(q / 64, q / 64, nz), 256, 0, s>>>(q, -1.0, h->d_Minv.p + (size_t)q * (R + 1), R, diag, Y, q, dense, h->d_Minv.p + (size_t)q * R, R, diag);
    }
    const GsBatch whole{(long long)RR, 0, 1};
    k_gs_expand<<<dim3(B / 64, B / 4, nb), 256, 0, s>>>(n, np, h->d_T6.p, h->d_rec0.p, B, nb, 0, B, h->d_gsAT.p, R, (long long)RR);
    k_gs_gemm<true><<<dim3(R / 64, R / 64, nb), 256, 0, s>>>(R, 1.0, h->d_Minv.p, R, whole, h->d_gsAT.p, R, whole, h->d_gsN.p, R, whole);
    k_gs_cb_init<<<nblk(n, 256), 256, 0, s>>>(n, h->d_rec0.p, h->d_ef_s.p, h->d_F.p, h->d_cb.p);
    int prev = -1, flip = 0;
    const int grid = R / POLAR_GS_WAVES + nblk(n, POLAR_GS_WAVES), npart = nb * (R / POLAR_GS_WAVES);
    h->d_gs_part.ensure((size_t)npart + 1);
    HIPCHECK(hipMemsetAsync(h->d_gs_part.p, 0, (size_t)npart * sizeof(double), s));
    // the sweep's last launch carries the end-of-sweep logic (GsTail) unless something has to happen between the two (`debug yes`)
    const bool tail_on = !st.debug;
    if (tail_on && !h->d_gs_cnt.p) { h->d_gs_cnt.ensure(8); HIPCHECK(hipMemsetAsync(h->d_gs_cnt.p, 0, 8 * sizeof(int), s)); }
    for (int sw = 0; sw < max_sweeps; sw++) {
      for (int b0 = 0; b0 < n; b0 += B) {
        const int next0 = b0 + B < n ? b0 + B : 0;
        const GsTail tail = (tail_on && b0 + B >= n) ? GsTail{h->d_gs_cnt.p, npart, norm_count(h), st.fixed_iteration, st.iterations_max, st.polar_precision}
                                                     : GsTail{nullptr, 0, 0, 0, 0, 0.0};
#define POLAR_GS_BLK(BB) k_gs_blk<BB><<<grid, 64 * POLAR_GS_WAVES, 0, s>>>(n, np, b0, prev, next0, h->d_T6.p, h->d_Minv.p, h->d_gsN.p, h->d_rec0.p, h->d_ef_s.p, h->d_F.p, h->d_cb.p, \
                                                            h->d_dmu.p + R * flip, h->d_dmu.p + R * (flip ^ 1), h->d_scal.p, h->d_gs_part.p, tail)
        if (B == 256) POLAR_GS_BLK(256); else if (B == 128) POLAR_GS_BLK(128); else POLAR_GS_BLK(64);
#undef POLAR_GS_BLK
        prev = b0; flip ^= 1;
      }
      debug_trace(h, sw, false);
      if (!tail_on) k_solver_step<<<1, POLAR_NSLOT, 0, s>>>(h->d_scal.p, h->d_slots.p, norm_count(h), st.fixed_iteration, st.iterations_max, st.polar_precision, 0, nullptr, 1, h->d_gs_part.p, npart);
      if (!st.fixed_iteration && look_at_state(h, sw, check_every)) {
        read_scal(h);
        if (h->h_scal->done) break;
      }
    }
  } else {  // exact-order blocked Gauss-Seidel, matrix-free (systems whose tensor does not fit)
    std::vector<int> order(n), pos(n);
    std::iota(order.begin(), order.end(), 0);
    if (st.polar_gs_ranked) {  // stable descending sort == the reference's bubble sort (PS.cpp:1130-1143)
      std::vector<double> rk(n);
      HIPCHECK(hipMemcpyAsync(rk.data(), h->d_rank.p, n * sizeof(double), hipMemcpyDeviceToHost, s));
      HIPCHECK(hipStreamSynchronize(s));
      std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return rk[a] > rk[b]; });
    }
    for (int k = 0; k < n; k++) pos[order[k]] = k;
    h->d_order.ensure(n); h->d_pos.ensure(n); h->d_dmu.ensure(3 * 64);
    HIPCHECK(hipMemcpyAsync(h->d_order.p, order.data(), n * sizeof(int), hipMemcpyHostToDevice, s));
    HIPCHECK(hipMemcpyAsync(h->d_pos.p, pos.data(), n * sizeof(int), hipMemcpyHostToDevice, s));
    HIPCHECK(hipStreamSynchronize(s));  // order/pos are stack vectors
    launch_field_dyn<EP_FIELD>(h, true, n, nullptr);
    for (int sw = 0; sw < max_sweeps; sw++) {
      for (int b0 = 0; b0 < n; b0 += 64) {
        if (expd) {
          k_gs_block_seq<0><<<1, 64, 0, s>>>(n, b0, h->d_order.p, h->d_rec0.p, h->box, st.polar_damp, h->d_ef_s.p, h->d_F.p, h->d_dmu.p, h->d_scal.p, h->d_slots.p);
          k_gs_block_push<0><<<nblk(n, POLAR_ROWS_PER_BLOCK), POLAR_BLOCK, 0, s>>>(n, b0, h->d_order.p, h->d_pos.p, h->d_rec0.p, h->box, st.polar_damp, h->d_dmu.p, h->d_F.p, h->d_scal.p);
        } else {
          k_gs_block_seq<1><<<1, 64, 0, s>>>(n, b0, h->d_order.p, h->d_rec0.p, h->box, st.polar_damp, h->d_ef_s.p, h->d_F.p, h->d_dmu.p, h->d_scal.p, h->d_slots.p);
          k_gs_block_push<1><<<nblk(n, POLAR_ROWS_PER_BLOCK), POLAR_BLOCK, 0, s>>>(n, b0, h->d_order.p, h->d_pos.p, h->d_rec0.p, h->box, st.polar_damp, h->d_dmu.p, h->d_F.p, h->d_scal.p);
        }
      }
      debug_trace(h, sw, false);
      k_solver_step<<<1, POLAR_NSLOT, 0, s>>>(h->d_scal.p, h->d_slots.p, norm_count(h), st.fixed_iteration, st.iterations_max, st.polar_precision, 0, nullptr, 1, nullptr, 0);
      if (!st.fixed_iteration && look_at_state(h, sw, check_every)) {
        read_scal(h);
        if (h->h_scal->done) break;
      }
    }
  }
}

// PS.cpp:125-386: everything before the solve
void phase_begin(polar_handle *h, int eflag, int vflag, const double *mu_host) {
  need_device(h);
  if (!h->types_set || !h->coul_set) throw std::runtime_error("polar_compute before the pair tables were set: polar_pair_init (or polar_set_types), then polar_set_coul with the Coulomb tables of Pair::init_tables unless pair_modify table 0");
  if (!h->box_set || !h->atoms_set || !h->neigh_set) throw std::runtime_error("polar_compute before polar_set_box/polar_set_atoms/polar_set_neighbors");
  const polar_settings &st = h->ph.st;
  const int n = h->nlocal, nall = h->nlocal + h->nghost;
  const bool ap = !(st.dd_cutoff > 0.0);
  if (ap && own_n(h) != n) throw InputError("row sharding needs dd_cutoff > 0 (exact all-pairs mode runs as replicas only)");
  if (!ap && h->box.triclinic && !(h->sweep_kernel == 4 || (h->sweep_kernel == 2 && h->lp_depth == 0)))
    throw InputError("dd_cutoff (list) mode in a triclinic box needs the row sweep (k_field_lp) or, in the lab build, the tile sweep");
  const int vmode = vflag % 4;
  hipStream_t s = h->stream;
  h->warn.clear();
  h->ntrace = 0;
  h->step_eflag = eflag; h->step_vflag = vflag;

  h->d_f.ensure(3 * (size_t)nall); h->d_ef.ensure(3 * (size_t)n); h->d_F.ensure(3 * (size_t)n);
  h->d_mu.ensure(3 * (size_t)n); h->d_rank.ensure(n); h->d_rec0.ensure(n + 1); h->d_rec1.ensure(n + 1);  // + the dummy record
  h->d_ef_s.ensure(3 * (size_t)n); h->d_mol_s.ensure(n + 1);
  HIPCHECK(hipEventRecord(h->ev[0], s));
  HIPCHECK(hipMemsetAsync(h->d_f.p, 0, 3 * (size_t)nall * sizeof(double), s));
  // per-atom tallies (eflag/2, vflag/4: src/pair.cpp:760-764), zeroed like ev_setup does (:789-806)
  double *eatom = nullptr, *vatom = nullptr;
  if (eflag / 2) { h->d_eatom.ensure(nall + 1); eatom = h->d_eatom.p; HIPCHECK(hipMemsetAsync(eatom, 0, (size_t)nall * sizeof(double), s)); }
  if (vflag / 4) { h->d_vatom.ensure(6 * (size_t)nall + 6); vatom = h->d_vatom.p; HIPCHECK(hipMemsetAsync(vatom, 0, 6 * (size_t)nall * sizeof(double), s)); }
  // (on a row-sharded handle every kernel tallies into the rows it walks -- half of each pair per row atom -- so the
  //  shards' arrays add up to the unsharded ones: LJ/Coulomb over the rows of the list the shard was given, polarization
  //  over the own rows)
  k_zero_slots<<<nblk(POLAR_NSLOT, 256), 256, 0, s>>>(h->d_slots.p, h->d_scal.p);
  auto launch_lj = [&]() {  // a3 -- on a low-priority side stream when overlap is on: it depends on nothing below, and fills
    // whatever the list build, the static field and the latency-bound solver launches leave idle
    hipStream_t ms = s;
    h->lj_forked = h->overlap_lj && h->inum > 0;
    if (h->lj_forked) {
      HIPCHECK(hipEventRecord(h->ev_fork, ms));
      HIPCHECK(hipStreamWaitEvent(h->lj_stream, h->ev_fork, 0));
    }
    hipStream_t s = h->lj_forked ? h->lj_stream : ms;  // shadows the main stream inside this block
    if (h->list_up_pending) {   // the uploaded list may still be on its way (upload_neighbor_rows returns before its transfer has ended)
      HIPCHECK(hipStreamWaitEvent(s, h->ev_list_up, 0));   // (every step until the next list: free once the event has fired)
    }
    HIPCHECK(hipEventRecord(h->ev_lj0, s));
    LJCoulParams P = h->P;
    P.newton_pair = h->newton_pair; P.nlocal = n; P.cut_coulsq = st.cut_coul * st.cut_coul; P.full_list = h->full_list; P.ablate = 0;
    h->d_xq.ensure(nall + 1);
    k_pack_lj<<<nblk(nall, 256), 256, 0, s>>>(nall, h->d_x.p, h->d_q.p, h->d_xq.p);
    const size_t ljlds = (size_t)(h->ntypes + 1) * (h->ntypes + 1) * 8 * sizeof(double);
    if (ljlds > 64 * 1024) throw InputError("too many atom types for the LDS-resident LJ table (max 31)");
    dim3 block(POLAR_BLOCK);
    bool symmetrise = !h->full_list;
#ifdef POLAR_LAB
    if (getenv("POLAR_LJ_ATOMICS")) symmetrise = false;  // round 1's FP64 atomics on f[j]
#endif
    // A full (newton-off) list puts no force on ghosts, so sum f.x over locals+ghosts (fdotr) would miss
    // the image terms: the LJ/Coulomb virial is then tallied pairwise (the same number), and only the
    // polarization forces -- local atoms, minimum-image displacements -- go through f.x (phase_finish)
    const bool lj_pairwise_virial = h->full_list && vmode == 2;
    if ((eatom || vatom) && !symmetrise && !h->full_list) throw InputError("per-atom tallies need the row-complete pair list");
    if (symmetrise && !h->sym_valid && h->inum > 0) {  // once per uploaded list (reneighbor steps)
      dim3 g0(nblk(h->inum, POLAR_ROWS_PER_BLOCK));
      h->d_sym_cnt.ensure(nall + 1); h->d_sym_fill.ensure(nall + 1); h->d_sym_first.ensure(nall + 2);
      h->d_sym_j.ensure(2 * (size_t)h->nneigh + 64);
      HIPCHECK(hipMemsetAsync(h->d_sym_cnt.p, 0, (nall + 1) * sizeof(int), s));
      HIPCHECK(hipMemsetAsync(h->d_sym_fill.p, 0, (nall + 1) * sizeof(int), s));
      k_sym_count<<<g0, block, 0, s>>>(h->inum, h->d_ilist.p, h->d_numneigh.p, h->d_first.p, h->d_neigh.p, h->d_sym_cnt.p);
      launch_scan(nall, h->d_sym_cnt.p, h->d_sym_first.p, h->d_scan_b, s);
      h->sym_typed = h->lj_typed && nall < (1 << 24) && h->ntypes < 64;
      k_sym_fill<<<g0, block, 0, s>>>(h->inum, h->d_ilist.p, h->d_numneigh.p, h->d_first.p, h->d_neigh.p, h->d_sym_first.p,
                                      h->d_sym_fill.p, h->d_sym_j.p, h->sym_typed ? h->d_type.p : nullptr);
      h->sym_valid = true;
    }
    // newton off: ghosts receive no force and tally nothing (PS.cpp:293, ev_tally's 0.5 per LOCAL atom), so only the
    // local rows of the symmetrised list are walked: a local-ghost pair then counts 0.5, a local-local pair 0.5 + 0.5
    const int nrows_lj = symmetrise ? (h->newton_pair ? nall : n) : h->inum;
    dim3 grid(nblk(nrows_lj, POLAR_ROWS_PER_BLOCK));
    if (symmetrise) P.full_list = 1;  // rows of the symmetrised list: force on the row atom only, tallies halved
    P.typed_list = symmetrise ? (h->sym_typed ? 1 : 0) : (h->device_list && h->dev_typed ? 1 : 0);
    const int *il = symmetrise ? nullptr : h->d_ilist.p;
    const int *nn = symmetrise ? nullptr : h->d_numneigh.p;
    const long long *fi = symmetrise ? h->d_sym_first.p : h->d_first.p;
    const int *nj = symmetrise ? h->d_sym_j.p : h->d_neigh.p;
    if (h->inum > 0) {
      const bool vrow = vmode == 1 || vatom || lj_pairwise_virial;
      // the persistent form with the Coulomb bins in LDS (k_ljcoul_pers) where the bins fit beside the LJ table and there are
      // rows enough to fill the chip's 16 waves per CU; else one wave per row with the bins in memory
      const size_t ljlds_al = ((ljlds / sizeof(double) + 3) & ~(size_t)3) * sizeof(double);
      const size_t plds = ljlds_al + ((size_t)32 << P.ncoultablebits);
      const bool pers = h->lj_pers && P.ncoultablebits > 0 && plds <= (size_t)160 * 1024 && h->ncu > 0 && (nrows_lj >= 16 * h->ncu || h->lj_pers >= 2);   // (POLAR_LJ_PERS=2: also for small systems -- tests)
      if (pers) {
#define LJP(E, V, T) do {                                                                                                              \
          static size_t raised[64] = {0};   /* (per template instance and device: the attribute belongs to the function ON a device) */ \
          size_t &rs = raised[h->device & 63];                                                                                          \
          if (rs < plds) { HIPCHECK(hipFuncSetAttribute((const void *)k_ljcoul_pers<E, V, T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)plds)); rs = plds; } \
          k_ljcoul_pers<E, V, T><<<h->ncu, h->lj_pers_threads, plds, s>>>(P, nrows_lj, il, nn, fi, nj, h->d_xq.p, h->d_type.p, h->d_f.p, h->d_slots.p, eatom, vatom, vmode == 1 || lj_pairwise_virial); \
        } while (0)
#define LJPT(E, V) do { if (h->lj_tab_arith) LJP(E, V, 2); else LJP(E, V, 1); } while (0)
        if (eflag) { if (vrow) LJPT(true, true); else LJPT(true, false); }
        else       { if (vrow) LJPT(false, true); else LJPT(false, false); }
#undef LJPT
#undef LJP
      } else {
#define LJ(E, V) k_ljcoul<E, V><<<grid, block, ljlds, s>>>(P, nrows_lj, il, nn, fi, nj, h->d_xq.p, h->d_type.p, h->d_f.p, h->d_slots.p, eatom, vatom, vmode == 1 || lj_pairwise_virial)
        if (eflag) { if (vrow) LJ(true, true); else LJ(true, false); }
        else       { if (vrow) LJ(false, true); else LJ(false, false); }
#undef LJ
      }
    }
    HIPCHECK(hipEventRecord(h->ev_lj1, s));
    if (h->lj_forked) HIPCHECK(hipEventRecord(h->ev_join, s));
  };
  bool lj_late = false;
#ifdef POLAR_LAB
  lj_late = getenv("POLAR_LJ_LATE") != nullptr;  // fork a3 after the static field instead
#endif
  if (!lj_late) launch_lj();
  const double *mu0 = nullptr;
  if (st.use_previous) {
    // the caller's mu_induced (PS.cpp:376-386 reads atom->mu_induced).  Between two polar_set_atoms calls the atoms keep
    // their places and the caller's array is what the last compute call wrote into it: the resident copy is the same
    // numbers (a caller that edits mu_induced in between goes through polar_upload_mu or polar_set_atoms)
    if (h->attempt > 0) {
      // a retry after an outgrown pitch starts from the SAME guess as the first attempt: the failed attempt has overwritten
      // d_mu (k_unpack) and, through the early download, the caller's array
      if (h->mu0_saved) { HIPCHECK(hipMemcpyAsync(h->d_mu.p, h->d_mu0.p, 3 * (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, s)); mu0 = h->d_mu.p; }
    } else {
      if (mu_host && !(h->mu_resident && h->mu_host_in_sync)) {
        HIPCHECK(hipMemcpyAsync(h->d_mu.p, mu_host, 3 * (size_t)n * sizeof(double), hipMemcpyHostToDevice, s));
        mu0 = h->d_mu.p;
      } else if (h->mu_resident) mu0 = h->d_mu.p;
      h->mu0_saved = mu0 != nullptr;
      if (mu0) { h->d_mu0.ensure(3 * (size_t)n + 3); HIPCHECK(hipMemcpyAsync(h->d_mu0.p, h->d_mu.p, 3 * (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, s)); }
    }
  }
  h->sorted = false;
  h->dense_gs = false;
  const bool gs_mode = (st.polar_gs || st.polar_gs_ranked) && !st.zodid;
  bool ranked_done = false;
  if (ap && gs_mode && n > 64 && 48.0 * (double)n * (double)n <= 3.2e10 && !getenv("POLAR_NO_DENSE_GS")) {   // (n <= 64: one block, the matrix-free form; 32 GB of tensor = 25,819 atoms)
    // exact-order Gauss-Seidel on the HBM-resident tensor: put the atoms in SWEEP order first
    // (s space = ranked order), so blocks of the sweep are contiguous rows/columns of T6
    h->dense_gs = true;
    if (st.polar_gs_ranked) {
      launch_rank<true>(h, 1); launch_rank<true>(h, 2);  // a2 (orig space: needs only x/alpha/mol)
      ranked_done = true;
      // stable descending sort == the reference's bubble sort (PS.cpp:1130-1143), on the device: no host round trip
      h->d_perm.ensure(n + 1); h->d_inv.ensure(n + 1);
      k_rank_order<<<nblk(n, POLAR_ROWS_PER_BLOCK), POLAR_BLOCK, 0, s>>>(n, h->d_rank.p, h->d_perm.p, h->d_inv.p);
      h->sorted = true;
    }
  }
  if (!ap) { build_cells(h); h->d_pos4.ensure(n + 1); if (h->static_xq) h->d_xq_s.ensure(n + 2); }  // cell order: perm / inv
  k_pack<<<nblk(n + 1, 256), 256, 0, s>>>(n, h->sorted ? h->d_perm.p : nullptr, h->d_x.p, h->d_q.p, h->d_alpha.p, h->d_mol.p, mu0,
                                      h->d_rec0.p, h->d_rec1.p, h->d_mol_s.p, ap ? nullptr : h->d_pos4.p, (!ap && h->static_xq) ? h->d_xq_s.p : nullptr,
                                      (!ap && h->sweep_kernel == 4) ? 1 : 0, h->box, h->boxlo[0], h->boxlo[1], h->boxlo[2]);
  if (!ap) {
    if (h->colors_valid && h->sweep_kernel < 3) map_color_rows(h);  // the colour rows in this step's cell order
    if (h->sweep_kernel == 2) compute_slots(h);
    build_lists(h);
    if (h->colors_valid && h->sweep_kernel == 3) build_cluster_lists(h);
    if (h->sweep_kernel == 4) build_tiles(h);
  }
  HIPCHECK(hipEventRecord(h->ev[1], s));

  // a2: every step in exact mode (reference); in cutoff mode only when the colour phases are rebuilt
  if (st.polar_gs_ranked && (ap || (!h->colors_valid && h->sweep_kernel != 4)) && !ranked_done) {
    if (ap) { launch_rank<true>(h, 1); launch_rank<true>(h, 2); }
    else    { launch_rank<false>(h, 1); launch_rank<false>(h, 2); }
  }
  HIPCHECK(hipEventRecord(h->ev[2], s));

  HIPCHECK(hipEventRecord(h->ev[3], s));

  {  // a4 + a5
    dim3 grid(nblk(own_n(h), POLAR_ROWS_PER_BLOCK)), block(POLAR_BLOCK);
    const double ccs = st.cut_coul * st.cut_coul, e2s = std::sqrt(h->P.qqrd2e);
    if (ap) k_static_field<true><<<grid, block, 0, s>>>(nullptr, n, n, h->d_rec0.p, h->d_mol_s.p, h->box, RowList{nullptr, 0}, nullptr, ccs, e2s, st.polar_gamma, st.use_previous, h->d_ef_s.p, h->d_rec0.p, h->d_rec1.p, nullptr);
    else    k_static_field<false><<<grid, block, 0, s>>>(own_rows(h), own_n(h), n, h->d_rec0.p, h->d_mol_s.p, h->box, RowList{h->d_nl_cnt.p, h->nl_pitch}, h->d_nl_j.p, ccs, e2s, st.polar_gamma, st.use_previous, h->d_ef_s.p, h->d_rec0.p, h->d_rec1.p, h->static_xq ? h->d_xq_s.p : nullptr);
  }
  // tile sweep: the solve works on 48-byte sweep records {position, dipole}; the initial dipoles are in the AtomRecs now
#ifdef POLAR_LAB
  if (!ap && h->sweep_kernel == 4 && !st.zodid)
    k_srec_pack<<<nblk(n, 256), 256, 0, s>>>(n, h->d_rec0.p, h->d_srec0.p, h->d_srec1.p);
#endif
  HIPCHECK(hipEventRecord(h->ev[4], s));
  if (lj_late) launch_lj();
}

// PS.cpp:406-645: everything after the solve
int phase_finish(polar_handle *h, polar_result *out) {
  const polar_settings &st = h->ph.st;
  const int n = h->nlocal, nall = h->nlocal + h->nghost;
  const bool ap = !(st.dd_cutoff > 0.0);
  const bool expd = st.damping_type == POLAR_DAMP_EXPONENTIAL;
  const int eflag = h->step_eflag, vmode = h->step_vflag % 4;
  hipStream_t s = h->stream;
#ifdef POLAR_LAB
  if (!ap && h->sweep_kernel == 4 && !st.zodid)  // the solved dipoles back into the AtomRecs the remaining kernels read
    k_srec_unpack<<<nblk(n, 256), 256, 0, s>>>(n, h->d_scal.p, h->d_srec0.p, h->d_srec1.p, h->d_rec0.p, h->d_rec1.p);
#endif
  k_fallback<<<nblk(n, 256), 256, 0, s>>>(n, h->d_scal.p, h->d_rec0.p, h->d_rec1.p, h->d_ef_s.p);
  HIPCHECK(hipEventRecord(h->ev[5], s));
  // the dipoles and the static field are final: back into the caller's atom order now, so that polar_compute can send them
  // to the host while the force kernel runs (nothing below writes the records)
  k_unpack<<<nblk(n, 256), 256, 0, s>>>(n, h->sorted ? h->d_perm.p : nullptr, h->d_scal.p, h->d_rec0.p, h->d_rec1.p, h->d_ef_s.p, h->d_mu.p, h->d_ef.p);
  if (h->early_mu && n > 0) {
    HIPCHECK(hipEventRecord(h->ev_mu_ready, s));
    HIPCHECK(hipStreamWaitEvent(h->dl_stream, h->ev_mu_ready, 0));
    HIPCHECK(hipMemcpyAsync(h->early_mu, h->d_mu.p, 3 * (size_t)n * sizeof(double), hipMemcpyDeviceToHost, h->dl_stream));
    if (h->early_ef) HIPCHECK(hipMemcpyAsync(h->early_ef, h->d_ef.p, 3 * (size_t)n * sizeof(double), hipMemcpyDeviceToHost, h->dl_stream));
    HIPCHECK(hipEventRecord(h->ev_dl1, h->dl_stream));
  }
  double *vatom = (h->step_vflag / 4) ? h->d_vatom.p : nullptr;
  // join a3: before the force kernel when both write the (non-atomic) per-atom virial rows,
  // otherwise only before the accumulators are read
  auto join_lj = [&]() { if (h->lj_forked) { HIPCHECK(hipStreamWaitEvent(s, h->ev_join, 0)); h->lj_forked = false; } };
  if (vatom) join_lj();
  // full-list mode with an fdotr virial: the polarization forces go to their own array first, so that
  // sum f_pol . x can be formed without the LJ forces (whose virial was tallied pairwise)
  const bool split_f = h->full_list && vmode == 2;
  double *fdst = h->d_f.p;
  if (split_f) {
    h->d_fpol.ensure(3 * (size_t)n + 3);
    HIPCHECK(hipMemsetAsync(h->d_fpol.p, 0, 3 * (size_t)n * sizeof(double), s));
    fdst = h->d_fpol.p;
  }
  if (ap) { if (expd) launch_force<true, 0>(h, eflag, vmode == 1, vatom, fdst); else launch_force<true, 1>(h, eflag, vmode == 1, vatom, fdst); }
  else    { if (expd) launch_force<false, 0>(h, eflag, vmode == 1, vatom, fdst); else launch_force<false, 1>(h, eflag, vmode == 1, vatom, fdst); }
  join_lj();
  if (split_f) {
    k_virial_fdotr<<<std::min(1024, nblk(n, 256)), 256, 0, s>>>(n, h->d_x.p, h->d_fpol.p, h->d_slots.p);
    k_add_into<<<nblk(3 * (long long)n, 256), 256, 0, s>>>(3 * (long long)n, h->d_fpol.p, h->d_f.p);
  } else if (vmode == 2) k_virial_fdotr<<<std::min(1024, nblk(nall, 256)), 256, 0, s>>>(nall, h->d_x.p, h->d_f.p, h->d_slots.p);  // a10
  k_fold_scal<<<1, POLAR_NSLOT, 0, s>>>(h->d_scal.p, h->d_slots.p, 0);
  HIPCHECK(hipEventRecord(h->ev[6], s));
  if (h->early_mu && h->user_mu && n > 0) {  // the force kernel is still running: the dipoles and the field, on the host by now, go into the caller's arrays meanwhile
    HIPCHECK(hipEventSynchronize(h->ev_dl1));
    const double *sm = h->early_mu, *se = h->early_ef;
    double *um = h->user_mu, *ue = h->user_ef;
    host_chunks(3 * (size_t)n, [&](size_t a, size_t b) {
      memcpy(um + a, sm + a, (b - a) * sizeof(double));
      if (ue && se) memcpy(ue + a, se + a, (b - a) * sizeof(double));
    });
  }
  read_scal(h);
  h->mu_resident = true;
  h->mu_host_in_sync = false;  // d_mu has new numbers; polar_compute / polar_compute_peratom set this again once the caller's array holds them

  const Scal &sc = *h->h_scal;
  out->eng_vdwl = sc.eng_vdwl; out->eng_coul = sc.eng_coul;
  out->u_self = sc.u_self; out->u_ef = sc.u_ef; out->u_dd = sc.u_dd;
  out->eng_pol = sc.u_self + sc.u_ef + sc.u_dd;  // PS.cpp:632 (all zero when eflag == 0)
  for (int k = 0; k < 6; k++) out->virial[k] = sc.virial[k];
  long long rb = (long long)sc.rmin_bits;
  memcpy(&out->rmin, &rb, sizeof(double));
  out->rms_dmu = std::sqrt(std::max(0.0, sc.last_change));
  h->ntrace = std::min(h->ntrace, sc.sweeps);
  out->iterations = sc.iterations; out->sweeps = sc.sweeps; out->status = sc.status ? POLAR_WARN_NOT_CONVERGED : POLAR_OK;
  h->last_sweeps = (!st.fixed_iteration && !st.zodid && !sc.status && sc.done) ? sc.sweeps : 0;   // (where the next solve's first look at the loop state goes)
  if (!ap) {
    unsigned long long tot = 0;
    for (int k = 0; k < 64; k++) tot += h->h_ddtot[16 * k];
    h->dd_pairs = (long long)tot;
  }
  out->dd_pairs = ap ? (long long)n * (n - 1) : h->dd_pairs;
  out->ms_color_host = h->ms_color_host;
  h->ms_color_host = 0.0;
  float ms;
  auto el = [&](int a, int b) { HIPCHECK(hipEventElapsedTime(&ms, h->ev[a], h->ev[b])); return (double)ms; };
  out->ms_list = el(0, 1); out->ms_rank = el(1, 2); out->ms_static = el(3, 4);
  HIPCHECK(hipEventElapsedTime(&ms, h->ev_lj0, h->ev_lj1));
  out->ms_ljcoul = ms;  // on its own stream: overlaps the other phases, so the parts no longer add up to ms_total
  out->ms_solve = el(4, 5); out->ms_force = el(5, 6); out->ms_total = el(0, 6);
  if (sc.status) h->warn = "Number of iterations exceeding max_iterations, setting dipoles to alpha*E";  // PS.cpp:1233
  return out->status;
}

// end-of-step flags of the pitched lists (pinned, copied at the end of the list builds): true = a list did not fit, the
// pitches have been enlarged and the step must be repeated
void clear_flags(polar_handle *h) { for (int k = 0; k < 16; k++) h->h_flags[k] = 0; }
void tile_fallback(polar_handle *h) {
  h->sweep_kernel = 2;
  h->colors_valid = false; h->slots_by_color = false;
  h->nl_pitch = h->dd_pitch = 0;  // the cell grid changes with the sweep kernel
  h->warn = "tile sweep unavailable for this system (cell neighbourhood beyond the LDS, or a cell beyond the builder's limits): row sweep in use";
}
bool grow_pitches(polar_handle *h) {
  bool again = false;
  if (h->h_flags[0] != 0) {
    const long long need = ((long long)(1.25 * h->h_flags[0]) / 64 + 1) * 64;
    h->nl_pitch = h->dd_pitch = std::max(need, h->nl_pitch + 64);
    again = true;
  }
  if (h->h_flags[4] != 0) { h->cl_pitch = (((long long)(1.25 * h->h_flags[4]) + 255) / 256) * 256; again = true; }
  if (h->sweep_kernel == 4) {
    if (h->h_flags[7] != 0) { tile_fallback(h); return true; }
    if (h->h_flags[5] != 0) {  // a union list beyond the pitch, or beyond what the sweep's LDS request holds
      const int need = (int)(((long long)(1.15 * h->h_flags[5]) + 16 + 63) / 64 * 64);
      if (need > tile_lds_cap()) { tile_fallback(h); return true; }
      h->un_pitch = std::max(h->un_pitch, need); h->un_lds = h->un_pitch;
      again = true;
    }
    if (h->h_flags[6] != 0) { h->pitch16 = ((long long)(1.25 * h->h_flags[6]) + 511) / 512 * 512; again = true; }
    if (!again && h->h_flags[9] > 0) {  // next step: ask only for the LDS the unions need (two workgroups per CU below 80 KB)
      h->tile_max_u = h->h_flags[9];
      h->un_lds = std::min(h->un_pitch, (int)((long long)(1.06 * (h->tile_max_u + 1)) + 8 + 7) / 8 * 8);
      if (getenv("POLAR_DEBUG") && !h->tile_reported) {
        h->tile_reported = true;
        fprintf(stderr, "[polar] tile sweep: %lld cells (%d x %d x %d), %d launches per sweep, largest union %d records, union pitch %d, "
                "LDS request %zu bytes (%d records), row pitch %lld entries\n", h->ncell, h->grid.nc[0], h->grid.nc[1], h->grid.nc[2],
                (int)h->tile_launches.size(), h->tile_max_u, h->un_pitch, tile_lds_bytes(std::min(h->un_pitch, h->un_lds)), h->un_lds, h->pitch16);
      }
    }
  }
  return again;
}

int do_compute(polar_handle *h, int eflag, int vflag, const double *mu_host, polar_result *out) {
  memset(out, 0, sizeof(*out));
  const bool ap = !(h->ph.st.dd_cutoff > 0.0);
  int rc = 0;
  bool done = false;
  struct Attempt { polar_handle *h; ~Attempt() { h->attempt = 0; } } guard{h};
  for (int attempt = 0; attempt < 5; attempt++) {
    clear_flags(h);
    h->attempt = attempt;
    try {
      phase_begin(h, eflag, vflag, mu_host);
    } catch (const TileUnavailable &) {  // density beyond what a workgroup can stage: the row sweep takes over
      tile_fallback(h);
      continue;
    }
    if (!h->ph.st.zodid) solve(h, ap, out);  // a6 + a7 (PS.cpp:389)
    const int nc = out->ncolors;
    rc = phase_finish(h, out);
    out->ncolors = nc;
    if (ap || !grow_pitches(h)) { done = true; break; }
    memset(out, 0, sizeof(*out));  // a row did not fit its pitch: the pitches have grown, redo the step
  }
  if (!done) throw std::runtime_error("neighbor list pitch overflow persists");
  return rc;
}

// polar_step_begin in two halves, so that a multi-GPU driver can agree on failures and on a common colouring in between:
// step_begin_lists = everything up to the point where the colour phases are needed (PS.cpp:125-386: lists, a3 forked, static
// field, initial guess; on reneighbor steps the verdict on the colouring in use); step_begin_finish = the colouring if it is
// still missing (this handle's own), rows into launch order, descriptors.
void step_begin_lists(polar_handle *h, int eflag, int vflag) {
  HIPCHECK(hipSetDevice(h->device));
  clear_flags(h);
  try {
    phase_begin(h, eflag, vflag, nullptr);
  } catch (const TileUnavailable &) {
    tile_fallback(h);
    phase_begin(h, eflag, vflag, nullptr);
  }
  if (h->ph.st.dd_cutoff > 0.0 && !tile_mode(h)) resolve_colors(h);
}
bool step_needs_colors(const polar_handle *h) {
  const polar_settings &st = h->ph.st;
  const bool clm = st.dd_cutoff > 0.0 && h->sweep_kernel == 3;
  return st.dd_cutoff > 0.0 && !tile_mode(h) && !st.zodid && (st.polar_gs || st.polar_gs_ranked || clm) && !h->colors_valid;
}
void step_begin_finish(polar_handle *h) {
  const polar_settings &st = h->ph.st;
  const bool clm = st.dd_cutoff > 0.0 && h->sweep_kernel == 3;
  if (step_needs_colors(h)) { ensure_colors(h); if (clm) build_cluster_lists(h); else map_color_rows(h); }
  if (!st.zodid && st.dd_cutoff > 0.0 && (st.polar_gs || st.polar_gs_ranked) && h->sweep_kernel == 2 && !slots_current(h)) { compute_slots(h); build_lists(h); }
  if (!st.zodid && st.dd_cutoff > 0.0 && h->sweep_kernel == 2) prepare_lp(h);
  h->in_step = true;
}
