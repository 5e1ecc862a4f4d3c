// polar_color.hip -- colour phases of the list-mode Gauss-Seidel: atoms of one colour are relaxed by one launch and must lie
// farther apart than the colour distance.  Built on the device (sequential DSATUR cell by cell + local repair), re-validated
// on reneighbor steps.  No reference counterpart: the reference's sweep is serial (PS.cpp:1158-1180).
#include "polar_handle.hpp"

#ifdef POLAR_LAB
#include "lab/color_clusters_host.inc"
#else
inline void build_cluster_colors(polar_handle *, const std::vector<double> &, const std::vector<std::vector<int>> &, const std::vector<std::vector<int>> &) { throw std::logic_error("lab build only"); }
#endif  // POLAR_LAB

#ifdef POLAR_LAB
#include "lab/color_host_dsatur.inc"
#else
inline void build_colors(polar_handle *, const std::vector<double> &) { throw std::logic_error("host-side colouring: lab build only"); }
#endif  // POLAR_LAB

// ---- the colour phases on the device (polar_lists.hpp, k_color_*): sequential DSATUR cell by cell (parity classes of the
//      cell grid), the small top class repaired by local exhaustive search, Jones-Plassmann rounds as the fallback; phase
//      order and the rows of every phase in cell order.
//      Needs this step's cell order (phase_begin has run) and, for the ranked flavour, the rank metric in d_rank (s space).
namespace {
struct Lap {   // POLAR_DEBUG: wall time since the last lap, the device drained first
  hipStream_t s; bool on; std::chrono::steady_clock::time_point t;
  explicit Lap(hipStream_t st) : s(st), on(getenv("POLAR_DEBUG") != nullptr), t(std::chrono::steady_clock::now()) {}
  void operator()(const char *what) {
    if (!on) return;
    HIPCHECK(hipStreamSynchronize(s));
    const auto tn = std::chrono::steady_clock::now();
    fprintf(stderr, "[polar] colouring: %-28s %7.3f ms\n", what, std::chrono::duration<double, std::milli>(tn - t).count());
    t = tn;
  }
};
// rows and rank sums per colour of the OWN rows -> h_cstat (pinned), number of colours in use on this handle
int color_stats(polar_handle *h, bool ranked) {
  const int n = h->nlocal;
  hipStream_t s = h->stream;
  HIPCHECK(hipMemsetAsync(h->d_cstat.p, 0, 128 * sizeof(double), s));
  k_color_stats<<<nblk(n, 256), 256, 0, s>>>(n, h->d_color_s.p, ranked ? h->d_rank.p : nullptr, h->d_cstat.p, h->d_perm.p, own_lo(h), own_lo(h) + own_n(h));
  HIPCHECK(hipMemcpyAsync(h->h_cstat, h->d_cstat.p, 128 * sizeof(double), hipMemcpyDeviceToHost, s));
  HIPCHECK(hipStreamSynchronize(s));
  int nc = 0;
  for (int c = 0; c < 64; c++) if (h->h_cstat[2 * c] > 0.0) nc = c + 1;
  return nc;
}
// stage 1: buffers and the conflict lists.  color_s = -1 everywhere afterwards.
void color_adjacency(polar_handle *h, bool with_halo) {
  const int n = h->nlocal;
  hipStream_t s = h->stream;
  Lap lap(s);
  lap("(work queued before)");
  if (!h->h_cflags) {
    HIPCHECK(hipHostMalloc((void **)&h->h_cflags, 96 * sizeof(int)));
    HIPCHECK(hipHostMalloc((void **)&h->h_cstat, 128 * sizeof(double)));
    HIPCHECK(hipHostMalloc((void **)&h->h_coff, POLAR_MAX_CLASS_OFF * sizeof(long long)));
  }
  h->d_cdeg.ensure(n + 1); h->d_cprio.ensure(n + 1);
  h->d_color_s.ensure(n + 1); h->d_color_orig.ensure(n + 1); h->d_cflags.ensure(96); h->d_cstat.ensure(128); h->d_crelabel.ensure(64);
  int *flags = h->d_cflags.p;  // [0] conflict-list overflow, [1] atoms deferred in the last round, [2] a row that saw 64 colours, [65..] atoms that could not leave a folded class
  HIPCHECK(hipMemsetAsync(h->d_color_orig.p, 0xFF, (size_t)(n + 1) * sizeof(int), s));
  const int lo = own_lo(h), hi = own_lo(h) + own_n(h);
  const double dc2 = h->color_dist * h->color_dist;
  int sw = 1;
  {  // k_color_adj looks at the cells within `sw` of an atom's cell: the colour distance must fit (it used to look at
     // 3 x 3 x 3 cells whatever the distance -- with the small cells of a short cutoff the lists then missed pairs)
    double wdt[3];
    box_widths(h->box, wdt);
    const double wmin = std::min({wdt[0] / h->grid.nc[0], wdt[1] / h->grid.nc[1], wdt[2] / h->grid.nc[2]});
    if (h->color_dist > wmin) sw = 2;
    if (h->color_dist > 2.0 * wmin) throw InputError("colour distance (POLAR_COLOR_DIST) beyond two cell widths of the list grid (a cell is half the list cutoff wide): the conflict lists would miss pairs");
  }
  for (int attempt = 0;; attempt++) {  // conflict lists; an atom with more neighbours than the lists hold makes them wider
    h->d_cadj.ensure((size_t)n * h->cadj_pitch + 16);
    HIPCHECK(hipMemsetAsync(flags, 0, 96 * sizeof(int), s));
    k_color_adj<<<nblk(n, 128), 128, 0, s>>>(n, h->d_pos4.p, h->d_perm.p, lo, hi, h->box, h->grid, h->d_cell_first.p, h->d_cell_fill.p, dc2,
                                             h->cadj_pitch, h->d_cadj.p, h->d_cdeg.p, h->d_cprio.p, h->d_color_s.p, flags, with_halo ? 1 : 0, sw);
    HIPCHECK(hipMemcpyAsync(h->h_cflags, flags, sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHECK(hipStreamSynchronize(s));
    if (h->h_cflags[0] <= h->cadj_pitch) break;
    if (h->h_cflags[0] > 62 || attempt > 3) throw std::runtime_error("colouring: more than 62 polarizable atoms within the colour distance of one atom (64 colours at most): reduce POLAR_COLOR_DIST");
    h->cadj_pitch = (h->h_cflags[0] + 4 + 7) / 8 * 8;
  }
  lap("conflict lists");
}
// stage 2: colour the own rows (colours of other ranks' rows, where known, are fixed constraints).  Returns the number of
// colours in use among the own rows.
int color_assign(polar_handle *h, bool ranked) {
  const int n = h->nlocal;
  hipStream_t s = h->stream;
  Lap lap(s);
  int *flags = h->d_cflags.p;
  const int ap_ = h->cadj_pitch;
  // sequential DSATUR cell by cell (k_color_cells): one launch per parity class of the cell grid -- per dimension the even
  // cells, the odd cells and, when a periodic dimension has an odd count, its last cell on its own.  Safe while two cells of
  // a class (a whole cell apart) cannot hold neighbours: colour distance below the shortest cell edge.
  const double min_edge = std::min({h->box.prd[0] / h->grid.nc[0], h->box.prd[1] / h->grid.nc[1], h->box.prd[2] / h->grid.nc[2]});
  // (tilted box: a cell's perpendicular width is below its edge along the lattice vector -- by at most 1/sqrt(1.5) at LAMMPS'
  //  tilt limit of half a box length)
  bool cell_pass = h->color_dist < (h->box.triclinic ? 0.5 : 1.0) * min_edge;
#ifdef POLAR_LAB
  if (getenv("POLAR_COLOR_JP")) cell_pass = false;  // lab: Jones-Plassmann alone (the round-3 first version: 5 classes)
#endif
  if (cell_pass) {
    struct Cls { int start, stride, count; };
    std::vector<Cls> cls[3];
    for (int k = 0; k < 3; k++) {
      const int nc = h->grid.nc[k];
      const bool seam = h->box.periodic[k] && (nc & 1) && nc > 1;
      const int lim = seam ? nc - 1 : nc;
      if ((lim + 1) / 2 > 0) cls[k].push_back(Cls{0, 2, (lim + 1) / 2});
      if (lim / 2 > 0) cls[k].push_back(Cls{1, 2, lim / 2});
      if (seam) cls[k].push_back(Cls{nc - 1, 1, 1});
    }
    for (const Cls &cz : cls[2]) for (const Cls &cy : cls[1]) for (const Cls &cx : cls[0])
      k_color_cells<<<cx.count * cy.count * cz.count, 64, 0, s>>>(cx.start, cy.start, cz.start, cx.stride, cy.stride, cz.stride, cx.count, cy.count,
                                                                  cz.count, h->grid.nc[0], h->grid.nc[1], h->d_cell_first.p, h->d_cell_fill.p, ap_,
                                                                  h->d_cadj.p, h->d_cdeg.p, h->d_color_s.p, flags + 2);
  }
  // Jones-Plassmann rounds for whatever is still uncoloured (nothing after the cell pass; everything without it): an uncoloured
  // row whose priority beats every uncoloured neighbour's takes the lowest free colour.  First look after one round.
  bool coloured = false;
  int rounds = 0;
  const int max_rounds = 4096, look_every = 32;
  while (!coloured && rounds < max_rounds) {   // rounds (priorities, then decisions), then a look at how many atoms the last one deferred
    const int hashed = rounds >= 1024 ? 1 : 0;  // index-ordered ties while the chains stay short (see k_color_prio)
    const int batch = rounds == 0 ? 1 : look_every;
    for (int k = 0; k < batch; k++) {
      k_color_prio<<<nblk(n, 256), 256, 0, s>>>(n, ap_, h->d_cadj.p, h->d_cdeg.p, h->d_color_s.p, h->d_perm.p, h->d_cprio.p, hashed);
      if (k == batch - 1) HIPCHECK(hipMemsetAsync(flags + 1, 0, sizeof(int), s));
      k_color_round<<<nblk(n, 256), 256, 0, s>>>(n, ap_, h->d_cadj.p, h->d_cdeg.p, h->d_cprio.p, h->d_color_s.p, flags + 1);
    }
    rounds += batch;
    HIPCHECK(hipMemcpyAsync(h->h_cflags, flags, 3 * sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHECK(hipStreamSynchronize(s));
    if (h->h_cflags[2] >= 1000) throw std::runtime_error("colouring needs more than 64 colours: reduce POLAR_COLOR_DIST");
    coloured = h->h_cflags[1] == 0;
  }
  if (!coloured) throw std::runtime_error("colouring: Jones-Plassmann did not finish");
  lap("cell pass + rounds");
  if (getenv("POLAR_DEBUG")) fprintf(stderr, "[polar] device colouring: %d rounds\n", rounds);
  auto stats = [&]() { return color_stats(h, ranked); };
  int ncolors = stats();
  auto fold = [&]() {  // dissolve the highest class while that works
    for (int pass = 0; pass < 6 && ncolors > 1; pass++) {
      k_color_fold<<<nblk(n, 256), 256, 0, s>>>(n, ap_, ncolors - 1, h->d_cadj.p, h->d_cdeg.p, h->d_color_s.p, flags + 65 + pass);
      const int before = ncolors;
      ncolors = stats();
      if (ncolors == before) break;
    }
  };
  fold();
  lap("fold");
  // local repair of a small top class (k_color_ball): the cell-by-cell pass leaves about one row per unit cell in a fifth
  // class on the MOF boxes; first balls of one conflict step, then of two, then of three
  const int kcap = 8192;
  // (only a SMALL top class of at most six is worth it -- at most 2 % of the rows: where atoms overlap, e.g. sorbates flying through the
  //  framework in bench.py's ballistic leg, the extra classes are needed and no local search removes them)
  const double krows = std::min((double)kcap, 0.02 * (double)own_n(h));
  for (int hops = 1; hops <= 3 && ncolors > 2 && ncolors <= 6 && h->h_cstat[2 * (ncolors - 1)] <= (hops < 3 ? krows : 128.0); hops++) {   // (the last, long search only for a handful of rows)
    h->d_klist.ensure(4 * (size_t)kcap + 8);
    int *raw = h->d_klist.p, *list = raw + kcap, *st0 = raw + 2 * kcap, *st1 = raw + 3 * kcap, *cnt = raw + 4 * kcap;
    HIPCHECK(hipMemsetAsync(cnt, 0, sizeof(int), s));
    k_color_collect<<<nblk(n, 256), 256, 0, s>>>(n, ncolors - 1, h->d_color_s.p, kcap, raw, cnt);
    k_sort_small<<<8, 256, 0, s>>>(cnt, kcap, raw, list, st0, st1);
    const int waves = (int)h->h_cstat[2 * (ncolors - 1)];
    const double reach = (2 * hops + 1) * h->color_dist;
    const int budget = hops == 1 ? 768 : hops == 2 ? 2048 : 8192;   // search steps per ball: the later stages see few rows
    for (int round = 0; round < 2 + 2 * hops; round++)
      k_color_ball<<<waves, 64, 0, s>>>(cnt, kcap, list, (round & 1) ? st1 : st0, (round & 1) ? st0 : st1, h->d_pos4.p, h->box, reach * reach,
                                        ncolors - 1, hops, budget, ap_, h->d_cadj.p, h->d_cdeg.p, h->d_color_s.p);
    const int before = ncolors;
    ncolors = stats();
    if (getenv("POLAR_DEBUG")) fprintf(stderr, "[polar] colour repair (%d-step balls): %d rows in the top class, %d classes -> %d\n", hops, waves, before, ncolors);
    if (ncolors < before) break;
  }
  lap("repair");
  // iterated greedy (Culberson) for what is still above four classes -- overlapping atoms, or the Jones-Plassmann fallback,
  // whose parallel rounds decide on stale saturation counts: regrouping by old classes in another order never adds a class
  // and sometimes removes one
  for (int ig = 0, stale = 0; ig < 2 && stale < 2 && ncolors > 4; ig++) {
    std::vector<int> ord((size_t)ncolors), rank(64, 0);
    std::iota(ord.begin(), ord.end(), 0);
    if (ig % 2 == 0) std::reverse(ord.begin(), ord.end());                                    // highest class first
    else std::stable_sort(ord.begin(), ord.end(), [&](int a, int b) {                         // smallest / largest class first
      return (ig % 4 == 1) ? h->h_cstat[2 * a] < h->h_cstat[2 * b] : h->h_cstat[2 * a] > h->h_cstat[2 * b]; });
    for (int c = 0; c < ncolors; c++) rank[ord[c]] = c;
    HIPCHECK(hipMemcpyAsync(h->d_crelabel.p, rank.data(), 64 * sizeof(int), hipMemcpyHostToDevice, s));
    k_color_regroup<<<nblk(n, 256), 256, 0, s>>>(n, h->d_cdeg.p, h->d_crelabel.p, h->d_perm.p, h->d_color_s.p, h->d_cprio.p);
    for (int k = 0; k < ncolors + 1; k++)
      k_color_round<<<nblk(n, 256), 256, 0, s>>>(n, ap_, h->d_cadj.p, h->d_cdeg.p, h->d_cprio.p, h->d_color_s.p, flags + 80);
    const int before = ncolors;
    ncolors = stats();   // (synchronises: `rank` may go)
    fold();
    stale = ncolors < before ? 0 : stale + 1;
  }
  lap("iterated greedy");
  return ncolors;
}
// stage 3: phase order from h_cstat (rows and rank sums per colour -- of this handle, or summed over the ranks by the
// caller), relabel, and the rows of every phase in cell order (boundary rows first when the handle has boundary flags)
void color_finish(polar_handle *h, bool ranked, int ncolors) {
  const int n = h->nlocal;
  hipStream_t s = h->stream;
  Lap lap(s);
  const long long ncell = h->ncell;
  if (ncolors > 64) throw std::runtime_error("colouring needs more than 64 colours: reduce POLAR_COLOR_DIST");
  // phase order: "ranked" flavour = colours by descending mean rank metric (PS.cpp:192-227 ranks the dipoles most likely to
  // change first); otherwise by descending size
  std::vector<int> ord((size_t)ncolors), relabel(64, 0);
  std::iota(ord.begin(), ord.end(), 0);
  auto key = [&](int c) { return ranked ? h->h_cstat[2 * c + 1] / std::max(h->h_cstat[2 * c], 1.0) : h->h_cstat[2 * c]; };
  std::stable_sort(ord.begin(), ord.end(), [&](int a, int b) { return key(a) > key(b); });
  for (int c = 0; c < ncolors; c++) relabel[ord[c]] = c;
  HIPCHECK(hipMemcpyAsync(h->d_crelabel.p, relabel.data(), 64 * sizeof(int), hipMemcpyHostToDevice, s));
  k_color_relabel<<<nblk(n, 256), 256, 0, s>>>(n, h->d_crelabel.p, h->d_perm.p, h->d_color_s.p, h->d_color_orig.p);
  // rows of every phase in cell order, by sub-class inside a phase: multi-GPU: boundary rows first (polar_dist_set_halo)
  const int *sub = nullptr;
  int nsub = 1;
  if (h->bflag_n == n && sharded(h) && !deterministic(h)) { sub = h->d_bflag.p; nsub = 2; }
  const int nclass = nsub * ncolors;
  if (nclass + 1 > POLAR_MAX_CLASS_OFF) throw std::runtime_error("colouring: too many phase classes");   // (cannot happen: ncolors <= 64, nsub <= 2)
  const size_t ncc = (size_t)nclass * ncell;
  const int lo = own_lo(h), hi = own_lo(h) + own_n(h);
  h->d_ccnt.ensure(ncc + 1); h->d_coff.ensure(ncc + 2);
  k_color_cellcount<<<nblk(ncell, 128), 128, 0, s>>>(ncell, nclass, h->d_cell_first.p, h->d_cell_fill.p, h->d_color_s.p, h->d_ccnt.p, h->d_perm.p, lo, hi, sub, nsub);
  launch_scan((long long)ncc, h->d_ccnt.p, h->d_coff.p, h->d_scan_a, s);
  for (int q = 0; q <= nclass; q++)
    HIPCHECK(hipMemcpyAsync(h->h_coff + q, h->d_coff.p + (size_t)q * ncell, sizeof(long long), hipMemcpyDeviceToHost, s));
  HIPCHECK(hipStreamSynchronize(s));  // (also: `relabel` is a stack vector)
  h->color_off.assign((size_t)ncolors + 1, 0);
  h->color_sub.assign((size_t)nclass + 1, 0);
  h->color_nsub = nsub;
  for (int q = 0; q <= nclass; q++) h->color_sub[q] = (int)h->h_coff[q];
  for (int q = 0; q <= ncolors; q++) h->color_off[q] = (int)h->h_coff[nsub * q];
  const int tot = h->color_off[ncolors];
  h->d_rows_orig.ensure((size_t)tot + 1); h->d_rows.ensure((size_t)tot + 1);
  k_color_fill<<<nblk(ncell, 128), 128, 0, s>>>(ncell, nclass, h->d_cell_first.p, h->d_cell_fill.p, h->d_color_s.p, h->d_perm.p, h->d_coff.p,
                                               h->d_rows_orig.p, lo, hi, sub, nsub);
  lap("phase order + rows");
#ifdef POLAR_LAB
#include "lab/color_sort_by_trips.inc"
#endif
  h->h_color.assign((size_t)n, 0);  // (its size says "a colouring for n atoms exists": the colours themselves live on the device)
  h->color_epoch++;
  h->colors_rebuilt++;
  if (getenv("POLAR_DEBUG")) {
    fprintf(stderr, "[polar] %d colour phases (device, dist %.2f):", ncolors, h->color_dist);
    for (int c = 0; c < ncolors; c++) fprintf(stderr, " %d", h->color_off[c + 1] - h->color_off[c]);
    fprintf(stderr, "\n");
  }
  h->colors_valid = true;
}
}  // namespace

void build_colors_device(polar_handle *h, bool ranked) {
  color_adjacency(h, false);
  const int ncolors = color_assign(h, ranked);
  color_finish(h, ranked, ncolors);
  h->colors_global = false;
}

// ONE colouring consistent across the ranks of a multi-GPU run (polar_dist): no two rows of one colour closer than the colour
// distance, whichever ranks own them -- so that the colour phases of all ranks together are the single-GPU iteration and a
// per-phase exchange of boundary dipoles loses nothing.  Ranks take turns by class (no two peers share a class, polar_dist
// derives the classes from the peer graph): a rank colours its rows with the sequential DSATUR above against the colours its
// already-coloured neighbours' rows hold in its halo; after every turn the colours of the halo rows travel (cc.exchange).
// Phase order and labels come from the all-reduced rows / rank sums per colour, identical on every rank.
void build_colors_distributed(polar_handle *h, bool ranked, ColorComm &cc) {
  const auto t0 = std::chrono::steady_clock::now();
  if (!(h->sorted && h->ph.st.dd_cutoff > 0.0 && h->sweep_kernel == 2 && h->pol_first)) throw std::logic_error("distributed colouring: needs the row sweep of list mode");
  if (ranked) { launch_rank_pass(h, false, 1); launch_rank_pass(h, false, 2); }   // a2 of the own rows for the phase order
  color_adjacency(h, true);
  for (int cls = 0; cls < cc.nclasses; cls++) {
    if (cls == cc.my_class) (void)color_assign(h, ranked);
    cc.exchange(h->d_color_s.p);     // halo rows take their owners' colours (collective: every rank, every turn)
  }
  (void)color_stats(h, ranked);      // own rows -> h_cstat
  cc.allreduce(h->h_cstat, 128);     // -> rows and rank sums per colour over all ranks
  int ncolors = 0;
  for (int c = 0; c < 64; c++) if (h->h_cstat[2 * c] > 0.0) ncolors = c + 1;
  color_finish(h, ranked, ncolors);
  h->colors_global = true;
  h->ms_color_host = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

// polar_set_colors: a colouring handed in by the caller (own rows and, on a sharded handle, the halo rows): phases in the
// order of the colour numbers, rows of a phase in cell order.  Needs this step's cell order.
void apply_imposed_colors(polar_handle *h) {
  const int n = h->nlocal;
  hipStream_t s = h->stream;
  if ((int)h->user_colors.size() != n) throw std::logic_error("imposed colours: atom count changed");
  if (!h->h_cflags) {
    HIPCHECK(hipHostMalloc((void **)&h->h_cflags, 96 * sizeof(int)));
    HIPCHECK(hipHostMalloc((void **)&h->h_cstat, 128 * sizeof(double)));
    HIPCHECK(hipHostMalloc((void **)&h->h_coff, POLAR_MAX_CLASS_OFF * sizeof(long long)));
  }
  h->d_color_s.ensure(n + 1); h->d_color_orig.ensure(n + 1); h->d_cstat.ensure(128); h->d_crelabel.ensure(64);
  HIPCHECK(hipMemcpyAsync(h->d_color_orig.p, h->user_colors.data(), (size_t)n * sizeof(int), hipMemcpyHostToDevice, s));
  k_color_map<<<nblk(n, 256), 256, 0, s>>>(n, h->d_perm.p, h->d_color_orig.p, h->d_color_s.p);
  int ncolors = 0;
  for (int c : h->user_colors) ncolors = std::max(ncolors, c + 1);
  for (int c = 0; c < 64; c++) { h->h_cstat[2 * c] = (double)(64 - c); h->h_cstat[2 * c + 1] = (double)(64 - c); }   // identity relabel: keep the caller's order
  color_finish(h, false, ncolors);
  h->colors_global = true;
}
// ---- the solve: a6+a7 (PS.cpp:1113-1238) -----------------------------------------------------
void ensure_colors(polar_handle *h) {
  if (h->colors_valid) return;
  const polar_settings &st = h->ph.st;
  const int n = h->nlocal;
  const auto t0 = std::chrono::steady_clock::now();
  const bool on_device = h->sorted && st.dd_cutoff > 0.0 && h->sweep_kernel == 2 && !h->host_colors && h->pol_first;
  if (on_device && !h->user_colors.empty() && (int)h->user_colors.size() == n && !h->user_colors_clashed) {
    apply_imposed_colors(h);
  } else if (on_device) {
    build_colors_device(h, st.polar_gs_ranked != 0);
  } else {  // lab paths (cluster rows, POLAR_HOST_COLORS): the host-side conflict graph + DSATUR of rounds 1-2
    std::vector<double> rk;
    if (st.polar_gs_ranked && !sharded(h)) {  // a sharded handle only knows its own rows' metric
      std::vector<double> rs(n);
      std::vector<int> perm(n);
      rk.assign(n, 0.0);
      HIPCHECK(hipMemcpyAsync(rs.data(), h->d_rank.p, n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
      HIPCHECK(hipMemcpyAsync(perm.data(), h->d_perm.p, n * sizeof(int), hipMemcpyDeviceToHost, h->stream));
      HIPCHECK(hipStreamSynchronize(h->stream));
      for (int k = 0; k < n; k++) rk[perm[k]] = rs[k];  // rank metric was computed in s space
    }
    build_colors(h, rk);
  }
  // (wall time of the rebuild, host work and the waits for the device included)
  h->ms_color_host = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}
// per step: the colour rows (orig ids) -> s space of this step's cell order
void map_color_rows(polar_handle *h) {
  const int tot = h->color_off.empty() ? 0 : h->color_off.back();
  if (tot > 0) k_map_rows<<<nblk(tot, 256), 256, 0, h->stream>>>(tot, h->d_inv.p, h->d_rows_orig.p, h->d_rows.p);
}

// reneighbor steps: k_nl_build has checked the colouring in use against the new positions; rebuild only on a clash
void resolve_colors(polar_handle *h) {
  if (!h->colors_recheck) return;
  h->colors_recheck = false;
  if (!h->colors_valid) return;
  int clash = 0;
  HIPCHECK(hipMemcpyAsync(&clash, h->d_overflow.p + 8, sizeof(int), hipMemcpyDeviceToHost, h->stream));
  HIPCHECK(hipStreamSynchronize(h->stream));
  if (!clash) { h->colors_reused++; return; }
  h->colors_valid = false;
  if (!h->user_colors.empty()) h->user_colors_clashed = true;   // the caller's colouring no longer separates its classes: the library's own takes over
  if (h->ph.st.polar_gs_ranked) { launch_rank_pass(h, false, 1); launch_rank_pass(h, false, 2); }  // a2 for the phase order
}

