/* shim_host_harness.cpp -- TEST INFRASTRUCTURE (build container only; needs /root/reference, never shipped).
 *
 * Runs the HOST-SIDE paths of the LAMMPS shim (lammps_shim/pair_lj_cut_coul_long_polarization_mi355x.cpp) inside the
 * reference's own Pair base class (src/pair.cpp, compiled from where it lies) NEXT TO the reference's pair style, in one
 * process, and compares them: settings / coeff / Pair::init (init_style, init_one, init_tables), cutsq, single(),
 * extract(), and the restart records in both directions (a file written by the reference read by the shim and the other
 * way round; the tagged record of `restart_polar yes`).  No GPU is needed: the library's host mirror works without one.
 * The object graph and the framework stubs are the link seam's (oracle/ref_seam/seam_harness.cpp), included here. */
#define private public
#define protected public
#include "atom.h"   /* (map_array is set up by hand below) */
#include "pair_lj_cut_coul_long_polarization_mi355x.h"
#undef private
#undef protected
#include <algorithm>
#include "../../oracle/ref_seam/seam_harness.cpp"

int LAMMPS_NS::Atom::map_find_hash(int) { return -1; }

namespace {
struct HostCtx {
  LAMMPS *lmp;
  std::vector<std::vector<char> > store;
  std::vector<char *> argv(const char *const *a, int n) {
    std::vector<char *> v;
    for (int k = 0; k < n; k++) { store.emplace_back(a[k], a[k] + strlen(a[k]) + 1); v.push_back(store.back().data()); }
    return v;
  }
};
LAMMPS *make_lammps(int ntypes, double g_ewald, double qqrd2e) {
  LAMMPS *lmp = blank<LAMMPS>();
  lmp->screen = stdout;
  lmp->error = blank<Error>();
  lmp->atom = blank<Atom>();
  lmp->force = blank<Force>();
  lmp->domain = blank<Domain>();
  lmp->neighbor = blank<Neighbor>();
  lmp->update = blank<Update>();
  lmp->comm = blank<Comm>();
  lmp->memory = new Memory(lmp);
  static void *vtab[64];
  for (int k = 0; k < 64; k++) vtab[k] = (void *)&seam_nop;
  *(void ***)lmp->comm = vtab;
  lmp->comm->me = 0; lmp->comm->nprocs = 1; lmp->comm->nthreads = 1;
  KSpace *ks = blank<KSpace>();
  ks->g_ewald = g_ewald;
  lmp->force->kspace = ks;
  lmp->force->qqrd2e = qqrd2e; lmp->force->qqr2e = qqrd2e; lmp->force->dielectric = 1.0;
  lmp->force->newton = lmp->force->newton_pair = lmp->force->newton_bond = 1;
  const double slj[4] = {1.0, 0.0, 0.0, 0.5}, sc[4] = {1.0, 0.0, 0.0, 0.8333333333333333};
  for (int k = 0; k < 4; k++) { lmp->force->special_lj[k] = slj[k]; lmp->force->special_coul[k] = sc[k]; }
  lmp->update->whichflag = 1;
  static char verlet[] = "verlet";
  lmp->update->integrate_style = verlet;
  lmp->domain->dimension = 3;
  lmp->atom->ntypes = ntypes; lmp->atom->q_flag = 1; lmp->atom->static_polarizability_flag = 1; lmp->atom->molecular = 1;
  lmp->atom->nlocal = 0; lmp->atom->nghost = 0;
  return lmp;
}
template <class P>
void feed(HostCtx &H, P *pair, int nstyle, const char *const *style, int nmod, const char *const *mod, int ncoeff,
          const char *const *rows) {
  { std::vector<char *> a = H.argv(style, nstyle); pair->settings((int)a.size(), a.data()); }
  if (nmod) { std::vector<char *> a = H.argv(mod, nmod); pair->modify_params((int)a.size(), a.data()); }
  for (int r = 0; r < ncoeff; r++) {
    std::vector<char> row(rows[r], rows[r] + strlen(rows[r]) + 1);
    std::vector<char *> a;
    for (char *t = strtok(row.data(), " "); t; t = strtok(NULL, " ")) a.push_back(t);
    pair->coeff((int)a.size(), a.data());
  }
}
// largest |difference| of single() (energy and fforce, relative to the larger magnitude) and of cutsq between two pair objects
double g_max_energy = 0.0, g_max_force = 0.0;  // what the comparisons saw (a test of nothing would leave these at zero)
double compare(Pair *a, Pair *b, int ntypes, double rmax) {
  double worst = 0.0;
  const double fc[3] = {1.0, 0.5, 0.0}, fl[3] = {1.0, 0.0, 0.5};
  for (int i = 1; i <= ntypes; i++)
    for (int j = 1; j <= ntypes; j++) {
      worst = std::max(worst, fabs(a->cutsq[i][j] - b->cutsq[i][j]));
      for (double r = 0.8; r < rmax; r += 0.173)
        for (int s = 0; s < 3; s++) {
          double fa = 0, fb = 0;
          const double ea = a->single(0, 1, i, j, r * r, fc[s], fl[s], fa), eb = b->single(0, 1, i, j, r * r, fc[s], fl[s], fb);
          g_max_energy = std::max(g_max_energy, fabs(ea)); g_max_force = std::max(g_max_force, fabs(fa));
          worst = std::max(worst, fabs(ea - eb) / std::max(1e-300, std::max(fabs(ea), fabs(eb))) * (ea == eb ? 0.0 : 1.0));
          worst = std::max(worst, fabs(fa - fb) / std::max(1e-300, std::max(fabs(fa), fabs(fb))) * (fa == fb ? 0.0 : 1.0));
        }
    }
  return worst;
}
}  // namespace

extern "C" {
/* report[0] init/cutsq/single: shim vs reference              report[1] reference-written restart read by the shim
 * report[2] shim-written (default format) restart read by the reference
 * report[3] shim restart_polar yes -> shim: 0 if every polarization keyword came back, else a positive code
 * report[4] extract() mismatches (0 expected)       report[5], report[6] largest |energy| and |fforce| the comparisons saw
 * report[7] 0 if write_data / write_data_all produce the reference's text byte for byte */
int shimhost_check(const char *tmpdir, int ntypes, double g_ewald, double qqrd2e, int nstyle, const char *const *style,
                   int nmod, const char *const *mod, int ncoeff, const char *const *rows, double *report, char *msg, int nmsg) {
  g_last_error.clear();
  for (int k = 0; k < 8; k++) report[k] = -1.0;
  g_max_energy = g_max_force = 0.0;
  HostCtx H;
  H.lmp = make_lammps(ntypes, g_ewald, qqrd2e);
  LAMMPS *lmp = H.lmp;
  /* single() reads atom->q[i], atom->q[j]: two charges */
  lmp->atom->nlocal = 2; lmp->atom->nmax = 2;
  lmp->memory->create(lmp->atom->q, 2, "q");
  lmp->memory->create(lmp->atom->type, 2, "t");
  lmp->atom->q[0] = 0.4; lmp->atom->q[1] = -0.7; lmp->atom->type[0] = 1; lmp->atom->type[1] = 1;
  int rc = 0;
  try {
    PairLJCutCoulLongPolarization *ref = new PairLJCutCoulLongPolarization(lmp);
    PairLJCutCoulLongPolarizationMI355X *shim = new PairLJCutCoulLongPolarizationMI355X(lmp);
    feed(H, ref, nstyle, style, nmod, mod, ncoeff, rows);
    feed(H, shim, nstyle, style, nmod, mod, ncoeff, rows);
    lmp->force->pair = ref;  ref->init();
    lmp->force->pair = shim; shim->init();
    const double rmax = 14.0;
    report[0] = compare(ref, shim, ntypes, rmax);
    /* extract(): the three names of PS.cpp:1101-1109 */
    {
      int da = -1, db = -1, bad = 0;
      const double *ca = (const double *)ref->extract("cut_coul", da), *cb = (const double *)shim->extract("cut_coul", db);
      if (!ca || !cb || da != db || *ca != *cb) bad++;
      double **ea = (double **)ref->extract("epsilon", da), **eb = (double **)shim->extract("epsilon", db);
      double **sa = (double **)ref->extract("sigma", da), **sb = (double **)shim->extract("sigma", db);
      if (!ea || !eb || !sa || !sb || da != db) bad++;
      else
        for (int i = 1; i <= ntypes; i++)
          for (int j = i; j <= ntypes; j++)
            if (ea[i][j] != eb[i][j] || sa[i][j] != sb[i][j]) bad++;
      int dn = 0;
      if (shim->extract("nonsense", dn) != NULL) bad++;
      report[4] = bad;
    }
    {  /* write_data / write_data_all: the "PairIJ Coeffs" sections of a data file (PS.cpp:1013-1031) */
      auto slurp = [](const std::string &p) { std::string t; FILE *f = fopen(p.c_str(), "rb"); int c; while ((c = fgetc(f)) != EOF) t.push_back((char)c); fclose(f); return t; };
      std::string a = std::string(tmpdir) + "/ref.data", b = std::string(tmpdir) + "/shim.data";
      FILE *fa = fopen(a.c_str(), "wb"); ref->write_data(fa); ref->write_data_all(fa); fclose(fa);
      FILE *fb = fopen(b.c_str(), "wb"); shim->write_data(fb); shim->write_data_all(fb); fclose(fb);
      const std::string ta = slurp(a), tb = slurp(b);
      report[7] = (ta == tb && ta.size() > 20) ? 0.0 : 1.0;
    }
    std::string f1 = std::string(tmpdir) + "/ref.restart", f2 = std::string(tmpdir) + "/shim.restart",
                f3 = std::string(tmpdir) + "/shim_polar.restart";
    /* 1: the reference writes, the shim reads */
    { FILE *fp = fopen(f1.c_str(), "wb"); ref->write_restart(fp); int tail = 12345; fwrite(&tail, sizeof(int), 1, fp); fclose(fp); }
    {
      PairLJCutCoulLongPolarizationMI355X *s2 = new PairLJCutCoulLongPolarizationMI355X(lmp);
      FILE *fp = fopen(f1.c_str(), "rb"); s2->read_restart(fp);
      int tail = 0; size_t got = fread(&tail, sizeof(int), 1, fp); fclose(fp);
      if (got != 1 || tail != 12345) throw SeamError{"the shim did not leave the stream behind the reference's record"};
      lmp->force->pair = s2; s2->init();
      report[1] = compare(ref, s2, ntypes, rmax);
    }
    /* 2: the shim writes (default format), the reference reads */
    { FILE *fp = fopen(f2.c_str(), "wb"); shim->write_restart(fp); int tail = 54321; fwrite(&tail, sizeof(int), 1, fp); fclose(fp); }
    {
      PairLJCutCoulLongPolarization *r2 = new PairLJCutCoulLongPolarization(lmp);
      FILE *fp = fopen(f2.c_str(), "rb"); r2->read_restart(fp);
      int tail = 0; size_t got = fread(&tail, sizeof(int), 1, fp); fclose(fp);
      if (got != 1 || tail != 54321) throw SeamError{"the reference did not end where the shim's record ends"};
      lmp->force->pair = r2; r2->init();
      report[2] = compare(r2, shim, ntypes, rmax);
    }
    /* 3: restart_polar yes: the polarization keywords travel in the tagged record */
    {
      PairLJCutCoulLongPolarizationMI355X *s3 = new PairLJCutCoulLongPolarizationMI355X(lmp);
      std::vector<const char *> st(style, style + nstyle);
      const char *more[] = {"restart_polar", "yes", "max_iterations", "77", "polar_gamma", "1.01", "dd_cutoff", "11.5",
                            "polar_gs_ranked", "no", "polar_gs", "yes", "damp", "1.9", "use_previous", "yes"};
      st.insert(st.end(), more, more + 16);
      feed(H, s3, (int)st.size(), st.data(), nmod, mod, ncoeff, rows);
      FILE *fp = fopen(f3.c_str(), "wb"); s3->write_restart(fp); int tail = 777; fwrite(&tail, sizeof(int), 1, fp); fclose(fp);
      PairLJCutCoulLongPolarizationMI355X *s4 = new PairLJCutCoulLongPolarizationMI355X(lmp);
      fp = fopen(f3.c_str(), "rb"); s4->read_restart(fp);
      tail = 0; size_t got = fread(&tail, sizeof(int), 1, fp); fclose(fp);
      polar_settings a, b;
      polar_get_settings(s3->h, &a); polar_get_settings(s4->h, &b);
      double code = 0.0;
      if (got != 1 || tail != 777) code += 1.0;
      if (a.iterations_max != b.iterations_max || b.iterations_max != 77) code += 2.0;
      if (a.polar_gamma != b.polar_gamma || a.dd_cutoff != b.dd_cutoff || a.polar_damp != b.polar_damp) code += 4.0;
      if (a.polar_gs != b.polar_gs || a.polar_gs_ranked != b.polar_gs_ranked || a.use_previous != b.use_previous) code += 8.0;
      if (a.polar_precision != b.polar_precision || a.damping_type != b.damping_type || !b.restart_polar) code += 16.0;
      report[3] = code;
    }
  } catch (SeamError &e) {
    g_last_error = e.msg; rc = -1;
  }
  report[5] = g_max_energy; report[6] = g_max_force;
  snprintf(msg, nmsg, "%s", g_last_error.c_str());
  return rc;
}

/* The same (possibly faulty) input through the reference's pair style (which = 0) or the shim (which = 1): settings, coeff
 * rows, Pair::init.  Returns 0 and an empty message when everything was accepted, else -1 and the text error->all was given.
 * flags: bit 0 clears atom->q_flag, bit 1 clears atom->static_polarizability_flag, bit 2 removes the KSpace style. */
int shimhost_message(int which, int ntypes, int flags, int nstyle, const char *const *style, int ncoeff,
                     const char *const *rows, int do_init, char *msg, int nmsg) {
  g_last_error.clear();
  HostCtx H;
  H.lmp = make_lammps(ntypes, 0.2, 332.06371);
  if (flags & 1) H.lmp->atom->q_flag = 0;
  if (flags & 2) H.lmp->atom->static_polarizability_flag = 0;
  if (flags & 4) H.lmp->force->kspace = NULL;
  int rc = 0;
  try {
    Pair *p = which ? (Pair *)new PairLJCutCoulLongPolarizationMI355X(H.lmp) : (Pair *)new PairLJCutCoulLongPolarization(H.lmp);
    if (which) feed(H, (PairLJCutCoulLongPolarizationMI355X *)p, nstyle, style, 0, NULL, ncoeff, rows);
    else feed(H, (PairLJCutCoulLongPolarization *)p, nstyle, style, 0, NULL, ncoeff, rows);
    if (do_init) { H.lmp->force->pair = p; p->init(); }
  } catch (SeamError &e) {
    g_last_error = e.msg; rc = -1;
  }
  snprintf(msg, nmsg, "%s", g_last_error.c_str());
  return rc;
}

/* build_halo_map (one MPI rank per GPU): LAMMPS order -> library order [own | halo: one ghost per foreign tag | other ghosts]
 * and the half list re-indexed accordingly, special bits kept.  A small hand-made rank: 4 local atoms (tags 11..14), ghosts =
 * two images of local atoms, three foreign atoms of which one appears twice.  Returns 0 when every invariant holds. */
int shimhost_halo_map(char *msg, int nmsg) {
  g_last_error.clear();
  std::string log;
  int bad = 0;
  auto expect = [&](bool ok, const char *what) { if (!ok) { bad++; log += what; log += "; "; } };
  try {
    LAMMPS *lmp = make_lammps(1, 0.2, 332.06371);
    Atom *atom = lmp->atom;
    const int nlocal = 4, nghost = 6, nall = 10;
    atom->nlocal = nlocal; atom->nghost = nghost; atom->nmax = nall;
    lmp->memory->create(atom->tag, nall, "tag");
    const int tags[nall] = {11, 12, 13, 14, /* ghosts: */ 12, 21, 22, 21, 14, 23};
    for (int i = 0; i < nall; i++) atom->tag[i] = tags[i];
    atom->map_style = 1;
    lmp->memory->create(atom->map_array, 32, "map");
    for (int t = 0; t < 32; t++) atom->map_array[t] = -1;
    for (int i = nall - 1; i >= 0; i--) atom->map_array[tags[i]] = i;   // locals win over their images, first ghost of a foreign tag wins
    PairLJCutCoulLongPolarizationMI355X *shim = new PairLJCutCoulLongPolarizationMI355X(lmp);
    /* half list of the rank: rows = locals; entries carry special bits in bits 30-31 */
    NeighList *list = blank<NeighList>();
    static int il[4] = {0, 1, 2, 3};
    static int nn[4] = {3, 2, 2, 1};
    static int r0[3] = {1, 5 | (1 << 30), 7}, r1[2] = {4, 6 | (3 << 30)}, r2[2] = {9, 8}, r3[1] = {5};
    static int *first[4] = {r0, r1, r2, r3};
    list->inum = 4; list->ilist = il; list->numneigh = nn; list->firstneigh = first;
    shim->list = list;
    shim->device_neigh = 0;
    shim->build_halo_map();
    expect(shim->nhalo == 3 && shim->sh_n == nlocal + 3, "halo = one ghost per foreign tag");
    std::vector<int> seen(nall, 0);
    for (int a = 0; a < nall; a++) {
      const int k = shim->lib_of_lammps[a];
      expect(k >= 0 && k < nall && shim->lammps_of_lib[k] == a, "the two maps are inverse");
      if (k >= 0 && k < nall) seen[k]++;
    }
    for (int k = 0; k < nall; k++) expect(seen[k] == 1, "library order is a permutation");
    for (int i = 0; i < nlocal; i++) expect(shim->lib_of_lammps[i] == i, "own atoms keep their indices");
    /* halo slots: tags 21, 22, 23 once each; images of own atoms (tags 12, 14) and the second image of 21 are plain ghosts */
    std::vector<int> halo_tags;
    for (int k = nlocal; k < nlocal + shim->nhalo; k++) halo_tags.push_back(tags[shim->lammps_of_lib[k]]);
    std::sort(halo_tags.begin(), halo_tags.end());
    expect(halo_tags.size() == 3 && halo_tags[0] == 21 && halo_tags[1] == 22 && halo_tags[2] == 23, "halo tags");
    for (int k = nlocal + shim->nhalo; k < nall; k++) {
      const int t = tags[shim->lammps_of_lib[k]];
      expect(t == 12 || t == 14 || t == 21, "plain ghosts");
    }
    /* the re-indexed list: same pairs, same special bits */
    long long at = 0;
    for (int i = 0; i < nlocal; i++) {
      expect(shim->sh_nn[i] == nn[i] && shim->sh_first[i] == at, "row layout");
      for (int k = 0; k < nn[i]; k++) {
        const int e = shim->sh_flat[at + k], o = first[i][k];
        expect((e & ~0x3FFFFFFF) == (o & ~0x3FFFFFFF), "special bits kept");
        expect(shim->lammps_of_lib[e & 0x3FFFFFFF] == (o & 0x3FFFFFFF), "neighbor re-indexed");
      }
      at += nn[i];
    }
  } catch (SeamError &e) {
    bad++; log += "LAMMPS error: " + e.msg;
  }
  snprintf(msg, nmsg, "%s", log.c_str());
  return bad;
}
}
