/* shim_compute_harness.cpp -- TEST INFRASTRUCTURE (build container only; needs /root/reference, never shipped).
 *
 * Executes PairLJCutCoulLongPolarizationMI355X::compute() (lammps_shim/..._mi355x.cpp; counterpart PS.cpp:125-188, 632-645,
 * called from src/verlet.cpp:310) inside the reference's own Pair base class (src/pair.cpp: ev_setup, virial_fdotr_compute)
 * against a RECORDING STUB of the C-ABI: every polar_* entry point the shim calls is defined below, writes down what it was
 * given and returns canned results.  What is checked is the marshalling of atom-> / list-> / ev_setup state into the calls
 * (pointer identities, counts, the eflag / vflag mapping, lists only on reneighbor steps, f += semantics, energy / virial
 * write-back, virial_fdotr_compute reached exactly when the reference reaches it, warning -> error->warning).  No GPU, no
 * library: this file is linked INSTEAD of libpolar_mi355x.so.  It contains no pair arithmetic. */
#define private public
#define protected public
#include "atom.h"
#include "neigh_request.h"
#include "pair_lj_cut_coul_long_polarization_mi355x.h"
#undef private
#undef protected
#include <algorithm>
#include <sstream>
#include "../../oracle/ref_seam/seam_harness.cpp"

int LAMMPS_NS::Atom::map_find_hash(int) { return -1; }

struct polar_handle { int dummy; };

namespace {
struct Rec {
  std::vector<std::string> calls;
  int nlocal, nghost; const double *x, *q, *alpha; const int *type, *mol;
  double boxlo[3], prd[3], tilt[3]; int periodic[3], triclinic;
  int inum; const int *ilist, *numneigh; int *const *firstneigh;
  const double *cutneighsq; const int *tag, *nspecial, *special, *special_flag; int maxspecial, excl;
  int newton;
  int eflag, vflag; double *f, *mu, *ef, *eatom, *vatom;
  polar_settings st;
  int rc_compute;
  std::string modify;
  /* one-MPI-rank-per-GPU path (compute_sharded): what the stepwise entry points were given */
  bool sharded = false;
  int row_lo = -1, row_hi = -1, sweeps = 0; long long global = -1;
  double last_change = 0.0;
  std::vector<double> xcopy, put_val;
  std::vector<int> csr_ilist, csr_nn, csr_flat, put_idx;
  std::vector<long long> csr_first;
  long long get_lo = -1, get_n = -1;
} R;
polar_handle g_handle;
double g_tab[3][64];
void canned(polar_result *res) {
  memset(res, 0, sizeof(*res));
  res->eng_vdwl = 1.25; res->eng_coul = -2.5; res->eng_pol = -0.75; res->u_self = 0.5; res->u_ef = -1.0; res->u_dd = -0.25;
  for (int k = 0; k < 6; k++) res->virial[k] = 10.0 + k;
  res->iterations = 7; res->sweeps = 8; res->status = R.rc_compute;
}
void deposit(int nall, int nlocal, double *f, double *mu, double *ef) {
  for (int k = 0; k < 3 * nall; k++) f[k] += 0.001 * (k + 1);      /* the library ADDS its forces (PS.cpp:293-297, 617-623) */
  for (int k = 0; k < 3 * nlocal; k++) { mu[k] = 100.0 + k; if (ef) ef[k] = 200.0 + k; }
}
}  // namespace

extern "C" {
int polar_device_count(void) { return 1; }
const char *polar_kernel_version(void) { return "recording-stub"; }
int polar_create(int, polar_handle **out) { *out = &g_handle; R.calls.push_back("create"); return POLAR_OK; }
int polar_destroy(polar_handle *) { return POLAR_OK; }
const char *polar_last_error(const polar_handle *) { return "stub error"; }
const char *polar_last_warning(const polar_handle *) { return "Number of iterations exceeding max_iterations, setting dipoles to alpha*E"; }
int polar_pair_settings(polar_handle *, int, const char *const *) { R.calls.push_back("pair_settings"); return POLAR_OK; }
int polar_pair_coeff(polar_handle *, int, int, const char *const *) { R.calls.push_back("pair_coeff"); return POLAR_OK; }
int polar_pair_modify(polar_handle *, int n, const char *const *a) {
  R.calls.push_back("pair_modify");
  R.modify.clear();
  for (int k = 0; k < n; k++) { R.modify += a[k]; R.modify += ' '; }
  return POLAR_OK;
}
int polar_pair_init(polar_handle *, double, double, const double *, const double *) { R.calls.push_back("pair_init"); return POLAR_OK; }
double polar_pair_cut(const polar_handle *, int, int) { return 9.0; }
double polar_pair_single(const polar_handle *, double, double, int, int, double, double, double, double *ff) { if (ff) *ff = 0; return 0.0; }
const void *polar_pair_extract(const polar_handle *, const char *name, int *dim) {
  if (dim) *dim = 2;
  if (!strcmp(name, "epsilon")) return g_tab[0];
  if (!strcmp(name, "sigma")) return g_tab[1];
  if (!strcmp(name, "cut_lj")) return g_tab[2];
  if (!strcmp(name, "cut_coul")) { if (dim) *dim = 0; return &R.st.cut_coul; }
  return NULL;
}
int polar_get_settings(const polar_handle *, polar_settings *out) { *out = R.st; return POLAR_OK; }
int polar_set_settings(polar_handle *, const polar_settings *s) { R.st = *s; return POLAR_OK; }
int polar_restart_pack(const polar_handle *, void *, int) { return 0; }
int polar_restart_unpack(polar_handle *, const void *, int) { return POLAR_OK; }
int polar_set_types(polar_handle *, int, const double *, const double *, const double *, const double *, const double *, const double *, const double *) { return POLAR_OK; }
int polar_set_coul(polar_handle *, double, double, const double *, const double *, int, int, int, double, const double *,
                   const double *, const double *, const double *, const double *, const double *, const double *, const double *) {
  R.calls.push_back("set_coul");
  return POLAR_OK;
}
int polar_set_box(polar_handle *, const double lo[3], const double prd[3], const double tilt[3], const int per[3], int tri) {
  R.calls.push_back("set_box");
  for (int k = 0; k < 3; k++) { R.boxlo[k] = lo[k]; R.prd[k] = prd[k]; R.tilt[k] = tilt[k]; R.periodic[k] = per[k]; }
  R.triclinic = tri;
  return POLAR_OK;
}
int polar_set_atoms(polar_handle *, int nlocal, int nghost, const double *x, const double *q, const double *alpha, const int *type,
                    const int *mol) {
  R.calls.push_back("set_atoms");
  R.nlocal = nlocal; R.nghost = nghost; R.x = x; R.q = q; R.alpha = alpha; R.type = type; R.mol = mol;
  if (R.sharded) R.xcopy.assign(x, x + 3 * (size_t)(nlocal + nghost));   /* (a temporary in library order: keep the numbers) */
  return POLAR_OK;
}
int polar_set_positions(polar_handle *, int nlocal, int nghost, const double *x) {
  R.calls.push_back("set_positions");
  if (nlocal != R.nlocal || nghost != R.nghost) return POLAR_ERR_INPUT;
  R.x = x;
  return POLAR_OK;
}
int polar_set_neighbors(polar_handle *, int inum, const int *ilist, const int *numneigh, int *const *firstneigh) {
  R.calls.push_back("set_neighbors");
  R.inum = inum; R.ilist = ilist; R.numneigh = numneigh; R.firstneigh = firstneigh;
  return POLAR_OK;
}
int polar_set_neighbors_csr(polar_handle *, int inum, const int *il, const int *nn, const long long *first, const int *flat) {
  R.calls.push_back("set_neighbors_csr");
  R.csr_ilist.assign(il, il + inum);
  const int n = R.nlocal;
  R.csr_nn.assign(nn, nn + n); R.csr_first.assign(first, first + n);
  long long tot = 0;
  for (int i = 0; i < inum; i++) tot = std::max(tot, first[il[i]] + nn[il[i]]);
  R.csr_flat.assign(flat, flat + tot);
  return POLAR_OK;
}
int polar_build_neighbors(polar_handle *, const double *cn, const int *tag, const int *nsp, const int *sp, int maxsp, const int sf[4], int excl) {
  R.calls.push_back("build_neighbors");
  R.cutneighsq = cn; R.tag = tag; R.nspecial = nsp; R.special = sp; R.maxspecial = maxsp; R.special_flag = sf; R.excl = excl;
  return POLAR_OK;
}
int polar_set_newton(polar_handle *, int n) { R.calls.push_back("set_newton"); R.newton = n; return POLAR_OK; }
int polar_set_list_style(polar_handle *, int) { return POLAR_OK; }
int polar_compute(polar_handle *, int eflag, int vflag, double *f, double *mu, double *ef, polar_result *out) {
  R.calls.push_back("compute");
  R.eflag = eflag; R.vflag = vflag; R.f = f; R.mu = mu; R.ef = ef; R.eatom = R.vatom = NULL;
  canned(out);
  deposit(R.nlocal + R.nghost, R.nlocal, f, mu, ef);
  return R.rc_compute;
}
int polar_compute_peratom(polar_handle *, int eflag, int vflag, double *f, double *mu, double *ef, double *eatom, double *vatom,
                          polar_result *out) {
  R.calls.push_back("compute_peratom");
  R.eflag = eflag; R.vflag = vflag; R.f = f; R.mu = mu; R.ef = ef; R.eatom = eatom; R.vatom = vatom;
  canned(out);
  deposit(R.nlocal + R.nghost, R.nlocal, f, mu, ef);
  const int nall = R.nlocal + R.nghost;
  if (eatom) for (int k = 0; k < nall; k++) eatom[k] += 0.5 + k;
  if (vatom) for (int k = 0; k < 6 * nall; k++) vatom[k] += 0.25 + k;
  return R.rc_compute;
}
int polar_compute_resident(polar_handle *, int, int, polar_result *) { return POLAR_ERR_STATE; }
void *polar_dev_ptr(polar_handle *, const char *) { return NULL; }
int polar_download(polar_handle *, const char *name, double *dst, long long n) {
  if (!R.sharded) return POLAR_ERR_STATE;
  R.calls.push_back(std::string("download:") + name);
  const double base = !strcmp(name, "f") ? 0.0 : !strcmp(name, "mu") ? 300.0 : !strcmp(name, "ef_static") ? 400.0 : !strcmp(name, "eatom") ? 0.5 : 0.25;
  for (long long k = 0; k < n; k++) dst[k] = !strcmp(name, "f") ? 0.001 * (k + 1) : base + k;   /* library order */
  return POLAR_OK;
}
int polar_upload_mu(polar_handle *, const double *, long long) { if (!R.sharded) return POLAR_ERR_STATE; R.calls.push_back("upload_mu"); return POLAR_OK; }
int polar_get_debug_trace(polar_handle *, double *, int) { return 0; }
int polar_get_debug_forces(polar_handle *, double *o) { for (int k = 0; k < 6; k++) o[k] = 0.0; return 0; }
int polar_set_stream(polar_handle *, void *) { return POLAR_OK; }
int polar_set_row_range(polar_handle *, int lo, int hi) { R.calls.push_back("set_row_range"); R.row_lo = lo; R.row_hi = hi; return POLAR_OK; }
int polar_set_global_count(polar_handle *, long long n) { R.global = n; return POLAR_OK; }
int polar_step_begin(polar_handle *, int ef, int vf) { if (!R.sharded) return POLAR_ERR_STATE; R.calls.push_back("step_begin"); R.eflag = ef; R.vflag = vf; R.sweeps = 0; return POLAR_OK; }
int polar_step_sweep(polar_handle *) { if (!R.sharded) return POLAR_ERR_STATE; R.calls.push_back("sweep"); R.sweeps++; return POLAR_OK; }
int polar_step_sweep_end(polar_handle *, const double *) { if (!R.sharded) return POLAR_ERR_STATE; R.calls.push_back("sweep_end"); return POLAR_OK; }
int polar_step_sweep_end_host(polar_handle *, double v) { if (!R.sharded) return POLAR_ERR_STATE; R.calls.push_back("sweep_end_host"); R.last_change = v; return POLAR_OK; }
int polar_step_state(polar_handle *, int *done, int *it, int *st) { if (!R.sharded) return POLAR_ERR_STATE; R.calls.push_back("state"); *done = R.sweeps >= 3; *it = R.sweeps; *st = 0; return POLAR_OK; }
int polar_step_finish(polar_handle *, polar_result *out) { if (!R.sharded) return POLAR_ERR_STATE; R.calls.push_back("finish"); canned(out); return R.rc_compute; }
int polar_step_mu_get(polar_handle *, long long lo, long long n, double *dst) {
  if (!R.sharded) return POLAR_ERR_STATE;
  R.calls.push_back("mu_get"); R.get_lo = lo; R.get_n = n;
  for (long long k = 0; k < 3 * n; k++) dst[k] = 100.0 + 10.0 * R.sweeps + k;   /* the own dipoles after sweep R.sweeps */
  return POLAR_OK;
}
int polar_step_mu_put_idx(polar_handle *, long long n, const int *idx, const double *src) {
  if (!R.sharded) return POLAR_ERR_STATE;
  R.calls.push_back("mu_put");
  R.put_idx.assign(idx, idx + n); R.put_val.assign(src, src + 3 * n);
  return POLAR_OK;
}
int polar_step_change_get(polar_handle *, double *c) { if (!R.sharded) return POLAR_ERR_STATE; R.calls.push_back("change_get"); *c = 1.0 / R.sweeps; return POLAR_OK; }
// the multi-rank driver (rccl_halo yes): recorded like the rest.  polar_dist_step returns the sums over the (two) ranks,
// polar_dist_local_result this rank's own share -- what the shim must add to LAMMPS' per-rank accumulators
int polar_dist_unique_id(void *) { return POLAR_ERR_STATE; }
int polar_dist_create(const void *, int, int, int, polar_dist **) { return POLAR_ERR_STATE; }
int polar_dist_destroy(polar_dist *) { return POLAR_OK; }
const char *polar_dist_last_error(const polar_dist *) { return "stub"; }
int polar_dist_set_halo(polar_dist *, polar_handle *, int, const int *, const int *, const int *, const int *, const int *) { return POLAR_ERR_STATE; }
int polar_dist_set_schedule(polar_dist *, int, int, int) { return POLAR_ERR_STATE; }
int polar_dist_step(polar_dist *, polar_handle *, int ef, int vf, polar_result *out) {
  if (!R.sharded) return POLAR_ERR_STATE;
  R.calls.push_back("dist_step"); R.eflag = ef; R.vflag = vf;
  canned(out);
  out->eng_vdwl *= 2.0; out->eng_coul *= 2.0; out->eng_pol *= 2.0; out->u_self *= 2.0; out->u_ef *= 2.0; out->u_dd *= 2.0;   // two ranks with equal shares
  for (int k = 0; k < 6; k++) out->virial[k] *= 2.0;
  return R.rc_compute;
}
int polar_dist_local_result(const polar_dist *, polar_result *out) { R.calls.push_back("dist_local"); canned(out); return POLAR_OK; }
}

namespace {
struct Fail { std::string msg; };
#define EXPECT(cond, what)                                                                    \
  do {                                                                                        \
    if (!(cond)) { std::ostringstream o_; o_ << tag << ": " << what << " [" #cond "]"; throw Fail{o_.str()}; } \
  } while (0)
}  // namespace

/* the LAMMPS objects the shim reads in compute(): five local atoms, three ghosts, a paged half list */
struct World {
  LAMMPS *lmp; Force *force; Domain *dom; Atom *atom; Neighbor *nb; NeighList *list; Memory *mem;
  int *ilist, *numneigh; int **first;
};
static World make_world(int nlocal, int nghost, int ntypes) {
  const int nall = nlocal + nghost;
  LAMMPS *lmp = blank<LAMMPS>();
  lmp->screen = NULL;
  lmp->error = blank<Error>(); lmp->atom = blank<Atom>(); lmp->force = blank<Force>(); lmp->domain = blank<Domain>();
  lmp->neighbor = blank<Neighbor>(); lmp->update = blank<Update>(); lmp->comm = blank<Comm>();
  lmp->memory = new Memory(lmp);
  static void *vtab[64];
  for (int k = 0; k < 64; k++) vtab[k] = (void *)&seam_nop;
  *(void ***)lmp->comm = vtab;
  lmp->comm->me = 0; lmp->comm->nprocs = 1; lmp->comm->nthreads = 1;
  KSpace *ks = blank<KSpace>();
  ks->g_ewald = 0.2;
  Force *force = lmp->force;
  force->kspace = ks; force->qqrd2e = force->qqr2e = 332.06371; force->dielectric = 1.0;
  force->newton = force->newton_pair = force->newton_bond = 1;
  lmp->update->whichflag = 1;
  static char verlet[] = "verlet";
  lmp->update->integrate_style = verlet;
  Domain *dom = lmp->domain;
  dom->dimension = 3; dom->triclinic = 0;
  dom->xperiodic = 1; dom->yperiodic = 1; dom->zperiodic = 0;
  dom->boxlo[0] = -1.0; dom->boxlo[1] = -2.0; dom->boxlo[2] = -3.0;
  dom->xprd = dom->prd[0] = 30.0; dom->yprd = dom->prd[1] = 31.0; dom->zprd = dom->prd[2] = 32.0;
  dom->xy = 0.5; dom->xz = 0.25; dom->yz = 0.125;
  Atom *atom = lmp->atom;
  atom->nlocal = nlocal; atom->nghost = nghost; atom->nmax = nall + 2; atom->ntypes = ntypes; atom->natoms = nlocal;
  atom->q_flag = 1; atom->static_polarizability_flag = 1; atom->molecular = 1; atom->maxspecial = 3;
  Memory *mem = lmp->memory;
  mem->create(atom->x, atom->nmax, 3, "x"); mem->create(atom->f, atom->nmax, 3, "f");
  mem->create(atom->ef_static, atom->nmax, 3, "ef"); mem->create(atom->mu_induced, atom->nmax, 3, "mu");
  mem->create(atom->q, atom->nmax, "q"); mem->create(atom->static_polarizability, atom->nmax, "a");
  mem->create(atom->type, atom->nmax, "t"); mem->create(atom->molecule, atom->nmax, "m"); mem->create(atom->tag, atom->nmax, "tag");
  mem->create(atom->nspecial, atom->nmax, 3, "ns"); mem->create(atom->special, atom->nmax, atom->maxspecial, "sp");
  for (int i = 0; i < atom->nmax; i++) {
    for (int k = 0; k < 3; k++) { atom->x[i][k] = 1.0 + 0.37 * i + 0.11 * k; atom->nspecial[i][k] = 0; atom->special[i][k] = 0; }
    atom->q[i] = 0.1 * (i % 3 - 1); atom->static_polarizability[i] = 0.5; atom->type[i] = 1 + i % 2; atom->molecule[i] = i / 2; atom->tag[i] = i + 1;
  }
  Neighbor *nb = lmp->neighbor;
  nb->includegroup = 0; nb->nex_type = nb->nex_group = nb->nex_mol = 0;
  static NeighRequest *reqs[2];
  reqs[0] = blank<NeighRequest>(); reqs[1] = blank<NeighRequest>();
  nb->requests = reqs;
  mem->create(nb->cutneighsq, ntypes + 1, ntypes + 1, "cn");
  for (int i = 0; i <= ntypes; i++) for (int j = 0; j <= ntypes; j++) nb->cutneighsq[i][j] = 121.0;
  nb->special_flag[0] = 1; nb->special_flag[1] = 2; nb->special_flag[2] = 2; nb->special_flag[3] = 2;
  NeighList *list = blank<NeighList>();
  static int ilist[5] = {0, 1, 2, 3, 4}, numneigh[5] = {2, 1, 1, 0, 1};
  static int r0[2] = {1, 5}, r1[1] = {2}, r2[1] = {6}, r4[1] = {7};
  static int *first[5] = {r0, r1, r2, NULL, r4};
  list->inum = 5; list->ilist = ilist; list->numneigh = numneigh; list->firstneigh = first;

  World W{lmp, force, dom, atom, nb, list, mem, ilist, numneigh, first};
  return W;
}

extern "C" {
/* runs every combination; returns 0, or -1 with the first mismatch in msg.  *ncombos = combinations executed */
int shimcompute_check(int *ncombos, char *msg, int nmsg) {
  g_last_error.clear();
  *ncombos = 0;
  const int nlocal = 5, nghost = 3, nall = nlocal + nghost, ntypes = 2;
  World W = make_world(nlocal, nghost, ntypes);
  LAMMPS *lmp = W.lmp; Force *force = W.force; Atom *atom = W.atom; Neighbor *nb = W.nb; NeighList *list = W.list;
  int *ilist = W.ilist, *numneigh = W.numneigh; int **first = W.first;
  (void)nall;
  int rc = 0;
  try {
    for (int dn = 0; dn < 2; dn++) {
      memset(&R.st, 0, sizeof(R.st));
      R.st.cut_lj_global = 9.0; R.st.cut_coul = 9.0; R.st.iterations_max = 50; R.st.device_neigh = dn;
      R.calls.clear();
      PairLJCutCoulLongPolarizationMI355X *shim = new PairLJCutCoulLongPolarizationMI355X(lmp);
      force->pair = shim;
      shim->ncoultablebits = 0;                 /* pair_modify table 0: init_style then has no table to build */
      {
        std::string tag = "init";
        char a0[] = "9.0", a1[] = "9.0"; char *sa[2] = {a0, a1};
        shim->settings(2, sa);
        char c0[] = "*", c1[] = "*", c2[] = "0.1", c3[] = "3.0"; char *ca[4] = {c0, c1, c2, c3};
        shim->coeff(4, ca);
        shim->init_style();
        shim->init_list(0, list);
        EXPECT(shim->device_neigh == dn, "init_style takes device_neigh from the library's settings");
        EXPECT(shim->no_virial_fdotr_compute == dn, "a device-built full list must keep the base class off the fdotr virial");
        EXPECT(R.modify.find("table 0") != std::string::npos && R.modify.find("mix geometric") != std::string::npos, "pair_modify state mirrored into the library: " << R.modify);
      }
      bool first_call = true;
      for (int rcw = 0; rcw < 2; rcw++)
      for (int ago = 3; ago >= 0; ago -= 3)       /* (ago 3 first: the very first call must hand the atoms over all the same) */
      for (int eflag = 0; eflag <= 3; eflag++)
      for (int vflag = 0; vflag <= 6; vflag++) {
        if (vflag == 3) continue;               /* integrate::ev_set never produces 3 or 7 */
        std::ostringstream t; t << "device_neigh " << dn << " ago " << ago << " eflag " << eflag << " vflag " << vflag << " rc " << rcw;
        const std::string tag = t.str();
        nb->ago = ago;
        R.rc_compute = rcw ? POLAR_WARN_NOT_CONVERGED : POLAR_OK;
        g_warnings = 0; g_last_warning.clear();
        R.calls.clear();
        std::vector<double> f0(3 * nall);
        for (int k = 0; k < 3 * nall; k++) { f0[k] = 7.0 + 0.5 * k; atom->f[0][k] = f0[k]; atom->mu_induced[0][k] = -1.0; atom->ef_static[0][k] = -2.0; }
        shim->eng_vdwl = 1000.0; shim->eng_coul = 2000.0; shim->eng_pol = 3000.0;   /* what an earlier step left behind */
        for (int k = 0; k < 6; k++) shim->virial[k] = 500.0 + k;
        shim->compute(eflag, vflag);
        (*ncombos)++;
        /* ---- the call sequence (PS.cpp:125-188: everything compute() reads, lists only when they are new) ---- */
        /* (the first call of a run and every reneighbor step hand all per-atom arrays over; steps in between only x) */
        std::vector<std::string> want = {"set_box", (ago == 0 || first_call) ? "set_atoms" : "set_positions"};
        first_call = false;
        if (ago == 0) { want.push_back("set_newton"); want.push_back(dn ? "build_neighbors" : "set_neighbors"); }
        const bool peratom = (eflag / 2) || (vflag / 4);
        want.push_back(peratom ? "compute_peratom" : "compute");
        std::string got, exp;
        for (auto &c : R.calls) got += c + " ";
        for (auto &c : want) exp += c + " ";
        EXPECT(got == exp, "call sequence: got '" << got << "' expected '" << exp << "'");
        /* ---- pointer identities and counts ---- */
        EXPECT(R.nlocal == nlocal && R.nghost == nghost, "atom counts");
        EXPECT(R.x == &atom->x[0][0] && R.q == atom->q && R.alpha == atom->static_polarizability && R.type == atom->type &&
               (const void *)R.mol == (const void *)atom->molecule, "atom arrays handed over in place");
        EXPECT(R.prd[0] == 30.0 && R.prd[1] == 31.0 && R.prd[2] == 32.0 && R.boxlo[2] == -3.0 && R.tilt[0] == 0.5 && R.tilt[1] == 0.25 &&
               R.tilt[2] == 0.125 && R.periodic[0] == 1 && R.periodic[2] == 0 && R.triclinic == 0, "box");
        if (ago == 0 && !dn) EXPECT(R.inum == 5 && R.ilist == ilist && R.numneigh == numneigh && R.firstneigh == first, "LAMMPS' paged list handed over as it is");
        if (ago == 0 && dn) EXPECT(R.cutneighsq == &nb->cutneighsq[0][0] && (const void *)R.tag == (const void *)atom->tag && R.nspecial == &atom->nspecial[0][0] &&
                                   (const void *)R.special == (const void *)&atom->special[0][0] && R.maxspecial == 3 && R.special_flag == nb->special_flag && R.excl == 0, "device list inputs");
        if (ago == 0) EXPECT(R.newton == 1, "newton_pair");
        EXPECT(R.f == &atom->f[0][0] && R.mu == &atom->mu_induced[0][0] && R.ef == &atom->ef_static[0][0], "output arrays in place");
        /* ---- eflag / vflag mapping (shim line "const int vf = ...") ---- */
        const int eg = eflag % 2, ea = eflag / 2, vg = vflag % 4, va = vflag / 4;
        const bool fdotr = vg == 2 && !dn;       /* ev_setup, src/pair.cpp:812-817 */
        const int vgl = fdotr ? 0 : vg;          /* vflag_global after ev_setup */
        const int want_ef = (eflag ? 1 : 0) | (ea ? 2 : 0);     /* eflag_either | eflag_atom << 1 */
        const int want_vf = (dn ? vgl : (vgl ? 1 : 0)) | (va ? 4 : 0);
        if (eflag || vflag) {
          EXPECT(R.eflag == want_ef, "eflag handed to the library: " << R.eflag << " expected " << want_ef);
          EXPECT(R.vflag == want_vf, "vflag handed to the library: " << R.vflag << " expected " << want_vf);
        } else EXPECT(R.eflag == 0 && R.vflag == 0, "no tallies asked for");
        if (peratom) {
          EXPECT((R.eatom != NULL) == (ea != 0) && (R.vatom != NULL) == (va != 0), "per-atom arrays only when asked for");
          if (ea) EXPECT(R.eatom == shim->eatom && shim->eatom[2] == 0.5 + 2, "Pair::eatom zeroed by ev_setup, then filled");
          if (va) EXPECT(R.vatom == &shim->vatom[0][0] && shim->vatom[1][1] == 0.25 + 7, "Pair::vatom zeroed by ev_setup, then filled");
        }
        /* ---- results: f += , dipoles and static field in place ---- */
        for (int k = 0; k < 3 * nall; k++) EXPECT(atom->f[0][k] == f0[k] + 0.001 * (k + 1), "forces are ADDED to atom->f, entry " << k);
        EXPECT(atom->mu_induced[0][4] == 104.0 && atom->ef_static[0][4] == 204.0, "mu_induced / ef_static of the local atoms");
        EXPECT(atom->mu_induced[nlocal][0] == -1.0, "ghost dipoles untouched");
        /* ---- energies (ev_setup zeroes them only when eflag_global; PS.cpp:641 sets eng_pol every step) ---- */
        if (eg) EXPECT(shim->eng_vdwl == 1.25 && shim->eng_coul == -2.5, "eng_vdwl / eng_coul");
        else EXPECT(shim->eng_vdwl == 1000.0 && shim->eng_coul == 2000.0, "energies untouched without eflag_global");
        EXPECT(force->pair->eng_pol == -0.75, "eng_pol");
        /* ---- virial: pairwise from the library, or f.x by the base class exactly when the reference's compute() would ---- */
        double fx[6] = {0, 0, 0, 0, 0, 0};
        for (int i = 0; i < nall; i++) {
          const double *ff = atom->f[i], *xx = atom->x[i];
          fx[0] += ff[0] * xx[0]; fx[1] += ff[1] * xx[1]; fx[2] += ff[2] * xx[2]; fx[3] += ff[1] * xx[0]; fx[4] += ff[2] * xx[0]; fx[5] += ff[2] * xx[1];
        }
        for (int k = 0; k < 6; k++) {
          double expect;
          if (fdotr) expect = fx[k];                        /* zeroed by ev_setup (vflag_global was still 2 there), then f.x (PS.cpp:644) */
          else if (vgl) expect = 10.0 + k;                  /* zeroed by ev_setup, the library's tally added */
          else expect = 500.0 + k;                          /* untouched */
          EXPECT(fabs(shim->virial[k] - expect) <= 1e-12 * fabs(expect), "virial[" << k << "] = " << shim->virial[k] << " expected " << expect);
        }
        EXPECT(shim->vflag_fdotr == 0, "virial_fdotr_compute ran (it clears the flag) or was never asked for");
        /* ---- non-convergence -> error->warning with the reference's text (PS.cpp:1233) ---- */
        EXPECT(g_warnings == (rcw ? 1 : 0), "warnings raised: " << g_warnings);
        if (rcw) EXPECT(g_last_warning == "Number of iterations exceeding max_iterations, setting dipoles to alpha*E", "warning text");
      }
      /* a library error reaches error->all with the library's message */
      {
        std::string tag = "error path";
        R.rc_compute = POLAR_ERR_HIP;
        bool thrown = false;
        try { shim->compute(1, 2); } catch (SeamError &e) { thrown = e.msg == "stub error"; }
        EXPECT(thrown, "a negative status must stop the run through error->all");
      }
    }
  } catch (Fail &f) {
    snprintf(msg, nmsg, "%s", f.msg.c_str());
    rc = -1;
  } catch (SeamError &e) {
    snprintf(msg, nmsg, "error->all: %s", e.msg.c_str());
    rc = -1;
  }
  return rc;
}

/* ---- one MPI rank per GPU: compute() with comm->nprocs > 1 goes through compute_sharded() ------------------------------
 * Serial MPI (the reference's STUBS), comm->nprocs set to 2: the path a rank of a multi-rank run takes, with this rank's
 * own contribution as "the sum over ranks".  Comm::forward_comm_pair is emulated through the shim's own pack / unpack:
 * images of own atoms get their owner's dipole, atoms owned elsewhere the numbers "the other rank" sent.
 * Checked: library order [own | halo | other ghosts], row range and global count, the re-indexed CSR list, the per-sweep
 * sequence (sweep, sum, MPI_Allreduce, end of sweep, dipoles out, forward communication, halo dipoles in, state), the stop
 * at the sweep the library reports, results scattered back in LAMMPS order. */
}  /* extern "C" */
namespace {
LAMMPS_NS::Atom *g_atom = NULL;
int g_nlocal = 0, g_nall = 0, g_fwd = 0;
void seam_forward_comm(void *, void *pair_) {
  PairLJCutCoulLongPolarizationMI355X *pair = (PairLJCutCoulLongPolarizationMI355X *)pair_;
  g_fwd++;
  for (int g = g_nlocal; g < g_nall; g++) {
    double buf[3];
    const int owner = g_atom->map(g_atom->tag[g]);
    if (owner >= 0 && owner < g_nlocal) { int one = owner; pair->pack_forward_comm(1, &one, buf, 0, NULL); }
    else { buf[0] = 900.0 + 10 * g + g_fwd; buf[1] = buf[0] + 1.0; buf[2] = buf[0] + 2.0; }   /* what its owner's rank sent */
    pair->unpack_forward_comm(1, g, buf);
  }
}
}  // namespace
extern "C" {
int shimsharded_check(int *ncombos, char *msg, int nmsg) {
  g_last_error.clear();
  *ncombos = 0;
  const int nlocal = 5, nghost = 3, nall = nlocal + nghost, ntypes = 2;
  World W = make_world(nlocal, nghost, ntypes);
  LAMMPS *lmp = W.lmp; Force *force = W.force; Atom *atom = W.atom; Neighbor *nb = W.nb; NeighList *list = W.list;
  /* two ranks: ghost 5 and ghost 7 belong to the other one (tags 101, 102), ghost 6 is an image of own atom 1 (tag 2) */
  lmp->comm->nprocs = 2;
  lmp->comm->cutghost[0] = lmp->comm->cutghost[1] = lmp->comm->cutghost[2] = 20.0;
  static void *vtab2[64];
  for (int k = 0; k < 64; k++) vtab2[k] = (void *)&seam_forward_comm;
  *(void ***)lmp->comm = vtab2;
  atom->natoms = 12;
  atom->tag[5] = 101; atom->tag[6] = 2; atom->tag[7] = 102;
  static int map_array[128];
  for (int k = 0; k < 128; k++) map_array[k] = -1;
  for (int i = nall - 1; i >= 0; i--) map_array[atom->tag[i]] = i;     /* owned atoms win over their images, like Atom::map_set */
  atom->map_style = 1; atom->map_array = map_array; atom->map_tag_max = 127;
  g_atom = atom; g_nlocal = nlocal; g_nall = nall;
  R.sharded = true;
  int rc = 0;
  try {
    for (int dn = 0; dn < 2; dn++)
    for (int fixed = 0; fixed < 2; fixed++)
    for (int eflag = 0; eflag <= 3; eflag += 3)
    for (int vflag = 0; vflag <= 6; vflag += 2) {
      std::ostringstream t; t << "sharded: device_neigh " << dn << " fixed_iteration " << fixed << " eflag " << eflag << " vflag " << vflag;
      const std::string tag = t.str();
      memset(&R.st, 0, sizeof(R.st));
      R.st.cut_lj_global = 9.0; R.st.cut_coul = 9.0; R.st.dd_cutoff = 9.0; R.st.iterations_max = 4; R.st.device_neigh = dn; R.st.fixed_iteration = fixed;
      R.rc_compute = POLAR_OK;
      PairLJCutCoulLongPolarizationMI355X *shim = new PairLJCutCoulLongPolarizationMI355X(lmp);
      force->pair = shim;
      shim->ncoultablebits = 0;
      char a0[] = "9.0", a1[] = "9.0"; char *sa[2] = {a0, a1};
      shim->settings(2, sa);
      char c0[] = "*", c1[] = "*", c2[] = "0.1", c3[] = "3.0"; char *ca[4] = {c0, c1, c2, c3};
      shim->coeff(4, ca);
      shim->init_style();
      shim->init_list(0, list);
      nb->ago = 0;
      g_fwd = 0;
      std::vector<double> f0(3 * nall);
      for (int k = 0; k < 3 * nall; k++) { f0[k] = 7.0 + 0.5 * k; atom->f[0][k] = f0[k]; atom->mu_induced[0][k] = -1.0; atom->ef_static[0][k] = -2.0; }
      R.calls.clear();
      shim->compute(eflag, vflag);
      (*ncombos)++;
      /* library order: own atoms keep their places, then ONE ghost per foreign tag, then the images of own atoms */
      EXPECT(R.nlocal == nlocal + 2 && R.nghost == 1, "set_atoms: " << R.nlocal << " + " << R.nghost << " (own + halo, other ghosts)");
      const int lib_of[8] = {0, 1, 2, 3, 4, 5, 7, 6};
      for (int a = 0; a < nall; a++)
        for (int c = 0; c < 3; c++) EXPECT(R.xcopy[3 * lib_of[a] + c] == atom->x[a][c], "positions in library order, atom " << a);
      EXPECT(R.row_lo == 0 && R.row_hi == nlocal, "row range = the own atoms");
      EXPECT(R.global == 12, "global atom count for the stop rule");
      EXPECT(R.newton == 1, "newton_pair");
      if (!dn) {
        const int flat_want[5] = {1, 5, 2, 7, 6};     /* rows 0: {1, 5}, 1: {2}, 2: {6 -> 7}, 4: {7 -> 6} */
        EXPECT(R.csr_ilist.size() == 5 && R.csr_flat.size() == 5, "CSR list sizes");
        for (int k = 0; k < 5; k++) EXPECT(R.csr_flat[k] == flat_want[k], "neighbor " << k << " re-indexed into library order: " << R.csr_flat[k]);
        EXPECT(R.csr_nn[3] == 0 && R.csr_nn[0] == 2, "row lengths");
      } else {
        EXPECT(R.tag[5] == 101 && R.tag[6] == 102 && R.tag[7] == 2, "tags in library order for the device list");
      }
      /* the call sequence */
      std::string want = "set_box set_atoms set_row_range set_newton ";
      want += dn ? "build_neighbors " : "set_neighbors_csr ";
      want += "step_begin mu_get mu_put ";
      const int nsweeps = fixed ? R.st.iterations_max + 1 : 3;       /* fixed: every sweep; otherwise until the library says done */
      for (int sw = 0; sw < nsweeps; sw++) want += fixed ? "sweep sweep_end mu_get mu_put " : "sweep change_get sweep_end_host mu_get mu_put state ";
      want += "finish download:f download:mu download:ef_static ";
      if (eflag / 2) want += "download:eatom ";
      if (vflag / 4) want += "download:vatom ";
      std::string got;
      for (auto &c : R.calls) got += c + " ";
      EXPECT(got == want, "call sequence: got '" << got << "' expected '" << want << "'");
      if (!fixed) EXPECT(R.last_change == 1.0 / 3.0, "the all-reduced sum goes back in (serial MPI: this rank's own)");
      EXPECT(R.get_lo == 0 && R.get_n == nlocal, "own dipoles fetched for the forward communication");
      /* halo dipoles: slot k of the halo = ghost halo_ghost[k], as the last forward communication left it */
      EXPECT(R.put_idx.size() == 2 && R.put_idx[0] == nlocal && R.put_idx[1] == nlocal + 1, "halo rows");
      EXPECT(R.put_val[0] == 900.0 + 50 + g_fwd && R.put_val[3] == 900.0 + 70 + g_fwd, "halo dipoles of ghosts 5 and 7: " << R.put_val[0] << " " << R.put_val[3]);
      EXPECT(g_fwd == nsweeps + 1, "one forward communication after the initial guess and one per sweep: " << g_fwd);
      /* images of own atoms carry their owner's dipole after the forward communication (pack / unpack of the shim) */
      /* results in LAMMPS order */
      for (int a = 0; a < nall; a++)
        for (int c = 0; c < 3; c++)
          EXPECT(fabs(atom->f[a][c] - (f0[3 * a + c] + 0.001 * (3 * lib_of[a] + c + 1))) < 1e-12, "forces added in LAMMPS order, atom " << a);
      EXPECT(atom->mu_induced[0][4] == 304.0 && atom->ef_static[0][4] == 404.0, "dipoles / static field of the own atoms");
      const int eg = eflag % 2;
      if (eg) EXPECT(shim->eng_vdwl == 1.25 && shim->eng_coul == -2.5, "energies of this rank's rows");
      EXPECT(force->pair->eng_pol == -0.75, "eng_pol");
      if (eflag / 2) EXPECT(shim->eatom[6] == 0.5 + 7 && shim->eatom[7] == 0.5 + 6, "per-atom energies back in LAMMPS order");
      if (vflag / 4) EXPECT(shim->vatom[6][0] == 0.25 + 6 * 7, "per-atom virial back in LAMMPS order");
    }
    /* rccl_halo yes: the whole solve in the library's RCCL driver; energies and virial must be THIS rank's share, not the
       driver's sums over the ranks (LAMMPS adds the ranks up itself) */
    {
      struct ShimNoPlan : PairLJCutCoulLongPolarizationMI355X {
        explicit ShimNoPlan(LAMMPS *l) : PairLJCutCoulLongPolarizationMI355X(l) {}
        void build_rccl_plan() override {}          /* (the plan exchange needs real MPI ranks) */
      };
      for (int dn = 0; dn < 2; dn++) {
        std::ostringstream t; t << "sharded rccl_halo: device_neigh " << dn;
        const std::string tag = t.str();
        memset(&R.st, 0, sizeof(R.st));
        R.st.cut_lj_global = 9.0; R.st.cut_coul = 9.0; R.st.dd_cutoff = 9.0; R.st.iterations_max = 4; R.st.device_neigh = dn; R.st.rccl_halo = 1;
        R.rc_compute = POLAR_OK;
        ShimNoPlan *shim = new ShimNoPlan(lmp);
        force->pair = shim; shim->ncoultablebits = 0;
        char a0[] = "9.0", a1[] = "9.0"; char *sa[2] = {a0, a1};
        shim->settings(2, sa);
        char c0[] = "*", c1[] = "*", c2[] = "0.1", c3[] = "3.0"; char *ca[4] = {c0, c1, c2, c3};
        shim->coeff(4, ca); shim->init_style(); shim->init_list(0, list);
        nb->ago = 0;
        for (int k = 0; k < 3 * nall; k++) { atom->f[0][k] = 0.0; atom->mu_induced[0][k] = -1.0; atom->ef_static[0][k] = -2.0; }
        shim->eng_vdwl = shim->eng_coul = 0.0;
        for (int k = 0; k < 6; k++) shim->virial[k] = 0.0;
        R.calls.clear();
        shim->compute(1, 1);
        (*ncombos)++;
        std::string got;
        for (auto &c : R.calls) got += c + " ";
        std::string want = "set_box set_atoms set_row_range set_newton ";
        want += dn ? "build_neighbors " : "set_neighbors_csr ";
        want += "dist_step dist_local download:f download:mu download:ef_static ";
        EXPECT(got == want, "call sequence: got '" << got << "' expected '" << want << "'");
        EXPECT(shim->eng_vdwl == 1.25 && shim->eng_coul == -2.5 && force->pair->eng_pol == -0.75, "energies = this rank's share: " << shim->eng_vdwl << " " << shim->eng_coul << " " << force->pair->eng_pol);
        for (int k = 0; k < 6; k++) EXPECT(shim->virial[k] == 10.0 + k, "virial[" << k << "] = this rank's share: " << shim->virial[k]);
      }
    }
    /* the checks compute_sharded makes before anything moves */
    {
      std::string tag = "sharded: exact mode refused";
      memset(&R.st, 0, sizeof(R.st));
      R.st.cut_lj_global = 9.0; R.st.cut_coul = 9.0; R.st.dd_cutoff = 0.0; R.st.iterations_max = 4;
      PairLJCutCoulLongPolarizationMI355X *shim = new PairLJCutCoulLongPolarizationMI355X(lmp);
      force->pair = shim; shim->ncoultablebits = 0;
      char a0[] = "9.0", a1[] = "9.0"; char *sa[2] = {a0, a1};
      shim->settings(2, sa);
      char c0[] = "*", c1[] = "*", c2[] = "0.1", c3[] = "3.0"; char *ca[4] = {c0, c1, c2, c3};
      shim->coeff(4, ca); shim->init_style(); shim->init_list(0, list);
      bool thrown = false;
      try { shim->compute(1, 2); } catch (SeamError &e) { thrown = e.msg.find("dd_cutoff") != std::string::npos; }
      EXPECT(thrown, "several ranks without dd_cutoff must stop with the dd_cutoff message");
      R.st.dd_cutoff = 9.0;
      lmp->comm->cutghost[1] = 5.0;
      thrown = false;
      try { shim->compute(1, 2); } catch (SeamError &e) { thrown = e.msg.find("ghost cutoff") != std::string::npos; }
      EXPECT(thrown, "a ghost shell shorter than the list reach must stop the run");
      lmp->comm->cutghost[1] = 20.0;
    }
  } catch (Fail &f) {
    snprintf(msg, nmsg, "%s", f.msg.c_str());
    rc = -1;
  } catch (SeamError &e) {
    snprintf(msg, nmsg, "error->all: %s", e.msg.c_str());
    rc = -1;
  }
  R.sharded = false;
  return rc;
}
}
