/* polar_mi355x.h -- C-ABI of the MI355X-native lj/cut/coul/long/polarization pair style.
 *
 * Drop-in boundary for ONE hot path of aehogan/lammps-induced-dipole-polarization-pair-style:
 * PairLJCutCoulLongPolarization::compute and the text/host interface around it
 * (reference: src/pair_lj_cut_coul_long_polarization.{h,cpp}; "PS.cpp"/"PS.h" below).
 * The reference has no FFI (the pair style is compiled into LAMMPS); the entry points here are
 * what a `Pair` subclass shim binds instead of the reference's in-class code
 * (lammps_shim/pair_lj_cut_coul_long_polarization_mi355x.cpp).
 *
 * Plain C types only, FP64, row-major [n][3] arrays, host pointers unless a name says `dev`.
 * Every function returns an int status: 0 ok, >0 warning, <0 fatal (message via
 * polar_last_error).  Not thread-safe per handle; calls are synchronous on return.
 * The implementation is hand-written HIP for gfx950; there is NO CPU fallback: without a
 * usable GPU every compute entry point fails with POLAR_ERR_NO_DEVICE.
 */
#ifndef POLAR_MI355X_H
#define POLAR_MI355X_H

#ifdef __cplusplus
extern "C" {
#endif

#define POLAR_OK 0
#define POLAR_WARN_NOT_CONVERGED 1 /* PS.cpp:1227-1235: mu reset to alpha*E, warning text in polar_last_warning */
#define POLAR_RETRY_STEP 2         /* polar_step_finish only: a list row outgrew its pitch; the pitch was enlarged, the
                                      results of this step are void and the driver repeats the step (all ranks together) */
#define POLAR_ERR_INPUT -1         /* reference error->all() conditions; message = reference text */
#define POLAR_ERR_NO_DEVICE -2
#define POLAR_ERR_HIP -3
#define POLAR_ERR_UNSUPPORTED -4   /* not offered in this mode (e.g. exact mode on a row-sharded handle) */
#define POLAR_ERR_STATE -5         /* call order (e.g. compute before set_atoms) */

enum { POLAR_DAMP_EXPONENTIAL = 0, POLAR_DAMP_NONE = 1 }; /* PS.cpp:51 */

typedef struct polar_handle polar_handle;

/* pair_style keyword state.  Defaults = PS.cpp:65-78. */
typedef struct {
  double cut_lj_global, cut_coul;
  double polar_precision, polar_damp, polar_gamma;
  int iterations_max, damping_type, zodid, fixed_iteration;
  int polar_gs, polar_gs_ranked, use_previous, debug;
  /* --- extension, not in the reference ---
   * dd_cutoff <= 0 : reference semantics: static field / charge-dipole cut at cut_coul,
   *                  dipole-dipole over ALL minimum-image pairs (exact, O(N^2)).
   * dd_cutoff  > 0 : dipole tensor and dipole-dipole force truncated at rsq < dd_cutoff^2 and all
   *                  polarization loops run over a device-built cell/neighbor list (O(N K)).
   *                  Identical to the reference when dd_cutoff >= sqrt(3)/2 * L.               */
  double dd_cutoff;
  /* device_neigh yes|no (extension keyword): the LAMMPS shim lets the library build the LJ/Coulomb
   * neighbor list on the device (polar_build_neighbors) instead of uploading Neighbor's list. */
  int device_neigh;
  /* restart_polar yes|no (extension keyword): write_restart_settings appends the polarization keywords behind the
   * stock record (polar_restart_pack); off by default, so restart files keep the reference's layout (PS.cpp:976-985),
   * which stores none of them.  The dipoles persist through the atom style (AtomVecFullPolar::pack_restart). */
  int restart_polar;
  /* deterministic yes|no (extension keyword, list mode): no sweep launch reads a dipole that another wave of the same
   * launch writes -- updates are committed between launches and between the sub-phases of a tile -- so that a run,
   * `fixed_iteration` runs included, is reproducible bit for bit like the reference's serial loop (PS.cpp:1158-1180).
   * Keyword not given (POLAR_DET_AUTO = 0, also what a zero-initialised struct says): ON for `fixed_iteration yes` -- the
   * reference returns one well-defined unconverged iterate there (PS.cpp:1211-1215), and the in-place race of a colour phase
   * would make it differ from run to run at 4e-6 of the largest dipole --, OFF for precision runs: they converge to the same
   * fixed point either way (run-to-run spread 1e-9) and save the commit launches (+5 %).  `deterministic no` (POLAR_DET_NO)
   * keeps the in-place update for fixed-iteration runs as well. */
#define POLAR_DET_AUTO 0
#define POLAR_DET_YES 1
#define POLAR_DET_NO 2
  int deterministic;
  /* polar_sor <omega> (extension keyword, list mode with polar_gs / polar_gs_ranked): successive over-relaxation of the
   * Gauss-Seidel update, mu <- mu + omega (alpha (E_static + E_ind) - mu).  1 (default) is the reference's update
   * (PS.cpp:1170-1180); 1.1 - 1.2 reaches the same fixed point under the same stop rule in about 30 % fewer sweeps on the MOF
   * boxes (profiles/r03_lab_det_sor.txt).  0 < omega < 2.  Single-handle runs only: across ranks (block-Jacobi) over-relaxation
   * slows the iteration down (60 instead of 38 sweeps on 8 slabs at 1.15). */
  double polar_sor;
  /* rccl_halo yes|no (extension keyword, one MPI rank per GPU on one node): the LAMMPS shim hands the per-sweep exchange of
   * the halo dipoles and the all-reduced stop rule to the library's own RCCL driver (polar_dist_step) instead of staging
   * them through atom->mu_induced and Comm::forward_comm_pair -- the capability PS.h:51-52 / PS.cpp:1320-1362
   * (pack_comm / unpack_comm, never called) was meant to provide.  Off by default. */
  int rccl_halo;
  /* polar_accel <m> (extension keyword, list mode with polar_gs / polar_gs_ranked; 0 = off, the default): Anderson mixing of
   * depth m (1 .. POLAR_ACCEL_MAX) on the sweep map.  With G(mu) = one Gauss-Seidel sweep (PS.cpp:1158-1180) and residual
   * r_k = G(mu_k) - mu_k, the next iterate is G(mu_k) minus the combination of the last m differences of G that minimises
   * |r_k - sum gamma_j (r_{k-j+1} - r_{k-j})|: the same fixed point, the same stop rule on |G(mu_k) - mu_k| (PS.cpp:1194-1210),
   * fewer sweeps.  Per sweep it adds m dot products over 3N doubles; across ranks they ride the stop rule's all-reduce. */
  int polar_accel;
} polar_settings;
#define POLAR_ACCEL_MAX 8

typedef struct {
  double eng_vdwl, eng_coul, eng_pol; /* Pair::eng_vdwl/eng_coul/eng_pol (src/pair.h:36) */
  double u_self, u_ef, u_dd;          /* the three parts of eng_pol (PS.cpp:632) */
  double virial[6];                   /* Pair::virial when vflag asks for it */
  double rmin;                        /* PS.cpp:196-209 */
  double rms_dmu;                     /* sqrt(sum(dmu^2)/(3N)) of the last sweep */
  int iterations;                     /* return value of DipoleSolverIterative (PS.cpp:1113) */
  int sweeps;                         /* sweeps actually executed on the device */
  int status;                         /* POLAR_OK or POLAR_WARN_NOT_CONVERGED */
  int ncolors;                        /* colour phases per GS sweep in cutoff mode (0 otherwise) */
  /* device time of the last compute, milliseconds (HIP events on the compute stream) */
  double ms_total, ms_rank, ms_ljcoul, ms_static, ms_solve, ms_force, ms_list;
  long long dd_pairs;                 /* entries of the dipole-dipole list swept per sweep */
  double ms_color_host;               /* host wall time of the colour-phase rebuild (conflict graph + DSATUR) when this
                                         step rebuilt it (reneighbor steps in list mode), else 0 */
} polar_result;

/* ---- lifetime --------------------------------------------------------------------------- */
int polar_create(int device, polar_handle **out);
int polar_destroy(polar_handle *h);
const char *polar_last_error(const polar_handle *h);
const char *polar_last_warning(const polar_handle *h);
int polar_device_count(void); /* 0 when no GPU is usable; never throws */
/* identifies the kernel generation of the built library (bench.py matches PMC traffic files in profiles/ against it) */
const char *polar_kernel_version(void);

/* ---- host mirror of the Pair text interface (same grammar, defaults and error strings) --- */
/* PairLJCutCoulLongPolarization::settings, PS.cpp:678-766. argv = pair_style args after the name. */
int polar_pair_settings(polar_handle *h, int narg, const char *const *arg);
/* ::coeff, PS.cpp:772-800.  arg = "I J epsilon sigma [cut_lj]"; ntypes = atom->ntypes. */
int polar_pair_coeff(polar_handle *h, int ntypes, int narg, const char *const *arg);
/* Pair::modify_params subset used with this style (src/pair.cpp:125-185): mix, shift, table, tabinner. */
int polar_pair_modify(polar_handle *h, int narg, const char *const *arg);
/* Pair::init + ::init_style + ::init_one for all pairs (src/pair.cpp:189-263, PS.cpp:806-921): mixing, lj1..lj4,
 * offset, cutsq, uploaded to the device.  g_ewald = force->kspace->g_ewald (PS.cpp:847), qqrd2e/special_* = Force
 * members.  The Coulomb lookup tables of Pair::init_tables (src/pair.cpp:313-520, called at PS.cpp:851) stay LAMMPS
 * host code: unless `pair_modify table 0` is in force, hand them over with polar_set_coul afterwards --
 * compute entry points and polar_pair_single refuse to run without them. */
int polar_pair_init(polar_handle *h, double g_ewald, double qqrd2e, const double special_lj[4],
                    const double special_coul[4]);
/* ::init_one return value (cutoff) for a type pair, after polar_pair_init. */
double polar_pair_cut(const polar_handle *h, int itype, int jtype);
/* ::single, PS.cpp:1035-1097 (LJ + real-space Coulomb only).  Host arithmetic.  Returns NaN (message in
 * polar_last_error) when the Coulomb tables it needs were not handed over. */
double polar_pair_single(const polar_handle *h, double qi, double qj, int itype, int jtype, double rsq,
                         double factor_coul, double factor_lj, double *fforce);
/* ::extract, PS.cpp:1101-1109: "cut_coul" (dim 0), "epsilon"/"sigma" (dim 2, [(n+1)*(n+1)] row-major). */
const void *polar_pair_extract(const polar_handle *h, const char *name, int *dim);
int polar_get_settings(const polar_handle *h, polar_settings *out);

/* Restart persistence of the polarization keywords (the reference persists only the stock
 * lj/cut/coul/long fields, PS.cpp:976-1009, so after read_restart its defaults apply).  polar_restart_pack writes a
 * self-describing record -- int32 magic 'POLR', int32 version, int32 payload bytes, payload = precision, damp, gamma,
 * dd_cutoff (doubles), max_iterations, damp_type, zodid, fixed_iteration, polar_gs, polar_gs_ranked, use_previous, debug,
 * device_neigh, restart_polar (int32) -- into buf and returns its length (with buf == NULL: the length needed).
 * polar_restart_unpack applies such a record (cut_lj / cut_coul stay as the stock record set them) and returns
 * POLAR_ERR_INPUT, changing nothing, when buf does not start with one: a reader can probe the stream and seek back. */
int polar_restart_pack(const polar_handle *h, void *buf, int max_bytes);
int polar_restart_unpack(polar_handle *h, const void *buf, int nbytes);
#define POLAR_RESTART_HEADER_BYTES 12 /* magic, version, payload length */

/* ---- raw setters for a LAMMPS shim that keeps LAMMPS' own tables ------------------------- */
int polar_set_settings(polar_handle *h, const polar_settings *s);
/* tables are [(ntypes+1)*(ntypes+1)] row-major as lj1[i][j] (PS.h:62) */
int polar_set_types(polar_handle *h, int ntypes, const double *lj1, const double *lj2, const double *lj3,
                    const double *lj4, const double *offset, const double *cut_ljsq, const double *cutsq);
/* Pair::{rtable,drtable,ftable,dftable,ctable,dctable,etable,detable} with Pair::ncoultablebits, ncoulmask,
 * ncoulshiftbits, tabinnersq as Pair::init_tables left them (src/pair.h:205-216); ncoultablebits == 0: no table.
 * A host copy serves polar_pair_single. */
int polar_set_coul(polar_handle *h, double g_ewald, double qqrd2e, const double special_lj[4],
                   const double special_coul[4], int ncoultablebits, int ncoulmask, int ncoulshiftbits,
                   double tabinnersq, const double *rtable, const double *drtable, const double *ftable,
                   const double *dftable, const double *ctable, const double *dctable, const double *etable,
                   const double *detable);

/* ---- per-run / per-step data (what compute() reads through atom->, domain->, list->) ----- */
/* Domain: boxlo, prd, tilt (xy,xz,yz), periodicity, triclinic (domain.cpp:1220-1312 inputs) */
int polar_set_box(polar_handle *h, const double boxlo[3], const double prd[3], const double tilt[3],
                  const int periodic[3], int triclinic);
/* atom->x,q,static_polarizability,type,molecule for nlocal+nghost atoms (src/atom.h:160-163) */
int polar_set_atoms(polar_handle *h, int nlocal, int nghost, const double *x, const double *q,
                    const double *alpha, const int *type, const int *molecule);
/* positions only, between two neighbor-list builds (neighbor->ago != 0): LAMMPS neither reorders nor exchanges atoms on such
 * steps, so q, static_polarizability, type and molecule on the device still hold (what PS.cpp:125-188 re-reads through atom->
 * every step is, on these steps, only x).  Same nlocal / nghost as the last polar_set_atoms, or POLAR_ERR_INPUT. */
int polar_set_positions(polar_handle *h, int nlocal, int nghost, const double *x);
/* positions of the atoms [lo, hi) only (x = [hi-lo][3]): a rank of a multi-GPU run uploads its own atoms and lets
 * polar_dist_positions fetch the rest from their owners */
int polar_set_positions_range(polar_handle *h, int lo, int hi, const double *x);
/* NeighList inum/ilist/numneigh/firstneigh (src/neigh_list.h:46-50); call when neighbor->ago == 0.  The rows are validated and
 * copied into pinned memory before the call returns (the caller's arrays may change afterwards); the transfer to the device
 * finishes under the calls that follow -- the LJ/Coulomb loop of the next polar_compute waits for it. */
int polar_set_neighbors(polar_handle *h, int inum, const int *ilist, const int *numneigh,
                        int *const *firstneigh);
/* same list already flattened: firstneigh[i] = offset of atom i's entries in neigh[] */
int polar_set_neighbors_csr(polar_handle *h, int inum, const int *ilist, const int *numneigh,
                            const long long *firstneigh, const int *neigh);

/* force->newton_pair (src/force.h; read by the reference at PS.cpp:293 `if (newton_pair || j < nlocal)` and by
 * Pair::ev_tally, src/pair.cpp:854-950).  1 (default): half list with newton on -- every pair once, -F deposited on j,
 * ghost forces returned for reverse_comm.  0: LAMMPS' newton-off half list -- a pair of a local atom with a ghost is
 * listed by the local atom (both owners list it), ghosts receive no force, and such a pair tallies half its energy and
 * virial (ev_tally's 0.5 per local atom).  Call before polar_set_neighbors*. */
int polar_set_newton(polar_handle *h, int newton_pair);

/* Device-side neighbor build for the LJ + Ewald-real loop, INSTEAD of polar_set_neighbors when
 * neighbor->ago == 0.  Replaces what Neighbor builds for this style
 * (src/neighbor.cpp, src/npair_half_bin_newton.cpp) with the rules of NPair::exclusion() and
 * NPair::find_special() (src/npair.cpp): pairs with rsq <= cutneighsq[itype][jtype] among the atoms of
 * the last polar_set_atoms (locals + ghosts); same-molecule pairs dropped when exclude_molecule_intra
 * (neigh_modify exclude molecule/intra all); special partners dropped / kept plain / kept with their
 * 1-2,1-3,1-4 code in bits 30-31 according to special_flag[1..3] = neighbor->special_flag
 * (src/neighbor.cpp: 0, 1, 2), except images beyond half a box (Domain::minimum_image_check).
 * The result is a FULL list for the local rows (newton off): such steps leave no force on ghosts,
 * and the LJ/Coulomb part of a vflag = 2 virial is tallied pairwise (the same number as LAMMPS'
 * fdotr over locals + ghosts) -- a shim using this entry point sets Pair::no_virial_fdotr_compute = 1 (src/pair.h:53).
 *   cutneighsq[(ntypes+1)^2]  neighbor->cutneighsq (cut_ij + skin)^2, 1-based types
 *   tag[nall]                 atom->tag (may be NULL when special is NULL)
 *   nspecial[nlocal][3], special[nlocal][maxspecial]   atom->nspecial (cumulative counts) and
 *                             atom->special (partner tags); both NULL for systems without bonds */
int polar_build_neighbors(polar_handle *h, const double *cutneighsq, const int *tag, const int *nspecial,
                          const int *special, int maxspecial, const int special_flag[4],
                          int exclude_molecule_intra);

/* ---- the hot path: PairLJCutCoulLongPolarization::compute(eflag,vflag), PS.cpp:125-645 ---- */
/* f[nall][3] is ACCUMULATED (+=) like atom->f; mu[nlocal][3] is atom->mu_induced (read when
 * use_previous, always written); ef_static[nlocal][3] (may be NULL) receives atom->ef_static.
 * eflag: 0 / 1 (global).  vflag: 0, 1 (pairwise global virial), 2 (fdotr global virial).
 * Per-atom flags (eflag & 2, vflag & 4) -> POLAR_ERR_INPUT here: use polar_compute_peratom. */
int polar_compute(polar_handle *h, int eflag, int vflag, double *f, double *mu, double *ef_static,
                  polar_result *out);
/* The same step with LAMMPS' per-atom tallies (ev_setup: eflag_atom = eflag/2, vflag_atom = vflag/4,
 * src/pair.cpp:760-764).  eatom[nall] (Pair::eatom) and vatom[nall][6] (Pair::vatom) are ACCUMULATED
 * (+=) for locals and ghosts exactly as ev_tally (src/pair.cpp:881-885, 925-942) and ev_tally_xyz
 * (src/pair.cpp:1065-1082) fill them: 1/2 of each LJ/Coulomb pair energy and of each pair virial to
 * both atoms; the polarization pairs add to vatom only (PS.cpp:629 passes zero energies).
 * eatom may be NULL unless eflag & 2, vatom may be NULL unless vflag & 4. */
int polar_compute_peratom(polar_handle *h, int eflag, int vflag, double *f, double *mu, double *ef_static,
                          double *eatom, double *vatom, polar_result *out);

/* `debug yes` (PS.cpp:1182-1191): u_polar = -1/2 sum E_static . mu after every sweep of the last solve, in the
 * units the library works in (the reference prints it times 22.432653052265^2).  Writes at most `max` values and
 * returns how many sweeps were recorded (0 unless the debug keyword is on), < 0 on error. */
int polar_get_debug_trace(polar_handle *h, double *u_polar, int max);

/* `debug yes`, the two force lines of PS.cpp:637-638: out6 = {polarization force on the caller's atom 0 (PS.cpp:612-626),
 * its dipole-dipole part (PS.cpp:542-556, 583-597)} of the last compute, in the units of `f`.  Returns 1 when the numbers
 * are there, 0 (and zeros) without the debug keyword or when atom 0 is not a row of this handle, < 0 on error. */
int polar_get_debug_forces(polar_handle *h, double *out6);

/* Diagnostics (no reference counterpart: the reference's sweep is serial, PS.cpp:1158-1180): the colour phase of every
 * local atom in the list-mode Gauss-Seidel of the last compute, in the caller's atom order; -1 for atoms that are not rows
 * (not polarizable, or outside the handle's row range).  Atoms of one colour are relaxed by one launch and must lie farther
 * apart than the colour distance (tests check exactly that).  Returns the number of colours, < 0 on error. */
int polar_get_colors(polar_handle *h, int *color, int n);

/* The colour phases handed in by the caller instead of built by the library (no reference counterpart): color[nlocal] in the
 * caller's atom order, 0 .. 63 for polarizable atoms, -1 for the others; phases run in the order of the colour numbers.  Atoms
 * of one colour must lie farther apart than the colour distance.  On a sharded handle the halo rows carry the colours their
 * owners gave them: all ranks then sweep ONE colouring (polar_dist_set_schedule).  If atoms of one colour later come closer than
 * the library tolerates, its own colouring takes over.  n = 0 withdraws the colours. */
int polar_set_colors(polar_handle *h, const int *color, int n);

/* ---- device-resident variant (bench, multi-GPU driver): no host<->device traffic ---------- */
/* Runs compute() on the atoms/lists already resident from polar_set_*; results stay on the
 * device (polar_dev_ptr) and only the scalars in polar_result come back. */
int polar_compute_resident(polar_handle *h, int eflag, int vflag, polar_result *out);
/* device pointers: "f" [nall][3], "mu" [nlocal][3], "ef_static" [nlocal][3], "x" [nall][3],
 * "eatom" [nall], "vatom" [nall][6] (the last two are valid after a step run with eflag&2 / vflag&4) */
void *polar_dev_ptr(polar_handle *h, const char *name);
/* copy a device-resident per-atom array back: name as above, n doubles */
int polar_download(polar_handle *h, const char *name, double *dst, long long n);
/* overwrite the device-resident dipoles (multi-GPU halo, use_previous across steps) */
int polar_upload_mu(polar_handle *h, const double *mu, long long n);


/* ---- stepwise / sharded interface (multi-GPU driver: one process per GPU, rows sharded) ------
 * The reference is single-process only (README.md:5; its pack_comm/unpack_comm are dead code,
 * PS.h:51-52).  With dd_cutoff > 0 the path shards by rows: every rank holds all atoms, owns the
 * rows [lo,hi), and exchanges dipoles between sweeps (RCCL, driven from parallel.py).          */
/* run all kernels on this HIP stream (e.g. torch's current stream) instead of the private one */
int polar_set_stream(polar_handle *h, void *hip_stream);
int polar_set_row_range(polar_handle *h, int lo, int hi); /* hi < 0: all rows */
/* the N of the stop rule sum(dmu^2)/(3N) <= precision^2 (PS.cpp:1194-1210) when the handle holds only a
 * part of the system (own + halo atoms): the global atom count, identical on every rank; 0 = nlocal */
int polar_set_global_count(polar_handle *h, long long natoms);
/* 1: the LJ/coul list is a LAMMPS *full* list (each pair in both rows): force on i only, tallies halved */
int polar_set_list_style(polar_handle *h, int full);
/* compute() split at the exchange points: begin = list build, LJ+coul, static field, initial guess.
 * eflag & 2 / vflag & 4 ask for the per-atom tallies as polar_compute_peratom does; they stay on the device
 * ("eatom" [nall], "vatom" [nall][6] of polar_dev_ptr / polar_download, valid after polar_step_finish) and hold what THIS
 * handle's rows tallied: the shards' arrays add up to the unsharded ones. */
int polar_step_begin(polar_handle *h, int eflag, int vflag);
int polar_step_sweep(polar_handle *h); /* one sweep over the owned rows */
/* one colour phase of the list-mode Gauss-Seidel sweep: part 0 = all rows of colour `color`, 1 = its boundary rows (the rows a
 * peer receives, polar_dist_set_halo; all rows on a handle without boundary flags), 2 = its interior rows.  Phases 0 .. ncolors-1
 * in order, parts 1 then 2 (or 0), equal polar_step_sweep. */
int polar_step_sweep_phase(polar_handle *h, int color, int part);
/* one sweep in `nparts` pieces: piece `part` runs its share of the colour phases (colour-phase Gauss-Seidel of list mode only),
 * so that a driver can exchange halo dipoles INSIDE a sweep -- neighbours then see the phases already done one piece, not one
 * sweep, late.  Calling all pieces 0 .. nparts-1 in order equals polar_step_sweep. */
int polar_step_sweep_part(polar_handle *h, int part, int nparts);
/* end-of-sweep control (PS.cpp:1193-1236) on the device; dev_global_change = all-reduced sum of
 * (dmu)^2 in device memory, or NULL to use this handle's own sum */
int polar_step_sweep_end(polar_handle *h, const double *dev_global_change);
/* the end-of-sweep logic of `count` sweeps at once: fixed-iteration Gauss-Seidel takes no decision between
 * sweeps, so a driver may call this after the last-but-one sweep (count = sweeps so far) and after the last
 * (count = 1) instead of once per sweep */
int polar_step_sweep_end_n(polar_handle *h, const double *dev_global_change, int count);
int polar_step_state(polar_handle *h, int *done, int *iterations, int *status); /* synchronises */
int polar_step_finish(polar_handle *h, polar_result *out); /* forces/energies of the owned rows */
/* dipoles of rows [lo,hi) <-> a packed device buffer [(hi-lo)][3] */
int polar_mu_gather(polar_handle *h, long long lo, long long hi, double *dev_dst);
int polar_mu_scatter(polar_handle *h, long long lo, long long hi, const double *dev_src);
/* halo form: dipoles of the atoms listed in dev_idx (orig ids, device memory; negative = padding) <->
 * packed buffer [n][3]; the scatter skips padding and rows this handle owns */
int polar_mu_gather_idx(polar_handle *h, const int *dev_idx, long long n, double *dev_dst);
int polar_mu_scatter_idx(polar_handle *h, const int *dev_idx, long long n, const double *dev_src);
int polar_change_export(polar_handle *h, double *dev_dst); /* this handle's running sum of (dmu)^2 */
/* Host-pointer forms of the exchange, for a host whose communication layer works on host arrays
 * (LAMMPS: Comm::forward_comm_pair -> Pair::pack_forward_comm / unpack_forward_comm, src/pair.h:165-166,
 * one MPI rank per GPU; lammps_shim/ uses them):
 *   polar_step_mu_get      dipoles of the rows [lo,hi) -> mu_host [(hi-lo)][3]      (synchronises)
 *   polar_step_mu_put_idx  mu_host [n][3] -> the atoms idx[0..n) (handle-local indices, host array)
 *   polar_step_change_get  this handle's sum of (dmu)^2 of the sweep -> *sum        (synchronises)
 *   polar_step_sweep_end_host  end-of-sweep control with the globally reduced sum passed by value */
int polar_step_mu_get(polar_handle *h, long long lo, long long hi, double *mu_host);
int polar_step_mu_put_idx(polar_handle *h, long long n, const int *idx_host, const double *mu_host);
int polar_step_change_get(polar_handle *h, double *sum);
int polar_step_sweep_end_host(polar_handle *h, double global_change);

/* ---- multi-GPU driver inside the library: one process (or MPI rank) per GPU, RCCL over xGMI -------------------------
 * The capability the reference declares and never wires up (pack_comm / unpack_comm, PS.h:51-52, PS.cpp:1320-1362; the
 * pair style is single-process, README.md:5).  A rank's handle holds [own atoms | halo atoms | ghost images] with
 * polar_set_row_range = the own rows; the driver runs a whole Pair::compute for it: per sweep, on the handle's stream,
 *     pack kernel -> ncclGroupStart, ncclRecv + ncclSend per peer, ncclGroupEnd -> unpack kernel
 * and one ncclAllReduce of the stop rule's double every `reduce_every` sweeps; the host reads the device-resident loop
 * state every `check_every` sweeps.  Energies, virial and pair counts of the result are summed over the ranks; a pitch
 * overflow on any rank makes all ranks repeat the step.  RCCL is opened at run time (dlopen): the library has no link-time
 * dependency on it, and a process that already holds a copy (PyTorch's) shares it.
 *   polar_dist_unique_id   rank 0: the 128-byte ncclUniqueId, to be broadcast by the caller (MPI_Bcast, a file, a store)
 *   polar_dist_create      ncclCommInitRank on `device` (collective over the ranks)
 *   polar_dist_set_halo    this rank's exchange plan for handle h: for peer k (a rank; the rank itself is allowed) the
 *                          handle-local indices of the atoms whose dipoles it sends, and of the atoms it receives dipoles for,
 *                          in the order the peer sends them (concatenated over the peers).  The rows it sends become the
 *                          handle's BOUNDARY rows: they are swept first in their colour phase.
 *   polar_dist_set_cadence sweeps per all-reduce of the stop rule (1 = the reference's rule after every sweep) and
 *                          sweeps per look at the loop state
 *   polar_dist_set_schedule  how the Gauss-Seidel solve exchanges.  nclasses > 0: the ranks build ONE colouring together -- they
 *                          take turns by class (my_class in 0 .. nclasses-1; no two peers may share a class), a rank colours its
 *                          rows against the colours its peers' rows already hold -- so that the colour phases of all ranks
 *                          together are the single-GPU iteration; after phase c only colour c's boundary rows travel, on a
 *                          second stream, and a phase waits for the exchange issued lag + 1 phases earlier: lag 0 = the
 *                          single-GPU iterates exactly (the exchange hides behind the interior rows of its own phase only),
 *                          lag 1 (default) = a halo dipole may be one phase old, every exchange has a whole phase to hide
 *                          behind; lag -1 or nclasses 0 = one exchange of all halo dipoles per sweep, every rank colouring for
 *                          itself (block-Jacobi across the ranks: more sweeps).  A colouring handed in with polar_set_colors
 *                          (own AND halo rows) counts as shared.
 *   polar_dist_set_ghosts  the periodic images the handle holds behind its local atoms: owner (handle-local index) and shift
 *                          of every ghost, for polar_dist_positions
 *   polar_dist_positions   per step, before polar_dist_step: the positions of the halo atoms from their owners (the same plan
 *                          as the dipoles, 24 B per halo atom) and of the ghost images from theirs -- "ghost x over RCCL";
 *                          the caller has uploaded its OWN atoms' new positions (polar_set_positions_range)
 *   polar_dist_step        one Pair::compute across the ranks (collective); forces, dipoles and per-atom arrays stay on the
 *                          device (polar_download / polar_dev_ptr).  out = energies, virial and pair count SUMMED over the
 *                          ranks; polar_dist_local_result = this rank's own share of the same step (what a host that sums
 *                          per-rank accumulators itself -- LAMMPS: compute pe, thermo, pressure -- must add).
 *                          A rank that fails before the first exchange (bad input, allocation) makes EVERY rank return an
 *                          error: the begin status is max-reduced first.
 *                          A wait for the device that sees no progress for POLAR_DIST_TIMEOUT_S (180 s; or an asynchronous
 *                          RCCL error) aborts the communicator and ends the step with an error: a peer that is not answering an
 *                          exchange or an all-reduce is reported, not waited for for ever.
 *   polar_dist_exchange    one dipole exchange by itself (tests)
 *   polar_dist_profile     enable != 0: the following polar_dist_step calls put timed events between the parts of their sweep
 *                          loop (a few us each: not for a timed region); polar_dist_profile_get = the last such step's device
 *                          time by part, in ms[POLAR_DIST_PROF_PARTS]: [0] other (waiting for an exchange issued on the
 *                          communication stream, host looks at the loop state), [1] sweep kernels, [2] stop rule (fold,
 *                          all-reduce, end-of-sweep logic), [3] exchanges on the compute stream (pack, send / receive, unpack),
 *                          [4] polar_accel's mixing kernels -- the figures a scaling model of the schedule is calibrated on
 *   polar_dist_comm_count  ncclCommCount of the communicator (the number of ranks RCCL itself sees) */
#define POLAR_DIST_ID_BYTES 128
typedef struct polar_dist polar_dist;
int polar_dist_unique_id(void *id128);
int polar_dist_create(const void *id128, int rank, int nranks, int device, polar_dist **out);
int polar_dist_destroy(polar_dist *d);
const char *polar_dist_last_error(const polar_dist *d);
int polar_dist_comm_count(const polar_dist *d);
int polar_dist_set_cadence(polar_dist *d, int reduce_every, int check_every);
int polar_dist_set_schedule(polar_dist *d, int lag, int my_class, int nclasses);
int polar_dist_set_halo(polar_dist *d, polar_handle *h, int npeers, const int *peers, const int *send_count, const int *send_idx,
                        const int *recv_count, const int *recv_idx);
int polar_dist_set_ghosts(polar_dist *d, polar_handle *h, int nghost, const int *owner, const double *shift);
int polar_dist_positions(polar_dist *d, polar_handle *h);
int polar_dist_exchange(polar_dist *d, polar_handle *h);
int polar_dist_step(polar_dist *d, polar_handle *h, int eflag, int vflag, polar_result *out);
int polar_dist_local_result(const polar_dist *d, polar_result *out);
int polar_dist_counters(const polar_dist *d, int *exchanges, int *allreduces);
#define POLAR_DIST_PROF_PARTS 5
int polar_dist_profile(polar_dist *d, int enable);
int polar_dist_profile_get(const polar_dist *d, double *ms_by_part, int *intervals);

#ifdef __cplusplus
}
#endif
#endif
