/* polar_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C CPU restatement of the reference pair style
 *   lj/cut/coul/long/polarization
 * (reference: src/pair_lj_cut_coul_long_polarization.cpp, abbreviated PS.cpp below,
 *  src/domain.cpp Domain::closest_image, src/pair.cpp init_tables / ev_tally /
 *  virial_fdotr_compute).  Written from scratch following the reference's algorithm
 *  and quirks line by line; each function cites the file:line range it restates.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use
 * this library, and only as the checker.  The product path (libpolar_mi355x.so,
 * hand-written HIP) never links or calls it.
 *
 * Parity pin: the oracle is pinned against
 *   (1) the reference's own logs (polarization/examples/Bulk H2/log.lammps and
 *       MOF5+Methane/log.lammps: E_pol, E_vdwl, E_coul at step 0), and
 *   (2) per-atom forces / energies produced by the reference's unmodified
 *       PS.cpp compiled into oracle/_ref (see oracle/Makefile, oracle/ref_seam/).
 * See tests/test_oracle_golden.py.
 */
#ifndef POLAR_ORACLE_H
#define POLAR_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

enum { ORC_DAMP_EXPONENTIAL = 0, ORC_DAMP_NONE = 1 }; /* PS.cpp:51 */

typedef struct {
  /* ---- atoms (LAMMPS layout: locals first, then ghosts) ---- */
  int nlocal, nghost;
  const double *x;        /* [nall][3] */
  const double *q;        /* [nall] */
  const double *alpha;    /* [nall]  atom->static_polarizability */
  const int *type;        /* [nall]  1-based */
  const int *molecule;    /* [nall] */
  /* ---- box: what Domain::closest_image reads (domain.cpp:1220-1312) ---- */
  double prd[3];          /* xprd,yprd,zprd */
  double tilt[3];         /* xy,xz,yz */
  int periodic[3];
  int triclinic;
  /* ---- LJ / Coulomb parameters (PS.cpp:858-921 init_one outputs) ---- */
  int ntypes;             /* tables are [(ntypes+1)*(ntypes+1)], row-major, index 0 unused */
  const double *lj1, *lj2, *lj3, *lj4, *offset, *cut_ljsq, *cutsq;
  double cut_coul, g_ewald, qqrd2e;
  double special_lj[4], special_coul[4];
  int newton_pair;
  /* ---- Coulomb tables (pair.cpp:313-520); ncoultablebits==0 => no tables ---- */
  int ncoultablebits, ncoulmask, ncoulshiftbits;
  double tabinnersq;
  const double *rtable, *drtable, *ftable, *dftable, *ctable, *dctable, *etable, *detable;
  /* ---- half neighbor list, CSR-flattened (j carries special bits 30-31) ---- */
  int inum;
  const int *ilist;       /* [inum] */
  const int *numneigh;    /* [nlocal] indexed by atom i */
  const long long *firstneigh; /* [nlocal] offset of atom i's list in neigh[] */
  const int *neigh;
  /* ---- polarization settings (PS.cpp:65-78 defaults, 686-756 parser) ---- */
  int iterations_max, damping_type, zodid, fixed_iteration;
  int polar_gs, polar_gs_ranked, use_previous, debug;
  double polar_damp, polar_precision, polar_gamma;
  /* ---- extension (NOT in the reference): dipole-dipole cutoff.
   *      <= 0 : reference semantics (all minimum-image pairs, dense matrix)
   *      >  0 : truncate T_ij (a6/a7) and the dd force/energy (a8B) at rsq < dd_cut^2;
   *             all polarization loops then run over a cell-list neighbor list. ---- */
  double dd_cutoff;
} orc_system;

typedef struct {
  double eng_vdwl, eng_coul, eng_pol;
  double u_self, u_ef, u_dd;
  double virial[6];
  double rmin;
  int iterations;
  int status;       /* 0 ok, 1 = not converged -> mu = alpha*E fallback (PS.cpp:1227-1235) */
  double rms_dmu;   /* sqrt(change/(3N)) of the last sweep */
  double t_rank, t_ljcoul, t_static, t_matrix, t_solve, t_force; /* seconds */
  int sweeps;       /* number of sweeps actually executed */
  /* the reference's `debug yes` lines PS.cpp:637-638: polarization force on atom 0 (PS.cpp:617-626) and its dipole-dipole
   * part (PS.cpp:548-556, 589-597), accumulated with + for i == 0 and - for j == 0 */
  double force_atom0[3], dipole_force_atom0[3];
} orc_result;

/* a9: Domain::closest_image, domain.cpp:1220-1312 */
void orc_closest_image(const orc_system *s, const double *xi, const double *xj, double *xjimage);

/* pair.cpp:1676-1723 init_bitmap + pair.cpp:313-520 init_tables (cut_respa == NULL, no msm).
 * tables: 8 arrays of 2^nbits doubles in the order r,dr,f,df,c,dc,e,de. */
int orc_init_tables(double cut_coul, double g_ewald, double qqrd2e, double tabinner,
                    int ncoultablebits, int *ncoulmask, int *ncoulshiftbits,
                    double *tabinnersq, double *tables);

/* PS.cpp:858-921 init_one (mixing pair.cpp:660-690 when not set explicitly) for all type pairs.
 * eps,sig,cutlj: [(n+1)^2] with [i][j], i<=j filled where setflag!=0.
 * mix_flag: 0 geometric, 1 arithmetic, 2 sixthpower. Outputs [(n+1)^2] arrays. */
void orc_init_one_all(int ntypes, const int *setflag, double *eps, double *sig, double *cutlj,
                      int mix_flag, int offset_flag, double cut_coul,
                      double *lj1, double *lj2, double *lj3, double *lj4, double *offset,
                      double *cut_ljsq, double *cutsq);

/* a2: PS.cpp:192-227 */
void orc_rank_metric(const orc_system *s, double *rank_metric, double *rmin);
/* a3: PS.cpp:232-321 (+ ev_tally pair.cpp:854-950 when vflag_global==1) */
void orc_ljcoul(const orc_system *s, int eflag, int vflag_pairwise, double *f,
                double *eng_vdwl, double *eng_coul, double *virial, double *eatom,
                double *vatom); /* eatom/vatom may be NULL: ev_tally per-atom parts, pair.cpp:881-942 */
/* a4: PS.cpp:324-361 */
void orc_static_field(const orc_system *s, double *ef_static);
/* a6: PS.cpp:1243-1316 ; matrix is [3N][3N] row-major */
void orc_build_dipole_field_matrix(const orc_system *s, double *matrix);
/* a7: PS.cpp:1113-1238 ; returns iterations */
int orc_dipole_solver(const orc_system *s, const double *matrix, const double *ef_static,
                      const double *rank_metric, double *mu, orc_result *res, double *utrace);
/* a8: PS.cpp:406-641 */
void orc_polar_forces(const orc_system *s, int eflag, int vflag_pairwise, const double *mu,
                      double *f, orc_result *res, double *vatom /* may be NULL; pair.cpp:1065-1082 */);
/* a10: pair.cpp:1495-1540 */
void orc_virial_fdotr(const orc_system *s, const double *f, double *virial);

/* Whole compute(): PS.cpp:125-645.  f[nall][3] is accumulated (+=); mu[nlocal][3] in/out;
 * ef_static[nlocal][3] out (already scaled by sqrt(qqrd2e)); utrace (may be NULL) receives the
 * `debug` u_polar value after each sweep (PS.cpp:1182-1191, without the K conversion).
 * eflag: 0/1 global energy; vflag: 0 none, 1 pairwise global virial, 2 fdotr virial. */
int orc_compute(const orc_system *s, int eflag, int vflag, double *f, double *mu,
                double *ef_static, orc_result *res, double *utrace);
/* Same, plus the per-atom tallies of ev_tally / ev_tally_xyz: eflag & 2 -> eatom[nall] +=,
 * vflag & 4 -> vatom[nall][6] += (zeroed by the caller, as ev_setup does, pair.cpp:789-806). */
int orc_compute_peratom(const orc_system *s, int eflag, int vflag, double *f, double *mu,
                        double *ef_static, orc_result *res, double *utrace, double *eatom,
                        double *vatom);

#ifdef __cplusplus
}
#endif
#endif
