/* -*- c++ -*- ----------------------------------------------------------
   LAMMPS-side shim of the MI355X-native polarization pair style.

   Registers under the SAME style name as the reference's
   src/pair_lj_cut_coul_long_polarization.h:14-18, so input decks are unchanged:
       pair_style lj/cut/coul/long/polarization ...
   Replace the reference's two pair_lj_cut_coul_long_polarization.{h,cpp} files by this pair and
   link libpolar_mi355x.so.  All arithmetic of compute() runs in the library
   (hand-written HIP, include/polar_mi355x.h); this class only maps LAMMPS' object graph onto
   the C-ABI and the C-ABI's status codes onto error->all / error->warning.
------------------------------------------------------------------------- */

#ifdef PAIR_CLASS

PairStyle(lj/cut/coul/long/polarization,PairLJCutCoulLongPolarizationMI355X)

#else

#ifndef LMP_PAIR_LJ_CUT_COUL_LONG_POLARIZATION_MI355X_H
#define LMP_PAIR_LJ_CUT_COUL_LONG_POLARIZATION_MI355X_H

#include "pair.h"
#include <vector>
#include "polar_mi355x.h"


namespace LAMMPS_NS {

class PairLJCutCoulLongPolarizationMI355X : public Pair {
 public:
  PairLJCutCoulLongPolarizationMI355X(class LAMMPS *);
  virtual ~PairLJCutCoulLongPolarizationMI355X();
  virtual void compute(int, int);
  virtual void settings(int, char **);
  void coeff(int, char **);
  virtual void init_style();
  virtual double init_one(int, int);
  void write_restart(FILE *);
  void read_restart(FILE *);
  virtual void write_restart_settings(FILE *);
  virtual void read_restart_settings(FILE *);
  void write_data(FILE *);
  void write_data_all(FILE *);
  virtual double single(int, int, int, int, double, double, double, double &);
  virtual void *extract(const char *, int &);
  virtual int pack_forward_comm(int, int *, double *, int, int *);
  virtual void unpack_forward_comm(int, int, double *);

 protected:
  polar_handle *h;
  double cut_lj_global, cut_coul;
  double **epsilon, **sigma, **cut_lj;   // row-pointer views into the library's tables (extract())
  int pair_inited;
  int device_neigh;                      // extension keyword: list built by polar_build_neighbors
  int debug_flag;                        // keyword `debug yes`: the reference's prints (debug_prints)
  int atoms_sent, sent_nlocal, sent_nghost;   // what the last polar_set_atoms handed over (positions alone travel in between)
  std::vector<double> q_sent;                 // ... and the charges it carried (a fix may change atom->q between two list builds)
  // one MPI rank per GPU: library order = [own | halo (one ghost per foreign tag) | other ghosts]
  int nhalo, sh_n;
  std::vector<int> lib_of_lammps, lammps_of_lib, halo_ghost, sh_nn, sh_flat, sh_idx, sh_t, sh_t2, sh_m;
  std::vector<long long> sh_first;
  std::vector<double> sh_x, sh_q, sh_a, sh_f, sh_mu;
  // `rccl_halo yes`: the sweeps of a multi-rank step run inside the library (polar_dist_step, RCCL over xGMI)
  polar_dist *dist;
  int device_index;
  virtual void build_rccl_plan();   // (virtual: the host-path tests replace the MPI plan exchange)
  void compute_sharded(int, int);
  void build_halo_map();
  void exchange_dipoles();
  virtual void allocate();
  void check(int rc);                    // C-ABI status -> error->all / error->warning
  void debug_prints(const polar_result &res);
  void sync_views();
};

}

#endif
#endif
