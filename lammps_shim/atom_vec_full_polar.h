/* -*- c++ -*- ----------------------------------------------------------
   Atom style support for the polarization pair style.

   The reference declares the per-atom polarization attributes (src/atom.h:160-163:
   static_polarizability, ef_static, mu_induced, static_polarizability_flag) and uses them in
   `set`, `thermo`, `compute pe` and the pair style, but the checkout's atom_style full
   (src/MOLECULE/atom_vec_full.cpp:52-130) never allocates or communicates them: the author's
   patched file was lost to a stale .gitignore.  This class supplies that support WITHOUT touching the
   stock file: it derives from AtomVecFull and adds the three arrays to every place an atom style
   owns per-atom data (grow, copy, border, exchange, restart, create/data_atom, memory_usage), plus
   names for `compute property/atom`.

   Registration: under the name `full`, so decks keep `atom_style full`.  The stock header's line
   `AtomStyle(full,AtomVecFull)` (src/MOLECULE/atom_vec_full.h:16) must be renamed (e.g. to
   `full/plain`) or removed when this file is added -- the one edit INTEGRATION.md lists.
------------------------------------------------------------------------- */

#ifdef ATOM_CLASS

AtomStyle(full,AtomVecFullPolar)

#else

#ifndef LMP_ATOM_VEC_FULL_POLAR_H
#define LMP_ATOM_VEC_FULL_POLAR_H

#include "atom_vec_full.h"

namespace LAMMPS_NS {

class AtomVecFullPolar : public AtomVecFull {
 public:
  AtomVecFullPolar(class LAMMPS *);
  virtual ~AtomVecFullPolar() {}
  void grow(int);
  void grow_reset();
  void copy(int, int, int);
  virtual int pack_border(int, int *, double *, int, int *);
  virtual int pack_border_vel(int, int *, double *, int, int *);
  virtual void unpack_border(int, int, double *);
  virtual void unpack_border_vel(int, int, double *);
  virtual int pack_exchange(int, double *);
  virtual int unpack_exchange(double *);
  int size_restart();
  int pack_restart(int, double *);
  int unpack_restart(double *);
  void create_atom(int, double *);
  void data_atom(double *, imageint, char **);
  int property_atom(char *);
  void pack_property_atom(int, double *, int, int);
  bigint memory_usage();

 protected:
  double *static_polarizability;
  double **ef_static, **mu_induced;
  void grow_polar();
  void clear_polar(int);
};

}

#endif
#endif
