/* ----------------------------------------------------------------------
   AtomVecFullPolar: atom_style full + the per-atom polarization attributes.  See the header.

   Layout of the added data in the communication buffers (chosen so that the stock routines can be
   called unchanged):
     border    [alpha of the n atoms][the stock border block]      alpha first: the stock unpack may
                                                                   grow the arrays, so it runs before
                                                                   alpha is stored
     exchange  [stock record, buf[0] = its length][alpha, mu_x, mu_y, mu_z]   buf[0] re-written
     restart   same as exchange
   E_static is scratch of the pair style (recomputed every step, PS.cpp:329-361): allocated and
   copied, never communicated.  mu_induced travels with the atom (exchange, restart) because
   `use_previous yes` reads it as the initial guess of the next step (PS.cpp:376-386).
------------------------------------------------------------------------- */

#include <string.h>
#include "atom_vec_full_polar.h"
#include "atom.h"
#include "memory.h"

using namespace LAMMPS_NS;

/* ---------------------------------------------------------------------- */

AtomVecFullPolar::AtomVecFullPolar(LAMMPS *lmp) : AtomVecFull(lmp)
{
  // what the pair style and `set ... static_polarizability` test for (PS.cpp:812-813, src/set.cpp:178)
  atom->static_polarizability_flag = 1;
  size_border += 1;       // alpha
  size_data_atom += 0;    // the data file format is unchanged: alpha comes from `set`, as in the examples
  static_polarizability = NULL;
  ef_static = mu_induced = NULL;
}

/* ---------------------------------------------------------------------- */

void AtomVecFullPolar::grow_polar()
{
  static_polarizability = memory->grow(atom->static_polarizability,nmax,"atom:static_polarizability");
  ef_static = memory->grow(atom->ef_static,nmax,3,"atom:ef_static");
  mu_induced = memory->grow(atom->mu_induced,nmax,3,"atom:mu_induced");
}

void AtomVecFullPolar::clear_polar(int i)
{
  static_polarizability[i] = 0.0;
  ef_static[i][0] = ef_static[i][1] = ef_static[i][2] = 0.0;
  mu_induced[i][0] = mu_induced[i][1] = mu_induced[i][2] = 0.0;
}

void AtomVecFullPolar::grow(int n)
{
  const int old = nmax;
  AtomVecFull::grow(n);          // sets nmax = atom->nmax and grows the stock arrays
  grow_polar();
  for (int i = old; i < nmax; i++) clear_polar(i);   // use_previous reads mu before anything wrote it (PS.cpp:376)
}

void AtomVecFullPolar::grow_reset()
{
  AtomVecFull::grow_reset();
  static_polarizability = atom->static_polarizability;
  ef_static = atom->ef_static;
  mu_induced = atom->mu_induced;
}

void AtomVecFullPolar::copy(int i, int j, int delflag)
{
  static_polarizability[j] = static_polarizability[i];
  for (int k = 0; k < 3; k++) {
    ef_static[j][k] = ef_static[i][k];
    mu_induced[j][k] = mu_induced[i][k];
  }
  AtomVecFull::copy(i,j,delflag);
}

/* ---------------------------------------------------------------------- */

int AtomVecFullPolar::pack_border(int n, int *list, double *buf, int pbc_flag, int *pbc)
{
  for (int i = 0; i < n; i++) buf[i] = static_polarizability[list[i]];
  return n + AtomVecFull::pack_border(n,list,&buf[n],pbc_flag,pbc);
}

int AtomVecFullPolar::pack_border_vel(int n, int *list, double *buf, int pbc_flag, int *pbc)
{
  for (int i = 0; i < n; i++) buf[i] = static_polarizability[list[i]];
  return n + AtomVecFull::pack_border_vel(n,list,&buf[n],pbc_flag,pbc);
}

void AtomVecFullPolar::unpack_border(int n, int first, double *buf)
{
  AtomVecFull::unpack_border(n,first,&buf[n]);    // may grow(): the arrays below are valid afterwards
  for (int i = 0; i < n; i++) {
    static_polarizability[first+i] = buf[i];
    ef_static[first+i][0] = ef_static[first+i][1] = ef_static[first+i][2] = 0.0;
    mu_induced[first+i][0] = mu_induced[first+i][1] = mu_induced[first+i][2] = 0.0;  // ghosts: filled by forward_comm_pair
  }
}

void AtomVecFullPolar::unpack_border_vel(int n, int first, double *buf)
{
  AtomVecFull::unpack_border_vel(n,first,&buf[n]);
  for (int i = 0; i < n; i++) {
    static_polarizability[first+i] = buf[i];
    ef_static[first+i][0] = ef_static[first+i][1] = ef_static[first+i][2] = 0.0;
    mu_induced[first+i][0] = mu_induced[first+i][1] = mu_induced[first+i][2] = 0.0;
  }
}

/* ----------------------------------------------------------------------
   exchange / restart: the stock record first (it stores its own length in buf[0]), ours appended
------------------------------------------------------------------------- */

int AtomVecFullPolar::pack_exchange(int i, double *buf)
{
  int m = AtomVecFull::pack_exchange(i,buf);
  buf[m++] = static_polarizability[i];
  buf[m++] = mu_induced[i][0];
  buf[m++] = mu_induced[i][1];
  buf[m++] = mu_induced[i][2];
  buf[0] = m;
  return m;
}

int AtomVecFullPolar::unpack_exchange(double *buf)
{
  int m = AtomVecFull::unpack_exchange(buf);      // stores the atom at index nlocal and increments atom->nlocal
  const int i = atom->nlocal - 1;
  static_polarizability[i] = buf[m++];
  mu_induced[i][0] = buf[m++];
  mu_induced[i][1] = buf[m++];
  mu_induced[i][2] = buf[m++];
  ef_static[i][0] = ef_static[i][1] = ef_static[i][2] = 0.0;
  return m;
}

int AtomVecFullPolar::size_restart()
{
  return AtomVecFull::size_restart() + 4 * atom->nlocal;
}

int AtomVecFullPolar::pack_restart(int i, double *buf)
{
  int m = AtomVecFull::pack_restart(i,buf);
  buf[m++] = static_polarizability[i];
  buf[m++] = mu_induced[i][0];
  buf[m++] = mu_induced[i][1];
  buf[m++] = mu_induced[i][2];
  buf[0] = m;
  return m;
}

int AtomVecFullPolar::unpack_restart(double *buf)
{
  // the stock routine hands EVERYTHING between its own fields and buf[0] to the fixes' per-atom restart data
  // (atom->extra, src/MOLECULE/atom_vec_full.cpp:890-894): our four values must be outside that span while it runs
  const double total = buf[0];
  buf[0] = total - 4.0;
  int m = AtomVecFull::unpack_restart(buf);
  buf[0] = total;
  const int i = atom->nlocal - 1;
  static_polarizability[i] = buf[m++];
  mu_induced[i][0] = buf[m++];
  mu_induced[i][1] = buf[m++];
  mu_induced[i][2] = buf[m++];
  ef_static[i][0] = ef_static[i][1] = ef_static[i][2] = 0.0;
  return m;
}

/* ---------------------------------------------------------------------- */

void AtomVecFullPolar::create_atom(int itype, double *coord)
{
  AtomVecFull::create_atom(itype,coord);
  clear_polar(atom->nlocal - 1);
}

void AtomVecFullPolar::data_atom(double *coord, imageint imagetmp, char **values)
{
  AtomVecFull::data_atom(coord,imagetmp,values);
  clear_polar(atom->nlocal - 1);    // alpha is assigned by `set type T static_polarizability a` (src/set.cpp:719-721)
}

/* ----------------------------------------------------------------------
   names for compute property/atom (and through it dump custom): the reference exposes the solver's
   per-atom results only through `debug yes` prints
------------------------------------------------------------------------- */

int AtomVecFullPolar::property_atom(char *name)
{
  static const char *names[] = {"static_polarizability","mu_inducedx","mu_inducedy","mu_inducedz",
                                "ef_staticx","ef_staticy","ef_staticz"};
  for (int k = 0; k < 7; k++)
    if (strcmp(name,names[k]) == 0) return k;
  return -1;
}

void AtomVecFullPolar::pack_property_atom(int index, double *buf, int nvalues, int groupbit)
{
  const int nlocal = atom->nlocal;
  const int *mask = atom->mask;
  int n = 0;
  for (int i = 0; i < nlocal; i++) {
    double v = 0.0;
    if (mask[i] & groupbit) {
      if (index == 0) v = static_polarizability[i];
      else if (index <= 3) v = mu_induced[i][index-1];
      else v = ef_static[i][index-4];
    }
    buf[n] = v;
    n += nvalues;
  }
}

bigint AtomVecFullPolar::memory_usage()
{
  bigint bytes = AtomVecFull::memory_usage();
  if (atom->memcheck("static_polarizability")) bytes += memory->usage(static_polarizability,nmax);
  if (atom->memcheck("ef_static")) bytes += memory->usage(ef_static,nmax,3);
  if (atom->memcheck("mu_induced")) bytes += memory->usage(mu_induced,nmax,3);
  return bytes;
}
