/* atom_vec_harness.cpp -- TEST INFRASTRUCTURE (build container only; needs /root/reference, never shipped).
 *
 * Runs lammps_shim/atom_vec_full_polar.cpp for real on top of the reference's own AtomVec / AtomVecFull (src/atom_vec.cpp,
 * src/MOLECULE/atom_vec_full.cpp, compiled from where they lie): grow, copy, border and exchange buffers, restart records
 * (with and without fixes' per-atom restart data behind the stock fields), create_atom, property names.  Every routine of the
 * derived class is checked by a round trip: what was packed for an atom must come back for it, stock fields included. */
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include <mpi.h>
#include "lammps.h"
#include "atom.h"
#include "comm.h"
#include "domain.h"
#include "error.h"
#include "memory.h"
#include "atom_vec_full_polar.h"

using namespace LAMMPS_NS;

namespace {
struct HErr { std::string msg; };
template <class T> T *blank() { return (T *)calloc(1, sizeof(T) + 64); }
int g_fail = 0;
std::string g_log;
void expect(bool ok, const char *what) { if (!ok) { g_fail++; g_log += what; g_log += "; "; } }
}  // namespace

void Error::all(const char *, int, const char *str) { throw HErr{str}; }
void Error::one(const char *, int, const char *str) { throw HErr{str}; }
int Atom::memcheck(const char *) { return 1; }

static void fill(Atom *a, int i, int seed) {
  for (int k = 0; k < 3; k++) { a->x[i][k] = seed + 0.1 * k; a->v[i][k] = -seed - 0.01 * k; a->f[i][k] = 0.0; }
  a->tag[i] = 100 + seed; a->type[i] = 1 + seed % 3; a->mask[i] = 1; a->image[i] = ((imageint)IMGMAX << IMG2BITS) | ((imageint)IMGMAX << IMGBITS) | IMGMAX;
  a->q[i] = 0.25 * seed; a->molecule[i] = 7 + seed;
  a->num_bond[i] = a->num_angle[i] = a->num_dihedral[i] = a->num_improper[i] = 0;
  a->nspecial[i][0] = a->nspecial[i][1] = a->nspecial[i][2] = 0;
  a->static_polarizability[i] = 1.5 + seed;
  for (int k = 0; k < 3; k++) { a->mu_induced[i][k] = 0.001 * seed + k; a->ef_static[i][k] = 9.0 + seed + k; }
}
static bool same_atom(Atom *a, int i, int seed, bool with_mu, bool with_v) {
  bool ok = a->tag[i] == 100 + seed && a->type[i] == 1 + seed % 3 && a->q[i] == 0.25 * seed && a->molecule[i] == 7 + seed &&
            a->static_polarizability[i] == 1.5 + seed;
  for (int k = 0; k < 3; k++) {
    ok = ok && a->x[i][k] == seed + 0.1 * k;
    if (with_v) ok = ok && a->v[i][k] == -seed - 0.01 * k;
    if (with_mu) ok = ok && a->mu_induced[i][k] == 0.001 * seed + k;
    else ok = ok && a->mu_induced[i][k] == 0.0;
  }
  return ok;
}

extern "C" int atomvec_check(char *msg, int nmsg) {
  g_fail = 0; g_log.clear();
  try {
    LAMMPS *lmp = blank<LAMMPS>();
    lmp->error = blank<Error>();
    lmp->atom = blank<Atom>();
    lmp->domain = blank<Domain>();
    lmp->comm = blank<Comm>();   // only its data members are read (nthreads)
    lmp->comm->nthreads = 1; lmp->comm->me = 0; lmp->comm->nprocs = 1;
    lmp->memory = new Memory(lmp);
    Atom *atom = lmp->atom;
    atom->nlocal = atom->nghost = 0; atom->nmax = 0;
    atom->bond_per_atom = atom->angle_per_atom = atom->dihedral_per_atom = atom->improper_per_atom = 1;
    atom->maxspecial = 1;
    AtomVecFullPolar *av = new AtomVecFullPolar(lmp);
    atom->avec = av;
    expect(atom->static_polarizability_flag == 1, "flag not set");
    expect(av->size_border == 9, "size_border != 8 + 1");
    av->grow(16);
    expect(atom->static_polarizability && atom->mu_induced && atom->ef_static, "arrays not allocated");
    expect(atom->nmax >= 16, "nmax");
    bool zero = true;
    for (int i = 0; i < atom->nmax; i++) zero = zero && atom->static_polarizability[i] == 0.0 && atom->mu_induced[i][2] == 0.0;
    expect(zero, "grown entries not cleared");
    for (int i = 0; i < 5; i++) fill(atom, i, i + 1);
    atom->nlocal = 5;
    /* copy */
    av->copy(1, 9, 0);
    expect(same_atom(atom, 9, 2, true, true) && atom->ef_static[9][1] == 9.0 + 2 + 1, "copy");
    /* exchange: atom 3 leaves, comes back as a new atom at index nlocal */
    std::vector<double> buf(4096, -777.0);
    int m = av->pack_exchange(3, buf.data());
    expect((int)buf[0] == m, "exchange buf[0]");
    int m2 = av->unpack_exchange(buf.data());
    expect(m2 == m && atom->nlocal == 6, "exchange length");
    expect(same_atom(atom, 5, 4, true, true), "exchange round trip");
    expect(atom->ef_static[5][0] == 0.0, "exchange: E_static is scratch, must arrive cleared");
    /* border: atoms {0, 2, 4} become ghosts behind the locals; alpha travels, mu does not */
    int list[3] = {0, 2, 4};
    int pbc[6] = {0, 0, 0, 0, 0, 0};
    std::fill(buf.begin(), buf.end(), -777.0);
    m = av->pack_border(3, list, buf.data(), 0, pbc);
    expect(m == 3 * av->size_border, "border length");
    av->unpack_border(3, atom->nlocal, buf.data());
    expect(same_atom(atom, 6, 1, false, false) && same_atom(atom, 7, 3, false, false) && same_atom(atom, 8, 5, false, false), "border round trip");
    std::fill(buf.begin(), buf.end(), -777.0);
    m = av->pack_border_vel(3, list, buf.data(), 0, pbc);
    expect(m == 3 * (av->size_border + av->size_velocity), "border_vel length");
    av->unpack_border_vel(3, atom->nlocal + 3, buf.data());
    expect(same_atom(atom, 9, 1, false, true) && same_atom(atom, 11, 5, false, true), "border_vel round trip");
    /* a border that forces the arrays to grow inside the stock unpack */
    {
      std::vector<int> big(40, 1);
      std::vector<double> b2(40 * 16, 0.0);
      m = av->pack_border(40, big.data(), b2.data(), 0, pbc);
      av->unpack_border(40, atom->nlocal, b2.data());
      expect(atom->nmax >= atom->nlocal + 40 && same_atom(atom, atom->nlocal + 39, 2, false, false), "border with grow");
    }
    /* restart: record of atom 2, read back as a new atom */
    std::fill(buf.begin(), buf.end(), -777.0);
    const int before = atom->nlocal;
    expect(av->size_restart() == before * (17 + 4), "size_restart");   // 17 stock values per bond-less atom + alpha, mu
    m = av->pack_restart(2, buf.data());
    expect((int)buf[0] == m && m == 21, "restart length");
    m2 = av->unpack_restart(buf.data());
    expect(m2 == m && atom->nlocal == before + 1 && same_atom(atom, before, 3, true, true), "restart round trip");
    /* the same record with two values of a fix's per-atom restart data between the stock fields and ours */
    {
      std::vector<double> b3(buf.begin(), buf.begin() + m);
      b3.insert(b3.begin() + (m - 4), 41.0);
      b3.insert(b3.begin() + (m - 3), 42.0);
      b3[0] = m + 2;
      atom->nextra_store = 2;
      lmp->memory->create(atom->extra, atom->nmax, 2, "extra");
      const int at = atom->nlocal;
      m2 = av->unpack_restart(b3.data());
      expect(m2 == m + 2 && same_atom(atom, at, 3, true, true), "restart with fix data: our values");
      expect(atom->extra[at][0] == 41.0 && atom->extra[at][1] == 42.0, "restart with fix data: the fix's values");
      expect(b3[0] == m + 2, "restart: buf[0] restored");
      atom->nextra_store = 0;
    }
    /* create_atom: cleared attributes */
    {
      double c[3] = {1, 2, 3};
      const int at = atom->nlocal;
      atom->static_polarizability[at] = 5.0; atom->mu_induced[at][1] = 5.0;
      av->create_atom(2, c);
      expect(atom->nlocal == at + 1 && atom->static_polarizability[at] == 0.0 && atom->mu_induced[at][1] == 0.0, "create_atom");
    }
    /* property names */
    {
      char n0[] = "static_polarizability", n1[] = "mu_inducedy", n2[] = "ef_staticz", n3[] = "nonsense";
      expect(av->property_atom(n0) == 0 && av->property_atom(n1) == 2 && av->property_atom(n2) == 6 && av->property_atom(n3) < 0, "property names");
      std::vector<double> pb(atom->nlocal, -1.0);
      av->pack_property_atom(0, pb.data(), 1, 1);
      expect(pb[1] == 1.5 + 2 && pb[4] == 1.5 + 5, "pack_property_atom alpha");
      av->pack_property_atom(2, pb.data(), 1, 1);
      expect(pb[2] == 0.001 * 3 + 1, "pack_property_atom mu_y");
    }
  } catch (HErr &e) {
    g_fail++; g_log += "LAMMPS error: " + e.msg;
  }
  snprintf(msg, nmsg, "%s", g_log.c_str());
  return g_fail;
}
