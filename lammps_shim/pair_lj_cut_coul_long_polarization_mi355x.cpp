/* ----------------------------------------------------------------------
   LAMMPS shim of the MI355X-native lj/cut/coul/long/polarization pair style.
   See the header.  Reference counterparts are cited per function
   ("PS.cpp" = the reference's src/pair_lj_cut_coul_long_polarization.cpp).
------------------------------------------------------------------------- */

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <mpi.h>
#include <map>
#include <vector>
#include <algorithm>
#include "pair_lj_cut_coul_long_polarization_mi355x.h"
#include "atom.h"
#include "comm.h"
#include "domain.h"
#include "error.h"
#include "force.h"
#include "kspace.h"
#include "memory.h"
#include "neigh_list.h"
#include "neigh_request.h"
#include "neighbor.h"
#include "update.h"

#include "polar_mi355x.h"

using namespace LAMMPS_NS;

static_assert(sizeof(tagint) == sizeof(int), "the C-ABI takes 32-bit molecule ids (default LAMMPS_SMALLBIG build)");

/* ---------------------------------------------------------------------- */

PairLJCutCoulLongPolarizationMI355X::PairLJCutCoulLongPolarizationMI355X(LAMMPS *lmp) : Pair(lmp)
{
  // same flags as the reference constructor, PS.cpp:55-61
  ewaldflag = pppmflag = 1;
  respa_enable = 0;
  writedata = 1;
  ftable = NULL;
  epsilon = sigma = cut_lj = NULL;
  pair_inited = 0;
  device_neigh = 0;
  debug_flag = 0;
  atoms_sent = sent_nlocal = sent_nghost = 0;
  cut_lj_global = cut_coul = 0.0;
  h = NULL;
  // one MPI rank per GPU: the dipoles of ghost atoms travel through
  // Comm::forward_comm_pair -> pack_forward_comm / unpack_forward_comm, 3 doubles per atom.
  // (The reference declares pack_comm/unpack_comm but never calls them: it is single-process only.)
  comm_forward = 3;
  nhalo = 0;
  dist = NULL;
  const char *dev = getenv("POLAR_DEVICE");
  int ndev = polar_device_count();
  device_index = dev ? atoi(dev) : (ndev > 0 ? comm->me % ndev : 0);
  int rc = polar_create(device_index, &h);
  if (rc < 0) error->all(FLERR,"Pair style lj/cut/coul/long/polarization (MI355X): cannot create the library handle");
  // (a machine without a GPU may still parse a deck, read and write restart files and call single(): the host mirror of
  //  the library needs no device.  init_style refuses to go on without one, and every compute entry point fails loudly.)
}

PairLJCutCoulLongPolarizationMI355X::~PairLJCutCoulLongPolarizationMI355X()
{
  if (dist) polar_dist_destroy(dist);
  if (allocated) {
    memory->destroy(setflag);
    memory->destroy(cutsq);
    memory->sfree(epsilon);
    memory->sfree(sigma);
    memory->sfree(cut_lj);
  }
  polar_destroy(h);
}

void PairLJCutCoulLongPolarizationMI355X::check(int rc)
{
  // several ranks: a library error is local to one rank (its GPU, its atoms), and error->all from inside the
  // per-sweep loop would leave the others waiting in MPI_Allreduce / forward_comm -> error->one (MPI_Abort)
  if (rc < 0) { if (comm->nprocs > 1) error->one(FLERR,polar_last_error(h)); else error->all(FLERR,polar_last_error(h)); }
  else if (rc == POLAR_WARN_NOT_CONVERGED) error->warning(FLERR,polar_last_warning(h));  // PS.cpp:1233
}

/* ----------------------------------------------------------------------
   `debug yes`: the reference's prints (PS.cpp:391-404, 633-640, 1182-1191), same formats.  The two
   "polar force on atom 0" lines of the reference print loop temporaries of its last pair and are
   not reproduced.
------------------------------------------------------------------------- */

void PairLJCutCoulLongPolarizationMI355X::debug_prints(const polar_result &res)
{
  std::vector<double> trace((size_t) res.sweeps + 8);
  const int nt = polar_get_debug_trace(h,trace.data(),(int) trace.size());
  for (int k = 0; k < nt; k++)
    printf("u_polar (K) %d: %.18f\n",k,trace[k]*22.432653052265*22.432653052265);     // PS.cpp:1190
  if (screen) fprintf(screen,"iterations: %d\n",res.iterations);                       // PS.cpp:391
  double u_polar = 0.0;
  for (int i = 0; i < atom->nlocal; i++)
    u_polar += atom->ef_static[i][0]*atom->mu_induced[i][0] + atom->ef_static[i][1]*atom->mu_induced[i][1] +
               atom->ef_static[i][2]*atom->mu_induced[i][2];
  u_polar *= -0.5;
  printf("u_polar: %.18f\n",u_polar);                                                  // PS.cpp:403
  printf("self: %.18f\nef: %.18f\ndd: %.18f\n",res.u_self,res.u_ef,res.u_dd);        // PS.cpp:635
  printf("u_polar calc: %.18f\n",res.eng_pol);
  double f0[6] = {0.0,0.0,0.0,0.0,0.0,0.0};
  polar_get_debug_forces(h,f0);                                                        // PS.cpp:637-638
  printf("polar force on atom 0: %.18f,%.18f,%.18f\n",f0[0],f0[1],f0[2]);
  printf("polar dipole force on atom 0: %.18f,%.18f,%.18f\n",f0[3],f0[4],f0[5]);
  if (atom->nlocal > 0) printf("pos of atom 0: %.5f,%.5f,%.5f\n",atom->x[0][0],atom->x[0][1],atom->x[0][2]);
}

/* ---------------------------------------------------------------------- */

void PairLJCutCoulLongPolarizationMI355X::compute(int eflag, int vflag)
{
  if (eflag || vflag) ev_setup(eflag,vflag);
  else {
    // PS.cpp:126-127 clears only evflag and vflag_fdotr and then reads its ARGUMENTS; this function reads the flags
    // ev_setup derives from them, which would otherwise still hold the last thermo step's values (Pair::ev_unset)
    evflag = vflag_fdotr = 0;
    eflag_either = eflag_global = eflag_atom = 0;
    vflag_either = vflag_global = vflag_atom = 0;
  }
  if (comm->nprocs > 1) { compute_sharded(eflag,vflag); return; }
  // what compute() reads through domain->, atom-> and list-> (PS.cpp:125-188)
  double tilt[3] = {domain->xy,domain->xz,domain->yz};
  int periodic[3] = {domain->xperiodic,domain->yperiodic,domain->zperiodic};
  check(polar_set_box(h,domain->boxlo,domain->prd,tilt,periodic,domain->triclinic));
  // atoms are reordered, exchanged and re-ghosted only when the lists are rebuilt: in between only their positions move
  // (the reference re-reads atom->q every step, PS.cpp:125-188: a fix that changes charges between two list builds --
  //  fix qeq/*, fix adapt charge -- is noticed by comparing with the charges last sent, 8 B per atom, and answered with a
  //  full upload.  static_polarizability, type and molecule only change through commands that force a new list.)
  const int nall_now = atom->nlocal + atom->nghost;
  const bool q_changed = atoms_sent && (int) q_sent.size() == nall_now && nall_now > 0 &&
                         memcmp(q_sent.data(),atom->q,(size_t) nall_now*sizeof(double)) != 0;
  if (neighbor->ago == 0 || !atoms_sent || atom->nlocal != sent_nlocal || atom->nghost != sent_nghost || q_changed) {
    check(polar_set_atoms(h,atom->nlocal,atom->nghost,&atom->x[0][0],atom->q,atom->static_polarizability,
                          atom->type,(const int *) atom->molecule));
    atoms_sent = 1; sent_nlocal = atom->nlocal; sent_nghost = atom->nghost;
    q_sent.assign(atom->q,atom->q + nall_now);
  } else check(polar_set_positions(h,atom->nlocal,atom->nghost,&atom->x[0][0]));
  if (neighbor->ago == 0) {
    check(polar_set_newton(h,force->newton_pair));       // PS.cpp:293 and the ev_tally weights read force->newton_pair
    if (device_neigh)
      check(polar_build_neighbors(h,&neighbor->cutneighsq[0][0],(const int *) atom->tag,
                                  atom->molecular ? &atom->nspecial[0][0] : NULL,
                                  atom->molecular ? (const int *) &atom->special[0][0] : NULL,atom->maxspecial,
                                  neighbor->special_flag,neighbor->nex_mol > 0));
    else
      check(polar_set_neighbors(h,list->inum,list->ilist,list->numneigh,list->firstneigh));
  }

  // global virial: fdotr is left to the base class (it needs the reverse-communicated ghosts'
  // x, which LAMMPS owns); a pairwise global virial (vflag_global == 1) is tallied on the device
  // per-atom tallies go straight into Pair::eatom / Pair::vatom, which ev_setup has just zeroed
  // (src/pair.cpp:789-806); both are contiguous (src/memory.h:118-131)
  polar_result res;
  const int ef = (eflag_either ? 1 : 0) | (eflag_atom ? 2 : 0);
  // with device_neigh, no_virial_fdotr_compute leaves vflag_global = 2 for an fdotr-style request: the library
  // then tallies the LJ/Coulomb part pairwise and the polarization part as sum f.x, like the reference
  const int vf = (device_neigh ? vflag_global : (vflag_global ? 1 : 0)) | (vflag_atom ? 4 : 0);
  int rc;
  if (eflag_atom || vflag_atom)
    rc = polar_compute_peratom(h,ef,vf,&atom->f[0][0],&atom->mu_induced[0][0],&atom->ef_static[0][0],
                               eflag_atom ? eatom : NULL,vflag_atom ? &vatom[0][0] : NULL,&res);
  else
    rc = polar_compute(h,ef,vf,&atom->f[0][0],&atom->mu_induced[0][0],&atom->ef_static[0][0],&res);
  check(rc);

  if (eflag_global) {
    eng_vdwl += res.eng_vdwl;
    eng_coul += res.eng_coul;
  }
  force->pair->eng_pol = res.eng_pol;   // PS.cpp:641 (zero on steps without eflag)
  if (vflag_global) for (int k = 0; k < 6; k++) virial[k] += res.virial[k];
  if (vflag_fdotr) virial_fdotr_compute();
  if (debug_flag) debug_prints(res);
}

/* ----------------------------------------------------------------------
   More than one MPI rank (one rank per GPU).  The reference cannot do this at all: its
   polarization loops run over local atoms only (PS.cpp:330-641).  Here every rank hands the library
   [own atoms | halo atoms | remaining ghosts]: halo = one ghost per foreign atom (distinct tag not
   owned by this rank); inside the GLOBAL periodic box the minimum image of an (own, halo) pair is the
   image LAMMPS' ghost shell carries, provided the box is at least twice the cutoff (list mode,
   keyword dd_cutoff).  The library sweeps the own rows (polar_set_row_range) and between sweeps the
   dipoles make the round trip  device -> atom->mu_induced -> Comm::forward_comm_pair -> device.
------------------------------------------------------------------------- */

void PairLJCutCoulLongPolarizationMI355X::build_halo_map()
{
  const int nlocal = atom->nlocal, nall = atom->nlocal + atom->nghost;
  if (atom->map_style == 0)
    error->all(FLERR,"Pair style lj/cut/coul/long/polarization on several MPI ranks needs an atom map");
  lib_of_lammps.assign(nall,-1);
  halo_ghost.clear();
  for (int i = 0; i < nlocal; i++) lib_of_lammps[i] = i;
  std::map<tagint,int> seen;
  std::vector<int> rest;
  for (int g = nlocal; g < nall; g++) {
    const int owner = atom->map(atom->tag[g]);           // closest image; an index < nlocal means "mine"
    if (owner >= 0 && owner < nlocal) { rest.push_back(g); continue; }
    if (seen.insert(std::make_pair(atom->tag[g],g)).second) halo_ghost.push_back(g);
    else rest.push_back(g);
  }
  nhalo = (int) halo_ghost.size();
  for (int k = 0; k < nhalo; k++) lib_of_lammps[halo_ghost[k]] = nlocal + k;
  for (size_t k = 0; k < rest.size(); k++) lib_of_lammps[rest[k]] = nlocal + nhalo + (int) k;
  lammps_of_lib.assign(nall,0);
  for (int a = 0; a < nall; a++) lammps_of_lib[lib_of_lammps[a]] = a;

  sh_n = nlocal + nhalo;
  if (device_neigh) return;        // the library builds the rows of the own atoms itself (polar_build_neighbors)

  // the half list with its indices in library order (special bits kept), rows = own atoms
  const int inum = list->inum;
  std::vector<int> nn(nlocal + nhalo,0);
  std::vector<long long> first(nlocal + nhalo,0);
  long long total = 0;
  for (int ii = 0; ii < inum; ii++) { const int i = list->ilist[ii]; first[i] = total; nn[i] = list->numneigh[i]; total += nn[i]; }
  std::vector<int> flat((size_t) total + 1);
  for (int ii = 0; ii < inum; ii++) {
    const int i = list->ilist[ii];
    const int *jl = list->firstneigh[i];
    for (int k = 0; k < nn[i]; k++)
      flat[first[i] + k] = lib_of_lammps[jl[k] & NEIGHMASK] | (jl[k] & ~NEIGHMASK);
  }
  sh_nn.swap(nn); sh_first.swap(first); sh_flat.swap(flat);
}

/* ----------------------------------------------------------------------
   `rccl_halo yes`: the halo plan of the library's RCCL driver, rebuilt with the halo map (reneighbor steps).
   Every rank publishes the tags of its halo atoms; the owner of a tag (atom->map gives an own index) answers with the
   slot it will fill.  Own atoms keep their LAMMPS indices in library order, halo slot k is library atom nlocal + k.
   The communicator is made once: rank 0 draws the RCCL id, MPI_Bcast carries it.
------------------------------------------------------------------------- */

void PairLJCutCoulLongPolarizationMI355X::build_rccl_plan()
{
  const int me = comm->me, np = comm->nprocs, nlocal = atom->nlocal;
  if (!dist) {
    char id[POLAR_DIST_ID_BYTES];
    memset(id,0,sizeof(id));
    int ok = 1;
    if (me == 0) ok = polar_dist_unique_id(id) == POLAR_OK;
    MPI_Bcast(&ok,1,MPI_INT,0,world);
    if (!ok) error->all(FLERR,"Pair style lj/cut/coul/long/polarization rccl_halo: RCCL is not available");
    MPI_Bcast(id,POLAR_DIST_ID_BYTES,MPI_CHAR,0,world);
    int rc = polar_dist_create(id,me,np,device_index,&dist);
    int bad = rc < 0 ? 1 : 0, any = 0;
    MPI_Allreduce(&bad,&any,1,MPI_INT,MPI_MAX,world);
    if (any) error->all(FLERR,"Pair style lj/cut/coul/long/polarization rccl_halo: cannot create the RCCL communicator");
  }
  // the halo tags of all ranks
  std::vector<int> cnt(np,0), disp(np + 1,0);
  int mine = nhalo;
  MPI_Allgather(&mine,1,MPI_INT,cnt.data(),1,MPI_INT,world);
  for (int r = 0; r < np; r++) disp[r + 1] = disp[r] + cnt[r];
  std::vector<tagint> mytags(nhalo + 1), alltags((size_t) disp[np] + 1);
  for (int k = 0; k < nhalo; k++) mytags[k] = atom->tag[halo_ghost[k]];
  MPI_Allgatherv(mytags.data(),nhalo,MPI_LMP_TAGINT,alltags.data(),cnt.data(),disp.data(),MPI_LMP_TAGINT,world);
  // the slots of the others' halos that this rank fills, and with which own atoms
  std::vector<int> scount(np,0), sdisp(np + 1,0), rcount(np,0), rdisp(np + 1,0), sslot, sidx;
  for (int r = 0; r < np; r++) {
    if (r != me)
      for (int k = 0; k < cnt[r]; k++) {
        const int a = atom->map(alltags[(size_t) disp[r] + k]);
        if (a >= 0 && a < nlocal) { sslot.push_back(k); sidx.push_back(a); scount[r]++; }
      }
    sdisp[r + 1] = sdisp[r] + scount[r];
  }
  MPI_Alltoall(scount.data(),1,MPI_INT,rcount.data(),1,MPI_INT,world);
  for (int r = 0; r < np; r++) rdisp[r + 1] = rdisp[r] + rcount[r];
  std::vector<int> rslot((size_t) rdisp[np] + 1);
  sslot.push_back(0);
  MPI_Alltoallv(sslot.data(),scount.data(),sdisp.data(),MPI_INT,rslot.data(),rcount.data(),rdisp.data(),MPI_INT,world);
  // every halo slot has exactly one owner
  int bad = rdisp[np] != nhalo ? 1 : 0, any = 0;
  std::vector<char> filled(nhalo + 1,0);
  for (int k = 0; k < rdisp[np] && !bad; k++) {
    if (rslot[k] < 0 || rslot[k] >= nhalo || filled[rslot[k]]) bad = 1;
    else filled[rslot[k]] = 1;
  }
  MPI_Allreduce(&bad,&any,1,MPI_INT,MPI_MAX,world);
  if (any) error->all(FLERR,"Pair style lj/cut/coul/long/polarization rccl_halo: a halo atom has no owner or several");
  std::vector<int> peers, sc, rc_, sflat, rflat;
  for (int r = 0; r < np; r++) {
    if (scount[r] == 0 && rcount[r] == 0) continue;
    peers.push_back(r); sc.push_back(scount[r]); rc_.push_back(rcount[r]);
    for (int k = sdisp[r]; k < sdisp[r + 1]; k++) sflat.push_back(sidx[k]);
    for (int k = rdisp[r]; k < rdisp[r + 1]; k++) rflat.push_back(nlocal + rslot[k]);
  }
  sflat.push_back(0); rflat.push_back(0); peers.push_back(0); sc.push_back(0); rc_.push_back(0);   // (never empty pointers)
  if (polar_dist_set_halo(dist,h,(int) peers.size() - 1,peers.data(),sc.data(),sflat.data(),rc_.data(),rflat.data()) < 0)
    error->one(FLERR,polar_dist_last_error(dist));
  // Schedule (polar_dist_set_schedule).  Default: every rank colours for itself and ALL halo dipoles travel once per sweep on
  // the compute stream (lag -1) -- one stream, one communicator, the only form that has run on real RCCL, and by the one-rank
  // measurements the cheaper one below ~100k own atoms per rank.  POLAR_DIST_LAG=0|1|2 switches to ONE
  // colouring shared by the ranks -- they colour in turns, no two peers sharing a turn: greedy colouring of the peer graph in
  // rank order, computed identically by every rank from the gathered peer lists -- with colour c's boundary dipoles travelling
  // after phase c on the library's communication stream, at most `lag` phases late.  lag >= 1 lets an unpack land while the next
  // phase reads the halo dipoles: results are then not reproducible run to run, and `deterministic yes` refuses it.
  int lag = -1;
  polar_settings pst;
  check(polar_get_settings(h,&pst));
  if (const char *e = getenv("POLAR_DIST_LAG")) lag = atoi(e);
  if (lag < -1 || lag > 2) error->all(FLERR,"Pair style lj/cut/coul/long/polarization rccl_halo: POLAR_DIST_LAG is -1, 0, 1 or 2");
  const bool det = pst.deterministic == POLAR_DET_YES || (pst.deterministic == POLAR_DET_AUTO && pst.fixed_iteration && !pst.polar_accel);
  if (lag >= 1 && det)
    error->all(FLERR,"Pair style lj/cut/coul/long/polarization rccl_halo: deterministic yes (the default of fixed_iteration runs) needs POLAR_DIST_LAG <= 0");
  std::vector<int> ispeer(np,0), allpeer((size_t) np*np,0), cls(np,0);
  for (size_t k = 0; k + 1 < peers.size(); k++) if (peers[k] != me) ispeer[peers[k]] = 1;
  MPI_Allgather(ispeer.data(),np,MPI_INT,allpeer.data(),np,MPI_INT,world);
  int ncls = 0;
  for (int r = 0; r < np; r++) {
    int c = 0;
    for (bool again = true; again; ) {
      again = false;
      for (int q2 = 0; q2 < r; q2++)
        if ((allpeer[(size_t) r*np + q2] || allpeer[(size_t) q2*np + r]) && cls[q2] == c) { c++; again = true; break; }
    }
    cls[r] = c;
    if (c + 1 > ncls) ncls = c + 1;
  }
  if (polar_dist_set_schedule(dist,lag,cls[me],lag >= 0 ? ncls : 0) < 0) error->one(FLERR,polar_dist_last_error(dist));
}

void PairLJCutCoulLongPolarizationMI355X::exchange_dipoles()
{
  const int nlocal = atom->nlocal;
  check(polar_step_mu_get(h,0,nlocal,&atom->mu_induced[0][0]));
  comm->forward_comm_pair(this);                       // ghosts of atom->mu_induced now hold their owners' dipoles
  sh_mu.resize(3 * (size_t) nhalo + 3);
  sh_idx.resize((size_t) nhalo + 1);
  for (int k = 0; k < nhalo; k++) {
    const double *m = atom->mu_induced[halo_ghost[k]];
    sh_mu[3*k] = m[0]; sh_mu[3*k+1] = m[1]; sh_mu[3*k+2] = m[2];
    sh_idx[k] = nlocal + k;
  }
  check(polar_step_mu_put_idx(h,nhalo,sh_idx.data(),sh_mu.data()));
}

void PairLJCutCoulLongPolarizationMI355X::compute_sharded(int eflag, int vflag)
{
  polar_settings pst;
  check(polar_get_settings(h,&pst));
  if (!(pst.dd_cutoff > 0.0))
    error->all(FLERR,"Pair style lj/cut/coul/long/polarization on several MPI ranks needs the dd_cutoff keyword (list mode)");
  if (!force->newton_pair)
    error->all(FLERR,"Pair style lj/cut/coul/long/polarization on several MPI ranks needs newton_pair on");
  // halo atoms are LAMMPS ghosts: the ghost shell must reach as far as the polarization loops do
  const double reach = MAX(pst.cut_coul,pst.dd_cutoff);
  if (comm->cutghost[0] < reach || comm->cutghost[1] < reach || comm->cutghost[2] < reach)
    error->all(FLERR,"Pair style lj/cut/coul/long/polarization: ghost cutoff is shorter than max(cut_coul,dd_cutoff): use comm_modify cutoff");
  const int nlocal = atom->nlocal, nall = atom->nlocal + atom->nghost;
  const bool relist = neighbor->ago == 0 || (int) lib_of_lammps.size() != nall;
  if (relist) build_halo_map();

  // per-atom inputs in library order
  sh_x.resize(3 * (size_t) nall); sh_q.resize(nall); sh_a.resize(nall); sh_t.resize(nall); sh_m.resize(nall);
  for (int k = 0; k < nall; k++) {
    const int a = lammps_of_lib[k];
    sh_x[3*k] = atom->x[a][0]; sh_x[3*k+1] = atom->x[a][1]; sh_x[3*k+2] = atom->x[a][2];
    sh_q[k] = atom->q[a]; sh_a[k] = atom->static_polarizability[a];
    sh_t[k] = atom->type[a]; sh_m[k] = (int) atom->molecule[a];
  }
  double tilt[3] = {domain->xy,domain->xz,domain->yz};
  int periodic[3] = {domain->xperiodic,domain->yperiodic,domain->zperiodic};
  check(polar_set_box(h,domain->boxlo,domain->prd,tilt,periodic,domain->triclinic));
  check(polar_set_atoms(h,sh_n,nall - sh_n,sh_x.data(),sh_q.data(),sh_a.data(),sh_t.data(),sh_m.data()));
  check(polar_set_row_range(h,0,nlocal));
  check(polar_set_global_count(h,(long long) atom->natoms));
  check(polar_set_newton(h,1));
  if (relist && device_neigh) {
    // full list for the own rows, built on the device among [own | halo | other ghosts]: tags in library order; the
    // special lists are read for the own atoms only, which keep their LAMMPS indices
    sh_t2.resize(nall);
    for (int k = 0; k < nall; k++) sh_t2[k] = (int) atom->tag[lammps_of_lib[k]];
    check(polar_build_neighbors(h,&neighbor->cutneighsq[0][0],sh_t2.data(),
                                atom->molecular ? &atom->nspecial[0][0] : NULL,
                                atom->molecular ? (const int *) &atom->special[0][0] : NULL,atom->maxspecial,
                                neighbor->special_flag,neighbor->nex_mol > 0));
  } else if (relist) {
    std::vector<int> il(list->inum);
    for (int ii = 0; ii < list->inum; ii++) il[ii] = list->ilist[ii];
    check(polar_set_neighbors_csr(h,list->inum,il.data(),sh_nn.data(),sh_first.data(),sh_flat.data()));
  }
  if (pst.use_previous) {
    // the initial guess is atom->mu_induced (PS.cpp:376-386): it migrates with the atoms (AtomVecFullPolar packs it
    // into exchange records) and reaches the ghosts through forward_comm; the library's resident copy is in LAST
    // step's library order and must not be trusted across a relist, a sort or a migration
    comm->forward_comm_pair(this);
    sh_mu.resize(3 * (size_t) sh_n + 3);
    for (int k = 0; k < sh_n; k++) {
      const double *m = atom->mu_induced[lammps_of_lib[k]];
      sh_mu[3*k] = m[0]; sh_mu[3*k+1] = m[1]; sh_mu[3*k+2] = m[2];
    }
    check(polar_upload_mu(h,sh_mu.data(),3 * (long long) sh_n));
  }

  // eflag & 2 / vflag & 4: per-atom tallies (src/pair.cpp:760-764), fetched after the step
  const int ef = (eflag_either ? 1 : 0) | (eflag_atom ? 2 : 0), vf = (vflag_global ? 1 : 0) | (vflag_atom ? 4 : 0);
  polar_result res;
  if (pst.rccl_halo) {
    // the whole solve inside the library: per sweep a pack kernel, one grouped RCCL send/receive with the halo peers, an
    // unpack kernel and (every sweep, PS.cpp:1194-1210 over all ranks) one all-reduced double, all on the compute stream;
    // retries after a pitch overflow are agreed on by the driver itself
    if (relist) build_rccl_plan();
    int rc = polar_dist_step(dist,h,ef,vf,&res);
    if (rc < 0) error->one(FLERR,polar_dist_last_error(dist));
    check(rc);
    // polar_dist_step returns energies and virial SUMMED over the ranks; LAMMPS sums the per-rank accumulators itself
    // (compute_pe.cpp:80, thermo.cpp:2227, compute pressure): this rank contributes its own share only
    polar_result mine;
    if (polar_dist_local_result(dist,&mine) < 0) error->one(FLERR,polar_dist_last_error(dist));
    res.eng_vdwl = mine.eng_vdwl; res.eng_coul = mine.eng_coul; res.eng_pol = mine.eng_pol;
    res.u_self = mine.u_self; res.u_ef = mine.u_ef; res.u_dd = mine.u_dd;
    for (int k = 0; k < 6; k++) res.virial[k] = mine.virial[k];
  } else
  for (int attempt = 0;; attempt++) {
    check(polar_step_begin(h,ef,vf));
    exchange_dipoles();                                // the other ranks' initial guess
    if (!pst.zodid)
      for (int sw = 0; sw <= pst.iterations_max; sw++) {
        check(polar_step_sweep(h));
        if (!pst.fixed_iteration) {                    // PS.cpp:1194-1210 with the sum over all ranks
          double mine = 0.0, all = 0.0;
          check(polar_step_change_get(h,&mine));
          MPI_Allreduce(&mine,&all,1,MPI_DOUBLE,MPI_SUM,world);
          check(polar_step_sweep_end_host(h,all));
        } else check(polar_step_sweep_end(h,NULL));
        exchange_dipoles();
        if (!pst.fixed_iteration) {
          int done = 0, it = 0, st = 0;
          check(polar_step_state(h,&done,&it,&st));    // identical on every rank: same global sum
          if (done) break;
        }
      }
    int rc = polar_step_finish(h,&res);
    int retry = rc == POLAR_RETRY_STEP ? 1 : 0, any = 0;
    MPI_Allreduce(&retry,&any,1,MPI_INT,MPI_MAX,world);  // a row outgrew its pitch somewhere: all repeat
    if (!any) { check(rc); break; }
    if (attempt >= 4) error->all(FLERR,"Pair style lj/cut/coul/long/polarization: neighbor row pitch overflow persists");
  }

  // results back in LAMMPS order: forces on locals AND ghosts (newton on), dipoles / static field of the own atoms
  sh_f.resize(3 * (size_t) nall);
  check(polar_download(h,"f",sh_f.data(),3 * (long long) nall));
  for (int k = 0; k < nall; k++) {
    double *fa = atom->f[lammps_of_lib[k]];
    fa[0] += sh_f[3*k]; fa[1] += sh_f[3*k+1]; fa[2] += sh_f[3*k+2];
  }
  sh_x.resize(3 * (size_t) std::max(sh_n,nall));
  check(polar_download(h,"mu",sh_x.data(),3 * (long long) sh_n));
  memcpy(&atom->mu_induced[0][0],sh_x.data(),3 * (size_t) nlocal * sizeof(double));
  check(polar_download(h,"ef_static",sh_x.data(),3 * (long long) sh_n));
  memcpy(&atom->ef_static[0][0],sh_x.data(),3 * (size_t) nlocal * sizeof(double));

  // per-atom tallies of this rank's rows, locals and ghosts (the ghosts' halves travel home with the reverse
  // communication compute pe/atom and compute stress/atom do), accumulated like ev_tally does
  if (eflag_atom) {
    sh_f.resize((size_t) nall);
    check(polar_download(h,"eatom",sh_f.data(),(long long) nall));
    for (int k = 0; k < nall; k++) eatom[lammps_of_lib[k]] += sh_f[k];
  }
  if (vflag_atom) {
    sh_f.resize(6 * (size_t) nall);
    check(polar_download(h,"vatom",sh_f.data(),6 * (long long) nall));
    for (int k = 0; k < nall; k++) {
      double *va = vatom[lammps_of_lib[k]];
      for (int c = 0; c < 6; c++) va[c] += sh_f[6*k+c];
    }
  }

  if (eflag_global) {
    eng_vdwl += res.eng_vdwl;
    eng_coul += res.eng_coul;
  }
  if (debug_flag && comm->me == 0) debug_prints(res);
  force->pair->eng_pol = res.eng_pol;   // this rank's share; compute pe / thermo sum over ranks
  if (vflag_global) for (int k = 0; k < 6; k++) virial[k] += res.virial[k];
  if (vflag_fdotr) virial_fdotr_compute();
}

/* ---------------------------------------------------------------------- */

int PairLJCutCoulLongPolarizationMI355X::pack_forward_comm(int n, int *list_, double *buf, int /*pbc_flag*/, int * /*pbc*/)
{
  int m = 0;
  for (int i = 0; i < n; i++) {
    const double *mu = atom->mu_induced[list_[i]];
    buf[m++] = mu[0]; buf[m++] = mu[1]; buf[m++] = mu[2];
  }
  return m;
}

void PairLJCutCoulLongPolarizationMI355X::unpack_forward_comm(int n, int first, double *buf)
{
  int m = 0;
  for (int i = first; i < first + n; i++) {
    double *mu = atom->mu_induced[i];
    mu[0] = buf[m++]; mu[1] = buf[m++]; mu[2] = buf[m++];
  }
}

/* ---------------------------------------------------------------------- */

void PairLJCutCoulLongPolarizationMI355X::allocate()
{
  allocated = 1;
  int n = atom->ntypes;
  memory->create(setflag,n+1,n+1,"pair:setflag");
  for (int i = 1; i <= n; i++)
    for (int j = i; j <= n; j++)
      setflag[i][j] = 0;
  memory->create(cutsq,n+1,n+1,"pair:cutsq");
  epsilon = (double **) memory->smalloc((n+1)*sizeof(double *),"pair:epsilon");
  sigma = (double **) memory->smalloc((n+1)*sizeof(double *),"pair:sigma");
  cut_lj = (double **) memory->smalloc((n+1)*sizeof(double *),"pair:cut_lj");
}

// row-pointer views into the library's [(n+1)*(n+1)] tables so that extract("epsilon"/"sigma")
// hands out the double** fix adapt and friends expect (PS.cpp:1101-1109)
void PairLJCutCoulLongPolarizationMI355X::sync_views()
{
  int dim, n = atom->ntypes;
  double *e = (double *) polar_pair_extract(h,"epsilon",&dim);
  double *s = (double *) polar_pair_extract(h,"sigma",&dim);
  double *c = (double *) polar_pair_extract(h,"cut_lj",&dim);
  for (int i = 0; i <= n; i++) {
    epsilon[i] = e + (size_t) i*(n+1);
    sigma[i] = s + (size_t) i*(n+1);
    cut_lj[i] = c + (size_t) i*(n+1);
  }
}

/* ---------------------------------------------------------------------- */

void PairLJCutCoulLongPolarizationMI355X::settings(int narg, char **arg)
{
  // identical grammar, ordering quirks and error strings: parsed by the library, PS.cpp:678-766
  check(polar_pair_settings(h,narg,arg));
  polar_settings st;
  polar_get_settings(h,&st);
  cut_lj_global = st.cut_lj_global;
  cut_coul = st.cut_coul;
}

void PairLJCutCoulLongPolarizationMI355X::coeff(int narg, char **arg)
{
  if (narg < 4 || narg > 5) error->all(FLERR,"Incorrect args for pair coefficients");
  if (!allocated) allocate();
  check(polar_pair_coeff(h,atom->ntypes,narg,arg));   // PS.cpp:772-800
  sync_views();
  int ilo,ihi,jlo,jhi;
  force->bounds(FLERR,arg[0],atom->ntypes,ilo,ihi);
  force->bounds(FLERR,arg[1],atom->ntypes,jlo,jhi);
  for (int i = ilo; i <= ihi; i++)
    for (int j = MAX(jlo,i); j <= jhi; j++) setflag[i][j] = 1;
}

/* ---------------------------------------------------------------------- */

void PairLJCutCoulLongPolarizationMI355X::init_style()
{
  // PS.cpp:806-852
  if (!atom->q_flag)
    error->all(FLERR,"Pair style lj/cut/coul/long requires atom attribute q");
  if (!atom->static_polarizability_flag)
    error->all(FLERR,"Pair style lj/cut/coul/long/polarization requires atom attribute polarizability");
  if (strstr(update->integrate_style,"respa"))
    error->all(FLERR,"Pair style lj/cut/coul/long/polarization does not support rRESPA");  // respa_enable = 0
  if (polar_device_count() < 1 && !getenv("POLAR_HOST_PATHS_ONLY"))   // (the variable: tests of the host-side paths)
    error->all(FLERR,"Pair style lj/cut/coul/long/polarization (MI355X) found no usable HIP device");
  int irequest = neighbor->request(this,instance_me);     // default half list, newton on
  polar_settings pst;
  check(polar_get_settings(h,&pst));
  device_neigh = pst.device_neigh;
  debug_flag = pst.debug;
  if (device_neigh) {
    // extension keyword `device_neigh yes`: the library bins locals + ghosts and builds the (full) list
    // itself (polar_build_neighbors), so Neighbor never builds this request -- it still decides WHEN to
    // reneighbor (neighbor->ago) and provides cutneighsq / special_flag / the exclusion settings.
    // A full list puts no force on ghosts: the base class must not form the fdotr virial.
    neighbor->requests[irequest]->occasional = 1;
    no_virial_fdotr_compute = 1;
    if (neighbor->nex_type || neighbor->nex_group)
      error->all(FLERR,"Pair style lj/cut/coul/long/polarization device_neigh supports only neigh_modify exclude molecule/intra all");
    for (int k = 0; k < neighbor->nex_mol; k++)
      if (!neighbor->ex_mol_intra[k])
        error->all(FLERR,"Pair style lj/cut/coul/long/polarization device_neigh supports only neigh_modify exclude molecule/intra all");
  }
  if (force->kspace == NULL) error->all(FLERR,"Pair style requires a KSpace style");

  // pair_modify state lives in the Pair base class; mirror it into the library, then let it build
  // lj1..lj4/offset/cutsq (init_one) and the Coulomb tables (init_tables) and upload them
  char bits[32],inner[64];
  sprintf(bits,"%d",ncoultablebits);
  sprintf(inner,"%.17g",tabinner);
  const char *mix = mix_flag == GEOMETRIC ? "geometric" : (mix_flag == ARITHMETIC ? "arithmetic" : "sixthpower");
  const char *mod[10] = {"mix",mix,"shift",offset_flag ? "yes" : "no","table",bits,"tabinner",inner,
                         "tail",tail_flag ? "yes" : "no"};
  check(polar_pair_modify(h,10,mod));
  check(polar_pair_init(h,force->kspace->g_ewald,force->qqrd2e,force->special_lj,force->special_coul));
  // the Coulomb lookup tables stay LAMMPS host code: Pair::init_tables (src/pair.cpp:313-520), as PS.cpp:851 calls it
  if (ncoultablebits) {
    init_tables(cut_coul,NULL);
    check(polar_set_coul(h,force->kspace->g_ewald,force->qqrd2e,force->special_lj,force->special_coul,
                         ncoultablebits,ncoulmask,ncoulshiftbits,tabinnersq,rtable,drtable,ftable,dftable,
                         ctable,dctable,etable,detable));
  }
  sync_views();
  pair_inited = 1;
}

double PairLJCutCoulLongPolarizationMI355X::init_one(int i, int j)
{
  // mixing / lj1..lj4 / offset were computed for every pair by polar_pair_init (PS.cpp:858-890)
  double cut = polar_pair_cut(h,i,j);

  // long-range tail correction, PS.cpp:897-918 (host-side bookkeeping of the Pair base class)
  if (tail_flag) {
    int *type = atom->type;
    int nlocal = atom->nlocal;
    double count[2],all[2];
    count[0] = count[1] = 0.0;
    for (int k = 0; k < nlocal; k++) {
      if (type[k] == i) count[0] += 1.0;
      if (type[k] == j) count[1] += 1.0;
    }
    MPI_Allreduce(count,all,2,MPI_DOUBLE,MPI_SUM,world);
    const double PI = 3.14159265358979323846;
    double sig2 = sigma[i][j]*sigma[i][j];
    double sig6 = sig2*sig2*sig2;
    double rc3 = cut_lj[i][j]*cut_lj[i][j]*cut_lj[i][j];
    double rc6 = rc3*rc3;
    double rc9 = rc3*rc6;
    etail_ij = 8.0*PI*all[0]*all[1]*epsilon[i][j] * sig6 * (sig6 - 3.0*rc6) / (9.0*rc9);
    ptail_ij = 16.0*PI*all[0]*all[1]*epsilon[i][j] * sig6 * (2.0*sig6 - 3.0*rc6) / (9.0*rc9);
  }
  return cut;
}

/* ----------------------------------------------------------------------
   restart / data files: same records as the reference (PS.cpp:927-1031); like the reference,
   the polarization keywords and the dipoles are NOT persisted
------------------------------------------------------------------------- */

void PairLJCutCoulLongPolarizationMI355X::write_restart(FILE *fp)
{
  write_restart_settings(fp);
  for (int i = 1; i <= atom->ntypes; i++)
    for (int j = i; j <= atom->ntypes; j++) {
      fwrite(&setflag[i][j],sizeof(int),1,fp);
      if (setflag[i][j]) {
        fwrite(&epsilon[i][j],sizeof(double),1,fp);
        fwrite(&sigma[i][j],sizeof(double),1,fp);
        fwrite(&cut_lj[i][j],sizeof(double),1,fp);
      }
    }
}

void PairLJCutCoulLongPolarizationMI355X::read_restart(FILE *fp)
{
  read_restart_settings(fp);
  allocate();
  int me = comm->me;
  for (int i = 1; i <= atom->ntypes; i++)
    for (int j = i; j <= atom->ntypes; j++) {
      int flag = 0;
      double v[3] = {0.0,0.0,0.0};
      if (me == 0) fread(&flag,sizeof(int),1,fp);
      MPI_Bcast(&flag,1,MPI_INT,0,world);
      if (flag) {
        if (me == 0) fread(v,sizeof(double),3,fp);
        MPI_Bcast(v,3,MPI_DOUBLE,0,world);
        char a[5][32];
        sprintf(a[0],"%d",i); sprintf(a[1],"%d",j);
        sprintf(a[2],"%.17g",v[0]); sprintf(a[3],"%.17g",v[1]); sprintf(a[4],"%.17g",v[2]);
        const char *argv[5] = {a[0],a[1],a[2],a[3],a[4]};
        check(polar_pair_coeff(h,atom->ntypes,5,argv));
        setflag[i][j] = 1;
      }
    }
  sync_views();
}

void PairLJCutCoulLongPolarizationMI355X::write_restart_settings(FILE *fp)
{
  fwrite(&cut_lj_global,sizeof(double),1,fp);
  fwrite(&cut_coul,sizeof(double),1,fp);
  fwrite(&offset_flag,sizeof(int),1,fp);
  fwrite(&mix_flag,sizeof(int),1,fp);
  fwrite(&tail_flag,sizeof(int),1,fp);
  fwrite(&ncoultablebits,sizeof(int),1,fp);
  fwrite(&tabinner,sizeof(double),1,fp);
  // extension (keyword restart_polar yes): the polarization keywords behind the stock record, tagged and
  // length-prefixed; without the keyword the file has the reference's layout (PS.cpp:976-985)
  polar_settings pst;
  check(polar_get_settings(h,&pst));
  if (pst.restart_polar) {
    char rec[256];
    int n = polar_restart_pack(h,rec,(int) sizeof(rec));
    if (n < 0) error->one(FLERR,"Pair style lj/cut/coul/long/polarization: cannot pack its restart record");
    fwrite(rec,1,n,fp);
  }
}

void PairLJCutCoulLongPolarizationMI355X::read_restart_settings(FILE *fp)
{
  if (comm->me == 0) {
    fread(&cut_lj_global,sizeof(double),1,fp);
    fread(&cut_coul,sizeof(double),1,fp);
    fread(&offset_flag,sizeof(int),1,fp);
    fread(&mix_flag,sizeof(int),1,fp);
    fread(&tail_flag,sizeof(int),1,fp);
    fread(&ncoultablebits,sizeof(int),1,fp);
    fread(&tabinner,sizeof(double),1,fp);
  }
  MPI_Bcast(&cut_lj_global,1,MPI_DOUBLE,0,world);
  MPI_Bcast(&cut_coul,1,MPI_DOUBLE,0,world);
  MPI_Bcast(&offset_flag,1,MPI_INT,0,world);
  MPI_Bcast(&mix_flag,1,MPI_INT,0,world);
  MPI_Bcast(&tail_flag,1,MPI_INT,0,world);
  MPI_Bcast(&ncoultablebits,1,MPI_INT,0,world);
  MPI_Bcast(&tabinner,1,MPI_DOUBLE,0,world);
  // a reference-format file holds no polarization keywords: defaults apply (reference behaviour) ...
  char a[2][32];
  sprintf(a[0],"%.17g",cut_lj_global); sprintf(a[1],"%.17g",cut_coul);
  const char *argv[2] = {a[0],a[1]};
  check(polar_pair_settings(h,2,argv));
  // ... unless the writer ran with restart_polar yes: probe for the tagged record, seek back when it is not there
  char rec[256];
  int nrec = 0;
  if (comm->me == 0) {
    const long pos = ftell(fp);
    size_t got = fread(rec,1,POLAR_RESTART_HEADER_BYTES,fp);
    int head[3] = {0,0,0};
    if (got == POLAR_RESTART_HEADER_BYTES) memcpy(head,rec,sizeof(head));
    if (got == POLAR_RESTART_HEADER_BYTES && head[0] == 0x524C4F50 && head[2] > 0 &&
        head[2] <= (int) sizeof(rec) - POLAR_RESTART_HEADER_BYTES &&
        fread(rec + POLAR_RESTART_HEADER_BYTES,1,head[2],fp) == (size_t) head[2])
      nrec = POLAR_RESTART_HEADER_BYTES + head[2];
    else fseek(fp,pos,SEEK_SET);
  }
  MPI_Bcast(&nrec,1,MPI_INT,0,world);
  if (nrec) {
    MPI_Bcast(rec,nrec,MPI_CHAR,0,world);
    check(polar_restart_unpack(h,rec,nrec));
  }
}

void PairLJCutCoulLongPolarizationMI355X::write_data(FILE *fp)
{
  for (int i = 1; i <= atom->ntypes; i++)
    fprintf(fp,"%d %g %g\n",i,epsilon[i][i],sigma[i][i]);
}

void PairLJCutCoulLongPolarizationMI355X::write_data_all(FILE *fp)
{
  for (int i = 1; i <= atom->ntypes; i++)
    for (int j = i; j <= atom->ntypes; j++)
      fprintf(fp,"%d %d %g %g %g\n",i,j,epsilon[i][j],sigma[i][j],cut_lj[i][j]);
}

/* ---------------------------------------------------------------------- */

double PairLJCutCoulLongPolarizationMI355X::single(int i, int j, int itype, int jtype, double rsq,
                                                   double factor_coul, double factor_lj, double &fforce)
{
  return polar_pair_single(h,atom->q[i],atom->q[j],itype,jtype,rsq,factor_coul,factor_lj,&fforce);  // PS.cpp:1035-1097
}

void *PairLJCutCoulLongPolarizationMI355X::extract(const char *str, int &dim)
{
  dim = 0;
  if (strcmp(str,"cut_coul") == 0) return (void *) &cut_coul;
  dim = 2;
  if (strcmp(str,"epsilon") == 0) return (void *) epsilon;
  if (strcmp(str,"sigma") == 0) return (void *) sigma;
  return NULL;
}
