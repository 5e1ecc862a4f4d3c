// polar_api.hip -- the C-ABI entry points (include/polar_mi355x.h): handle lifetime, the host mirror of the Pair text
// interface, per-step data, the compute calls and the stepwise / sharded interface.  The orchestration of a step lives in
// polar_step.hip, the colour phases in polar_color.hip, the multi-GPU driver in polar_dist.hip.
#include "polar_handle.hpp"

namespace {
// upload LJ tables + Coulomb tables from the host mirror (or raw setters) into P, repacked so that
// the half-list loop reads one 64-byte line per type pair / per Coulomb bin
void upload_types(polar_handle *h, int ntypes, const double *const t[7]) {
  need_device(h);
  const size_t m = (size_t)(ntypes + 1) * (ntypes + 1);
  std::vector<double> pack(8 * m, 0.0);
  // t = lj1, lj2, lj3, lj4, offset, cut_ljsq, cutsq  ->  cutsq, cut_ljsq, lj1, lj2, lj3, lj4, offset, pad
  for (size_t k = 0; k < m; k++) {
    pack[8 * k + 0] = t[6][k]; pack[8 * k + 1] = t[5][k]; pack[8 * k + 2] = t[0][k]; pack[8 * k + 3] = t[1][k];
    pack[8 * k + 4] = t[2][k]; pack[8 * k + 5] = t[3][k]; pack[8 * k + 6] = t[4][k];
  }
  h->d_lj.ensure(8 * m);
  HIPCHECK(hipMemcpy(h->d_lj.p, pack.data(), 8 * m * sizeof(double), hipMemcpyHostToDevice));
  h->P.ntypes = ntypes;
  h->P.ljpack = h->d_lj.p;
  h->ntypes = ntypes;
  h->types_set = true;
}

void upload_coul(polar_handle *h, double g_ewald, double qqrd2e, const double *slj, const double *scoul, int nbits,
                 int mask, int shift, double tabinnersq, const double *const t[8]) {
  need_device(h);
  h->P.g_ewald = g_ewald; h->P.qqrd2e = qqrd2e;
  for (int k = 0; k < 4; k++) { h->P.special_lj[k] = slj[k]; h->P.special_coul[k] = scoul[k]; }
  h->P.ncoultablebits = nbits; h->P.ncoulmask = mask; h->P.ncoulshiftbits = shift; h->P.tabinnersq = tabinnersq;
  const size_t nt = nbits ? ((size_t)1 << nbits) : 1;
  std::vector<double> pack(8 * nt, 0.0);
  // t = r, dr, f, df, c, dc, e, de  ->  r, dr, f, df, e, de, c, dc
  static const int order[8] = {0, 1, 2, 3, 6, 7, 4, 5};
  if (nbits)
    for (size_t b = 0; b < nt; b++)
      for (int k = 0; k < 8; k++) pack[8 * b + k] = t[order[k]][b];
  h->d_tab.ensure(8 * nt);
  HIPCHECK(hipMemcpy(h->d_tab.p, pack.data(), 8 * nt * sizeof(double), hipMemcpyHostToDevice));
  h->P.ctab = h->d_tab.p;
  // How k_ljcoul_pers may find a pair's bin (polar_rows.hpp, ljcoul_row TAB): checked here, bin by bin, against the tables the
  // caller handed over (Pair::init_tables, src/pair.cpp): r = a float whose low `shift` mantissa bits are zero and whose masked
  // bits give the bin's own index; dr = the reciprocal of one bin step of that binade, an exact power of two.  Up to two bins may
  // carry another dr (the last bin before the cutoff gets 1 / (cut_coulsq - r); the bin where the index wraps): they travel as
  // parameters.  Anything else -> the tables stay in memory for r and dr (TAB 1).
  h->P.tab_lowmask = nbits ? (1 << shift) - 1 : 0;
  h->P.i_special[0] = h->P.i_special[1] = -1;
  h->P.dr_special[0] = h->P.dr_special[1] = 0.0;
  h->lj_tab_arith = false;
  if (nbits) {
    bool ok = shift > 0 && shift < 23;
    int nspecial = 0;
    for (size_t b = 0; b < nt && ok; b++) {
      const double r = t[0][b], dr = t[1][b];
      const float rf = (float)r;
      int bits;
      memcpy(&bits, &rf, sizeof(int));
      if ((double)rf != r || (bits & h->P.tab_lowmask) != 0 || (size_t)((bits & mask) >> shift) != b) { ok = false; break; }
      const int e = (bits >> 23) & 0xFF;
      const double want = std::ldexp(1.0, 150 - shift - e);
      if (dr != want) {
        if (nspecial < 2) { h->P.i_special[nspecial] = (int)b; h->P.dr_special[nspecial] = dr; nspecial++; }
        else ok = false;
      }
    }
    h->lj_tab_arith = ok;
  }
  h->coul_set = true;
}
}  // namespace

// =============================================================================================
namespace {
// The half list of a reneighbor step on its way to the device (PS.cpp:232-247 reads list->firstneigh): rows are packed in
// ilist order into two pinned 32 MB buffers by the helper threads -- each thread checking its entries on the way -- and a
// buffer travels (hipMemcpyAsync from pinned memory: the full PCIe rate) while the next one is packed.  270 MB at 135k atoms:
// a single-threaded flatten + check + four copies out of pageable memory took 20 ms; this takes what the link takes.
// `row(i)` = where atom i's numneigh[i] entries lie on the host (LAMMPS' pages, or a caller's flat array).
template <typename RowFn>
void upload_neighbor_rows(polar_handle *h, int inum, const int *ilist, const int *numneigh, RowFn row) {
  // The list leaves on its OWN stream and the call returns when the last row has been packed into pinned memory, not when the
  // last byte has arrived: the tail of the transfer (270 MB at 134,900 atoms: ~5 ms of PCIe) runs under whatever the caller does
  // next -- polar_set_atoms, the first kernels of polar_compute --; the LJ/Coulomb loop's stream waits for `ev_list_up` before
  // it touches the list (launch_lj).  One pinned 32-MB buffer per chunk of the list (kept for the next list; a ring beyond 2 GB).
  const int n = h->nlocal, nall = h->nlocal + h->nghost;
  if (!h->up_stream) {
    HIPCHECK(hipStreamCreateWithFlags(&h->up_stream, hipStreamNonBlocking));
    HIPCHECK(hipEventCreateWithFlags(&h->ev_list_up, hipEventDisableTiming));
  }
  hipStream_t s = h->up_stream;
  HIPCHECK(hipStreamSynchronize(s));            // the previous list's transfer reads the staging buffers refilled below
  HIPCHECK(hipStreamSynchronize(h->stream));    // (and its consumers are done with the device arrays that may be re-allocated)
  if (h->lj_stream) HIPCHECK(hipStreamSynchronize(h->lj_stream));
  h->list_up_pending = false;
  std::vector<long long> &first = h->h_first;
  std::vector<int> &nn = h->h_nn;
  first.assign((size_t)std::max(n, 1), 0);
  nn.assign((size_t)std::max(n, 1), 0);
  std::vector<long long> at((size_t)inum + 1, 0);
  long long total = 0;
  for (int ii = 0; ii < inum; ii++) {
    const int i = ilist[ii];
    if (i < 0 || i >= n) throw InputError("neighbor list row index out of range");
    if (numneigh[i] < 0) throw InputError("negative neighbor count/offset");
    first[i] = total; nn[i] = numneigh[i]; at[ii] = total;
    total += numneigh[i];
  }
  at[inum] = total;
  h->inum = inum; h->nneigh = total;
  h->d_ilist.ensure(inum + 1); h->d_numneigh.ensure(n + 1); h->d_first.ensure(n + 1); h->d_neigh.ensure((size_t)total + 1);
  h->h_ilist.assign(ilist, ilist + inum);       // (the caller's ilist may change after the call)
  HIPCHECK(hipMemcpyAsync(h->d_ilist.p, h->h_ilist.data(), inum * sizeof(int), hipMemcpyHostToDevice, s));
  HIPCHECK(hipMemcpyAsync(h->d_numneigh.p, nn.data(), n * sizeof(int), hipMemcpyHostToDevice, s));
  HIPCHECK(hipMemcpyAsync(h->d_first.p, first.data(), n * sizeof(long long), hipMemcpyHostToDevice, s));
  const long long CH = 8ll << 20;   // entries per buffer (32 MB)
  constexpr size_t kRing = 64;      // at most 2 GB of pinned staging; longer lists reuse buffers as their transfers end
  HostPool &pool = HostPool::get();
  const int nt = pool.width();
  std::vector<int> bad((size_t)nt * 16, 0);
  size_t chunk = 0;
  for (int ii0 = 0; ii0 < inum;) {
    int ii1 = ii0;
    while (ii1 < inum && at[ii1 + 1] - at[ii0] <= CH) ii1++;
    if (ii1 == ii0) throw InputError("a neighbor row of more than 8M entries");
    const size_t buf = chunk % kRing;
    if (buf >= h->h_nl_stage.size()) {
      int *p = nullptr; hipEvent_t e = nullptr;
      HIPCHECK(hipHostMalloc((void **)&p, (size_t)CH * sizeof(int)));
      HIPCHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
      h->h_nl_stage.push_back(p); h->ev_nl.push_back(e);
    }
    if (chunk >= kRing) HIPCHECK(hipEventSynchronize(h->ev_nl[buf]));   // the transfer that last used this buffer is over
    int *dst = h->h_nl_stage[buf];
    const long long t0 = at[ii0], cnt = at[ii1] - t0;
    pool.run([&](int part) {
      // thread `part` packs the rows whose first entry lies in its share of the chunk's entries
      const long long lo = t0 + cnt * part / nt, hi = t0 + cnt * (part + 1) / nt;
      int a = (int)(std::lower_bound(at.begin() + ii0, at.begin() + ii1, lo) - at.begin());
      const int b = (int)(std::lower_bound(at.begin() + ii0, at.begin() + ii1, hi) - at.begin());
      int worst = 0;
      for (; a < b; a++) {
        const int i = ilist[a], m = numneigh[i];
        const int *src = row(i);
        int *d = dst + (at[a] - t0);
        int mx = 0;
        for (int k = 0; k < m; k++) { const int v = src[k]; d[k] = v; const int j = v & 0x3FFFFFFF; mx = j > mx ? j : mx; }
        worst = mx > worst ? mx : worst;
      }
      if (worst >= nall) bad[(size_t)part * 16] = 1;
    }, nt);
    for (int t = 0; t < nt; t++) if (bad[(size_t)t * 16]) { HIPCHECK(hipStreamSynchronize(s)); throw InputError("neighbor index out of range"); }
    if (cnt > 0) HIPCHECK(hipMemcpyAsync(h->d_neigh.p + t0, dst, (size_t)cnt * sizeof(int), hipMemcpyHostToDevice, s));
    HIPCHECK(hipEventRecord(h->ev_nl[buf], s));
    ii0 = ii1; chunk++;
  }
  HIPCHECK(hipEventRecord(h->ev_list_up, s));
  h->list_up_pending = true;
  h->neigh_set = true;
  h->sym_valid = false;
  if (h->colors_valid) h->colors_recheck = true;  // reneighbor step: the colour phases are re-validated (k_nl_build)
  h->device_list = false;
  h->full_list = h->user_full_list;
}
}  // namespace


extern "C" {

const char *polar_kernel_version(void) { return POLAR_KERNEL_VERSION; }

int polar_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int polar_create(int device, polar_handle **out) {
  if (!out) return POLAR_ERR_STATE;
  polar_handle *h = new polar_handle();
  *out = h;
  h->device = device;
  if (const char *e = getenv("POLAR_COLOR_DIST")) h->color_dist = atof(e);
  h->color_keep = std::max(h->color_dist - 0.4, 0.75 * h->color_dist);
  if (const char *e = getenv("POLAR_COLOR_KEEP")) h->color_keep = atof(e);
  if (const char *e = getenv("POLAR_DETERMINISTIC")) h->deterministic = atoi(e) != 0;  // the same as the keyword `deterministic yes`
#ifdef POLAR_LAB
#include "lab/api_env_knobs.inc"
#endif
  int n = polar_device_count();
  if (n <= 0 || device < 0 || device >= n) {
    h->have_device = false;  // host mirror still usable; compute entry points will fail loudly
    h->err = "no usable HIP device";
    return POLAR_OK;
  }
  return guarded(h, [&]() {
    HIPCHECK(hipSetDevice(device));
    HIPCHECK(hipStreamCreate(&h->stream));
    for (auto &e : h->ev) HIPCHECK(hipEventCreate(&e));
    {
      int prio_lo = 0, prio_hi = 0;  // numerically greatest = lowest priority
      HIPCHECK(hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi));
      HIPCHECK(hipStreamCreateWithPriority(&h->lj_stream, hipStreamNonBlocking, prio_lo));
    }
    HIPCHECK(hipEventCreate(&h->ev_fork)); HIPCHECK(hipEventCreate(&h->ev_join));
    HIPCHECK(hipEventCreate(&h->ev_lj0)); HIPCHECK(hipEventCreate(&h->ev_lj1));
    HIPCHECK(hipEventCreateWithFlags(&h->ev_dl0, hipEventDisableTiming)); HIPCHECK(hipEventCreateWithFlags(&h->ev_dl1, hipEventDisableTiming));
    HIPCHECK(hipEventCreateWithFlags(&h->ev_mu_ready, hipEventDisableTiming));
    for (auto &e : h->ev_fchunk) HIPCHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    HIPCHECK(hipStreamCreateWithFlags(&h->dl_stream, hipStreamNonBlocking));
    if (getenv("POLAR_NO_OVERLAP")) h->overlap_lj = false;
    if (const char *e = getenv("POLAR_LJ_PERS")) h->lj_pers = atoi(e);
    if (const char *e = getenv("POLAR_LJ_PERS_THREADS")) h->lj_pers_threads = std::max(256, std::min(POLAR_LJ_PERS_THREADS, (atoi(e) / 64) * 64));
    { int ncu = 0; if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess) h->ncu = ncu; }
    HIPCHECK(hipHostMalloc((void **)&h->h_scal, sizeof(Scal)));
    HIPCHECK(hipHostMalloc((void **)&h->h_flags, 16 * sizeof(int)));
    HIPCHECK(hipHostMalloc((void **)&h->h_ddtot, 64 * 16 * sizeof(unsigned long long)));
    h->d_overflow.ensure(16); h->d_ddtot.ensure(64 * 16);
    h->d_scal.ensure(1);
    h->d_slots.ensure((size_t)POLAR_NSLOT * POLAR_SLOT_STRIDE);
    h->have_device = true;
    return POLAR_OK;
  });
}

int polar_destroy(polar_handle *h) {
  if (!h) return POLAR_OK;
  if (h->have_device) {
    (void)hipSetDevice(h->device);
    (void)hipStreamSynchronize(h->stream);
    if (h->lj_stream) { (void)hipStreamSynchronize(h->lj_stream); (void)hipStreamDestroy(h->lj_stream); }
    for (hipEvent_t e : {h->ev_fork, h->ev_join, h->ev_lj0, h->ev_lj1, h->ev_dl0, h->ev_dl1, h->ev_mu_ready}) if (e) (void)hipEventDestroy(e);
    if (h->dl_stream) (void)hipStreamDestroy(h->dl_stream);
    if (h->up_stream) { (void)hipStreamSynchronize(h->up_stream); (void)hipStreamDestroy(h->up_stream); }
    if (h->ev_list_up) (void)hipEventDestroy(h->ev_list_up);
    for (int *p : h->h_nl_stage) (void)hipHostFree(p);
    for (hipEvent_t e : h->ev_nl) (void)hipEventDestroy(e);
    for (hipEvent_t e : h->ev_fchunk) if (e) (void)hipEventDestroy(e);
    h->d_ljpos.release(); h->d_ljaux.release(); h->d_tag.release(); h->d_nspecial.release(); h->d_special.release();
    h->d_ljcell_id.release(); h->d_ljcell_cnt.release(); h->d_ljcell_fill.release(); h->d_ljcell_first.release(); h->d_cutneighsq.release();
    h->d_xchg.release(); h->d_xidx.release();
    h->d_eatom.release(); h->d_vatom.release(); h->d_dd_r2.release(); h->d_fpol.release();
    h->d_x.release(); h->d_q.release(); h->d_alpha.release(); h->d_f.release(); h->d_ef.release(); h->d_F.release();
    h->d_mu.release(); h->d_mu0.release(); h->d_acc_x.release(); h->d_acc_f.release(); h->d_acc_g.release(); h->d_acc_dF.release(); h->d_acc_dG.release(); h->d_acc_part.release(); h->d_acc_state.release(); h->d_bflag.release(); h->d_rank.release(); h->d_dmu.release(); h->d_tab.release(); h->d_lj.release();
    h->d_type.release(); h->d_mol.release(); h->d_order.release(); h->d_pos.release(); h->d_ilist.release();
    h->d_numneigh.release(); h->d_neigh.release(); h->d_rows.release(); h->d_first.release();
    h->d_sym_first.release(); h->d_sym_cnt.release(); h->d_sym_fill.release(); h->d_sym_j.release();
    h->d_mol_s.release(); h->d_perm.release(); h->d_inv.release(); h->d_rows_orig.release(); h->d_ownrows.release(); h->d_ef_s.release(); h->d_scan_a.release(); h->d_scan_b.release(); h->d_gs_cnt.release(); h->d_T6.release(); h->d_Minv.release(); h->d_gsN.release(); h->d_gsAT.release(); h->d_cb.release(); h->d_gs_part.release();
    h->d_rec0.release(); h->d_rec1.release(); h->d_scal.release(); h->d_slots.release();
    h->d_cell_id.release(); h->d_cell_cnt.release(); h->d_cell_fill.release();
    h->d_nl_cnt.release(); h->d_dd_cnt.release(); h->d_dd_wrap.release(); h->d_lpdesc.release(); h->d_slot.release(); h->d_color_orig.release(); h->d_color_s.release(); h->d_trace.release(); h->d_cl_orig.release(); h->d_cl_cnt.release(); h->d_cl_wrap.release(); h->d_cl_tw.release(); h->d_cl_s.release(); h->d_nl_j.release(); h->d_dd_j.release();
    h->d_cell_first.release(); h->d_nl_first.release(); h->d_dd_first.release(); h->d_dd_s.release(); h->d_xq.release(); h->d_pos4.release();
    h->d_overflow.release(); h->d_ddtot.release();
    h->d_cadj.release(); h->d_cdeg.release(); h->d_ccnt.release(); h->d_cflags.release(); h->d_crelabel.release(); h->d_klist.release(); h->d_dbgf.release(); h->d_ulead.release(); h->d_udd_j.release(); h->d_upos.release(); h->d_unit.release(); h->d_udesc.release(); h->d_cprio.release(); h->d_cstat.release(); h->d_coff.release(); h->d_lp_pend.release(); h->d_lp_part.release();
    if (h->h_cflags) (void)hipHostFree(h->h_cflags);
    if (h->h_cstat) (void)hipHostFree(h->h_cstat);
    if (h->h_coff) (void)hipHostFree(h->h_coff);
    h->d_srec0.release(); h->d_srec1.release(); h->d_thdr.release(); h->d_trow.release(); h->d_un_j.release(); h->d_dd16.release(); h->d_pend.release();
    if (h->h_flags) (void)hipHostFree(h->h_flags);
    if (h->h_ddtot) (void)hipHostFree(h->h_ddtot);
    if (h->h_scal) (void)hipHostFree(h->h_scal);
    if (h->h_stage) (void)hipHostFree(h->h_stage);
    for (auto &e : h->ev) if (e) (void)hipEventDestroy(e);
    if (h->stream && h->own_stream) (void)hipStreamDestroy(h->stream);
  }
  delete h;
  return POLAR_OK;
}

const char *polar_last_error(const polar_handle *h) { return h ? h->err.c_str() : "null handle"; }
const char *polar_last_warning(const polar_handle *h) { return h ? h->warn.c_str() : ""; }

int polar_pair_settings(polar_handle *h, int narg, const char *const *arg) {
  return guarded(h, [&]() { h->ph.settings(narg, arg); h->colors_valid = false; h->nl_pitch = h->dd_pitch = 0; h->cl_pitch = 0; h->un_pitch = 0; h->pitch16 = 0; return POLAR_OK; });
}
int polar_pair_coeff(polar_handle *h, int ntypes, int narg, const char *const *arg) {
  return guarded(h, [&]() { h->ph.coeff(ntypes, narg, arg); return POLAR_OK; });
}
int polar_pair_modify(polar_handle *h, int narg, const char *const *arg) {
  return guarded(h, [&]() { h->ph.modify(narg, arg); return POLAR_OK; });
}
int polar_pair_init(polar_handle *h, double g_ewald, double qqrd2e, const double special_lj[4],
                    const double special_coul[4]) {
  return guarded(h, [&]() {
    PairHost &p = h->ph;
    p.init(g_ewald, qqrd2e, special_lj, special_coul);
    h->coul_set = false;
    if (h->have_device) {
      HIPCHECK(hipSetDevice(h->device));
      const double *t[7] = {p.lj1.data(), p.lj2.data(), p.lj3.data(), p.lj4.data(), p.offset.data(), p.cut_ljsq.data(), p.cutsq.data()};
      upload_types(h, p.ntypes, t);
      if (p.ncoultablebits == 0) {  // pair_modify table 0: the erfc polynomial everywhere, no table to wait for
        const double *c[8];
        double dummy = 0.0;
        for (int k = 0; k < 8; k++) c[k] = &dummy;
        upload_coul(h, g_ewald, qqrd2e, special_lj, special_coul, 0, 0, 0, 0.0, c);
      }
      // otherwise the Coulomb tables of Pair::init_tables (PS.cpp:851) arrive through polar_set_coul
    }
    return POLAR_OK;
  });
}
double polar_pair_cut(const polar_handle *h, int i, int j) {
  if (!h || !h->ph.inited || i < 1 || j < 1 || i > h->ph.ntypes || j > h->ph.ntypes) return -1.0;
  return std::sqrt(h->ph.cutsq[(size_t)i * h->ph.w() + j]);
}
double polar_pair_single(const polar_handle *h, double qi, double qj, int itype, int jtype, double rsq,
                         double factor_coul, double factor_lj, double *fforce) {
  double ff = 0.0, e;
  try {
    e = h->ph.single(qi, qj, itype, jtype, rsq, factor_coul, factor_lj, ff);
  } catch (const std::exception &ex) {  // no status channel in this signature: NaN + polar_last_error
    const_cast<polar_handle *>(h)->err = ex.what();
    e = ff = std::nan("");
  }
  if (fforce) *fforce = ff;
  return e;
}
const void *polar_pair_extract(const polar_handle *h, const char *name, int *dim) {
  if (!h || !name) return nullptr;
  if (dim) *dim = 0;
  if (strcmp(name, "cut_coul") == 0) return &h->ph.st.cut_coul;
  if (dim) *dim = 2;
  if (strcmp(name, "epsilon") == 0) return h->ph.epsilon.data();
  if (strcmp(name, "sigma") == 0) return h->ph.sigma.data();
  if (strcmp(name, "cut_lj") == 0) return h->ph.cut_lj.data();
  return nullptr;
}
int polar_get_settings(const polar_handle *h, polar_settings *out) {
  if (!h || !out) return POLAR_ERR_STATE;
  *out = h->ph.st;
  return POLAR_OK;
}
namespace {
const int kRestartMagic = 0x524C4F50;  // 'POLR' (little endian)
const int kRestartVersion = 2;         // 2: + deterministic, rccl_halo, polar_accel (int32) and polar_sor (double) behind the version-1 payload
const int kRestartPayloadV1 = 4 * (int)sizeof(double) + 10 * (int)sizeof(int);
const int kRestartPayload = kRestartPayloadV1 + 4 * (int)sizeof(int) + (int)sizeof(double);
}
int polar_restart_pack(const polar_handle *h, void *buf, int max_bytes) {
  if (!h) return POLAR_ERR_STATE;
  const int total = POLAR_RESTART_HEADER_BYTES + kRestartPayload;
  if (!buf) return total;
  if (max_bytes < total) return POLAR_ERR_INPUT;
  const polar_settings &st = h->ph.st;
  char *p = (char *)buf;
  const int head[3] = {kRestartMagic, kRestartVersion, kRestartPayload};
  memcpy(p, head, sizeof(head)); p += sizeof(head);
  const double d[4] = {st.polar_precision, st.polar_damp, st.polar_gamma, st.dd_cutoff};
  memcpy(p, d, sizeof(d)); p += sizeof(d);
  const int iv[10] = {st.iterations_max, st.damping_type, st.zodid, st.fixed_iteration, st.polar_gs, st.polar_gs_ranked,
                      st.use_previous, st.debug, st.device_neigh, st.restart_polar};
  memcpy(p, iv, sizeof(iv)); p += sizeof(iv);
  const int iv2[4] = {st.deterministic, st.rccl_halo, st.polar_accel, 0};
  memcpy(p, iv2, sizeof(iv2)); p += sizeof(iv2);
  memcpy(p, &st.polar_sor, sizeof(double));
  return total;
}
int polar_restart_unpack(polar_handle *h, const void *buf, int nbytes) {
  return guarded(h, [&]() {
    if (!buf || nbytes < POLAR_RESTART_HEADER_BYTES) throw InputError("not a polarization restart record");
    const char *p = (const char *)buf;
    int head[3];
    memcpy(head, p, sizeof(head)); p += sizeof(head);
    if (head[0] != kRestartMagic) throw InputError("not a polarization restart record");
    const bool v1 = head[1] == 1 && head[2] == kRestartPayloadV1, v2 = head[1] == kRestartVersion && head[2] == kRestartPayload;
    if (!(v1 || v2) || nbytes < POLAR_RESTART_HEADER_BYTES + head[2])
      throw InputError("polarization restart record of an unknown version or length");
    double d[4]; int iv[10];
    memcpy(d, p, sizeof(d)); p += sizeof(d);
    memcpy(iv, p, sizeof(iv)); p += sizeof(iv);
    polar_settings s = h->ph.st;  // the cutoffs stay as the stock record set them
    s.polar_precision = d[0]; s.polar_damp = d[1]; s.polar_gamma = d[2]; s.dd_cutoff = d[3];
    s.iterations_max = iv[0]; s.damping_type = iv[1]; s.zodid = iv[2]; s.fixed_iteration = iv[3]; s.polar_gs = iv[4];
    s.polar_gs_ranked = iv[5]; s.use_previous = iv[6]; s.debug = iv[7]; s.device_neigh = iv[8]; s.restart_polar = iv[9];
    if (v2) {  // (a version-1 record leaves these at the defaults in force)
      int iv2[4]; double sor;
      memcpy(iv2, p, sizeof(iv2)); p += sizeof(iv2);
      memcpy(&sor, p, sizeof(double));
      s.deterministic = iv2[0]; s.rccl_halo = iv2[1]; s.polar_accel = iv2[2]; s.polar_sor = sor;
      if (!(s.polar_sor > 0.0 && s.polar_sor < 2.0) || s.polar_accel < 0 || s.polar_accel > POLAR_ACCEL_MAX) throw InputError("polarization restart record holds an illegal polar_sor / polar_accel");
    }
    if (s.zodid && (s.polar_gs || s.polar_gs_ranked)) throw InputError("Zodid doesn't work with polar_gs or polar_gs_ranked");
    if (s.polar_gs && s.polar_gs_ranked) throw InputError("polar_gs and polar_gs_ranked are mutually exclusive");
    h->ph.st = s;
    h->colors_valid = false; h->nl_pitch = h->dd_pitch = 0; h->cl_pitch = 0; h->un_pitch = 0; h->pitch16 = 0;
    return POLAR_OK;
  });
}
int polar_set_settings(polar_handle *h, const polar_settings *s) {
  return guarded(h, [&]() {
    if (s->zodid && (s->polar_gs || s->polar_gs_ranked)) throw InputError("Zodid doesn't work with polar_gs or polar_gs_ranked");
    if (s->polar_gs && s->polar_gs_ranked) throw InputError("polar_gs and polar_gs_ranked are mutually exclusive");
    polar_settings v = *s;
    if (v.polar_sor == 0.0) v.polar_sor = 1.0;   // a zero-initialised struct means "the reference's update", not "freeze the dipoles"
    if (!(v.polar_sor > 0.0 && v.polar_sor < 2.0)) throw InputError("polar_sor must lie in (0, 2)");
    if (v.polar_accel < 0 || v.polar_accel > POLAR_ACCEL_MAX) throw InputError("polar_accel must lie in 0 .. 8");
    h->ph.st = v;
    h->colors_valid = false;
    return POLAR_OK;
  });
}
int polar_set_types(polar_handle *h, int ntypes, const double *lj1, const double *lj2, const double *lj3,
                    const double *lj4, const double *offset, const double *cut_ljsq, const double *cutsq) {
  return guarded(h, [&]() {
    HIPCHECK(hipSetDevice(h->device));
    const double *t[7] = {lj1, lj2, lj3, lj4, offset, cut_ljsq, cutsq};
    upload_types(h, ntypes, t);
    return POLAR_OK;
  });
}
int polar_set_coul(polar_handle *h, double g_ewald, double qqrd2e, const double special_lj[4],
                   const double special_coul[4], int nbits, int mask, int shift, double tabinnersq,
                   const double *rtable, const double *drtable, const double *ftable, const double *dftable,
                   const double *ctable, const double *dctable, const double *etable, const double *detable) {
  return guarded(h, [&]() {
    const double *c[8] = {rtable, drtable, ftable, dftable, ctable, dctable, etable, detable};
    h->ph.set_tables(nbits, mask, shift, tabinnersq, c);  // host copy: polar_pair_single reads the same table
    h->ph.g_ewald = g_ewald; h->ph.qqrd2e = qqrd2e;
    for (int k = 0; k < 4; k++) { h->ph.special_lj[k] = special_lj[k]; h->ph.special_coul[k] = special_coul[k]; }
    if (h->have_device) {
      HIPCHECK(hipSetDevice(h->device));
      double dummy = 0.0;
      if (nbits <= 0) for (int k = 0; k < 8; k++) c[k] = &dummy;
      upload_coul(h, g_ewald, qqrd2e, special_lj, special_coul, nbits > 0 ? nbits : 0, mask, shift, tabinnersq, c);
    }
    return POLAR_OK;
  });
}

int polar_set_box(polar_handle *h, const double boxlo[3], const double prd[3], const double tilt[3],
                  const int periodic[3], int triclinic) {
  return guarded(h, [&]() {
    // triclinic boxes: supported by the exact (all-pairs) kernels through the triclinic branch of
    // closest_image; the list mode's cell grid is orthogonal only
    const bool tri = triclinic != 0;
    Box nb{};
    nb.triclinic = tri ? 1 : 0;
    nb.xy = tri && tilt ? tilt[0] : 0.0;
    nb.xz = tri && tilt ? tilt[1] : 0.0;
    nb.yz = tri && tilt ? tilt[2] : 0.0;
    bool same = h->box_set;
    for (int k = 0; k < 3; k++) {
      if (!(prd[k] > 0.0) || !std::isfinite(prd[k])) throw InputError("box lengths must be positive and finite");
      nb.prd[k] = prd[k]; nb.half[k] = 0.5 * prd[k]; nb.inv[k] = 1.0 / prd[k]; nb.periodic[k] = periodic[k] ? 1 : 0;
      same = same && h->boxlo[k] == boxlo[k] && h->box.prd[k] == nb.prd[k] && h->box.periodic[k] == nb.periodic[k];
      h->boxlo[k] = boxlo[k];
    }
    same = same && h->box.triclinic == nb.triclinic && h->box.xy == nb.xy && h->box.xz == nb.xz && h->box.yz == nb.yz;
    h->box = nb;
    h->box_set = true;
    // a shim hands the box over every step: the colour phases (host-side colouring, rank metric) are rebuilt only when
    // the box really changed -- and on reneighbor steps, through polar_set_neighbors* / polar_build_neighbors
    if (!same) h->colors_valid = false;
    return POLAR_OK;
  });
}

int polar_set_atoms(polar_handle *h, int nlocal, int nghost, const double *x, const double *q, const double *alpha,
                    const int *type, const int *molecule) {
  return guarded(h, [&]() {
    need_device(h);
    HIPCHECK(hipSetDevice(h->device));
    if (nlocal < 0 || nghost < 0) throw InputError("negative atom count");
    const size_t nall = (size_t)nlocal + nghost;
    // the checks and the copies into the pinned staging area by the helper threads (a reneighbor step of an MD run hands all
    // five arrays over: single-threaded checks + five copies out of pageable memory cost 1.3 ms at 258k atoms)
    const size_t words = 3 * nall + nall + nall + (nall + 1) / 2 + (nall + 1) / 2 + 8;
    double *st = staging(h, std::max(words, 3 * nall + 6 * (size_t)nlocal));
    double *sx = st, *sq = sx + 3 * nall, *sa = sq + nall;
    int *stype = reinterpret_cast<int *>(sa + nall), *smol = stype + 2 * ((nall + 1) / 2);
    struct Ext { double lo[3], hi[3]; int bad, neg; char pad[56]; };
    const int nparts = nall < (1u << 14) ? 1 : std::min(8, HostPool::get().width());
    Ext ext[8];
    for (auto &e : ext) { for (int k = 0; k < 3; k++) { e.lo[k] = 1e300; e.hi[k] = -1e300; } e.bad = e.neg = 0; }
    const size_t per = ((nall + nparts - 1) / nparts + 3) / 4 * 4;
    HostPool::get().run([&](int part) {
      Ext &e = ext[part];
      const size_t a0 = std::min(nall, (size_t)part * per), a1 = std::min(nall, ((size_t)part + 1) * per);
      memcpy(sx + 3 * a0, x + 3 * a0, 3 * (a1 - a0) * sizeof(double));
      memcpy(sq + a0, q + a0, (a1 - a0) * sizeof(double));
      memcpy(sa + a0, alpha + a0, (a1 - a0) * sizeof(double));
      memcpy(stype + a0, type + a0, (a1 - a0) * sizeof(int));
      memcpy(smol + a0, molecule + a0, (a1 - a0) * sizeof(int));
      double lo[12], hi[12], acc = 0.0, amin = 0.0;
      for (int j = 0; j < 12; j++) { lo[j] = 1e300; hi[j] = -1e300; }
      size_t a = a0;
      for (; a + 4 <= a1; a += 4) {
        const double *p = x + 3 * a;
        for (int j = 0; j < 12; j++) { const double v = p[j]; lo[j] = v < lo[j] ? v : lo[j]; hi[j] = v > hi[j] ? v : hi[j]; acc += v - v; }
      }
      for (; a < a1; a++)
        for (int k = 0; k < 3; k++) { const double v = x[3 * a + k]; lo[k] = v < lo[k] ? v : lo[k]; hi[k] = v > hi[k] ? v : hi[k]; acc += v - v; }
      for (size_t k = a0; k < a1; k++) amin = alpha[k] < amin ? alpha[k] : amin;
      for (int j = 0; j < 12; j++) { e.lo[j % 3] = std::min(e.lo[j % 3], lo[j]); e.hi[j % 3] = std::max(e.hi[j % 3], hi[j]); }
      if (!(acc == 0.0)) e.bad = 1;
      if (amin < 0.0) e.neg = 1;
    }, nparts);
    for (const auto &e : ext) {
      if (e.bad) throw InputError("non-finite atom coordinate");
      if (e.neg) throw InputError("Invalid value in set command");  // src/set.cpp:174-184 rejects negatives
    }
    if (nlocal != h->nlocal) { h->colors_valid = false; h->mu_resident = false; }
    else if (h->colors_valid) {  // the rows of the colour phases are the polarizable atoms: same atoms, same rows
      for (int k = 0; k < nlocal; k++)
        if ((alpha[k] != 0.0) != (h->halpha[k] != 0.0)) { h->colors_valid = false; break; }
    }
    h->nlocal = nlocal; h->nghost = nghost;
    h->d_x.ensure(3 * nall + 3); h->d_q.ensure(nall + 1); h->d_alpha.ensure(nall + 1); h->d_type.ensure(nall + 1); h->d_mol.ensure(nall + 1);
    hipStream_t s = h->stream;
    HIPCHECK(hipMemcpyAsync(h->d_x.p, sx, 3 * nall * sizeof(double), hipMemcpyHostToDevice, s));
    HIPCHECK(hipMemcpyAsync(h->d_q.p, sq, nall * sizeof(double), hipMemcpyHostToDevice, s));
    HIPCHECK(hipMemcpyAsync(h->d_alpha.p, sa, nall * sizeof(double), hipMemcpyHostToDevice, s));
    HIPCHECK(hipMemcpyAsync(h->d_type.p, stype, nall * sizeof(int), hipMemcpyHostToDevice, s));
    HIPCHECK(hipMemcpyAsync(h->d_mol.p, smol, nall * sizeof(int), hipMemcpyHostToDevice, s));
    h->hx.assign(x, x + 3 * (size_t)nlocal);
    h->halpha.assign(alpha, alpha + nlocal);
    for (int k = 0; k < 3; k++) { h->bbox_lo[k] = 1e300; h->bbox_hi[k] = -1e300; }
    for (const auto &e : ext)
      for (int k = 0; k < 3; k++) { h->bbox_lo[k] = std::min(h->bbox_lo[k], e.lo[k]); h->bbox_hi[k] = std::max(h->bbox_hi[k], e.hi[k]); }
    HIPCHECK(hipStreamSynchronize(s));
    h->atoms_set = true;
    h->mu_host_in_sync = false;  // the atoms may sit in a new order
    return POLAR_OK;
  });
}

int polar_set_positions(polar_handle *h, int nlocal, int nghost, const double *x) {
  return guarded(h, [&]() {
    need_device(h);
    HIPCHECK(hipSetDevice(h->device));
    if (!h->atoms_set || nlocal != h->nlocal || nghost != h->nghost) throw InputError("polar_set_positions: atom counts differ from the last polar_set_atoms");
    if (!x) throw InputError("polar_set_positions: null pointer");
    const size_t nall = (size_t)nlocal + nghost;
    // through the pinned staging area: a pageable source is staged by the runtime in small pieces at a third of the rate
    // (the caller's array is NOT registered in place, as VERDICT r2 suggested: a registration outlives a free + malloc that
    //  returns the same address with other pages behind it -- LAMMPS re-creates its per-atom arrays between runs -- and the
    //  next DMA would then land in the stale pages.  A copy by four threads costs 0.1 ms at 260k atoms.)
    double *st = staging(h, 3 * nall + 6 * (size_t)nlocal);
    // one pass: the copy, the check polar_set_atoms makes (non-finite coordinates must not reach the cell build) and the
    // bounding box polar_build_neighbors lays its grid over -- so that polar_set_positions -> polar_build_neighbors is as
    // good as polar_set_atoms -> polar_build_neighbors
    struct Ext { double lo[3], hi[3]; int bad; char pad[64]; };
    const int nparts = nall < (1u << 14) ? 1 : std::min(8, HostPool::get().width());
    Ext ext[8];
    for (auto &e : ext) { for (int k = 0; k < 3; k++) { e.lo[k] = 1e300; e.hi[k] = -1e300; } e.bad = 0; }
    const size_t per = ((nall + nparts - 1) / nparts + 3) / 4 * 4;
    HostPool::get().run([&](int part) {
      Ext &e = ext[part];
      const size_t a0 = std::min(nall, (size_t)part * per), a1 = std::min(nall, ((size_t)part + 1) * per);
      memcpy(st + 3 * a0, x + 3 * a0, 3 * (a1 - a0) * sizeof(double));
      // four atoms = twelve doubles at a time: twelve independent running minima / maxima (the compiler keeps them in vector
      // registers; one accumulator per coordinate, fed with a stride of three, ran at a fifth of the copy's rate)
      double lo[12], hi[12], acc = 0.0;
      for (int j = 0; j < 12; j++) { lo[j] = 1e300; hi[j] = -1e300; }
      size_t a = a0;
      for (; a + 4 <= a1; a += 4) {
        const double *p = x + 3 * a;
        for (int j = 0; j < 12; j++) { const double v = p[j]; lo[j] = v < lo[j] ? v : lo[j]; hi[j] = v > hi[j] ? v : hi[j]; acc += v - v; }
      }
      for (; a < a1; a++)
        for (int k = 0; k < 3; k++) { const double v = x[3 * a + k]; lo[k] = v < lo[k] ? v : lo[k]; hi[k] = v > hi[k] ? v : hi[k]; acc += v - v; }
      for (int j = 0; j < 12; j++) { e.lo[j % 3] = std::min(e.lo[j % 3], lo[j]); e.hi[j % 3] = std::max(e.hi[j % 3], hi[j]); }
      if (!(acc == 0.0)) e.bad = 1;   // (v - v is 0 for a finite v, NaN for an infinite or NaN one)
    }, nparts);
    for (int k = 0; k < 3; k++) { h->bbox_lo[k] = 1e300; h->bbox_hi[k] = -1e300; }
    for (const auto &e : ext) {
      if (e.bad) throw InputError("non-finite atom coordinate");
      for (int k = 0; k < 3; k++) { h->bbox_lo[k] = std::min(h->bbox_lo[k], e.lo[k]); h->bbox_hi[k] = std::max(h->bbox_hi[k], e.hi[k]); }
    }
    if (h->hx.size() == 3 * (size_t)nlocal) memcpy(h->hx.data(), st, 3 * (size_t)nlocal * sizeof(double));
    HIPCHECK(hipMemcpyAsync(h->d_x.p, st, 3 * nall * sizeof(double), hipMemcpyHostToDevice, h->stream));
    // (no synchronisation: the staging area is next written by the downloads of the compute call, which wait for the stream)
    return POLAR_OK;
  });
}

int polar_set_neighbors_csr(polar_handle *h, int inum, const int *ilist, const int *numneigh,
                            const long long *firstneigh, const int *neigh) {
  return guarded(h, [&]() {
    need_device(h);
    HIPCHECK(hipSetDevice(h->device));
    if (!h->atoms_set) throw std::runtime_error("polar_set_neighbors before polar_set_atoms");
    if (inum < 0 || (inum > 0 && (!ilist || !numneigh || !firstneigh || !neigh))) throw InputError("polar_set_neighbors_csr: null pointer");
    for (int ii = 0; ii < inum; ii++) {
      const int i = ilist[ii];
      if (i < 0 || i >= h->nlocal) throw InputError("neighbor list row index out of range");
      if (firstneigh[i] < 0) throw InputError("negative neighbor count/offset");
    }
    upload_neighbor_rows(h, inum, ilist, numneigh, [&](int i) { return neigh + firstneigh[i]; });
    return POLAR_OK;
  });
}

int polar_build_neighbors(polar_handle *h, const double *cutneighsq, const int *tag, const int *nspecial,
                          const int *special, int maxspecial, const int special_flag[4], int exclude_molecule_intra) {
  return guarded(h, [&]() {
    need_device(h);
    HIPCHECK(hipSetDevice(h->device));
    if (h->list_up_pending) { HIPCHECK(hipStreamSynchronize(h->up_stream)); h->list_up_pending = false; }   // (an uploaded list still on its way into the arrays rebuilt here)
    if (!h->atoms_set || !h->box_set) throw std::runtime_error("polar_build_neighbors before polar_set_box/polar_set_atoms");
    if (!h->types_set) throw std::runtime_error("polar_build_neighbors before the pair tables were set");
    if (!cutneighsq || !special_flag) throw InputError("polar_build_neighbors: null cutneighsq/special_flag");
    if ((nspecial == nullptr) != (special == nullptr) || (special && maxspecial <= 0)) throw InputError("polar_build_neighbors: nspecial/special/maxspecial are inconsistent");
    const int n = h->nlocal, nall = h->nlocal + h->nghost, w = h->ntypes + 1;
    // a sharded handle (polar_set_row_range) holds [own | halo | ghosts]: only the own atoms get list rows -- the halo
    // atoms are somebody else's rows and would otherwise be counted into this rank's energies and virial
    const int nown = own_n(h);
    hipStream_t s = h->stream;
    double cutmax2 = 0.0;
    for (int a = 1; a < w; a++) for (int b = 1; b < w; b++) cutmax2 = std::max(cutmax2, cutneighsq[a * w + b]);
    if (!(cutmax2 > 0.0)) throw InputError("polar_build_neighbors: no positive neighbor cutoff");
    const double cutmax = std::sqrt(cutmax2);
    // grid over the bounding box of locals + ghosts (recorded by polar_set_atoms), cell edge >= cutmax / 2
    LJGrid g;
    long long ncell = 1;
    for (int k = 0; k < 3; k++) {
      const double ext = std::max(h->bbox_hi[k] - h->bbox_lo[k], 1e-9);
      int nc = (int)std::floor(ext / (0.5 * cutmax));
      nc = std::max(1, std::min(nc, 512));
      g.nc[k] = nc; g.lo[k] = h->bbox_lo[k]; g.inv[k] = nc / ext;
      ncell *= nc;
    }
    h->d_ljcell_id.ensure(nall + 1); h->d_ljcell_cnt.ensure(ncell + 1); h->d_ljcell_fill.ensure(ncell + 1); h->d_ljcell_first.ensure(ncell + 2);
    h->d_ljpos.ensure(nall + 1); h->d_ljaux.ensure(nall + 1); h->d_cutneighsq.ensure((size_t)w * w);
    HIPCHECK(hipMemcpyAsync(h->d_cutneighsq.p, cutneighsq, (size_t)w * w * sizeof(double), hipMemcpyHostToDevice, s));
    const int *d_tag = nullptr, *d_nsp = nullptr, *d_sp = nullptr;
    if (tag) { h->d_tag.ensure(nall + 1); HIPCHECK(hipMemcpyAsync(h->d_tag.p, tag, (size_t)nall * sizeof(int), hipMemcpyHostToDevice, s)); d_tag = h->d_tag.p; }
    if (special) {
      if (!tag) throw InputError("polar_build_neighbors: special lists need the atom tags");
      h->d_nspecial.ensure(3 * (size_t)n + 3); h->d_special.ensure((size_t)n * maxspecial + 1);
      HIPCHECK(hipMemcpyAsync(h->d_nspecial.p, nspecial, 3 * (size_t)n * sizeof(int), hipMemcpyHostToDevice, s));
      HIPCHECK(hipMemcpyAsync(h->d_special.p, special, (size_t)n * maxspecial * sizeof(int), hipMemcpyHostToDevice, s));
      d_nsp = h->d_nspecial.p; d_sp = h->d_special.p;
    }
    HIPCHECK(hipMemsetAsync(h->d_ljcell_cnt.p, 0, (ncell + 1) * sizeof(int), s));
    HIPCHECK(hipMemsetAsync(h->d_ljcell_fill.p, 0, (ncell + 1) * sizeof(int), s));
    k_lj_cell_count<<<nblk(nall, 256), 256, 0, s>>>(nall, h->d_x.p, g, h->d_ljcell_id.p, h->d_ljcell_cnt.p);
    launch_scan(ncell, h->d_ljcell_cnt.p, h->d_ljcell_first.p, h->d_scan_a, s);
    k_lj_cell_fill<<<nblk(nall, 256), 256, 0, s>>>(nall, h->d_ljcell_id.p, h->d_ljcell_first.p, h->d_ljcell_fill.p, h->d_x.p,
                                                    h->d_type.p, h->d_mol.p, d_tag, h->d_ljpos.p, h->d_ljaux.p);
    if (h->lj_pitch == 0 && getenv("POLAR_INIT_PITCH")) h->lj_pitch = ((std::max(64, atoi(getenv("POLAR_INIT_PITCH"))) + 63) / 64) * 64;  // tests: force the overflow path
    if (h->lj_pitch == 0) {  // first build: 1.3x the mean sphere population, rounded to 64
      const double vol = std::max((h->bbox_hi[0] - h->bbox_lo[0]) * (h->bbox_hi[1] - h->bbox_lo[1]) * (h->bbox_hi[2] - h->bbox_lo[2]), 1e-9);
      const double mean = 4.18879 * cutmax * cutmax2 * nall / vol;
      h->lj_pitch = (((long long)(1.3 * mean) + 64) / 64 + 1) * 64;
    }
    h->d_numneigh.ensure(n + 1); h->d_ilist.ensure(n + 1); h->d_first.ensure(n + 1);
    h->dev_typed = h->lj_typed && nall < (1 << 24) && h->ntypes < 64;
    const size_t lds = (size_t)w * w * sizeof(double);
    if (lds > 64 * 1024) throw InputError("too many atom types for the LDS-resident cutoff table");
    for (int attempt = 0;; attempt++) {
      h->d_neigh.ensure((size_t)std::max(n, 1) * h->lj_pitch + 64);
      HIPCHECK(hipMemsetAsync(h->d_overflow.p, 0, 16 * sizeof(int), s));
      HIPCHECK(hipMemsetAsync(h->d_ddtot.p, 0, 64 * 16 * sizeof(unsigned long long), s));
      k_lj_nl_build<<<nblk(nown, POLAR_ROWS_PER_BLOCK), POLAR_BLOCK, lds, s>>>(
          own_lo(h), nown, h->ntypes, h->d_x.p, h->d_type.p, h->d_mol.p, h->d_ljpos.p, h->d_ljaux.p, g, h->d_ljcell_first.p,
          h->d_cutneighsq.p, h->box, exclude_molecule_intra, d_nsp, d_sp, maxspecial, special_flag[1], special_flag[2],
          special_flag[3], h->lj_pitch, h->d_numneigh.p, h->d_neigh.p, h->d_overflow.p, h->d_ddtot.p, h->dev_typed ? 1 : 0);
      HIPCHECK(hipMemcpyAsync(h->h_flags, h->d_overflow.p, sizeof(int), hipMemcpyDeviceToHost, s));
      HIPCHECK(hipMemcpyAsync(h->h_ddtot, h->d_ddtot.p, 64 * 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
      HIPCHECK(hipStreamSynchronize(s));
      HIPCHECK(hipGetLastError());
      if (h->h_flags[0] == 0) break;
      if (attempt >= 3) throw std::runtime_error("polar_build_neighbors: row pitch overflow persists");
      h->lj_pitch = ((long long)(1.25 * h->h_flags[0]) / 64 + 1) * 64;
    }
    unsigned long long tot = 0;
    for (int k = 0; k < 64; k++) tot += h->h_ddtot[16 * k];
    h->h_flags[0] = 0;
    k_lj_rows<<<nblk(nown, 256), 256, 0, s>>>(own_lo(h), nown, h->lj_pitch, h->d_ilist.p, h->d_first.p);
    h->inum = nown; h->nneigh = (long long)tot;
    h->neigh_set = true; h->sym_valid = false;
    if (h->colors_valid) h->colors_recheck = true;
    h->device_list = true; h->full_list = 1;
    return POLAR_OK;
  });
}

int polar_set_neighbors(polar_handle *h, int inum, const int *ilist, const int *numneigh, int *const *firstneigh) {
  return guarded(h, [&]() {
    need_device(h);
    HIPCHECK(hipSetDevice(h->device));
    if (!h->atoms_set) throw std::runtime_error("polar_set_neighbors before polar_set_atoms");
    if (inum < 0 || (inum > 0 && (!ilist || !numneigh || !firstneigh))) throw InputError("polar_set_neighbors: null pointer");
    // LAMMPS' paged int** rows, packed straight into the pinned upload buffers (no flat copy in between)
    upload_neighbor_rows(h, inum, ilist, numneigh, [&](int i) { return (const int *)firstneigh[i]; });
    return POLAR_OK;
  });
}

int polar_compute(polar_handle *h, int eflag, int vflag, double *f, double *mu, double *ef_static, polar_result *out) {
  return guarded(h, [&]() {
    if (!f || !mu || !out) throw std::runtime_error("polar_compute: null output pointer");
    if (eflag / 2 || vflag / 4) throw InputError("per-atom tallies (eflag & 2, vflag & 4) are returned by polar_compute_peratom");
    need_device(h);
    HIPCHECK(hipSetDevice(h->device));
    const size_t n = h->nlocal, nall = (size_t)h->nlocal + h->nghost;
    // results come back through one pinned staging area (pageable destinations cost ~3x the PCIe time):
    // [f nall*3 | mu n*3 | ef n*3]; mu and ef leave as soon as the solve is over (phase_finish, their own stream, beside the
    // force kernel), f after the force kernel; the host adds the forces in while whatever is left still travels
    double *st = staging(h, 3 * nall + 6 * n);
    struct Early {  // (cleared on every way out: the stepwise interface shares phase_finish)
      polar_handle *h;
      ~Early() { h->early_mu = h->early_ef = h->user_mu = h->user_ef = nullptr; }
    } early{h};
    h->early_mu = st + 3 * nall;
    h->early_ef = ef_static ? st + 3 * nall + 3 * n : nullptr;
    h->user_mu = mu; h->user_ef = ef_static;
    const bool dbg = getenv("POLAR_DEBUG") != nullptr;
    auto t0 = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
      if (!dbg) return;
      const auto t1 = std::chrono::steady_clock::now();
      fprintf(stderr, "[polar] compute: %-22s %7.3f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t0).count());
      t0 = t1;
    };
    int rc = do_compute(h, eflag, vflag, mu, out);
    if (rc < 0) return rc;
    lap("device step + sync");
    // the forces: four pieces, each added into the caller's array while the next one travels
    const size_t tot = 3 * nall, piece = (tot + 3) / 4;
    for (int k = 0; k < 4; k++) {
      const size_t a = std::min(tot, k * piece), b = std::min(tot, (k + 1) * piece);
      if (b > a) HIPCHECK(hipMemcpyAsync(st + a, h->d_f.p + a, (b - a) * sizeof(double), hipMemcpyDeviceToHost, h->stream));
      HIPCHECK(hipEventRecord(h->ev_fchunk[k], h->stream));
    }
    for (int k = 0; k < 4; k++) {
      const size_t a = std::min(tot, k * piece), b = std::min(tot, (k + 1) * piece);
      HIPCHECK(hipEventSynchronize(h->ev_fchunk[k]));
      host_chunks(b - a, [&](size_t lo, size_t hi) { for (size_t i = a + lo; i < a + hi; i++) f[i] += st[i]; });
    }
    lap("f arrived and added");
    h->mu_host_in_sync = true;
    return rc;
  });
}

int polar_compute_peratom(polar_handle *h, int eflag, int vflag, double *f, double *mu, double *ef_static,
                          double *eatom, double *vatom, polar_result *out) {
  return guarded(h, [&]() {
    if (!f || !mu || !out) throw std::runtime_error("polar_compute_peratom: null output pointer");
    if ((eflag / 2 && !eatom) || (vflag / 4 && !vatom)) throw InputError("polar_compute_peratom: eflag & 2 needs eatom, vflag & 4 needs vatom");
    need_device(h);
    HIPCHECK(hipSetDevice(h->device));
    int rc = do_compute(h, eflag, vflag, mu, out);
    if (rc < 0) return rc;
    const size_t n = h->nlocal, nall = (size_t)h->nlocal + h->nghost;
    auto add_from = [&](const double *dev, double *host, size_t cnt) {
      h->h_tmp.resize(cnt);
      HIPCHECK(hipMemcpy(h->h_tmp.data(), dev, cnt * sizeof(double), hipMemcpyDeviceToHost));
      for (size_t k = 0; k < cnt; k++) host[k] += h->h_tmp[k];
    };
    add_from(h->d_f.p, f, 3 * nall);
    if (eflag / 2) add_from(h->d_eatom.p, eatom, nall);
    if (vflag / 4) add_from(h->d_vatom.p, vatom, 6 * nall);
    HIPCHECK(hipMemcpy(mu, h->d_mu.p, 3 * n * sizeof(double), hipMemcpyDeviceToHost));
    if (ef_static) HIPCHECK(hipMemcpy(ef_static, h->d_ef.p, 3 * n * sizeof(double), hipMemcpyDeviceToHost));
    h->mu_host_in_sync = true;
    return rc;
  });
}

int polar_compute_resident(polar_handle *h, int eflag, int vflag, polar_result *out) {
  return guarded(h, [&]() {
    if (!out) throw std::runtime_error("polar_compute_resident: null result pointer");
    need_device(h);
    HIPCHECK(hipSetDevice(h->device));
    return do_compute(h, eflag, vflag, nullptr, out);
  });
}

void *polar_dev_ptr(polar_handle *h, const char *name) {
  if (!h || !name || !h->have_device) return nullptr;
  if (strcmp(name, "f") == 0) return h->d_f.p;
  if (strcmp(name, "mu") == 0) return h->d_mu.p;
  if (strcmp(name, "ef_static") == 0) return h->d_ef.p;
  if (strcmp(name, "x") == 0) return h->d_x.p;
  if (strcmp(name, "eatom") == 0) return h->d_eatom.p;
  if (strcmp(name, "vatom") == 0) return h->d_vatom.p;
  return nullptr;
}
int polar_download(polar_handle *h, const char *name, double *dst, long long n) {
  return guarded(h, [&]() {
    need_device(h);
    void *p = polar_dev_ptr(h, name);
    if (!p) throw std::runtime_error("polar_download: unknown array name");
    HIPCHECK(hipMemcpy(dst, p, (size_t)n * sizeof(double), hipMemcpyDeviceToHost));
    return POLAR_OK;
  });
}
int polar_upload_mu(polar_handle *h, const double *mu, long long n) {
  return guarded(h, [&]() {
    need_device(h);
    h->d_mu.ensure((size_t)n);
    HIPCHECK(hipMemcpy(h->d_mu.p, mu, (size_t)n * sizeof(double), hipMemcpyHostToDevice));
    h->mu_resident = true;
    return POLAR_OK;
  });
}

/* ---- stepwise / sharded interface (multi-GPU driver, parallel.py) --------------------------- */
int polar_set_stream(polar_handle *h, void *hip_stream) {
  return guarded(h, [&]() {
    need_device(h);
    if (h->own_stream && h->stream) { HIPCHECK(hipStreamSynchronize(h->stream)); HIPCHECK(hipStreamDestroy(h->stream)); }
    h->stream = (hipStream_t)hip_stream;
    h->own_stream = false;
    return POLAR_OK;
  });
}
int polar_set_row_range(polar_handle *h, int lo, int hi) {
  return guarded(h, [&]() {
    if (lo < 0 || (hi >= 0 && hi < lo)) throw InputError("bad row range");
    h->row_lo = lo; h->row_hi = hi;
    return POLAR_OK;
  });
}
int polar_set_global_count(polar_handle *h, long long natoms) {
  if (!h) return POLAR_ERR_STATE;
  if (natoms < 0 || natoms > 2147483647LL) return fail(h, POLAR_ERR_INPUT, "bad global atom count");
  h->global_count = natoms;
  return POLAR_OK;
}
int polar_get_debug_trace(polar_handle *h, double *u_polar, int max) {
  if (!h) return POLAR_ERR_STATE;
  int n = 0;
  int rc = guarded(h, [&]() {
    need_device(h);
    n = std::min(h->ntrace, max);
    if (n > 0) HIPCHECK(hipMemcpy(u_polar, h->d_trace.p, (size_t)n * sizeof(double), hipMemcpyDeviceToHost));
    return (int)POLAR_OK;
  });
  return rc < 0 ? rc : n;
}
int polar_get_colors(polar_handle *h, int *color, int n) {
  if (!h) return POLAR_ERR_STATE;
  int nc = 0;
  int rc = guarded(h, [&]() {
    need_device(h);
    if (!color || n < 0) throw InputError("polar_get_colors: null pointer");
    const int m = std::min(n, h->nlocal);
    for (int k = 0; k < n; k++) color[k] = -1;
    if (!h->colors_valid || (int)h->color_off.size() < 2) return (int)POLAR_OK;
    nc = (int)h->color_off.size() - 1;
    if (h->d_color_orig.p && h->d_color_orig.cap >= (size_t)m && m > 0)   // colours by original index (both colourings leave them here)
      HIPCHECK(hipMemcpy(color, h->d_color_orig.p, (size_t)m * sizeof(int), hipMemcpyDeviceToHost));
    return (int)POLAR_OK;
  });
  return rc < 0 ? rc : nc;
}
int polar_get_debug_forces(polar_handle *h, double *out6) {
  return guarded(h, [&]() {
    need_device(h);
    if (!out6) throw InputError("polar_get_debug_forces: null pointer");
    for (int k = 0; k < 6; k++) out6[k] = 0.0;
    if (!h->ph.st.debug || !h->d_dbgf.p) return 0;
    HIPCHECK(hipMemcpy(out6, h->d_dbgf.p, 6 * sizeof(double), hipMemcpyDeviceToHost));
    return 1;
  });
}
int polar_set_newton(polar_handle *h, int newton_pair) {
  if (!h) return POLAR_ERR_STATE;
  h->newton_pair = newton_pair ? 1 : 0;
  return POLAR_OK;
}
int polar_set_list_style(polar_handle *h, int full) {
  if (!h) return POLAR_ERR_STATE;
  h->user_full_list = full ? 1 : 0;
  if (!h->device_list) h->full_list = h->user_full_list;
  return POLAR_OK;
}
int polar_step_begin(polar_handle *h, int eflag, int vflag) {
  return guarded(h, [&]() {
    step_begin_lists(h, eflag, vflag);
    step_begin_finish(h);
    return POLAR_OK;
  });
}
int polar_step_sweep_phase(polar_handle *h, int color, int part) {
  return guarded(h, [&]() {
    if (!h->in_step) throw std::runtime_error("polar_step_sweep_phase outside polar_step_begin/finish");
    const polar_settings &st = h->ph.st;
    if (!(st.dd_cutoff > 0.0 && (st.polar_gs || st.polar_gs_ranked) && h->sweep_kernel == 2)) throw InputError("polar_step_sweep_phase: colour phases exist for the Gauss-Seidel sweep of list mode only");
    if (color < 0 || color >= (int)h->color_off.size() - 1 || part < 0 || part > 2) throw InputError("polar_step_sweep_phase: bad colour or part");
    sweep_phase(h, color, part);
    return POLAR_OK;
  });
}
int polar_set_colors(polar_handle *h, const int *color, int n) {
  return guarded(h, [&]() {
    if (!color && n != 0) throw InputError("polar_set_colors: null pointer");
    if (n == 0) { h->user_colors.clear(); h->user_colors_clashed = false; h->colors_valid = false; return POLAR_OK; }
    if (!h->atoms_set || n != h->nlocal) throw InputError("polar_set_colors: one colour per local atom of the last polar_set_atoms");
    for (int k = 0; k < n; k++) {
      if (color[k] < -1 || color[k] > 63) throw InputError("polar_set_colors: colours are -1 (none) or 0 .. 63");
      if ((color[k] >= 0) != (h->halpha[k] != 0.0)) throw InputError("polar_set_colors: exactly the polarizable atoms carry a colour");
    }
    h->user_colors.assign(color, color + n);
    h->user_colors_clashed = false;
    h->colors_valid = false;
    return POLAR_OK;
  });
}
int polar_set_positions_range(polar_handle *h, int lo, int hi, const double *x) {
  return guarded(h, [&]() {
    need_device(h);
    HIPCHECK(hipSetDevice(h->device));
    if (!h->atoms_set || lo < 0 || hi < lo || hi > h->nlocal + h->nghost) throw InputError("polar_set_positions_range: range outside the atoms of the last polar_set_atoms");
    if (hi > lo && !x) throw InputError("polar_set_positions_range: null pointer");
    const size_t cnt = 3 * (size_t)(hi - lo);
    for (size_t k = 0; k < cnt; k++) if (!(x[k] - x[k] == 0.0)) throw InputError("non-finite atom coordinate");
    double *st = staging(h, 3 * ((size_t)h->nlocal + h->nghost) + 6 * (size_t)h->nlocal);
    if (cnt) {
      memcpy(st, x, cnt * sizeof(double));
      HIPCHECK(hipMemcpyAsync(h->d_x.p + 3 * (size_t)lo, st, cnt * sizeof(double), hipMemcpyHostToDevice, h->stream));
      for (int a = lo; a < std::min(hi, h->nlocal); a++) for (int k = 0; k < 3; k++) h->hx[3 * (size_t)a + k] = x[3 * (size_t)(a - lo) + k];
      for (int a = lo; a < hi; a++) for (int k = 0; k < 3; k++) { const double v = x[3 * (size_t)(a - lo) + k]; h->bbox_lo[k] = std::min(h->bbox_lo[k], v); h->bbox_hi[k] = std::max(h->bbox_hi[k], v); }
      HIPCHECK(hipStreamSynchronize(h->stream));   // (the staging area is the caller's to reuse)
    }
    return POLAR_OK;
  });
}
int polar_step_sweep(polar_handle *h) {
  return guarded(h, [&]() {
    if (!h->in_step) throw std::runtime_error("polar_step_sweep outside polar_step_begin/finish");
    sweep_once(h, false);
    return POLAR_OK;
  });
}
int polar_step_sweep_part(polar_handle *h, int part, int nparts) {
  return guarded(h, [&]() {
    if (!h->in_step) throw std::runtime_error("polar_step_sweep_part outside polar_step_begin/finish");
    if (nparts < 1 || part < 0 || part >= nparts) throw InputError("polar_step_sweep_part: bad part");
    const polar_settings &st = h->ph.st;
    if (nparts > 1 && !(st.dd_cutoff > 0.0 && (st.polar_gs || st.polar_gs_ranked) && (h->sweep_kernel == 2 || h->sweep_kernel == 4)))
      throw InputError("polar_step_sweep_part: parts exist for the colour-phase Gauss-Seidel sweep of list mode only");
    struct Window {  // restored on every way out: a failed launch must not leave the next full sweep truncated
      polar_handle *h;
      ~Window() { h->part_k = 0; h->part_n = 1; }
    } window{h};
    h->part_k = part; h->part_n = nparts;
    sweep_once(h, false);
    return POLAR_OK;
  });
}
int polar_step_sweep_end_n(polar_handle *h, const double *dev_global_change, int count) {
  return guarded(h, [&]() {
    if (!h->in_step) throw std::runtime_error("polar_step_sweep_end outside polar_step_begin/finish");
    if (count < 1) throw InputError("polar_step_sweep_end_n: count must be >= 1");
    const polar_settings &st = h->ph.st;
    const bool gs = st.polar_gs || st.polar_gs_ranked;
    if (count > 1 && !(gs && st.fixed_iteration)) throw InputError("polar_step_sweep_end_n: count > 1 needs fixed-iteration Gauss-Seidel");
    k_solver_step<<<1, POLAR_NSLOT, 0, h->stream>>>(h->d_scal.p, h->d_slots.p, norm_count(h), st.fixed_iteration, st.iterations_max, st.polar_precision,
                                          gs ? 0 : 1, dev_global_change, count, det_part(h), det_npart(h));
    return POLAR_OK;
  });
}
int polar_step_sweep_end(polar_handle *h, const double *dev_global_change) { return polar_step_sweep_end_n(h, dev_global_change, 1); }
int polar_step_state(polar_handle *h, int *done, int *iterations, int *status) {
  return guarded(h, [&]() {
    need_device(h);
    read_scal(h);
    if (done) *done = h->h_scal->done;
    if (iterations) *iterations = h->h_scal->iterations;
    if (status) *status = h->h_scal->status;
    return POLAR_OK;
  });
}
int polar_step_finish(polar_handle *h, polar_result *out) {
  return guarded(h, [&]() {
    if (!h->in_step) throw std::runtime_error("polar_step_finish without polar_step_begin");
    memset(out, 0, sizeof(*out));
    int rc = phase_finish(h, out);
    out->ncolors = (h->ph.st.polar_gs || h->ph.st.polar_gs_ranked) ? (tile_mode(h) ? (int)h->tile_launches.size() : (int)h->color_off.size() - 1) : 0;
    h->in_step = false;
    if (grow_pitches(h)) {  // the driver must redo the step (all ranks see their own flag)
      out->status = POLAR_RETRY_STEP;
      h->warn = "neighbor list pitch overflow: pitch enlarged, repeat the step";
      return POLAR_RETRY_STEP;
    }
    return rc;
  });
}
int polar_mu_gather(polar_handle *h, long long lo, long long hi, double *dev_dst) {
  return guarded(h, [&]() {
    need_device(h);
    if (hi > lo) k_mu_gather<<<nblk(hi - lo, 256), 256, 0, h->stream>>>(lo, hi, h->sorted ? h->d_inv.p : nullptr, h->d_scal.p, mu_view(h), dev_dst);
    return POLAR_OK;
  });
}
int polar_mu_scatter(polar_handle *h, long long lo, long long hi, const double *dev_src) {
  return guarded(h, [&]() {
    need_device(h);
    if (hi > lo) k_mu_scatter<<<nblk(hi - lo, 256), 256, 0, h->stream>>>(lo, hi, h->sorted ? h->d_inv.p : nullptr, h->d_scal.p, mu_view(h), dev_src);
    return POLAR_OK;
  });
}
int polar_mu_gather_idx(polar_handle *h, const int *dev_idx, long long n, double *dev_dst) {
  return guarded(h, [&]() {
    need_device(h);
    if (n > 0) k_mu_gather_idx<<<nblk(n, 256), 256, 0, h->stream>>>(n, dev_idx, h->sorted ? h->d_inv.p : nullptr, h->d_scal.p, mu_view(h), dev_dst);
    return POLAR_OK;
  });
}
int polar_mu_scatter_idx(polar_handle *h, const int *dev_idx, long long n, const double *dev_src) {
  return guarded(h, [&]() {
    need_device(h);
    if (n > 0) k_mu_scatter_idx<<<nblk(n, 256), 256, 0, h->stream>>>(n, dev_idx, h->sorted ? h->d_inv.p : nullptr, h->d_scal.p, mu_view(h), dev_src, own_lo(h), own_lo(h) + own_n(h));
    return POLAR_OK;
  });
}
int polar_change_export(polar_handle *h, double *dev_dst) {
  return guarded(h, [&]() {
    need_device(h);
    k_fold_change<<<1, POLAR_NSLOT, 0, h->stream>>>(h->d_scal.p, h->d_slots.p, dev_dst, det_part(h), det_npart(h));
    return POLAR_OK;
  });
}

int polar_step_mu_get(polar_handle *h, long long lo, long long hi, double *mu_host) {
  return guarded(h, [&]() {
    need_device(h);
    if (!mu_host || lo < 0 || hi < lo || hi > h->nlocal) throw InputError("polar_step_mu_get: bad range or null pointer");
    const size_t cnt = 3 * (size_t)(hi - lo);
    if (cnt == 0) return (int)POLAR_OK;
    h->d_xchg.ensure(cnt);
    k_mu_gather<<<nblk(hi - lo, 256), 256, 0, h->stream>>>(lo, hi, h->sorted ? h->d_inv.p : nullptr, h->d_scal.p, mu_view(h), h->d_xchg.p);
    HIPCHECK(hipMemcpyAsync(mu_host, h->d_xchg.p, cnt * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHECK(hipStreamSynchronize(h->stream));
    return (int)POLAR_OK;
  });
}
int polar_step_mu_put_idx(polar_handle *h, long long n, const int *idx_host, const double *mu_host) {
  return guarded(h, [&]() {
    need_device(h);
    if (n < 0 || (n > 0 && (!idx_host || !mu_host))) throw InputError("polar_step_mu_put_idx: null pointer");
    if (n == 0) return (int)POLAR_OK;
    for (long long k = 0; k < n; k++)
      if (idx_host[k] >= h->nlocal) throw InputError("polar_step_mu_put_idx: atom index out of range");
    h->d_xchg.ensure(3 * (size_t)n); h->d_xidx.ensure((size_t)n);
    HIPCHECK(hipMemcpyAsync(h->d_xidx.p, idx_host, (size_t)n * sizeof(int), hipMemcpyHostToDevice, h->stream));
    HIPCHECK(hipMemcpyAsync(h->d_xchg.p, mu_host, 3 * (size_t)n * sizeof(double), hipMemcpyHostToDevice, h->stream));
    k_mu_scatter_idx<<<nblk(n, 256), 256, 0, h->stream>>>(n, h->d_xidx.p, h->sorted ? h->d_inv.p : nullptr, h->d_scal.p, mu_view(h), h->d_xchg.p, own_lo(h), own_lo(h) + own_n(h));
    HIPCHECK(hipStreamSynchronize(h->stream));  // the host buffers may be reused by the caller
    return (int)POLAR_OK;
  });
}
int polar_step_change_get(polar_handle *h, double *sum) {
  return guarded(h, [&]() {
    need_device(h);
    if (!sum) throw InputError("polar_step_change_get: null pointer");
    h->d_xchg.ensure(8);
    k_fold_change<<<1, POLAR_NSLOT, 0, h->stream>>>(h->d_scal.p, h->d_slots.p, h->d_xchg.p, det_part(h), det_npart(h));
    HIPCHECK(hipMemcpyAsync(sum, h->d_xchg.p, sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHECK(hipStreamSynchronize(h->stream));
    return (int)POLAR_OK;
  });
}
int polar_step_sweep_end_host(polar_handle *h, double global_change) {
  if (!h) return POLAR_ERR_STATE;
  int rc = guarded(h, [&]() {
    need_device(h);
    h->d_xchg.ensure(8);
    HIPCHECK(hipMemcpyAsync(h->d_xchg.p + 4, &global_change, sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIPCHECK(hipStreamSynchronize(h->stream));  // global_change lives on the caller's stack
    return (int)POLAR_OK;
  });
  if (rc < 0) return rc;
  return polar_step_sweep_end(h, h->d_xchg.p + 4);
}


}  // extern "C"
