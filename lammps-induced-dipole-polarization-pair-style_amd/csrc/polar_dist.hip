// polar_dist.hip -- multi-GPU driver inside the library: one process (or MPI rank) per GPU, RCCL over xGMI.
#include <dlfcn.h>
#include <rccl/rccl.h>   // types only: the entry points are resolved at run time, the library does not link librccl
#include "polar_handle.hpp"

/* ---- multi-GPU driver inside the library: one rank per GPU, RCCL over xGMI --------------------------------------------
 * What the reference's dead pack_comm / unpack_comm (PS.h:51-52, PS.cpp:1320-1362) never delivered, without a host
 * language in the per-sweep loop: per sweep the library enqueues, on its compute stream,
 *     pack kernel -> ncclGroupStart / ncclRecv + ncclSend per peer / ncclGroupEnd -> unpack kernel
 * and every `reduce_every` sweeps one ncclAllReduce of the stop rule's double; the host looks at the device-resident loop
 * state every `check_every` sweeps only. */
namespace {
struct RcclApi {
  void *lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
};
RcclApi &rccl() {
  static RcclApi api;
  if (api.lib) return api;
  // the copy already in the process first (PyTorch ships its own librccl.so and two copies would not see each other's state)
  // POLAR_RCCL_LIB=<path>: open this library instead (a site's own RCCL build; tests/dist_mock: an in-process stand-in that
  // lets several ranks of ONE process drive this code on one GPU)
  if (const char *e = getenv("POLAR_RCCL_LIB")) {
    api.lib = dlopen(e, RTLD_NOW | RTLD_GLOBAL);
    if (!api.lib) throw std::runtime_error(std::string("polar_dist: cannot open POLAR_RCCL_LIB: ") + e);
  }
  const char *names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
  for (int pass = 0; pass < 2 && !api.lib; pass++)
    for (const char *nm : names) {
      api.lib = dlopen(nm, RTLD_NOW | RTLD_GLOBAL | (pass == 0 ? RTLD_NOLOAD : 0));
      if (api.lib) break;
    }
  if (!api.lib) throw std::runtime_error("polar_dist: librccl.so not found (the multi-GPU driver needs RCCL)");
#define POLAR_RCCL_SYM(field, name) \
  *(void **)(&api.field) = dlsym(api.lib, name); \
  if (!api.field) throw std::runtime_error(std::string("polar_dist: RCCL symbol missing: ") + name)
  POLAR_RCCL_SYM(GetUniqueId, "ncclGetUniqueId"); POLAR_RCCL_SYM(CommInitRank, "ncclCommInitRank");
  POLAR_RCCL_SYM(CommDestroy, "ncclCommDestroy"); POLAR_RCCL_SYM(GroupStart, "ncclGroupStart");
  POLAR_RCCL_SYM(GroupEnd, "ncclGroupEnd"); POLAR_RCCL_SYM(Send, "ncclSend"); POLAR_RCCL_SYM(Recv, "ncclRecv");
  POLAR_RCCL_SYM(AllReduce, "ncclAllReduce"); POLAR_RCCL_SYM(GetErrorString, "ncclGetErrorString");
#undef POLAR_RCCL_SYM
  return api;
}
#define RCCLCHECK(expr)                                                                                     \
  do {                                                                                                      \
    ncclResult_t r_ = (expr);                                                                               \
    if (r_ != ncclSuccess) throw HipError(std::string(#expr) + " failed: " + rccl().GetErrorString(r_));    \
  } while (0)
}  // namespace

struct polar_dist {
  ncclComm_t comm = nullptr;
  int rank = 0, nranks = 1;
  std::string err;
  // halo plan of the handle this driver steps: peers, and per peer the rows it sends / the rows it receives (atom indices)
  std::vector<int> peers, send_off, recv_off;   // offsets into the packed buffers, in atoms; size npeers + 1
  DBuf<int> d_send_idx, d_recv_idx;
  DBuf<double> d_send, d_recv, d_red;           // packed dipoles; [0] this rank's sum (dmu)^2 -> all-reduced, [1] +inf, [2..17] end-of-step sums
  double *h_red = nullptr;                      // pinned
  int reduce_every = 1, check_every = 4;
  int exchanges = 0, allreduces = 0;            // of the last step (diagnostics)
};

namespace {
template <typename F>
int dist_guarded(polar_dist *d, F &&fn) {
  if (!d) return POLAR_ERR_STATE;
  try {
    return fn();
  } catch (const InputError &e) { d->err = e.what(); return POLAR_ERR_INPUT;
  } catch (const NoDevice &e) { d->err = e.what(); return POLAR_ERR_NO_DEVICE;
  } catch (const HipError &e) { d->err = e.what(); return POLAR_ERR_HIP;
  } catch (const std::exception &e) { d->err = e.what(); return POLAR_ERR_STATE; }
}
// one dipole exchange with the peers, enqueued on the handle's stream
void dist_exchange(polar_dist *d, polar_handle *h) {
  const int np = (int)d->peers.size();
  if (np == 0) return;
  const long long ns = d->send_off[np], nr = d->recv_off[np];
  hipStream_t s = h->stream;
  const MuView mv = mu_view(h);
  if (ns > 0) k_mu_gather_idx<<<nblk(ns, 256), 256, 0, s>>>(ns, d->d_send_idx.p, h->sorted ? h->d_inv.p : nullptr, h->d_scal.p, mv, d->d_send.p);
  RcclApi &R = rccl();
  RCCLCHECK(R.GroupStart());
  for (int k = 0; k < np; k++) {
    const long long a = d->recv_off[k], b = d->recv_off[k + 1];
    if (b > a) RCCLCHECK(R.Recv(d->d_recv.p + 3 * a, 3 * (size_t)(b - a), ncclDouble, d->peers[k], d->comm, s));
  }
  for (int k = 0; k < np; k++) {
    const long long a = d->send_off[k], b = d->send_off[k + 1];
    if (b > a) RCCLCHECK(R.Send(d->d_send.p + 3 * a, 3 * (size_t)(b - a), ncclDouble, d->peers[k], d->comm, s));
  }
  RCCLCHECK(R.GroupEnd());
  // (own_lo = own_hi = 0: the plan lists exactly the rows to overwrite; a self-exchange rewrites own rows with their own values)
  if (nr > 0) k_mu_scatter_idx<<<nblk(nr, 256), 256, 0, s>>>(nr, d->d_recv_idx.p, h->sorted ? h->d_inv.p : nullptr, h->d_scal.p, mv, d->d_recv.p, 0, 0);
  d->exchanges++;
}
}  // namespace

extern "C" {

int polar_dist_unique_id(void *id128) {
  if (!id128) return POLAR_ERR_STATE;
  try {
    ncclUniqueId id;
    if (rccl().GetUniqueId(&id) != ncclSuccess) return POLAR_ERR_HIP;
    static_assert(sizeof(id) == POLAR_DIST_ID_BYTES, "ncclUniqueId size");
    memcpy(id128, &id, sizeof(id));
    return POLAR_OK;
  } catch (const std::exception &) { return POLAR_ERR_STATE; }
}
int polar_dist_create(const void *id128, int rank, int nranks, int device, polar_dist **out) {
  if (!out || !id128) return POLAR_ERR_STATE;
  polar_dist *d = new polar_dist();
  *out = d;
  d->rank = rank; d->nranks = nranks;
  return dist_guarded(d, [&]() {
    if (rank < 0 || nranks < 1 || rank >= nranks) throw InputError("polar_dist_create: bad rank");
    HIPCHECK(hipSetDevice(device));
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    RCCLCHECK(rccl().CommInitRank(&d->comm, nranks, id, rank));
    d->d_red.ensure(32);
    HIPCHECK(hipHostMalloc((void **)&d->h_red, 32 * sizeof(double)));
    const double inf = INFINITY;
    HIPCHECK(hipMemcpy(d->d_red.p + 1, &inf, sizeof(double), hipMemcpyHostToDevice));
    return (int)POLAR_OK;
  });
}
int polar_dist_destroy(polar_dist *d) {
  if (!d) return POLAR_OK;
  if (d->comm) (void)rccl().CommDestroy(d->comm);
  d->d_send_idx.release(); d->d_recv_idx.release(); d->d_send.release(); d->d_recv.release(); d->d_red.release();
  if (d->h_red) (void)hipHostFree(d->h_red);
  delete d;
  return POLAR_OK;
}
const char *polar_dist_last_error(const polar_dist *d) { return d ? d->err.c_str() : "null driver"; }
int polar_dist_set_cadence(polar_dist *d, int reduce_every, int check_every) {
  if (!d || reduce_every < 1 || check_every < 1) return POLAR_ERR_STATE;
  d->reduce_every = reduce_every; d->check_every = check_every;
  return POLAR_OK;
}
int polar_dist_set_halo(polar_dist *d, int npeers, const int *peers, const int *send_count, const int *send_idx,
                        const int *recv_count, const int *recv_idx) {
  return dist_guarded(d, [&]() {
    if (npeers < 0 || (npeers > 0 && (!peers || !send_count || !recv_count))) throw InputError("polar_dist_set_halo: null pointer");
    d->peers.assign(peers, peers + npeers);
    d->send_off.assign((size_t)npeers + 1, 0); d->recv_off.assign((size_t)npeers + 1, 0);
    for (int k = 0; k < npeers; k++) {
      if (peers[k] < 0 || peers[k] >= d->nranks || send_count[k] < 0 || recv_count[k] < 0) throw InputError("polar_dist_set_halo: bad peer or count");
      d->send_off[k + 1] = d->send_off[k] + send_count[k];
      d->recv_off[k + 1] = d->recv_off[k] + recv_count[k];
    }
    const size_t ns = (size_t)d->send_off[npeers], nr = (size_t)d->recv_off[npeers];
    if ((ns && !send_idx) || (nr && !recv_idx)) throw InputError("polar_dist_set_halo: null index list");
    d->d_send_idx.ensure(ns + 1); d->d_recv_idx.ensure(nr + 1); d->d_send.ensure(3 * ns + 3); d->d_recv.ensure(3 * nr + 3);
    if (ns) HIPCHECK(hipMemcpy(d->d_send_idx.p, send_idx, ns * sizeof(int), hipMemcpyHostToDevice));
    if (nr) HIPCHECK(hipMemcpy(d->d_recv_idx.p, recv_idx, nr * sizeof(int), hipMemcpyHostToDevice));
    return (int)POLAR_OK;
  });
}
int polar_dist_exchange(polar_dist *d, polar_handle *h) {
  return dist_guarded(d, [&]() {
    if (!h) throw InputError("polar_dist_exchange: null handle");
    need_device(h);
    dist_exchange(d, h);
    return (int)POLAR_OK;
  });
}
int polar_dist_step(polar_dist *d, polar_handle *h, int eflag, int vflag, polar_result *out) {
  return dist_guarded(d, [&]() {
    if (!h || !out) throw InputError("polar_dist_step: null pointer");
    need_device(h);
    HIPCHECK(hipSetDevice(h->device));
    RcclApi &R = rccl();
    const polar_settings &st = h->ph.st;
    if (!(st.dd_cutoff > 0.0)) throw InputError("polar_dist_step needs list mode (dd_cutoff > 0): exact mode runs as replicas only");
    const bool gs = st.polar_gs || st.polar_gs_ranked;
    const int max_sweeps = st.iterations_max + 1;
    int rc = POLAR_OK;
    for (int attempt = 0;; attempt++) {
      d->exchanges = d->allreduces = 0;
      rc = polar_step_begin(h, eflag, vflag);
      if (rc < 0) { d->err = h->err; return rc; }
      hipStream_t s = h->stream;
      dist_exchange(d, h);  // the other ranks' initial guess
      if (!st.zodid) {
        const bool lazy = st.fixed_iteration && gs;
        for (int sw = 0; sw < max_sweeps; sw++) {
          sweep_once(h, false);
          if (!st.fixed_iteration) {
            // the stop rule (PS.cpp:1194-1210) needs the sum over all ranks: one all-reduced double every `reduce_every` sweeps;
            // in between the end-of-sweep logic is told "not converged yet" (+inf)
            const double *gc = d->d_red.p + 1;
            if ((sw % d->reduce_every) == d->reduce_every - 1 || sw >= st.iterations_max) {
              k_fold_change<<<1, POLAR_NSLOT, 0, s>>>(h->d_scal.p, h->d_slots.p, d->d_red.p, det_part(h), det_npart(h));
              RCCLCHECK(R.AllReduce(d->d_red.p, d->d_red.p, 1, ncclDouble, ncclSum, d->comm, s));
              d->allreduces++;
              gc = d->d_red.p;
            }
            k_solver_step<<<1, POLAR_NSLOT, 0, s>>>(h->d_scal.p, h->d_slots.p, norm_count(h), st.fixed_iteration, st.iterations_max, st.polar_precision,
                                                  gs ? 0 : 1, gc, 1, gc == d->d_red.p ? nullptr : det_part(h), gc == d->d_red.p ? 0 : det_npart(h));
          } else if (lazy) {
            if (sw == max_sweeps - 2 || sw == max_sweeps - 1 || max_sweeps == 1)
              k_solver_step<<<1, POLAR_NSLOT, 0, s>>>(h->d_scal.p, h->d_slots.p, norm_count(h), st.fixed_iteration, st.iterations_max, st.polar_precision,
                                                    0, nullptr, (sw == max_sweeps - 2) ? max_sweeps - 1 : 1, det_part(h), det_npart(h));
          } else {
            k_solver_step<<<1, POLAR_NSLOT, 0, s>>>(h->d_scal.p, h->d_slots.p, norm_count(h), st.fixed_iteration, st.iterations_max, st.polar_precision,
                                                  gs ? 0 : 1, nullptr, 1, det_part(h), det_npart(h));
          }
          dist_exchange(d, h);
          if (!st.fixed_iteration && (sw % d->check_every) == d->check_every - 1) {
            read_scal(h);  // identical on every rank: same all-reduced sum (sweeps past the end are no-ops on the device)
            if (h->h_scal->done) break;
          }
        }
      }
      rc = polar_step_finish(h, out);
      // a rank whose rows outgrew their pitch reports POLAR_RETRY_STEP: the flag is max-reduced so that all ranks repeat
      // together; the same call sums energies, virial and pair counts over the ranks
      double *hr = d->h_red;
      hr[0] = rc == POLAR_RETRY_STEP ? 1.0 : 0.0;
      hr[1] = out->eng_vdwl; hr[2] = out->eng_coul; hr[3] = out->eng_pol; hr[4] = out->u_self; hr[5] = out->u_ef; hr[6] = out->u_dd;
      for (int k = 0; k < 6; k++) hr[7 + k] = out->virial[k];
      hr[13] = (double)out->dd_pairs;
      hr[14] = rc < 0 ? 1.0 : 0.0;
      HIPCHECK(hipMemcpyAsync(d->d_red.p + 2, hr, 15 * sizeof(double), hipMemcpyHostToDevice, s));
      RCCLCHECK(R.AllReduce(d->d_red.p + 3, d->d_red.p + 3, 14, ncclDouble, ncclSum, d->comm, s));
      RCCLCHECK(R.AllReduce(d->d_red.p + 2, d->d_red.p + 2, 1, ncclDouble, ncclMax, d->comm, s));
      HIPCHECK(hipMemcpyAsync(hr, d->d_red.p + 2, 15 * sizeof(double), hipMemcpyDeviceToHost, s));
      HIPCHECK(hipStreamSynchronize(s));
      if (hr[14] > 0.0) { if (rc >= 0) { d->err = "polar_dist_step: another rank failed"; rc = POLAR_ERR_STATE; } else d->err = h->err; return rc; }
      if (hr[0] == 0.0) {
        out->eng_vdwl = hr[1]; out->eng_coul = hr[2]; out->eng_pol = hr[3]; out->u_self = hr[4]; out->u_ef = hr[5]; out->u_dd = hr[6];
        for (int k = 0; k < 6; k++) out->virial[k] = hr[7 + k];
        out->dd_pairs = (long long)hr[13];
        break;
      }
      if (attempt >= 4) throw std::runtime_error("polar_dist_step: neighbor list pitch overflow persists");
    }
    return rc;
  });
}
int polar_dist_counters(const polar_dist *d, int *exchanges, int *allreduces) {
  if (!d) return POLAR_ERR_STATE;
  if (exchanges) *exchanges = d->exchanges;
  if (allreduces) *allreduces = d->allreduces;
  return POLAR_OK;
}

}  // extern "C"
