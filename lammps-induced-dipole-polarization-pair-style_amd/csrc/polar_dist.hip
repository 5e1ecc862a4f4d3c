// polar_dist.hip -- multi-GPU driver inside the library: one process (or MPI rank) per GPU, RCCL over xGMI.
#include <dlfcn.h>
#include <rccl/rccl.h>   // types only: the entry points are resolved at run time, the library does not link librccl
#include "polar_handle.hpp"

#include <mutex>

/* ---- multi-GPU driver inside the library: one rank per GPU, RCCL over xGMI --------------------------------------------
 * What the reference's dead pack_comm / unpack_comm (PS.h:51-52, PS.cpp:1320-1362) never delivered, without a host
 * language in the per-sweep loop.
 *
 * Schedule of a Gauss-Seidel solve (list mode) when the ranks share ONE colouring (build_colors_distributed, or colours
 * handed in with polar_set_colors): the colour phases of all ranks together are the single-GPU iteration, and after phase c
 * only the dipoles of colour c's boundary rows have to travel.  Per phase, on the compute stream: boundary rows -> event ->
 * interior rows; on the communication stream, behind the event: pack kernel -> ncclGroupStart / ncclRecv + ncclSend per peer /
 * ncclGroupEnd -> unpack kernel -> event.  A phase waits for the exchange issued `lag` + 1 phases before it: lag 0 = every
 * phase sees all earlier phases of all ranks (the single-GPU iterates exactly; the exchange is hidden behind the interior
 * rows only), lag 1 (default) = a halo dipole may be one phase old and every exchange has a whole phase of compute to hide
 * behind.  The stop rule's all-reduce (one double every `reduce_every` sweeps) stays on the compute stream.
 * Without a shared colouring (no classes set) or for Jacobi: one exchange per sweep, the round-3 schedule.
 * The host looks at the device-resident loop state every `check_every` sweeps only. */
namespace {
struct RcclApi {
  void *lib = nullptr;
  bool ready = false;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
  // optional (absent from the test stand-in): rank count, abort of a communicator after a rank-local failure
  ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
  ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
  ncclResult_t (*CommGetAsyncError)(ncclComm_t, ncclResult_t *) = nullptr;
  ncclResult_t (*CommSplit)(ncclComm_t, int, int, ncclComm_t *, void *) = nullptr;
};
std::once_flag g_rccl_once;
RcclApi g_rccl;
std::string g_rccl_error;
void rccl_open() {
  RcclApi api;
  // the copy already in the process first (PyTorch ships its own librccl.so and two copies would not see each other's state)
  // POLAR_RCCL_LIB=<path>: open this library instead (a site's own RCCL build; tests/dist_mock: an in-process stand-in that
  // lets several ranks of ONE process drive this code on one GPU)
  if (const char *e = getenv("POLAR_RCCL_LIB")) {
    api.lib = dlopen(e, RTLD_NOW | RTLD_GLOBAL);
    if (!api.lib) { g_rccl_error = std::string("polar_dist: cannot open POLAR_RCCL_LIB: ") + e; return; }
  }
  const char *names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
  for (int pass = 0; pass < 2 && !api.lib; pass++)
    for (const char *nm : names) {
      api.lib = dlopen(nm, RTLD_NOW | RTLD_GLOBAL | (pass == 0 ? RTLD_NOLOAD : 0));
      if (api.lib) break;
    }
  if (!api.lib) { g_rccl_error = "polar_dist: librccl.so not found (the multi-GPU driver needs RCCL)"; return; }
#define POLAR_RCCL_SYM(field, name) \
  *(void **)(&api.field) = dlsym(api.lib, name); \
  if (!api.field) { g_rccl_error = std::string("polar_dist: RCCL symbol missing: ") + name; return; }
  POLAR_RCCL_SYM(GetUniqueId, "ncclGetUniqueId"); POLAR_RCCL_SYM(CommInitRank, "ncclCommInitRank");
  POLAR_RCCL_SYM(CommDestroy, "ncclCommDestroy"); POLAR_RCCL_SYM(GroupStart, "ncclGroupStart");
  POLAR_RCCL_SYM(GroupEnd, "ncclGroupEnd"); POLAR_RCCL_SYM(Send, "ncclSend"); POLAR_RCCL_SYM(Recv, "ncclRecv");
  POLAR_RCCL_SYM(AllReduce, "ncclAllReduce"); POLAR_RCCL_SYM(GetErrorString, "ncclGetErrorString");
#undef POLAR_RCCL_SYM
  *(void **)(&api.CommCount) = dlsym(api.lib, "ncclCommCount");
  *(void **)(&api.CommAbort) = dlsym(api.lib, "ncclCommAbort");
  *(void **)(&api.CommGetAsyncError) = dlsym(api.lib, "ncclCommGetAsyncError");
  *(void **)(&api.CommSplit) = dlsym(api.lib, "ncclCommSplit");
  api.ready = true;
  g_rccl = api;   // published only when complete: a failed attempt leaves no half-filled table behind
}
RcclApi &rccl() {
  std::call_once(g_rccl_once, rccl_open);
  if (!g_rccl.ready) throw std::runtime_error(g_rccl_error.empty() ? "polar_dist: RCCL unavailable" : g_rccl_error);
  return g_rccl;
}
#define RCCLCHECK(expr)                                                                                     \
  do {                                                                                                      \
    ncclResult_t r_ = (expr);                                                                               \
    if (r_ != ncclSuccess) throw HipError(std::string(#expr) + " failed: " + rccl().GetErrorString(r_));    \
  } while (0)
const int kRing = 8;   // events kept per stream: a phase waits for an exchange at most lag + 1 <= 3 phases old
}  // namespace

struct polar_dist {
  ncclComm_t comm = nullptr;      // point-to-point exchanges (communication stream)
  ncclComm_t comm_red = nullptr;  // all-reduces (compute stream): a communicator of its own where ncclCommSplit exists -- RCCL orders the
                                  // operations of ONE communicator across streams, and the stop rule's all-reduce would wait for the
                                  // exchange in flight on the other stream; the same communicator otherwise
  int rank = 0, nranks = 1, device = 0;
  std::string err;
  // halo plan of the handle this driver steps: peers, and per peer the rows it sends / the rows it receives (atom indices)
  std::vector<int> peers, send_off, recv_off;   // offsets into the packed buffers, in atoms; size npeers + 1
  std::vector<int> h_send_idx, h_recv_idx;      // host copies (re-sorted by colour after every colouring)
  DBuf<int> d_send_idx, d_recv_idx;
  DBuf<double> d_send, d_recv, d_red;           // packed dipoles; d_red: [0] this rank's sum (dmu)^2 -> all-reduced, [1] +inf, [2..] agreement flags and end-of-step sums
  double *h_red = nullptr;                      // pinned
  int reduce_every = 1, check_every = 4;
  int exchanges = 0, allreduces = 0, phase_exchanges = 0;   // of the last step (diagnostics)
  // schedule
  int lag = 1;                                  // phases a halo dipole may be late (0, 1, 2); -1: one exchange per sweep
  int my_class = -1, nclasses = 0;              // turn of this rank in a distributed colouring (no two peers share a class); -1: each rank colours for itself
  hipStream_t xs = nullptr;                     // communication stream
  hipEvent_t ev_phase[kRing] = {}, ev_xdone[kRing] = {};
  // the plan by colour (valid for colouring `plan_epoch` of the handle): per colour c and peer k the atoms
  // [coff[c * np + k], coff[c * np + k + 1]) of the colour-sorted lists
  long long plan_epoch = -1;
  int plan_nc = 0;
  std::vector<int> csend_off, crecv_off;
  DBuf<int> d_csend_idx, d_crecv_idx;
  DBuf<double> d_csend, d_crecv;
  // periodic images held by the handle (atoms gfirst .. gfirst + ng): owner atom and shift, for polar_dist_positions
  DBuf<int> d_gowner; DBuf<double> d_gshift;
  long long ng = 0, gfirst = 0;
  polar_result local{};                         // this rank's own share of the last step (energies, virial, pairs)
  double host_us_rccl = 0.0, host_us_loop = 0.0; // host time of the last step: inside RCCL calls / issuing the whole sweep loop (POLAR_DEBUG prints them)
  // polar_dist_profile: timed events between the parts of the sweep loop; the interval that ENDS at mark k belongs to part prof_kind[k]
  double timeout_s = 180.0;        // POLAR_DIST_TIMEOUT_S: how long a wait for the device may see no progress before the step gives up (a peer that is not answering)
  bool prof_on = false;
  std::vector<hipEvent_t> prof_ev; std::vector<int> prof_kind; size_t prof_n = 0;
  double prof_ms[POLAR_DIST_PROF_PARTS] = {0, 0, 0, 0, 0}; int prof_intervals = 0;
};

namespace {
enum { PK_OTHER = 0, PK_SWEEP, PK_REDUCE, PK_XCHG, PK_ACCEL };
inline void prof_mark(polar_dist *d, hipStream_t s, int kind) {
  if (!d->prof_on) return;
  if (d->prof_n == d->prof_ev.size()) { hipEvent_t e; HIPCHECK(hipEventCreate(&e)); d->prof_ev.push_back(e); d->prof_kind.push_back(0); }
  d->prof_kind[d->prof_n] = kind;
  HIPCHECK(hipEventRecord(d->prof_ev[d->prof_n++], s));
}
// after the step's last synchronisation: the device time of the sweep loop by part
inline void prof_fold(polar_dist *d) {
  if (!d->prof_on) return;
  for (double &v : d->prof_ms) v = 0.0;
  d->prof_intervals = 0;
  for (size_t k = 1; k < d->prof_n; k++) {
    float ms = 0.f;
    HIPCHECK(hipEventElapsedTime(&ms, d->prof_ev[k - 1], d->prof_ev[k]));
    d->prof_ms[d->prof_kind[k]] += ms;
    d->prof_intervals++;
  }
}
// Wait for stream `s` to drain -- every synchronisation of a step goes through here -- but not for ever: an exchange or an
// all-reduce whose peer never answers (a rank that died, a schedule the ranks do not agree on) would otherwise leave this rank
// inside hipStreamSynchronize with nothing to report.  After `timeout_s` without the stream draining, or as soon as RCCL
// reports an asynchronous error, the communicators are aborted and the step ends with an error that says so.
void dist_wait(polar_dist *d, hipStream_t s) {
  const auto t0 = std::chrono::steady_clock::now();
  for (long long spin = 0;; spin++) {
    const hipError_t q = hipStreamQuery(s);
    if (q == hipSuccess) return;
    if (q != hipErrorNotReady) throw HipError(std::string("polar_dist: hipStreamQuery failed: ") + hipGetErrorString(q));
    if (spin < 2000) continue;                                   // (a look at the loop state normally ends within microseconds)
    std::this_thread::sleep_for(std::chrono::microseconds(50));
    if ((spin & 1023) != 0) continue;
    ncclResult_t async = ncclSuccess;
    const bool rccl_bad = d->comm && g_rccl.CommGetAsyncError && g_rccl.CommGetAsyncError(d->comm, &async) == ncclSuccess && async != ncclSuccess && async != ncclInProgress;
    const double waited = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (rccl_bad || waited > d->timeout_s) {
      if (g_rccl.CommAbort && d->comm) {
        if (d->comm_red && d->comm_red != d->comm) (void)g_rccl.CommAbort(d->comm_red);
        (void)g_rccl.CommAbort(d->comm);
        d->comm = d->comm_red = nullptr;
      }
      throw HipError(rccl_bad ? std::string("polar_dist: RCCL reported an asynchronous error while rank ") + std::to_string(d->rank) + " waited for the device: " + g_rccl.GetErrorString(async)
                              : "polar_dist: rank " + std::to_string(d->rank) + " saw no progress on the device for " + std::to_string((int)waited) +
                                    " s (POLAR_DIST_TIMEOUT_S): a peer rank is not answering an exchange or an all-reduce; the communicator was aborted");
    }
  }
}
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
// grouped send / receive with the peers: segment k of the packed buffers = peer k; `unit` doubles per atom
void p2p(polar_dist *d, const int *soff, const int *roff, const double *sendbuf, double *recvbuf, int unit, hipStream_t s) {
  const int np = (int)d->peers.size();
  RcclApi &R = rccl();
  struct Clock { polar_dist *d; std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
                 ~Clock() { d->host_us_rccl += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count(); } } clock{d};
  RCCLCHECK(R.GroupStart());
  for (int k = 0; k < np; k++) {
    const long long a = roff[k], b = roff[k + 1];
    if (b > a) RCCLCHECK(R.Recv(recvbuf + unit * a, (size_t)unit * (size_t)(b - a), ncclDouble, d->peers[k], d->comm, s));
  }
  for (int k = 0; k < np; k++) {
    const long long a = soff[k], b = soff[k + 1];
    if (b > a) RCCLCHECK(R.Send(sendbuf + unit * a, (size_t)unit * (size_t)(b - a), ncclDouble, d->peers[k], d->comm, s));
  }
  RCCLCHECK(R.GroupEnd());
}
// one dipole exchange of ALL halo atoms with the peers, enqueued on the handle's stream
void dist_exchange(polar_dist *d, polar_handle *h, bool packed = false) {   // packed: the send buffer has been filled already (k_solver_step_gather)
  const int np = (int)d->peers.size();
  if (np == 0) return;
  const long long ns = d->send_off[np], nr = d->recv_off[np];
  hipStream_t s = h->stream;
  const MuView mv = mu_view(h);
  if (ns > 0 && !packed) k_mu_gather_idx<<<nblk(ns, 256), 256, 0, s>>>(ns, d->d_send_idx.p, h->sorted ? h->d_inv.p : nullptr, h->d_scal.p, mv, d->d_send.p);
  p2p(d, d->send_off.data(), d->recv_off.data(), d->d_send.p, d->d_recv.p, 3, s);
  // (own_lo = own_hi = 0: the plan lists exactly the rows to overwrite; a self-exchange rewrites own rows with their own values)
  if (nr > 0) k_mu_scatter_idx<<<nblk(nr, 256), 256, 0, s>>>(nr, d->d_recv_idx.p, h->sorted ? h->d_inv.p : nullptr, h->d_scal.p, mv, d->d_recv.p, 0, 0);
  d->exchanges++;
}
// the dipoles of colour c's boundary rows, on stream `s` (the communication stream)
void dist_exchange_color(polar_dist *d, polar_handle *h, int c, hipStream_t s) {
  const int np = (int)d->peers.size();
  if (np == 0) return;
  const int *so = d->csend_off.data() + (size_t)c * np, *ro = d->crecv_off.data() + (size_t)c * np;
  const long long s0 = so[0], s1 = so[np], r0 = ro[0], r1 = ro[np];
  const MuView mv = mu_view(h);
  if (s1 > s0) k_mu_gather_idx<<<nblk(s1 - s0, 256), 256, 0, s>>>(s1 - s0, d->d_csend_idx.p + s0, h->d_inv.p, h->d_scal.p, mv, d->d_csend.p + 3 * s0);
  if (s1 > s0 || r1 > r0) p2p(d, so, ro, d->d_csend.p, d->d_crecv.p, 3, s);
  if (r1 > r0) k_mu_scatter_idx<<<nblk(r1 - r0, 256), 256, 0, s>>>(r1 - r0, d->d_crecv_idx.p + r0, h->d_inv.p, h->d_scal.p, mv, d->d_crecv.p + 3 * r0, 0, 0);
  d->phase_exchanges++;
}
// host doubles -> all-reduce over the ranks -> host doubles (synchronises): agreement on flags, sums of small tables
void host_allreduce(polar_dist *d, polar_handle *h, double *vals, int count, ncclRedOp_t op) {
  if (count > 24) throw std::logic_error("host_allreduce: at most 24 doubles at a time");
  hipStream_t s = h->stream;
  double *hr = d->h_red + 32;   // (its own pinned slots: d->h_red[0..31] belong to the end-of-step sums)
  memcpy(hr, vals, count * sizeof(double));
  HIPCHECK(hipMemcpyAsync(d->d_red.p + 32, hr, count * sizeof(double), hipMemcpyHostToDevice, s));
  RCCLCHECK(rccl().AllReduce(d->d_red.p + 32, d->d_red.p + 32, count, ncclDouble, op, d->comm, s));
  HIPCHECK(hipMemcpyAsync(hr, d->d_red.p + 32, count * sizeof(double), hipMemcpyDeviceToHost, s));
  dist_wait(d, s);
  memcpy(vals, hr, count * sizeof(double));
}
// After a (re)colouring shared by the ranks: the exchange lists sorted by colour (stable inside a peer's segment: both ends
// of a link walk the same atoms in the same order, since a halo row carries its owner's colour).  The per-colour counts are
// checked against the peers' before anything depends on them: a mismatch would make the next exchange hang.
void build_phase_plan(polar_dist *d, polar_handle *h) {
  const int np = (int)d->peers.size(), n = h->nlocal;
  std::vector<int> color((size_t)n + 1, -1);
  HIPCHECK(hipMemcpy(color.data(), h->d_color_orig.p, (size_t)n * sizeof(int), hipMemcpyDeviceToHost));
  double ncd = (double)((int)h->color_off.size() - 1);
  host_allreduce(d, h, &ncd, 1, ncclMax);   // (a rank may hold no row of the last colours)
  const int nc = (int)ncd;
  auto sort_by_color = [&](const std::vector<int> &idx, const std::vector<int> &off, std::vector<int> &coff, std::vector<int> &out) {
    coff.assign((size_t)nc * np + 1, 0);
    out.clear();
    for (int c = 0; c < nc; c++)
      for (int k = 0; k < np; k++) {
        coff[(size_t)c * np + k] = (int)out.size();
        for (int t = off[k]; t < off[k + 1]; t++) if (color[idx[t]] == c) out.push_back(idx[t]);
      }
    coff[(size_t)nc * np] = (int)out.size();
  };
  std::vector<int> sidx, ridx;
  sort_by_color(d->h_send_idx, d->send_off, d->csend_off, sidx);
  sort_by_color(d->h_recv_idx, d->recv_off, d->crecv_off, ridx);
  d->d_csend_idx.ensure(sidx.size() + 1); d->d_crecv_idx.ensure(ridx.size() + 1);
  d->d_csend.ensure(3 * sidx.size() + 3); d->d_crecv.ensure(3 * ridx.size() + 3);
  if (!sidx.empty()) HIPCHECK(hipMemcpy(d->d_csend_idx.p, sidx.data(), sidx.size() * sizeof(int), hipMemcpyHostToDevice));
  if (!ridx.empty()) HIPCHECK(hipMemcpy(d->d_crecv_idx.p, ridx.data(), ridx.size() * sizeof(int), hipMemcpyHostToDevice));
  // cross-check: peer k is told how many rows of every colour this rank will send it
  double bad = 0.0;
  if (np > 0) {
    if (nc > 64) throw std::runtime_error("polar_dist: more than 64 colours");
    std::vector<double> mine((size_t)np * 64, 0.0), theirs((size_t)np * 64, -1.0);
    std::vector<int> off((size_t)np + 1);
    for (int k = 0; k <= np; k++) off[k] = 64 * k;
    for (int k = 0; k < np; k++)
      for (int c = 0; c < nc; c++) mine[(size_t)k * 64 + c] = d->csend_off[(size_t)c * np + k + 1] - d->csend_off[(size_t)c * np + k];
    d->d_send.ensure((size_t)np * 64 + 64); d->d_recv.ensure((size_t)np * 64 + 64);
    HIPCHECK(hipMemcpy(d->d_send.p, mine.data(), mine.size() * sizeof(double), hipMemcpyHostToDevice));
    p2p(d, off.data(), off.data(), d->d_send.p, d->d_recv.p, 1, h->stream);
    dist_wait(d, h->stream);
    HIPCHECK(hipMemcpy(theirs.data(), d->d_recv.p, theirs.size() * sizeof(double), hipMemcpyDeviceToHost));
    for (int k = 0; k < np; k++)
      for (int c = 0; c < nc; c++)
        if ((int)theirs[(size_t)k * 64 + c] != d->crecv_off[(size_t)c * np + k + 1] - d->crecv_off[(size_t)c * np + k]) bad = 1.0;
    // (the buffers are sized for the dipole exchange again below)
    d->d_send.ensure(3 * (size_t)d->send_off[np] + 3); d->d_recv.ensure(3 * (size_t)d->recv_off[np] + 3);
  }
  host_allreduce(d, h, &bad, 1, ncclMax);
  if (bad > 0.0) throw std::runtime_error("polar_dist: the ranks disagree about the colours of halo rows (a colouring handed in with polar_set_colors must give a halo row its owner's colour)");
  d->plan_nc = nc;
  d->plan_epoch = h->color_epoch;
}
// the colouring shared by the ranks (polar_color.hip) over this transport
void color_together(polar_dist *d, polar_handle *h) {
  const int np = (int)d->peers.size();
  ColorComm cc;
  cc.my_class = d->my_class; cc.nclasses = d->nclasses;
  cc.exchange = [&](int *color_s) {
    if (np == 0) return;
    const long long ns = d->send_off[np], nr = d->recv_off[np];
    hipStream_t s = h->stream;
    if (ns > 0) k_color_gather_idx<<<nblk(ns, 256), 256, 0, s>>>(ns, d->d_send_idx.p, h->d_inv.p, color_s, d->d_send.p);
    p2p(d, d->send_off.data(), d->recv_off.data(), d->d_send.p, d->d_recv.p, 1, s);
    if (nr > 0) k_color_scatter_idx<<<nblk(nr, 256), 256, 0, s>>>(nr, d->d_recv_idx.p, h->d_inv.p, color_s, d->d_recv.p);
  };
  cc.allreduce = [&](double *host, int count) {
    for (int a = 0; a < count; a += 16) host_allreduce(d, h, host + a, std::min(16, count - a), ncclSum);
  };
  h->colors_valid = false;
  build_colors_distributed(h, h->ph.st.polar_gs_ranked != 0, cc);
  map_color_rows(h);
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
  d->rank = rank; d->nranks = nranks; d->device = device;
  return dist_guarded(d, [&]() {
    if (rank < 0 || nranks < 1 || rank >= nranks) throw InputError("polar_dist_create: bad rank");
    HIPCHECK(hipSetDevice(device));
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    RCCLCHECK(rccl().CommInitRank(&d->comm, nranks, id, rank));
    d->comm_red = d->comm;
    // OFF by default: two communicators whose kernels run side by side have never met a second GPU here, one communicator on
    // two streams is the form the one-rank RCCL test exercises (POLAR_DIST_SPLIT_COMM=1 switches the second one on)
    if (rccl().CommSplit && getenv("POLAR_DIST_SPLIT_COMM") && atoi(getenv("POLAR_DIST_SPLIT_COMM")) != 0) {
      ncclComm_t second = nullptr;
      if (rccl().CommSplit(d->comm, 0, rank, &second, nullptr) == ncclSuccess && second) d->comm_red = second;   // (collective; on failure: one communicator)
    }
    d->d_red.ensure(128);
    HIPCHECK(hipHostMalloc((void **)&d->h_red, 128 * sizeof(double)));
    const double inf = INFINITY;
    HIPCHECK(hipMemcpy(d->d_red.p + 1, &inf, sizeof(double), hipMemcpyHostToDevice));
    HIPCHECK(hipStreamCreateWithFlags(&d->xs, hipStreamNonBlocking));
    for (auto &e : d->ev_phase) HIPCHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    for (auto &e : d->ev_xdone) HIPCHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    if (const char *e = getenv("POLAR_DIST_LAG")) d->lag = std::max(-1, std::min(2, atoi(e)));
    if (const char *e = getenv("POLAR_DIST_TIMEOUT_S")) d->timeout_s = std::max(1.0, atof(e));
    return (int)POLAR_OK;
  });
}
int polar_dist_destroy(polar_dist *d) {
  if (!d) return POLAR_OK;
  (void)hipSetDevice(d->device);
  if (d->xs) (void)hipStreamSynchronize(d->xs);
  if (d->comm_red && d->comm_red != d->comm) { try { (void)rccl().CommDestroy(d->comm_red); } catch (const std::exception &) {} }
  if (d->comm) { try { (void)rccl().CommDestroy(d->comm); } catch (const std::exception &) {} }
  for (auto &e : d->ev_phase) if (e) (void)hipEventDestroy(e);
  for (auto &e : d->ev_xdone) if (e) (void)hipEventDestroy(e);
  for (auto &e : d->prof_ev) if (e) (void)hipEventDestroy(e);
  if (d->xs) (void)hipStreamDestroy(d->xs);
  d->d_send_idx.release(); d->d_recv_idx.release(); d->d_send.release(); d->d_recv.release(); d->d_red.release();
  d->d_csend_idx.release(); d->d_crecv_idx.release(); d->d_csend.release(); d->d_crecv.release(); d->d_gowner.release(); d->d_gshift.release();
  if (d->h_red) (void)hipHostFree(d->h_red);
  delete d;
  return POLAR_OK;
}
const char *polar_dist_last_error(const polar_dist *d) { return d ? d->err.c_str() : "null driver"; }
int polar_dist_comm_count(const polar_dist *d) {
  if (!d || !d->comm) return POLAR_ERR_STATE;
  try {
    int n = d->nranks;
    if (rccl().CommCount && rccl().CommCount(d->comm, &n) != ncclSuccess) return POLAR_ERR_HIP;
    return n;
  } catch (const std::exception &) { return POLAR_ERR_STATE; }
}
int polar_dist_set_cadence(polar_dist *d, int reduce_every, int check_every) {
  if (!d || reduce_every < 1 || check_every < 1) return POLAR_ERR_STATE;
  d->reduce_every = reduce_every; d->check_every = check_every;
  return POLAR_OK;
}
int polar_dist_set_schedule(polar_dist *d, int lag, int my_class, int nclasses) {
  return dist_guarded(d, [&]() {
    if (lag < -1 || lag > 2) throw InputError("polar_dist_set_schedule: lag is -1 (one exchange per sweep), 0, 1 or 2 phases");
    if (nclasses < 0 || (nclasses > 0 && (my_class < 0 || my_class >= nclasses)) || nclasses > 64) throw InputError("polar_dist_set_schedule: bad colouring class");
    d->lag = lag;
    d->my_class = nclasses > 0 ? my_class : -1; d->nclasses = nclasses;
    return (int)POLAR_OK;
  });
}
int polar_dist_set_halo(polar_dist *d, polar_handle *h, int npeers, const int *peers, const int *send_count, const int *send_idx,
                        const int *recv_count, const int *recv_idx) {
  return dist_guarded(d, [&]() {
    if (npeers < 0 || (npeers > 0 && (!peers || !send_count || !recv_count))) throw InputError("polar_dist_set_halo: null pointer");
    HIPCHECK(hipSetDevice(d->device));
    d->peers.assign(peers, peers + npeers);
    d->send_off.assign((size_t)npeers + 1, 0); d->recv_off.assign((size_t)npeers + 1, 0);
    for (int k = 0; k < npeers; k++) {
      if (peers[k] < 0 || peers[k] >= d->nranks || send_count[k] < 0 || recv_count[k] < 0) throw InputError("polar_dist_set_halo: bad peer or count");
      d->send_off[k + 1] = d->send_off[k] + send_count[k];
      d->recv_off[k + 1] = d->recv_off[k] + recv_count[k];
    }
    const size_t ns = (size_t)d->send_off[npeers], nr = (size_t)d->recv_off[npeers];
    if ((ns && !send_idx) || (nr && !recv_idx)) throw InputError("polar_dist_set_halo: null index list");
    if (h) {
      const int nall = h->nlocal + h->nghost;
      for (size_t t = 0; t < ns; t++) if (send_idx[t] < 0 || send_idx[t] >= h->nlocal) throw InputError("polar_dist_set_halo: send index outside the handle's atoms");
      for (size_t t = 0; t < nr; t++) if (recv_idx[t] < 0 || recv_idx[t] >= h->nlocal) throw InputError("polar_dist_set_halo: receive index outside the handle's atoms");
      (void)nall;
    }
    d->h_send_idx.assign(send_idx, send_idx + ns); d->h_recv_idx.assign(recv_idx, recv_idx + nr);
    d->d_send_idx.ensure(ns + 1); d->d_recv_idx.ensure(nr + 1); d->d_send.ensure(3 * ns + 3); d->d_recv.ensure(3 * nr + 3);
    if (ns) HIPCHECK(hipMemcpy(d->d_send_idx.p, send_idx, ns * sizeof(int), hipMemcpyHostToDevice));
    if (nr) HIPCHECK(hipMemcpy(d->d_recv_idx.p, recv_idx, nr * sizeof(int), hipMemcpyHostToDevice));
    d->plan_epoch = -1;   // the lists by colour are rebuilt before the next solve
    if (h) {  // boundary rows of the handle: the rows a peer receives (they come first in their colour phase)
      std::vector<int> flag((size_t)h->nlocal + 1, 1);   // sub-class: 0 = boundary (comes first in its phase), 1 = interior
      for (size_t t = 0; t < ns; t++) flag[send_idx[t]] = 0;
      h->d_bflag.ensure((size_t)h->nlocal + 1);
      HIPCHECK(hipMemcpy(h->d_bflag.p, flag.data(), (size_t)h->nlocal * sizeof(int), hipMemcpyHostToDevice));
      h->bflag_n = h->nlocal;
      h->colors_valid = false;   // (rows of a phase are laid out boundary-first from now on)
    }
    return (int)POLAR_OK;
  });
}
int polar_dist_set_ghosts(polar_dist *d, polar_handle *h, int nghost, const int *owner, const double *shift) {
  return dist_guarded(d, [&]() {
    if (!h) throw InputError("polar_dist_set_ghosts: null handle");
    if (nghost != h->nghost) throw InputError("polar_dist_set_ghosts: one entry per ghost atom of the handle");
    if (nghost > 0 && (!owner || !shift)) throw InputError("polar_dist_set_ghosts: null pointer");
    for (int g = 0; g < nghost; g++) if (owner[g] < 0 || owner[g] >= h->nlocal) throw InputError("polar_dist_set_ghosts: owner outside the handle's local atoms");
    HIPCHECK(hipSetDevice(d->device));
    d->d_gowner.ensure((size_t)nghost + 1); d->d_gshift.ensure(3 * (size_t)nghost + 3);
    if (nghost) {
      HIPCHECK(hipMemcpy(d->d_gowner.p, owner, (size_t)nghost * sizeof(int), hipMemcpyHostToDevice));
      HIPCHECK(hipMemcpy(d->d_gshift.p, shift, 3 * (size_t)nghost * sizeof(double), hipMemcpyHostToDevice));
    }
    d->ng = nghost; d->gfirst = h->nlocal;
    return (int)POLAR_OK;
  });
}
// positions of the halo atoms from their owners (north_star: "ghost x/q/mu over RCCL"; the counterpart of the x part of
// Comm::forward_comm on steps between two neighbor-list builds), then the periodic images the handle holds
int polar_dist_positions(polar_dist *d, polar_handle *h) {
  return dist_guarded(d, [&]() {
    if (!h) throw InputError("polar_dist_positions: null handle");
    need_device(h);
    HIPCHECK(hipSetDevice(h->device));
    const int np = (int)d->peers.size();
    hipStream_t s = h->stream;
    if (np > 0) {
      const long long ns = d->send_off[np], nr = d->recv_off[np];
      if (ns > 0) k_vec3_gather_idx<<<nblk(ns, 256), 256, 0, s>>>(ns, d->d_send_idx.p, h->d_x.p, d->d_send.p);
      p2p(d, d->send_off.data(), d->recv_off.data(), d->d_send.p, d->d_recv.p, 3, s);
      if (nr > 0) k_vec3_scatter_idx<<<nblk(nr, 256), 256, 0, s>>>(nr, d->d_recv_idx.p, h->d_x.p, d->d_recv.p);
    }
    if (d->ng > 0) {
      if (d->ng != h->nghost || d->gfirst != h->nlocal) throw InputError("polar_dist_positions: the ghost map is of another atom set (polar_dist_set_ghosts)");
      k_ghost_images<<<nblk(d->ng, 256), 256, 0, s>>>(d->ng, d->gfirst, d->d_gowner.p, d->d_gshift.p, h->d_x.p);
    }
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
    if (!d->comm) throw std::runtime_error("polar_dist_step: the communicator was aborted after a failure inside an earlier step");
    need_device(h);
    HIPCHECK(hipSetDevice(h->device));
    RcclApi &R = rccl();
    const polar_settings &st = h->ph.st;
    const bool gs = st.polar_gs || st.polar_gs_ranked;
    const int max_sweeps = st.iterations_max + 1;
    int rc = POLAR_OK;
    for (int attempt = 0;; attempt++) {
      d->exchanges = d->allreduces = d->phase_exchanges = 0;
      // ---- begin, then AGREE before the first exchange: a rank that failed here (bad input, a refused mode, an allocation)
      //      must not leave the others waiting in ncclRecv.  The same all-reduce says whether any rank needs colours.
      int brc = POLAR_OK;
      std::string berr;
      try {
        if (!(st.dd_cutoff > 0.0)) throw InputError("polar_dist_step needs list mode (dd_cutoff > 0): exact mode runs as replicas only");
        // (everything the sweeps could refuse is refused HERE, where the ranks can still agree on it)
        if (st.polar_accel > 0 && !(gs && h->sweep_kernel == 2 && !st.zodid))
          throw InputError("polar_accel across ranks needs the Gauss-Seidel sweep of list mode (polar_gs / polar_gs_ranked)");
        if (st.polar_accel > 0 && deterministic(h)) throw InputError("polar_accel and `deterministic yes` exclude each other");
        step_begin_lists(h, eflag, vflag);
      } catch (const InputError &e) { brc = POLAR_ERR_INPUT; berr = e.what();
      } catch (const NoDevice &e) { brc = POLAR_ERR_NO_DEVICE; berr = e.what();
      } catch (const HipError &e) { brc = POLAR_ERR_HIP; berr = e.what();
      } catch (const std::exception &e) { brc = POLAR_ERR_STATE; berr = e.what(); }
      const bool shared = d->my_class >= 0 && gs && !st.zodid && h->sweep_kernel == 2;
      // (the same all-reduce makes the ranks agree on WHERE they look at the loop state -- look_at_state works from the sweep count
      //  of the handle's last solve, and ranks that looked at different sweeps would leave the loop at different sweeps: one of
      //  them would wait in an exchange for ever.  The counts are the same on every rank by construction; if they ever are not,
      //  all ranks fall back to the fixed cadence)
      double flags[4] = {brc < 0 ? 1.0 : 0.0, (brc >= 0 && shared && (step_needs_colors(h) || !h->colors_global)) ? 1.0 : 0.0,
                         (double)h->last_sweeps, -(double)h->last_sweeps};
      host_allreduce(d, h, flags, 4, ncclMax);
      h->last_sweeps = (flags[2] == -flags[3]) ? (int)flags[2] : 0;
      if (flags[0] > 0.0) {
        h->in_step = false;
        if (brc < 0) { d->err = berr; h->err = berr; return brc; }
        d->err = "polar_dist_step: another rank failed at the start of the step"; return (int)POLAR_ERR_STATE;
      }
      if (flags[1] > 0.0) color_together(d, h);   // every rank, also those whose own colours still held: the colouring is one object
      step_begin_finish(h);
      hipStream_t s = h->stream;
      const bool phased = shared && d->lag >= 0 && h->colors_global && !st.zodid;
      if (phased && d->plan_epoch != h->color_epoch) build_phase_plan(d, h);
      dist_exchange(d, h);  // the other ranks' initial guess (all halo atoms)
      // ---- the sweeps.  From here on a rank-local failure cannot be agreed on any more (the others are inside their
      //      exchanges): the communicator is aborted so that they see an error instead of waiting.
      d->host_us_rccl = d->host_us_loop = 0.0;
      const auto t_loop = std::chrono::steady_clock::now();
      d->prof_n = 0;
      prof_mark(d, s, PK_OTHER);
      try {
      if (!st.zodid && phased) {
        const int nc = d->plan_nc, lag = deterministic(h) ? 0 : d->lag;
        const bool lazy = st.fixed_iteration != 0;
        // `polar_accel m` across the ranks: every rank mixes with the SAME coefficients -- the dot products ride the stop
        // rule's all-reduce (1 + 16 doubles, every sweep) -- and the mixed boundary dipoles travel in one exchange of all
        // halo rows after the mix (every row changed, not one colour)
        const bool accel = accel_begin(h, false);
        double *ared = d->d_red.p + 64;   // [0] sum (dmu)^2, [1 .. 16] the dot products
        // boundary rows first, the exchange behind the interior rows -- where there are interior rows worth a launch of their
        // own (8 slabs two cutoffs thick have none: every row is some peer's halo)
        long long interior = 0, rows = h->color_off.empty() ? 0 : h->color_off.back();
        if (h->color_nsub == 2)
          for (size_t c = 0; c + 1 < h->color_off.size(); c++) interior += h->color_sub[2 * c + 2] - h->color_sub[2 * c + 1];
        const bool split = !deterministic(h) && interior * 8 >= rows && interior > 0;
        long long g = 0;   // phases issued so far
        for (int sw = 0; sw < max_sweeps; sw++) {
          for (int c = 0; c < nc; c++, g++) {
            if (g - 1 - lag >= 0) HIPCHECK(hipStreamWaitEvent(s, d->ev_xdone[(g - 1 - lag) % kRing], 0));
            prof_mark(d, s, PK_OTHER);   // (stamped once the wait is over: what ends here is the exposed part of an exchange)
            const bool mine = c < (int)h->color_off.size() - 1;   // (a rank may hold no row of the last colours)
            if (mine) sweep_phase(h, c, split ? 1 : 0);
            HIPCHECK(hipEventRecord(d->ev_phase[g % kRing], s));
            if (mine && split) sweep_phase(h, c, 2);
            prof_mark(d, s, PK_SWEEP);
            HIPCHECK(hipStreamWaitEvent(d->xs, d->ev_phase[g % kRing], 0));
            dist_exchange_color(d, h, c, d->xs);
            HIPCHECK(hipEventRecord(d->ev_xdone[g % kRing], d->xs));
          }
          if (accel && (!lazy || sw < max_sweeps - 1)) {
            if (g > 0) HIPCHECK(hipStreamWaitEvent(s, d->ev_xdone[(g - 1) % kRing], 0));   // no late unpack may land on the mixed dipoles
            prof_mark(d, s, PK_OTHER);
            if (lazy) accel_export(h, ared + 1); else accel_export_fused(h, ared);   // (precision mode: the rank's change and its dot products leave one launch)
            prof_mark(d, s, PK_ACCEL);
            RCCLCHECK(R.AllReduce(ared, ared, 1 + 2 * POLAR_ACCEL_MAXM, ncclDouble, ncclSum, d->comm_red, s));
            d->allreduces++;
            prof_mark(d, s, PK_REDUCE);
            if (lazy) accel_step(h, ared + 1); else accel_decide_mix(h, ared, false);   // (the decision on the all-reduced change and the coefficients: one launch)
            prof_mark(d, s, PK_ACCEL);
            dist_exchange(d, h);
            prof_mark(d, s, PK_XCHG);
          } else if (!st.fixed_iteration) {
            const double *gc = d->d_red.p + 1;   // +inf: "not converged yet"
            if ((sw % d->reduce_every) == d->reduce_every - 1 || sw >= st.iterations_max) {
              k_fold_change<<<1, POLAR_NSLOT, 0, s>>>(h->d_scal.p, h->d_slots.p, d->d_red.p, det_part(h), det_npart(h));
              RCCLCHECK(R.AllReduce(d->d_red.p, d->d_red.p, 1, ncclDouble, ncclSum, d->comm_red, s));
              d->allreduces++;
              gc = d->d_red.p;
            }
            k_solver_step<<<1, POLAR_NSLOT, 0, s>>>(h->d_scal.p, h->d_slots.p, norm_count(h), st.fixed_iteration, st.iterations_max, st.polar_precision,
                                                  0, gc, 1, gc == d->d_red.p ? nullptr : det_part(h), gc == d->d_red.p ? 0 : det_npart(h));
            prof_mark(d, s, PK_REDUCE);
          }
          if (lazy) {
            if (sw == max_sweeps - 2 || sw == max_sweeps - 1 || max_sweeps == 1)
              k_solver_step<<<1, POLAR_NSLOT, 0, s>>>(h->d_scal.p, h->d_slots.p, norm_count(h), st.fixed_iteration, st.iterations_max, st.polar_precision,
                                                    0, nullptr, (sw == max_sweeps - 2) ? max_sweeps - 1 : 1, det_part(h), det_npart(h));
          }
          if (!st.fixed_iteration && look_at_state(h, sw, d->check_every, d->reduce_every)) {
            dist_wait(d, s);
            read_scal(h);  // identical on every rank: same all-reduced sum (sweeps past the end are no-ops on the device)
            if (h->h_scal->done) break;
          }
        }
        if (g > 0) HIPCHECK(hipStreamWaitEvent(s, d->ev_xdone[(g - 1) % kRing], 0));   // every halo dipole is its owner's final one
        prof_mark(d, s, PK_OTHER);
      } else if (!st.zodid) {
        const bool lazy = st.fixed_iteration && gs;
        // `polar_accel m` with one exchange per sweep: the same mixing as above on the map "one sweep of every rank against the
        // halo dipoles of the previous one" (block-Jacobi across the ranks) -- everything on the compute stream, one all-reduce
        // (1 + 16 doubles) and one exchange of all halo rows per sweep
        const bool accel = gs && accel_begin(h, false);
        double *ared = d->d_red.p + 64;   // [0] sum (dmu)^2, [1 .. 16] the dot products
        for (int sw = 0; sw < max_sweeps; sw++) {
          bool packed = false;
          sweep_once(h, false);
          prof_mark(d, s, PK_SWEEP);
          if (accel) {
            if (!lazy || sw < max_sweeps - 1) {
              if (lazy) accel_export(h, ared + 1); else accel_export_fused(h, ared);   // (precision mode: the rank's change and its dot products leave one launch)
              prof_mark(d, s, PK_ACCEL);
              RCCLCHECK(R.AllReduce(ared, ared, 1 + 2 * POLAR_ACCEL_MAXM, ncclDouble, ncclSum, d->comm_red, s));
              d->allreduces++;
              prof_mark(d, s, PK_REDUCE);
              if (lazy) accel_step(h, ared + 1); else accel_decide_mix(h, ared, false);   // (the decision on the all-reduced change and the coefficients: one launch)
              prof_mark(d, s, PK_ACCEL);
            }
            if (lazy && (sw == max_sweeps - 2 || sw == max_sweeps - 1 || max_sweeps == 1))
              k_solver_step<<<1, POLAR_NSLOT, 0, s>>>(h->d_scal.p, h->d_slots.p, norm_count(h), st.fixed_iteration, st.iterations_max, st.polar_precision,
                                                    0, nullptr, (sw == max_sweeps - 2) ? max_sweeps - 1 : 1, det_part(h), det_npart(h));
          } else if (!st.fixed_iteration) {
            // the stop rule (PS.cpp:1194-1210) needs the sum over all ranks: one all-reduced double every `reduce_every` sweeps;
            // in between the end-of-sweep logic is told "not converged yet" (+inf)
            const double *gc = d->d_red.p + 1;
            if ((sw % d->reduce_every) == d->reduce_every - 1 || sw >= st.iterations_max) {
              k_fold_change<<<1, POLAR_NSLOT, 0, s>>>(h->d_scal.p, h->d_slots.p, d->d_red.p, det_part(h), det_npart(h));
              RCCLCHECK(R.AllReduce(d->d_red.p, d->d_red.p, 1, ncclDouble, ncclSum, d->comm_red, s));
              d->allreduces++;
              gc = d->d_red.p;
            }
            // Gauss-Seidel: the end-of-sweep logic and the pack of the halo dipoles in one launch (Jacobi flips its buffers in the
            // logic and packs afterwards)
            const long long ns = d->peers.empty() ? 0 : d->send_off[d->peers.size()];
            if (gs && ns > 0) {
              k_solver_step_gather<<<1 + nblk(ns, POLAR_NSLOT), POLAR_NSLOT, 0, s>>>(
                  h->d_scal.p, h->d_slots.p, norm_count(h), st.fixed_iteration, st.iterations_max, st.polar_precision, gc,
                  gc == d->d_red.p ? nullptr : det_part(h), gc == d->d_red.p ? 0 : det_npart(h), ns, d->d_send_idx.p,
                  h->sorted ? h->d_inv.p : nullptr, mu_view(h), d->d_send.p);
              packed = true;
            } else
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
          if (!accel) prof_mark(d, s, PK_REDUCE);
          dist_exchange(d, h, packed);
          prof_mark(d, s, PK_XCHG);
          if (!st.fixed_iteration && look_at_state(h, sw, d->check_every, d->reduce_every)) {
            dist_wait(d, s);
            read_scal(h);  // identical on every rank: same all-reduced sum (sweeps past the end are no-ops on the device)
            if (h->h_scal->done) break;
            prof_mark(d, s, PK_OTHER);
          }
        }
      }
      } catch (...) {
        if (R.CommAbort && d->comm) {
          if (d->comm_red && d->comm_red != d->comm) (void)R.CommAbort(d->comm_red);
          (void)R.CommAbort(d->comm);
          d->comm = d->comm_red = nullptr;
        }
        h->in_step = false;
        throw;
      }
      d->host_us_loop = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_loop).count();
      if (getenv("POLAR_DEBUG")) fprintf(stderr, "[polar] dist step (rank %d): sweep loop on the host %.0f us (of it inside RCCL calls %.0f us, state reads included), %d + %d exchanges, %d all-reduces\n",
                                         d->rank, d->host_us_loop, d->host_us_rccl, d->exchanges, d->phase_exchanges, d->allreduces);
      dist_wait(d, s);          // (the solve's last exchanges: polar_step_finish synchronises unconditionally)
      rc = polar_step_finish(h, out);
      prof_fold(d);
      d->local = *out;
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
      dist_wait(d, s);
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
int polar_dist_local_result(const polar_dist *d, polar_result *out) {
  if (!d || !out) return POLAR_ERR_STATE;
  *out = d->local;
  return POLAR_OK;
}
int polar_dist_profile(polar_dist *d, int enable) {
  if (!d) return POLAR_ERR_STATE;
  d->prof_on = enable != 0;
  return POLAR_OK;
}
int polar_dist_profile_get(const polar_dist *d, double *ms_by_part, int *intervals) {
  if (!d || !ms_by_part) return POLAR_ERR_STATE;
  for (int k = 0; k < POLAR_DIST_PROF_PARTS; k++) ms_by_part[k] = d->prof_ms[k];
  if (intervals) *intervals = d->prof_intervals;
  return POLAR_OK;
}
int polar_dist_counters(const polar_dist *d, int *exchanges, int *allreduces) {
  if (!d) return POLAR_ERR_STATE;
  if (exchanges) *exchanges = d->exchanges + d->phase_exchanges;
  if (allreduces) *allreduces = d->allreduces;
  return POLAR_OK;
}

}  // extern "C"
