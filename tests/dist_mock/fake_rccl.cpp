// TEST INFRASTRUCTURE -- an in-process stand-in for the nine RCCL entry points the library's multi-GPU driver uses
// (polar_dist_*, csrc/polar_dist.hip).  The "ranks" are THREADS of one process that share one GPU: a send / receive pair
// is a device-to-device copy, an all-reduce goes through the host.  Loaded through POLAR_RCCL_LIB; it exists so that the
// driver's own logic -- halo plans with several peers, the all-reduced stop rule and its cadence, the agreed retry, the
// summed results -- runs with more than one rank on the one-GPU test box.  It says nothing about RCCL itself.
//
// Semantics kept from the real thing: calls are made per rank, sends and receives of a group are matched pairwise by
// (source, destination) in issue order, collectives need every rank.  Semantics NOT kept: everything here is synchronous
// (the stream is drained before data moves), so no overlap is exercised.
#include <hip/hip_runtime.h>

#include <condition_variable>
#include <cstring>
#include <deque>
#include <map>
#include <mutex>
#include <string>
#include <vector>

extern "C" {
typedef struct fakeComm *ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclSuccess = 0, ncclInternalError = 3, ncclInvalidArgument = 4 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclDouble = 8 } ncclDataType_t;   // (only ncclDouble is used)
typedef enum { ncclSum = 0, ncclProd = 1, ncclMax = 2, ncclMin = 3 } ncclRedOp_t;
}

namespace {
struct Post { const void *ptr; size_t count; };
struct World {
  int nranks = 0, joined = 0;
  std::mutex m;
  std::condition_variable cv;
  std::map<std::pair<int, int>, std::deque<Post>> box;      // (src, dst) -> posted sends, in order
  std::map<std::pair<int, int>, long long> taken;            // (src, dst) -> sends the receiver has copied
  std::map<std::pair<int, int>, long long> posted;           // (src, dst) -> sends posted so far
  // all-reduce: generation barrier
  long long gen = 0;
  int arrived = 0;
  std::vector<std::vector<double>> contrib;
  std::vector<double> result;
};
std::mutex g_m;
std::map<std::string, World *> g_worlds;
unsigned g_next_id = 1;
}  // namespace

struct fakeComm {
  World *w;
  int rank;
  bool in_group = false;
  struct Op { bool send; void *ptr; size_t count; int peer; hipStream_t stream; };
  std::vector<Op> ops;
};

namespace {
ncclResult_t run_ops(fakeComm *c) {
  World *w = c->w;
  for (auto &o : c->ops) if (hipStreamSynchronize(o.stream) != hipSuccess) return ncclInternalError;   // send buffers are packed
  {
    std::lock_guard<std::mutex> g(w->m);
    for (auto &o : c->ops)
      if (o.send) { w->box[{c->rank, o.peer}].push_back(Post{o.ptr, o.count}); w->posted[{c->rank, o.peer}]++; }
  }
  w->cv.notify_all();
  for (auto &o : c->ops) {
    if (o.send) continue;
    Post p;
    {
      std::unique_lock<std::mutex> g(w->m);
      auto key = std::make_pair(o.peer, c->rank);
      w->cv.wait(g, [&] { return !w->box[key].empty(); });
      p = w->box[key].front();
    }
    if (p.count != o.count) return ncclInvalidArgument;   // the two ranks' plans disagree
    // on the receiver's stream, drained before the call returns: the driver's streams need not be ordered with the null
    // stream (its communication stream is a non-blocking one), and a device-to-device hipMemcpy may return early
    if (hipMemcpyAsync(o.ptr, p.ptr, o.count * sizeof(double), hipMemcpyDeviceToDevice, o.stream) != hipSuccess) return ncclInternalError;
    if (hipStreamSynchronize(o.stream) != hipSuccess) return ncclInternalError;
    {
      std::lock_guard<std::mutex> g(w->m);
      auto key = std::make_pair(o.peer, c->rank);
      w->box[key].pop_front();
      w->taken[key]++;
    }
    w->cv.notify_all();
  }
  // a sender may reuse its buffer only after the receiver has copied it
  {
    std::unique_lock<std::mutex> g(w->m);
    for (auto &o : c->ops) {
      if (!o.send) continue;
      auto key = std::make_pair(c->rank, o.peer);
      const long long want = w->posted[key];
      w->cv.wait(g, [&] { return w->taken[key] >= want; });
    }
  }
  c->ops.clear();
  return ncclSuccess;
}
}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId *id) {
  std::lock_guard<std::mutex> g(g_m);
  memset(id, 0, sizeof(*id));
  snprintf(id->internal, sizeof(id->internal), "fake-rccl-%u", g_next_id++);
  return ncclSuccess;
}
ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId id, int rank) {
  World *w;
  {
    std::lock_guard<std::mutex> g(g_m);
    std::string key(id.internal, strnlen(id.internal, sizeof(id.internal)));
    auto it = g_worlds.find(key);
    if (it == g_worlds.end()) { w = new World(); w->nranks = nranks; w->contrib.resize(nranks); g_worlds[key] = w; }
    else w = it->second;
  }
  if (w->nranks != nranks || rank < 0 || rank >= nranks) return ncclInvalidArgument;
  {
    std::unique_lock<std::mutex> g(w->m);
    w->joined++;
    w->cv.notify_all();
    w->cv.wait(g, [&] { return w->joined >= w->nranks; });   // like the real call: returns when every rank is there
  }
  *comm = new fakeComm{w, rank};
  return ncclSuccess;
}
ncclResult_t ncclCommDestroy(ncclComm_t c) { delete c; return ncclSuccess; }
ncclResult_t ncclGroupStart() { return ncclSuccess; }   // (ops are queued on the communicator and run at ncclGroupEnd)
namespace { thread_local std::vector<fakeComm *> t_open; }
ncclResult_t ncclSend(const void *buf, size_t count, ncclDataType_t dt, int peer, ncclComm_t c, hipStream_t s) {
  if (dt != ncclDouble) return ncclInvalidArgument;
  c->ops.push_back(fakeComm::Op{true, const_cast<void *>(buf), count, peer, s});
  if (t_open.empty() || t_open.back() != c) t_open.push_back(c);
  return ncclSuccess;
}
ncclResult_t ncclRecv(void *buf, size_t count, ncclDataType_t dt, int peer, ncclComm_t c, hipStream_t s) {
  if (dt != ncclDouble) return ncclInvalidArgument;
  c->ops.push_back(fakeComm::Op{false, buf, count, peer, s});
  if (t_open.empty() || t_open.back() != c) t_open.push_back(c);
  return ncclSuccess;
}
ncclResult_t ncclGroupEnd() {
  ncclResult_t rc = ncclSuccess;
  for (fakeComm *c : t_open) { const ncclResult_t r = run_ops(c); if (r != ncclSuccess) rc = r; }
  t_open.clear();
  return rc;
}
ncclResult_t ncclAllReduce(const void *send, void *recv, size_t count, ncclDataType_t dt, ncclRedOp_t op, ncclComm_t c, hipStream_t s) {
  if (dt != ncclDouble || (op != ncclSum && op != ncclMax)) return ncclInvalidArgument;
  World *w = c->w;
  std::vector<double> mine(count);
  if (hipStreamSynchronize(s) != hipSuccess) return ncclInternalError;
  if (hipMemcpyAsync(mine.data(), send, count * sizeof(double), hipMemcpyDeviceToHost, s) != hipSuccess) return ncclInternalError;
  if (hipStreamSynchronize(s) != hipSuccess) return ncclInternalError;
  std::vector<double> out;
  {
    std::unique_lock<std::mutex> g(w->m);
    const long long my_gen = w->gen;
    w->contrib[c->rank] = mine;
    if (++w->arrived == w->nranks) {
      w->result.assign(count, op == ncclSum ? 0.0 : -1.0e300);
      for (int r = 0; r < w->nranks; r++) {       // rank order: the same sum on every rank, as a real ring gives
        if (w->contrib[r].size() != count) { w->result.clear(); break; }
        for (size_t k = 0; k < count; k++)
          w->result[k] = op == ncclSum ? w->result[k] + w->contrib[r][k] : (w->contrib[r][k] > w->result[k] ? w->contrib[r][k] : w->result[k]);
      }
      w->arrived = 0;
      w->gen++;
      w->cv.notify_all();
    } else {
      w->cv.wait(g, [&] { return w->gen != my_gen; });
    }
    out = w->result;
  }
  if (out.size() != count) return ncclInvalidArgument;   // the ranks disagree about the collective
  if (hipMemcpyAsync(recv, out.data(), count * sizeof(double), hipMemcpyHostToDevice, s) != hipSuccess) return ncclInternalError;
  if (hipStreamSynchronize(s) != hipSuccess) return ncclInternalError;   // (`out` is a stack vector)
  return ncclSuccess;
}
const char *ncclGetErrorString(ncclResult_t r) {
  return r == ncclSuccess ? "no error" : r == ncclInvalidArgument ? "invalid argument (fake RCCL: the ranks' calls do not match)" : "internal error (fake RCCL)";
}

}  // extern "C"
