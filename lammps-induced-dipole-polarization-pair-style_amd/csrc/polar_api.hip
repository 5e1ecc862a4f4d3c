// polar_api.hip -- C-ABI implementation (include/polar_mi355x.h): handle, HBM-resident state,
// and the orchestration of PairLJCutCoulLongPolarization::compute (PS.cpp:125-645) on one MI355X.
// All arithmetic of the hot path runs in the kernels of polar_kernels.hpp; this file holds no
// CPU implementation of it (no fallback: without a GPU every compute call fails).
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <rccl/rccl.h>   // types only: the entry points are resolved at run time (polar_dist_*), the library does not link librccl

#include <algorithm>
#include <initializer_list>
#include <utility>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <numeric>
#include <string>
#include <thread>
#include <mutex>
#include <condition_variable>
#include <functional>
#include <memory>
#include <unistd.h>
#include <vector>

#include "pair_host.hpp"
#include "polar_kernels.hpp"

using namespace polar;

// bump when a kernel on the hot path changes: PMC files under profiles/ are keyed by it (bench.py, roofline.traffic)
#define POLAR_KERNEL_VERSION "r03-lp3-v3"

namespace {

struct HipError : std::runtime_error {
  explicit HipError(const std::string &m) : std::runtime_error(m) {}
};
struct NoDevice : std::runtime_error {
  NoDevice() : std::runtime_error("no usable HIP device: this library has no CPU fallback") {}
};
#define HIPCHECK(expr)                                                                             \
  do {                                                                                             \
    hipError_t e_ = (expr);                                                                        \
    if (e_ != hipSuccess)                                                                          \
      throw HipError(std::string(#expr) + " failed: " + hipGetErrorString(e_) + " (" + __FILE__ + \
                     ":" + std::to_string(__LINE__) + ")");                                      \
  } while (0)

template <typename T>
struct DBuf {
  T *p = nullptr;
  size_t cap = 0;
  void ensure(size_t n) {
    if (n <= cap) return;
    if (p) HIPCHECK(hipFree(p));
    size_t want = n + n / 8 + 64;
    HIPCHECK(hipMalloc((void **)&p, want * sizeof(T)));
    cap = want;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr; cap = 0;
  }
};

// at least one workgroup: every kernel bounds-checks its index, and a zero-sized grid is a launch error
inline int nblk(long long n, int per) { return n <= 0 ? 1 : (int)((n + per - 1) / per); }
// one launch zeroing up to six small device buffers (sizes in bytes, multiples of 4)
inline void zero_many(hipStream_t s, std::initializer_list<std::pair<void *, size_t>> bufs) {
  ZeroJobs jobs;
  jobs.n = 0;
  unsigned long long most = 0;
  for (const auto &b : bufs) {
    jobs.p[jobs.n] = (unsigned int *)b.first;
    jobs.nwords[jobs.n] = b.second / 4;
    most = std::max<unsigned long long>(most, b.second / 4);
    jobs.n++;
  }
  const int blocks = (int)std::min<unsigned long long>(1024, (most + 255) / 256);
  k_zero_many<<<std::max(blocks, 1), 256, 0, s>>>(jobs);
}
// grid of a kernel that places its workgroups with xcd_block(): 8 * ceil(nblocks / 8)
inline int nblk_xcd(long long n, int per) { return ((nblk(n, per) + 7) / 8) * 8; }

// widths of the simulation cell between opposite faces: the box lengths, or V / |face area| when the box is tilted
inline void box_widths(const Box &b, double w[3]) {
  w[0] = b.prd[0]; w[1] = b.prd[1]; w[2] = b.prd[2];
  if (!b.triclinic) return;
  const double vol = b.prd[0] * b.prd[1] * b.prd[2];
  const double bxc[3] = {b.prd[1] * b.prd[2], -b.xy * b.prd[2], b.xy * b.yz - b.prd[1] * b.xz};  // b x c
  w[0] = vol / std::sqrt(bxc[0] * bxc[0] + bxc[1] * bxc[1] + bxc[2] * bxc[2]);
  w[1] = vol / (b.prd[0] * std::sqrt(b.prd[2] * b.prd[2] + b.yz * b.yz));                      // |a x c|
}
// squared minimum-image distance on the host (colouring, clustering): the rule of min_image_rint
inline double min_image_dist2(const Box &b, const double *xi, const double *xj) {
  double d[3] = {xi[0] - xj[0], xi[1] - xj[1], xi[2] - xj[2]};
  if (b.triclinic) {
    if (b.periodic[2]) { const double n = std::nearbyint(d[2] / b.prd[2]); d[2] -= n * b.prd[2]; d[1] -= n * b.yz; d[0] -= n * b.xz; }
    if (b.periodic[1]) { const double n = std::nearbyint(d[1] / b.prd[1]); d[1] -= n * b.prd[1]; d[0] -= n * b.xy; }
    if (b.periodic[0]) d[0] -= b.prd[0] * std::nearbyint(d[0] / b.prd[0]);
  } else {
    for (int k = 0; k < 3; k++)
      if (b.periodic[k]) d[k] -= b.prd[k] * std::nearbyint(d[k] / b.prd[k]);
  }
  return d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
}

}  // namespace

struct polar_handle {
  int device = -1;
  bool have_device = false;
  hipStream_t stream = nullptr;
  std::string err, warn;
  PairHost ph;
  bool types_set = false, coul_set = false, box_set = false, atoms_set = false, neigh_set = false;
  // raw-setter copies (when the shim passes LAMMPS' own tables)
  int ntypes = 0;
  LJCoulParams P{};
  Box box{};
  double boxlo[3] = {0, 0, 0};
  int nlocal = 0, nghost = 0;
  int row_lo = 0, row_hi = -1;   // rows this handle owns (multi-GPU row sharding); -1 = all
  int full_list = 0;             // LJ/coul list is a LAMMPS full list
  int newton_pair = 1;           // force->newton_pair of the uploaded half list (polar_set_newton)
  int step_eflag = 0, step_vflag = 0;
  bool in_step = false;
  bool own_stream = true;
  // host mirrors needed by host-side colouring
  std::vector<double> hx, halpha;
  // device state
  DBuf<double> d_x, d_q, d_alpha, d_f, d_ef, d_F, d_mu, d_rank, d_dmu, d_tab, d_lj;
  DBuf<int> d_type, d_mol, d_order, d_pos, d_ilist, d_numneigh, d_neigh, d_rows;
  DBuf<int> d_mol_s, d_perm, d_inv, d_rows_orig, d_ownrows;  // s-space bookkeeping (see polar_kernels.hpp)
  DBuf<double> d_ef_s, d_T6, d_eatom, d_vatom, d_dd_r2, d_fpol;
  bool dense_gs = false;   // exact-order GS on the HBM-resident tensor (atoms in sweep order)
  CellGrid grid{};
  long long ncell = 0;
  bool sorted = false;  // true while the records are in cell order (list mode)
  DBuf<long long> d_first, d_sym_first;
  DBuf<int> d_sym_cnt, d_sym_fill, d_sym_j;
  bool sym_valid = false;  // symmetrised list matches the uploaded half list
  DBuf<AtomRec> d_rec0, d_rec1;
  DBuf<Scal> d_scal;
  DBuf<double> d_slots;
  // cutoff-mode lists
  DBuf<int> d_cell_id, d_cell_cnt, d_cell_fill, d_nl_cnt, d_dd_cnt, d_nl_j, d_dd_j, d_dd_wrap;
  DBuf<long long> d_cell_first, d_nl_first, d_dd_first;
  DBuf<double2> d_dd_s;
  DBuf<int2> d_lpdesc;  // row descriptors of k_field_lp
  DBuf<double> d_lp_pend;  // `deterministic yes`: {mu_x, mu_y, mu_z, (dmu)^2} per launch row until k_lp_commit
  DBuf<double> d_lp_part;  // ... and the sums of (dmu)^2 per 256 launch rows (k_solver_step adds them in order)
  int lp_npart = 0;
  DBuf<int> d_slot;     // lp sweep: launch row of every atom's dd row (s space), -1: none
  // cluster rows (sweep_kernel 3, k_field_cl): clusters sorted by colour; color_off then counts clusters
  std::vector<int> h_cl;        // [ncl][4] member atoms (orig ids, -1 padded)
  DBuf<int> d_cl_orig, d_cl_cnt, d_cl_wrap, d_cl_tw;
  DBuf<int4> d_cl_s;            // members in this step's s space
  long long cl_pitch = 0;
  int ncl = 0;
  double cluster_dist = 2.0;    // A: largest distance between two members (POLAR_CLUSTER_DIST)
  int cluster_max = 4;          // members per cluster, 1..4 (POLAR_CLUSTER_MAX)
  long long cl_slots = 0;       // entries of the union lists (gathered records per sweep)
  DBuf<double4> d_xq, d_pos4, d_xq_s;
  long long nl_pairs = 0, dd_pairs = 0;
  long long nl_pitch = 0, dd_pitch = 0;   // pitched row lists (see polar_kernels.hpp RowList)
  DBuf<int> d_overflow;
  DBuf<unsigned long long> d_ddtot;
  int *h_flags = nullptr;                   // pinned: [0] overflow (needed count), [1..] unused
  unsigned long long *h_ddtot = nullptr;    // pinned: 64 x 16 partial totals
  int inum = 0;
  long long nneigh = 0;
  bool mu_resident = false;
  DBuf<double> d_dbgf;           // `debug yes`: {force on atom 0, its dipole-dipole part} of the last compute
  bool mu_host_in_sync = false;  // the caller's mu array still holds what the last polar_compute returned (no polar_set_atoms since)
  // colour phases (cutoff-mode Gauss-Seidel)
  std::vector<int> color_off;  // [ncolors+1] offsets into d_rows
  std::vector<int> h_rows;     // rows sorted by colour (host copy)
  bool colors_valid = false;
  double color_keep = 2.0;      // A (POLAR_COLOR_KEEP): a colouring built at color_dist stays in use on later lists while
                                // every same-colour pair is still farther apart than this (hysteresis: atoms move)
  bool slots_by_color = false;  // the dd rows of the current lists are laid out in colour-phase order (compute_slots)
  long long color_epoch = 0;    // counts colourings (build_colors); the dd rows are usable only while laid out for the current one
  long long slots_epoch = -1;   // the colouring compute_slots laid the rows out for
  bool colors_recheck = false;  // a new neighbor list arrived: keep the colouring if it still separates every same-colour pair
  std::vector<int> h_color;     // colour of every atom (orig ids), -1: none
  DBuf<int> d_color_orig, d_color_s;
  DBuf<int> d_klist;              // device colouring: rows of the top class and their repair states (k_color_kempe)
  DBuf<int> d_cadj, d_cdeg, d_ccnt, d_cflags, d_crelabel;  // device colouring: conflict lists, degrees, rows per (colour, cell), round counters
  DBuf<unsigned long long> d_cprio;
  DBuf<double> d_cstat;
  DBuf<long long> d_coff;
  int *h_cflags = nullptr;        // pinned: round counters / fold counters of the device colouring
  double *h_cstat = nullptr;      // pinned: rows and rank sums per colour
  long long *h_coff = nullptr;    // pinned: first row of every phase
  int cadj_pitch = 16;            // conflict-list entries per atom (grown when an atom has more neighbours within the colour distance)
  int host_colors = 0;            // lab (POLAR_HOST_COLORS): rounds 1-2's host-side conflict graph + DSATUR instead of the device colouring
  int colors_reused = 0, colors_rebuilt = 0;
  double ms_color_host = 0.0;  // host time of the last colour rebuild; reported once, then cleared
  double color_dist = 2.4;  // A (POLAR_COLOR_DIST).  profiles/r01_lab_color_distance.txt: 2.4 -> 4 phases, 2.5-2.6 -> 5, with the same
                            // number of sweeps to 1e-11 (33); <= 2.2 -> 3 phases but 36-37 sweeps; <= 1.2 does not converge
  int field_block = 256;
  double bbox_lo[3] = {0, 0, 0}, bbox_hi[3] = {0, 0, 0};  // locals + ghosts, recorded by polar_set_atoms
  long long global_count = 0;  // N of the stop rule when the handle holds a part of the system (0: nlocal)
  DBuf<double> d_xchg; DBuf<int> d_xidx;  // staging of the host-pointer exchange forms
  bool device_list = false;  // the a3 list was built by polar_build_neighbors (always a full list)
  int user_full_list = 0;    // polar_set_list_style for uploaded lists
  long long lj_pitch = 0;
  DBuf<double4> d_ljpos; DBuf<int2> d_ljaux; DBuf<int> d_tag, d_nspecial, d_special, d_ljcell_id, d_ljcell_cnt, d_ljcell_fill; DBuf<long long> d_ljcell_first; DBuf<double> d_cutneighsq;
  int lj_typed = 1;              // LJ/Coulomb list entries carry the partner's type (POLAR_LJ_TYPED=0)
  bool sym_typed = false, dev_typed = false;
  int static_xq = 1;             // the static-field rows gather 32-byte {x, y, z, q} records instead of whole AtomRecs (POLAR_STATIC_XQ=0)
  int pol_first = 1;             // polarizable atoms first inside a cell (POLAR_POL_FIRST=0: arrival order)
  int part_k = 0, part_n = 1;    // polar_step_sweep_part: which share of the colour phases the next sweep_once runs
  int lp_wg_per_cu = 0;          // lab (POLAR_LP_WG_PER_CU): workgroups of k_field_lp resident per CU, capped through the LDS size
  int lp_quad_major = 1;         // slot order of the lp index stream (lp_slot), POLAR_LP_QM=0: lane = entry
  int quad_block = POLAR_BLOCK;  // workgroup size of k_field_quad / k_field_lp (POLAR_QUAD_BLOCK)
  int lp_tiles = 2;              // LDS tiles per wave of k_field_lp (POLAR_LP_TILES: 1 or 2)
  int lp_rows = 1;               // launch rows per wave (k_field_lpr when > 1; POLAR_LP_ROWS)
  int lp_pairs = 0;              // lab (POLAR_LP_PAIRS=1): paired rows, k_field_lp2 over union lists
  DBuf<int> d_ulead, d_udd_j;    // paired rows: leader flags per launch row, union lists
  DBuf<long long> d_upos;        // ... unit number of every leader (scan)
  DBuf<int2> d_unit;             // ... {row atom A, row atom B or -1} per unit
  DBuf<int4> d_udesc;            // ... {A, B, trips | wrap << 30, entries}
  std::vector<int> unit_off;     // ... first unit of every phase
  long long upitch = 0;
  int lp_depth = 0;              // >= 2: k_field_lpa with the gathers that many trips ahead (POLAR_LP_DEPTH: 0, 2, 3)
  int cache_r2 = -1;      // sweep stream (POLAR_CACHE_R2): 0 cached (s3,s5), 20 B/pair; 1 cached r^2, 12 B/pair; 2 nothing,
                          // 4 B/pair (r^2 rebuilt from the gathered positions); -1: 1 or 2 by size, see build_lists
  int stream_mode = 1;    // the choice in force for the current lists
  int sweep_kernel = 2;   // list-mode sweep (POLAR_SWEEP_KERNEL): 4 k_field_tile (one workgroup per cell, neighbour records staged in LDS),
                          // 2 k_field_lp (one wave per row, LDS-DMA gathers), 0 k_field_quad (component-per-lane, round 1),
                          // 1 k_field (register-staged lane-per-pair), 3 k_field_cl (cluster rows, experimental)
  int ablate = 0;  // lab switches for k_field (POLAR_ABLATE), 0 in production
  // tile sweep (sweep_kernel 4, polar_tiles.hpp): sweep records, tile headers, row table, union lists, 16-bit row lists
  DBuf<SRec> d_srec0, d_srec1;
  DBuf<TileHdr> d_thdr;
  DBuf<TileRowEnt> d_trow;
  DBuf<int> d_un_j;
  DBuf<unsigned short> d_dd16;
  DBuf<double> d_pend;             // `deterministic yes`: dipoles of a launch's rows until k_tile_commit
  int un_pitch = 0;                // union entries per tile (pitch of d_un_j)
  int un_lds = 0;                  // records (dummy included) the sweep's LDS request holds; the builder refuses larger unions
  long long pitch16 = 0;           // entries per row of d_dd16 (a multiple of 512 = 8 trips)
  int tile_max_u = 0;              // largest union of the last step
  bool tile_reported = false;
  double dens = 0.0;               // atoms per A^3 of the occupied part of the box (first list build)
  std::vector<TileLaunch> tile_launches;  // Gauss-Seidel: one launch per tile colour; tile_all: every cell (Jacobi)
  TileLaunch tile_all{};
  size_t tile_lds_attr[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};  // dynamic-LDS limit already raised per kernel instance
  size_t tile_build_lds_attr = 0;
  int tile_waves = 4;              // waves of a sweep workgroup = rows of a tile's sub-phase (POLAR_TILE_WAVES: 4 or 8)
  int tile_wide = 0;               // cells of a whole cutoff in x (tiles of ~30 rows) instead of half a cutoff (~15) (POLAR_TILE_WIDE)
  int tile_sw[3] = {2, 2, 2};      // stencil half-widths of the tile builder, in cells
  int deterministic = 0;           // POLAR_DETERMINISTIC / `deterministic yes`: no sweep reads a dipole another wave of the same launch writes
  Scal *h_scal = nullptr;  // pinned
  hipEvent_t ev[8] = {};
  // a3 runs on its own stream beside the list build / static field / dipole solve (it only shares the
  // force and tally accumulators with them): fork after the accumulators are zeroed, join before they are read
  hipStream_t lj_stream = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_lj0 = nullptr, ev_lj1 = nullptr, ev_dl0 = nullptr, ev_dl1 = nullptr;
  // polar_compute: the dipoles and the static field are final before the force kernel starts; they travel to the host on
  // their own stream while it runs (early_mu / early_ef: where in the pinned staging area; null = not asked for)
  hipStream_t dl_stream = nullptr;
  hipEvent_t ev_mu_ready = nullptr;
  double *early_mu = nullptr, *early_ef = nullptr, *user_mu = nullptr, *user_ef = nullptr;  // staging slots; the caller's arrays
  hipEvent_t ev_fchunk[4] = {nullptr, nullptr, nullptr, nullptr};
  bool overlap_lj = true;  // POLAR_NO_OVERLAP=1 keeps a3 on the main stream
  bool lj_forked = false;
  std::vector<double> h_tmp;
  double *h_stage = nullptr;  // pinned staging area for downloads
  DBuf<double> d_trace;       // `debug yes`: u_polar after every sweep of the last solve
  int ntrace = 0;
  size_t h_stage_cap = 0;
};

namespace {

int fail(polar_handle *h, int code, const std::string &m) {
  if (h) h->err = m;
  return code;
}

template <typename F>
int guarded(polar_handle *h, F &&fn) {
  if (!h) return POLAR_ERR_STATE;
  try {
    return fn();
  } catch (const InputError &e) {
    return fail(h, POLAR_ERR_INPUT, e.what());
  } catch (const NoDevice &e) {
    return fail(h, POLAR_ERR_NO_DEVICE, e.what());
  } catch (const HipError &e) {
    return fail(h, POLAR_ERR_HIP, e.what());
  } catch (const std::exception &e) {
    return fail(h, POLAR_ERR_STATE, e.what());
  }
}

// pinned host staging area of at least `count` doubles (grown geometrically, freed with the handle)
double *staging(polar_handle *h, size_t count) {
  if (count > h->h_stage_cap) {
    if (h->h_stage) {  // (an upload or an early download may still be using the old area)
      if (h->stream) (void)hipStreamSynchronize(h->stream);
      if (h->dl_stream) (void)hipStreamSynchronize(h->dl_stream);
      (void)hipHostFree(h->h_stage);
    }
    h->h_stage = nullptr; h->h_stage_cap = 0;
    const size_t want = count + count / 4 + 1024;
    HIPCHECK(hipHostMalloc((void **)&h->h_stage, want * sizeof(double)));
    h->h_stage_cap = want;
  }
  return h->h_stage;
}

void need_device(polar_handle *h) {
  if (!h->have_device) throw NoDevice();
}

// host-side array work of a compute call (adding 4 MB of forces into the caller's array, copying dipoles and fields out of
// the staging area): one thread moves ~10 GB/s, the PCIe link brings the data three times faster -- a few short-lived
// threads, each on its own contiguous quarter
// Three helper threads that live as long as the library (a std::thread per call cost 30-50 us each: 0.3 ms per MD step over
// the three copies of a step).  run(fn, parts): fn(k) for k = 1 .. parts-1 on the helpers, fn(0) on the caller; returns when
// all are done.  One job at a time (the library's host copies are serial per process anyway).
class HostPool {
 public:
  static HostPool &get() { static HostPool p; return p; }
  int width() const { return (int)th_.size() + 1; }
  void run(const std::function<void(int)> &fn, int parts) {
    if (parts <= 1 || th_.empty() || getpid() != pid_) { for (int k = 0; k < parts; k++) fn(k); return; }   // (a forked child has no helpers)
    std::unique_lock<std::mutex> job(job_m_);   // one job at a time
    {
      std::lock_guard<std::mutex> g(m_);
      fn_ = &fn; parts_ = parts; pending_ = std::min(parts - 1, (int)th_.size()); gen_++;
    }
    cv_.notify_all();
    fn(0);
    for (int k = (int)th_.size() + 1; k < parts; k++) fn(k);   // (more parts than threads: the caller takes the rest)
    std::unique_lock<std::mutex> g(m_);
    done_.wait(g, [&] { return pending_ == 0; });
    fn_ = nullptr;
  }
 private:
  HostPool() {
    pid_ = getpid();
    const unsigned hw = std::thread::hardware_concurrency();
    const int n = hw >= 4 ? 3 : (hw >= 2 ? (int)hw - 1 : 0);
    for (int t = 0; t < n; t++) th_.emplace_back([this, t]() { loop(t); });
  }
  ~HostPool() {
    { std::lock_guard<std::mutex> g(m_); stop_ = true; gen_++; }
    cv_.notify_all();
    for (auto &t : th_) t.join();
  }
  void loop(int t) {
    unsigned long long seen = 0;
    for (;;) {
      const std::function<void(int)> *fn = nullptr;
      int part = -1;
      {
        std::unique_lock<std::mutex> g(m_);
        cv_.wait(g, [&] { return gen_ != seen; });
        seen = gen_;
        if (stop_) return;
        if (t + 1 < parts_) { fn = fn_; part = t + 1; }
      }
      if (fn) {
        (*fn)(part);
        std::lock_guard<std::mutex> g(m_);
        if (--pending_ == 0) done_.notify_all();
      }
    }
  }
  std::vector<std::thread> th_;
  std::mutex m_, job_m_;
  std::condition_variable cv_, done_;
  const std::function<void(int)> *fn_ = nullptr;
  int parts_ = 0, pending_ = 0;
  unsigned long long gen_ = 0;
  bool stop_ = false;
  pid_t pid_ = 0;
};
template <typename F>
void host_chunks(size_t total, F &&fn) {
  HostPool &pool = HostPool::get();
  const size_t nt = total < (1u << 16) ? 1 : (size_t)pool.width();
  if (nt <= 1) { fn((size_t)0, total); return; }
  const size_t per = (total + nt - 1) / nt;
  pool.run([&](int k) { fn(std::min(total, (size_t)k * per), std::min(total, ((size_t)k + 1) * per)); }, (int)nt);
}

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
  h->coul_set = true;
}

// ---- cutoff-mode: cell sort (perm/inv), then CSR lists in s space, all on the device ----------
inline int own_lo(const polar_handle *h) { return h->row_lo; }
inline int own_n(const polar_handle *h) { return (h->row_hi < 0 ? h->nlocal : h->row_hi) - h->row_lo; }
inline int norm_count(const polar_handle *h) { return (int)(h->global_count > 0 ? h->global_count : h->nlocal); }
inline bool sharded(const polar_handle *h) { return own_n(h) != h->nlocal; }
// the dd rows of the current lists sit in the launch order of the colouring in force (a colouring rebuilt after the
// lists were laid out -- a clash found on a reneighbor step, a changed alpha pattern -- makes them stale)
inline bool slots_current(const polar_handle *h) { return h->slots_by_color && h->slots_epoch == h->color_epoch; }
inline bool deterministic(const polar_handle *h) { return h->deterministic || h->ph.st.deterministic; }
// `deterministic yes` with the row sweep: where the end-of-sweep kernels find the sweep's partial sums of (dmu)^2
inline const double *det_part(const polar_handle *h) { return (deterministic(h) && h->ph.st.dd_cutoff > 0.0 && h->sweep_kernel == 2 && h->lp_npart > 0) ? h->d_lp_part.p : nullptr; }
inline int det_npart(const polar_handle *h) { return det_part(h) ? h->lp_npart : 0; }
inline bool tile_mode(const polar_handle *h) { return h->ph.st.dd_cutoff > 0.0 && h->sweep_kernel == 4; }
// where the dipoles live during a solve (exchange and debug kernels): the sweep records in tile mode, else the AtomRecs
inline MuView mu_view(const polar_handle *h) {
  if (tile_mode(h)) return MuView{reinterpret_cast<char *>(h->d_srec0.p), reinterpret_cast<char *>(h->d_srec1.p), (int)sizeof(SRec)};
  return MuView{reinterpret_cast<char *>(h->d_rec0.p), reinterpret_cast<char *>(h->d_rec1.p), (int)sizeof(AtomRec)};
}
// rows a per-row kernel should visit: nullptr = all rows 0..n-1 (identity)
inline const int *own_rows(const polar_handle *h) { return (h->sorted && sharded(h)) ? h->d_ownrows.p : nullptr; }

void build_cells(polar_handle *h) {
  const polar_settings &st = h->ph.st;
  const int n = h->nlocal;
  const double cutall = std::max(st.cut_coul, st.dd_cutoff);
  double width[3];
  box_widths(h->box, width);
  for (int k = 0; k < 3; k++)
    if (h->box.periodic[k] && width[k] < 2.0 * cutall * (1.0 - 1e-12))
      throw InputError("dd_cutoff mode needs box lengths >= 2*max(cut_coul,dd_cutoff); use exact mode (dd_cutoff 0)");
  CellGrid &g = h->grid;
  g.trim = 1;
#ifdef POLAR_LAB
  if (const char *e = getenv("POLAR_NL_TRIM")) g.trim = atoi(e);
#endif
  if (h->box.triclinic) g.trim = 0;  // the per-atom stencil trimming measures orthogonal distances
  long long ncell = 1;
  for (int k = 0; k < 3; k++) {
    g.nc[k] = std::max(1, (int)std::floor(width[k] / (0.5 * cutall)));  // cell height >= cutoff/2: +-2 stencil
    // tile sweep: the launches are the parity classes of the cells; an even count in a periodic dimension needs no third
    // class for the seam (cells grow by at most 1/6)
    if (h->sweep_kernel == 4 && k == 0 && h->tile_wide) g.nc[k] = std::max(1, (int)std::floor(width[k] / cutall));
    if (h->sweep_kernel == 4 && h->box.periodic[k] && g.nc[k] >= 7 && (g.nc[k] & 1)) g.nc[k] -= 1;
    h->tile_sw[k] = std::max(1, std::min(2, (int)std::ceil(st.dd_cutoff / (width[k] / g.nc[k]) - 1e-9)));
    g.lo[k] = h->boxlo[k];
    g.inv[k] = g.nc[k] / h->box.prd[k];
    ncell *= g.nc[k];
  }
  h->ncell = ncell;
  if (h->sweep_kernel == 4) {  // tile colours: per dimension the even cells, the odd cells and -- odd count, periodic -- the last cell
    struct Cls { int start, stride, count; };
    std::vector<Cls> cls[3];
    for (int k = 0; k < 3; k++) {
      const int nc = g.nc[k];
      const bool seam = h->box.periodic[k] && (nc & 1) && nc > 1;
      const int lim = seam ? nc - 1 : nc;
      if ((lim + 1) / 2 > 0) cls[k].push_back(Cls{0, 2, (lim + 1) / 2});
      if (lim / 2 > 0) cls[k].push_back(Cls{1, 2, lim / 2});
      if (seam) cls[k].push_back(Cls{nc - 1, 1, 1});
    }
    h->tile_launches.clear();
    for (const Cls &cz : cls[2]) for (const Cls &cy : cls[1]) for (const Cls &cx : cls[0]) {
      TileLaunch L;
      L.start[0] = cx.start; L.stride[0] = cx.stride; L.count[0] = cx.count;
      L.start[1] = cy.start; L.stride[1] = cy.stride; L.count[1] = cy.count;
      L.start[2] = cz.start; L.stride[2] = cz.stride; L.count[2] = cz.count;
      for (int k = 0; k < 3; k++) L.nc[k] = g.nc[k];
      h->tile_launches.push_back(L);
    }
    for (int k = 0; k < 3; k++) { h->tile_all.start[k] = 0; h->tile_all.stride[k] = 1; h->tile_all.count[k] = g.nc[k]; h->tile_all.nc[k] = g.nc[k]; }
  }
  hipStream_t s = h->stream;
  h->d_cell_id.ensure(n); h->d_cell_cnt.ensure(ncell + 1); h->d_cell_fill.ensure(2 * (ncell + 1));
  h->d_cell_first.ensure(ncell + 2); h->d_perm.ensure(n + 1); h->d_inv.ensure(n + 1);
  zero_many(s, {{h->d_cell_cnt.p, (size_t)(ncell + 1) * sizeof(int)}, {h->d_cell_fill.p, 2 * (size_t)(ncell + 1) * sizeof(int)}});
  k_cell_count<<<nblk(n, 256), 256, 0, s>>>(n, h->d_x.p, g, h->box, h->d_cell_id.p, h->d_cell_cnt.p);
  k_exclusive_scan<int><<<1, 1024, 0, s>>>(ncell, h->d_cell_cnt.p, h->d_cell_first.p);
  k_cell_fill<<<nblk(n, 256), 256, 0, s>>>(n, h->d_cell_id.p, h->d_cell_first.p, h->d_cell_fill.p,
                                           (h->pol_first || h->sweep_kernel == 4) ? h->d_cell_fill.p + ncell + 1 : nullptr, h->d_alpha.p, h->d_perm.p, h->d_inv.p);
  // the order inside a cell follows the atomics of k_cell_fill: put it into atom order -- always, not only for `deterministic
  // yes` (reproducible sums): the device colouring breaks its ties by position in the cell, and a colouring that changed from
  // run to run would make unconverged (`fixed_iteration`) results differ at 1e-6 instead of the 1e-9 of the in-place race
  k_cell_sort<<<nblk(ncell, 4), 256, 0, s>>>(ncell, h->d_cell_first.p, (h->pol_first || h->sweep_kernel == 4) ? h->d_cell_fill.p : nullptr, h->d_perm.p, h->d_inv.p);
  h->sorted = true;
  if (sharded(h)) {
    h->d_ownrows.ensure(own_n(h) + 1);
    k_map_range<<<nblk(own_n(h), 256), 256, 0, s>>>(own_lo(h), own_n(h), h->d_inv.p, h->d_ownrows.p);
  }
}

// lp sweep: where k_nl_build stores every atom's dd row = its row in launch order (colour phases back to back for GS,
// own rows for Jacobi).  While no colouring exists yet (first step, or after a clash) the rows are laid out in atom
// order and the lists are built once more after the colouring (solve / polar_step_begin).
void compute_slots(polar_handle *h) {
  const polar_settings &st = h->ph.st;
  const bool gs = st.polar_gs || st.polar_gs_ranked;
  const int n = h->nlocal;
  hipStream_t s = h->stream;
  h->d_slot.ensure(n + 1);
  HIPCHECK(hipMemsetAsync(h->d_slot.p, 0xFF, (size_t)(n + 1) * sizeof(int), s));
  if (gs && h->colors_valid) {
    const int tot = h->color_off.empty() ? 0 : h->color_off.back();
    if (tot > 0) k_slot_from_rows<<<nblk(tot, 256), 256, 0, s>>>(tot, h->d_rows.p, h->d_slot.p);
    h->slots_by_color = true;
    h->slots_epoch = h->color_epoch;
  } else {
    k_slot_from_rows<<<nblk(own_n(h), 256), 256, 0, s>>>(own_n(h), own_rows(h), h->d_slot.p);
    h->slots_by_color = false;
  }
}

void build_lists(polar_handle *h) {
  const polar_settings &st = h->ph.st;
  const int n = h->nlocal;
  const double cutall = std::max(st.cut_coul, st.dd_cutoff);
  hipStream_t s = h->stream;
  const CellGrid &g = h->grid;
  if (h->nl_pitch == 0 && getenv("POLAR_INIT_PITCH")) h->nl_pitch = h->dd_pitch = ((std::max(64, atoi(getenv("POLAR_INIT_PITCH"))) + 63) / 64) * 64;  // tests: force the overflow path
  if (h->nl_pitch == 0) {  // first build: 1.5x the mean sphere population, rounded to 64
    // density over the OCCUPIED part of the box (a shard handle holds one slab plus its halo, not the
    // whole box): count the non-empty cells of a coarse host grid with edge ~cutoff
    double vol = h->box.prd[0] * h->box.prd[1] * h->box.prd[2];
    {
      int nc[3];
      long long tot = 1;
      for (int k = 0; k < 3; k++) { nc[k] = std::max(1, std::min(64, (int)(h->box.prd[k] / cutall))); tot *= nc[k]; }
      std::vector<char> occ((size_t)tot, 0);
      for (int a = 0; a < n; a++) {
        long long c = 0, mul = 1;
        for (int k = 0; k < 3; k++) {
          double fr = (h->hx[3 * (size_t)a + k] - h->grid.lo[k]) / h->box.prd[k];
          fr -= std::floor(fr);
          c += mul * std::min(nc[k] - 1, (int)(fr * nc[k]));
          mul *= nc[k];
        }
        occ[(size_t)c] = 1;
      }
      long long filled = 0;
      for (char v : occ) filled += v;
      if (filled > 0) vol *= (double)filled / (double)tot;
    }
    h->dens = n / vol;
    double mean = n / vol * 4.18879020478639 * cutall * cutall * cutall;
    h->nl_pitch = h->dd_pitch = (((long long)(1.5 * mean) + 64) / 64 + 1) * 64;
  }
  if (h->sweep_kernel == 2) h->dd_pitch = ((h->dd_pitch + 255) / 256) * 256;  // whole 4-trip chunks (k_field_lp)
  // the lp / cluster index streams hold byte offsets j << 6 in 32-bit words: the record table must stay below 2^31 bytes
  if ((h->sweep_kernel == 2 || h->sweep_kernel == 3) && ((long long)n + 1) * (long long)sizeof(AtomRec) >= (1ll << 31))
    throw InputError("more than 2^25 atoms on one handle: the 32-bit record offsets of the sweep's index stream would wrap (shard the system)");
  h->d_nl_cnt.ensure(n + 1); h->d_dd_cnt.ensure(n + 1);
  h->d_nl_j.ensure((size_t)n * h->nl_pitch + 64);
  if (h->sweep_kernel < 3) h->d_dd_j.ensure((size_t)n * h->dd_pitch + 1024);  // slack: k_field_lp requests two index chunks per row up front
  else h->d_dd_j.ensure(64);
  // What the sweep streams per pair.  Measured (tools/exp_nocache.sh): while index + r^2 of all pairs
  // (12 B/pair) stay resident in the 256 MB Infinity Cache between sweeps the cached r^2 wins (36k atoms,
  // 160 MB: 99 vs 107 us/sweep); beyond that the stream comes from HBM every sweep and rebuilding r^2 from
  // the gathered positions wins (135k atoms, 595 MB: 320 vs 350 us/sweep).
  int mode = h->cache_r2;
  if (h->sweep_kernel == 1) mode = 0;
  if (h->sweep_kernel == 2) mode = 3;  // lane-per-pair sweep: only the index (as a byte offset) is streamed
  if (h->sweep_kernel >= 3) mode = 4;  // cluster sweep: the dd lists are the clusters' union lists (build_cluster_lists); tile sweep: build_tiles
  if (mode < 0) {
    const double est_pairs = h->dd_pairs > 0 ? (double)h->dd_pairs : 0.35 * (double)own_n(h) * (double)h->dd_pitch;
    mode = (12.0 * est_pairs < 200.0e6) ? 1 : 2;
  }
  h->stream_mode = mode;
  const bool r2c = mode == 1;
  if (mode == 1) h->d_dd_r2.ensure((size_t)n * h->dd_pitch + 64);
  else if (mode == 0) h->d_dd_s.ensure((size_t)n * h->dd_pitch + 64);
  double *r2p = r2c ? h->d_dd_r2.p : nullptr;
  double2 *sp = mode == 0 ? h->d_dd_s.p : nullptr;
  bool fuse = mode != 0 && mode != 4;  // r^2 and padding written by k_nl_build (no k_dd_scalars pass)
#ifdef POLAR_LAB
  if (getenv("POLAR_NO_FUSE_R2")) fuse = false;
#endif
  const double cutallsq = cutall * cutall, ddsq = mode == 4 ? -1.0 : st.dd_cutoff * st.dd_cutoff;
  const int nr = own_n(h);
  const int *rows = own_rows(h);
  zero_many(s, {{h->d_nl_cnt.p, (size_t)(n + 1) * sizeof(int)}, {h->d_dd_cnt.p, (size_t)(n + 1) * sizeof(int)},
                {h->d_overflow.p, 16 * sizeof(int)}, {h->d_ddtot.p, 64 * 16 * sizeof(unsigned long long)}});
  const bool lp = mode == 3;
  if (lp) h->d_dd_wrap.ensure(n + 1);
  const bool recheck = h->colors_valid && h->colors_recheck && h->sweep_kernel < 3 && (int)h->h_color.size() == n;
  if (h->colors_recheck && !recheck) { h->colors_valid = false; h->colors_recheck = false; }
  if (recheck) {
    h->d_color_s.ensure(n + 1);
    k_color_map<<<nblk(n, 256), 256, 0, s>>>(n, h->d_perm.p, h->d_color_orig.p, h->d_color_s.p);
  }
#define NLB(TRI, RC)                                                                                                          \
  k_nl_build<TRI, RC><<<nblk(nr, POLAR_ROWS_PER_BLOCK), POLAR_BLOCK, 0, s>>>(                                                 \
      rows, nr, h->d_pos4.p, h->box, g, h->d_cell_first.p, cutallsq, ddsq, h->nl_pitch, h->dd_pitch, h->d_nl_cnt.p,            \
      h->d_dd_cnt.p, h->d_nl_j.p, h->d_dd_j.p, fuse ? r2p : nullptr, fuse ? 1 : 0, lp ? (6 | (h->lp_quad_major << 8)) : 0,    \
      lp ? n : -1, lp ? h->d_dd_wrap.p : nullptr, recheck ? h->d_color_s.p : nullptr, h->color_keep * h->color_keep,           \
      h->d_overflow.p + 8, lp ? h->d_slot.p : nullptr, h->d_overflow.p, h->d_ddtot.p)
  // the box shape and "colours are being re-validated" are compile-time: the kernel is bound by its vector instructions
  if (h->box.triclinic) { if (recheck) NLB(true, true); else NLB(true, false); }
  else                  { if (recheck) NLB(false, true); else NLB(false, false); }
#undef NLB
  const RowList ddl{h->d_dd_cnt.p, h->dd_pitch};
  if (fuse || mode == 4) {
    // modes 1 and 2: the list build wrote r^2 (mode 1) and the padding itself; mode 4: no per-atom dd rows at all
  }
#ifdef POLAR_LAB
  else if (st.damping_type == POLAR_DAMP_EXPONENTIAL)
    k_dd_scalars<0><<<nblk(nr, POLAR_ROWS_PER_BLOCK), POLAR_BLOCK, 0, s>>>(rows, nr, h->d_rec0.p, h->box, ddl, h->d_dd_j.p, st.polar_damp, sp, r2p);
  else
    k_dd_scalars<1><<<nblk(nr, POLAR_ROWS_PER_BLOCK), POLAR_BLOCK, 0, s>>>(rows, nr, h->d_rec0.p, h->box, ddl, h->d_dd_j.p, st.polar_damp, sp, r2p);
#endif
  // overflow flag and dd total come back with the end-of-step read (no sync here)
  HIPCHECK(hipMemcpyAsync(h->h_flags, h->d_overflow.p, sizeof(int), hipMemcpyDeviceToHost, s));
  HIPCHECK(hipMemcpyAsync(h->h_ddtot, h->d_ddtot.p, 64 * 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
}

#ifdef POLAR_LAB
// ---- clusters of rows for k_field_cl and their colouring ------------------------------------------
// Clusters: greedy, in cell order -- a seed atom takes its nearest unassigned polarizable neighbours while every
// member stays within cluster_dist of every other (adjacency lists hold the atoms within color_dist, so
// cluster_dist <= color_dist).  Two clusters conflict when any two of their members are closer than color_dist;
// DSATUR colours the cluster graph, phases are ordered by the mean rank metric, members by descending rank metric
// (the in-cluster update is sequential: PS.cpp:1130-1143's "most coupled first", restricted to the cluster).
void build_cluster_colors(polar_handle *h, const std::vector<double> &rank, const std::vector<std::vector<int>> &cells,
                          const std::vector<std::vector<int>> &adj) {
  const int n = h->nlocal;
  const polar_settings &st = h->ph.st;
  const bool gs = st.polar_gs || st.polar_gs_ranked;
  const int lo = own_lo(h), hi = own_lo(h) + own_n(h);
  const double dcl = std::min(h->cluster_dist, h->color_dist), dcl2 = dcl * dcl;
  auto dist2 = [&](int i, int j) { return min_image_dist2(h->box, &h->hx[3 * (size_t)i], &h->hx[3 * (size_t)j]); };
  auto row_atom = [&](int i) { return i >= lo && i < hi && h->halpha[i] != 0.0; };
  std::vector<int> cl_of((size_t)n, -1);
  std::vector<int> mem;  // 4 per cluster
  for (const auto &cell : cells)
    for (int i : cell) {
      if (!row_atom(i) || cl_of[i] >= 0) continue;
      const int c = (int)(mem.size() / 4);
      int m[4] = {i, -1, -1, -1}, cnt = 1;
      cl_of[i] = c;
      if (h->cluster_max > 1) {
        std::vector<std::pair<double, int>> cand;
        for (int j : adj[i])
          if (row_atom(j) && cl_of[j] < 0) { const double d2 = dist2(i, j); if (d2 <= dcl2) cand.push_back({d2, j}); }
        std::sort(cand.begin(), cand.end());
        for (const auto &cj : cand) {
          if (cnt >= h->cluster_max) break;
          bool ok = true;
          for (int k = 1; k < cnt; k++) ok = ok && dist2(m[k], cj.second) <= dcl2;
          if (!ok) continue;
          m[cnt++] = cj.second;
          cl_of[cj.second] = c;
        }
      }
      if (!rank.empty()) std::stable_sort(m, m + cnt, [&](int a, int b) { return rank[a] > rank[b]; });
      mem.insert(mem.end(), m, m + 4);
    }
  const int ncl = (int)(mem.size() / 4);
  // cluster graph
  std::vector<std::vector<int>> cadj((size_t)ncl);
  if (gs)
    for (int c = 0; c < ncl; c++) {
      for (int k = 0; k < 4; k++) {
        const int a = mem[4 * (size_t)c + k];
        if (a < 0) continue;
        for (int b : adj[a]) { const int o = cl_of[b]; if (o >= 0 && o != c) cadj[c].push_back(o); }
      }
      std::sort(cadj[c].begin(), cadj[c].end());
      cadj[c].erase(std::unique(cadj[c].begin(), cadj[c].end()), cadj[c].end());
    }
  std::vector<int> color((size_t)ncl, gs ? -1 : 0), satur((size_t)ncl, 0);
  int ncolors = gs ? 0 : (ncl > 0 ? 1 : 0);
  if (gs) {
    std::vector<unsigned long long> seenmask((size_t)ncl, 0ull);
    struct Key { int sat, deg, idx; };
    auto lessk = [](const Key &a, const Key &b) {
      if (a.sat != b.sat) return a.sat < b.sat;
      if (a.deg != b.deg) return a.deg < b.deg;
      return a.idx > b.idx;
    };
    std::vector<Key> heap;
    heap.reserve((size_t)ncl * 2);
    for (int c = 0; c < ncl; c++) heap.push_back(Key{0, (int)cadj[c].size(), c});
    std::make_heap(heap.begin(), heap.end(), lessk);
    while (!heap.empty()) {
      std::pop_heap(heap.begin(), heap.end(), lessk);
      const Key kx = heap.back();
      heap.pop_back();
      const int c = kx.idx;
      if (color[c] >= 0 || kx.sat != satur[c]) continue;
      int col = 0;
      while (col < 64 && ((seenmask[c] >> col) & 1ull)) col++;
      if (col >= 64) throw std::runtime_error("colouring needs more than 64 colours: reduce POLAR_COLOR_DIST");
      color[c] = col;
      ncolors = std::max(ncolors, col + 1);
      for (int o : cadj[c]) {
        if (color[o] >= 0) continue;
        if (!((seenmask[o] >> col) & 1ull)) {
          seenmask[o] |= 1ull << col;
          satur[o]++;
          heap.push_back(Key{satur[o], (int)cadj[o].size(), o});
          std::push_heap(heap.begin(), heap.end(), lessk);
        }
      }
    }
    // balance the phases: DSATUR leaves classes as uneven as 3k / 9k / 6k / 25k clusters, and a phase with few
    // clusters cannot fill the GPU.  A cluster of the heaviest class moves to the lightest class none of its
    // neighbours uses, while that narrows the spread (weights = rows per cluster).
    if (getenv("POLAR_COLOR_BALANCE")) {  // measured: no gain (135k atoms: 246 vs 232 us per sweep), off by default
      std::vector<long long> wsum((size_t)ncolors, 0);
      std::vector<int> wcl((size_t)ncl, 0);
      for (int c = 0; c < ncl; c++) {
        for (int k = 0; k < 4; k++) wcl[c] += mem[4 * (size_t)c + k] >= 0;
        wsum[color[c]] += wcl[c];
      }
      for (int pass = 0; pass < 8; pass++) {
        long long moved = 0;
        for (int c = 0; c < ncl; c++) {
          const int from = color[c];
          unsigned long long used = 0ull;
          for (int o : cadj[c]) used |= 1ull << color[o];
          int best = -1;
          for (int k = 0; k < ncolors; k++)
            if (k != from && !((used >> k) & 1ull) && wsum[k] + wcl[c] < wsum[from] && (best < 0 || wsum[k] < wsum[best])) best = k;
          if (best >= 0) { wsum[from] -= wcl[c]; wsum[best] += wcl[c]; color[c] = best; moved++; }
        }
        if (!moved) break;
      }
    }
    // phase order: colours by descending mean rank metric (ranked flavour) or by descending size
    std::vector<double> key((size_t)ncolors, 0.0);
    std::vector<int> cnt((size_t)ncolors, 0), ord((size_t)ncolors), relabel((size_t)ncolors);
    for (int c = 0; c < ncl; c++)
      for (int k = 0; k < 4; k++) {
        const int a = mem[4 * (size_t)c + k];
        if (a < 0) continue;
        cnt[color[c]]++;
        key[color[c]] += rank.empty() ? 1.0 : rank[a];
      }
    if (!rank.empty())
      for (int c = 0; c < ncolors; c++) key[c] /= std::max(cnt[c], 1);
    std::iota(ord.begin(), ord.end(), 0);
    std::stable_sort(ord.begin(), ord.end(), [&](int a, int b) { return key[a] > key[b]; });
    for (int c = 0; c < ncolors; c++) relabel[ord[c]] = c;
    for (int c = 0; c < ncl; c++) color[c] = relabel[color[c]];
  }
  // clusters sorted by colour (cell order inside a colour, i.e. the order they were formed in)
  h->color_off.assign((size_t)ncolors + 1, 0);
  for (int c = 0; c < ncl; c++) h->color_off[color[c] + 1]++;
  for (int c = 0; c < ncolors; c++) h->color_off[c + 1] += h->color_off[c];
  std::vector<int> fill(h->color_off.begin(), h->color_off.end() - 1);
  h->h_cl.assign((size_t)ncl * 4, -1);
  long long natoms = 0;
  for (int c = 0; c < ncl; c++) {
    const int slot = fill[color[c]]++;
    for (int k = 0; k < 4; k++) { h->h_cl[4 * (size_t)slot + k] = mem[4 * (size_t)c + k]; natoms += mem[4 * (size_t)c + k] >= 0; }
  }
  h->ncl = ncl;
  h->d_cl_orig.ensure((size_t)ncl * 4 + 4);
  h->d_cl_s.ensure((size_t)ncl + 1);
  if (ncl > 0) HIPCHECK(hipMemcpy(h->d_cl_orig.p, h->h_cl.data(), (size_t)ncl * 4 * sizeof(int), hipMemcpyHostToDevice));
  if (getenv("POLAR_DEBUG")) {
    fprintf(stderr, "[polar] %d clusters of %lld rows (%.2f per cluster, dist %.2f), %d colour phases:", ncl, natoms,
            ncl ? (double)natoms / ncl : 0.0, dcl, ncolors);
    for (int c = 0; c < ncolors; c++) fprintf(stderr, " %d", h->color_off[c + 1] - h->color_off[c]);
    fprintf(stderr, "\n");
  }
  h->color_epoch++;
  h->colors_valid = true;
}

#else
inline void build_cluster_colors(polar_handle *, const std::vector<double> &, const std::vector<std::vector<int>> &, const std::vector<std::vector<int>> &) { throw std::logic_error("lab build only"); }
#endif  // POLAR_LAB

#ifdef POLAR_LAB
// ---- host-side greedy distance colouring for the colour-phase Gauss-Seidel (cutoff mode) ----
// Atoms of one colour are >= color_dist apart, so the couplings treated Jacobi-style inside a
// phase are weak and the splitting M = D + L_colour stays convergent for the SPD dipole system
// (DESIGN.md "colour-phase Gauss-Seidel").  Visit order = ranked order when polar_gs_ranked.
void build_colors(polar_handle *h, const std::vector<double> &rank) {
  const int n = h->nlocal;
  const double dc = h->color_dist, dcsq = dc * dc;
  int nc[3];
  long long ncell = 1;
  double width[3];
  box_widths(h->box, width);
  for (int k = 0; k < 3; k++) { nc[k] = std::max(1, (int)std::floor(width[k] / dc)); nc[k] = std::min(nc[k], 512); ncell *= nc[k]; }
  auto cellof = [&](int i, int c[3]) {
    double fr3[3];
    frac_coords(h->box, h->boxlo, h->hx[3 * (size_t)i], h->hx[3 * (size_t)i + 1], h->hx[3 * (size_t)i + 2], fr3);
    for (int k = 0; k < 3; k++) {
      double fr = fr3[k];
      fr -= std::floor(fr);
      c[k] = std::min(nc[k] - 1, (int)(fr * nc[k]));
    }
  };
  // 1. conflict graph: polarizable atoms closer than color_dist (cell grid of edge >= color_dist)
  std::vector<std::vector<int>> cells((size_t)ncell);
  for (int i = 0; i < n; i++) {
    if (h->halpha[i] == 0.0) continue;  // never updated: needs no phase
    int c[3];
    cellof(i, c);
    cells[((size_t)c[2] * nc[1] + c[1]) * nc[0] + c[0]].push_back(i);
  }
  std::vector<std::vector<int>> adj((size_t)n);
  for (int i = 0; i < n; i++) {
    if (h->halpha[i] == 0.0) continue;
    int c[3];
    cellof(i, c);
    int seen[27], nseen = 0;
    for (int dz = -1; dz <= 1; dz++)
      for (int dy = -1; dy <= 1; dy++)
        for (int dx = -1; dx <= 1; dx++) {
          int b[3] = {c[0] + dx, c[1] + dy, c[2] + dz};
          bool ok = true;
          for (int k = 0; k < 3; k++) {
            if (h->box.periodic[k]) b[k] = (b[k] + nc[k]) % nc[k];
            else if (b[k] < 0 || b[k] >= nc[k]) ok = false;
          }
          if (!ok) continue;
          const int cj = (int)(((size_t)b[2] * nc[1] + b[1]) * nc[0] + b[0]);
          bool dup = false;
          for (int t = 0; t < nseen; t++) dup |= seen[t] == cj;
          if (dup) continue;  // tiny grids: a cell reached through two offsets
          seen[nseen++] = cj;
          for (int j : cells[cj]) {
            if (j == i) continue;
            const double rsq = min_image_dist2(h->box, &h->hx[3 * (size_t)i], &h->hx[3 * (size_t)j]);
            if (rsq < dcsq) adj[i].push_back(j);
          }
        }
  }
  if (h->sweep_kernel == 3) { build_cluster_colors(h, rank, cells, adj); return; }
  // 2. DSATUR (Brelaz): always colour the vertex that sees the most distinct colours; ties by degree,
  //    then by index (deterministic: every rank of a multi-GPU run derives the same colouring).
  //    One colour fewer, and better balanced, than first-fit on the MOF test systems -> one launch
  //    fewer per sweep.  Lazy max-heap: stale entries are skipped when popped.
  std::vector<int> color((size_t)n, -1), satur((size_t)n, 0);
  std::vector<unsigned long long> seenmask((size_t)n, 0ull);  // colours 0..63 seen by the neighbours
  struct Key { int sat, deg, idx; };
  auto lessk = [](const Key &a, const Key &b) {
    if (a.sat != b.sat) return a.sat < b.sat;
    if (a.deg != b.deg) return a.deg < b.deg;
    return a.idx > b.idx;
  };
  std::vector<Key> heap;
  heap.reserve((size_t)n * 2);
  for (int i = 0; i < n; i++)
    if (h->halpha[i] != 0.0) heap.push_back(Key{0, (int)adj[i].size(), i});
  std::make_heap(heap.begin(), heap.end(), lessk);
  int ncolors = 0;
  while (!heap.empty()) {
    std::pop_heap(heap.begin(), heap.end(), lessk);
    const Key kx = heap.back();
    heap.pop_back();
    const int i = kx.idx;
    if (color[i] >= 0 || kx.sat != satur[i]) continue;  // already coloured, or a stale entry
    int col = 0;
    while (col < 64 && ((seenmask[i] >> col) & 1ull)) col++;
    if (col >= 64) throw std::runtime_error("colouring needs more than 64 colours: reduce POLAR_COLOR_DIST");
    color[i] = col;
    ncolors = std::max(ncolors, col + 1);
    for (int j : adj[i]) {
      if (color[j] >= 0) continue;
      if (!((seenmask[j] >> col) & 1ull)) {
        seenmask[j] |= 1ull << col;
        satur[j]++;
        heap.push_back(Key{satur[j], (int)adj[j].size(), j});
        std::push_heap(heap.begin(), heap.end(), lessk);
      }
    }
  }
  // 3. phase order: "ranked" flavour = colours by descending mean rank metric (PS.cpp:192-227 ranks the
  //    dipoles most likely to change first); otherwise by descending size.  Relabel accordingly.
  {
    std::vector<double> key((size_t)ncolors, 0.0);
    std::vector<int> cnt((size_t)ncolors, 0), ord((size_t)ncolors), relabel((size_t)ncolors);
    for (int i = 0; i < n; i++)
      if (color[i] >= 0) { cnt[color[i]]++; key[color[i]] += rank.empty() ? 1.0 : rank[i]; }
    if (!rank.empty())
      for (int c = 0; c < ncolors; c++) key[c] /= std::max(cnt[c], 1);
    std::iota(ord.begin(), ord.end(), 0);
    std::stable_sort(ord.begin(), ord.end(), [&](int a, int b) { return key[a] > key[b]; });
    for (int c = 0; c < ncolors; c++) relabel[ord[c]] = c;
    for (int i = 0; i < n; i++)
      if (color[i] >= 0) color[i] = relabel[color[i]];
  }
  std::vector<int> rows;
  rows.reserve(n);
  h->color_off.assign((size_t)ncolors + 1, 0);
  // the colouring is global (every rank computes the same one); a sharded handle keeps only its rows
  const int lo = own_lo(h), hi = own_lo(h) + own_n(h);
  auto mine = [&](int i) { return color[i] >= 0 && i >= lo && i < hi; };
  for (int i = 0; i < n; i++)
    if (mine(i)) h->color_off[color[i] + 1]++;
  for (int c = 0; c < ncolors; c++) h->color_off[c + 1] += h->color_off[c];
  rows.resize((size_t)h->color_off[ncolors]);
  std::vector<int> fill(h->color_off.begin(), h->color_off.end() - 1);
  // inside a colour, keep the rows in cell order of the colouring grid: neighbouring waves of a
  // phase then work on neighbouring atoms (shared records in L1/L2)
  for (auto &cell : cells)
    for (int i : cell)
      if (mine(i)) rows[fill[color[i]]++] = i;
  h->h_rows = rows;
  h->color_epoch++;  // the launch order changed: dd rows laid out for an earlier colouring are stale (slots_current)
  h->h_color.assign(color.begin(), color.end());
  h->d_color_orig.ensure((size_t)n + 1);
  if (n > 0) HIPCHECK(hipMemcpy(h->d_color_orig.p, h->h_color.data(), (size_t)n * sizeof(int), hipMemcpyHostToDevice));
  h->colors_rebuilt++;
  h->d_rows_orig.ensure(rows.size() + 1);
  h->d_rows.ensure(rows.size() + 1);
  if (!rows.empty()) HIPCHECK(hipMemcpy(h->d_rows_orig.p, rows.data(), rows.size() * sizeof(int), hipMemcpyHostToDevice));
  if (getenv("POLAR_DEBUG")) {
    fprintf(stderr, "[polar] %d colour phases (dist %.2f):", ncolors, h->color_dist);
    for (int c = 0; c < ncolors; c++) fprintf(stderr, " %d", h->color_off[c + 1] - h->color_off[c]);
    fprintf(stderr, "\n");
  }
  h->colors_valid = true;
}

#else
inline void build_colors(polar_handle *, const std::vector<double> &) { throw std::logic_error("host-side colouring: lab build only"); }
#endif  // POLAR_LAB

// ---- the colour phases on the device (polar_lists.hpp, k_color_*): sequential DSATUR cell by cell (parity classes of the
//      cell grid), the small top class repaired by local exhaustive search, Jones-Plassmann rounds as the fallback; phase
//      order and the rows of every phase in cell order.
//      Needs this step's cell order (phase_begin has run) and, for the ranked flavour, the rank metric in d_rank (s space).
void build_colors_device(polar_handle *h, bool ranked) {
  const int n = h->nlocal;
  hipStream_t s = h->stream;
  const bool dbg = getenv("POLAR_DEBUG") != nullptr;
  auto tprev = std::chrono::steady_clock::now();
  auto lap = [&](const char *what) {   // POLAR_DEBUG: wall time since the last lap, the device drained first
    if (!dbg) return;
    HIPCHECK(hipStreamSynchronize(s));
    const auto tn = std::chrono::steady_clock::now();
    fprintf(stderr, "[polar] colouring: %-28s %7.3f ms\n", what, std::chrono::duration<double, std::milli>(tn - tprev).count());
    tprev = tn;
  };
  lap("(work queued before)");
  const long long ncell = h->ncell;
  if (!h->h_cflags) {
    HIPCHECK(hipHostMalloc((void **)&h->h_cflags, 96 * sizeof(int)));
    HIPCHECK(hipHostMalloc((void **)&h->h_cstat, 128 * sizeof(double)));
    HIPCHECK(hipHostMalloc((void **)&h->h_coff, 72 * sizeof(long long)));
  }
  h->d_cdeg.ensure(n + 1); h->d_cprio.ensure(n + 1);
  h->d_color_s.ensure(n + 1); h->d_color_orig.ensure(n + 1); h->d_cflags.ensure(96); h->d_cstat.ensure(128); h->d_crelabel.ensure(64);
  int *flags = h->d_cflags.p;  // [0] conflict-list overflow, [1] atoms deferred in the last round, [2] a row that saw 64 colours, [65..] atoms that could not leave a folded class
  HIPCHECK(hipMemsetAsync(h->d_color_orig.p, 0xFF, (size_t)(n + 1) * sizeof(int), s));
  const int lo = own_lo(h), hi = own_lo(h) + own_n(h);
  const double dc2 = h->color_dist * h->color_dist;
  for (int attempt = 0;; attempt++) {  // conflict lists; an atom with more neighbours than the lists hold makes them wider
    h->d_cadj.ensure((size_t)n * h->cadj_pitch + 16);
    HIPCHECK(hipMemsetAsync(flags, 0, 96 * sizeof(int), s));
    k_color_adj<<<nblk(n, 128), 128, 0, s>>>(n, h->d_pos4.p, h->d_perm.p, lo, hi, h->box, h->grid, h->d_cell_first.p, h->d_cell_fill.p, dc2,
                                             h->cadj_pitch, h->d_cadj.p, h->d_cdeg.p, h->d_cprio.p, h->d_color_s.p, flags);
    HIPCHECK(hipMemcpyAsync(h->h_cflags, flags, sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHECK(hipStreamSynchronize(s));
    if (h->h_cflags[0] <= h->cadj_pitch) break;
    if (h->h_cflags[0] > 62 || attempt > 3) throw std::runtime_error("colouring: more than 62 polarizable atoms within the colour distance of one atom (64 colours at most): reduce POLAR_COLOR_DIST");
    h->cadj_pitch = (h->h_cflags[0] + 4 + 7) / 8 * 8;
  }
  const int ap_ = h->cadj_pitch;
  lap("conflict lists");
  // sequential DSATUR cell by cell (k_color_cells): one launch per parity class of the cell grid -- per dimension the even
  // cells, the odd cells and, when a periodic dimension has an odd count, its last cell on its own.  Safe while two cells of
  // a class (a whole cell apart) cannot hold neighbours: colour distance below the shortest cell edge.
  const double min_edge = std::min({h->box.prd[0] / h->grid.nc[0], h->box.prd[1] / h->grid.nc[1], h->box.prd[2] / h->grid.nc[2]});
  // (tilted box: a cell's perpendicular width is below its edge along the lattice vector -- by at most 1/sqrt(1.5) at LAMMPS'
  //  tilt limit of half a box length)
  bool cell_pass = h->color_dist < (h->box.triclinic ? 0.5 : 1.0) * min_edge;
#ifdef POLAR_LAB
  if (getenv("POLAR_COLOR_JP")) cell_pass = false;  // lab: Jones-Plassmann alone (the round-3 first version: 5 classes)
#endif
  if (cell_pass) {
    struct Cls { int start, stride, count; };
    std::vector<Cls> cls[3];
    for (int k = 0; k < 3; k++) {
      const int nc = h->grid.nc[k];
      const bool seam = h->box.periodic[k] && (nc & 1) && nc > 1;
      const int lim = seam ? nc - 1 : nc;
      if ((lim + 1) / 2 > 0) cls[k].push_back(Cls{0, 2, (lim + 1) / 2});
      if (lim / 2 > 0) cls[k].push_back(Cls{1, 2, lim / 2});
      if (seam) cls[k].push_back(Cls{nc - 1, 1, 1});
    }
    for (const Cls &cz : cls[2]) for (const Cls &cy : cls[1]) for (const Cls &cx : cls[0])
      k_color_cells<<<cx.count * cy.count * cz.count, 64, 0, s>>>(cx.start, cy.start, cz.start, cx.stride, cy.stride, cz.stride, cx.count, cy.count,
                                                                  cz.count, h->grid.nc[0], h->grid.nc[1], h->d_cell_first.p, h->d_cell_fill.p, ap_,
                                                                  h->d_cadj.p, h->d_cdeg.p, h->d_color_s.p, flags + 2);
  }
  // Jones-Plassmann rounds for whatever is still uncoloured (nothing after the cell pass; everything without it): an uncoloured
  // row whose priority beats every uncoloured neighbour's takes the lowest free colour.  First look after one round.
  bool coloured = false;
  int rounds = 0;
  const int max_rounds = 4096, look_every = 32;
  while (!coloured && rounds < max_rounds) {   // rounds (priorities, then decisions), then a look at how many atoms the last one deferred
    const int hashed = rounds >= 1024 ? 1 : 0;  // index-ordered ties while the chains stay short (see k_color_prio)
    const int batch = rounds == 0 ? 1 : look_every;
    for (int k = 0; k < batch; k++) {
      k_color_prio<<<nblk(n, 256), 256, 0, s>>>(n, ap_, h->d_cadj.p, h->d_cdeg.p, h->d_color_s.p, h->d_perm.p, h->d_cprio.p, hashed);
      if (k == batch - 1) HIPCHECK(hipMemsetAsync(flags + 1, 0, sizeof(int), s));
      k_color_round<<<nblk(n, 256), 256, 0, s>>>(n, ap_, h->d_cadj.p, h->d_cdeg.p, h->d_cprio.p, h->d_color_s.p, flags + 1);
    }
    rounds += batch;
    HIPCHECK(hipMemcpyAsync(h->h_cflags, flags, 3 * sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHECK(hipStreamSynchronize(s));
    if (h->h_cflags[2] >= 1000) throw std::runtime_error("colouring needs more than 64 colours: reduce POLAR_COLOR_DIST");
    coloured = h->h_cflags[1] == 0;
  }
  if (!coloured) throw std::runtime_error("colouring: Jones-Plassmann did not finish");
  lap("cell pass + rounds");
  if (getenv("POLAR_DEBUG")) fprintf(stderr, "[polar] device colouring: %d rounds\n", rounds);
  auto stats = [&]() {  // rows and rank sums per colour -> number of colours in use
    HIPCHECK(hipMemsetAsync(h->d_cstat.p, 0, 128 * sizeof(double), s));
    k_color_stats<<<nblk(n, 256), 256, 0, s>>>(n, h->d_color_s.p, ranked ? h->d_rank.p : nullptr, h->d_cstat.p);
    HIPCHECK(hipMemcpyAsync(h->h_cstat, h->d_cstat.p, 128 * sizeof(double), hipMemcpyDeviceToHost, s));
    HIPCHECK(hipStreamSynchronize(s));
    int nc = 0;
    for (int c = 0; c < 64; c++) if (h->h_cstat[2 * c] > 0.0) nc = c + 1;
    return nc;
  };
  int ncolors = stats();
  auto fold = [&]() {  // dissolve the highest class while that works
    for (int pass = 0; pass < 6 && ncolors > 1; pass++) {
      k_color_fold<<<nblk(n, 256), 256, 0, s>>>(n, ap_, ncolors - 1, h->d_cadj.p, h->d_cdeg.p, h->d_color_s.p, flags + 65 + pass);
      const int before = ncolors;
      ncolors = stats();
      if (ncolors == before) break;
    }
  };
  fold();
  lap("fold");
  // local repair of a small top class (k_color_ball): the cell-by-cell pass leaves about one row per unit cell in a fifth
  // class on the MOF boxes; first balls of one conflict step, then of two, then of three
  const int kcap = 8192;
  // (only a SMALL top class of at most six is worth it -- at most 2 % of the rows: where atoms overlap, e.g. sorbates flying through the
  //  framework in bench.py's ballistic leg, the extra classes are needed and no local search removes them)
  const double krows = std::min((double)kcap, 0.02 * (double)own_n(h));
  for (int hops = 1; hops <= 3 && ncolors > 2 && ncolors <= 6 && h->h_cstat[2 * (ncolors - 1)] <= (hops < 3 ? krows : 128.0); hops++) {   // (the last, long search only for a handful of rows)
    h->d_klist.ensure(4 * (size_t)kcap + 8);
    int *raw = h->d_klist.p, *list = raw + kcap, *st0 = raw + 2 * kcap, *st1 = raw + 3 * kcap, *cnt = raw + 4 * kcap;
    HIPCHECK(hipMemsetAsync(cnt, 0, sizeof(int), s));
    k_color_collect<<<nblk(n, 256), 256, 0, s>>>(n, ncolors - 1, h->d_color_s.p, kcap, raw, cnt);
    k_sort_small<<<8, 256, 0, s>>>(cnt, kcap, raw, list, st0, st1);
    const int waves = (int)h->h_cstat[2 * (ncolors - 1)];
    const double reach = (2 * hops + 1) * h->color_dist;
    const int budget = hops == 1 ? 768 : hops == 2 ? 2048 : 8192;   // search steps per ball: the later stages see few rows
    for (int round = 0; round < 2 + 2 * hops; round++)
      k_color_ball<<<waves, 64, 0, s>>>(cnt, kcap, list, (round & 1) ? st1 : st0, (round & 1) ? st0 : st1, h->d_pos4.p, h->box, reach * reach,
                                        ncolors - 1, hops, budget, ap_, h->d_cadj.p, h->d_cdeg.p, h->d_color_s.p);
    const int before = ncolors;
    ncolors = stats();
    if (getenv("POLAR_DEBUG")) fprintf(stderr, "[polar] colour repair (%d-step balls): %d rows in the top class, %d classes -> %d\n", hops, waves, before, ncolors);
    if (ncolors < before) break;
  }
  lap("repair");
  // iterated greedy (Culberson) for what is still above four classes -- overlapping atoms, or the Jones-Plassmann fallback,
  // whose parallel rounds decide on stale saturation counts: regrouping by old classes in another order never adds a class
  // and sometimes removes one
  for (int ig = 0, stale = 0; ig < 2 && stale < 2 && ncolors > 4; ig++) {
    std::vector<int> ord((size_t)ncolors), rank(64, 0);
    std::iota(ord.begin(), ord.end(), 0);
    if (ig % 2 == 0) std::reverse(ord.begin(), ord.end());                                    // highest class first
    else std::stable_sort(ord.begin(), ord.end(), [&](int a, int b) {                         // smallest / largest class first
      return (ig % 4 == 1) ? h->h_cstat[2 * a] < h->h_cstat[2 * b] : h->h_cstat[2 * a] > h->h_cstat[2 * b]; });
    for (int c = 0; c < ncolors; c++) rank[ord[c]] = c;
    HIPCHECK(hipMemcpyAsync(h->d_crelabel.p, rank.data(), 64 * sizeof(int), hipMemcpyHostToDevice, s));
    k_color_regroup<<<nblk(n, 256), 256, 0, s>>>(n, h->d_cdeg.p, h->d_crelabel.p, h->d_perm.p, h->d_color_s.p, h->d_cprio.p);
    for (int k = 0; k < ncolors + 1; k++)
      k_color_round<<<nblk(n, 256), 256, 0, s>>>(n, ap_, h->d_cadj.p, h->d_cdeg.p, h->d_cprio.p, h->d_color_s.p, flags + 80);
    const int before = ncolors;
    ncolors = stats();   // (synchronises: `rank` may go)
    fold();
    stale = ncolors < before ? 0 : stale + 1;
  }
  lap("iterated greedy");
  if (ncolors > 64) throw std::runtime_error("colouring needs more than 64 colours: reduce POLAR_COLOR_DIST");
  // phase order: "ranked" flavour = colours by descending mean rank metric (PS.cpp:192-227 ranks the dipoles most likely to
  // change first); otherwise by descending size
  std::vector<int> ord((size_t)ncolors), relabel(64, 0);
  std::iota(ord.begin(), ord.end(), 0);
  auto key = [&](int c) { return ranked ? h->h_cstat[2 * c + 1] / std::max(h->h_cstat[2 * c], 1.0) : h->h_cstat[2 * c]; };
  std::stable_sort(ord.begin(), ord.end(), [&](int a, int b) { return key(a) > key(b); });
  for (int c = 0; c < ncolors; c++) relabel[ord[c]] = c;
  HIPCHECK(hipMemcpyAsync(h->d_crelabel.p, relabel.data(), 64 * sizeof(int), hipMemcpyHostToDevice, s));
  k_color_relabel<<<nblk(n, 256), 256, 0, s>>>(n, h->d_crelabel.p, h->d_perm.p, h->d_color_s.p, h->d_color_orig.p);
  // rows of every phase in cell order
  const size_t ncc = (size_t)ncolors * ncell;
  h->d_ccnt.ensure(ncc + 1); h->d_coff.ensure(ncc + 2);
  k_color_cellcount<<<nblk(ncell, 128), 128, 0, s>>>(ncell, ncolors, h->d_cell_first.p, h->d_cell_fill.p, h->d_color_s.p, h->d_ccnt.p);
  k_exclusive_scan<int><<<1, 1024, 0, s>>>((long long)ncc, h->d_ccnt.p, h->d_coff.p);
  for (int c = 0; c <= ncolors; c++)
    HIPCHECK(hipMemcpyAsync(h->h_coff + c, h->d_coff.p + (size_t)c * ncell, sizeof(long long), hipMemcpyDeviceToHost, s));
  HIPCHECK(hipStreamSynchronize(s));  // (also: `relabel` is a stack vector)
  h->color_off.assign((size_t)ncolors + 1, 0);
  for (int c = 0; c <= ncolors; c++) h->color_off[c] = (int)h->h_coff[c];
  const int tot = h->color_off[ncolors];
  h->d_rows_orig.ensure((size_t)tot + 1); h->d_rows.ensure((size_t)tot + 1);
  k_color_fill<<<nblk(ncell, 128), 128, 0, s>>>(ncell, ncolors, h->d_cell_first.p, h->d_cell_fill.p, h->d_color_s.p, h->d_perm.p, h->d_coff.p,
                                               h->d_rows_orig.p);
  lap("phase order + rows");
#ifdef POLAR_LAB
  if (getenv("POLAR_LP_SORT_T") && tot > 0) {   // lab: inside a phase the rows with the most trips first (stable: cell order inside a trip count)
    std::vector<int> rows((size_t)tot), cnt((size_t)n), inv((size_t)n);
    HIPCHECK(hipMemcpyAsync(rows.data(), h->d_rows_orig.p, (size_t)tot * sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHECK(hipMemcpyAsync(cnt.data(), h->d_dd_cnt.p, (size_t)n * sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHECK(hipMemcpyAsync(inv.data(), h->d_inv.p, (size_t)n * sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHECK(hipStreamSynchronize(s));
    for (int c = 0; c < ncolors; c++)
      std::stable_sort(rows.begin() + h->color_off[c], rows.begin() + h->color_off[c + 1],
                       [&](int a, int b) { return (cnt[inv[a]] + 63) / 64 > (cnt[inv[b]] + 63) / 64; });
    HIPCHECK(hipMemcpyAsync(h->d_rows_orig.p, rows.data(), (size_t)tot * sizeof(int), hipMemcpyHostToDevice, s));
    HIPCHECK(hipStreamSynchronize(s));
  }
#endif
  h->h_color.assign((size_t)n, 0);  // (its size says "a colouring for n atoms exists": the colours themselves live on the device)
  h->color_epoch++;
  h->colors_rebuilt++;
  if (getenv("POLAR_DEBUG")) {
    fprintf(stderr, "[polar] %d colour phases (device, dist %.2f):", ncolors, h->color_dist);
    for (int c = 0; c < ncolors; c++) fprintf(stderr, " %d", h->color_off[c + 1] - h->color_off[c]);
    fprintf(stderr, "\n");
  }
  h->colors_valid = true;
}

template <bool AP>
void launch_rank(polar_handle *h, int pass) {
  const int n = h->nlocal, ntot = h->nlocal + h->nghost;
  dim3 grid(nblk(n, POLAR_ROWS_PER_BLOCK)), block(POLAR_BLOCK);
  if (pass == 2) k_fold_scal<<<1, POLAR_NSLOT, 0, h->stream>>>(h->d_scal.p, h->d_slots.p, 1);
  if (pass == 1)
    k_rank<AP, 1><<<grid, block, 0, h->stream>>>(n, ntot, h->d_x.p, h->d_alpha.p, h->d_mol.p, h->box, RowList{h->d_nl_cnt.p, h->nl_pitch},
                                                 h->d_nl_j.p, h->d_rec0.p, h->d_mol_s.p, h->d_scal.p, h->d_slots.p, h->d_rank.p);
  else
    k_rank<AP, 2><<<grid, block, 0, h->stream>>>(n, ntot, h->d_x.p, h->d_alpha.p, h->d_mol.p, h->box, RowList{h->d_nl_cnt.p, h->nl_pitch},
                                                 h->d_nl_j.p, h->d_rec0.p, h->d_mol_s.p, h->d_scal.p, h->d_slots.p, h->d_rank.p);
}

template <bool AP, int DAMP, int EP>
void launch_field(polar_handle *h, int nrows, const int *rows) {
  const polar_settings &st = h->ph.st;
  if (nrows <= 0) return;
  const int fb = h->field_block;  // threads per workgroup = 64 x rows that share one L1
  const size_t lds = AP ? 0 : (size_t)(fb / 64) * 64 * 5 * sizeof(double2);  // per-wave staging tiles (list mode)
  k_field<AP, DAMP, EP><<<nblk(nrows, fb / 64), fb, lds, h->stream>>>(
      nrows, rows, h->nlocal, h->d_rec0.p, h->d_rec1.p, h->box, RowList{h->d_dd_cnt.p, h->dd_pitch}, h->d_dd_j.p, h->d_dd_s.p,
      st.dd_cutoff * st.dd_cutoff, st.polar_damp, h->d_ef_s.p, h->d_F.p, h->d_scal.p, h->d_slots.p
#ifdef POLAR_LAB
      , h->ablate
#endif
      );
}
template <int EP>
void launch_field_dyn(polar_handle *h, bool ap, int nrows, const int *rows) {
  const bool expd = h->ph.st.damping_type == POLAR_DAMP_EXPONENTIAL;
  if (ap) { if (expd) launch_field<true, 0, EP>(h, nrows, rows); else launch_field<true, 1, EP>(h, nrows, rows); }
#ifdef POLAR_LAB
  else    { if (expd) launch_field<false, 0, EP>(h, nrows, rows); else launch_field<false, 1, EP>(h, nrows, rows); }
#else
  else throw std::logic_error("k_field's list branch exists in the lab build only");
#endif
}

#ifdef POLAR_LAB
// list-mode production sweep: component-per-lane quads (k_field_quad), one wave per row
template <int EP>
void launch_field_quad(polar_handle *h, int nrows, const int *rows) {
  if (nrows <= 0) return;
  const polar_settings &st = h->ph.st;
  const int qb = h->quad_block;
#define FQ(M) k_field_quad<EP, M><<<nblk_xcd(nrows, qb / 64), qb, 0, h->stream>>>(                 \
      nrows, rows, h->d_rec0.p, h->d_rec1.p, h->box, RowList{h->d_dd_cnt.p, h->dd_pitch}, h->d_dd_j.p, h->d_dd_s.p, \
      h->d_dd_r2.p, st.polar_damp, make_expcoef(), h->d_ef_s.p, h->d_scal.p, h->d_slots.p, h->ablate)
  const bool expd = st.damping_type == POLAR_DAMP_EXPONENTIAL;
  if (h->stream_mode == 0) FQ(0);
  else if (h->stream_mode == 2) { if (expd) FQ(3); else FQ(4); }
  else if (expd) FQ(1);
  else FQ(2);
#undef FQ
}

#else
template <int EP> inline void launch_field_quad(polar_handle *, int, const int *) { throw std::logic_error("lab build only"); }
#endif  // POLAR_LAB

// list-mode sweep, lane-per-pair with LDS-DMA gathers (k_field_lp), one wave per row
// row descriptors of this step for k_field_lp: the colour phases back to back (GS), or the own rows (Jacobi)
void prepare_lp(polar_handle *h) {
  const polar_settings &st = h->ph.st;
  const bool gs = st.polar_gs || st.polar_gs_ranked;
  const int tot = gs ? (h->color_off.empty() ? 0 : h->color_off.back()) : own_n(h);
  h->d_lpdesc.ensure((size_t)tot + 1);
  h->lp_npart = 0;
  if (deterministic(h)) {  // one partial sum per 256 rows of every launch; launches are 256-aligned in the array
    h->d_lp_pend.ensure(4 * (size_t)tot + 4);
    const int nlaunch = gs ? (int)h->color_off.size() - 1 : 1;
    h->lp_npart = tot / 256 + nlaunch + 1;
    h->d_lp_part.ensure((size_t)h->lp_npart + 1);
    HIPCHECK(hipMemsetAsync(h->d_lp_part.p, 0, (size_t)h->lp_npart * sizeof(double), h->stream));
  }
  if (tot > 0)
    k_lp_desc<<<nblk(tot, 256), 256, 0, h->stream>>>(tot, gs ? h->d_rows.p : own_rows(h), RowList{h->d_dd_cnt.p, h->dd_pitch},
                                                     h->d_dd_wrap.p, h->d_lpdesc.p);
}
template <int EP>
void launch_field_lp(polar_handle *h, int nrows, const int2 *desc) {
  const long long row0 = desc - h->d_lpdesc.p;  // launch row of the first descriptor = its dd row
  if (nrows <= 0) return;
  const polar_settings &st = h->ph.st;
  const int qb = h->quad_block;
  const int nt = h->lp_tiles;
  size_t lds = (size_t)(qb / 64) * nt * POLAR_LP_TILE;
  if (h->lp_wg_per_cu > 0) lds = std::max(lds, std::min((size_t)64 * 1024, (size_t)160 * 1024 / h->lp_wg_per_cu));  // lab: cap the residency
  // the launch requests dynamic LDS without raising the kernel's limit: beyond 64 KB (workgroups of more than 512 threads
  // with two tiles per wave) it would fail, and the failure would only surface at the next read of the loop state
  if (lds > (size_t)64 * 1024) throw InputError("k_field_lp: workgroup size x tiles needs more than 64 KB of LDS (POLAR_QUAD_BLOCK <= 512 with two tiles)");
  const bool det = deterministic(h);
  const double omega = EP == EP_INPLACE ? st.polar_sor : 1.0;
#ifdef POLAR_LAB
#define POLAR_LAB_ARG , h->ablate
#else
#define POLAR_LAB_ARG
#endif
#define FL(D, NT, DT) k_field_lp<EP, D, NT, DT><<<nblk_xcd(nrows, qb / 64), qb, lds, h->stream>>>(                   \
      nrows, row0, desc, h->d_rec0.p, h->d_rec1.p, h->box, h->dd_pitch, h->d_dd_j.p,                               \
      st.polar_damp, make_expcoef(), h->d_ef_s.p, h->d_scal.p, h->d_slots.p, omega, h->d_lp_pend.p POLAR_LAB_ARG)
  const bool expd = st.damping_type == POLAR_DAMP_EXPONENTIAL;
#ifdef POLAR_LAB
  if (h->lp_depth >= 2) {  // gathers kept lp_depth trips ahead (hand-counted waits), four tiles per wave
    const int pb = std::min(qb, 256);
    const size_t plds = (size_t)(pb / 64) * 4 * POLAR_LP_TILE;
#define FA(D, DEPTH) k_field_lpa<EP, D, DEPTH><<<nblk_xcd(nrows, pb / 64), pb, plds, h->stream>>>(                  \
      nrows, row0, desc, h->d_rec0.p, h->d_rec1.p, h->box, h->dd_pitch, h->d_dd_j.p,                                     \
      st.polar_damp, make_expcoef(), h->d_ef_s.p, h->d_scal.p, h->d_slots.p, h->ablate)
    if (h->lp_depth == 2) { if (expd) FA(0, 2); else FA(1, 2); }
    else                  { if (expd) FA(0, 3); else FA(1, 3); }
#undef FA
    return;
  }
  if (nt == 1 && !det) { if (expd) FL(0, 1, false); else FL(1, 1, false); return; }
#endif
#ifdef POLAR_LAB
  if (h->lp_rows > 1 && qb <= 256) {  // several launch rows per wave (k_field_lpr)
    const int R = h->lp_rows;
    const size_t rlds = (size_t)(qb / 64) * 2 * POLAR_LP_TILE;
#define FR(D, DT) k_field_lpr<EP, D, DT><<<nblk_xcd(nrows, (qb / 64) * R), qb, rlds, h->stream>>>(                         \
      nrows, row0, desc, h->d_rec0.p, h->d_rec1.p, h->box, h->dd_pitch, h->d_dd_j.p,                               \
      st.polar_damp, make_expcoef(), h->d_ef_s.p, h->d_scal.p, h->d_slots.p, omega, h->d_lp_pend.p, R)
    if (det) { if (expd) FR(0, true); else FR(1, true); }
    else     { if (expd) FR(0, false); else FR(1, false); }
#undef FR
    if (det) {
      int launch_no = 0;
      if (EP == EP_INPLACE) while (launch_no + 1 < (int)h->color_off.size() && h->color_off[launch_no] < row0) launch_no++;
      k_lp_commit<<<nblk(nrows, 256), 256, 0, h->stream>>>(nrows, row0, desc, h->d_lp_pend.p, h->d_rec0.p, h->d_rec1.p, EP == EP_JACOBI ? 1 : 0,
                                                        h->d_scal.p, h->d_lp_part.p + row0 / 256 + launch_no);
    }
    return;
  }
#endif
  if (det) {
    if (expd) FL(0, 2, true); else FL(1, 2, true);
    // the launch only read the record table: its rows' new dipoles and the sum of their changes are folded in now
    // (launch `l` of the sweep starts at row0: its partial sums start at slot row0 / 256 + l -- disjoint for every launch)
    int launch_no = 0;
    if (EP == EP_INPLACE) while (launch_no + 1 < (int)h->color_off.size() && h->color_off[launch_no] < row0) launch_no++;
    k_lp_commit<<<nblk(nrows, 256), 256, 0, h->stream>>>(nrows, row0, desc, h->d_lp_pend.p, h->d_rec0.p, h->d_rec1.p, EP == EP_JACOBI ? 1 : 0,
                                                      h->d_scal.p, h->d_lp_part.p + row0 / 256 + launch_no);
  } else {
    if (expd) FL(0, 2, false); else FL(1, 2, false);
  }
#undef FL
#undef POLAR_LAB_ARG
}

#ifdef POLAR_LAB
// cluster sweep (k_field_cl): one wave per cluster, clusters [first, first + ncl) of the colour-sorted table
template <int EP>
void launch_field_cl(polar_handle *h, int ncl, int first) {
  if (ncl <= 0) return;
  const polar_settings &st = h->ph.st;
  const int qb = std::min(h->quad_block, 256);
  const int nt = h->lp_tiles;
  const size_t lds = (size_t)(qb / 64) * nt * POLAR_LP_TILE;
#define FC(D, NT) k_field_cl<EP, D, NT><<<nblk_xcd(ncl, qb / 64), qb, lds, h->stream>>>(                              \
      ncl, first, h->d_cl_s.p, h->d_cl_tw.p, h->d_rec0.p, h->d_rec1.p, h->box, h->cl_pitch, h->d_dd_j.p,              \
      st.dd_cutoff * st.dd_cutoff, st.polar_damp, make_expcoef(), h->d_ef_s.p, h->d_scal.p, h->d_slots.p, h->ablate)
  const bool expd = st.damping_type == POLAR_DAMP_EXPONENTIAL;
  if (nt == 1) { if (expd) FC(0, 1); else FC(1, 1); }
  else         { if (expd) FC(0, 2); else FC(1, 2); }
#undef FC
}

// ---- tile sweep: per-step tables (polar_tiles.hpp) ---------------------------------------------------
struct TileUnavailable : std::runtime_error {
  explicit TileUnavailable(const std::string &m) : std::runtime_error(m) {}
};
inline size_t tile_lds_bytes(int records) { return POLAR_TILE_LDS_REC + (size_t)(records + 1) * sizeof(SRec) + POLAR_TILE_LDS_SLACK; }
inline int tile_lds_cap() { return (int)((160 * 1024 - POLAR_TILE_LDS_REC - POLAR_TILE_LDS_SLACK - 256) / sizeof(SRec)) - 1; }  // records one workgroup can stage at all
void build_tiles(polar_handle *h) {
  const polar_settings &st = h->ph.st;
  const int n = h->nlocal;
  hipStream_t s = h->stream;
  const double rc = st.dd_cutoff;
  if (h->un_pitch == 0) {
    long long npol = 0;
    for (int a = 0; a < n; a++) npol += h->halpha[a] != 0.0;
    const double dens = (h->dens > 0.0 ? h->dens : n / (h->box.prd[0] * h->box.prd[1] * h->box.prd[2])) * (double)npol / std::max(n, 1);
    double e[3];
    for (int k = 0; k < 3; k++) e[k] = h->box.prd[k] / h->grid.nc[k];
    // what a cell's rows can see: the cell widened by the cutoff (Minkowski sum of a box and a sphere)
    const double vol = e[0] * e[1] * e[2] + 2.0 * (e[0] * e[1] + e[1] * e[2] + e[0] * e[2]) * rc + M_PI * (e[0] + e[1] + e[2]) * rc * rc +
                       4.18879020478639 * rc * rc * rc;
    h->un_pitch = (int)(((long long)(1.25 * dens * vol) + 64 + 63) / 64 * 64);
    h->pitch16 = ((long long)(1.5 * dens * 4.18879020478639 * rc * rc * rc) + 64 + 511) / 512 * 512;
    if (const char *ip = getenv("POLAR_INIT_PITCH")) {  // tests: force the overflow paths
      h->un_pitch = std::max(64, atoi(ip) / 64 * 64);
      h->pitch16 = 512;
    }
    h->un_lds = 0;
  }
  if (h->un_pitch > tile_lds_cap()) throw TileUnavailable("tile sweep: a cell's neighbourhood does not fit the LDS of a compute unit");
  if (h->un_lds <= 0 || h->un_lds > h->un_pitch) h->un_lds = h->un_pitch;
  const long long ncell = h->ncell;
  h->d_thdr.ensure((size_t)ncell + 1); h->d_trow.ensure((size_t)n + 1);
  {  // the sweep requests entry words before it knows how many a tile holds: the table never contains stale garbage
    const int *before = h->d_un_j.p;
    h->d_un_j.ensure((size_t)ncell * h->un_pitch + 64);
    if (h->d_un_j.p != before) HIPCHECK(hipMemsetAsync(h->d_un_j.p, 0, h->d_un_j.cap * sizeof(int), s));
  }
  h->d_dd16.ensure((size_t)std::max(n, 1) * h->pitch16 + 1024);
  h->d_srec0.ensure((size_t)n + 1); h->d_srec1.ensure((size_t)n + 1);
  if (deterministic(h)) h->d_pend.ensure(3 * (size_t)n + 3);
  // (build_lists has just zeroed the flag words and the pair totals; its own k_nl_build lists no dd pair in this mode)
  const size_t lds = 24 * (size_t)h->un_pitch + (4 * 128 + 2 * POLAR_TILE_MAXROWS) * sizeof(int) + 24 * sizeof(double) + 12 * sizeof(int) + 16;
  if (lds > h->tile_build_lds_attr) {
    HIPCHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_tile_build), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    h->tile_build_lds_attr = lds;
  }
  const int cap = std::min(h->un_pitch, h->un_lds);
  k_tile_build<<<(int)ncell, 256, lds, s>>>(h->grid, h->box, h->d_pos4.p, h->d_cell_first.p, h->d_cell_fill.p, h->d_perm.p, own_lo(h),
                                            own_lo(h) + own_n(h), rc * rc, h->color_dist * h->color_dist, h->tile_waves, h->tile_sw[0], h->tile_sw[1], h->tile_sw[2], cap, h->d_un_j.p, h->pitch16,
                                            h->d_dd16.p, h->d_thdr.p, h->d_trow.p, h->d_rec0.p, h->d_overflow.p, h->d_ddtot.p);
  HIPCHECK(hipMemcpyAsync(h->h_flags + 5, h->d_overflow.p + 5, 5 * sizeof(int), hipMemcpyDeviceToHost, s));
  HIPCHECK(hipMemcpyAsync(h->h_ddtot, h->d_ddtot.p, 64 * 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
}
// NOTE: the builder indexes the union lists with the pitch it is given (`cap`), so the sweep must use the same value
inline int tile_pitch(const polar_handle *h) { return std::min(h->un_pitch, h->un_lds); }

template <int EP>
void launch_field_tile(polar_handle *h, const TileLaunch &L) {
  const polar_settings &st = h->ph.st;
  const long long nt = (long long)L.count[0] * L.count[1] * L.count[2];
  if (nt <= 0) return;
  const bool expd = st.damping_type == POLAR_DAMP_EXPONENTIAL;
  const bool det = deterministic(h) && EP == EP_INPLACE;
  size_t lds = tile_lds_bytes(tile_pitch(h));
  if (const char *e = getenv("POLAR_TILE_LDS_PAD")) lds += (size_t)atoi(e);  // LAB (temporary): residency experiment
  const int inst = ((EP == EP_JACOBI ? 0 : (det ? 2 : 1)) * 2 + (expd ? 0 : 1)) * 2 + (h->tile_waves == 8 ? 1 : 0);
#define FT(D, DT, W)                                                                                                           \
  {                                                                                                                            \
    if (lds > h->tile_lds_attr[inst]) {                                                                                        \
      HIPCHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_field_tile<EP, D, DT, W>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
      h->tile_lds_attr[inst] = lds;                                                                                            \
    }                                                                                                                          \
    k_field_tile<EP, D, DT, W><<<nblk_xcd(nt, 1), 64 * W, lds, h->stream>>>(L, h->d_thdr.p, h->d_trow.p, h->d_un_j.p, tile_pitch(h),   \
        h->d_dd16.p, h->pitch16, h->d_srec0.p, h->d_srec1.p, h->d_pend.p, h->d_ef_s.p, h->box, st.polar_damp,       \
        make_expcoef(), h->d_scal.p, h->d_slots.p);                                                                              \
  }
#define FW(D, DT) { if (h->tile_waves == 8) FT(D, DT, 8) else FT(D, DT, 4) }
  if (det) { if (expd) FW(0, true) else FW(1, true) }
  else     { if (expd) FW(0, false) else FW(1, false) }
#undef FW
#undef FT
  if (det) k_tile_commit<<<(int)nt, 64, 0, h->stream>>>(L, h->d_thdr.p, h->d_trow.p, h->d_pend.p, h->d_srec0.p, h->d_scal.p);
}

#else
template <int EP> inline void launch_field_cl(polar_handle *, int, int) { throw std::logic_error("lab build only"); }
struct TileUnavailable : std::runtime_error { explicit TileUnavailable(const std::string &m) : std::runtime_error(m) {} };
inline size_t tile_lds_bytes(int) { return 0; }
inline int tile_lds_cap() { return 0; }
inline void build_tiles(polar_handle *) { throw std::logic_error("lab build only"); }
template <int EP> inline void launch_field_tile(polar_handle *, const TileLaunch &) { throw std::logic_error("lab build only"); }
#endif  // POLAR_LAB

// one sweep over the rows this handle owns (Jacobi, or the colour phases)
#ifdef POLAR_LAB
// paired rows (lab): this step's units and their union lists; needs the colour rows in s space (map_color_rows) and the cells
void build_units(polar_handle *h) {
  const polar_settings &st = h->ph.st;
  hipStream_t s = h->stream;
  const bool gs = st.polar_gs || st.polar_gs_ranked;
  PhaseOff P;
  if (gs) {
    P.n = (int)h->color_off.size() - 1;
    if (P.n > 64) throw InputError("paired rows: more than 64 colour phases");
    for (int q = 0; q <= P.n; q++) P.off[q] = h->color_off[q];
  } else { P.n = 1; P.off[0] = 0; P.off[1] = own_n(h); }
  const int tot = P.off[P.n];
  h->unit_off.assign((size_t)P.n + 1, 0);
  if (tot <= 0) return;
  const int *rows = gs ? h->d_rows.p : own_rows(h);
  h->d_ulead.ensure((size_t)tot + 1); h->d_upos.ensure((size_t)tot + 2); h->d_unit.ensure((size_t)tot + 1); h->d_udesc.ensure((size_t)tot + 1);
  k_unit_flag<<<nblk(tot, 256), 256, 0, s>>>(tot, rows, h->d_perm.p, h->d_cell_id.p, P, h->d_ulead.p);
  k_exclusive_scan<int><<<1, 1024, 0, s>>>((long long)tot, h->d_ulead.p, h->d_upos.p);
  k_unit_fill<<<nblk(tot, 256), 256, 0, s>>>(tot, rows, h->d_perm.p, h->d_cell_id.p, P, h->d_ulead.p, h->d_upos.p, h->d_unit.p);
  std::vector<long long> first((size_t)P.n + 1);
  for (int q = 0; q <= P.n; q++)
    HIPCHECK(hipMemcpyAsync(&first[q], h->d_upos.p + P.off[q], sizeof(long long), hipMemcpyDeviceToHost, s));
  HIPCHECK(hipStreamSynchronize(s));
  for (int q = 0; q <= P.n; q++) h->unit_off[q] = (int)first[q];
  const int nunits = h->unit_off[P.n];
  if (h->upitch == 0) h->upitch = ((h->dd_pitch * 3 / 2 + 255) / 256) * 256;
  for (int attempt = 0;; attempt++) {
    h->d_udd_j.ensure((size_t)nunits * h->upitch + 1024);
    zero_many(s, {{h->d_overflow.p + 4, 4 * sizeof(int)}, {h->d_ddtot.p, 64 * 16 * sizeof(unsigned long long)}});
    if (h->box.triclinic)
      k_dd_units<true><<<nblk(nunits, POLAR_ROWS_PER_BLOCK), POLAR_BLOCK, 0, s>>>(nunits, h->d_unit.p, h->d_pos4.p, h->box, h->grid, h->d_cell_first.p, st.dd_cutoff * st.dd_cutoff,
                                                                                 h->upitch, h->d_udd_j.p, h->d_udesc.p, h->lp_quad_major ? 1 : 0, h->nlocal, h->d_overflow.p + 4, h->d_ddtot.p);
    else
      k_dd_units<false><<<nblk(nunits, POLAR_ROWS_PER_BLOCK), POLAR_BLOCK, 0, s>>>(nunits, h->d_unit.p, h->d_pos4.p, h->box, h->grid, h->d_cell_first.p, st.dd_cutoff * st.dd_cutoff,
                                                                                  h->upitch, h->d_udd_j.p, h->d_udesc.p, h->lp_quad_major ? 1 : 0, h->nlocal, h->d_overflow.p + 4, h->d_ddtot.p);
    int over = 0;
    HIPCHECK(hipMemcpyAsync(&over, h->d_overflow.p + 4, sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHECK(hipStreamSynchronize(s));
    if (over <= h->upitch) break;
    if (attempt > 2) throw std::runtime_error("paired rows: union list pitch overflow persists");
    h->upitch = (((long long)over * 9 / 8 + 255) / 256) * 256;
  }
  if (getenv("POLAR_DEBUG")) fprintf(stderr, "[polar] paired rows: %d units for %d rows, union pitch %lld\n", nunits, tot, h->upitch);
}
template <int EP>
void launch_field_lp2(polar_handle *h, int q) {
  const int nunits = h->unit_off[q + 1] - h->unit_off[q];
  if (nunits <= 0) return;
  const polar_settings &st = h->ph.st;
  const size_t lds = (size_t)4 * 2 * POLAR_LP_TILE;
  const double omega = EP == EP_INPLACE ? st.polar_sor : 1.0;
  if (st.damping_type == POLAR_DAMP_EXPONENTIAL)
    k_field_lp2<EP, 0><<<nblk_xcd(nunits, 4), 256, lds, h->stream>>>(nunits, (long long)h->unit_off[q], h->d_udesc.p, h->d_rec0.p, h->d_rec1.p, h->box, h->upitch,
                                                                     h->d_udd_j.p, st.polar_damp, make_expcoef(), h->d_ef_s.p, h->d_scal.p, h->d_slots.p, omega);
  else
    k_field_lp2<EP, 1><<<nblk_xcd(nunits, 4), 256, lds, h->stream>>>(nunits, (long long)h->unit_off[q], h->d_udesc.p, h->d_rec0.p, h->d_rec1.p, h->box, h->upitch,
                                                                     h->d_udd_j.p, st.polar_damp, make_expcoef(), h->d_ef_s.p, h->d_scal.p, h->d_slots.p, omega);
}
#endif

void sweep_once(polar_handle *h, bool ap) {
  const polar_settings &st = h->ph.st;
  const bool gs = st.polar_gs || st.polar_gs_ranked;
  if (!ap && h->sweep_kernel == 4) {
    if (!gs) { launch_field_tile<EP_JACOBI>(h, h->tile_all); return; }
    const int ncol = (int)h->tile_launches.size();
    const int c0 = h->part_n > 1 ? ncol * h->part_k / h->part_n : 0, c1 = h->part_n > 1 ? ncol * (h->part_k + 1) / h->part_n : ncol;
    for (int c = c0; c < c1; c++) launch_field_tile<EP_INPLACE>(h, h->tile_launches[c]);  // (polar_step_sweep_part: a window of the tile colours)
    return;
  }
  if (!ap && h->sweep_kernel == 3) {
    const int ncol = (int)h->color_off.size() - 1;
    for (int c = 0; c < ncol; c++) {
      if (gs) launch_field_cl<EP_INPLACE>(h, h->color_off[c + 1] - h->color_off[c], h->color_off[c]);
      else launch_field_cl<EP_JACOBI>(h, h->color_off[c + 1] - h->color_off[c], h->color_off[c]);
    }
    return;
  }
#ifdef POLAR_LAB
  if (!ap && h->sweep_kernel == 2 && h->lp_pairs && !deterministic(h) && h->part_n <= 1) {
    if (!gs) { launch_field_lp2<EP_JACOBI>(h, 0); return; }
    for (int c = 0; c + 1 < (int)h->unit_off.size(); c++) launch_field_lp2<EP_INPLACE>(h, c);
    return;
  }
#endif
  if (!ap && h->sweep_kernel == 2) {
    if (!gs) { launch_field_lp<EP_JACOBI>(h, own_n(h), h->d_lpdesc.p); return; }
    const int ncol = (int)h->color_off.size() - 1;
    const int c0 = h->part_n > 1 ? ncol * h->part_k / h->part_n : 0, c1 = h->part_n > 1 ? ncol * (h->part_k + 1) / h->part_n : ncol;
    for (int c = c0; c < c1; c++)  // (polar_step_sweep_part: a window of the colour phases)
      launch_field_lp<EP_INPLACE>(h, h->color_off[c + 1] - h->color_off[c], h->d_lpdesc.p + h->color_off[c]);
    return;
  }
  if (!ap && h->sweep_kernel == 0) {
    if (!gs) { launch_field_quad<EP_JACOBI>(h, own_n(h), own_rows(h)); return; }
    const int ncol = (int)h->color_off.size() - 1;
    for (int c = 0; c < ncol; c++)
      launch_field_quad<EP_INPLACE>(h, h->color_off[c + 1] - h->color_off[c], h->d_rows.p + h->color_off[c]);
    return;
  }
  if (!gs) {
    launch_field_dyn<EP_JACOBI>(h, ap, own_n(h), own_rows(h));
    return;
  }
  const int ncol = (int)h->color_off.size() - 1;
  for (int c = 0; c < ncol; c++) {
    const int cnt = h->color_off[c + 1] - h->color_off[c];
    if (cnt <= 0) continue;
    launch_field_dyn<EP_INPLACE>(h, false, cnt, h->d_rows.p + h->color_off[c]);
  }
}

template <bool AP, int DAMP>
void launch_force(polar_handle *h, int eflag, int vglobal, double *vatom, double *fdst) {
  const bool vpair = vglobal || vatom;
  const polar_settings &st = h->ph.st;
  dim3 grid(nblk(own_n(h), POLAR_ROWS_PER_BLOCK)), block(POLAR_BLOCK);
  const double ccs = st.cut_coul * st.cut_coul, dds = st.dd_cutoff * st.dd_cutoff, e2s = std::sqrt(h->P.qqrd2e);
  double *dbg6 = nullptr;   // `debug yes`: polarization force on the caller's atom 0 and its dipole-dipole part (PS.cpp:637-638)
  if (st.debug) {
    h->d_dbgf.ensure(8);
    HIPCHECK(hipMemsetAsync(h->d_dbgf.p, 0, 8 * sizeof(double), h->stream));
    dbg6 = h->d_dbgf.p;
  }
#define LF(E, V)                                                                                                    \
  k_polar_force<AP, DAMP, E, V><<<grid, block, 0, h->stream>>>(own_rows(h), own_n(h), h->sorted ? h->d_perm.p : nullptr, h->nlocal, h->d_scal.p, h->d_rec0.p, h->d_rec1.p,  \
                                                               h->d_mol_s.p, h->box, RowList{h->d_nl_cnt.p, h->nl_pitch}, h->d_nl_j.p,  \
                                                               ccs, dds, st.polar_damp, e2s, fdst, h->d_slots.p, vatom, vglobal, make_expcoef(),  \
                                                               dbg6)
  if (eflag) { if (vpair) LF(true, true); else LF(true, false); }
  else       { if (vpair) LF(false, true); else LF(false, false); }
#undef LF
}

void read_scal(polar_handle *h) {
  HIPCHECK(hipGetLastError());  // a failed kernel launch must not go unnoticed
  HIPCHECK(hipMemcpyAsync(h->h_scal, h->d_scal.p, sizeof(Scal), hipMemcpyDeviceToHost, h->stream));
  HIPCHECK(hipStreamSynchronize(h->stream));
}

// ---- the solve: a6+a7 (PS.cpp:1113-1238) -----------------------------------------------------
void ensure_colors(polar_handle *h) {
  if (h->colors_valid) return;
  const polar_settings &st = h->ph.st;
  const int n = h->nlocal;
  const auto t0 = std::chrono::steady_clock::now();
  const bool on_device = h->sorted && st.dd_cutoff > 0.0 && h->sweep_kernel == 2 && !h->host_colors && h->pol_first;
  if (on_device) {
    build_colors_device(h, st.polar_gs_ranked != 0);
  } else {  // lab paths (cluster rows, POLAR_HOST_COLORS): the host-side conflict graph + DSATUR of rounds 1-2
    std::vector<double> rk;
    if (st.polar_gs_ranked && !sharded(h)) {  // a sharded handle only knows its own rows' metric
      std::vector<double> rs(n);
      std::vector<int> perm(n);
      rk.assign(n, 0.0);
      HIPCHECK(hipMemcpyAsync(rs.data(), h->d_rank.p, n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
      HIPCHECK(hipMemcpyAsync(perm.data(), h->d_perm.p, n * sizeof(int), hipMemcpyDeviceToHost, h->stream));
      HIPCHECK(hipStreamSynchronize(h->stream));
      for (int k = 0; k < n; k++) rk[perm[k]] = rs[k];  // rank metric was computed in s space
    }
    build_colors(h, rk);
  }
  // (wall time of the rebuild, host work and the waits for the device included)
  h->ms_color_host = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}
// per step: the colour rows (orig ids) -> s space of this step's cell order
void map_color_rows(polar_handle *h) {
  const int tot = h->color_off.empty() ? 0 : h->color_off.back();
  if (tot > 0) k_map_rows<<<nblk(tot, 256), 256, 0, h->stream>>>(tot, h->d_inv.p, h->d_rows_orig.p, h->d_rows.p);
}

// reneighbor steps: k_nl_build has checked the colouring in use against the new positions; rebuild only on a clash
void resolve_colors(polar_handle *h) {
  if (!h->colors_recheck) return;
  h->colors_recheck = false;
  if (!h->colors_valid) return;
  int clash = 0;
  HIPCHECK(hipMemcpyAsync(&clash, h->d_overflow.p + 8, sizeof(int), hipMemcpyDeviceToHost, h->stream));
  HIPCHECK(hipStreamSynchronize(h->stream));
  if (!clash) { h->colors_reused++; return; }
  h->colors_valid = false;
  if (h->ph.st.polar_gs_ranked) { launch_rank<false>(h, 1); launch_rank<false>(h, 2); }  // a2 for the phase order
}

#ifdef POLAR_LAB
// cluster mode, per step: members -> s space, union lists, descriptors (needs the colours AND this step's cell order)
void build_cluster_lists(polar_handle *h) {
  const polar_settings &st = h->ph.st;
  hipStream_t s = h->stream;
  const int ncl = h->ncl;
  if (h->cl_pitch == 0) h->cl_pitch = ((h->dd_pitch * 5 / 4 + 255) / 256) * 256;
  h->d_cl_cnt.ensure(ncl + 1); h->d_cl_wrap.ensure(ncl + 1); h->d_cl_tw.ensure(ncl + 1);
  h->d_dd_j.ensure((size_t)std::max(ncl, 1) * h->cl_pitch + 256);
  if (ncl <= 0) return;
  k_map_rows_pad<<<nblk(4 * (long long)ncl, 256), 256, 0, s>>>(4 * ncl, h->d_inv.p, h->d_cl_orig.p, reinterpret_cast<int *>(h->d_cl_s.p));
  zero_many(s, {{h->d_overflow.p + 4, 4 * sizeof(int)}, {h->d_ddtot.p, 64 * 16 * sizeof(unsigned long long)}});
  k_cl_build<<<nblk(ncl, POLAR_ROWS_PER_BLOCK), POLAR_BLOCK, 0, s>>>(
      ClusterRows{h->d_cl_s.p, ncl}, h->d_pos4.p, h->box, h->grid, h->d_cell_first.p, st.dd_cutoff * st.dd_cutoff, h->cl_pitch,
      h->d_cl_cnt.p, h->d_dd_j.p, h->nlocal, h->d_cl_wrap.p, h->d_overflow.p + 4, h->d_ddtot.p);
  k_cl_desc<<<nblk(ncl, 256), 256, 0, s>>>(ncl, h->d_cl_cnt.p, h->cl_pitch, h->d_cl_wrap.p, h->d_cl_tw.p);
  HIPCHECK(hipMemcpyAsync(h->h_flags + 4, h->d_overflow.p + 4, sizeof(int), hipMemcpyDeviceToHost, s));
  HIPCHECK(hipMemcpyAsync(h->h_ddtot, h->d_ddtot.p, 64 * 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
}

#else
inline void build_cluster_lists(polar_handle *) { throw std::logic_error("lab build only"); }
#endif  // POLAR_LAB

// `debug yes` (PS.cpp:1182-1191): u_polar after sweep `sw`, kept on the device until polar_get_debug_trace
void debug_trace(polar_handle *h, int sw, bool /*jacobi*/) {
  if (!h->ph.st.debug) return;
  h->d_trace.ensure((size_t)h->ph.st.iterations_max + 8);
  if (sw > h->ph.st.iterations_max + 1) return;
  const MuView mv = mu_view(h);
  k_debug_upolar<<<1, 1024, 0, h->stream>>>(h->nlocal, h->d_scal.p, mv.a, mv.b, mv.stride, h->d_ef_s.p, h->d_trace.p, sw, 0);  // Jacobi: like the reference, the value is formed BEFORE "mu = mu_new" (jacobi unused)
  h->ntrace = sw + 1;
}

void solve(polar_handle *h, bool ap, polar_result *out) {
  const polar_settings &st = h->ph.st;
  const int n = h->nlocal;
  hipStream_t s = h->stream;
  const bool gs = st.polar_gs || st.polar_gs_ranked;
  const bool expd = st.damping_type == POLAR_DAMP_EXPONENTIAL;
  const int max_sweeps = st.iterations_max + 1;
  const int check_every = 4;
  out->ncolors = 0;

  if (!gs || !ap) {  // Jacobi (reference "polar_gs no") or colour-phase Gauss-Seidel over the dd list
    const bool clm = !ap && h->sweep_kernel == 3, tile = !ap && h->sweep_kernel == 4;  // tile sweep: no host-side colours at all
    if (!ap && !tile) resolve_colors(h);
    if (!tile && (gs || clm) && !h->colors_valid) { ensure_colors(h); if (clm) build_cluster_lists(h); else map_color_rows(h); }
    if (!ap && gs && h->sweep_kernel == 2 && !slots_current(h)) { compute_slots(h); build_lists(h); }  // rows into launch order
    if (gs) out->ncolors = tile ? (int)h->tile_launches.size() : (int)h->color_off.size() - 1;
    if (!ap && h->sweep_kernel == 2) prepare_lp(h);
#ifdef POLAR_LAB
    if (!ap && h->sweep_kernel == 2 && h->lp_pairs && !deterministic(h)) build_units(h);
#endif
    // fixed-iteration GS takes no decision between sweeps: its end-of-sweep logic is applied in two
    // launches (all sweeps but the last, then the last one, whose sum |dmu|^2 is the one reported)
    const bool lazy = st.fixed_iteration && gs;
    for (int sw = 0; sw < max_sweeps; sw++) {
      sweep_once(h, ap);
      debug_trace(h, sw, !gs);
      if (lazy && sw < max_sweeps - 2) continue;
      const int count = (lazy && sw == max_sweeps - 2) ? max_sweeps - 1 : 1;
      k_solver_step<<<1, POLAR_NSLOT, 0, s>>>(h->d_scal.p, h->d_slots.p, norm_count(h), st.fixed_iteration, st.iterations_max, st.polar_precision, gs ? 0 : 1, nullptr, count, det_part(h), det_npart(h));
      if (!st.fixed_iteration && (sw % check_every) == check_every - 1) {
        read_scal(h);
        if (h->h_scal->done) break;
      }
    }
  } else if (h->dense_gs) {  // exact-order blocked Gauss-Seidel on the HBM-resident tensor
    h->d_T6.ensure((size_t)n * n * 6 + 64); h->d_dmu.ensure(3 * 64);
    if (expd) k_build_T6<0><<<nblk(n, POLAR_ROWS_PER_BLOCK), POLAR_BLOCK, 0, s>>>(n, h->d_rec0.p, h->box, st.polar_damp, h->d_T6.p);
    else      k_build_T6<1><<<nblk(n, POLAR_ROWS_PER_BLOCK), POLAR_BLOCK, 0, s>>>(n, h->d_rec0.p, h->box, st.polar_damp, h->d_T6.p);
    k_dense_field<<<nblk(n, POLAR_ROWS_PER_BLOCK), POLAR_BLOCK, 0, s>>>(n, h->d_T6.p, h->d_rec0.p, h->d_F.p);
    for (int sw = 0; sw < max_sweeps; sw++) {
      for (int b0 = 0; b0 < n; b0 += 64) {
        k_gs_seq_T6<<<1, 64, 0, s>>>(n, b0, h->d_T6.p, h->d_rec0.p, h->d_ef_s.p, h->d_F.p, h->d_dmu.p, h->d_scal.p, h->d_slots.p);
        k_gs_push_T6<<<nblk(n, POLAR_ROWS_PER_BLOCK), POLAR_BLOCK, 0, s>>>(n, b0, h->d_T6.p, h->d_rec0.p, h->d_dmu.p, h->d_F.p, h->d_scal.p);
      }
      debug_trace(h, sw, false);
      k_solver_step<<<1, POLAR_NSLOT, 0, s>>>(h->d_scal.p, h->d_slots.p, norm_count(h), st.fixed_iteration, st.iterations_max, st.polar_precision, 0, nullptr, 1, nullptr, 0);
      if (!st.fixed_iteration && (sw % check_every) == check_every - 1) {
        read_scal(h);
        if (h->h_scal->done) break;
      }
    }
  } else {  // exact-order blocked Gauss-Seidel, matrix-free (systems whose tensor does not fit)
    std::vector<int> order(n), pos(n);
    std::iota(order.begin(), order.end(), 0);
    if (st.polar_gs_ranked) {  // stable descending sort == the reference's bubble sort (PS.cpp:1130-1143)
      std::vector<double> rk(n);
      HIPCHECK(hipMemcpyAsync(rk.data(), h->d_rank.p, n * sizeof(double), hipMemcpyDeviceToHost, s));
      HIPCHECK(hipStreamSynchronize(s));
      std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return rk[a] > rk[b]; });
    }
    for (int k = 0; k < n; k++) pos[order[k]] = k;
    h->d_order.ensure(n); h->d_pos.ensure(n); h->d_dmu.ensure(3 * 64);
    HIPCHECK(hipMemcpyAsync(h->d_order.p, order.data(), n * sizeof(int), hipMemcpyHostToDevice, s));
    HIPCHECK(hipMemcpyAsync(h->d_pos.p, pos.data(), n * sizeof(int), hipMemcpyHostToDevice, s));
    HIPCHECK(hipStreamSynchronize(s));  // order/pos are stack vectors
    launch_field_dyn<EP_FIELD>(h, true, n, nullptr);
    for (int sw = 0; sw < max_sweeps; sw++) {
      for (int b0 = 0; b0 < n; b0 += 64) {
        if (expd) {
          k_gs_block_seq<0><<<1, 64, 0, s>>>(n, b0, h->d_order.p, h->d_rec0.p, h->box, st.polar_damp, h->d_ef_s.p, h->d_F.p, h->d_dmu.p, h->d_scal.p, h->d_slots.p);
          k_gs_block_push<0><<<nblk(n, POLAR_ROWS_PER_BLOCK), POLAR_BLOCK, 0, s>>>(n, b0, h->d_order.p, h->d_pos.p, h->d_rec0.p, h->box, st.polar_damp, h->d_dmu.p, h->d_F.p, h->d_scal.p);
        } else {
          k_gs_block_seq<1><<<1, 64, 0, s>>>(n, b0, h->d_order.p, h->d_rec0.p, h->box, st.polar_damp, h->d_ef_s.p, h->d_F.p, h->d_dmu.p, h->d_scal.p, h->d_slots.p);
          k_gs_block_push<1><<<nblk(n, POLAR_ROWS_PER_BLOCK), POLAR_BLOCK, 0, s>>>(n, b0, h->d_order.p, h->d_pos.p, h->d_rec0.p, h->box, st.polar_damp, h->d_dmu.p, h->d_F.p, h->d_scal.p);
        }
      }
      debug_trace(h, sw, false);
      k_solver_step<<<1, POLAR_NSLOT, 0, s>>>(h->d_scal.p, h->d_slots.p, norm_count(h), st.fixed_iteration, st.iterations_max, st.polar_precision, 0, nullptr, 1, nullptr, 0);
      if (!st.fixed_iteration && (sw % check_every) == check_every - 1) {
        read_scal(h);
        if (h->h_scal->done) break;
      }
    }
  }
}

// PS.cpp:125-386: everything before the solve
void phase_begin(polar_handle *h, int eflag, int vflag, const double *mu_host) {
  need_device(h);
  if (!h->types_set || !h->coul_set) throw std::runtime_error("polar_compute before the pair tables were set: polar_pair_init (or polar_set_types), then polar_set_coul with the Coulomb tables of Pair::init_tables unless pair_modify table 0");
  if (!h->box_set || !h->atoms_set || !h->neigh_set) throw std::runtime_error("polar_compute before polar_set_box/polar_set_atoms/polar_set_neighbors");
  const polar_settings &st = h->ph.st;
  const int n = h->nlocal, nall = h->nlocal + h->nghost;
  const bool ap = !(st.dd_cutoff > 0.0);
  if (ap && own_n(h) != n) throw InputError("row sharding needs dd_cutoff > 0 (exact all-pairs mode runs as replicas only)");
  if (!ap && h->box.triclinic && !(h->sweep_kernel == 4 || (h->sweep_kernel == 2 && h->lp_depth == 0)))
    throw InputError("dd_cutoff (list) mode in a triclinic box needs the row sweep (k_field_lp) or, in the lab build, the tile sweep");
  const int vmode = vflag % 4;
  hipStream_t s = h->stream;
  h->warn.clear();
  h->ntrace = 0;
  h->step_eflag = eflag; h->step_vflag = vflag;

  h->d_f.ensure(3 * (size_t)nall); h->d_ef.ensure(3 * (size_t)n); h->d_F.ensure(3 * (size_t)n);
  h->d_mu.ensure(3 * (size_t)n); h->d_rank.ensure(n); h->d_rec0.ensure(n + 1); h->d_rec1.ensure(n + 1);  // + the dummy record
  h->d_ef_s.ensure(3 * (size_t)n); h->d_mol_s.ensure(n + 1);
  HIPCHECK(hipEventRecord(h->ev[0], s));
  HIPCHECK(hipMemsetAsync(h->d_f.p, 0, 3 * (size_t)nall * sizeof(double), s));
  // per-atom tallies (eflag/2, vflag/4: src/pair.cpp:760-764), zeroed like ev_setup does (:789-806)
  double *eatom = nullptr, *vatom = nullptr;
  if (eflag / 2) { h->d_eatom.ensure(nall + 1); eatom = h->d_eatom.p; HIPCHECK(hipMemsetAsync(eatom, 0, (size_t)nall * sizeof(double), s)); }
  if (vflag / 4) { h->d_vatom.ensure(6 * (size_t)nall + 6); vatom = h->d_vatom.p; HIPCHECK(hipMemsetAsync(vatom, 0, 6 * (size_t)nall * sizeof(double), s)); }
  // (on a row-sharded handle every kernel tallies into the rows it walks -- half of each pair per row atom -- so the
  //  shards' arrays add up to the unsharded ones: LJ/Coulomb over the rows of the list the shard was given, polarization
  //  over the own rows)
  k_zero_slots<<<nblk(POLAR_NSLOT, 256), 256, 0, s>>>(h->d_slots.p, h->d_scal.p);
  auto launch_lj = [&]() {  // a3 -- on a low-priority side stream when overlap is on: it depends on nothing below, and fills
    // whatever the list build, the static field and the latency-bound solver launches leave idle
    hipStream_t ms = s;
    h->lj_forked = h->overlap_lj && h->inum > 0;
    if (h->lj_forked) {
      HIPCHECK(hipEventRecord(h->ev_fork, ms));
      HIPCHECK(hipStreamWaitEvent(h->lj_stream, h->ev_fork, 0));
    }
    hipStream_t s = h->lj_forked ? h->lj_stream : ms;  // shadows the main stream inside this block
    HIPCHECK(hipEventRecord(h->ev_lj0, s));
    LJCoulParams P = h->P;
    P.newton_pair = h->newton_pair; P.nlocal = n; P.cut_coulsq = st.cut_coul * st.cut_coul; P.full_list = h->full_list; P.ablate = 0;
    h->d_xq.ensure(nall + 1);
    k_pack_lj<<<nblk(nall, 256), 256, 0, s>>>(nall, h->d_x.p, h->d_q.p, h->d_xq.p);
    const size_t ljlds = (size_t)(h->ntypes + 1) * (h->ntypes + 1) * 8 * sizeof(double);
    if (ljlds > 64 * 1024) throw InputError("too many atom types for the LDS-resident LJ table (max 31)");
    dim3 block(POLAR_BLOCK);
    bool symmetrise = !h->full_list;
#ifdef POLAR_LAB
    if (getenv("POLAR_LJ_ATOMICS")) symmetrise = false;  // round 1's FP64 atomics on f[j]
#endif
    // A full (newton-off) list puts no force on ghosts, so sum f.x over locals+ghosts (fdotr) would miss
    // the image terms: the LJ/Coulomb virial is then tallied pairwise (the same number), and only the
    // polarization forces -- local atoms, minimum-image displacements -- go through f.x (phase_finish)
    const bool lj_pairwise_virial = h->full_list && vmode == 2;
    if ((eatom || vatom) && !symmetrise && !h->full_list) throw InputError("per-atom tallies need the row-complete pair list");
    if (symmetrise && !h->sym_valid && h->inum > 0) {  // once per uploaded list (reneighbor steps)
      dim3 g0(nblk(h->inum, POLAR_ROWS_PER_BLOCK));
      h->d_sym_cnt.ensure(nall + 1); h->d_sym_fill.ensure(nall + 1); h->d_sym_first.ensure(nall + 2);
      h->d_sym_j.ensure(2 * (size_t)h->nneigh + 64);
      HIPCHECK(hipMemsetAsync(h->d_sym_cnt.p, 0, (nall + 1) * sizeof(int), s));
      HIPCHECK(hipMemsetAsync(h->d_sym_fill.p, 0, (nall + 1) * sizeof(int), s));
      k_sym_count<<<g0, block, 0, s>>>(h->inum, h->d_ilist.p, h->d_numneigh.p, h->d_first.p, h->d_neigh.p, h->d_sym_cnt.p);
      k_exclusive_scan<int><<<1, 1024, 0, s>>>(nall, h->d_sym_cnt.p, h->d_sym_first.p);
      h->sym_typed = h->lj_typed && nall < (1 << 24) && h->ntypes < 64;
      k_sym_fill<<<g0, block, 0, s>>>(h->inum, h->d_ilist.p, h->d_numneigh.p, h->d_first.p, h->d_neigh.p, h->d_sym_first.p,
                                      h->d_sym_fill.p, h->d_sym_j.p, h->sym_typed ? h->d_type.p : nullptr);
      h->sym_valid = true;
    }
    // newton off: ghosts receive no force and tally nothing (PS.cpp:293, ev_tally's 0.5 per LOCAL atom), so only the
    // local rows of the symmetrised list are walked: a local-ghost pair then counts 0.5, a local-local pair 0.5 + 0.5
    const int nrows_lj = symmetrise ? (h->newton_pair ? nall : n) : h->inum;
    dim3 grid(nblk(nrows_lj, POLAR_ROWS_PER_BLOCK));
    if (symmetrise) P.full_list = 1;  // rows of the symmetrised list: force on the row atom only, tallies halved
    P.typed_list = symmetrise ? (h->sym_typed ? 1 : 0) : (h->device_list && h->dev_typed ? 1 : 0);
    const int *il = symmetrise ? nullptr : h->d_ilist.p;
    const int *nn = symmetrise ? nullptr : h->d_numneigh.p;
    const long long *fi = symmetrise ? h->d_sym_first.p : h->d_first.p;
    const int *nj = symmetrise ? h->d_sym_j.p : h->d_neigh.p;
    if (h->inum > 0) {
#define LJ(E, V) k_ljcoul<E, V><<<grid, block, ljlds, s>>>(P, nrows_lj, il, nn, fi, nj, h->d_xq.p, h->d_type.p, h->d_f.p, h->d_slots.p, eatom, vatom, vmode == 1 || lj_pairwise_virial)
      const bool vrow = vmode == 1 || vatom || lj_pairwise_virial;
      if (eflag) { if (vrow) LJ(true, true); else LJ(true, false); }
      else       { if (vrow) LJ(false, true); else LJ(false, false); }
#undef LJ
    }
    HIPCHECK(hipEventRecord(h->ev_lj1, s));
    if (h->lj_forked) HIPCHECK(hipEventRecord(h->ev_join, s));
  };
  bool lj_late = false;
#ifdef POLAR_LAB
  lj_late = getenv("POLAR_LJ_LATE") != nullptr;  // fork a3 after the static field instead
#endif
  if (!lj_late) launch_lj();
  const double *mu0 = nullptr;
  if (st.use_previous) {
    // the caller's mu_induced (PS.cpp:376-386 reads atom->mu_induced).  Between two polar_set_atoms calls the atoms keep
    // their places and the caller's array is what the last compute call wrote into it: the resident copy is the same
    // numbers (a caller that edits mu_induced in between goes through polar_upload_mu or polar_set_atoms)
    if (mu_host && !(h->mu_resident && h->mu_host_in_sync)) {
      HIPCHECK(hipMemcpyAsync(h->d_mu.p, mu_host, 3 * (size_t)n * sizeof(double), hipMemcpyHostToDevice, s));
      mu0 = h->d_mu.p;
    } else if (h->mu_resident) mu0 = h->d_mu.p;
  }
  h->sorted = false;
  h->dense_gs = false;
  const bool gs_mode = (st.polar_gs || st.polar_gs_ranked) && !st.zodid;
  bool ranked_done = false;
  if (ap && gs_mode && n > 0 && 48.0 * (double)n * (double)n <= 4.0e9 && !getenv("POLAR_NO_DENSE_GS")) {
    // exact-order Gauss-Seidel on the HBM-resident tensor: put the atoms in SWEEP order first
    // (s space = ranked order), so blocks of the sweep are contiguous rows/columns of T6
    h->dense_gs = true;
    if (st.polar_gs_ranked) {
      launch_rank<true>(h, 1); launch_rank<true>(h, 2);  // a2 (orig space: needs only x/alpha/mol)
      ranked_done = true;
      std::vector<double> rk(n);
      std::vector<int> order(n), pos(n);
      HIPCHECK(hipMemcpyAsync(rk.data(), h->d_rank.p, n * sizeof(double), hipMemcpyDeviceToHost, s));
      HIPCHECK(hipStreamSynchronize(s));
      std::iota(order.begin(), order.end(), 0);
      // stable descending sort == the reference's bubble sort (PS.cpp:1130-1143)
      std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return rk[a] > rk[b]; });
      for (int k = 0; k < n; k++) pos[order[k]] = k;
      h->d_perm.ensure(n + 1); h->d_inv.ensure(n + 1);
      HIPCHECK(hipMemcpy(h->d_perm.p, order.data(), n * sizeof(int), hipMemcpyHostToDevice));
      HIPCHECK(hipMemcpy(h->d_inv.p, pos.data(), n * sizeof(int), hipMemcpyHostToDevice));
      h->sorted = true;
    }
  }
  if (!ap) { build_cells(h); h->d_pos4.ensure(n + 1); if (h->static_xq) h->d_xq_s.ensure(n + 2); }  // cell order: perm / inv
  k_pack<<<nblk(n + 1, 256), 256, 0, s>>>(n, h->sorted ? h->d_perm.p : nullptr, h->d_x.p, h->d_q.p, h->d_alpha.p, h->d_mol.p, mu0,
                                      h->d_rec0.p, h->d_rec1.p, h->d_mol_s.p, ap ? nullptr : h->d_pos4.p, (!ap && h->static_xq) ? h->d_xq_s.p : nullptr,
                                      (!ap && h->sweep_kernel == 4) ? 1 : 0, h->box, h->boxlo[0], h->boxlo[1], h->boxlo[2]);
  if (!ap) {
    if (h->colors_valid && h->sweep_kernel < 3) map_color_rows(h);  // the colour rows in this step's cell order
    if (h->sweep_kernel == 2) compute_slots(h);
    build_lists(h);
    if (h->colors_valid && h->sweep_kernel == 3) build_cluster_lists(h);
    if (h->sweep_kernel == 4) build_tiles(h);
  }
  HIPCHECK(hipEventRecord(h->ev[1], s));

  // a2: every step in exact mode (reference); in cutoff mode only when the colour phases are rebuilt
  if (st.polar_gs_ranked && (ap || (!h->colors_valid && h->sweep_kernel != 4)) && !ranked_done) {
    if (ap) { launch_rank<true>(h, 1); launch_rank<true>(h, 2); }
    else    { launch_rank<false>(h, 1); launch_rank<false>(h, 2); }
  }
  HIPCHECK(hipEventRecord(h->ev[2], s));

  HIPCHECK(hipEventRecord(h->ev[3], s));

  {  // a4 + a5
    dim3 grid(nblk(own_n(h), POLAR_ROWS_PER_BLOCK)), block(POLAR_BLOCK);
    const double ccs = st.cut_coul * st.cut_coul, e2s = std::sqrt(h->P.qqrd2e);
    if (ap) k_static_field<true><<<grid, block, 0, s>>>(nullptr, n, n, h->d_rec0.p, h->d_mol_s.p, h->box, RowList{nullptr, 0}, nullptr, ccs, e2s, st.polar_gamma, st.use_previous, h->d_ef_s.p, h->d_rec0.p, h->d_rec1.p, nullptr);
    else    k_static_field<false><<<grid, block, 0, s>>>(own_rows(h), own_n(h), n, h->d_rec0.p, h->d_mol_s.p, h->box, RowList{h->d_nl_cnt.p, h->nl_pitch}, h->d_nl_j.p, ccs, e2s, st.polar_gamma, st.use_previous, h->d_ef_s.p, h->d_rec0.p, h->d_rec1.p, h->static_xq ? h->d_xq_s.p : nullptr);
  }
  // tile sweep: the solve works on 48-byte sweep records {position, dipole}; the initial dipoles are in the AtomRecs now
#ifdef POLAR_LAB
  if (!ap && h->sweep_kernel == 4 && !st.zodid)
    k_srec_pack<<<nblk(n, 256), 256, 0, s>>>(n, h->d_rec0.p, h->d_srec0.p, h->d_srec1.p);
#endif
  HIPCHECK(hipEventRecord(h->ev[4], s));
  if (lj_late) launch_lj();
}

// PS.cpp:406-645: everything after the solve
int phase_finish(polar_handle *h, polar_result *out) {
  const polar_settings &st = h->ph.st;
  const int n = h->nlocal, nall = h->nlocal + h->nghost;
  const bool ap = !(st.dd_cutoff > 0.0);
  const bool expd = st.damping_type == POLAR_DAMP_EXPONENTIAL;
  const int eflag = h->step_eflag, vmode = h->step_vflag % 4;
  hipStream_t s = h->stream;
#ifdef POLAR_LAB
  if (!ap && h->sweep_kernel == 4 && !st.zodid)  // the solved dipoles back into the AtomRecs the remaining kernels read
    k_srec_unpack<<<nblk(n, 256), 256, 0, s>>>(n, h->d_scal.p, h->d_srec0.p, h->d_srec1.p, h->d_rec0.p, h->d_rec1.p);
#endif
  k_fallback<<<nblk(n, 256), 256, 0, s>>>(n, h->d_scal.p, h->d_rec0.p, h->d_rec1.p, h->d_ef_s.p);
  HIPCHECK(hipEventRecord(h->ev[5], s));
  // the dipoles and the static field are final: back into the caller's atom order now, so that polar_compute can send them
  // to the host while the force kernel runs (nothing below writes the records)
  k_unpack<<<nblk(n, 256), 256, 0, s>>>(n, h->sorted ? h->d_perm.p : nullptr, h->d_scal.p, h->d_rec0.p, h->d_rec1.p, h->d_ef_s.p, h->d_mu.p, h->d_ef.p);
  if (h->early_mu && n > 0) {
    HIPCHECK(hipEventRecord(h->ev_mu_ready, s));
    HIPCHECK(hipStreamWaitEvent(h->dl_stream, h->ev_mu_ready, 0));
    HIPCHECK(hipMemcpyAsync(h->early_mu, h->d_mu.p, 3 * (size_t)n * sizeof(double), hipMemcpyDeviceToHost, h->dl_stream));
    if (h->early_ef) HIPCHECK(hipMemcpyAsync(h->early_ef, h->d_ef.p, 3 * (size_t)n * sizeof(double), hipMemcpyDeviceToHost, h->dl_stream));
    HIPCHECK(hipEventRecord(h->ev_dl1, h->dl_stream));
  }
  double *vatom = (h->step_vflag / 4) ? h->d_vatom.p : nullptr;
  // join a3: before the force kernel when both write the (non-atomic) per-atom virial rows,
  // otherwise only before the accumulators are read
  auto join_lj = [&]() { if (h->lj_forked) { HIPCHECK(hipStreamWaitEvent(s, h->ev_join, 0)); h->lj_forked = false; } };
  if (vatom) join_lj();
  // full-list mode with an fdotr virial: the polarization forces go to their own array first, so that
  // sum f_pol . x can be formed without the LJ forces (whose virial was tallied pairwise)
  const bool split_f = h->full_list && vmode == 2;
  double *fdst = h->d_f.p;
  if (split_f) {
    h->d_fpol.ensure(3 * (size_t)n + 3);
    HIPCHECK(hipMemsetAsync(h->d_fpol.p, 0, 3 * (size_t)n * sizeof(double), s));
    fdst = h->d_fpol.p;
  }
  if (ap) { if (expd) launch_force<true, 0>(h, eflag, vmode == 1, vatom, fdst); else launch_force<true, 1>(h, eflag, vmode == 1, vatom, fdst); }
  else    { if (expd) launch_force<false, 0>(h, eflag, vmode == 1, vatom, fdst); else launch_force<false, 1>(h, eflag, vmode == 1, vatom, fdst); }
  join_lj();
  if (split_f) {
    k_virial_fdotr<<<std::min(1024, nblk(n, 256)), 256, 0, s>>>(n, h->d_x.p, h->d_fpol.p, h->d_slots.p);
    k_add_into<<<nblk(3 * (long long)n, 256), 256, 0, s>>>(3 * (long long)n, h->d_fpol.p, h->d_f.p);
  } else if (vmode == 2) k_virial_fdotr<<<std::min(1024, nblk(nall, 256)), 256, 0, s>>>(nall, h->d_x.p, h->d_f.p, h->d_slots.p);  // a10
  k_fold_scal<<<1, POLAR_NSLOT, 0, s>>>(h->d_scal.p, h->d_slots.p, 0);
  HIPCHECK(hipEventRecord(h->ev[6], s));
  if (h->early_mu && h->user_mu && n > 0) {  // the force kernel is still running: the dipoles and the field, on the host by now, go into the caller's arrays meanwhile
    HIPCHECK(hipEventSynchronize(h->ev_dl1));
    const double *sm = h->early_mu, *se = h->early_ef;
    double *um = h->user_mu, *ue = h->user_ef;
    host_chunks(3 * (size_t)n, [&](size_t a, size_t b) {
      memcpy(um + a, sm + a, (b - a) * sizeof(double));
      if (ue && se) memcpy(ue + a, se + a, (b - a) * sizeof(double));
    });
  }
  read_scal(h);
  h->mu_resident = true;

  const Scal &sc = *h->h_scal;
  out->eng_vdwl = sc.eng_vdwl; out->eng_coul = sc.eng_coul;
  out->u_self = sc.u_self; out->u_ef = sc.u_ef; out->u_dd = sc.u_dd;
  out->eng_pol = sc.u_self + sc.u_ef + sc.u_dd;  // PS.cpp:632 (all zero when eflag == 0)
  for (int k = 0; k < 6; k++) out->virial[k] = sc.virial[k];
  long long rb = (long long)sc.rmin_bits;
  memcpy(&out->rmin, &rb, sizeof(double));
  out->rms_dmu = std::sqrt(std::max(0.0, sc.last_change));
  h->ntrace = std::min(h->ntrace, sc.sweeps);
  out->iterations = sc.iterations; out->sweeps = sc.sweeps; out->status = sc.status ? POLAR_WARN_NOT_CONVERGED : POLAR_OK;
  if (!ap) {
    unsigned long long tot = 0;
    for (int k = 0; k < 64; k++) tot += h->h_ddtot[16 * k];
    h->dd_pairs = (long long)tot;
  }
  out->dd_pairs = ap ? (long long)n * (n - 1) : h->dd_pairs;
  out->ms_color_host = h->ms_color_host;
  h->ms_color_host = 0.0;
  float ms;
  auto el = [&](int a, int b) { HIPCHECK(hipEventElapsedTime(&ms, h->ev[a], h->ev[b])); return (double)ms; };
  out->ms_list = el(0, 1); out->ms_rank = el(1, 2); out->ms_static = el(3, 4);
  HIPCHECK(hipEventElapsedTime(&ms, h->ev_lj0, h->ev_lj1));
  out->ms_ljcoul = ms;  // on its own stream: overlaps the other phases, so the parts no longer add up to ms_total
  out->ms_solve = el(4, 5); out->ms_force = el(5, 6); out->ms_total = el(0, 6);
  if (sc.status) h->warn = "Number of iterations exceeding max_iterations, setting dipoles to alpha*E";  // PS.cpp:1233
  return out->status;
}

// end-of-step flags of the pitched lists (pinned, copied at the end of the list builds): true = a list did not fit, the
// pitches have been enlarged and the step must be repeated
void clear_flags(polar_handle *h) { for (int k = 0; k < 16; k++) h->h_flags[k] = 0; }
void tile_fallback(polar_handle *h) {
  h->sweep_kernel = 2;
  h->colors_valid = false; h->slots_by_color = false;
  h->nl_pitch = h->dd_pitch = 0;  // the cell grid changes with the sweep kernel
  h->warn = "tile sweep unavailable for this system (cell neighbourhood beyond the LDS, or a cell beyond the builder's limits): row sweep in use";
}
bool grow_pitches(polar_handle *h) {
  bool again = false;
  if (h->h_flags[0] != 0) {
    const long long need = ((long long)(1.25 * h->h_flags[0]) / 64 + 1) * 64;
    h->nl_pitch = h->dd_pitch = std::max(need, h->nl_pitch + 64);
    again = true;
  }
  if (h->h_flags[4] != 0) { h->cl_pitch = (((long long)(1.25 * h->h_flags[4]) + 255) / 256) * 256; again = true; }
  if (h->sweep_kernel == 4) {
    if (h->h_flags[7] != 0) { tile_fallback(h); return true; }
    if (h->h_flags[5] != 0) {  // a union list beyond the pitch, or beyond what the sweep's LDS request holds
      const int need = (int)(((long long)(1.15 * h->h_flags[5]) + 16 + 63) / 64 * 64);
      if (need > tile_lds_cap()) { tile_fallback(h); return true; }
      h->un_pitch = std::max(h->un_pitch, need); h->un_lds = h->un_pitch;
      again = true;
    }
    if (h->h_flags[6] != 0) { h->pitch16 = ((long long)(1.25 * h->h_flags[6]) + 511) / 512 * 512; again = true; }
    if (!again && h->h_flags[9] > 0) {  // next step: ask only for the LDS the unions need (two workgroups per CU below 80 KB)
      h->tile_max_u = h->h_flags[9];
      h->un_lds = std::min(h->un_pitch, (int)((long long)(1.06 * (h->tile_max_u + 1)) + 8 + 7) / 8 * 8);
      if (getenv("POLAR_DEBUG") && !h->tile_reported) {
        h->tile_reported = true;
        fprintf(stderr, "[polar] tile sweep: %lld cells (%d x %d x %d), %d launches per sweep, largest union %d records, union pitch %d, "
                "LDS request %zu bytes (%d records), row pitch %lld entries\n", h->ncell, h->grid.nc[0], h->grid.nc[1], h->grid.nc[2],
                (int)h->tile_launches.size(), h->tile_max_u, h->un_pitch, tile_lds_bytes(std::min(h->un_pitch, h->un_lds)), h->un_lds, h->pitch16);
      }
    }
  }
  return again;
}

int do_compute(polar_handle *h, int eflag, int vflag, const double *mu_host, polar_result *out) {
  memset(out, 0, sizeof(*out));
  const bool ap = !(h->ph.st.dd_cutoff > 0.0);
  int rc = 0;
  bool done = false;
  for (int attempt = 0; attempt < 5; attempt++) {
    clear_flags(h);
    try {
      phase_begin(h, eflag, vflag, mu_host);
    } catch (const TileUnavailable &) {  // density beyond what a workgroup can stage: the row sweep takes over
      tile_fallback(h);
      continue;
    }
    if (!h->ph.st.zodid) solve(h, ap, out);  // a6 + a7 (PS.cpp:389)
    const int nc = out->ncolors;
    rc = phase_finish(h, out);
    out->ncolors = nc;
    if (ap || !grow_pitches(h)) { done = true; break; }
    memset(out, 0, sizeof(*out));  // a row did not fit its pitch: the pitches have grown, redo the step
  }
  if (!done) throw std::runtime_error("neighbor list pitch overflow persists");
  return rc;
}

}  // namespace

// =============================================================================================
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
  if (const char *e = getenv("POLAR_ABLATE")) h->ablate = atoi(e);
  if (const char *e = getenv("POLAR_HOST_COLORS")) h->host_colors = atoi(e) != 0;
  if (const char *e = getenv("POLAR_SWEEP_KERNEL")) h->sweep_kernel = atoi(e);
  if (const char *e = getenv("POLAR_TILE_WAVES")) h->tile_waves = atoi(e) == 8 ? 8 : 4;
  if (const char *e = getenv("POLAR_TILE_WIDE")) h->tile_wide = atoi(e) != 0;
  if (const char *e = getenv("POLAR_CACHE_R2")) h->cache_r2 = atoi(e);
  if (const char *e = getenv("POLAR_LJ_TYPED")) h->lj_typed = atoi(e) != 0;
  if (const char *e = getenv("POLAR_STATIC_XQ")) h->static_xq = atoi(e) != 0;
  if (const char *e = getenv("POLAR_POL_FIRST")) h->pol_first = atoi(e) != 0;
  if (const char *e = getenv("POLAR_LP_WG_PER_CU")) h->lp_wg_per_cu = atoi(e);
  if (const char *e = getenv("POLAR_LP_QM")) h->lp_quad_major = atoi(e) != 0;
  if (const char *e = getenv("POLAR_LP_TILES")) h->lp_tiles = atoi(e) == 1 ? 1 : 2;
  if (const char *e = getenv("POLAR_LP_ROWS")) h->lp_rows = std::max(1, std::min(64, atoi(e)));
  if (const char *e = getenv("POLAR_LP_PAIRS")) h->lp_pairs = atoi(e) != 0;
  if (const char *e = getenv("POLAR_CLUSTER_DIST")) h->cluster_dist = atof(e);
  if (const char *e = getenv("POLAR_CLUSTER_MAX")) h->cluster_max = std::max(1, std::min(4, atoi(e)));
  if (const char *e = getenv("POLAR_LP_DEPTH")) { int v = atoi(e); h->lp_depth = (v == 2 || v == 3) ? v : 0; }
  if (const char *e = getenv("POLAR_QUAD_BLOCK")) { int v = atoi(e); if (v >= 64 && v <= 1024 && v % 64 == 0) h->quad_block = v; }
  if (const char *e = getenv("POLAR_FIELD_BLOCK")) { int v = atoi(e); if (v >= 64 && v <= 1024 && v % 64 == 0) h->field_block = v; }
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
    for (hipEvent_t e : h->ev_fchunk) if (e) (void)hipEventDestroy(e);
    h->d_ljpos.release(); h->d_ljaux.release(); h->d_tag.release(); h->d_nspecial.release(); h->d_special.release();
    h->d_ljcell_id.release(); h->d_ljcell_cnt.release(); h->d_ljcell_fill.release(); h->d_ljcell_first.release(); h->d_cutneighsq.release();
    h->d_xchg.release(); h->d_xidx.release();
    h->d_eatom.release(); h->d_vatom.release(); h->d_dd_r2.release(); h->d_fpol.release();
    h->d_x.release(); h->d_q.release(); h->d_alpha.release(); h->d_f.release(); h->d_ef.release(); h->d_F.release();
    h->d_mu.release(); h->d_rank.release(); h->d_dmu.release(); h->d_tab.release(); h->d_lj.release();
    h->d_type.release(); h->d_mol.release(); h->d_order.release(); h->d_pos.release(); h->d_ilist.release();
    h->d_numneigh.release(); h->d_neigh.release(); h->d_rows.release(); h->d_first.release();
    h->d_sym_first.release(); h->d_sym_cnt.release(); h->d_sym_fill.release(); h->d_sym_j.release();
    h->d_mol_s.release(); h->d_perm.release(); h->d_inv.release(); h->d_rows_orig.release(); h->d_ownrows.release(); h->d_ef_s.release(); h->d_T6.release();
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
const int kRestartVersion = 1;
const int kRestartPayload = 4 * (int)sizeof(double) + 10 * (int)sizeof(int);
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
  memcpy(p, iv, sizeof(iv));
  return total;
}
int polar_restart_unpack(polar_handle *h, const void *buf, int nbytes) {
  return guarded(h, [&]() {
    if (!buf || nbytes < POLAR_RESTART_HEADER_BYTES) throw InputError("not a polarization restart record");
    const char *p = (const char *)buf;
    int head[3];
    memcpy(head, p, sizeof(head)); p += sizeof(head);
    if (head[0] != kRestartMagic) throw InputError("not a polarization restart record");
    if (head[1] != kRestartVersion || head[2] != kRestartPayload || nbytes < POLAR_RESTART_HEADER_BYTES + head[2])
      throw InputError("polarization restart record of an unknown version or length");
    double d[4]; int iv[10];
    memcpy(d, p, sizeof(d)); p += sizeof(d);
    memcpy(iv, p, sizeof(iv));
    polar_settings s = h->ph.st;  // the cutoffs stay as the stock record set them
    s.polar_precision = d[0]; s.polar_damp = d[1]; s.polar_gamma = d[2]; s.dd_cutoff = d[3];
    s.iterations_max = iv[0]; s.damping_type = iv[1]; s.zodid = iv[2]; s.fixed_iteration = iv[3]; s.polar_gs = iv[4];
    s.polar_gs_ranked = iv[5]; s.use_previous = iv[6]; s.debug = iv[7]; s.device_neigh = iv[8]; s.restart_polar = iv[9];
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
    h->ph.st = *s;
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
    for (size_t k = 0; k < 3 * nall; k++)
      if (!std::isfinite(x[k])) throw InputError("non-finite atom coordinate");
    for (size_t k = 0; k < nall; k++)
      if (alpha[k] < 0.0) throw InputError("Invalid value in set command");  // src/set.cpp:174-184 rejects negatives
    if (nlocal != h->nlocal) { h->colors_valid = false; h->mu_resident = false; }
    else if (h->colors_valid) {  // the rows of the colour phases are the polarizable atoms: same atoms, same rows
      for (int k = 0; k < nlocal; k++)
        if ((alpha[k] != 0.0) != (h->halpha[k] != 0.0)) { h->colors_valid = false; break; }
    }
    h->nlocal = nlocal; h->nghost = nghost;
    h->d_x.ensure(3 * nall + 3); h->d_q.ensure(nall + 1); h->d_alpha.ensure(nall + 1); h->d_type.ensure(nall + 1); h->d_mol.ensure(nall + 1);
    hipStream_t s = h->stream;
    HIPCHECK(hipMemcpyAsync(h->d_x.p, x, 3 * nall * sizeof(double), hipMemcpyHostToDevice, s));
    HIPCHECK(hipMemcpyAsync(h->d_q.p, q, nall * sizeof(double), hipMemcpyHostToDevice, s));
    HIPCHECK(hipMemcpyAsync(h->d_alpha.p, alpha, nall * sizeof(double), hipMemcpyHostToDevice, s));
    HIPCHECK(hipMemcpyAsync(h->d_type.p, type, nall * sizeof(int), hipMemcpyHostToDevice, s));
    HIPCHECK(hipMemcpyAsync(h->d_mol.p, molecule, nall * sizeof(int), hipMemcpyHostToDevice, s));
    h->hx.assign(x, x + 3 * (size_t)nlocal);
    h->halpha.assign(alpha, alpha + nlocal);
    for (int k = 0; k < 3; k++) { h->bbox_lo[k] = 1e300; h->bbox_hi[k] = -1e300; }
    for (size_t a = 0; a < nall; a++)
      for (int k = 0; k < 3; k++) {
        h->bbox_lo[k] = std::min(h->bbox_lo[k], x[3 * a + k]);
        h->bbox_hi[k] = std::max(h->bbox_hi[k], x[3 * a + k]);
      }
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
    host_chunks(3 * nall, [&](size_t a, size_t b) { memcpy(st + a, x + a, (b - a) * sizeof(double)); });
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
    const int n = h->nlocal, nall = h->nlocal + h->nghost;
    long long total = 0;
    for (int ii = 0; ii < inum; ii++) {
      const int i = ilist[ii];
      if (i < 0 || i >= n) throw InputError("neighbor list row index out of range");
      if (numneigh[i] < 0 || firstneigh[i] < 0) throw InputError("negative neighbor count/offset");
      total = std::max(total, firstneigh[i] + numneigh[i]);
      for (int k = 0; k < numneigh[i]; k++) {
        const int j = neigh[firstneigh[i] + k] & 0x3FFFFFFF;
        if (j >= nall) throw InputError("neighbor index out of range");
      }
    }
    h->inum = inum; h->nneigh = total;
    h->d_ilist.ensure(inum + 1); h->d_numneigh.ensure(n + 1); h->d_first.ensure(n + 1); h->d_neigh.ensure((size_t)total + 1);
    hipStream_t s = h->stream;
    HIPCHECK(hipMemcpyAsync(h->d_ilist.p, ilist, inum * sizeof(int), hipMemcpyHostToDevice, s));
    HIPCHECK(hipMemcpyAsync(h->d_numneigh.p, numneigh, n * sizeof(int), hipMemcpyHostToDevice, s));
    HIPCHECK(hipMemcpyAsync(h->d_first.p, firstneigh, n * sizeof(long long), hipMemcpyHostToDevice, s));
    HIPCHECK(hipMemcpyAsync(h->d_neigh.p, neigh, (size_t)total * sizeof(int), hipMemcpyHostToDevice, s));
    HIPCHECK(hipStreamSynchronize(s));
    h->neigh_set = true;
    h->sym_valid = false;
    if (h->colors_valid) h->colors_recheck = true;  // reneighbor step: the colour phases are re-validated (k_nl_build)
    h->device_list = false;
    h->full_list = h->user_full_list;
    return POLAR_OK;
  });
}

int polar_build_neighbors(polar_handle *h, const double *cutneighsq, const int *tag, const int *nspecial,
                          const int *special, int maxspecial, const int special_flag[4], int exclude_molecule_intra) {
  return guarded(h, [&]() {
    need_device(h);
    HIPCHECK(hipSetDevice(h->device));
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
    k_exclusive_scan<int><<<1, 1024, 0, s>>>(ncell, h->d_ljcell_cnt.p, h->d_ljcell_first.p);
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
  if (!h) return POLAR_ERR_STATE;
  // flatten LAMMPS' paged int** rows into one CSR buffer
  std::vector<long long> first((size_t)std::max(h->nlocal, 1), 0);
  long long total = 0;
  for (int ii = 0; ii < inum; ii++) {
    const int i = ilist[ii];
    if (i < 0 || i >= h->nlocal) return fail(h, POLAR_ERR_INPUT, "neighbor list row index out of range");
    first[i] = total;
    total += numneigh[i];
  }
  std::vector<int> flat((size_t)total + 1);
  std::vector<int> nn((size_t)std::max(h->nlocal, 1), 0);
  for (int ii = 0; ii < inum; ii++) {
    const int i = ilist[ii];
    nn[i] = numneigh[i];
    if (numneigh[i] > 0) memcpy(flat.data() + first[i], firstneigh[i], (size_t)numneigh[i] * sizeof(int));
  }
  return polar_set_neighbors_csr(h, inum, ilist, nn.data(), first.data(), flat.data());
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
    HIPCHECK(hipSetDevice(h->device));
    clear_flags(h);
    try {
      phase_begin(h, eflag, vflag, nullptr);
    } catch (const TileUnavailable &) {
      tile_fallback(h);
      phase_begin(h, eflag, vflag, nullptr);
    }
    const polar_settings &st = h->ph.st;
    const bool clm = st.dd_cutoff > 0.0 && h->sweep_kernel == 3, tile = tile_mode(h);
    if (st.dd_cutoff > 0.0 && !tile) resolve_colors(h);
    if (!tile && !st.zodid && (st.polar_gs || st.polar_gs_ranked || clm) && !h->colors_valid) { ensure_colors(h); if (clm) build_cluster_lists(h); else map_color_rows(h); }
    if (!st.zodid && st.dd_cutoff > 0.0 && (st.polar_gs || st.polar_gs_ranked) && h->sweep_kernel == 2 && !slots_current(h)) { compute_slots(h); build_lists(h); }
    if (!st.zodid && st.dd_cutoff > 0.0 && h->sweep_kernel == 2) prepare_lp(h);
    h->in_step = true;
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
