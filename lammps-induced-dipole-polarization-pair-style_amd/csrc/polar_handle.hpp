// polar_handle.hpp -- the handle (HBM-resident state of one Pair instance), the small host utilities every translation
// unit of the library shares, and the functions they call across each other.  Translation units:
//   polar_api.hip    the C-ABI entry points (include/polar_mi355x.h): lifetime, text interface, per-step data, compute calls,
//                    the stepwise / sharded interface
//   polar_step.hip   one Pair::compute on the device (PS.cpp:125-645): cells and lists, LJ + Ewald-real, static field, the
//                    solve (PS.cpp:1113-1238), forces; the sweep launchers
//   polar_color.hip  colour phases of the list-mode Gauss-Seidel (device colouring, re-validation)
//   polar_dist.hip   multi-GPU driver over RCCL (polar_dist_*)
// No CPU implementation of the hot path anywhere: without a GPU every compute call fails.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <initializer_list>
#include <utility>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <exception>
#include <cstring>
#include <numeric>
#include <string>
#include <thread>
#include <mutex>
#include <condition_variable>
#include <functional>
#include <memory>
#include <stdexcept>
#include <unistd.h>
#include <vector>

#include "pair_host.hpp"
#include "polar_kernels.hpp"

using namespace polar;

// bump when a kernel on the hot path changes: PMC files under profiles/ are keyed by it (bench.py, roofline.traffic)
#define POLAR_KERNEL_VERSION "r05-lp3-v3"
#define POLAR_MAX_CLASS_OFF 130   // phase classes of a colouring + 1: up to 64 colours, each split into boundary / interior rows on a sharded handle


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
    // POLAR_POISON=1 (debugging aid): a fresh process gets zero pages from hipMalloc, a long-lived one gets recycled memory --
    // fill every new buffer with 0x7F bytes (ints 2139062143, doubles 1.4e306) so that a read-before-write shows in the FIRST
    // call of a process instead of the ninetieth (the abort of round 4)
    static const bool poison = getenv("POLAR_POISON") && atoi(getenv("POLAR_POISON")) != 0;
    if (poison) {   // (the fill runs on the null stream, which the library's non-blocking streams do not wait for: finish it here)
      HIPCHECK(hipMemset(p, 0x7F, want * sizeof(T)));
      HIPCHECK(hipDeviceSynchronize());
    }
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
// (exclusive scans: launch_scan in polar_step.hip)
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
  DBuf<double> d_ef_s, d_T6, d_Minv, d_gsN, d_gsAT, d_cb, d_gs_part, d_eatom, d_vatom, d_dd_r2, d_fpol;
  bool dense_gs = false;   // exact-order GS on the HBM-resident tensor (atoms in sweep order)
  CellGrid grid{};
  long long ncell = 0;
  bool sorted = false;  // true while the records are in cell order (list mode)
  DBuf<long long> d_first, d_sym_first;
  DBuf<int> d_gs_cnt;   // a zeroed counter: exact mode's end-of-sweep logic on a sweep's last launch (GsTail)
  DBuf<long long> d_scan_a, d_scan_b;   // workgroup totals of launch_scan: main stream / the LJ-Coulomb side stream
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
  DBuf<double> d_acc_x, d_acc_f, d_acc_g, d_acc_dF, d_acc_dG, d_acc_part;   // `polar_accel`: iterate, last residual, last G(x), difference histories, partial dots
  DBuf<AccelState> d_acc_state;
  long long acc_pitch = 0; int acc_rows = 0;
  DBuf<double> d_mu0;            // use_previous: the initial guess of the running step (a retry after a pitch overflow starts from it again)
  bool mu0_saved = false;
  int attempt = 0;               // which attempt of the running step (do_compute)
  int last_sweeps = 0;           // sweeps of the last converged precision-mode solve of this handle (0: none yet): where the host first looks at the loop state (look_at_state)
  bool mu_host_in_sync = false;  // the caller's mu array still holds what the last polar_compute returned (no polar_set_atoms since)
  // colour phases (cutoff-mode Gauss-Seidel)
  std::vector<int> color_off;  // [ncolors+1] offsets into d_rows
  std::vector<int> color_sub;  // [ncolors * color_nsub + 1] offsets of the sub-classes inside the phases (class = colour * nsub + sub)
  int color_nsub = 1;          // 1: none; 2 (multi-GPU): boundary rows, then interior rows
  DBuf<int> d_bflag; int bflag_n = 0;   // multi-GPU: sub-class by original index (polar_dist_set_halo: 0 = a peer receives this row's dipole, 1 = not), bflag_n = atoms it was made for
  // the neighbor list's upload: its own stream, one pinned buffer (+ event) per 32-MB chunk, host copies of the row tables that
  // must outlive the call, the event its consumers wait for
  std::vector<int *> h_nl_stage; std::vector<hipEvent_t> ev_nl;
  hipStream_t up_stream = nullptr; hipEvent_t ev_list_up = nullptr; bool list_up_pending = false;
  std::vector<long long> h_first; std::vector<int> h_nn, h_ilist;
  bool colors_global = false;  // the colouring in force is consistent across the ranks of a multi-GPU run (halo rows carry their owners' colours)
  std::vector<int> user_colors;  // polar_set_colors: a colouring imposed by the caller (original order, -1 = none)
  bool user_colors_clashed = false;
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
  long long *h_coff = nullptr;    // pinned: first row of every phase class (POLAR_MAX_CLASS_OFF entries: 64 colours x 2 sub-classes + 1)
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
  bool lj_tab_arith = false;     // the Coulomb tables are regular: a bin's r and dr can be rebuilt from the bits of (float)rsq (upload_coul)
  int lj_pers = 1;               // a3 as the persistent kernel with the Coulomb bins in LDS where they fit (POLAR_LJ_PERS=0: one wave per row, bins from memory)
  int ncu = 0;                   // compute units of the device
  int lj_pers_threads = 768;     // threads of a3's persistent workgroup (POLAR_LJ_PERS_THREADS: 256 ... 1024).  Sixteen waves of 128 registers take a CU's whole register file: nothing
                                 // runs beside them; twelve leave a quarter of it to the list build.  Measured on one box (gpurun_out/r5l_*): 1024: 9.01, 768: 8.88, 512: 8.97, 384: 9.12, 256: 9.67 ms per step
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
inline int fail(polar_handle *h, int code, const std::string &m) {
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
inline double *staging(polar_handle *h, size_t count) {
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

inline void need_device(polar_handle *h) {
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
  // An exception in any part (helper or caller) is rethrown here, AFTER every part has finished: the helpers run `fn` through a
  // pointer into the caller's frame and must never outlive it, and an exception that escaped a helper thread would be
  // std::terminate -- an abort with no message under a test runner.
  void run(const std::function<void(int)> &fn, int parts) {
    if (parts <= 1 || th_.empty() || getpid() != pid_) { for (int k = 0; k < parts; k++) fn(k); return; }   // (a forked child has no helpers)
    std::unique_lock<std::mutex> job(job_m_);   // one job at a time
    {
      std::lock_guard<std::mutex> g(m_);
      fn_ = &fn; parts_ = parts; pending_ = std::min(parts - 1, (int)th_.size()); gen_++; failed_ = nullptr;
    }
    cv_.notify_all();
    std::exception_ptr mine;
    try {
      fn(0);
      for (int k = (int)th_.size() + 1; k < parts; k++) fn(k);   // (more parts than threads: the caller takes the rest)
    } catch (...) { mine = std::current_exception(); }
    std::unique_lock<std::mutex> g(m_);
    done_.wait(g, [&] { return pending_ == 0; });
    fn_ = nullptr;
    std::exception_ptr theirs = failed_;
    failed_ = nullptr;
    g.unlock();
    if (mine) std::rethrow_exception(mine);
    if (theirs) std::rethrow_exception(theirs);
  }
 private:
  HostPool() {
    pid_ = getpid();
    const unsigned hw = std::thread::hardware_concurrency();
    // eight-way where the cores are: packing 270 MB of list rows is the longest job (5.4 ms = the PCIe time of the same bytes);
    // sixteen-way was slower on a GPU box's share of its host (6.1-6.4 ms, gpurun_out/r4y_md.txt)
    const int n = hw >= 16 ? 7 : (hw >= 4 ? 3 : (hw >= 2 ? (int)hw - 1 : 0));
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
        std::exception_ptr ex;
        try { (*fn)(part); } catch (...) { ex = std::current_exception(); }
        std::lock_guard<std::mutex> g(m_);
        if (ex && !failed_) failed_ = ex;
        if (--pending_ == 0) done_.notify_all();
      }
    }
  }
  std::vector<std::thread> th_;
  std::mutex m_, job_m_;
  std::condition_variable cv_, done_;
  const std::function<void(int)> *fn_ = nullptr;
  std::exception_ptr failed_;   // first exception of a helper in the running job
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
// ---- cutoff-mode: cell sort (perm/inv), then CSR lists in s space, all on the device ----------
inline int own_lo(const polar_handle *h) { return h->row_lo; }
inline int own_n(const polar_handle *h) { return (h->row_hi < 0 ? h->nlocal : h->row_hi) - h->row_lo; }
inline int norm_count(const polar_handle *h) { return (int)(h->global_count > 0 ? h->global_count : h->nlocal); }
inline bool sharded(const polar_handle *h) { return own_n(h) != h->nlocal; }
// the dd rows of the current lists sit in the launch order of the colouring in force (a colouring rebuilt after the
// lists were laid out -- a clash found on a reneighbor step, a changed alpha pattern -- makes them stale)
inline bool slots_current(const polar_handle *h) { return h->slots_by_color && h->slots_epoch == h->color_epoch; }
// `deterministic yes`, or -- keyword not given -- a fixed-iteration run: the reference returns ONE well-defined unconverged
// iterate there (PS.cpp:1211-1215), and the in-place race of a colour phase would make it differ run to run (4e-6 of the
// largest dipole after 13 sweeps, gpurun_out/r4a_tests.log) -- within an order of magnitude of the 1e-5 parity bar.
// `deterministic no` keeps the in-place update; `polar_accel` (refused with the commit) keeps it too.
inline bool deterministic(const polar_handle *h) {
  const polar_settings &st = h->ph.st;
  if (h->deterministic || st.deterministic == POLAR_DET_YES) return true;
  return st.deterministic == POLAR_DET_AUTO && st.fixed_iteration && st.polar_accel == 0;
}
// `deterministic yes` with the row sweep: where the end-of-sweep kernels find the sweep's partial sums of (dmu)^2
inline const double *det_part(const polar_handle *h) { return (deterministic(h) && h->ph.st.dd_cutoff > 0.0 && h->sweep_kernel == 2 && h->lp_npart > 0) ? h->d_lp_part.p : nullptr; }
inline int det_npart(const polar_handle *h) { return det_part(h) ? h->lp_npart : 0; }
// When the host looks at the device-resident loop state (a stream synchronisation: the queue drains, ~12 us before the next
// sweep starts).  Without history: after every `every`-th sweep.  With the sweep count of the handle's LAST solve -- counts
// change by a sweep or two from step to step, in MD as in the resident bench -- the first look comes where that solve ended
// (one look instead of eight at 31 sweeps, no launches past the end), then after every sweep for a few, then the old cadence.
// Sweeps launched past the end of a finished solve are no-ops on the device either way.  `step` = sweeps per possible
// change of the state (multi-GPU: the all-reduce cadence; the state can only flip after an all-reduced sweep).
inline bool look_at_state(const polar_handle *h, int sw, int every, int step = 1) {
  const int done_sweeps = sw + 1;
  const int pred = h->last_sweeps;
  if (pred <= 0) return (sw % every) == every - 1;
  if (done_sweeps < pred) return (done_sweeps % 16) == 0;   // (a coarse look on the way: a solve that ends much earlier than the last one wastes at most 16 no-op sweeps)
  const int past = done_sweeps - pred;
  if (past <= 4 * step) return (past % step) == 0;
  return (past % every) == 0;
}
inline bool tile_mode(const polar_handle *h) { return h->ph.st.dd_cutoff > 0.0 && h->sweep_kernel == 4; }
// where the dipoles live during a solve (exchange and debug kernels): the sweep records in tile mode, else the AtomRecs
inline MuView mu_view(const polar_handle *h) {
  if (tile_mode(h)) return MuView{reinterpret_cast<char *>(h->d_srec0.p), reinterpret_cast<char *>(h->d_srec1.p), (int)sizeof(SRec)};
  return MuView{reinterpret_cast<char *>(h->d_rec0.p), reinterpret_cast<char *>(h->d_rec1.p), (int)sizeof(AtomRec)};
}
// rows a per-row kernel should visit: nullptr = all rows 0..n-1 (identity)
inline const int *own_rows(const polar_handle *h) { return (h->sorted && sharded(h)) ? h->d_ownrows.p : nullptr; }

struct TileUnavailable : std::runtime_error {
  explicit TileUnavailable(const std::string &m) : std::runtime_error(m) {}
};

// ---- polar_step.hip ---------------------------------------------------------------------------------------------------
void launch_scan(long long n, const int *in, long long *out, DBuf<long long> &tot, hipStream_t s);
void build_cells(polar_handle *h);
void compute_slots(polar_handle *h);
void build_lists(polar_handle *h);
void launch_rank_pass(polar_handle *h, bool allpairs, int pass);   // a2, PS.cpp:192-227 (pass 1: rmin, pass 2: the metric)
void prepare_lp(polar_handle *h);
void sweep_once(polar_handle *h, bool ap);
void sweep_phase(polar_handle *h, int color, int part);   // one colour phase of the list-mode Gauss-Seidel: part 0 = all its rows, 1 = boundary rows, 2 = interior rows
void read_scal(polar_handle *h);
void debug_trace(polar_handle *h, int sw, bool jacobi);
bool accel_begin(polar_handle *h, bool ap);
void accel_export(polar_handle *h, double *dev_local_dots);
void accel_step(polar_handle *h, const double *global_dots);
void accel_export_fused(polar_handle *h, double *ared);
void accel_decide_mix(polar_handle *h, const double *ared, bool diff_first);
void solve(polar_handle *h, bool ap, polar_result *out);
void phase_begin(polar_handle *h, int eflag, int vflag, const double *mu_host);
int phase_finish(polar_handle *h, polar_result *out);
void clear_flags(polar_handle *h);
void tile_fallback(polar_handle *h);
bool grow_pitches(polar_handle *h);
int do_compute(polar_handle *h, int eflag, int vflag, const double *mu_host, polar_result *out);
void step_begin_lists(polar_handle *h, int eflag, int vflag);   // polar_step_begin up to the point where the colour phases are needed
bool step_needs_colors(const polar_handle *h);
void step_begin_finish(polar_handle *h);                        // ... and from there on (rows into launch order, descriptors)
void build_cluster_lists(polar_handle *h);
// ---- polar_color.hip --------------------------------------------------------------------------------------------------
void ensure_colors(polar_handle *h);
struct ColorComm {   // what a distributed colouring needs from the transport (polar_dist.hip)
  int my_class = 0, nclasses = 1;
  std::function<void(int *color_s_dev)> exchange;           // halo rows <- their owners' colours (s space, device)
  std::function<void(double *host, int count)> allreduce;   // sum over the ranks, in place (pinned host memory)
};
void build_colors_distributed(polar_handle *h, bool ranked, ColorComm &cc);
void map_color_rows(polar_handle *h);
void resolve_colors(polar_handle *h);
