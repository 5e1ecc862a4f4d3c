// polar_kernels.hpp -- hand-written HIP kernels (gfx950 / CDNA4, wave64) for the
// lj/cut/coul/long/polarization hot path.  FP64 throughout (the reference is FP64).
//
// Mapping used by every pair kernel: ONE WAVEFRONT PER ATOM ROW.  A 256-thread workgroup holds
// four rows; the 64 lanes stride the row's neighbor stream (coalesced 4-byte index loads), gather
// the 64-byte atom records of their j's (one cache line each), and the three field / force
// components are reduced across the wave with DPP/permute shuffles (no LDS, no atomics on i).
// No MFMA anywhere: this is a sparse neighbor stencil, not a dense contraction.
//
// Index spaces: "orig" = LAMMPS' atom index (inputs x/q/alpha/mol, outputs f/mu/ef_static);
// "s" = the library's internal order.  In list (dd_cutoff) mode s is CELL ORDER (perm[s] = orig,
// inv[orig] = s): the atoms of a cell are contiguous, so a row's neighbor stream -- emitted cell by
// cell -- makes the 64 lanes of a wave gather CONSECUTIVE 64-byte records, and the rows of one
// workgroup (neighbouring atoms) re-read the same records out of L1.  In exact (all-pairs) mode
// s == orig (perm == nullptr).
//
// Reference line numbers ("PS.cpp") are into
// /root/reference/src/pair_lj_cut_coul_long_polarization.cpp.
//
// The kernels live in seven headers: polar_common.hpp (types, wave helpers, image rules),
// polar_rows.hpp (the per-row kernels of a step), polar_solver.hpp (the dipole solver: list-mode sweep, loop control),
// polar_exact.hpp (exact mode's exact-order Gauss-Seidel), polar_accel.hpp (Anderson mixing, `polar_accel`),
// polar_lists.hpp (list mode: cells, neighbor lists, exchange), polar_tiles.hpp (list mode: the tile sweep
// -- one workgroup per cell, neighbour records staged in LDS -- and its builder).
#pragma once

#include "polar_common.hpp"
#include "polar_rows.hpp"
#include "polar_solver.hpp"
#include "polar_exact.hpp"
#include "polar_accel.hpp"
#include "polar_lists.hpp"
#include "polar_tiles.hpp"
