// polar_tiles.hpp -- list mode, the TILE sweep (a6 + a7, PS.cpp:1158-1180 with the tensor of PS.cpp:1273-1306).
// Part of the hand-written HIP kernels (gfx950 / CDNA4, wave64) of the lj/cut/coul/long/polarization
// hot path; see polar_kernels.hpp for the mapping and the index spaces.
//
// Why.  The row sweep (k_field_lp) gathers one 64-byte record per pair from L2 and uses it once: it runs at the
// chip's L2 gather rate (about 15 TB/s) whatever else is done to it.  Here a WORKGROUP owns a TILE = the polarizable
// atoms of one half-cutoff cell (about 15 rows).  It copies the records every row of the tile can see -- the UNION
// list, about 2.5 single-row lists, each entry already shifted into the tile's periodic image -- into LDS once
// (coalesced: the union is made of runs of consecutive records), and every row of the tile then gathers its
// partners out of LDS by 16-bit positions.  Per computed pair the L2 sees 1/6 of a record instead of one, the index
// stream is 2 bytes instead of 4, and no pair needs a minimum-image wrap.
//
// Gauss-Seidel order.  Launches = tile colours: cells of one parity class (2 x 2 x 2; a third class per dimension
// holds the last cell when a periodic dimension has an odd cell count) are never adjacent, so rows of two tiles of one
// launch are at least a cell edge (>= cutoff / 2) apart.  Inside a tile the rows are coloured greedily ON THE DEVICE,
// every step (rows closer than the colour distance get different sub-phases), and the workgroup walks the sub-phases
// with a barrier in between, writing each new dipole into its LDS copy and into the record table.  No host-side
// colouring, no rank metric, nothing to re-validate on reneighbor steps.
#pragma once

#include "polar_common.hpp"
#include "polar_solver.hpp"
#include "polar_lists.hpp"

namespace polar {

// 48-byte sweep record: what a pair needs from its partner (position, dipole), in the piece order of AtomRec
struct __attribute__((aligned(16))) SRec {
  double x, mx, y, my, z, mz;
};

#define POLAR_TILE_MAXSUB 11   // sub-phases (in-tile colours) a tile header can describe
#define POLAR_TILE_MAXROWS 128 // polarizable atoms of one cell the builder can colour (and the sweep's LDS row table holds)
#define POLAR_TILE_RECMASK 0x03FFFFFF  // union entry = record index | image code << 26 (code = (sx+1) + 3 (sy+1) + 9 (sz+1))

struct __attribute__((aligned(64))) TileHdr {
  int r0;     // first record of the cell (s space); local row m is record r0 + m
  int nrows;  // rows of this tile (polarizable atoms this handle owns); 0: nothing to do
  int U;      // entries of the union list = staged records (the dummy sits at position U)
  int nsub;   // sub-phases (low byte) | 0x100: some union entry is another periodic image
  int sub_off[POLAR_TILE_MAXSUB + 1];  // rows of sub-phase p: trow[r0 + sub_off[p] .. r0 + sub_off[p + 1])
};

// row table entry (16 bytes): rows of a tile in sub-phase order
struct __attribute__((aligned(16))) TileRowEnt {
  int mT;        // local row m (record r0 + m) | trips << 16
  int self;      // position of the row atom's own record in the union list
  double alpha;  // its polarizability (the sweep's epilogue needs it: mu = alpha (E_static + E_ind))
};

// tiles of one launch: cells c_k = start_k + stride_k * i_k, i_k < count_k
struct TileLaunch {
  int start[3], stride[3], count[3], nc[3];
};

// row lists: entry e of a row (trip e >> 6, lane e & 63) as a 16-bit LDS position, stored in chunks of 8 trips so that
// ONE 16-byte load per lane brings that lane's partners of eight trips (a typical row is 8-9 trips long)
__host__ __device__ __forceinline__ long long tile_slot16(long long e) {
  const long long t = e >> 6, lane = e & 63;
  return ((t >> 3) << 9) + (lane << 3) + (t & 7);
}

// periodic image shift of a union entry (whole lattice vectors a, b, c; domain.cpp:1258-1305 order)
__device__ __forceinline__ void tile_shift(const Box &b, int code, double &sx, double &sy, double &sz) {
  const int s0 = code % 3 - 1, s1 = (code / 3) % 3 - 1, s2 = code / 9 - 1;
  sx = s0 * b.prd[0] + s1 * b.xy + s2 * b.xz;
  sy = s1 * b.prd[1] + s2 * b.yz;
  sz = s2 * b.prd[2];
}

// LDS image of a workgroup: [tile header, 64 B][row table, MAXROWS x 16 B][records, (U + 1) x 48 B][slack]
#define POLAR_TILE_LDS_ROWS 64
#define POLAR_TILE_LDS_REC (POLAR_TILE_LDS_ROWS + 16 * POLAR_TILE_MAXROWS)
#define POLAR_TILE_LDS_SLACK 1024  // the last DMA instruction of the staging pass writes a whole 64-piece block

#ifdef POLAR_LAB  // the kernels below exist in the lab build only (the types and layout constants above are shared with the host code)
#include "lab/tile_kernels.hpp"
#endif  // POLAR_LAB

}  // namespace polar
