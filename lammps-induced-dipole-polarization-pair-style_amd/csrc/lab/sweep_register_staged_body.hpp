// lab/sweep_register_staged_body.hpp -- LAB BUILD ONLY (-DPOLAR_LAB, libpolar_mi355x_lab.so): code that was built, measured and did not become the
// product path.  Included by polar_solver.hpp inside `#ifdef POLAR_LAB`; the product library never sees it.
// No include guard: it is a fragment of polar_solver.hpp, textually in that file's scope.
      // list mode (lab: the register-staged lane-per-pair sweep).  The damped tensor scalars (s3, s5) were cached per pair by k_dd_scalars, so a
      // sweep streams 20 B per pair (int32 j + two doubles) and gathers one 64-byte record.
      //   gather : scattered 16-byte loads cost one L1 (TCP) transaction per LANE, so the records of
      //            a trip's 64 pairs are fetched QUAD-cooperatively -- lane k of quad q loads piece k
      //            of record (r*16+q): 4 load instructions, each quad one coalesced 64-byte access;
      //   LDS    : the pieces are written to a per-wave staging tile (80-byte pitch: conflict-free
      //            b128 reads) and every lane reads back ITS pair's record: a wave-local transpose,
      //            no workgroup barrier (rows have different trip counts);
      //   math   : lane-per-pair, 64 pairs per VALU instruction.
      if (ablate & 1) end = beg;  // lab: no pair loop at all
      extern __shared__ double2 stage_all[];
      double2 *stage = stage_all + (size_t)(threadIdx.x >> 6) * (64 * 5);
      const int q4 = lane >> 2, k = lane & 3;
      // Three trips in flight (software pipeline): while trip t is transposed and computed, the
      // records of trip t+1 are being gathered and the index/scalar stream of trip t+2 is being read,
      // so a row pays its memory latencies once instead of twice per 64 pairs.
#define POLAR_LOAD_STREAM(BASE, JM, SC)                                   \
  {                                                                      \
    const long long p_ = (BASE) + lane;                                  \
    const bool ok_ = p_ < end;                                           \
    JM = (ok_ && !(ablate & 8)) ? dd_j[p_] : i;                          \
    SC = (ok_ && !(ablate & 4)) ? dd_s[p_] : make_double2(0.0, 0.0);     \
  }
#define POLAR_GATHER(JM, P0, P1, P2, P3)                                                   \
  {                                                                                        \
    int j0_, j1_, j2_, j3_;                                                                 \
    if (ablate & 128) { j0_ = JM; j1_ = JM ^ 1; j2_ = JM ^ 2; j3_ = JM ^ 3; } /* lab: no bpermute */ \
    else { j0_ = __shfl(JM, q4, 64); j1_ = __shfl(JM, 16 + q4, 64);                         \
           j2_ = __shfl(JM, 32 + q4, 64); j3_ = __shfl(JM, 48 + q4, 64); }                  \
    if (ablate & 2) j0_ = j1_ = j2_ = j3_ = i;                                             \
    P0 = reinterpret_cast<const double2 *>(src + j0_)[k];                                   \
    P1 = reinterpret_cast<const double2 *>(src + j1_)[k];                                   \
    P2 = reinterpret_cast<const double2 *>(src + j2_)[k];                                   \
    P3 = reinterpret_cast<const double2 *>(src + j3_)[k];                                   \
  }
      int jm0 = i, jm1 = i, jm2 = i;
      double2 sc0 = make_double2(0.0, 0.0), sc1 = sc0, sc2 = sc0;
      double2 pa0 = sc0, pa1 = sc0, pa2 = sc0, pa3 = sc0, pb0 = sc0, pb1 = sc0, pb2 = sc0, pb3 = sc0;
      if (beg < end) {
        POLAR_LOAD_STREAM(beg, jm0, sc0);
        POLAR_LOAD_STREAM(beg + 64, jm1, sc1);
        POLAR_GATHER(jm0, pa0, pa1, pa2, pa3);
      }
      for (long long base = beg; base < end; base += 64) {
        POLAR_LOAD_STREAM(base + 128, jm2, sc2);   // trip t+2 (predicated off past the row's end)
        POLAR_GATHER(jm1, pb0, pb1, pb2, pb3);     // trip t+1
        double2 a, b, c2;
        if (ablate & 64) {  // lab: no LDS transpose (wrong numbers, timing only)
          a = pa0; b = pa1; c2 = make_double2(pa2.x + pa3.x, pa2.y + pa3.y);
        } else {
        stage[(q4)*5 + k] = pa0; stage[(16 + q4) * 5 + k] = pa1;  // trip t
        stage[(32 + q4) * 5 + k] = pa2; stage[(48 + q4) * 5 + k] = pa3;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        a = stage[lane * 5]; b = stage[lane * 5 + 1]; c2 = stage[lane * 5 + 2];
        __builtin_amdgcn_wave_barrier();  // the tile is rewritten by the next trip
        }
        double dx, dy, dz;
        min_image_rint(box, ri.x, ri.y, ri.z, a.x, b.x, c2.x, dx, dy, dz);
        const double md = a.y * dx + b.y * dy + c2.y * dz;
        const double c = sc0.y * md;
        fx -= sc0.x * a.y - c * dx;
        fy -= sc0.x * b.y - c * dy;
        fz -= sc0.x * c2.y - c * dz;
        jm1 = jm2; sc0 = sc1; sc1 = sc2;
        pa0 = pb0; pa1 = pb1; pa2 = pb2; pa3 = pb3;
      }
#undef POLAR_LOAD_STREAM
#undef POLAR_GATHER
      fx = wave_sum(fx); fy = wave_sum(fy); fz = wave_sum(fz);
