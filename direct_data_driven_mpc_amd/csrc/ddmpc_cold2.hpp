// ddmpc_cold2.hpp -- the cold-solve kernel for gfx950: one workgroup per controller instance, the whole r x r system
// in MFMA accumulator tiles, blocked Cholesky with 16-wide panels:
//
//   * tiles are kept K-MAJOR: register j of lane (l4, l15) of tile (I,J), I >= J, holds K[16J + l4 + 4j][16I + l15].
//     In this orientation an accumulator tile is directly the B
//     operand of a LEFT multiplication, so the triangular solve of a whole tile is 4 MFMAs with no LDS round trip:
//         U(J,I) = L_JJ^-1 K(J,I)          <-  D = (-M) * T,  M = L_JJ^-1 from LDS, T = the (negated) accumulator tile
//     and a dumped tile (PB[k][16I + i], k-major) is both the A and the B operand of the trailing update
//         K(J1,J2) -= U(Jp,J1)' U(Jp,J2)   <-  acc += mfma(PB[.][16 J1 + .], PB[.][16 J2 + .])    (matrix kept negated)
//   * wave 0 (the "panel wave") owns the main-diagonal tiles.  It factors the 16x16 diagonal tile by itself in four
//     4-wide sub-steps that never leave the wave (no workgroup barrier): 32 lanes = 16 tile rows + 16 rows of an
//     identity block that ride along, so the same substitution that produces L_JJ also produces M = L_JJ^-1; the
//     rank-4 updates inside the tile are 2 MFMAs per sub-step.  The other waves do nothing but MFMAs.
//   * per 16 columns: 3 workgroup barriers (M ready -> TRSM of the next diagonal tile's neighbour; panel dumped ->
//     trailing update; next M ready).
//   * L y = t rides along as column rE of the matrix; L' beta = y runs tile column by tile column, right to left, with
//     ONE workgroup barrier per column: every wave sums its tiles' contributions (4 multiply-adds per tile + one 16-lane
//     DPP reduction) into LDS, then every wave forms x_J = M_J' (y_J - parts) redundantly from the M tiles in LDS.
//   * the panel wave's chain (four 4-wide sub-steps per diagonal tile) is the critical path: the other waves' TRSM and
//     trailing updates of step J run beside the factorisation of diagonal tile J+1 (its update with panel J comes first),
//     and diagonal tile 0 is factored beside the other waves' Gram walks.
//   * the beta / active-set workspace is written only when asked for (ddmpc_get_solution re-solves on demand).
//
// Reference formulation: direct_data_driven_mpc_controller.py:409-445 (variables), :506-677 (constraints),
// :679-722 (cost), :780-808 (extraction).  DESIGN.md sections 3-5.
#pragma once
#include "ddmpc_kernels.hpp"

namespace ddmpc {

// Tile ownership: the main diagonal goes to wave 0, the other tile diagonals are dealt (longest first, to the least
// loaded wave) to waves 1..W-1; with a single wave everything is on wave 0.  Whole diagonals stay on one wave so that
// the structured Gram can walk down a diagonal inside one lane's registers.
template <int NT, int W>
struct TileMap2 {
  struct Tab { int wave[NT]; int base[NT]; int maxs; };
  static constexpr Tab make() {
    Tab t{};
    int load[W] = {};
    t.wave[0] = 0; t.base[0] = 0; load[0] = NT;
    for (int d = 1; d < NT; ++d) {
      int w = (W > 1) ? 1 : 0;
      for (int i = w + 1; i < W; ++i) if (load[i] < load[w]) w = i;
      t.wave[d] = w;
      t.base[d] = load[w];
      load[w] += NT - d;
    }
    t.maxs = 0;
    for (int i = 0; i < W; ++i) if (load[i] > t.maxs) t.maxs = load[i];
    return t;
  }
  static constexpr Tab tab = make();
  static constexpr int MAXS = tab.maxs;
  static constexpr int wave(int I, int J) { return tab.wave[I - J]; }
  static constexpr int slot(int I, int J) { return tab.base[I - J] + J; }
  // does wave w own a tile (J, K) with K < J in tile row J?
  static constexpr bool has_row(int w, int J) {
    for (int K = 0; K < J; ++K) if (wave(J, K) == w) return true;
    return false;
  }
  // does wave w own a tile (I, J) with I > J in tile column J?
  static constexpr bool has_col(int w, int J) {
    for (int I = J + 1; I < NT; ++I) if (wave(I, J) == w) return true;
    return false;
  }
};

template <int NT, int W, int WAVE>
struct WaveTiles2 {
  struct Tab { int I[NT * (NT + 1) / 2]; int J[NT * (NT + 1) / 2]; int n; };
  static constexpr Tab make() {
    Tab t{};
    t.n = 0;
    for (int J = 0; J < NT; ++J)
      for (int I = J; I < NT; ++I)
        if (TileMap2<NT, W>::wave(I, J) == WAVE) { t.I[t.n] = I; t.J[t.n] = J; ++t.n; }
    return t;
  }
  static constexpr Tab tab = make();
};

// Sum over the 16 lanes of a DPP row (every lane of the row receives the total): four rotate-and-add steps.
template <int CTRL>
__device__ __forceinline__ double dpp_mov_f64(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double row16_total(double v) {
  v += dpp_mov_f64<0x128>(v);   // row_ror:8
  v += dpp_mov_f64<0x124>(v);   // row_ror:4
  v += dpp_mov_f64<0x122>(v);   // row_ror:2
  v += dpp_mov_f64<0x121>(v);   // row_ror:1
  return v;
}

// v_permlane16_swap on a pair of doubles: result[0] = rows [a.r0, b.r0, a.r2, b.r2], result[1] = rows [a.r1, b.r1, a.r3, b.r3]
// (a.rN = the 16 lanes of DPP row N of a; probed on gfx950: tools/permlane_probe.hip).
__device__ __forceinline__ d2 permlane16_swap_f64(double a, double b) {
  const auto lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
  const auto hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
  return d2{__hiloint2double((int)hi[0], (int)lo[0]), __hiloint2double((int)hi[1], (int)lo[1])};
}

// 1/sqrt(x) for the pivots of the in-tile factorisation: hardware seed (v_rsq_f64, ~2^-26) + ONE Newton step
// y0 (1 + e/2), e = 1 - x y0^2: relative error 3/8 e^2 ~ 1e-16, two dependent operations shorter than a third-order
// correction; the pivots sit on the workgroup's critical path.
__device__ __forceinline__ double rsq_n2(double x) {
  const double y0 = __builtin_amdgcn_rsq(x);
  const double e = fma(-x * y0, y0, 1.0);
  return fma(0.5 * y0, e, y0);
}

// Sum of one value per lane over the four 16-lane rows of the wave (every lane receives the total), without LDS:
// swap(a, a) pairs rows (0,1) and (2,3), then the two halves.
__device__ __forceinline__ double rows4_total(double v) {
  const d2 s1 = permlane16_swap_f64(v, v);
  v = s1[0] + s1[1];
  const auto lo2 = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(v), (unsigned)__double2loint(v), false, false);
  const auto hi2 = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(v), (unsigned)__double2hiint(v), false, false);
  return __hiloint2double((int)hi2[0], (int)lo2[0]) + __hiloint2double((int)hi2[1], (int)lo2[1]);
}

// Maximum of one value per lane over the wave (every lane receives it): row rotations, then the two row exchanges.
__device__ __forceinline__ double wave_max(double v) {
  v = fmax(v, dpp_mov_f64<0x128>(v));   // row_ror:8
  v = fmax(v, dpp_mov_f64<0x124>(v));   // row_ror:4
  v = fmax(v, dpp_mov_f64<0x122>(v));   // row_ror:2
  v = fmax(v, dpp_mov_f64<0x121>(v));   // row_ror:1
  const d2 s1 = permlane16_swap_f64(v, v);
  v = fmax(s1[0], s1[1]);
  const auto lo2 = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(v), (unsigned)__double2loint(v), false, false);
  const auto hi2 = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(v), (unsigned)__double2hiint(v), false, false);
  return fmax(__hiloint2double((int)hi2[0], (int)lo2[0]), __hiloint2double((int)hi2[1], (int)lo2[1]));
}

// LDS carve-up (doubles).  Every offset is a compile-time function of (NT, W); the only runtime length is the trajectory
// region at the end.  The host sizes the launch from the same struct (Lds2<NT, W>::total, ddmpc_api.hip), and every
// aliasing / capacity assumption the kernel makes is a static_assert here: an overrun is a compile error, not a fault
// that shows up in one instance only.
template <int NT, int W>
struct Lds2 {
  static constexpr int RP = 16 * NT;
  static constexpr int dvec = 0;
  static constexpr int tvec = dvec + RP;
  static constexpr int beta = tvec + RP;
  static constexpr int cd0 = beta + RP;                // per-component constants of this instance (kept out of the registers):
  static constexpr int cd1 = cd0 + RP;                 //   D (inactive / bound active), target (past window folded in),
  static constexpr int ct0 = cd1 + RP;                 //   kind (ints, RP/2 doubles)
  static constexpr int ckk = ct0 + RP;
  static constexpr int PART_LEN = 2 * 16 * W;          // back substitution: 2 x [waves][16] partial sums (alternating rounds)
  static constexpr int part = ckk + (RP + 1) / 2;
  static constexpr int RED_LEN = 34;                   // block reductions: slots [w], [8 + w], [16 + w], [24 + w], w < W; [32]
  static constexpr int red = part + PART_LEN;
  static constexpr int ints = red + RED_LEN;           // int act[RP], int flags[8]
  static constexpr int INTS_LEN = (RP + 8 + 1) / 2;
  static constexpr int pt2 = (ints + INTS_LEN + 2) & ~1;   // in-tile panel, row-major: [32 rows][4]
  static constexpr int PT2_LEN = 32 * 4;
  static constexpr int LRS = 36;                       // row stride of lt16: 32 rows + 4 (spreads the 4 k-rows of an operand read over the banks)
  static constexpr int lt16 = pt2 + PT2_LEN;           // in-tile factor, k-major: [16][LRS]: x < 16 -> L_JJ[x][k], 16 + x' -> M[k][x']
  static constexpr int LT_LEN = 16 * LRS;
  static constexpr int pastw = lt16;                   // prologue only (lt16 is first written two barriers later): [u_past; y_past],
  static constexpr int PASTW_CAP = LT_LEN / 2;         //   at most PASTW_CAP entries (checked at create time), and each
  static constexpr int cpp = lt16 + PASTW_CAP;         //   component's index into it (RP ints)
  static constexpr int RSB = RP + 4;                   // row stride of the panel buffer
  static constexpr int pb = lt16 + LT_LEN;             // panel buffer, k-major: PB[k][16 I + i] = U(Jp, I)[k][i]
  static constexpr int PB_LEN = 16 * RSB;
  static constexpr int ctab = pb + PB_LEN;             // lag blocks C[d][a][b], d < RP/4 (structured Gram)
  static constexpr int CTAB_LEN = 4 * RP;
  // one 16x16 tile on its way from a helper wave to the panel wave (the deferred updates of the next-but-one diagonal
  // tile, see wave_body2): two halves of 128 doubles.  During the factorisation dvec (consumed by the fix-up) and beta
  // (written by the back substitution) are idle, so from 8 tile rows on the halves live there -- the benchmark instance
  // sits 2.4 KB below the LDS size at which a CU holds three workgroups; smaller instances get their own 2 KB.
  static constexpr bool DST_ALIAS = RP >= 128;
  static constexpr int dst = ctab + CTAB_LEN;
  static constexpr int DST_LEN = (W > 1 && !DST_ALIAS) ? 256 : 0;
  static constexpr int dst_lo = DST_ALIAS ? dvec : dst;
  static constexpr int dst_hi = DST_ALIAS ? beta : dst + 128;
  static constexpr int xs = dst + DST_LEN;             // trajectory, channel-interleaved
  // scratch of the residual check behind the solve (everything between pt2 and xs is free by then)
  static constexpr int SCR = pt2;
  static constexpr int SCR_LEN = dst - pt2;
  __host__ __device__ static constexpr int total(int xs_len) { return (xs + xs_len + 1) & ~1; }

  static_assert(W >= 1 && W <= 8, "red[] keeps 8 per-wave slots per quantity; part[] is sized by W");
  static_assert(24 + W <= 32 && RED_LEN >= 33, "red[24 + w] and red[32] (output stage) must stay inside red[]");
  static_assert(pt2 >= ints + INTS_LEN && (pt2 & 1) == 0, "act[RP] + flags[8] end before the in-tile panel; 16-byte aligned");
  static_assert(cpp + (RP + 1) / 2 <= lt16 + LT_LEN, "prologue staging (past window + its index table) must fit inside lt16");
  static_assert(NT * 256 <= PB_LEN, "back substitution keeps the NT inverse diagonal tiles M_J in the panel buffer");
  static_assert(16 * (RP / 4) <= CTAB_LEN, "one 4x4 lag block per time step of the window");
  static_assert((xs & 1) == 0 && (lt16 & 1) == 0 && (pb & 1) == 0, "16-byte alignment of the regions accessed with b128");
};

// Host-side mirror of the runtime capacity conditions (ddmpc_create refuses shapes that break them).
template <int NT, int W>
struct Lds2Limits {
  static constexpr int max_past = Lds2<NT, W>::PASTW_CAP;     // n (m + p) entries of [u_past; y_past]
  static constexpr int scratch = Lds2<NT, W>::SCR_LEN;        // doubles free for the residual check
};

// REF: compile the iterative-refinement loop in.  The plain variant (REF = false) is the fast one; when asked to
// (refine_flag != nullptr) it checks its own answer -- the residual t - (H (H' beta) + lam D beta) evaluated with exact
// products with the implicit Hankel matrix, two small MFMA contractions from the trajectory already in LDS -- and
// records whether it exceeds the threshold; the API re-solves exactly those instances with the REF variant
// (DDMPC_REFINE_AUTO).  Keeping the loop out of the plain variant keeps its register allocation free of spills on the
// factorisation's critical path.
// CVX: compile the rank-k treatment of the slack box in (see `rank_update` below): active-set iterations after the first keep
// the factor of the EMPTY active set and treat the k <= KC switched components as a diagonal modification of rank k instead
// of forming and factoring the whole system again.  A separate instantiation (plain variant only), so that the kernel of
// controllers without the box -- the headline -- keeps its code and register allocation bit for bit.
template <int NT, int W, int WAVE, bool REF, bool CVX>
__device__ __forceinline__ void wave_body2(const KParams& P, double* __restrict__ sm,
                                           const double* __restrict__ up, const double* __restrict__ yp,
                                           double* __restrict__ u_opt, double* __restrict__ cost_out,
                                           int* __restrict__ status_out, int* __restrict__ iters_out,
                                           double* __restrict__ beta_ws, signed char* __restrict__ act_ws,
                                           unsigned long long* __restrict__ stamps, double* __restrict__ lfac,
                                           double* __restrict__ lfacT, int* __restrict__ refine_flag,
                                           int* __restrict__ refine_count, const double* __restrict__ gpre,
                                           const double* __restrict__ ud_b, const double* __restrict__ yd_b) {
  // (ud_b, yd_b: this instance's trajectories in global memory -- read by the refinement loop of the REF variant when the
  //  trajectory is not staged, KParams::stage_xs = 0)
  using TM = TileMap2<NT, W>;
  using WT = WaveTiles2<NT, W, WAVE>;
  using LD = Lds2<NT, W>;
  constexpr int RP = 16 * NT;
  constexpr int NTHR = 64 * W;
  constexpr int LRS = LD::LRS, RSB = LD::RSB;
  int nstamp = 1;
  auto stamp = [&]() __attribute__((always_inline)) {
    if (stamps != nullptr && WAVE == 0) {
      const unsigned long long t = __builtin_amdgcn_s_memtime();
      if (threadIdx.x == 0 && nstamp < 7) stamps[nstamp] = t;
      ++nstamp;
    }
  };
  double* xs = sm + LD::xs;
  double* dvec = sm + LD::dvec;
  double* tvec = sm + LD::tvec;
  double* beta = sm + LD::beta;
  double* part = sm + LD::part;
  double* red = sm + LD::red;
  double* ctab = sm + LD::ctab;
  double* PT2 = sm + LD::pt2;
  double* LT = sm + LD::lt16;
  double* PB = sm + LD::pb;
  int* act = reinterpret_cast<int*>(sm + LD::ints);
  int* flags = act + RP;            // [0] fail, [1] active set changed

  int tid0 = threadIdx.x;
  asm volatile("" : "+v"(tid0));   // opaque: LDS addresses formed from it in the kernel function would be shared by the four wave bodies, live in scratch
  const int r = P.r, rE = P.rE, nch = P.nch;
  const int NS = rE >> 2;           // 4-wide pivot groups
  const int IR = rE >> 4;           // tile column holding the rhs column (column index rE)
  const int rr = rE & 15;

  // Deferred diagonal updates (W > 1).  The trailing updates of the main-diagonal tiles J >= 2 with the panels Jb <= J - 2
  // are NOT executed by the panel wave (every MFMA in its instruction stream delays its pivot chain, the critical path of
  // the workgroup; measured: with those 112 MFMAs in the chain's gaps the kernel takes 7 % longer than without them): a
  // helper wave accumulates  Delta_J = sum_{Jb <= J-2} U(Jb,J)' U(Jb,J)  in its own registers from the operands it loads
  // for its off-diagonal updates anyway, and hands the finished tile over through LDS (`dst`) behind barrier B of step
  // J - 2; the panel wave adds it (4 vector adds) and applies the last panel, J - 1, itself as before.
  constexpr int NHELP = (W > 1) ? W - 1 : 1;
  constexpr int DFIRST = 3;     // tiles J >= Jb + DFIRST are deferred to the helpers; the panel wave itself applies panels J-2 and J-1
  constexpr int NDSLOT = (W > 1 && NT > DFIRST) ? (NT - DFIRST + NHELP - 1) / NHELP : 1;
  auto downer = [](int J) constexpr { return 1 + (J - DFIRST) % NHELP; };
  auto dslot = [](int J) constexpr { return (J - DFIRST) / NHELP; };
  static_assert(8 * (TM::MAXS + (W > 1 ? NDSLOT : 0)) + 40 <= 512 / DDMPC_MIN_WAVES(NT, W),
                "the accumulator tiles of the busiest wave must leave a working set of 40 VGPRs at the occupancy of __launch_bounds__");
  d4 acc[TM::MAXS];
  d4 dlt[NDSLOT];
  double* DSTlo = sm + LD::dst_lo;    // registers 0, 1 of the tile in flight
  double* DSThi = sm + LD::dst_hi;    // registers 2, 3

  constexpr int NE = (RP + NTHR - 1) / NTHR;
  double* cD0 = sm + LD::cd0;
  double* cD1 = sm + LD::cd1;
  double* cT = sm + LD::ct0;
  int* cK = reinterpret_cast<int*>(sm + LD::ckk);
  {   // (cD0, cD1, cK, setpoint targets and the past window were staged by the kernel function, all loads in flight at once)
    const int* cP = reinterpret_cast<const int*>(sm + LD::cpp);
    const double* pastw = sm + LD::pastw;
    static_for<NE>([&](auto e) __attribute__((always_inline)) {
      const int rho = tid0 + e * NTHR;
      if (rho < RP) {
        const int pidx = cP[rho];
        if (pidx >= 0) cT[rho] = pastw[pidx];
        act[rho] = 0;
      }
    });
  }
  if (tid0 < 8) flags[tid0] = 0;

  // the panel wave's dependency chain (in-tile factorisation) is the critical path of the workgroup: its instructions
  // win the SIMD's issue arbitration against the MFMA waves of the co-resident workgroups
  // (measured: raising the panel wave's priority with s_setprio costs 3 %: it starves the MFMA waves of the co-resident workgroups)
  int iter = 0;
  int status = 0;
  int nupd = 0;                      // CVX: solves done by rank-k update so far (iter == 1 + nupd <=> the factor is the first one)
  bool ctab_ok = false;              // CVX: the lag-block table is in LDS (the rank-k update keeps W = L^-1 E in its place)
  int tid = tid0;
  double kmax = 0.0;                 // panel wave: largest diagonal entry of G = H H' (scale of the residual bound of AUTO refinement)
  long long tphF = 0, tphB = 0, tphT = 0, tphA = 0, tphU = 0;
  const bool timing = (stamps != nullptr) && (WAVE == 0);
  auto now = [&]() __attribute__((always_inline)) -> long long { return timing ? (long long)__builtin_amdgcn_s_memtime() : 0; };
  for (;;) {
    ++iter;
    asm volatile("" : "+v"(tid));       // opaque per-iteration thread id (keeps LICM from hoisting every LDS address)
    // ... and opaque problem sizes: otherwise every wave-uniform test on them (one per tile column and pivot group) is
    // hoisted out of this loop as a 64-bit lane mask, ~100 SGPRs that spill into VGPR lanes and push VGPRs into scratch
    int r_it = P.r, rE_it = P.rE;
    asm volatile("" : "+s"(r_it), "+s"(rE_it));
    const int r = r_it, rE = rE_it;
    const int NS = rE >> 2, IR = rE >> 4, rr = rE & 15;
    const int lane = tid & 63;
    const int l15 = lane & 15, l4 = lane >> 4, l3 = lane & 3, lo = l15 >> 2;
    static_for<NE>([&](auto e) __attribute__((always_inline)) {
      const int rho = tid + e * NTHR;
      if (rho < RP) {
        const int s_act = act[rho];
        dvec[rho] = s_act ? cD1[rho] : cD0[rho];
        tvec[rho] = cT[rho] + s_act * P.bound;
        beta[rho] = 0.0;
      }
    });
    if (tid == 0) flags[1] = 0;
    __syncthreads();   // trajectory staged (first pass), tables visible
    stamp();           // 1

    bool f0_done = false;   // structured Gram: the panel wave has already factored diagonal tile 0 (beside the other waves' walks)
    static_for<NDSLOT>([&](auto S) __attribute__((always_inline)) { dlt[S] = d4{0.0, 0.0, 0.0, 0.0}; });
    bool dst_valid = false;  // a finished Delta tile waits in `dst` (workgroup-uniform)
    // -(G + lam*D) of one raw Gram tile; -identity on dummy rows; rhs COLUMN rE := -t (mirrored into the diagonal tile);
    // dense weighting matrices: lam * W^-1 is a full symmetric matrix shared by the batch (L2)
    auto fix_tile = [&](auto II, auto JJ, const d4& raw) __attribute__((always_inline)) -> d4 {
      constexpr int I = II, J = JJ;
      const int gc = 16 * I + l15;                        // global column (i side)
      d4 v = -raw;
      if constexpr (I == J) {
        static_for<4>([&](auto j) __attribute__((always_inline)) {
          const int gr = 16 * J + l4 + 4 * j;
          if (gr == gc && gr < r) { kmax = fmax(kmax, raw[j()]); v[j()] -= P.lam * dvec[gr]; }
        });
      }
      if (16 * I + 15 >= r) {                             // wave-uniform: tile columns that touch the padding / rhs column
        static_for<4>([&](auto j) __attribute__((always_inline)) {
          const int gr = 16 * J + l4 + 4 * j;
          if (gr >= r || gc >= r) v[j()] = 0.0;
          if (I == J && gr == gc && gr >= r && gr < rE) v[j()] = -1.0;
          if (gc == rE && gr < r) v[j()] = -tvec[gr];
          if (I == J && gr == rE && gc < r) v[j()] = -tvec[gc];
        });
      }
      if (P.dense_w) {
        const double* dm = P.dmat + (long long)(16 * J + l4) * RP + 16 * I + l15;
        static_for<4>([&](auto j) __attribute__((always_inline)) { v[j()] -= P.lam * dm[4 * j() * RP]; });
      }
      return v;
    };
    // `pend(h)`, h = 0..11: hooks at which the caller slips independent MFMAs (trailing updates of later diagonal tiles)
    // into the dependency chain of the sub-step; pinned with sched_barrier so that they run beside the chain's VALU
    // work and LDS waits instead of in front of it.
    auto substep = [&](d4& Ad, d4& Et, auto QQ, int nq, auto&& pend) __attribute__((always_inline)) {
      constexpr int q = QQ;
      constexpr int c0 = 4 * q;
#define DDMPC_HOOK(h) do { pend(std::integral_constant<int, h>{}); } while (0)
      const int x = lane & 31;                                      // lanes 32..63 mirror lanes 0..31
      const double* Pd = PT2 + c0 * 4;                              // rows c0..c0+3 of the tile, 4 panel entries each
      const double p00 = -Pd[0];
      const double p10 = -Pd[4], p11 = -Pd[5];
      const double p20 = -Pd[8], p21 = -Pd[9], p22 = -Pd[10];
      const double p30 = -Pd[12], p31 = -Pd[13], p32 = -Pd[14], p33 = -Pd[15];
      const double r0 = PT2[x * 4 + 0], r1 = PT2[x * 4 + 1], r2 = PT2[x * 4 + 2], r3 = PT2[x * 4 + 3];
      DDMPC_HOOK(0);
      const double i0 = rsq_n2(p00);
      DDMPC_HOOK(1);
      const double l10 = p10 * i0, l20 = p20 * i0, l30 = p30 * i0;
      const double i1 = rsq_n2(p11 - l10 * l10);
      DDMPC_HOOK(2);
      const double l21 = (p21 - l20 * l10) * i1, l31 = (p31 - l30 * l10) * i1;
      const double i2 = rsq_n2(p22 - l20 * l20 - l21 * l21);
      DDMPC_HOOK(3);
      const double l32 = (p32 - l30 * l20 - l31 * l21) * i2;
      const double i3 = rsq_n2(p33 - l30 * l30 - l31 * l31 - l32 * l32);
      DDMPC_HOOK(4);
      const double x0 = -r0 * i0;
      const double x1 = -(r1 + x0 * l10) * i1;
      DDMPC_HOOK(5);
      const double x2 = -(r2 + x0 * l20 + x1 * l21) * i2;
      const double x3 = -(r3 + x0 * l30 + x1 * l31 + x2 * l32) * i3;
      DDMPC_HOOK(6);
      if (lane < 32) {                                              // kept for M / L_JJ / y; not on the chain
        LT[(c0 + 0) * LRS + x] = x0; LT[(c0 + 1) * LRS + x] = x1;
        LT[(c0 + 2) * LRS + x] = x2; LT[(c0 + 3) * LRS + x] = x3;
      }
      // MFMA operands without an LDS round trip.  Lane rows hold [tile rows, identity rows, tile rows, identity rows]
      // (x = lane & 31); v_permlane16_swap exchanges the odd rows of its first operand with the even rows of its second:
      //   swap(x0, x1) -> [x0.A x1.A x0.A x1.A], [x0.E x1.E x0.E x1.E];  swap(x2, x3) likewise,
      // so that operand row kk = x_kk of the tile rows (opA) / of the identity rows (opE) is one select away.
      const d2 s01a = permlane16_swap_f64(x0, x1), s23a = permlane16_swap_f64(x2, x3);
      const bool lowhalf = lane < 32;
      const double opA = lowhalf ? s01a[0] : s23a[0];
      const double opE = lowhalf ? s01a[1] : s23a[1];
      DDMPC_HOOK(7);
      Ad = __builtin_amdgcn_mfma_f64_16x16x4f64(opA, opA, Ad, 0, 0, 0);
      Et = __builtin_amdgcn_mfma_f64_16x16x4f64(opE, opA, Et, 0, 0, 0);
      DDMPC_HOOK(8);
      DDMPC_HOOK(9);
      DDMPC_HOOK(10);
      DDMPC_HOOK(11);
      if constexpr (q < 3) {
        if (q + 1 < nq && lo == q + 1) {
          static_for<4>([&](auto j) __attribute__((always_inline)) {
            PT2[(l4 + 4 * j) * 4 + l3] = Ad[j()];
            PT2[(16 + l4 + 4 * j) * 4 + l3] = Et[j()];
          });
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");       // in-wave hand-off through LDS (PT2, LT), see factor_begin
#undef DDMPC_HOOK
    };
    auto no_pend = [](auto) __attribute__((always_inline)) {};
    auto factor_begin = [&](const d4& Ad, d4& Et) __attribute__((always_inline)) {
      // (opaque lane id: otherwise the -identity pattern is hoisted out of the tile loop as four loop-invariant doubles,
      //  which then live in scratch memory and are reloaded on the critical path of every diagonal tile)
      int lz = lane;
      asm volatile("" : "+v"(lz));
      static_for<4>([&](auto j) __attribute__((always_inline)) { Et[j()] = ((lz >> 4) + 4 * j() == (lz & 15)) ? -1.0 : 0.0; });
      if (lo == 0) {
        static_for<4>([&](auto j) __attribute__((always_inline)) {
          PT2[(l4 + 4 * j) * 4 + l3] = Ad[j()];
          PT2[(16 + l4 + 4 * j) * 4 + l3] = Et[j()];
        });
      }
      // In-wave hand-off through LDS: lanes read what OTHER lanes of the wave just wrote.  Without a fence the compiler
      // reasons per thread ("my own stores did not touch this address, so the value I loaded last time is still good")
      // and forwards stale loads; the hardware itself executes a wave's LDS operations in order.
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    };
    // after the last sub-step: M into the diagonal tile's registers, y of the last tile column, optional export
    auto factor_end = [&](auto JT, int nq) __attribute__((always_inline)) {
      constexpr int Jt = JT;
      constexpr int SD = TM::slot(Jt, Jt);
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");       // LT was written by other lanes of this wave
      static_for<4>([&](auto j) __attribute__((always_inline)) {
        const int k = l4 + 4 * j();
        const double mv = LT[k * LRS + 16 + l15];
        acc[SD][j()] = (k < 4 * nq) ? mv : 0.0;                     // register j of lane (l4, l15) = M[l4 + 4j][l15]
      });
      if (Jt == IR) {                                               // y of the last tile column = the substituted rhs row
        if (lane < 4 * nq) tvec[16 * Jt + lane] = LT[lane * LRS + rr];
      }
      if (lfac != nullptr) {                                        // ddmpc_prepare: L_JJ row-major and its transpose
        static_for<4>([&](auto e) __attribute__((always_inline)) {
          const int idx = lane + 64 * e(), xr = idx >> 4, kc = idx & 15;
          double v = (kc <= xr && kc < 4 * nq) ? LT[kc * LRS + xr] : 0.0;
          if (xr == kc && kc >= 4 * nq) v = 1.0;
          lfac[(Jt * (Jt + 1) / 2 + Jt) * 256 + idx] = v;
          lfacT[(Jt * (Jt + 1) / 2 + Jt) * 256 + kc * 16 + xr] = v;
        });
      }
    };
    if (gpre != nullptr) {
      // ---- G = H H' formed ahead of the launch (KParams::gpre: the Hankel-structured Gram for any channel count)
      // (one 32-byte load per lane and tile.  Four 8-byte loads per tile from a row-major matrix -- cheaper to produce -- moved
      //  the register allocation of the whole kernel: 168 VGPRs and 116 B of scratch for <9,4> instead of 162 and none)
      static_for<WT::tab.n>([&](auto K) __attribute__((always_inline)) {
        constexpr int I = WT::tab.I[K], J = WT::tab.J[K];
        acc[TM::slot(I, J)] = *reinterpret_cast<const d4*>(gpre + (I * (I + 1) / 2 + J) * 256 + 4 * lane);
      });
      stamp();   // 2
      stamp();   // 3
    } else if (P.gram_dense) {
      // ---- G = H H' by fp64 MFMA over the implicit Hankel operand (k-major: rows of tile column J are the A operand)
      static_for<TM::MAXS>([&](auto S) __attribute__((always_inline)) { acc[S] = d4{0.0, 0.0, 0.0, 0.0}; });
      const int c = P.c;
      const int cfull = c & ~3;
      const double* xp = xs + l4 * nch + l15;
      for (int i0 = 0; i0 < cfull; i0 += 4) {
        double op[NT];
        static_for<NT>([&](auto I) __attribute__((always_inline)) { op[I] = xp[16 * I]; });
        xp += 4 * nch;
        static_for<WT::tab.n>([&](auto K) __attribute__((always_inline)) {
          constexpr int I = WT::tab.I[K], J = WT::tab.J[K];
          acc[TM::slot(I, J)] = __builtin_amdgcn_mfma_f64_16x16x4f64(op[J], op[I], acc[TM::slot(I, J)], 0, 0, 0);
        });
      }
      if (cfull < c) {
        const bool kok = (cfull + l4) < c;
        double op[NT];
        static_for<NT>([&](auto I) __attribute__((always_inline)) { const double v = xp[16 * I]; op[I] = kok ? v : 0.0; });
        static_for<WT::tab.n>([&](auto K) __attribute__((always_inline)) {
          constexpr int I = WT::tab.I[K], J = WT::tab.J[K];
          acc[TM::slot(I, J)] = __builtin_amdgcn_mfma_f64_16x16x4f64(op[J], op[I], acc[TM::slot(I, J)], 0, 0, 0);
        });
      }
      stamp();   // 2
      stamp();   // 3
    } else {
      // ---- structured Gram (nch == 4): lag blocks by v_mfma_f64_4x4x4, then 2 MFMAs per tile down every tile diagonal
      //      (the Hankel sliding-window recurrence in matrix form, see ddmpc_kernels.hpp) -- k-major tiles.
      // Two channels (SISO plants) ride on the same code: row rho = 2 t + ch of H is row rho of a FOUR-channel Hankel matrix
      // over the same flat trajectory (pseudo time t / 2, pseudo channel 2 (t % 2) + ch), whose columns step by 4 entries where
      // H's step by 2: H = [the pseudo matrix over x, columns 0, 2, 4, .. | the pseudo matrix over x + 2, columns 1, 3, ..], so
      // G is the sum of two four-channel Gram matrices -- every phase below runs over nsub trajectories `x + 2 sub` with
      // csub(sub) columns each and adds up (lag blocks: one more trip of the k-loop; base tiles: the window terms of both;
      // walks: 2 MFMAs per tile and trajectory).
      const int nsub = (nch == 2) ? 2 : 1;
      const int Ln = (nch == 2) ? (P.Ln + 1) >> 1 : P.Ln;
      auto csub = [&](int sub) __attribute__((always_inline)) { return (nch == 2) ? ((P.c + 1 - sub) >> 1) : P.c; };
      {
        // Every wave takes ALL lag groups (a half of them with 8 waves) over a quarter of the time range: the A operand
        // of a k-step is shared by the groups (x[t + 4 g + blk], g + u = const), so a trip of 4 k-steps needs
        // 4 + (MAXG + 3) LDS loads for 4 MAXG MFMAs -- 16 loads per 36 MFMAs for the benchmark instead of the 10 per 12 of
        // a split by groups (this phase is bound by LDS operand delivery), and the waves carry equal shares (9 groups
        // over 4 waves used to leave one wave idle and three with 3 groups each).  The partial sums meet in LDS.
        constexpr int TS = (W >= 4) ? 4 : W;                // time slices
        constexpr int GS = W / TS;                          // group slices (8 waves: 2)
        constexpr int MAXG = (NT + GS - 1) / GS;            // lag groups per wave
        static_assert(TS * GS == W, "waves = time slices x group slices");
        static_assert(TS * GS * MAXG * 64 <= LD::PT2_LEN + LD::LT_LEN + LD::PB_LEN, "partial lag blocks fit between pt2 and ctab");
        const int ngroups = (Ln + 3) >> 2;
        const int kq = lane >> 4, blk = (lane >> 2) & 3, ij = lane & 3;
        double* PP = sm + LD::pt2;                          // partial lag blocks [time slice][group][lane] (PT2 / LT / PB are idle)
        constexpr int tsl = WAVE % TS, gsl = WAVE / TS;
        const bool need_lag = CVX ? !ctab_ok : (iter == 1);     // (workgroup-uniform)
        if (need_lag) {
          double cacc[MAXG];
          static_for<MAXG>([&](auto gi) __attribute__((always_inline)) { cacc[gi()] = 0.0; });
#pragma nounroll
          for (int sub = 0; sub < nsub; ++sub) {
          const int c = csub(sub);
          const double* xb = xs + 2 * sub;
          const int cfull = c & ~3;
          const int nks = cfull >> 2;                       // full k-steps (4 time steps each)
          const int kw = (nks + TS - 1) / TS;
          const int ks0 = tsl * kw;
          const int ks1 = (ks0 + kw) < nks ? (ks0 + kw) : nks;
          const double* pB = xb + 4 * (kq + 4 * ks0) + ij;                               // B[k][j] = x_j[t0 + k]
          const double* pA = xb + 4 * (kq + blk + 4 * ks0) + ij + 16 * (gsl * MAXG);     // A[i][k] = x_i[t0 + k + 4g + blk]
          // 4 k-steps per trip: 4 + (MAXG + 3) loads, then 4 MAXG MFMAs (a second operand set in flight under the MFMAs
          // costs 60 spilled VGPRs in the Cholesky that follows: slower overall)
          auto ld = [&](double (&bv)[4], double (&av)[MAXG + 3]) __attribute__((always_inline)) {
            static_for<4>([&](auto u) __attribute__((always_inline)) { bv[u()] = pB[16 * u]; });
            static_for<MAXG + 3>([&](auto q) __attribute__((always_inline)) { av[q()] = pA[16 * q]; });
            pA += 64; pB += 64;
          };
          auto mm = [&](const double (&bv)[4], const double (&av)[MAXG + 3]) __attribute__((always_inline)) {
            static_for<4>([&](auto u) __attribute__((always_inline)) {
              static_for<MAXG>([&](auto gi) __attribute__((always_inline)) {
                cacc[gi()] = __builtin_amdgcn_mfma_f64_4x4x4f64(av[gi() + u()], bv[u()], cacc[gi()], 0, 0, 0);
              });
            });
          };
          const int nk = ks1 > ks0 ? ks1 - ks0 : 0;
          const int ntrip = nk >> 2;
          double bv0[4], av0[MAXG + 3];
#pragma nounroll
          for (int it = 0; it < ntrip; ++it) {
            ld(bv0, av0);
            mm(bv0, av0);
          }
          for (int k = 4 * ntrip; k < nk; ++k) {
            const double bv = pB[0];
            static_for<MAXG>([&](auto gi) __attribute__((always_inline)) {
              cacc[gi()] = __builtin_amdgcn_mfma_f64_4x4x4f64(pA[16 * gi], bv, cacc[gi()], 0, 0, 0);
            });
            pA += 16; pB += 16;
          }
          if (tsl == TS - 1 && cfull < c) {                  // the ragged last k-step: the last time slice (pA / pB stand at cfull)
            const bool kok = (cfull + kq) < c;
            const double bv = kok ? pB[0] : 0.0;
            static_for<MAXG>([&](auto gi) __attribute__((always_inline)) {
              const double a1 = pA[16 * gi];
              cacc[gi()] = __builtin_amdgcn_mfma_f64_4x4x4f64(kok ? a1 : 0.0, bv, cacc[gi()], 0, 0, 0);
            });
          }
          }   // sub
          static_for<MAXG>([&](auto gi) __attribute__((always_inline)) {
            PP[((tsl * GS + gsl) * MAXG + gi()) * 64 + lane] = cacc[gi()];
          });
        }
        __syncthreads();
        if (need_lag) {
          for (int e = tid; e < GS * MAXG * 64; e += NTHR) {
            const int g = e >> 6, ln = e & 63;              // (group slice, group in slice) = (g / MAXG, g % MAXG): g is the group
            double sacc = 0.0;
            static_for<TS>([&](auto t) __attribute__((always_inline)) { sacc += PP[(t() * GS * MAXG + g) * 64 + ln]; });
            const int d = 4 * g + ((ln >> 2) & 3);          // D layout: i = lane>>4, j = lane&3
            if (g < ngroups && d < Ln) ctab[d * 16 + (ln >> 4) * 4 + (ln & 3)] = sacc;
          }
        }
      }
      __syncthreads();
      ctab_ok = true;
      stamp();   // 2
      static_for<TM::MAXS>([&](auto S) __attribute__((always_inline)) { acc[S] = d4{0.0, 0.0, 0.0, 0.0}; });
      // (2) first tile of every owned tile diagonal from the lag blocks.  Register j of lane (l4, l15 = 4 lo + l3) of
      //     tile (d, 0) is K[(time j, channel l4)][(time 4d + lo, channel l3)] = G(t1, t2)(a, b) with
      //       G(l + del, l)(a, b) = C_del(a, b) + sum_{q<l} ( x_a[q+c+del] x_b[q+c] - x_a[q+del] x_b[q] ),
      //     the later time index taking the role of "l + del" (the matrix is symmetric).  Diagonal tiles are filled
      //     on both sides of their diagonal.
      {
#pragma nounroll
        for (int d = 0; d < NT; ++d) {                     // runtime loop (an unrolled one gets hoisted into spills)
          bool mine = false;
          static_for<NT>([&](auto DD) __attribute__((always_inline)) {
            if constexpr (TM::tab.wave[DD] == WAVE) mine = mine || (d == DD);
          });
          if (!mine) continue;
          auto base = [&](int j) __attribute__((always_inline)) -> double {
            int del = 4 * d + lo - j;
            const bool neg = del < 0;
            del = neg ? -del : del;
            del = del >= Ln ? Ln - 1 : del;                // padded rows: cleared in the fix-up
            const int pa = neg ? l4 : l3, pb = neg ? l3 : l4;
            const int nl = neg ? lo : j;                   // window terms: the earlier of the two time indices
            double t = ctab[del * 16 + pa * 4 + pb];
#pragma nounroll
            for (int sub = 0; sub < nsub; ++sub) {
              const int c = csub(sub);
              const double* qa0 = xs + 2 * sub + 4 * del + pa;         // x_pa[del + .]
              const double* qa1 = qa0 + 4 * c;
              const double* qb0 = xs + 2 * sub + pb;
              const double* qb1 = qb0 + 4 * c;
              const double e0 = qa1[0] * qb1[0] - qa0[0] * qb0[0];
              const double e1 = qa1[4] * qb1[4] - qa0[4] * qb0[4];
              const double e2 = qa1[8] * qb1[8] - qa0[8] * qb0[8];
              t += (0 < nl) ? e0 : 0.0;
              t += (1 < nl) ? e1 : 0.0;
              t += (2 < nl) ? e2 : 0.0;
            }
            return t;
          };
          const d4 v = d4{base(0), base(1), base(2), base(3)};
          static_for<NT>([&](auto DD) __attribute__((always_inline)) {
            if constexpr (TM::tab.wave[DD] == WAVE) { if (d == DD) acc[TM::slot(DD, 0)] = v; }
          });
        }
      }
      // (3) walk down the diagonals: tile(I+1,J+1) = tile(I,J) - (first 4 Hankel columns) + (columns c..c+3)
      static_for<NT>([&](auto DD) __attribute__((always_inline)) {
        constexpr int d = DD;
        if constexpr (TM::tab.wave[d] == WAVE && d + 1 < NT) {
          const double* pb = xs + 4 * l4 + l15;          // rows of tile column J = t: the A operand (k side)
          const double* pa = pb + 16 * d;                // rows of tile row I = d + t: the B operand (i side)
          const int c = csub(0);
#pragma nounroll
          for (int t = 0; t + 1 < NT - d; ++t) {
            const double a1 = pa[0], b1 = pb[0], a2 = pa[4 * c], b2 = pb[4 * c];
            double a3 = 0.0, b3 = 0.0, a4 = 0.0, b4 = 0.0;
            if (nsub == 2) { const int c1 = csub(1); a3 = pa[2]; b3 = pb[2]; a4 = pa[2 + 4 * c1]; b4 = pb[2 + 4 * c1]; }   // (kernel-uniform)
            static_for<NT - d - 1>([&](auto T) __attribute__((always_inline)) {
              if (t == T) {
                d4 v = __builtin_amdgcn_mfma_f64_16x16x4f64(-b1, a1, acc[TM::slot(d + T, T)], 0, 0, 0);
                if (nsub == 2) {
                  v = __builtin_amdgcn_mfma_f64_16x16x4f64(-b3, a3, v, 0, 0, 0);
                  v = __builtin_amdgcn_mfma_f64_16x16x4f64(b4, a4, v, 0, 0, 0);
                }
                acc[TM::slot(d + T + 1, T + 1)] = __builtin_amdgcn_mfma_f64_16x16x4f64(b2, a2, v, 0, 0, 0);
              }
            });
            pa += 16; pb += 16;
          }
        }
      });
      // (4) panel wave: its walk is short (one diagonal); diagonal tile 0 is fixed up and factored NOW, beside the other
      //     waves' walks and fix-ups (its chain is the longest single piece of the workgroup's critical path)
      if constexpr (WAVE == 0 && W > 1) {
        if (rE >= 16) {                                   // (tiny systems keep the plain order: their rhs row sits in tile 0)
          constexpr int S0 = TM::slot(0, 0);
          d4 Ad = fix_tile(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, acc[S0]);
          d4 Et;
          factor_begin(Ad, Et);
          static_for<4>([&](auto q) __attribute__((always_inline)) { substep(Ad, Et, q, 4, no_pend); });
          factor_end(std::integral_constant<int, 0>{}, 4);
          f0_done = true;
        }
      }
      stamp();   // 3
    }

    // ---- accumulators := -(G + lam*D) (fix_tile above)
    static_for<WT::tab.n>([&](auto K) __attribute__((always_inline)) {
      constexpr int I = WT::tab.I[K], J = WT::tab.J[K];
      constexpr int S = TM::slot(I, J);
      if (!(I == 0 && J == 0 && f0_done)) acc[S] = fix_tile(std::integral_constant<int, I>{}, std::integral_constant<int, J>{}, acc[S]);
    });
    __syncthreads();      // every wave has read tvec (targets) for its rhs entries; LT / PT2 / PB are free

    // ---- blocked Cholesky, 16-wide panels, software-pipelined over the tile columns --------------------------------
    // Step Jb (M_Jb = L_JJ^-1 of tile column Jb is in LT):
    //   (T1) the owner of tile (Jb+1, Jb) solves it (4 MFMAs) and dumps it to PB            -- barrier A1
    //   panel wave: updates diagonal tile Jb+1 from that dump and starts factoring it (sub-step 0)
    //   other waves: (T2) solve + dump their other tiles of tile row Jb                       -- barrier A2
    //   other waves: (U) trailing update of their tiles; panel wave: sub-steps 1..3 of the factorisation, with the
    //   trailing updates of the diagonal tiles Jb+2.. slipped into its instruction stream      -- barrier B: M_{Jb+1} in LT
    // so the factorisation chain of the next diagonal tile runs beside the TRSM / trailing update of the current one.
    //
    // In-tile factorisation (panel wave, no workgroup barrier inside except A2): four 4-wide sub-steps.  Lane x < 16 owns
    // tile row x, lane 16 + x' row x' of a (negated) identity block that rides along: the same substitution that yields
    // the rows of L_JJ yields the columns of M = L_JJ^-1.  Every lane factors the 4x4 pivot block redundantly
    // (v_rsq_f64 + one third-order correction per pivot), substitutes its row, the rows go to LT (k-major); the rank-4
    // update inside the two tiles is one MFMA each, and the next panel's columns go back out to PT2.  A non-positive
    // pivot yields NaN, which reaches beta and is reported through the finite check of the output stage.  Rows above /
    // inside the pivot group carry don't-care values in LT (they only touch finished entries); the export masks them.
    // prologue: the panel wave factors diagonal tile 0 on its own
    {
      const int nq0 = NS < 4 ? NS : 4;
      const long long t0 = now();
      if constexpr (WAVE == 0) {
        if (!f0_done) {
          d4 Ad = acc[TM::slot(0, 0)];
          d4 Et;
          factor_begin(Ad, Et);
          static_for<4>([&](auto q) __attribute__((always_inline)) { if (q() < nq0) substep(Ad, Et, q, nq0, no_pend); });
          factor_end(std::integral_constant<int, 0>{}, nq0);
        }
      }
      const long long t1 = now();
      __syncthreads();                                              // (B) M_0 is in LT
      tphF += t1 - t0; tphB += now() - t1;
    }
    static_for<NT>([&](auto JB) __attribute__((always_inline)) {
      constexpr int Jb = JB;
      const int nqc = (NS - 4 * Jb) < 4 ? (NS - 4 * Jb) : 4;       // pivot groups of this tile column (<= 0: none)
      if (nqc > 0) {                                                // workgroup-uniform
        const bool more = 16 * (Jb + 1) < rE + 1;                   // live columns to the right of this panel
        const int nqn = (NS - 4 * (Jb + 1)) < 4 ? (NS - 4 * (Jb + 1)) : 4;   // pivot groups of the next tile column
        const long long t0 = now();
        if constexpr (WAVE == 0 && W > 1 && Jb >= 1 && Jb + DFIRST - 1 < NT) {
          if (dst_valid) {                                         // Delta of tile Jb + DFIRST - 1 (panels 0 .. Jb - 1), written before barrier B
            constexpr int SDn = TM::slot(Jb + DFIRST - 1, Jb + DFIRST - 1);
            acc[SDn][0] += DSTlo[lane]; acc[SDn][1] += DSTlo[64 + lane];
            acc[SDn][2] += DSThi[lane]; acc[SDn][3] += DSThi[64 + lane];
          }
        }
        dst_valid = false;
        // M operand of the triangular solves: read before A1, the panel wave overwrites LT after it
        double am[4];
        static_for<4>([&](auto ks) __attribute__((always_inline)) { am[ks()] = -LT[l15 * LRS + 16 + 4 * ks() + l4]; });
        auto trsm_tile = [&](auto II) __attribute__((always_inline)) {
          constexpr int I = II;
          if (16 * I < rE + 1) {
            constexpr int S = TM::slot(I, Jb);
            const d4 T = acc[S];
            d4 dd = d4{0.0, 0.0, 0.0, 0.0};
            dd = __builtin_amdgcn_mfma_f64_16x16x4f64(am[0], T[0], dd, 0, 0, 0);
            dd = __builtin_amdgcn_mfma_f64_16x16x4f64(am[1], T[1], dd, 0, 0, 0);
            dd = __builtin_amdgcn_mfma_f64_16x16x4f64(am[2], T[2], dd, 0, 0, 0);
            dd = __builtin_amdgcn_mfma_f64_16x16x4f64(am[3], T[3], dd, 0, 0, 0);
            acc[S] = dd;
            static_for<4>([&](auto j) __attribute__((always_inline)) { PB[(l4 + 4 * j) * RSB + 16 * I + l15] = dd[j()]; });
          }
        };
        // (T1) the tile the next diagonal tile waits for
        if constexpr (Jb + 1 < NT) {
          if constexpr (TM::wave(Jb + 1, Jb) == WAVE) { if (more) trsm_tile(std::integral_constant<int, Jb + 1>{}); }
        }
        const long long t1 = now();
        __syncthreads();                                            // (A1)
        const long long t2 = now();
        d4 Ad = d4{0.0, 0.0, 0.0, 0.0}, Et = d4{0.0, 0.0, 0.0, 0.0};
        if constexpr (WAVE == 0 && Jb + 1 < NT) {
          if (nqn > 0) {                                            // next diagonal tile: trailing update, first sub-step
            constexpr int S = TM::slot(Jb + 1, Jb + 1);
            Ad = acc[S];
            const long long tu0 = now();
            static_for<4>([&](auto ks) __attribute__((always_inline)) {
              const double o = PB[(4 * ks() + l4) * RSB + 16 * (Jb + 1) + l15];
              Ad = __builtin_amdgcn_mfma_f64_16x16x4f64(o, o, Ad, 0, 0, 0);
            });
            tphU += now() - tu0;
            factor_begin(Ad, Et);
            substep(Ad, Et, std::integral_constant<int, 0>{}, nqn, no_pend);
          }
        }
        // (T2) the other tiles of tile row Jb
        if (more) {
          static_for<NT>([&](auto I) __attribute__((always_inline)) {
            if constexpr (I > Jb + 1 && TM::wave(I, Jb) == WAVE) trsm_tile(I);
          });
        }
        const long long t3 = now();
        __syncthreads();                                            // (A2) the whole panel is in PB
        const long long t4 = now();
        if constexpr (WAVE == 0 && Jb + 1 < NT) {
          // sub-steps 1..3 of the next diagonal tile; the trailing updates of the diagonal tiles Jb+2.. (operands from
          // PB, which stays valid until barrier B) are dealt over the three sub-steps
          static_for<3>([&](auto QM) __attribute__((always_inline)) {
            constexpr int q = QM + 1;
            // diagonal tiles Jb+2+QM, +3, +6, ...: 4 MFMAs each, issued k-step-major (consecutive MFMAs hit different tiles)
            constexpr int first = Jb + 2 + QM;
            constexpr int last = (W > 1) ? ((Jb + DFIRST - 1 < NT - 1) ? Jb + DFIRST - 1 : NT - 1) : NT - 1;   // helpers take the rest
            constexpr int ntl = first <= last ? (last - first) / 3 + 1 : 0;
            auto pend = [&](auto H) __attribute__((always_inline)) {
              constexpr int h0 = H;
              static_for<(4 * ntl + 11) / 12>([&](auto R) __attribute__((always_inline)) {   // more than 12 pending: several per hook
                constexpr int h = h0 + 12 * R;
                if constexpr (h < 4 * ntl) {
                  constexpr int J2 = first + 3 * (h % ntl), ks = h / ntl;
                  constexpr int S2 = TM::slot(J2, J2);
                  const double o = PB[(4 * ks + l4) * RSB + 16 * J2 + l15];
                  acc[S2] = __builtin_amdgcn_mfma_f64_16x16x4f64(o, o, acc[S2], 0, 0, 0);
                }
              });
            };
            if (q < nqn) substep(Ad, Et, std::integral_constant<int, q>{}, nqn, pend);
          });
          if (nqn > 0) factor_end(std::integral_constant<int, Jb + 1>{}, nqn);
        }
        // (U) trailing update K(J, I) += U(Jb, J)' U(Jb, I) (negated matrix) of the off-diagonal tiles, 4 k-steps of 4
        if (more) {
          static_for<4>([&](auto ks) __attribute__((always_inline)) {
            double op[NT];
            static_for<NT>([&](auto X) __attribute__((always_inline)) {
              if constexpr (X > Jb) op[X] = PB[(4 * ks() + l4) * RSB + 16 * X + l15];
            });
            static_for<WT::tab.n>([&](auto K) __attribute__((always_inline)) {
              constexpr int I = WT::tab.I[K], J = WT::tab.J[K];
              if constexpr (J > Jb && I > J) {
                if (16 * I < rE + 1) {
                  constexpr int S = TM::slot(I, J);
                  acc[S] = __builtin_amdgcn_mfma_f64_16x16x4f64(op[J], op[I], acc[S], 0, 0, 0);
                }
              }
            });
            if constexpr (WAVE != 0 && W > 1) {                     // deferred diagonal updates of this wave's tiles
              static_for<NT>([&](auto JD) __attribute__((always_inline)) {
                constexpr int J = JD;
                if constexpr (J >= Jb + DFIRST) {
                  if constexpr (downer(J) == WAVE) {
                    if (16 * J < rE + 1) dlt[dslot(J)] = __builtin_amdgcn_mfma_f64_16x16x4f64(op[J], op[J], dlt[dslot(J)], 0, 0, 0);
                  }
                }
              });
            }
          });
          if constexpr (W > 1 && Jb + DFIRST < NT) {
            if constexpr (WAVE != 0) {
              if constexpr (downer(Jb + DFIRST) == WAVE) {          // Delta of tile Jb + DFIRST is complete: hand it over
                const d4 dv = dlt[dslot(Jb + DFIRST)];
                DSTlo[lane] = dv[0]; DSTlo[64 + lane] = dv[1]; DSThi[lane] = dv[2]; DSThi[64 + lane] = dv[3];
              }
            }
            dst_valid = true;
          }
        }
        const long long t5 = now();
        __syncthreads();                                            // (B) M_{Jb+1} is in LT
        const long long t6 = now();
        tphT += t1 - t0; tphA += (t2 - t1) + (t4 - t3); tphF += (t3 - t2) + (t5 - t4); tphB += t6 - t5;   // F includes U (wave 0)
      }
    });
    if (timing && threadIdx.x == 0) {
      stamps[7] = tphF; stamps[8] = tphB;
      if constexpr (CVX) { stamps[9] = 0; stamps[10] = 0; stamps[11] = 0; }         // (the rank-k update's three phases go there)
      else { stamps[9] = tphT; stamps[10] = tphA; stamps[11] = tphU; }
    }
    stamp();   // 4

    // ---- optional export of the factor (ddmpc_prepare): lower tiles L(I,J), row-major 16x16 each, at
    //      lfac[(I(I+1)/2 + J) * 256]; lfacT holds the transposed tiles (what the accumulators hold as they are)
    if (lfac != nullptr) {
      static_for<WT::tab.n>([&](auto K) __attribute__((always_inline)) {
        constexpr int I = WT::tab.I[K], J = WT::tab.J[K];
        if constexpr (I > J) {
          constexpr int S = TM::slot(I, J);
          static_for<4>([&](auto j) __attribute__((always_inline)) {
            lfacT[(I * (I + 1) / 2 + J) * 256 + (l4 + 4 * j) * 16 + l15] = acc[S][j()];
            lfac[(I * (I + 1) / 2 + J) * 256 + l15 * 16 + l4 + 4 * j] = acc[S][j()];
          });
        }
      });
    }

    // ---- y = L^-1 t: column rr of the tiles in tile row IR (the last tile column wrote its part above) ----------
    static_for<WT::tab.n>([&](auto K) __attribute__((always_inline)) {
      constexpr int I = WT::tab.I[K], J = WT::tab.J[K];
      if constexpr (I > J) {
        if (I == IR && l15 == rr) {
          constexpr int S = TM::slot(I, J);
          static_for<4>([&](auto j) __attribute__((always_inline)) { tvec[16 * J + l4 + 4 * j] = acc[S][j()]; });
        }
      }
    });
    // ---- back substitution U x = y, one tile column per round, right to left, ONE workgroup barrier per round.  The panel wave
    //      first puts its M tiles into the (now free) panel buffer.
    //          x_J = M_J' (y_J - sum_{I >= J+2} U(J,I) x_I)  -  (M_J' U(J,J+1)) x_{J+1}
    //      Only the last term needs the column just finished, and the whole first sub-diagonal of tiles belongs to one wave (the
    //      chain wave CW; TileMap2 deals whole diagonals).  That wave forms  Ct_J = U(J,J+1)' M_J  on the matrix pipe one round
    //      ahead -- both operands in their natural layouts: the accumulator tile as A, M_J from LDS as B -- and the product comes
    //      out with x_J's index on the lanes: the critical term is 4 multiply-adds and ONE reduction over the four lane rows
    //      (with the tile itself it is a 16-lane reduction per register: 48 instructions), x_{J+1} read back from LDS in the
    //      register-per-row form.  Per round J:
    //        chain wave: v = y_J - the far terms (formed during the round before), x_J = M_J' v - Ct_J x_{J+1} -> `out`; Ct_{J-1}
    //        every wave, beside it: the far terms I >= J + 1 of column J - 1 -> part[parity of the column][wave][k]
    //      (Until round 5 every wave summed ALL its terms of column J after the barrier of column J + 1, 16-lane reductions
    //      included, and every wave formed x_J redundantly: 1.6 K cycles per round, 15 K of the benchmark's 126 K.)
    auto back_substitute = [&](double* __restrict__ out, const double* __restrict__ yin, const int rE) __attribute__((always_inline)) {
      double* Mb = PB;                                               // Mb[J][k][i] = M_J[k][i]
      constexpr int CW = TM::wave(1, 0);
      auto far_col = [](int w, int J) constexpr { for (int I = J + 2; I < NT; ++I) if (TM::wave(I, J) == w) return true; return false; };
      if constexpr (WAVE == 0) {
        static_for<NT>([&](auto J) __attribute__((always_inline)) {
          if (16 * J < rE) {
            constexpr int SD = TM::slot(J, J);
            static_for<4>([&](auto j) __attribute__((always_inline)) { Mb[J * 256 + (l4 + 4 * j) * 16 + l15] = acc[SD][j()]; });
          }
        });
      }
      __syncthreads();                                               // M tiles and y are in LDS
      double xl[NT];                                                 // x_I[l15] of the rounds done so far
      static_for<NT>([&](auto I) __attribute__((always_inline)) { xl[I] = 0.0; });
      d4 Ct = d4{0.0, 0.0, 0.0, 0.0};                                // chain wave: Ct_J, register q of lane (l4, l15) = (M_J' U(J,J+1))[l15][l4 + 4q]
      d4 xk = d4{0.0, 0.0, 0.0, 0.0};                                //             x_{J+1}[l4 + 4q]
      static_for<NT>([&](auto JREV) __attribute__((always_inline)) {
        constexpr int J = NT - 1 - JREV;
        if (16 * J < rE) {                                            // workgroup-uniform
          if constexpr (WAVE == CW) {
            const double* pr = part + (J & 1) * (16 * W);
            double pi = 0.0;
            static_for<4>([&](auto j) __attribute__((always_inline)) {
              const int k = l4 + 4 * j();
              double v = yin[16 * J + k];
              if constexpr (J + 2 < NT) {
                if (16 * (J + 2) < rE) {
                  static_for<W>([&](auto w) __attribute__((always_inline)) {
                    if constexpr (far_col(w(), J)) v -= pr[w * 16 + k];
                  });
                }
              }
              pi = fma(Mb[J * 256 + k * 16 + l15], v, pi);
            });
            static_for<4>([&](auto q) __attribute__((always_inline)) { pi = fma(-Ct[q()], xk[q()], pi); });   // (zero for the first column)
            pi = rows4_total(pi);
            xl[J] = (16 * J + l15 < rE) ? pi : 0.0;
            if (l4 == 0) out[16 * J + l15] = xl[J];
            if constexpr (J >= 1) {
              constexpr int S = TM::slot(J, J - 1);                   // Ct_{J-1} = U(J-1,J)' M_{J-1}, and x_J in the register-per-row form
              Ct = d4{0.0, 0.0, 0.0, 0.0};
              static_for<4>([&](auto j) __attribute__((always_inline)) {
                Ct = __builtin_amdgcn_mfma_f64_16x16x4f64(acc[S][j()], Mb[(J - 1) * 256 + (4 * j() + l4) * 16 + l15], Ct, 0, 0, 0);
              });
              __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); // `out` was written by other lanes of this wave
              static_for<4>([&](auto q) __attribute__((always_inline)) { xk[q()] = out[16 * J + l4 + 4 * q()]; });
            }
          }
          // one column ahead: the far terms I >= J + 1 of column J - 1
          if constexpr (J >= 1) {
            if constexpr (far_col(WAVE, J - 1)) {
              if (16 * (J + 1) < rE) {
                d4 sv = d4{0.0, 0.0, 0.0, 0.0};
                static_for<NT>([&](auto I) __attribute__((always_inline)) {
                  if constexpr (I >= J + 1 && TM::wave(I, J - 1) == WAVE) {
                    constexpr int S = TM::slot(I, J - 1);
                    static_for<4>([&](auto j) __attribute__((always_inline)) { sv[j()] = fma(acc[S][j()], xl[I], sv[j()]); });
                  }
                });
                static_for<4>([&](auto j) __attribute__((always_inline)) { sv[j()] = row16_total(sv[j()]); });
                if (l15 == 0) {
                  double* pw = part + ((J - 1) & 1) * (16 * W);
                  static_for<4>([&](auto j) __attribute__((always_inline)) { pw[WAVE * 16 + l4 + 4 * j] = sv[j()]; });
                }
              }
            }
          }
          __syncthreads();                                            // x_J (and the far terms of column J - 1) are in LDS
          if constexpr (WAVE != CW) xl[J] = out[16 * J + l15];
        }
      });
    };
    // ---- forward substitution U' y = rho in place in tvec (refinement only; the first right-hand side rides along
    //      with the factorisation), left to right:
    //   (1) every wave: s[i] = sum over its tiles (J, K), K < J, of U(K,J)[.][i]' y_K  (4 multiply-adds per tile, one
    //       reduction over the four lane rows) -> part[wave][i]
    //   (2) panel wave: y_J = M_J (rho_J - sum of the parts)
    auto forward_substitute = [&]() __attribute__((always_inline)) {
      static_for<NT>([&](auto JJ) __attribute__((always_inline)) {
        constexpr int J = JJ;
        if (16 * J < rE) {
          if constexpr (TM::has_row(WAVE, J)) {
            double s = 0.0;
            static_for<J>([&](auto K) __attribute__((always_inline)) {
              if constexpr (TM::wave(J, K) == WAVE) {
                constexpr int S = TM::slot(J, K);
                static_for<4>([&](auto j) __attribute__((always_inline)) {
                  s = fma(acc[S][j()], tvec[16 * K + l4 + 4 * j()], s);
                });
              }
            });
            s = rows4_total(s);
            if (l4 == 0) part[WAVE * 16 + l15] = s;
          }
          __syncthreads();
          if constexpr (WAVE == 0) {
            constexpr int SD = TM::slot(J, J);
            double v = tvec[16 * J + l15];
            static_for<W>([&](auto w) __attribute__((always_inline)) {
              if constexpr (TM::has_row(w, J)) v -= part[w * 16 + l15];
            });
            v = (16 * J + l15 < rE) ? v : 0.0;                        // rhs / padding columns of the last tile carry no unknown
            d4 q;
            static_for<4>([&](auto j) __attribute__((always_inline)) { q[j()] = row16_total(acc[SD][j()] * v); });
            if (l15 == 0) {
              static_for<4>([&](auto j) __attribute__((always_inline)) { tvec[16 * J + l4 + 4 * j] = q[j()]; });
            }
          }
          __syncthreads();
        }
      });
    };
    back_substitute(beta, tvec, rE);

    // ---- iterative refinement: residual with EXACT products with the implicit Hankel matrix,
    //        rho = t - ( H (H' beta) + lam D beta ),
    //      solved against the factor at hand.  The Gram route squares cond(H); the residual is evaluated through H
    //      itself (two products from the trajectory in LDS), so the corrected beta is accurate to cond(H)-level like a
    //      full-space solve.  (AUTO mode launches this variant on the instances the plain variant's residual check flagged.)
    if constexpr (REF) {
    if (P.refine != 0 && P.lam != 0.0) {      // nominal scheme: z = t does not depend on beta, nothing to refine
      bool go = true;
      double prev = 1e300;
      for (int pass = 0; go && pass < P.refine_max; ++pass) {
        // alpha = H' beta in chunks that fit the (now free) panel buffer, z += H[:, chunk] alpha[chunk]
        // Trajectories beyond the LDS (P.stage_xs == 0, hankel_matrix.py:39-51 takes any N >= L): the xs region is a WINDOW of
        // P.xs_len doubles; every chunk of columns stages its rows [i0, i0 + nc + Ln - 1) of the trajectory from global memory
        // first (the refining variant is the rare path: plain loads, one division per entry)
        const int c = P.c;
        constexpr int CHF = 16 * RSB;
        const bool windowed = P.stage_xs == 0;
        const int CHW = P.xs_len / nch - (P.Ln - 1);
        const int CH = (windowed && CHW < CHF) ? CHW : CHF;
        double zacc[NE];
        static_for<NE>([&](auto e) __attribute__((always_inline)) { zacc[e()] = 0.0; });
        for (int i0 = 0; i0 < c; i0 += CH) {
          const int nc = (c - i0) < CH ? (c - i0) : CH;
          if (windowed) {
            const int nrow = nc + P.Ln - 1, nu = nrow * P.m, ny = nrow * P.p;
            const double* us = ud_b + (long long)i0 * P.m;
            const double* ys = yd_b + (long long)i0 * P.p;
            for (int i = tid; i < nu; i += NTHR) { const int t = i / P.m, ch = i - t * P.m; xs[t * nch + ch] = us[i]; }
            for (int i = tid; i < ny; i += NTHR) { const int t = i / P.p, ch = i - t * P.p; xs[t * nch + P.m + ch] = ys[i]; }
            __syncthreads();
          }
          const double* xw = windowed ? xs : xs + (long long)i0 * nch;       // column 0 of the chunk
          for (int i = tid; i < nc; i += NTHR) {
            const double* xp = xw + (long long)i * nch;
            double s = 0.0;
            for (int rho = 0; rho < r; ++rho) s = fma(xp[rho], beta[rho], s);
            PB[i] = s;
          }
          __syncthreads();
          static_for<NE>([&](auto e) __attribute__((always_inline)) {
            const int rho = tid + e * NTHR;
            if (rho < r) {
              const double* xq = xw + rho;
              double s = zacc[e()];
              for (int ii = 0; ii < nc; ++ii) s = fma(xq[(long long)ii * nch], PB[ii], s);
              zacc[e()] = s;
            }
          });
          __syncthreads();
        }
        static_for<NE>([&](auto e) __attribute__((always_inline)) {
          const int rho = tid + e * NTHR;
          if (rho < RP) {
            double rv = 0.0;
            if (rho < r) {
              const int s_act = act[rho];
              const double D = s_act ? cD1[rho] : cD0[rho];
              const double t = cT[rho] + s_act * P.bound;
              double db = D * beta[rho];
              if (P.dense_w) {
                const double* dr = P.dmat + (long long)rho * RP;
                for (int j = 0; j < r; ++j) db = fma(dr[j], beta[j], db);
              }
              rv = t - zacc[e()] - P.lam * db;
            }
            tvec[rho] = rv;
          }
        });
        __syncthreads();
        forward_substitute();
        back_substitute(dvec, tvec, rE);                                // delta (dvec is free: D lives in registers)
        double dmx = 0.0, bmx = 0.0;
        static_for<NE>([&](auto e) __attribute__((always_inline)) {
          const int rho = tid + e * NTHR;
          if (rho < r) {
            const double dl = dvec[rho];
            const double bn = beta[rho] + dl;
            beta[rho] = bn;
            dmx = fmax(dmx, fabs(dl)); bmx = fmax(bmx, fabs(bn));
          }
        });
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { dmx = fmax(dmx, __shfl_xor(dmx, off, 64)); bmx = fmax(bmx, __shfl_xor(bmx, off, 64)); }
        __syncthreads();
        if (lane == 0) { red[tid >> 6] = dmx; red[8 + (tid >> 6)] = bmx; }
        __syncthreads();
        double dall = 0.0, ball = 0.0;
        for (int w = 0; w < W; ++w) { dall = fmax(dall, red[w]); ball = fmax(ball, red[8 + w]); }
        const double rel = dall / fmax(ball, 1e-300);
        // stop when the correction is at rounding level, or no longer shrinking (or not finite)
        go = (rel > 1e-13) && (rel < 0.25 * prev);
        prev = rel;
        __syncthreads();
      }
    }
    }  // REF
    stamp();   // 5
    stamp();   // 6

    // ---- slack box: primal-dual active-set update --------------------------------
    // (writes the new active set; true when it changed and another solve is due)
    auto active_set_test = [&]() __attribute__((always_inline)) -> bool {
      static_for<NE>([&](auto e) __attribute__((always_inline)) {
        const int rho = tid + e * NTHR;
        if (rho < r && (cK[rho] == K_WPRED || cK[rho] == K_WTERM)) {   // sigma[n*p:], controller.py:659
          const double sh = P.sig_scale * beta[rho];
          const int ns = (sh > P.bound) ? 1 : (sh < -P.bound) ? -1 : 0;
          if (ns != act[rho]) { act[rho] = ns; flags[1] = 1; }
        }
      });
      __syncthreads();
      bool ag = (flags[1] != 0) && (flags[0] == 0);
      if (ag && iter >= P.max_iter) { ag = false; status = 4; }
      return ag;
    };
    // ---- the same solve WITHOUT a new factorisation (CVX).  An active-set iteration switches D_ii (and the target) of the
    // few components S whose slack reached its bound -- 1 or 2 of 136 on the benchmark data, never more than 2 in 640
    // instances of configs[1] / configs[3] (tools/convex_update_study.py) -- while until round 4 every iteration formed and
    // factored the whole matrix again.  With K0 = L L' the factor of the EMPTY active set, E = [e_s], s in S, k = |S| <= KC:
    //     K(act) = K0 - E diag(d) E',  d_s = lam (D0_s - D1_s) > 0        t(act) = t0 + bound E sgn
    //     beta = K(act)^-1 t(act) = L^-T ( y' + W (diag(1/d) - W'W)^-1 W'y' ),   W = L^-1 E,   y' = y + bound W sgn
    // (Woodbury; diag(1/d) - W'W is positive definite because K(act) is, and well conditioned: 1/d = lamb_sigma / lam = 1e4
    // against |W'W| <= 1 / (lam D0) ~ 30).  W: ONE forward substitution of a 16-column block on the matrix pipe, tile row by
    // tile row -- the accumulator tiles U(K,J) and the tile of right-hand sides are both contracted over their first index,
    // which is exactly what an accumulator register offers as an MFMA operand (A = acc[j], B = W_K[j]); then M_J from LDS.
    // Rows above the first switched component are zero and skipped.  W (r x KC, compact) lives in the lag-block table's LDS,
    // the per-wave partial products in PT2 / LT; the M tiles of the last back substitution are still in the panel buffer.
    // Then the k x k system by every thread redundantly, and ONE back substitution.  Anything unusual (k > KC, a component
    // whose D does not change, a non-positive pivot) returns false and the caller factors the system of the new active set
    // as before.  Results: the same active sets and iteration counts as re-factoring (the CPU study: 0 mismatches, solutions
    // 5e-13 apart); tests/test_gpu_round5.py.
    auto rank_update = [&]() __attribute__((always_inline)) -> bool {
      constexpr int KS = 4;                                          // size of the k x k system (padded with the identity)
      // columns of W kept: four where two sets of per-wave partial products fit PT2 + LT (up to four waves), else two
      constexpr int KC = (2 * W * 16 * 4 <= LD::PT2_LEN + LD::LT_LEN) ? 4 : 2;
      int rE_u = P.rE;                                               // opaque per call (see the top of the iteration loop): the
      asm volatile("" : "+s"(rE_u));                                 // wave-uniform tests below stay where they are written
      const int rE = rE_u;
      double* Wc = ctab;                                             // Wc[row * KC + j]
      double* wpart = PT2;                                           // [row parity][wave][16][KC]
      double* sS = part;                                             // [KS][KS] W'W, [16 + x] W'y
      static_assert(2 * W * 16 * KC <= LD::PT2_LEN + LD::LT_LEN, "two sets of per-wave partial products fit PT2 + LT");
      static_assert(RP * KC <= LD::CTAB_LEN, "W fits the lag-block table");
      const long long tu0 = now();
      // the switched components in ascending order (deterministic: column order decides the summation order)
      if constexpr (WAVE == 0) {
        int base = 0;
        for (int r0 = 0; r0 < RP; r0 += 64) {
          const int rho = r0 + lane;
          const int a = (rho < RP) ? act[rho] : 0;
          const unsigned long long mk = __ballot(a != 0);
          const int slot = base + __popcll(mk & ((1ull << lane) - 1ull));
          if (a != 0 && slot < KC) flags[4 + slot] = rho;
          base += __popcll(mk);
        }
        if (lane == 0) flags[2] = base;
      }
      __syncthreads();
      const int k = flags[2];
      if (k > KC || k < 1) return false;                             // (workgroup-uniform)
      if (tid == 0) flags[1] = 0;                                    // (everybody has read it: barrier above)
      ctab_ok = false;
      const int s0 = flags[4];
      const int col = l15 & 3;                                       // column of W this lane works on (replicated over the four quads)
      const int sj = (col == 0) ? s0 : (col == 1) ? flags[5] : (col == 2) ? flags[6] : flags[7];
      int J0 = s0 >> 4;                                              // first tile row with a nonzero row of W
      J0 = __builtin_amdgcn_readfirstlane(J0);
      // One workgroup barrier per tile row.  W_J = M_J (E_J - sum_{K<J} U(K,J)' W_K): only the term K = J - 1 needs the tile row
      // just finished, and the whole first sub-diagonal of tiles belongs to ONE wave (TileMap2 deals whole diagonals) -- the
      // chain wave CW.  It keeps W_{J-1} in registers, adds the other terms -- formed one row AHEAD by every wave from the rows
      // already in LDS, beside the chain -- and applies M_J itself; the panel wave (idle here: it owns the diagonal tiles only)
      // accumulates W'W and W'y one row behind.  (First version: partial products of row J by every wave -> barrier -> the
      // panel wave -> barrier: 28 K cycles per update on the benchmark against 15 K for the back substitution.)
      //
      // W has KC <= 4 columns: a v_mfma_f64_16x16x4 would spend 64 cycles of the matrix pipe on a product of which a quarter is
      // used.  The products run on v_mfma_f64_4x4x4 instead (four independent 4 x 4 x 4 blocks per instruction, ~16 cycles):
      //   U(K,J)' W_K: block b = rows 4 b .. 4 b + 3 of the result, chunk c of the contraction per instruction -- the A operand
      //     lane (l4 = k, l15 = 4 b + i) = U[4 c + k][4 b + i] IS accumulator register c, the B operand lane (l4 = k, l15 = 4 b + j)
      //     = W_K[4 c + k][j], the same in every block; the result P[4 b + i][j] comes back in lane (l4 = i, l15 = 4 b + j);
      //   M_J V: one instruction per four rows i0 .. i0 + 3, block b = chunk b of the contraction: B = V in exactly that result
      //     layout, A lane (l4 = k, l15 = 4 b + i) = M_J[i0 + i][4 b + k] from LDS; the four blocks' partial sums meet by two
      //     row rotations, after which EVERY quad of the row holds W_J[i0 + l4][j] -- the replicated form the next row's B operand
      //     and the store want.
      constexpr int CW = TM::wave(1, 0);
      auto noncrit = [](int w, int J) constexpr { for (int K = 0; K + 2 <= J; ++K) if (TM::wave(J, K) == w) return true; return false; };
      d4 sacc = d4{0.0, 0.0, 0.0, 0.0}, gacc = d4{0.0, 0.0, 0.0, 0.0};
      double Wrep[4] = {0.0, 0.0, 0.0, 0.0};                         // chain wave: W_{J-1}[4 c + l4][col], c = 0 .. 3
      const double* Mb = PB;
      const int prow = 4 * (l15 >> 2) + l4;                          // row of the tile this lane holds in the 4x4x4 result layout
      static_for<NT + 1>([&](auto JJ) __attribute__((always_inline)) {
        constexpr int J = JJ;
        asm volatile("" : "+s"(J0));                                 // (opaque per tile row: no hoisted lane masks)
        const bool do_chain = J < NT && J >= J0 && 16 * J < rE;      // workgroup-uniform
        const bool do_gram = J >= 1 && J - 1 >= J0 && 16 * (J - 1) < rE;
        if constexpr (J < NT) {
          if (do_chain) {
            if constexpr (WAVE == CW) {
              double pc = 0.0;
              if constexpr (J >= 1) {
                if (J > J0) {                                        // the term that waits for the row before
                  constexpr int S = TM::slot(J, J - 1);
                  static_for<4>([&](auto c) __attribute__((always_inline)) {
                    pc = __builtin_amdgcn_mfma_f64_4x4x4f64(acc[S][c()], Wrep[c()], pc, 0, 0, 0);
                  });
                }
              }
              double v = (col < k && sj == 16 * J + prow) ? 1.0 : 0.0;
              if constexpr (J >= 2) {
                if (J > J0 + 1) {                                    // the terms K <= J - 2, formed during the step before
                  static_for<W>([&](auto w) __attribute__((always_inline)) {
                    if constexpr (noncrit(w(), J)) v -= wpart[(((J & 1) * W + w()) * 16 + prow) * KC + (col & (KC - 1))];
                  });
                }
              }
              v = (col < KC && 16 * J + prow < rE) ? v - pc : 0.0;
              static_for<4>([&](auto g) __attribute__((always_inline)) {       // rows 4 g .. 4 g + 3 of W_J
                const double am = Mb[J * 256 + (4 * g() + col) * 16 + 4 * (l15 >> 2) + l4];
                double w4 = __builtin_amdgcn_mfma_f64_4x4x4f64(am, v, 0.0, 0, 0, 0);
                w4 += dpp_mov_f64<0x124>(w4);                        // row_ror:4
                w4 += dpp_mov_f64<0x128>(w4);                        // row_ror:8: every quad holds the sum over the four blocks
                Wrep[g()] = w4;
              });
              if (l15 < KC) {
                static_for<4>([&](auto g) __attribute__((always_inline)) { Wc[(16 * J + 4 * g() + l4) * KC + l15] = Wrep[g()]; });
              }
            }
          }
        }
        // one row ahead: the terms K <= J - 1 of tile row J + 1 (W_K is in LDS since the barrier of step K)
        if constexpr (J + 1 < NT && J >= 1) {
          if (J >= J0 + 1 && 16 * (J + 1) < rE) {
            if constexpr (noncrit(WAVE, J + 1)) {
              double pacc = 0.0;
              static_for<J>([&](auto K) __attribute__((always_inline)) {
                if constexpr (TM::wave(J + 1, K) == WAVE) {
                  if (K >= J0) {
                    constexpr int S = TM::slot(J + 1, K);
                    static_for<4>([&](auto c) __attribute__((always_inline)) {
                      const double wv = Wc[(16 * K + 4 * c() + l4) * KC + (col & (KC - 1))];
                      pacc = __builtin_amdgcn_mfma_f64_4x4x4f64(acc[S][c()], (col < KC) ? wv : 0.0, pacc, 0, 0, 0);
                    });
                  }
                }
              });
              if (col < KC) wpart[((((J + 1) & 1) * W + WAVE) * 16 + prow) * KC + col] = pacc;
            }
          }
        }
        // one row behind: W'W and W'y of tile row J - 1
        if constexpr (WAVE == 0 && J >= 1) {
          if (do_gram) {
            static_for<4>([&](auto q) __attribute__((always_inline)) {
              const double wv = Wc[(16 * (J - 1) + 4 * q() + l4) * KC + (l15 & (KC - 1))];
              const double wq = (l15 < KC) ? wv : 0.0;
              const double yv = tvec[16 * (J - 1) + 4 * q() + l4];
              sacc = __builtin_amdgcn_mfma_f64_16x16x4f64(wq, wq, sacc, 0, 0, 0);
              gacc = __builtin_amdgcn_mfma_f64_16x16x4f64(wq, (l15 == 0) ? yv : 0.0, gacc, 0, 0, 0);
            });
          }
        }
        if (do_chain) __syncthreads();                               // W_J (and the terms of row J + 1) are in LDS
      });
      const long long tu1 = now();
      // ---- the k x k system, by the panel wave (every lane redundantly; the other waves hold up to 13 accumulator tiles and
      //      have no registers to spare for it):  (diag(1/d) - W'W) cv = W'y',  y' = y + bound W sgn;  coefficients
      //      ev = bound sgn + cv of  y' + W cv = y + W ev  go to sS[24 ..], flags[3] = 1 when a pivot failed
      if constexpr (WAVE == 0) {
        if (l15 < KS) sS[l4 * KS + l15] = sacc[0];                   // (W'W)[x = l4][y = l15]
        if (l15 == 0) sS[16 + l4] = gacc[0];                         // (W'y)[x = l4]
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");       // in-wave hand-off through LDS (see factor_begin)
        if (k <= 2) {
          // One or two switched components -- every instance of the benchmark configurations (tools/convex_update_study.py): the
          // 2 x 2 system in closed form, every lane the same arithmetic (the general path below took 3.6 K cycles of a 40 K update)
          const int sr0 = flags[4], sr1 = (k == 2) ? flags[5] : sr0;
          const double d0 = P.lam * (cD0[sr0] - cD1[sr0]), d1 = P.lam * (cD0[sr1] - cD1[sr1]);
          const double sg0 = (double)act[sr0] * P.bound, sg1 = (k == 2) ? (double)act[sr1] * P.bound : 0.0;
          const double g00 = sS[0], g01 = (k == 2) ? sS[1] : 0.0, g11 = (k == 2) ? sS[KS + 1] : 0.0;
          const double h0 = sS[16] + g00 * sg0 + g01 * sg1, h1 = (k == 2) ? sS[17] + g01 * sg0 + g11 * sg1 : 0.0;
          const double a = 1.0 / d0 - g00, cc = (k == 2) ? 1.0 / d1 - g11 : 1.0;
          const double det = a * cc - g01 * g01;
          const bool ok2 = (d0 > 1e-300) && (d1 > 1e-300) && (a > 0.0) && (det > 0.0);
          const double c0 = (cc * h0 + g01 * h1) / det, c1 = (k == 2) ? (a * h1 + g01 * h0) / det : 0.0;
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
          if (lane < KS) sS[24 + lane] = (lane == 0) ? c0 + sg0 : (lane == 1 && k == 2) ? c1 + sg1 : 0.0;
          if (lane == 0) flags[3] = ok2 ? 0 : 1;
        } else {
        // lane 4 x + y (< 16) holds entry (x, y) of Sm = diag(1/d) - W'W (identity on the padding) in ONE register; the
        // Cholesky runs across the lanes (three shuffles per column) -- a register-resident 4 x 4 factorisation in every lane
        // took 56 VGPRs next to the nine accumulator tiles and pushed three of those into scratch for the whole kernel
        const int sx = (lane >> 2) & 3, sy = lane & 3;
        const bool on_x = sx < k, on_y = sy < k;
        const int srx = on_x ? flags[4 + sx] : 0, sry = on_y ? flags[4 + sy] : 0;
        const double ddx = P.lam * (cD0[srx] - cD1[srx]);
        bool ok = !on_x || ddx > 1e-300;
        const double sgy = on_y ? (double)act[sry] * P.bound : 0.0;   // bound sgn of component y
        const double gxy = (on_x && on_y) ? sS[sx * KS + sy] : 0.0;   // (W'W)[x][y]
        double av = -gxy;
        if (sx == sy) av = on_x ? 1.0 / ddx - gxy : 1.0;
        double hp = gxy * sgy;                                         // h_x = (W'y)_x + sum_y (W'W)[x][y] bound sgn_y
        hp += __shfl_xor(hp, 1, 64); hp += __shfl_xor(hp, 2, 64);
        const double hx = hp + (on_x ? sS[16 + sx] : 0.0);
        static_for<KS>([&](auto jj) __attribute__((always_inline)) {
          constexpr int j = jj;
          const double pj = __shfl(av, 5 * j, 64);
          ok = ok && (pj > 0.0);
          const double il = 1.0 / sqrt(pj);
          const double axj = __shfl(av, 4 * sx + j, 64) * il, ayj = __shfl(av, 4 * sy + j, 64) * il;
          if (sx > j && sy > j) av = fma(-axj, ayj, av);
          else if (sy == j && sx > j) av = axj;
          else if (sx == j && sy == j) av = il;                       // (the reciprocal pivot is kept)
        });
        ok = __ballot(!ok && lane < 16) == 0ull;
        if (lane < 16) sS[lane] = av;                                 // L (lower), reciprocal pivots on the diagonal
        if (lane < 16 && sy == 0) sS[20 + sx] = hx;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        if (lane < KS) {                                              // forward and backward substitution, then ev = bound sgn + cv
          double c0 = sS[20] * sS[0];
          double c1 = (sS[21] - sS[4] * c0) * sS[5];
          double c2 = (sS[22] - sS[8] * c0 - sS[9] * c1) * sS[10];
          double c3 = (sS[23] - sS[12] * c0 - sS[13] * c1 - sS[14] * c2) * sS[15];
          c3 = c3 * sS[15];
          c2 = (c2 - sS[14] * c3) * sS[10];
          c1 = (c1 - sS[9] * c2 - sS[13] * c3) * sS[5];
          c0 = (c0 - sS[4] * c1 - sS[8] * c2 - sS[12] * c3) * sS[0];
          const double cv = (lane == 0) ? c0 : (lane == 1) ? c1 : (lane == 2) ? c2 : c3;
          const int srl = (lane < k) ? flags[4 + lane] : 0;
          sS[24 + lane] = cv + ((lane < k) ? (double)act[srl] * P.bound : 0.0);
        }
        if (lane == 0) flags[3] = ok ? 0 : 1;
        }
      }
      __syncthreads();
      const long long tu2 = now();
      if (flags[3] != 0) return false;                               // (workgroup-uniform)
      const double e0 = sS[24], e1 = sS[25], e2 = sS[26], e3 = sS[27];
      static_assert(KS == 4 && 28 <= LD::PART_LEN, "four coefficients behind the k x k system in part[]");
      static_for<NE>([&](auto e) __attribute__((always_inline)) {
        const int rho = tid + e * NTHR;
        if (rho < RP) {
          double v = 0.0;
          if (rho < rE) {
            v = tvec[rho];
            if (rho >= 16 * J0) {
              const d2 w01 = *reinterpret_cast<const d2*>(Wc + rho * KC);
              v = fma(w01[0], e0, v); v = fma(w01[1], e1, v);
              if constexpr (KC == 4) {
                const d2 w23 = *reinterpret_cast<const d2*>(Wc + rho * KC + 2);
                v = fma(w23[0], e2, v); v = fma(w23[1], e3, v);
              }
            }
          }
          dvec[rho] = v;
        }
      });
      __syncthreads();
      back_substitute(beta, dvec, rE);
      if (timing && threadIdx.x == 0) {                              // diagnostics: forward block substitution | k x k system | back substitution
        stamps[9] += (unsigned long long)(tu1 - tu0); stamps[10] += (unsigned long long)(tu2 - tu1); stamps[11] += (unsigned long long)(now() - tu2);
      }
      return true;
    };
    bool again = false;
    if (P.convex) {
      for (;;) {
        again = active_set_test();
        if constexpr (!CVX) break;
        if (!again || iter != 1 + nupd) break;                        // (the factor at hand is not the empty active set's)
        if (!rank_update()) break;
        ++iter; ++nupd;
      }
    }
    if (!again) break;
  }
  if (flags[0] != 0) status = 4;
  const int lane = tid & 63;

  // ---- outputs: z = t - lam*D*beta; cost = control cost + lam*beta'z + lamb_sigma*|sigma|^2 ----------------------
  double partc = 0.0;
  double bmx = 0.0, tmx = 0.0;                      // max |beta|, max |t|: the residual bound of AUTO refinement
  bool finite = true;
  static_for<NE>([&](auto e) __attribute__((always_inline)) {
    const int rho = tid + e * NTHR;
    if (rho < r) {
      const int s_act = act[rho];
      const double b = beta[rho];
      const double D = s_act ? cD1[rho] : cD0[rho];
      const double t = cT[rho] + s_act * P.bound;
      bmx = fmax(bmx, fabs(b)); tmx = fmax(tmx, fabs(t));
      double z = t - P.lam * D * b;
      if (P.dense_w) {                              // z = t - lam (W^-1 beta): one row of the dense matrix
        const double* dr = P.dmat + (long long)rho * RP;
        double sdb = 0.0;
        for (int j = 0; j < r; ++j) sdb += dr[j] * beta[j];
        z = t - P.lam * (D * b + sdb);
      }
      const double wq = P.tabd[3 * RP + rho];
      const double tb = P.tabd[2 * RP + rho];
      const int oidx = P.tabi[2 * RP + rho];
      finite = finite && (fabs(b) < 1e300);
      double contrib = P.lam * b * z;
      const int kind = cK[rho];
      if (P.dense_w && (kind == K_UFREE || kind == K_YFREE || kind == K_WPRED)) {
        // (z - t)' W (z - t) summed over the weighted components equals -lam * beta' (z - t), t the (shifted) target;
        // a sigma held at its bound adds lamb_sigma * bound^2 (W^-1 = Q_ff^-1 + 1/lamb_sigma on the INACTIVE components)
        contrib -= P.lam * b * (z - t);
        if (s_act != 0) contrib += P.box_cost;
      } else
      if (kind == K_UFREE || kind == K_YFREE) { const double dlt = z - tb; contrib += wq * dlt * dlt; }
      else if (kind == K_WINT) { const double sg = z - cT[rho]; contrib += P.lamb_sigma * sg * sg; }
      else if (kind == K_WTERM) { const double sg = z - tb; contrib += P.lamb_sigma * sg * sg; }
      else if (kind == K_WPRED) {
        const double sg = (s_act != 0) ? s_act * P.bound : -P.lam * b / P.lamb_sigma;
        const double dlt = z - sg - tb;
        contrib += wq * dlt * dlt + P.lamb_sigma * sg * sg;
      }
      partc += contrib;
      if (oidx >= 0) u_opt[oidx] = z;               // ubar[n*m:], controller.py:799-805
      if (beta_ws) beta_ws[rho] = b;
      if (act_ws) act_ws[rho] = (signed char)s_act;
    }
  });
  partc = rows4_total(row16_total(partc));
  const unsigned long long okmask = __ballot(finite);
  const bool auto_check = !REF && refine_flag != nullptr && P.lam != 0.0;      // kernel-uniform
  if (auto_check) {
    bmx = wave_max(bmx); tmx = wave_max(tmx);
    if constexpr (WAVE == 0) kmax = wave_max(kmax);
  }
  if (lane == 0) {
    red[tid >> 6] = partc; red[16 + (tid >> 6)] = (okmask == ~0ull) ? 0.0 : 1.0;
    if (auto_check) { red[8 + (tid >> 6)] = bmx; red[24 + (tid >> 6)] = tmx; if (WAVE == 0) red[32] = kmax; }
  }
  __syncthreads();
  if (tid == 0) {
    double tot = 0.0, bad = 0.0;
    for (int w = 0; w < W; ++w) { tot += red[w]; bad += red[16 + w]; }
    if (bad != 0.0 || !(fabs(tot) < 1e300)) status = 4;
    *cost_out = tot;
    *status_out = status;
    if (iters_out) *iters_out = iter;
  }

  // ---- AUTO refinement trigger (plain variant only): a-posteriori check of the solve just finished ------------------
  // Two stages.  (i) The a-priori residual bound of the Gram route,  q = eps max G_kk |beta|_inf / |t|_inf  (the solve's
  // residual is E beta with |E| ~ eps |G|: rounding of the Gram sums and of the factorisation), from maxima the output
  // stage collected on its way.  On the calibration sweep (profiles/r03_refine_calib.log: 72 robust random-plant cases + the
  // benchmark batch) the true residual is 0.1 .. 10 times q wherever q <= 1e-12 and at most ~50 times q anywhere, so q is
  // only used to DISMISS: an instance with q <= threshold / 20 is far below the threshold -- the benchmark's instances
  // (q <= 9e-13, residual <= 4e-12): nothing more to do.  (ii) Everything else is CHECKED, not guessed:
  //   res = | t - ( H (H' beta) + lam D beta ) |_inf / | t |_inf  with EXACT products with the implicit Hankel matrix
  // (a conditioning estimate read off the pivots does not tell the Gram route's error apart from harmless
  // ill-conditioning: tools/auto_flag_calib_cpu.py, tools/refine_calib.py), and res > threshold flags the instance.
  // 2 r c multiply-adds on the vector pipe, straight from the trajectory in LDS; with four channels (the benchmark) both
  // products walk the trajectory in quads x[4 j .. 4 j + 3] (one time step) that slide through registers:
  //   alpha_i      = sum_f < Q(i + f), beta[4 f ..] >           lane = (4 columns, half of the f range):  16 FMAs per 4 loads
  //   g[4 f + ch]  = sum_i Q(i + f)[ch] alpha_i                  lane = (4 time offsets, a chunk of i):    16 FMAs per 3 loads
  // alpha and the per-chunk partial sums live in the LDS between pt2 and xs, all free behind the factorisation.
  if constexpr (!REF) {
    if (auto_check) {
      double kall = red[32], ball = 0.0, tall = 0.0;
      for (int w = 0; w < W; ++w) { ball = fmax(ball, red[8 + w]); tall = fmax(tall, red[24 + w]); }
      const double q = 1.1102230246251565e-16 * kall * ball / fmax(tall, 1e-300);
      int flag = 0;
      double shown = q;
      if (!(q <= 0.05 * P.refine_res)) {                                // workgroup-uniform (NaN: checked, and flagged there)
        if (!P.res_fits) flag = P.epoch;                                // the check cannot be staged for this shape: refine
        else {
        const int c = P.c;
          const int xlast = P.xs_len - 1;
          double* al = sm + LD::SCR;                                      // alpha[i], i < c
          double* pp = al + ((c + 1) & ~1);                               // nch == 4: partial sums [i-chunk][16 FB]
          double g[NE];
          static_for<NE>([&](auto e) __attribute__((always_inline)) { g[e()] = 0.0; });
          __syncthreads();                                                // beta complete, PB / LT / ctab no longer read
          if (nch == 4) {
            const d2* xq = reinterpret_cast<const d2*>(xs);               // Q(j) = (xq[2 j], xq[2 j + 1])
            const d2* bq = reinterpret_cast<const d2*>(beta);
            const int jmax = (xlast - 3) >> 2;                            // last quad inside the region
            auto quad = [&](int j, d2& lo2, d2& hi2) __attribute__((always_inline)) {
              j = j < jmax ? j : jmax;
              lo2 = xq[2 * j]; hi2 = xq[2 * j + 1];
            };
            // (1) alpha
            {
              const int Ln = r >> 2;
              const int Lh = (Ln + 1) >> 1;
              const int ntask = 2 * ((c + 3) >> 2);
              for (int t0 = 0; t0 < ntask; t0 += NTHR) {
                const int task = t0 + tid;
                const int i0 = 4 * (task >> 1), h = task & 1;
                const int fa = h ? Lh : 0, fb = h ? Ln : Lh;
                double acc0 = 0.0, acc1 = 0.0, acc2 = 0.0, acc3 = 0.0;
                d2 q0l, q0h, q1l, q1h, q2l, q2h, q3l, q3h;
                quad(i0 + fa, q0l, q0h); quad(i0 + fa + 1, q1l, q1h); quad(i0 + fa + 2, q2l, q2h);
                auto stepA = [&](int f, const d2& al_, const d2& ah_, const d2& bl_, const d2& bh_, const d2& cl_, const d2& ch_,
                                 d2& nl_, d2& nh_) __attribute__((always_inline)) {
                  quad(i0 + f + 3, nl_, nh_);
                  const d2 b0 = bq[2 * f], b1 = bq[2 * f + 1];
                  acc0 = fma(al_[0], b0[0], acc0); acc0 = fma(al_[1], b0[1], acc0); acc0 = fma(ah_[0], b1[0], acc0); acc0 = fma(ah_[1], b1[1], acc0);
                  acc1 = fma(bl_[0], b0[0], acc1); acc1 = fma(bl_[1], b0[1], acc1); acc1 = fma(bh_[0], b1[0], acc1); acc1 = fma(bh_[1], b1[1], acc1);
                  acc2 = fma(cl_[0], b0[0], acc2); acc2 = fma(cl_[1], b0[1], acc2); acc2 = fma(ch_[0], b1[0], acc2); acc2 = fma(ch_[1], b1[1], acc2);
                  acc3 = fma(nl_[0], b0[0], acc3); acc3 = fma(nl_[1], b0[1], acc3); acc3 = fma(nh_[0], b1[0], acc3); acc3 = fma(nh_[1], b1[1], acc3);
                };
                int f = fa;
#pragma nounroll
                for (; f + 4 <= fb; f += 4) {                             // the window rotates through the four register quads
                  stepA(f, q0l, q0h, q1l, q1h, q2l, q2h, q3l, q3h);
                  stepA(f + 1, q1l, q1h, q2l, q2h, q3l, q3h, q0l, q0h);
                  stepA(f + 2, q2l, q2h, q3l, q3h, q0l, q0h, q1l, q1h);
                  stepA(f + 3, q3l, q3h, q0l, q0h, q1l, q1h, q2l, q2h);
                }
                if (f < fb) { stepA(f, q0l, q0h, q1l, q1h, q2l, q2h, q3l, q3h); ++f;
                  if (f < fb) { stepA(f, q1l, q1h, q2l, q2h, q3l, q3h, q0l, q0h); ++f;
                    if (f < fb) { stepA(f, q2l, q2h, q3l, q3h, q0l, q0h, q1l, q1h); } } }
                // the two halves of the f range sit in neighbouring lanes
                acc0 += __shfl_xor(acc0, 1, 64); acc1 += __shfl_xor(acc1, 1, 64);
                acc2 += __shfl_xor(acc2, 1, 64); acc3 += __shfl_xor(acc3, 1, 64);
                if (h == 0 && task < ntask) {
                  if (i0 < c) al[i0] = acc0;
                  if (i0 + 1 < c) al[i0 + 1] = acc1;
                  if (i0 + 2 < c) al[i0 + 2] = acc2;
                  if (i0 + 3 < c) al[i0 + 3] = acc3;
                }
              }
            }
            __syncthreads();
            // (2) g: lane = (block of 4 time offsets, chunk of columns); the partial sums of the chunks meet in LDS
            {
              const int Ln = r >> 2;
              const int FB = (Ln + 3) >> 2;                               // blocks of time offsets
              int NC = NTHR / FB;                                         // column chunks: one task per thread at most,
              const int ncap = (LD::SCR_LEN - ((c + 1) & ~1)) / (16 * FB);  //   partial sums inside the scratch region (>= 1: P.res_fits)
              NC = NC < ncap ? NC : ncap;
              const int CL = (c + NC - 1) / NC;
              if (tid < FB * NC) {
                const int fbk = tid / NC, ic = tid - fbk * NC;
                const int f0 = 4 * fbk;
                const int ia = ic * CL;
                const int ib = (ia + CL) < c ? (ia + CL) : c;
                d2 a0l = d2{0.0, 0.0}, a0h = a0l, a1l = a0l, a1h = a0l, a2l = a0l, a2h = a0l, a3l = a0l, a3h = a0l;
                d2 q0l, q0h, q1l, q1h, q2l, q2h, q3l, q3h;
                quad(ia + f0, q0l, q0h); quad(ia + f0 + 1, q1l, q1h); quad(ia + f0 + 2, q2l, q2h);
                auto stepB = [&](int i, const d2& al_, const d2& ah_, const d2& bl_, const d2& bh_, const d2& cl_, const d2& ch_,
                                 d2& nl_, d2& nh_) __attribute__((always_inline)) {
                  quad(i + f0 + 3, nl_, nh_);
                  const double av = al[i];
                  a0l[0] = fma(al_[0], av, a0l[0]); a0l[1] = fma(al_[1], av, a0l[1]); a0h[0] = fma(ah_[0], av, a0h[0]); a0h[1] = fma(ah_[1], av, a0h[1]);
                  a1l[0] = fma(bl_[0], av, a1l[0]); a1l[1] = fma(bl_[1], av, a1l[1]); a1h[0] = fma(bh_[0], av, a1h[0]); a1h[1] = fma(bh_[1], av, a1h[1]);
                  a2l[0] = fma(cl_[0], av, a2l[0]); a2l[1] = fma(cl_[1], av, a2l[1]); a2h[0] = fma(ch_[0], av, a2h[0]); a2h[1] = fma(ch_[1], av, a2h[1]);
                  a3l[0] = fma(nl_[0], av, a3l[0]); a3l[1] = fma(nl_[1], av, a3l[1]); a3h[0] = fma(nh_[0], av, a3h[0]); a3h[1] = fma(nh_[1], av, a3h[1]);
                };
                int i = ia;
#pragma nounroll
                for (; i + 4 <= ib; i += 4) {
                  stepB(i, q0l, q0h, q1l, q1h, q2l, q2h, q3l, q3h);
                  stepB(i + 1, q1l, q1h, q2l, q2h, q3l, q3h, q0l, q0h);
                  stepB(i + 2, q2l, q2h, q3l, q3h, q0l, q0h, q1l, q1h);
                  stepB(i + 3, q3l, q3h, q0l, q0h, q1l, q1h, q2l, q2h);
                }
                if (i < ib) { stepB(i, q0l, q0h, q1l, q1h, q2l, q2h, q3l, q3h); ++i;
                  if (i < ib) { stepB(i, q1l, q1h, q2l, q2h, q3l, q3h, q0l, q0h); ++i;
                    if (i < ib) { stepB(i, q2l, q2h, q3l, q3h, q0l, q0h, q1l, q1h); } } }
                d2* po = reinterpret_cast<d2*>(pp + (ic * FB + fbk) * 16);
                po[0] = a0l; po[1] = a0h; po[2] = a1l; po[3] = a1h; po[4] = a2l; po[5] = a2h; po[6] = a3l; po[7] = a3h;
              }
                __syncthreads();
              static_for<NE>([&](auto e) __attribute__((always_inline)) {
                const int rho = tid + e * NTHR;
                if (rho < r) {
                  double sacc = 0.0;
                  for (int ic = 0; ic < NC; ++ic) sacc += pp[ic * FB * 16 + rho];
                  g[e()] = sacc;
                }
              });
            }
          } else {
            // any other channel count: one column per lane, then one row per lane (no operand reuse; these shapes are not
            // the benchmark's)
            for (int i = tid; i < c; i += NTHR) {
              const double* xp = xs + (long long)i * nch;
              double sacc = 0.0;
              for (int rho = 0; rho < r; ++rho) sacc = fma(xp[rho], beta[rho], sacc);
              al[i] = sacc;
            }
            __syncthreads();
            static_for<NE>([&](auto e) __attribute__((always_inline)) {
              const int rho = tid + e * NTHR;
              if (rho < r) {
                const double* xq1 = xs + rho;
                double sacc = 0.0;
                for (int i = 0; i < c; ++i) sacc = fma(xq1[(long long)i * nch], al[i], sacc);
                g[e()] = sacc;
              }
            });
          }
          double rmx = 0.0;
          static_for<NE>([&](auto e) __attribute__((always_inline)) {
            const int rho = tid + e * NTHR;
            if (rho < r) {
              const int s_act = act[rho];
              const double D = s_act ? cD1[rho] : cD0[rho];
              const double t = cT[rho] + s_act * P.bound;
              double db = D * beta[rho];
              if (P.dense_w) {
                const double* dr = P.dmat + (long long)rho * RP;
                for (int j = 0; j < r; ++j) db = fma(dr[j], beta[j], db);
              }
              const double rv = t - g[e()] - P.lam * db;
              rmx = fmax(rmx, (rv == rv) ? fabs(rv) : 1e300);           // NaN (failed pivot) counts as "large"
            }
          });
          rmx = wave_max(rmx);
          __syncthreads();                                              // red[] was read above
          if (lane == 0) red[8 + (tid >> 6)] = rmx;
          __syncthreads();
          double rall = 0.0;
          for (int w = 0; w < W; ++w) rall = fmax(rall, red[8 + w]);
          const double res = rall / fmax(tall, 1e-300);
          flag = !(res <= P.refine_res) ? P.epoch : 0;
          shown = res;
        }
      }
      if (tid == 0) {
        // epoch stamps instead of counts: nothing has to be cleared between launches (P.epoch grows by one per launch)
        *refine_flag = flag;
        if (flag && refine_count != nullptr) atomicMax(refine_count, P.epoch);
        if (stamps != nullptr) stamps[12] = (unsigned long long)__double_as_longlong(shown);      // diagnostics: res if checked, else q
      }
    }
  }
  if (tid == 0) {
    if (stamps) { stamps[14] = __builtin_amdgcn_s_memtime(); stamps[13] = __builtin_amdgcn_s_memrealtime(); }
  }
}

// Grid: the plain variant runs one workgroup per instance (grid = batch).  The refining variant may also be launched as a
// short persistent grid (nbatch > 0: workgroup g takes instances g, g + gridDim.x, ... and skips the ones `only` filters
// out): the AUTO refinement pass usually finds few or no flagged instances, and 768 workgroups scanning flags cost less
// than a batch-sized grid of workgroups that exit at once.
template <int NT, int W, bool REF, bool CVX = false>
__global__ __launch_bounds__(64 * W, DDMPC_MIN_WAVES(NT, W)) void ddmpc_cold_solve_kernel2(
    KParams P, const double* __restrict__ u_d, const double* __restrict__ y_d,
    const double* __restrict__ u_past, const double* __restrict__ y_past, double* __restrict__ u_opt,
    double* __restrict__ cost, int* __restrict__ status, int* __restrict__ iters,
    double* __restrict__ beta_ws, signed char* __restrict__ act_ws, unsigned long long* __restrict__ stamps,
    double* __restrict__ lfac, double* __restrict__ lfacT, int* __restrict__ refine_flag,
    const int* __restrict__ only, long long nbatch, int* __restrict__ refine_count) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int tid = threadIdx.x;
  constexpr int NTHR = 64 * W;
  if constexpr (REF) {
    // persistent refinement pass: nothing flagged by the plain kernel (the usual case) -> one load and out
    if (nbatch > 0 && refine_count != nullptr && *refine_count != P.epoch) return;
  }
  const long long bend = (REF && nbatch > 0) ? nbatch : (long long)blockIdx.x + 1;
  const long long bstep = (REF && nbatch > 0) ? (long long)gridDim.x : 1;
  for (long long b = blockIdx.x; b < bend; b += bstep) {
  if (only != nullptr && only[b] == 0) continue;
  unsigned long long* st = stamps ? stamps + b * 16 : nullptr;
  if (st && tid == 0) { st[0] = __builtin_amdgcn_s_memtime(); st[15] = __builtin_amdgcn_s_memrealtime(); }
  double* xs = sm + Lds2<NT, W>::xs;
  // component tables (L2) and past window: independent loads, issued BEFORE the trajectory staging so that the three
  // global round trips of the prologue (tables -> past-window gather -> trajectory) overlap instead of following each other
  constexpr int RPk = 16 * NT;
  constexpr int NEk = (RPk + NTHR - 1) / NTHR;
  double tD0[NEk], tD1[NEk], tTb[NEk];
  int tK[NEk], tP[NEk];
  static_for<NEk>([&](auto e) __attribute__((always_inline)) {
    const int rho = tid + e * NTHR;
    tD0[e()] = 0.0; tD1[e()] = 0.0; tTb[e()] = 0.0; tK[e()] = K_PAD; tP[e()] = -1;
    if (rho < RPk) {
      tD0[e()] = P.tabd[0 * RPk + rho];
      tD1[e()] = P.tabd[1 * RPk + rho];
      tTb[e()] = P.tabd[2 * RPk + rho];
      tK[e()] = P.tabi[0 * RPk + rho];
      tP[e()] = P.tabi[1 * RPk + rho];
    }
  });
  const int npast = P.npu + (P.npu / P.m) * P.p;
  double pwv[NEk];
  static_for<NEk>([&](auto e) __attribute__((always_inline)) {
    const int i = tid + e * NTHR;
    pwv[e()] = 0.0;
    if (i < npast) pwv[e()] = (i < P.npu) ? u_past[b * (long long)P.npu + i] : y_past[b * (long long)(npast - P.npu) + (i - P.npu)];
  });
  if (P.stage_xs) {                          // (kernel-uniform; 0: trajectory beyond the LDS, see KParams::stage_xs)
    const double* ud = u_d + b * (long long)P.N * P.m;
    const double* yd = y_d + b * (long long)P.N * P.p;
    if (P.m == 2 && P.p == 2) {              // 16-byte loads, one time step per lane
      const d2* u2 = reinterpret_cast<const d2*>(ud);
      const d2* y2 = reinterpret_cast<const d2*>(yd);
      d2* x2 = reinterpret_cast<d2*>(xs);
      for (int t = tid; t < P.N; t += NTHR) {
        const d2 uu = u2[t], yy = y2[t];
        x2[2 * t] = uu;
        x2[2 * t + 1] = yy;
      }
    } else {
      const int nu = P.N * P.m, ny = P.N * P.p;
      for (int i = tid; i < nu; i += NTHR) {
        const int t = i / P.m, ch = i - t * P.m;
        xs[t * P.nch + ch] = ud[i];
      }
      for (int i = tid; i < ny; i += NTHR) {
        const int t = i / P.p, ch = i - t * P.p;
        xs[t * P.nch + P.m + ch] = yd[i];
      }
    }
    for (int i = P.N * P.nch + tid; i < P.xs_len; i += NTHR) xs[i] = 0.0;
  }
  {
    double* cD0 = sm + Lds2<NT, W>::cd0; double* cD1 = sm + Lds2<NT, W>::cd1; double* cT = sm + Lds2<NT, W>::ct0;
    int* cK = reinterpret_cast<int*>(sm + Lds2<NT, W>::ckk); int* cP = reinterpret_cast<int*>(sm + Lds2<NT, W>::cpp);
    double* pastw = sm + Lds2<NT, W>::pastw;
    static_for<NEk>([&](auto e) __attribute__((always_inline)) {
      const int rho = tid + e * NTHR;
      if (rho < RPk) { cD0[rho] = tD0[e()]; cD1[rho] = tD1[e()]; cT[rho] = tTb[e()]; cK[rho] = tK[e()]; cP[rho] = tP[e()]; }
      if (rho < npast) pastw[rho] = pwv[e()];
    });
  }
  __syncthreads();       // the past window is visible to the threads that own its components
  const int n = P.npu / P.m;
  const double* up = u_past + b * (long long)P.npu;
  const double* yp = y_past + b * (long long)(n * P.p);
  double* uo = u_opt + b * (long long)((P.Ln - n) * P.m);
  double* bw = beta_ws ? beta_ws + b * (long long)P.rE : nullptr;
  signed char* aw = act_ws ? act_ws + b * (long long)P.rE : nullptr;
  int* it = iters ? iters + b : nullptr;
  double* lf = lfac ? lfac + b * (long long)(NT * (NT + 1) / 2 * 256) : nullptr;
  double* lft = lfacT ? lfacT + b * (long long)(NT * (NT + 1) / 2 * 256) : nullptr;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  static_for<W>([&](auto WV) {
    if (wave == WV)
      wave_body2<NT, W, WV, REF, CVX && !REF>(P, sm, up, yp, uo, cost + b, status + b, it, bw, aw, st, lf, lft,
                                 refine_flag ? refine_flag + b : nullptr, REF ? nullptr : refine_count,
                                 P.gpre ? P.gpre + b * P.gpre_stride : nullptr, u_d + b * (long long)P.N * P.m,
                                 y_d + b * (long long)P.N * P.p);
  });
  if constexpr (REF) __syncthreads();          // persistent grid: LDS is reused by the next instance
  }
}

}  // namespace ddmpc
