// ddmpc_kernels.hpp -- gfx950 (CDNA4) device code of the batched Data-Driven MPC QP engine.
//
// One workgroup solves one controller instance end to end and keeps the whole
// problem on chip (DESIGN.md sections 3-5):
//
//   trajectory (u_d,y_d) --coalesced--> LDS  (channel-interleaved "xflat")
//   G = H H'   : never materialises H.  H[rho][i] = xflat[i*nch + rho], so
//                * structured mode (nch == 4): G(k,l) depends on (k-l, l) only through a
//                  sliding-window recurrence; lag blocks C_d = G(d,0) by v_mfma_f64_4x4x4, then every
//                  wave walks its own tile diagonals in registers with 2 MFMAs per tile (no barriers);
//                * dense mode: fp64 MFMA (v_mfma_f64_16x16x4_f64) over the implicit operand.
//   K = G + lam*D, rhs t: diagonal / extra-row fix-up in registers (dense weights: lam*W^-1 from L2)
//   K = L L'   : right-looking blocked Cholesky on register tiles (MFMA accumulators),
//                4-wide panels through LDS, rank-4 trailing updates by MFMA, two barriers per step
//   L y = t    : free -- t rides along as an extra matrix row
//   L' beta = y: back substitution per 16-row tile row: diagonal tile in registers, the tiles left of it
//                by MFMA with the accumulator registers as operand
//   slack box  : primal-dual active set around the above (CONVEX only)
//   outputs    : optimal_u, cost, status (+ beta / active-set workspace)
//
// Reference formulation: direct_data_driven_mpc_controller.py:409-445 (variables),
// :506-677 (constraints), :679-722 (cost), :780-808 (extraction).
//
// Internal component order ("rho"): time-major with the nch = m+p channels of one
// time step adjacent, rho = k*nch + ch, k = 0..L+n-1, ch < m -> ubar, else ybar(+sigma).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

// minimum waves per SIMD requested from the register allocator (keeps MFMA in VGPR form)
#ifndef DDMPC_MIN_WAVES
#define DDMPC_MIN_WAVES(NT, W) ((W) <= 2 ? 2 : ((W) <= 4 ? 3 : ((NT) <= 9 ? 4 : 2)))
#endif

namespace ddmpc {

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

// component kinds (what a row of the reduced system stands for)
enum : int { K_UFIX = 0, K_UFREE = 1, K_WINT = 2, K_WTERM = 3, K_WPRED = 4,
             K_YFIX = 5, K_YFREE = 6, K_PAD = 7 };

// Kernel parameters.  Everything that depends only on the controller parameters
// (shared by the whole batch) is tabulated per component on the host
// (ddmpc_api.hip: build_component_table) -- see DESIGN.md 3.2:
//   tabd: [4][RP] doubles  D0 (inactive 1/w), D1 (bound active), tb (target), wq (cost weight)
//   tabi: [3][RP] ints     kind, pidx (index into [u_past; y_past] or -1), oidx (index into u_opt or -1)
struct KParams {
  int N, nch, m, p;
  int Ln;           // L + n
  int r;            // nch * Ln           rows of H = [Hu; Hy]
  int rE;           // r rounded up to 4  (dummy identity rows in between)
  int c;            // N - Ln + 1         Hankel columns = len(alpha)
  int npu;          // n * m              length of u_past
  int convex;
  int max_iter;
  int gram_dense;   // 1: dense MFMA Gram, 0: structured (nch == 4 only)
  int xs_len;       // doubles reserved for xflat in LDS
  double lam;       // lamb_alpha * eps_max (0 for nominal)
  double lamb_sigma;
  double bound;     // c * eps_max
  double sig_scale; // -lam / lamb_sigma: sigma_hat = sig_scale * beta on the boxed components (host-computed: a uniform
  double box_cost;  // lamb_sigma * bound^2     fp64 value formed in the kernel would sit in a VGPR pair for its whole run)
  const double* tabd;
  const int* tabi;
  int refine;           // iterative refinement with exact Hankel products: 0 off, 1 auto (conditioning estimate), 2 always
  int refine_max;       // cap on refinement passes per factorisation
  double refine_cond;   // auto: refine when max K_kk * max 1/d_k (a lower bound of cond K) exceeds this
  int epoch;            // launch counter of the handle (AUTO refinement: flags / last-flagged stamp carry it, nothing is cleared)
  int dense_w;          // 1: dense weighting matrices -> lam * W^-1 is the full [RP][RP] matrix `dmat`
  const double* dmat;   //    (shared by the batch, zero outside the weighted components), tabd D0 = D1 = 0
};

template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (N > 0) {
    static_for<N - 1>(f);
    f(std::integral_constant<int, N - 1>{});
  }
}

// Lower-triangular 16x16 tile map.  Whole tile DIAGONALS are dealt to the W waves
// (longest first, to the least loaded wave): the structured Gram walks down a tile
// diagonal entirely inside one lane's registers, and every tile column still has its
// tiles spread over the waves for the panel extraction.
template <int NT, int W>
struct TileMap {
  struct Tab { int wave[NT]; int base[NT]; int maxs; };
  static constexpr Tab make() {
    Tab t{};
    int load[W] = {};
    t.maxs = 0;
    for (int d = 0; d < NT; ++d) {
      int w = 0;
      for (int i = 1; i < W; ++i) if (load[i] < load[w]) w = i;
      t.wave[d] = w;
      t.base[d] = load[w];
      load[w] += NT - d;
      if (load[w] > t.maxs) t.maxs = load[w];
    }
    return t;
  }
  static constexpr Tab tab = make();
  static constexpr int MAXS = tab.maxs;
  static constexpr int wave(int I, int J) { return tab.wave[I - J]; }
  static constexpr int slot(int I, int J) { return tab.base[I - J] + J; }
};

// The lower tiles one wave owns, in column-major (J, then I) order: lets the per-step loops instantiate one body per
// owned tile instead of one per (I, J) pair of the whole matrix (compile time of the large instances).
template <int NT, int W, int WAVE>
struct WaveTiles {
  struct Tab { int I[NT * (NT + 1) / 2]; int J[NT * (NT + 1) / 2]; int n; };
  static constexpr Tab make() {
    Tab t{};
    t.n = 0;
    for (int J = 0; J < NT; ++J)
      for (int I = J; I < NT; ++I)
        if (TileMap<NT, W>::wave(I, J) == WAVE) { t.I[t.n] = I; t.J[t.n] = J; ++t.n; }
    return t;
  }
  static constexpr Tab tab = make();
};

// By-value selects: keep ternaries over captured variables from turning into
// pointer selects (which would pin the accumulators in scratch memory).
__device__ __forceinline__ double sel4(int k, double a, double b, double c, double d) {
  double v = d;
  v = (k == 2) ? c : v;
  v = (k == 1) ? b : v;
  v = (k == 0) ? a : v;
  return v;
}

// 1/sqrt(x): hardware seed (v_rsq_f64) + one third-order correction step (full fp64
// accuracy for the ~2^-26 seed); no special-case handling -- pivots are checked separately.
__device__ __forceinline__ double rsq_nr(double x) {
  const double y0 = __builtin_amdgcn_rsq(x);
  const double e = fma(-x * y0, y0, 1.0);
  return fma(y0 * e, fma(e, 0.375, 0.5), y0);
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// LDS carve-up (doubles).  Everything lives in ONE dynamic array (16-B aligned).
// All offsets are compile-time; only the length of the trajectory region (last) is
// a runtime value.  Region U holds the two Cholesky panel buffers:
//   PT[4][RP] raw (negated) panel columns, LT[4][RP] the factored panel rows.
template <int NT>
struct Lds {
  static constexpr int RP = 16 * NT;
  static constexpr int dvec = 0;
  static constexpr int tvec = dvec + RP;
  static constexpr int beta = tvec + RP;
  static constexpr int msave = beta + RP;              // per step: i0..i3, m10, m20, m21, m30, m31, m32 (12 slots)
  static constexpr int red = msave + 3 * RP;           // 32
  static constexpr int ints = red + 32;                // int act[RP], int flags[8]
  static constexpr int U = (ints + (RP + 8 + 1) / 2 + 2) & ~1;
  static constexpr int RS = RP + 4;                    // row stride of PT/LT: +4 doubles spreads the 4 k-rows over the banks
  static constexpr int USIZE = 8 * RS;                 // PT[4][RS] + LT[4][RS]
  static constexpr int ctab = U + USIZE;               // lag blocks C[d][a][b], d < RP/4 (structured Gram)
  static constexpr int xs = ctab + 4 * RP;             // trajectory, channel-interleaved
  __host__ __device__ static constexpr int total(int xs_len) { return (xs + xs_len + 1) & ~1; }
};

// --------------------------------------------------------------------------
// The per-wave body.  WAVE is a compile-time wave index so that every access to
// the accumulator tiles is statically indexed (they must stay in registers).
// --------------------------------------------------------------------------
template <int NT, int W, int WAVE>
__device__ __forceinline__ void wave_body(const KParams& P, double* __restrict__ sm,
                                          const double* __restrict__ up, const double* __restrict__ yp,
                                          double* __restrict__ u_opt, double* __restrict__ cost_out,
                                          int* __restrict__ status_out, int* __restrict__ iters_out,
                                          double* __restrict__ beta_ws, signed char* __restrict__ act_ws,
                                          unsigned long long* __restrict__ stamps, double* __restrict__ lfac) {
  using TM = TileMap<NT, W>;
  using WT = WaveTiles<NT, W, WAVE>;
  using LD = Lds<NT>;
  constexpr int RP = 16 * NT;
  constexpr int NTHR = 64 * W;
  // diagnostic phase stamps (shader clock), wave 0 / lane 0 only, off unless requested
  int nstamp = 1;
  auto stamp = [&]() __attribute__((always_inline)) {
    if (stamps != nullptr && WAVE == 0) {
      const unsigned long long t = __builtin_amdgcn_s_memtime();
      if (threadIdx.x == 0 && nstamp < 13) stamps[nstamp] = t;
      ++nstamp;
    }
  };
  double* xs = sm + LD::xs;
  double* UU = sm + LD::U;
  double* dvec = sm + LD::dvec;
  double* tvec = sm + LD::tvec;
  double* beta = sm + LD::beta;
  double* msave = sm + LD::msave;
  double* red = sm + LD::red;
  double* ctab = sm + LD::ctab;
  int* act = reinterpret_cast<int*>(sm + LD::ints);
  int* flags = act + RP;            // [0] fail, [1] active set changed

  const int tid0 = threadIdx.x;
  const int r = P.r, rE = P.rE, nch = P.nch;
  const int NS = rE >> 2;           // panel steps
  const int IR = rE >> 4;           // tile row holding the rhs row (row index rE)
  const int rr = rE & 15;

  d4 acc[TM::MAXS];

  // per-component constants of this thread's rows (tables are L2-resident, shared by the batch)
  constexpr int NE = (RP + NTHR - 1) / NTHR;
  double cD0[NE], cD1[NE], cT[NE];
  int cK[NE];
  static_for<NE>([&](auto e) __attribute__((always_inline)) {
    const int rho = tid0 + e * NTHR;
    cD0[e()] = 0.0; cD1[e()] = 0.0; cT[e()] = 0.0; cK[e()] = K_PAD;
    if (rho < RP) {
      cD0[e()] = P.tabd[0 * RP + rho];
      cD1[e()] = P.tabd[1 * RP + rho];
      cT[e()] = P.tabd[2 * RP + rho];
      cK[e()] = P.tabi[0 * RP + rho];
      const int pidx = P.tabi[1 * RP + rho];
      if (pidx >= 0) cT[e()] = (pidx < P.npu) ? up[pidx] : yp[pidx - P.npu];
      act[rho] = 0;
    }
  });
  if (tid0 < 8) flags[tid0] = 0;

  int iter = 0;
  int status = 0;
  int tid = tid0;
  for (;;) {
    ++iter;
    // Opaque per-iteration copy of the thread id: without it LICM hoists every
    // lane-dependent LDS address of the body out of this loop and keeps hundreds of
    // them live across it (massive spilling).
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63;
    const int l15 = lane & 15, l4 = lane >> 4, l3 = lane & 3, lo = l15 >> 2;
    // ---- component tables for the current active set -----------------------
    static_for<NE>([&](auto e) __attribute__((always_inline)) {
      const int rho = tid + e * NTHR;
      if (rho < RP) {
        const int s_act = act[rho];
        dvec[rho] = s_act ? cD1[e()] : cD0[e()];
        tvec[rho] = cT[e()] + s_act * P.bound;
      }
    });
    if (tid == 0) flags[1] = 0;
    __syncthreads();   // trajectory staged (first pass), tables visible, U free
    stamp();           // 1

    if (P.gram_dense) {
      // ---- G = H H' by fp64 MFMA over the implicit Hankel operand ------------
      // Rows >= r of the padded operand read live trajectory data; the garbage they
      // produce lands only in padded rows/cols of G and is cleared in the fix-up below,
      // so the main loop carries no masks.  Only the last partial k-step is masked.
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
          acc[TM::slot(I, J)] =
              __builtin_amdgcn_mfma_f64_16x16x4f64(op[I], op[J], acc[TM::slot(I, J)], 0, 0, 0);
        });
      }
      if (cfull < c) {
        const bool kok = (cfull + l4) < c;
        double op[NT];
        static_for<NT>([&](auto I) __attribute__((always_inline)) { const double v = xp[16 * I]; op[I] = kok ? v : 0.0; });
        static_for<WT::tab.n>([&](auto K) __attribute__((always_inline)) {
          constexpr int I = WT::tab.I[K], J = WT::tab.J[K];
          acc[TM::slot(I, J)] =
              __builtin_amdgcn_mfma_f64_16x16x4f64(op[I], op[J], acc[TM::slot(I, J)], 0, 0, 0);
        });
      }
      stamp();   // 2
      stamp();   // 3
    } else {
      // ---- structured Gram (nch == 4), all on the matrix pipe -----------------------
      // With a_I(i)[r] = xflat[4i + 16I + r] the tile (I,J) of G is sum_{i<c} a_I(i) b_J(i)'.
      // Because a_{I+1}(i) = a_I(i+4), one step down a tile diagonal is
      //   tile(I+1,J+1) = tile(I,J) - sum_{i<4} a_I(i) b_J(i)' + sum_{c<=i<c+4} a_I(i) b_J(i)',
      // i.e. one rank-4 downdate and one rank-4 update = 2 MFMAs (the Hankel sliding-window
      // recurrence in matrix form).  Only the first tile of each diagonal needs the full sum.
      const int c = P.c, Ln = P.Ln;
      // (1) lag blocks C_d(a,b) = sum_{t<c} x_a[t+d] x_b[t] (= G(d,0)) by v_mfma_f64_4x4x4_4b: one
      //     instruction does 4 lags x 4 time steps with no wasted outputs (17 clk vs 64 for a
      //     16x16x4).  Lane layout (probed, tools/mfma_f64_4x4_probe.hip): A_blk[i][k] at lane
      //     (k<<4 | blk<<2 | i), B_blk[k][j] at (k<<4 | blk<<2 | j), D_blk[i][j] at (i<<4 | blk<<2 | j).
      {
        // Wave w takes the MAXG CONSECUTIVE lag groups g = w*MAXG + gi (4 lags each).  The A operand of
        // group g at k-step u is x[.. + 16 (g + u)]: it depends on g + u only, so one trip of 4 k-steps needs
        // MAXG + 3 A loads instead of 4 MAXG (this phase is bound by the LDS pipe, not by the MFMAs), and a
        // wave whose groups all lie past the last lag skips the phase.
        constexpr int MAXG = (NT + W - 1) / W;             // lag groups per wave
        const int ngroups = (Ln + 3) >> 2;
        const int kq = lane >> 4, blk = (lane >> 2) & 3, ij = lane & 3;
        // The table only depends on the trajectory: later active-set iterations reuse it (ctab has its own LDS region).
        if (iter == 1 && WAVE * MAXG < ngroups) {
          double cacc[MAXG];
          static_for<MAXG>([&](auto gi) __attribute__((always_inline)) { cacc[gi()] = 0.0; });
          const double* pB = xs + 4 * kq + ij;                                   // B[k][j] = x_j[t0 + k]
          const double* pA = xs + 4 * (kq + blk) + ij + 16 * (WAVE * MAXG);      // A[i][k] = x_i[t0 + k + 4g + blk]
          const int cfull = c & ~3;
          // Groups past the last lag (4g >= Ln) just compute unused lags: no branches in the loop
          // (reads stay inside the zero-padded trajectory region).
          int t0 = 0;
          for (; t0 + 16 <= cfull; t0 += 16) {             // 4 k-steps per trip: all loads first, then the MFMAs
            double bv[4], av[MAXG + 3];
            static_for<4>([&](auto u) __attribute__((always_inline)) { bv[u()] = pB[16 * u]; });
            static_for<MAXG + 3>([&](auto q) __attribute__((always_inline)) { av[q()] = pA[16 * q]; });
            static_for<4>([&](auto u) __attribute__((always_inline)) {
              static_for<MAXG>([&](auto gi) __attribute__((always_inline)) {
                cacc[gi()] = __builtin_amdgcn_mfma_f64_4x4x4f64(av[gi() + u()], bv[u()], cacc[gi()], 0, 0, 0);
              });
            });
            pA += 64; pB += 64;
          }
          for (; t0 < cfull; t0 += 4) {
            const double bv = pB[0];
            static_for<MAXG>([&](auto gi) __attribute__((always_inline)) {
              cacc[gi()] = __builtin_amdgcn_mfma_f64_4x4x4f64(pA[16 * gi], bv, cacc[gi()], 0, 0, 0);
            });
            pA += 16; pB += 16;
          }
          if (cfull < c) {
            const bool kok = (cfull + kq) < c;
            const double bv = kok ? pB[0] : 0.0;
            static_for<MAXG>([&](auto gi) __attribute__((always_inline)) {
              const double a1 = pA[16 * gi];
              cacc[gi()] = __builtin_amdgcn_mfma_f64_4x4x4f64(kok ? a1 : 0.0, bv, cacc[gi()], 0, 0, 0);
            });
          }
          static_for<MAXG>([&](auto gi) __attribute__((always_inline)) {
            constexpr int g = WAVE * MAXG + gi;
            const int d = 4 * g + blk;                     // D layout: i = lane>>4, j = lane&3
            if (g < ngroups && d < Ln) ctab[d * 16 + kq * 4 + ij] = cacc[gi()];
          });
        }
      }
      __syncthreads();
      stamp();   // 2
      // (explicit definition of every tile: keeps the previous active-set iteration's values from
      //  being considered live across the loop back-edge; placed after the lag blocks so that the
      //  accumulator registers are free during that phase)
      static_for<TM::MAXS>([&](auto S) __attribute__((always_inline)) { acc[S] = d4{0.0, 0.0, 0.0, 0.0}; });
      // (2) first tile of every owned tile diagonal from the lag blocks:
      //     G(l+d, l)(a,b) = C_d(a,b) + sum_{j<l} ( x_a[j+c+d] x_b[j+c] - x_a[j+d] x_b[j] ),  l = lo <= 3.
      //     Lane (a = l4, b = l3, lo), register j of tile (d,0): k = 4d+j, l = lo, lag 4d+j-lo.
      {
        const double* qb0 = xs + l3;                       // x_b[.]      at  qb0[4*time]
        const double* qb1 = qb0 + 4 * c;                   // x_b[. + c]
#pragma nounroll
        for (int d = 0; d < NT; ++d) {                     // runtime loop (an unrolled one gets hoisted into spills)
          bool mine = false;
          static_for<NT>([&](auto DD) __attribute__((always_inline)) {
            if constexpr (TM::tab.wave[DD] == WAVE) mine = mine || (d == DD);
          });
          if (!mine) continue;
          auto base = [&](int j) __attribute__((always_inline)) -> double {
            int del = 4 * d + j - lo;
            del = del < 0 ? 0 : del;                       // upper triangle of a diagonal tile: don't care
            del = del >= Ln ? Ln - 1 : del;                // padded rows: cleared in the fix-up
            const double* qa0 = xs + 4 * del + l4;         // x_a[del + .]
            const double* qa1 = qa0 + 4 * c;
            double t = ctab[del * 16 + l4 * 4 + l3];
            // branch-free: all three window terms are computed (reads are in range), unused ones dropped
            const double e0 = qa1[0] * qb1[0] - qa0[0] * qb0[0];
            const double e1 = qa1[4] * qb1[4] - qa0[4] * qb0[4];
            const double e2 = qa1[8] * qb1[8] - qa0[8] * qb0[8];
            t += (0 < lo) ? e0 : 0.0;
            t += (1 < lo) ? e1 : 0.0;
            t += (2 < lo) ? e2 : 0.0;
            return t;
          };
          const d4 v = d4{base(0), base(1), base(2), base(3)};
          static_for<NT>([&](auto DD) __attribute__((always_inline)) {
            if constexpr (TM::tab.wave[DD] == WAVE) { if (d == DD) acc[TM::slot(DD, 0)] = v; }
          });
        }
      }
      // (3) walk down the diagonals: 2 MFMAs per tile
      static_for<NT>([&](auto DD) __attribute__((always_inline)) {
        constexpr int d = DD;
        if constexpr (TM::tab.wave[d] == WAVE && d + 1 < NT) {
          const double* pb = xs + 4 * l4 + l15;          // b_J(i = l4), J = t
          const double* pa = pb + 16 * d;                // a_I(i = l4), I = d + t
#pragma nounroll
          for (int t = 0; t + 1 < NT - d; ++t) {         // runtime loop: keeps the loads from being hoisted wholesale
            const double a1 = pa[0], b1 = pb[0], a2 = pa[4 * c], b2 = pb[4 * c];
            static_for<NT - d - 1>([&](auto T) __attribute__((always_inline)) {
              if (t == T) {
                const d4 v = __builtin_amdgcn_mfma_f64_16x16x4f64(-a1, b1, acc[TM::slot(d + T, T)], 0, 0, 0);
                acc[TM::slot(d + T + 1, T + 1)] = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, b2, v, 0, 0, 0);
              }
            });
            pa += 16; pb += 16;
          }
        }
      });
      stamp();   // 3
    }

    // ---- accumulators := -(G + lam*D); identity on dummy rows; rhs row := -t -----------
    // The factorisation keeps the NEGATED matrix in the accumulators so that the rank-4
    // trailing updates are plain  acc += L_I L_J'  (MFMA has no operand-negate modifier).
    static_for<WT::tab.n>([&](auto K) __attribute__((always_inline)) {
      constexpr int I = WT::tab.I[K], J = WT::tab.J[K];
      constexpr int S = TM::slot(I, J);
      const int col = 16 * J + l15;
      d4 v = -acc[S];
      if constexpr (I == J) {
        static_for<4>([&](auto j) __attribute__((always_inline)) {
          const int row = 16 * I + l4 + 4 * j;
          if (row == col && row < r) v[j()] -= P.lam * dvec[row];
          if (row < col) v[j()] = 0.0;
        });
      }
      if (16 * I + 15 >= r) {                 // wave-uniform: tile rows that touch the padding / rhs row
        static_for<4>([&](auto j) __attribute__((always_inline)) {
          const int row = 16 * I + l4 + 4 * j;
          if (row >= r || col >= r) v[j()] = 0.0;
          if (I == J && row == col && row >= r && row < rE) v[j()] = -1.0;
          if (row == rE && col < r) v[j()] = -tvec[col];
        });
      }
      acc[S] = v;
    });

    // ---- dense weighting matrices (controller.py:708-710 with non-diagonal Q, R): the penalty
    //      term lam * W^-1 is a full matrix, identical for the whole batch, read from L2
    if (P.dense_w) {
      static_for<WT::tab.n>([&](auto K) __attribute__((always_inline)) {
        constexpr int I = WT::tab.I[K], J = WT::tab.J[K];
        constexpr int S = TM::slot(I, J);
        const double* dm = P.dmat + (long long)(16 * I + l4) * RP + 16 * J + l15;
        static_for<4>([&](auto j) __attribute__((always_inline)) {
          const double dv = dm[4 * j() * RP];
          if (I != J || l4 + 4 * j() >= l15) acc[S][j()] -= P.lam * dv;
        });
      });
    }

    // ---- blocked Cholesky, 4-wide panels, two barriers per step -----------------------
    //   (1) [at the end of the previous step] raw, negated panel columns -> PT
    //   (2) one thread per row: factor the 4x4 diagonal block (redundantly per thread),
    //       forward-substitute its own row, write the row of L to LT; the wave that owns
    //       no rows does the pivot check and forms M = L11^-1 for the back substitution
    //   (3) every wave: operands from LT; first the rank-4 update of the tiles that hold the
    //       next panel, whose columns go straight back out to PT (look-ahead); then the rest
    //       of the trailing update, which drains under the next step's chain; final L is
    //       written back into the panel columns (kept for the back substitution)
    constexpr int RS = LD::RS;
    double* PT = UU;
    double* LT = UU + 4 * RS;
    constexpr bool IDLE_WAVE_DOES_M = (W > 1) && (RP <= 64 * (W - 1)) && (NE == 1);
    long long tph0 = 0, tph1 = 0, tph2 = 0, tph3 = 0, tph4 = 0;
    const bool timing = (stamps != nullptr) && (WAVE == 0);
    auto now = [&]() __attribute__((always_inline)) -> long long { return timing ? (long long)__builtin_amdgcn_s_memtime() : 0; };
    auto save_m = [&](int s, double d0, double d1, double d2v, double d3, double i0, double i1, double i2, double i3,
                      double l10, double l20, double l21, double l30, double l31, double l32)
                      __attribute__((always_inline)) {
      const bool ok = (d0 > 0.0) && (d1 > 0.0) && (d2v > 0.0) && (d3 > 0.0) &&
                      (i0 < 1e150) && (i1 < 1e150) && (i2 < 1e150) && (i3 < 1e150);
      if (!ok) flags[0] = 1;
      const double m10 = -l10 * i0 * i1;
      const double m20 = -(l20 * i0 + l21 * m10) * i2, m21 = -(l21 * i1) * i2;
      const double m30 = -(l30 * i0 + l31 * m10 + l32 * m20) * i3, m31 = -(l31 * i1 + l32 * m21) * i3,
                   m32 = -(l32 * i2) * i3;
      double* ms = msave + 12 * s;
      ms[0] = i0; ms[1] = i1; ms[2] = i2; ms[3] = i3; ms[4] = m10; ms[5] = m20; ms[6] = m21;
      ms[7] = m30; ms[8] = m31; ms[9] = m32;
    };
    auto extract_panel = [&](auto JN, int qn) __attribute__((always_inline)) {
      constexpr int Jn = JN;
      if (lo == qn) {
        static_for<NT>([&](auto I) __attribute__((always_inline)) {
          if constexpr (I >= Jn && TM::wave(I, Jn) == WAVE) {
            constexpr int S = TM::slot(I, Jn);
            static_for<4>([&](auto j) __attribute__((always_inline)) {
              PT[l3 * RS + 16 * I + l4 + 4 * j] = acc[S][j()];
            });
          }
        });
      }
    };
    __syncthreads();   // U is free (dense path never used it; structured path neither)
    extract_panel(std::integral_constant<int, 0>{}, 0);
    static_for<NT>([&](auto JB) __attribute__((always_inline)) {
      constexpr int Jb = JB;
      const int qend = (NS - 4 * Jb) < 4 ? (NS - 4 * Jb) : 4;      // <=0 past the last panel
      for (int q = 0; q < qend; ++q) {
        const int s = 4 * Jb + q;
        const int c0 = 4 * s;
        const int lim = c0 + 4;                                     // rows/cols below this are final
        const long long ta = now();
        __syncthreads();
        const long long tc = now();
        // (2)
        if constexpr (IDLE_WAVE_DOES_M && WAVE == W - 1) {
          const double* Pd = PT + c0;
          const double p00 = -Pd[0 * RS + 0];
          const double p10 = -Pd[0 * RS + 1], p11 = -Pd[1 * RS + 1];
          const double p20 = -Pd[0 * RS + 2], p21 = -Pd[1 * RS + 2], p22 = -Pd[2 * RS + 2];
          const double p30 = -Pd[0 * RS + 3], p31 = -Pd[1 * RS + 3], p32 = -Pd[2 * RS + 3], p33 = -Pd[3 * RS + 3];
          const double d0 = p00;
          const double i0 = rsq_nr(d0);
          const double l10 = p10 * i0, l20 = p20 * i0, l30 = p30 * i0;
          const double d1 = p11 - l10 * l10;
          const double i1 = rsq_nr(d1);
          const double l21 = (p21 - l20 * l10) * i1, l31 = (p31 - l30 * l10) * i1;
          const double d2v = p22 - l20 * l20 - l21 * l21;
          const double i2 = rsq_nr(d2v);
          const double l32 = (p32 - l30 * l20 - l31 * l21) * i2;
          const double d3 = p33 - l30 * l30 - l31 * l31 - l32 * l32;
          const double i3 = rsq_nr(d3);
          if (lane == 0) save_m(s, d0, d1, d2v, d3, i0, i1, i2, i3, l10, l20, l21, l30, l31, l32);
        }
        static_for<NE>([&](auto e) __attribute__((always_inline)) {
          constexpr int row0 = 64 * WAVE + e * NTHR;               // first row of this wave's 64-row slab
          if constexpr (row0 < RP) {
            if (row0 + 63 >= c0) {                                  // wave-uniform: slab still has live rows
              const int row = tid + e * NTHR;
              const double* Pd = PT + c0;
              const double p00 = -Pd[0 * RS + 0];
              const double p10 = -Pd[0 * RS + 1], p11 = -Pd[1 * RS + 1];
              const double p20 = -Pd[0 * RS + 2], p21 = -Pd[1 * RS + 2], p22 = -Pd[2 * RS + 2];
              const double p30 = -Pd[0 * RS + 3], p31 = -Pd[1 * RS + 3], p32 = -Pd[2 * RS + 3], p33 = -Pd[3 * RS + 3];
              const double r0 = PT[0 * RS + row], r1 = PT[1 * RS + row], r2 = PT[2 * RS + row], r3 = PT[3 * RS + row];
              const double d0 = p00;
              const double i0 = rsq_nr(d0);
              const double l10 = p10 * i0, l20 = p20 * i0, l30 = p30 * i0;
              const double d1 = p11 - l10 * l10;
              const double i1 = rsq_nr(d1);
              const double l21 = (p21 - l20 * l10) * i1, l31 = (p31 - l30 * l10) * i1;
              const double d2v = p22 - l20 * l20 - l21 * l21;
              const double i2 = rsq_nr(d2v);
              const double l32 = (p32 - l30 * l20 - l31 * l21) * i2;
              const double d3 = p33 - l30 * l30 - l31 * l31 - l32 * l32;
              const double i3 = rsq_nr(d3);
              // forward substitution of this row (PT holds the negated entries).  For the rows of
              // the diagonal block the same substitution yields L11 itself; only its strictly
              // upper part has to be zeroed.
              const double x0 = -r0 * i0;
              double x1 = -(r1 + x0 * l10) * i1;
              double x2 = -(r2 + x0 * l20 + x1 * l21) * i2;
              double x3 = -(r3 + x0 * l30 + x1 * l31 + x2 * l32) * i3;
              const int ii = row - c0;
              x1 = (ii < 1) ? 0.0 : x1;
              x2 = (ii < 2) ? 0.0 : x2;
              x3 = (ii < 3) ? 0.0 : x3;
              if (row >= c0 && row < RP) {
                LT[0 * RS + row] = x0; LT[1 * RS + row] = x1; LT[2 * RS + row] = x2; LT[3 * RS + row] = x3;
              }
              if constexpr (!IDLE_WAVE_DOES_M) {
                if (row == c0) save_m(s, d0, d1, d2v, d3, i0, i1, i2, i3, l10, l20, l21, l30, l31, l32);
              }
            }
          }
        });
        const long long td = now();
        __syncthreads();
        const long long te = now();
        // (3)
        {
          double op[NT];
          static_for<NT>([&](auto I) __attribute__((always_inline)) {
            if constexpr (I >= Jb) {
              const double v = LT[l4 * RS + 16 * I + l15];
              if constexpr (I == Jb) op[I] = (16 * I + l15 >= lim) ? v : 0.0;
              else op[I] = v;                      // every row/col of a later tile is >= lim
            }
          });
          // (3a) tiles holding the NEXT panel's columns, then that panel goes out to PT
          //      (PT was last read before barrier 2, so it is free again)
          if (q < 3) {
            static_for<NT>([&](auto I) __attribute__((always_inline)) {
              if constexpr (I >= Jb && TM::wave(I, Jb) == WAVE) {
                constexpr int S = TM::slot(I, Jb);
                acc[S] = __builtin_amdgcn_mfma_f64_16x16x4f64(op[I], op[Jb], acc[S], 0, 0, 0);
              }
            });
            if (s + 1 < NS) extract_panel(std::integral_constant<int, Jb>{}, q + 1);
          } else if constexpr (Jb + 1 < NT) {
            static_for<NT>([&](auto I) __attribute__((always_inline)) {
              if constexpr (I >= Jb + 1 && TM::wave(I, Jb + 1) == WAVE) {
                constexpr int S = TM::slot(I, Jb + 1);
                acc[S] = __builtin_amdgcn_mfma_f64_16x16x4f64(op[I], op[Jb + 1], acc[S], 0, 0, 0);
              }
            });
            if (s + 1 < NS) extract_panel(std::integral_constant<int, Jb + 1>{}, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
          // (3b) the rest of the trailing update
          static_for<WT::tab.n>([&](auto K) __attribute__((always_inline)) {
            constexpr int I = WT::tab.I[K], J = WT::tab.J[K];
            if constexpr (J > Jb) {
              constexpr int S = TM::slot(I, J);
              if constexpr (J == Jb + 1) {
                if (q < 3) acc[S] = __builtin_amdgcn_mfma_f64_16x16x4f64(op[I], op[J], acc[S], 0, 0, 0);
              } else {
                acc[S] = __builtin_amdgcn_mfma_f64_16x16x4f64(op[I], op[J], acc[S], 0, 0, 0);
              }
            }
          });
          // (3c) keep the final L in the panel columns of this tile column
          if (lo == q) {
            static_for<NT>([&](auto I) __attribute__((always_inline)) {
              if constexpr (I >= Jb && TM::wave(I, Jb) == WAVE) {
                constexpr int S = TM::slot(I, Jb);
                static_for<4>([&](auto j) __attribute__((always_inline)) {
                  acc[S][j()] = LT[l3 * RS + 16 * I + l4 + 4 * j];
                });
              }
            });
          }
        }
        const long long tf = now();
        tph1 += tc - ta; tph2 += td - tc; tph3 += te - td; tph4 += tf - te;
      }
    });
    if (timing && threadIdx.x == 0) { stamps[7] = tph0; stamps[8] = tph1; stamps[9] = tph2; stamps[10] = tph3; stamps[11] = tph4; }
    stamp();   // 4
    stamp();   // 5

    // ---- optional export of the factor (ddmpc_prepare): lower tiles, row-major 16x16 each,
    //      tile (I,J) at lfac[(I(I+1)/2 + J) * 256]; register j of lane l is row l4+4j, col l15
    if (lfac != nullptr) {
      static_for<WT::tab.n>([&](auto K) __attribute__((always_inline)) {
        constexpr int I = WT::tab.I[K], J = WT::tab.J[K];
        constexpr int S = TM::slot(I, J);
        double* dst = lfac + (I * (I + 1) / 2 + J) * 256 + lane;
        static_for<4>([&](auto j) __attribute__((always_inline)) { dst[64 * j] = acc[S][j()]; });
      });
    }

    // ---- y = L^-1 t sits in row rE of the tiles -> tvec[] -----------------------------
    static_for<WT::tab.n>([&](auto K) __attribute__((always_inline)) {
      constexpr int I = WT::tab.I[K], J = WT::tab.J[K];
      if (I == IR) {
        constexpr int S = TM::slot(I, J);
        static_for<4>([&](auto j) __attribute__((always_inline)) {
          if (l4 + 4 * j == rr) tvec[16 * J + l15] = acc[S][j()];
        });
      }
    });
    __syncthreads();
    // ---- back substitution L' beta = y, one 16-row tile row per round --------------------
    //   (A) the owner of the diagonal tile solves its 16 rows in registers: 4 blocks of 4 rows,
    //       beta_blk = M_blk' y_blk with the saved inverse blocks; y values are broadcast with
    //       v_readlane, and the in-tile update of the rows above is ONE MFMA (beta in row 0 of
    //       A, accumulator register qq of the diagonal tile as B);
    //   (C) y_J -= L(Is,J)' beta_Is for every tile left of the diagonal: 4 MFMAs per tile with
    //       the accumulator registers themselves as B (register ks of a lane holds
    //       L[4ks + l4][l15] = B[k = l4][j = l15] of k-step ks).  The tile (Is, Is-1) that
    //       feeds the next diagonal solve goes first; the others are deferred past the barrier
    //       and overlap with that solve.
    {
      auto rl64 = [&](double v, int src) __attribute__((always_inline)) -> double {
        const int lo32 = __builtin_amdgcn_readlane(__double2loint(v), src);
        const int hi32 = __builtin_amdgcn_readlane(__double2hiint(v), src);
        return __hiloint2double(hi32, lo32);
      };
      auto tile_update = [&](auto IS, auto JJ, double a0, double a1, double a2, double a3) __attribute__((always_inline)) {
        constexpr int S = TM::slot(IS, JJ);
        const d4 Lt = acc[S];
        d4 dd = d4{0.0, 0.0, 0.0, 0.0};
        dd = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, Lt[0], dd, 0, 0, 0);
        dd = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, Lt[1], dd, 0, 0, 0);
        dd = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, Lt[2], dd, 0, 0, 0);
        dd = __builtin_amdgcn_mfma_f64_16x16x4f64(a3, Lt[3], dd, 0, 0, 0);
        if (l4 == 0) tvec[16 * JJ + l15] -= dd[0];
      };
      double pa0 = 0.0, pa1 = 0.0, pa2 = 0.0, pa3 = 0.0;          // operands of the previous (lower) tile row
      static_for<NT>([&](auto IREV) __attribute__((always_inline)) {
        constexpr int Is = NT - 1 - IREV;
        const bool live = 16 * Is < rE;                             // wave-uniform
        // (A)
        if constexpr (TM::wave(Is, Is) == WAVE) {
          if (live) {
            constexpr int S = TM::slot(Is, Is);
            const d4 Ld = acc[S];
            double yv = tvec[16 * Is + l15];
            const int qtop = (NS - 4 * Is) < 4 ? (NS - 4 * Is) : 4;
            static_for<4>([&](auto QR) __attribute__((always_inline)) {
              constexpr int QQ = 3 - QR;
              if (QQ < qtop) {
                const int s = 4 * Is + QQ;
                const double* ms = msave + 12 * s;
                const double y0 = rl64(yv, 4 * QQ + 0), y1 = rl64(yv, 4 * QQ + 1);
                const double y2 = rl64(yv, 4 * QQ + 2), y3 = rl64(yv, 4 * QQ + 3);
                const double b0 = ms[0] * y0 + ms[4] * y1 + ms[5] * y2 + ms[7] * y3;
                const double b1 = ms[1] * y1 + ms[6] * y2 + ms[8] * y3;
                const double b2 = ms[2] * y2 + ms[9] * y3;
                const double b3 = ms[3] * y3;
                if (lane == 0) { beta[4 * s + 0] = b0; beta[4 * s + 1] = b1; beta[4 * s + 2] = b2; beta[4 * s + 3] = b3; }
                if constexpr (QQ > 0) {
                  const double av = (l15 == 0) ? sel4(l4, b0, b1, b2, b3) : 0.0;
                  const d4 dd = __builtin_amdgcn_mfma_f64_16x16x4f64(av, Ld[QQ], d4{0.0, 0.0, 0.0, 0.0}, 0, 0, 0);
                  yv -= dd[0];                                      // meaningful in lanes 0..15 (row 0 of D)
                }
              }
            });
          }
        }
        // deferred tiles of the previous round (row Is+1, columns < Is): off the critical path
        if constexpr (Is + 1 < NT && Is >= 1) {
          if (16 * (Is + 1) < rE) {
            static_for<Is>([&](auto J) __attribute__((always_inline)) {
              if constexpr (TM::wave(Is + 1, J) == WAVE) tile_update(std::integral_constant<int, Is + 1>{}, J, pa0, pa1, pa2, pa3);
            });
          }
        }
        if constexpr (Is > 0) {
          __syncthreads();                                          // beta_Is visible; deferred updates ordered
          if (live) {
            const double* bp = beta + 16 * Is + l4;
            const int rw = 16 * Is + l4;                            // rows >= rE (rhs row, padding) carry no beta
            pa0 = (l15 == 0 && rw < rE) ? bp[0] : 0.0;
            pa1 = (l15 == 0 && rw + 4 < rE) ? bp[4] : 0.0;
            pa2 = (l15 == 0 && rw + 8 < rE) ? bp[8] : 0.0;
            pa3 = (l15 == 0 && rw + 12 < rE) ? bp[12] : 0.0;
            if constexpr (TM::wave(Is, Is - 1) == WAVE)
              tile_update(std::integral_constant<int, Is>{}, std::integral_constant<int, Is - 1>{}, pa0, pa1, pa2, pa3);
          }
          __syncthreads();                                          // y_{Is-1} complete
        }
      });
      __syncthreads();
    }
    stamp();   // 6


    // ---- slack box: primal-dual active-set update --------------------------------
    bool again = false;
    if (P.convex) {
      const double scale = -P.lam / P.lamb_sigma;
      static_for<NE>([&](auto e) __attribute__((always_inline)) {
        const int rho = tid + e * NTHR;
        if (rho < r && (cK[e()] == K_WPRED || cK[e()] == K_WTERM)) {   // sigma[n*p:], controller.py:659
          const double sh = scale * beta[rho];
          const int ns = (sh > P.bound) ? 1 : (sh < -P.bound) ? -1 : 0;
          if (ns != act[rho]) { act[rho] = ns; flags[1] = 1; }
        }
      });
      __syncthreads();
      again = (flags[1] != 0) && (flags[0] == 0);
      if (again && iter >= P.max_iter) { again = false; status = 4; }
    }
    if (!again) break;
  }
  if (flags[0] != 0) status = 4;
  const int lane = tid & 63;

  // ---- outputs --------------------------------------------------------------------
  // z = t - lam*D*beta; cost = control cost + lam*beta'z + lamb_sigma*|sigma|^2
  double part = 0.0;
  bool finite = true;
  static_for<NE>([&](auto e) __attribute__((always_inline)) {
    const int rho = tid + e * NTHR;
    if (rho < r) {
      const int s_act = act[rho];
      const double b = beta[rho];
      const double D = s_act ? cD1[e()] : cD0[e()];
      const double t = cT[e()] + s_act * P.bound;
      double z = t - P.lam * D * b;
      if (P.dense_w) {                              // z = t - lam (W^-1 beta): one row of the dense matrix
        const double* dr = P.dmat + (long long)rho * RP;
        double sdb = 0.0;
        for (int j = 0; j < r; ++j) sdb += dr[j] * beta[j];
        z = t - P.lam * (D * b + sdb);          // tabd keeps the diagonal-only components
      }
      const double wq = P.tabd[3 * RP + rho];
      const double tb = P.tabd[2 * RP + rho];       // setpoint of the component (u_s / y_s)
      const int oidx = P.tabi[2 * RP + rho];
      finite = finite && (fabs(b) < 1e300);
      double contrib = P.lam * b * z;
      const int kind = cK[e()];
      if (P.dense_w && (kind == K_UFREE || kind == K_YFREE || kind == K_WPRED)) {
        // (z - t)' W (z - t) summed over the weighted components equals -lam * beta' (z - t), t the (shifted) target;
        // a sigma held at its bound adds lamb_sigma * bound^2 (W^-1 = Q_ff^-1 + 1/lamb_sigma on the INACTIVE components)
        contrib -= P.lam * b * (z - t);
        if (s_act != 0) contrib += P.lamb_sigma * P.bound * P.bound;
      } else
      if (kind == K_UFREE || kind == K_YFREE) { const double dlt = z - tb; contrib += wq * dlt * dlt; }
      else if (kind == K_WINT) { const double sg = z - cT[e()]; contrib += P.lamb_sigma * sg * sg; }
      else if (kind == K_WTERM) { const double sg = z - tb; contrib += P.lamb_sigma * sg * sg; }
      else if (kind == K_WPRED) {
        const double sg = (s_act != 0) ? s_act * P.bound : -P.lam * b / P.lamb_sigma;
        const double dlt = z - sg - tb;
        contrib += wq * dlt * dlt + P.lamb_sigma * sg * sg;
      }
      part += contrib;
      if (oidx >= 0) u_opt[oidx] = z;               // ubar[n*m:], controller.py:799-805
      if (beta_ws) beta_ws[rho] = b;
      if (act_ws) act_ws[rho] = (signed char)s_act;
    }
  });
  part = wave_sum(part);
  const unsigned long long okmask = __ballot(finite);
  if (lane == 0) { red[tid >> 6] = part; red[16 + (tid >> 6)] = (okmask == ~0ull) ? 0.0 : 1.0; }
  __syncthreads();
  if (tid == 0) {
    double tot = 0.0, bad = 0.0;
    for (int w = 0; w < W; ++w) { tot += red[w]; bad += red[16 + w]; }
    if (bad != 0.0 || !(fabs(tot) < 1e300)) status = 4;
    *cost_out = tot;
    *status_out = status;
    if (iters_out) *iters_out = iter;
    if (stamps) { stamps[14] = __builtin_amdgcn_s_memtime(); stamps[13] = __builtin_amdgcn_s_memrealtime(); }
  }
}

// --------------------------------------------------------------------------
// Cold-solve kernel: grid = batch, block = 64*W threads.
// --------------------------------------------------------------------------
template <int NT, int W>
__global__ __launch_bounds__(64 * W, DDMPC_MIN_WAVES(NT, W)) void ddmpc_cold_solve_kernel(
    KParams P, const double* __restrict__ u_d, const double* __restrict__ y_d,
    const double* __restrict__ u_past, const double* __restrict__ y_past, double* __restrict__ u_opt,
    double* __restrict__ cost, int* __restrict__ status, int* __restrict__ iters,
    double* __restrict__ beta_ws, signed char* __restrict__ act_ws, unsigned long long* __restrict__ stamps,
    double* __restrict__ lfac, const int* __restrict__ only) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const long long b = blockIdx.x;
  // optional instance filter (ddmpc_step with the slack box: only the instances whose warm step found
  // an active bound are solved cold); uniform per workgroup
  if (only != nullptr && only[b] == 0) return;
  const int tid = threadIdx.x;
  constexpr int NTHR = 64 * W;
  unsigned long long* st = stamps ? stamps + b * 16 : nullptr;
  if (st && tid == 0) { st[0] = __builtin_amdgcn_s_memtime(); st[15] = __builtin_amdgcn_s_memrealtime(); }
  double* xs = sm + Lds<NT>::xs;
  // ---- stage the instance's trajectory, channel-interleaved: xs[t*nch + ch] ----
  {
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
  const int n = P.npu / P.m;
  const double* up = u_past + b * (long long)P.npu;
  const double* yp = y_past + b * (long long)(n * P.p);
  double* uo = u_opt + b * (long long)((P.Ln - n) * P.m);
  double* bw = beta_ws ? beta_ws + b * (long long)P.rE : nullptr;
  signed char* aw = act_ws ? act_ws + b * (long long)P.rE : nullptr;
  int* it = iters ? iters + b : nullptr;
  double* lf = lfac ? lfac + b * (long long)(NT * (NT + 1) / 2 * 256) : nullptr;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  static_for<W>([&](auto WV) {
    if (wave == WV) wave_body<NT, W, WV>(P, sm, up, yp, uo, cost + b, status + b, it, bw, aw, st, lf);
  });
}





}  // namespace ddmpc
