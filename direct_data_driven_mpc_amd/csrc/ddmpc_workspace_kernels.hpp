// ddmpc_workspace_kernels.hpp -- the one-workgroup-per-instance kernels for problems whose matrices do not fit the registers
// of the cold-solve kernel: packed storage helpers (pk_row), the semi-definite Cholesky and substitutions on packed matrices,
// the Hankel-structured Gram / products with the implicit Hankel matrix from global memory, ddmpc_nominal_rr_kernel (NOMINAL
// scheme on exact data: rank-revealing route; rescue path at the register-resident sizes, second implementation beyond them)
// and ddmpc_large_solve_kernel (ROBUST scheme beyond 271 rows: fall-back and second implementation of ddmpc_rr3.hpp).
// The phase pipelines (ddmpc_rr2.hpp, ddmpc_rr2_solve.hpp, ddmpc_rr3.hpp) build on the helpers here.  Split off
// ddmpc_aux_kernels.hpp in round 5.  Included by the API translation unit only.
#pragma once
#include "ddmpc_aux_kernels.hpp"

namespace ddmpc {

// --------------------------------------------------------------------------
// Rank-revealing solve of the NOMINAL scheme (controller.py:506-538,549-629,679-711) for instances whose
// Gram matrix is singular -- exact (noise-free) data, where rank H = m(L+n) + n_sys < r and the plain
// G beta = t of the cold kernel breaks down.  Only instances with status[b] == SOLVER_ERROR are processed.
//
//   order the components fixed-first:  G = [[G_FF, G_FR], [G_RF, G_RR]]   (F: hard values f, R: weighted)
//   semi-definite Cholesky with skipped (numerically zero) pivots, G = L L'
//   L_FF w = f  (consistency of the hard constraints is checked: otherwise "infeasible")
//   z_R = z0 + C v,  z0 = L_RF w,  C = L_RR  (its columns span what is left of range(H))
//   (C' W C) v = C' W (z_s - z0),  W = diag(weights)                  -> optimal_u = z on the free ubar rows
//
// Packed lower storage ((i,j) at i(i+1)/2 + j) for G (r rows) and the reduced normal matrix T (nR rows), in LDS
// when they fit (four-tank size), else in a global workspace; plain VALU code, one workgroup per instance.  A set-up-time / rescue path, not a
// throughput kernel.  Diagonal weights only.
// --------------------------------------------------------------------------
// Packed lower-triangular storage of the matrices in the global workspace (and in LDS at the four-tank sizes): row i holds
// its columns 0 .. i and STARTS ON A 128-BYTE BOUNDARY -- rows 16 t .. 16 t + 15 have 16 (t + 1) slots.  A 16-column piece of
// a row is then exactly one cache line (with rows packed back to back, i (i + 1) / 2, the four 32-byte lane pieces of a row
// straddle two lines fifteen times out of sixteen: twice the tag look-ups per load, and the vector-memory path of a CU that
// runs two of these workgroups is what their factorisations queue on).  +2.5 % of storage at 608 rows.
__host__ __device__ __forceinline__ size_t pk_row(size_t i) {
  const size_t t = i >> 4;
  return 128 * t * (t + 1) + (i & 15) * 16 * (t + 1);
}
__host__ __device__ __forceinline__ size_t pk_size(size_t n) { return pk_row(n); }   // (a multiple of 16: what follows stays aligned)

__device__ __forceinline__ int tri_row(int e) {       // row of entry e in the LOGICAL enumeration e = i (i + 1) / 2 + j
  int i = (int)((sqrt(8.0 * e + 1.0) - 1.0) * 0.5);
  while ((i + 1) * (i + 2) / 2 <= e) ++i;
  while (i * (i + 1) / 2 > e) --i;
  return i;
}

// In-place Cholesky of a packed lower matrix ((i,j) at i(i+1)/2 + j) with `n` rows; pivots <= tol_abs are skipped
// (column zeroed, skip[k] = 1).  `ncols` < n stops after the first ncols columns (rows >= ncols of those columns hold
// the factor's off-diagonal block, the trailing block is left untouched).  `pan`: PSD_PAN doubles of LDS.
constexpr int PSD_PAN = 2048;   // doubles of LDS scratch (1312 used by the factorisation; the rest widens the trajectory
                                // chunks of hankel_gram_packed)
//
// Left-looking over 32-wide panels, everything below the diagonal tiles done by v_mfma_f64_16x16x4:
//
//   * update:   P(I, C)' = A(I, C)' - sum_{j < k0} L(C, j) L(I, j)'     16x16 accumulator tiles, kept TRANSPOSED
//     (register q of lane (l4, l15) = entry [panel column l4 + 4q][row l15] of the tile), so that a finished tile is
//     directly the B operand of the left multiplications below.  Both operands are rows of the packed factor: a lane
//     loads 4 consecutive entries (32 B) of "its" row per 16 columns of j -- whole cache lines per wave -- and feeds
//     4 MFMAs per operand pair.
//   * the two 16x16 diagonal tiles of a panel are factored by ONE wave each in LDS (pivot rule above), which also
//     forms Mt = S L~^-1 (L~: unit diagonal where a pivot was skipped, S zeroes those rows), so that the rows below are
//     X' = Mt P' -- 4 MFMAs per tile -- and the second half of the panel is updated with the first by 4 more.
//   * traffic: the factor is streamed once per 32 columns (left-looking), half of what 16-wide panels read.
//
// Row tiles are dealt round-robin to the waves, PSD_TG tiles per wave and pass (48 accumulator VGPRs).
// LDS: 1312 doubles of `pan`.  Requires at least two waves (blockDim.x a multiple of 64, >= 128).
constexpr int PSD_TG = 3;
typedef double d2u8 __attribute__((ext_vector_type(2), aligned(8)));     // 16-byte load from an 8-byte aligned packed row

// One wave factors a 16x16 tile held row-major in LDS (lower triangle valid), columns [0, nbt): pivots <= tol_abs are
// skipped (column zeroed).  Also writes Ms[k][m] = Mt[m][k], Mt = S L~^-1 restricted to those columns (k-major: the A
// operand of X' = Mt P').  `skipout` receives nbt flags.
__device__ __forceinline__ void psd_tile_factor(double* Dg, double* Ms, double* Dinv, int nbt, double tol_abs, int* skipout) {
  const int lane = threadIdx.x & 63;
  const int rr = lane & 15, cg = lane >> 4;            // lane = (row of the tile, one of four column groups)
  for (int c = 0; c < nbt; ++c) {
    // one LDS round trip per column: every lane reads the pivot, its row's entry of column c and the column-c entries
    // of the (up to four) columns it updates, scales them itself and writes the results back
    const double dk = Dg[c * 16 + c];
    const double lrc = Dg[rr * 16 + c];
    double lcc[4], old[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int c2 = c + 1 + cg + 4 * t;
      const bool on = c2 < nbt && c2 <= rr;
      lcc[t] = on ? Dg[c2 * 16 + c] : 0.0;
      old[t] = on ? Dg[rr * 16 + c2] : 0.0;
    }
    const bool sk = !(dk > tol_abs);                   // the same value in every lane
    // 1/sqrt by the hardware seed + two Newton steps: `1.0 / sqrt(dk)` expands to two long software sequences on the
    // critical path of every column of every diagonal tile.  The seed is only good to ~2^-24..2^-26, so ONE step leaves
    // ~1.5 e0^2 ~ 5e-15 (some 50 eps in every pivot of the factor); the second brings it to rounding level.
    const double dks = sk ? 1.0 : dk;
    const double y0 = __builtin_amdgcn_rsq(dks);
    const double ye = fma(-dks * y0, y0, 1.0);
    const double y1 = fma(0.5 * y0, ye, y0);
    const double ye1 = fma(-dks * y1, y1, 1.0);
    const double inv = sk ? 0.0 : fma(0.5 * y1, ye1, y1);
    const double u = lrc * inv;
    if (cg == 0 && rr >= c) Dg[rr * 16 + c] = u;
    if (lane == 0) { skipout[c] = sk ? 1 : 0; Dinv[c] = inv; }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int c2 = c + 1 + cg + 4 * t;
      if (c2 < nbt && c2 <= rr) Dg[rr * 16 + c2] = old[t] - u * (lcc[t] * inv);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // in-wave hand-off through LDS
    __builtin_amdgcn_wave_barrier();
  }
  if (lane < 16) {                                     // column `lane` of L~^-1 by forward substitution, rows of skipped pivots zero
    double y[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      double sacc = (c == lane) ? 1.0 : 0.0;
#pragma unroll
      for (int c1 = 0; c1 < c; ++c1) sacc -= Dg[c * 16 + c1] * y[c1];
      y[c] = (c < nbt) ? sacc * Dinv[c] : 0.0;
    }
#pragma unroll
    for (int c = 0; c < 16; ++c) Ms[lane * 16 + c] = y[c];          // Ms[k = lane][m = c] = Mt[c][lane]
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
}

// `early` (full factorisations only): when a whole panel comes out without a pivot, the diagonal of the Schur complement
// behind it is formed (one pass over the rows below); if none of its entries exceeds tol_abs either, every remaining
// pivot would be skipped as well (the diagonal of a PSD Schur complement only shrinks): the remaining columns are
// marked skipped, the trailing block is zeroed and the factorisation stops (tested for panels from column `early_from` on:
// a dead panel inside the fixed block of the rank-revealing route has live rows behind it by construction, and every test
// is a latency-bound pass over the rows below -- three of them were 0.22 ms of a cfg-5 factorisation).  Returns the number of columns in front of
// that point (n when it ran to the end): all columns >= the return value are zero.  With the dependent rows ordered last
// (exact data: the rank-revealing kernel) this saves the factorisation of the dead half of the matrix.
__device__ __forceinline__ int packed_psd_cholesky(double* A, int n, double tol_abs, int* skip, double* pan, int ncols = -1,
                                                   bool early = false, int early_from = 0) {
  const int tid = threadIdx.x;
  const int lane = tid & 63, l15 = lane & 15, l4 = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), nwave = (int)(blockDim.x >> 6);
  double* Dga = pan;                 // diagonal tile of the first / second half of the panel, row-major
  double* Dgb = pan + 256;
  double* Msa = pan + 512;           // Mt of the two halves, k-major
  double* Msb = pan + 768;
  double* Xs = pan + 1024;           // X(1,0)' = rows k0+16.. of the first half's columns, k-major: Xs[k][row]
  double* Dinv = pan + 1280;         // 2 x 16 reciprocal pivots
  if (ncols < 0) ncols = n;
  auto rowp = [&](int i) -> const double* { i = i < n ? i : n - 1; return A + pk_row(i); };
  for (int k0 = 0; k0 < ncols; k0 += 32) {
    const int nba = (ncols - k0) < 16 ? (ncols - k0) : 16;
    const int nbb = (ncols - k0 - 16) < 0 ? 0 : ((ncols - k0 - 16) < 16 ? (ncols - k0 - 16) : 16);
    const int ntile = (n - k0 + 15) >> 4;
    // the panel's own rows: A operands of the update (clamped to a valid row; masked when P is formed)
    const double* pa = rowp(k0 + l15) + 4 * l4;
    const double* pb = rowp(k0 + 16 + l15) + 4 * l4;
    for (int g0 = 0; g0 < ntile; g0 += nwave * PSD_TG) {
      d4 acc[PSD_TG][2];
      const double* rp[PSD_TG];
      int ti[PSD_TG];
#pragma unroll
      for (int s = 0; s < PSD_TG; ++s) {
        ti[s] = g0 + wave + nwave * s;                                  // tile row (wave-uniform); >= ntile: idle slot
        rp[s] = rowp(k0 + 16 * ti[s] + l15) + 4 * l4;
        acc[s][0] = d4{0.0, 0.0, 0.0, 0.0};
        acc[s][1] = d4{0.0, 0.0, 0.0, 0.0};
      }
      // ---- update with the columns factored so far -------------------------------------------------------
      for (int j0 = 0; j0 < k0; j0 += 16) {
        // (all loads of the chunk unconditional and independent of one another -- pb is clamped to a valid row -- so that they
        //  share ONE memory round trip: with b0 = a0 as the fall-back the compiler waited for a0 before it issued the rest)
        const d2u8 a0 = *reinterpret_cast<const d2u8*>(pa + j0), a1 = *reinterpret_cast<const d2u8*>(pa + j0 + 2);
        const d2u8 b0 = *reinterpret_cast<const d2u8*>(pb + j0), b1 = *reinterpret_cast<const d2u8*>(pb + j0 + 2);
        // every load of the chunk is issued before the first MFMA (row pointers of idle slots are clamped to a valid
        // row): one memory round trip per 16 columns, not one per row tile
        d2u8 x0[PSD_TG], x1[PSD_TG];
#pragma unroll
        for (int s = 0; s < PSD_TG; ++s) {
          x0[s] = *reinterpret_cast<const d2u8*>(rp[s] + j0);
          x1[s] = *reinterpret_cast<const d2u8*>(rp[s] + j0 + 2);
        }
#pragma unroll
        for (int s = 0; s < PSD_TG; ++s) {
          if (ti[s] < ntile) {
            acc[s][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[0], x0[s][0], acc[s][0], 0, 0, 0);
            acc[s][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[1], x0[s][1], acc[s][0], 0, 0, 0);
            acc[s][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[0], x1[s][0], acc[s][0], 0, 0, 0);
            acc[s][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[1], x1[s][1], acc[s][0], 0, 0, 0);
            if (nbb > 0 && ti[s] > 0) {
              acc[s][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(b0[0], x0[s][0], acc[s][1], 0, 0, 0);
              acc[s][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(b0[1], x0[s][1], acc[s][1], 0, 0, 0);
              acc[s][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(b1[0], x1[s][0], acc[s][1], 0, 0, 0);
              acc[s][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(b1[1], x1[s][1], acc[s][1], 0, 0, 0);
            }
          }
        }
      }
      // ---- P' = A' - acc on the valid entries (row < n, column <= row, column < ncols), zero elsewhere -----
      {
        // (all 8 PSD_TG entries loaded unconditionally from positions clamped into the row, selected afterwards: with the
        //  test around the load every entry became an exec-masked block of its own and the loads waited for one another)
        double av[PSD_TG][2][4];
#pragma unroll
        for (int s = 0; s < PSD_TG; ++s) {
          const int i = k0 + 16 * ti[s] + l15;
          const int ic = i < n ? i : n - 1;
          const double* Ai = A + pk_row(ic);
#pragma unroll
          for (int C = 0; C < 2; ++C)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const int cc = k0 + 16 * C + l4 + 4 * q;
              av[s][C][q] = Ai[cc <= ic ? cc : ic];
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < PSD_TG; ++s) {
          const int i = k0 + 16 * ti[s] + l15;
#pragma unroll
          for (int C = 0; C < 2; ++C)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const int cc = k0 + 16 * C + l4 + 4 * q;
              const bool ok = ti[s] < ntile && i < n && cc <= i && cc < ncols;
              acc[s][C][q] = ok ? av[s][C][q] - acc[s][C][q] : 0.0;
            }
        }
      }
      if (g0 == 0) {
        // ---- diagonal tiles: wave 0 owns tile row 0, wave 1 (or wave 0 again, single-wave launch) tile row 1 ----
        if (wave == 0) {
#pragma unroll
          for (int q = 0; q < 4; ++q) Dga[l15 * 16 + l4 + 4 * q] = acc[0][0][q];
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
          psd_tile_factor(Dga, Msa, Dinv, nba, tol_abs, skip + k0);
#pragma unroll
          for (int e = 0; e < 4; ++e) {                                   // L(0,0) -> matrix
            const int idx = lane + 64 * e, rr = idx >> 4, c = idx & 15;
            if (c <= rr && c < nba && k0 + rr < n) A[pk_row(k0 + rr) + k0 + c] = Dga[rr * 16 + c];
          }
        }
        __syncthreads();                                                  // Msa visible
        constexpr int w1 = 1, s1 = 0;                                     // tile row 1: first slot of wave 1
#pragma unroll
        for (int s = 0; s < PSD_TG; ++s) {
          if (ti[s] >= 1 && ti[s] < ntile) {                              // X(t,0)' = Mta P(t,0)'
            d4 x = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int q = 0; q < 4; ++q) x = __builtin_amdgcn_mfma_f64_16x16x4f64(Msa[(l4 + 4 * q) * 16 + l15], acc[s][0][q], x, 0, 0, 0);
            acc[s][0] = x;
          }
        }
        if (nbb > 0) {
          if (wave == w1 && ntile > 1) {
            d4& x10 = acc[s1][0];
            d4& p11 = acc[s1][1];
#pragma unroll
            for (int q = 0; q < 4; ++q) Xs[(l4 + 4 * q) * 16 + l15] = x10[q];
#pragma unroll
            for (int q = 0; q < 4; ++q) p11 = __builtin_amdgcn_mfma_f64_16x16x4f64(-x10[q], x10[q], p11, 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 4; ++q) Dgb[l15 * 16 + l4 + 4 * q] = p11[q];
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            psd_tile_factor(Dgb, Msb, Dinv + 16, nbb, tol_abs, skip + k0 + 16);
#pragma unroll
            for (int e = 0; e < 4; ++e) {                                 // L(1,1) -> matrix
              const int idx = lane + 64 * e, rr = idx >> 4, c = idx & 15;
              const int gi = k0 + 16 + rr;
              if (c <= rr && c < nbb && gi < n) A[pk_row(gi) + k0 + 16 + c] = Dgb[rr * 16 + c];
            }
          }
          __syncthreads();                                                // Xs, Msb visible
        }
      }
      // ---- rows below the panel's diagonal tiles: second half updated with the first, X(t,1)' = Mtb P(t,1)'; stores ----
#pragma unroll
      for (int s = 0; s < PSD_TG; ++s) {
        if (ti[s] < ntile && ti[s] >= 1) {
          if (g0 != 0) {                                                  // later passes: Mta is long visible
            d4 x = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int q = 0; q < 4; ++q) x = __builtin_amdgcn_mfma_f64_16x16x4f64(Msa[(l4 + 4 * q) * 16 + l15], acc[s][0][q], x, 0, 0, 0);
            acc[s][0] = x;
          }
          const int i = k0 + 16 * ti[s] + l15;
          double* Ai = A + pk_row(i < n ? i : 0);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int cc = k0 + l4 + 4 * q;
            if (i < n && cc < ncols) Ai[cc] = acc[s][0][q];
          }
          if (nbb > 0 && ti[s] >= 2) {
            d4 p = acc[s][1];
#pragma unroll
            for (int q = 0; q < 4; ++q) p = __builtin_amdgcn_mfma_f64_16x16x4f64(-Xs[(l4 + 4 * q) * 16 + l15], acc[s][0][q], p, 0, 0, 0);
            d4 x = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int q = 0; q < 4; ++q) x = __builtin_amdgcn_mfma_f64_16x16x4f64(Msb[(l4 + 4 * q) * 16 + l15], p[q], x, 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const int cc = k0 + 16 + l4 + 4 * q;
              if (i < n && cc < ncols) Ai[cc] = x[q];
            }
          }
        }
      }
    }
    __syncthreads();                                                      // panel stored before the next update reads it; LDS tiles free
    if (early && ncols == n && k0 + 32 < n && k0 >= early_from) {
      bool alldead = true;
      for (int q = 0; q < 32; ++q) alldead = alldead && (skip[k0 + q] != 0);      // LDS, the same for every thread
      if (alldead) {
        const int i1 = k0 + 32;
        const int hw = tid >> 5, t32 = tid & 31, nhw = (int)(blockDim.x >> 5);
        double dm = 0.0;
        for (int ib = i1; ib < n; ib += nhw) {
          const int i = ib + hw;
          double s0 = 0.0, s1 = 0.0;
          if (i < n) {
            const double* Li = A + pk_row(i);
            int j = t32;
            for (; j + 32 < k0; j += 64) { const double l0 = Li[j], l1 = Li[j + 32]; s0 += l0 * l0; s1 += l1 * l1; }
            if (j < k0) { const double l0 = Li[j]; s0 += l0 * l0; }
          }
          double sacc = s0 + s1;
#pragma unroll
          for (int off = 16; off > 0; off >>= 1) sacc += __shfl_xor(sacc, off, 32);
          if (i < n) dm = fmax(dm, A[pk_row(i) + i] - sacc);
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) dm = fmax(dm, __shfl_xor(dm, off, 64));
        double* dred = pan + 1312;                                        // (behind the 1312 doubles the panels use)
        if ((tid & 63) == 0) dred[tid >> 6] = dm;
        __syncthreads();
        double dall = 0.0;
        for (int w = 0; w < nwave; ++w) dall = fmax(dall, dred[w]);
        __syncthreads();
        if (dall <= tol_abs) {
          for (int k = i1 + tid; k < n; k += (int)blockDim.x) skip[k] = 1;
          const size_t e0 = (size_t)i1 * (i1 + 1) / 2, e1 = (size_t)n * (n + 1) / 2;
          for (size_t e = e0 + tid; e < e1; e += blockDim.x) {             // rows >= i1: their entries in columns >= i1
            const int i = tri_row((int)e), j = (int)(e - (size_t)i * (i + 1) / 2);
            if (j >= i1) A[pk_row(i) + j] = 0.0;
          }
          __syncthreads();
          return k0;
        }
      }
    }
  }
  return n;
}

// One lane's double as a wave-uniform value (two v_readlane_b32 into an SGPR pair): the unknown-by-unknown chains of the
// blocked substitutions pass values on this way -- a __shfl is a trip through the LDS crossbar, ~100 cycles per step.
__device__ __forceinline__ double lane_value_f64(double v, int srclane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), srclane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), srclane);
  return __hiloint2double(hi, lo);
}

// Back substitution L' x = y for a packed lower factor, 16 rows at a time, with a one-block look-ahead: wave 0 solves
// block b -- it first takes the contribution of block b+1 (the 16x16 coupling block, four entries per lane) off its 16
// right-hand sides, then runs the 16x16 diagonal solve with the unknowns passed on by lane shuffles -- WHILE the other
// waves subtract block b+1's unknowns from all the rows above block b (column pieces of the factor, coalesced; their
// loads do not depend on anything computed here).  One workgroup barrier and no exposed memory round trip per 16 rows:
// the form without look-ahead (solve, barrier, update everything, barrier) spent ~3.3 us per block, twice this.
// y is consumed (overwritten), x must not alias it; `skip` (optional) marks rows whose unknown is zero.
constexpr int PSD_RPT = 2;      // r-vector entries per thread where a routine keeps them in registers (r <= PSD_RPT * blockDim.x)
__device__ __forceinline__ void packed_back_substitute(const double* Lm, int n, double* y, double* x, const int* skip) {
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int lane = tid & 63, a = lane & 15, g4 = lane >> 4;
  const int nblk = (n + 15) >> 4;
  for (int b = nblk - 1; b >= 0; --b) {
    const int k0 = 16 * b;
    const int nb = (n - k0) < 16 ? (n - k0) : 16;
    const size_t kb = pk_row(k0), krs = (size_t)k0 + 16;     // rows k0 .. k0+15 (one 16-row block): row k0 + q starts at kb + q * krs
    const bool below = b + 1 < nblk;                         // block b+1 exists: its unknowns x[k0+16 ..] were stored before the last barrier
    const int nbn = below ? ((n - k0 - 16) < 16 ? (n - k0 - 16) : 16) : 0;
    const size_t kbn = kb + 16 * krs, krn = krs + 16;        // rows of block b+1
    if (tid < 64) {
      // lane a owns unknown k0 + a and column a of the diagonal block: Lc[q] = L(k0 + q, k0 + a), q >= a
      // (unconditional loads from rows clamped into the block, selected afterwards: all 20 in flight at once)
      double Lc[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) Lc[q] = Lm[kb + (q < nb ? q : nb - 1) * krs + k0 + a];
      // coupling with block b+1: lane (g4, a) takes rows k0+16 + 4 g4 .. + 3 of column k0 + a
      double cq[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) cq[i] = Lm[(nbn > 0 ? kbn + (4 * g4 + i < nbn ? 4 * g4 + i : nbn - 1) * krn : kb) + k0 + a];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = 0; q < 16; ++q) Lc[q] = (q >= a && q < nb) ? Lc[q] : 0.0;
#pragma unroll
      for (int i = 0; i < 4; ++i) cq[i] = (4 * g4 + i < nbn && a < nb) ? cq[i] : 0.0;
      const bool dead = a >= nb || (skip != nullptr && skip[k0 + a] != 0);
      double inv = 0.0;
#pragma unroll
      for (int q = 0; q < 16; ++q) if (q == a) inv = dead ? 0.0 : 1.0 / Lc[q];
      double cs = 0.0;
#pragma unroll
      for (int i = 0; i < 4; ++i) cs += cq[i] * ((4 * g4 + i < nbn) ? x[k0 + 16 + 4 * g4 + i] : 0.0);
      cs += __shfl_xor(cs, 16, 64);
      cs += __shfl_xor(cs, 32, 64);
      double v = (a < nb) ? y[k0 + a] - cs : 0.0;
#pragma unroll
      for (int q = 15; q >= 0; --q) {
        const double xq = lane_value_f64(v * inv, q);   // final once every row below q has been subtracted (the four 16-lane groups hold the same values)
        if (a < q) v -= Lc[q] * xq;
      }
      if (tid < nb) x[k0 + a] = v * inv;
    } else if (below) {
      // rows above block b: y[j] -= sum_q L(k0+16+q, j) x[k0+16+q], j < k0 (block b's own rows get theirs from wave 0)
      for (int j = tid - 64; j < k0; j += nthr - 64) {
        double Lu[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) Lu[q] = Lm[kbn + (q < nbn ? q : nbn - 1) * krn + j];
        __builtin_amdgcn_sched_barrier(0);
        double sacc = 0.0;
#pragma unroll
        for (int q = 0; q < 16; ++q) sacc += Lu[q] * ((q < nbn) ? x[k0 + 16 + q] : 0.0);
        y[j] -= sacc;
      }
    }
    __syncthreads();
  }
}

__device__ __forceinline__ double block_sum(double v, double* red);       // defined below

// L y = rhs over the rows that are not skipped (y = 0 on skipped ones) for a packed lower factor, 16 rows at a time, with
// a one-block look-ahead: while wave 0 solves block b (right-hand sides minus the partial sums formed one block earlier
// minus the 16x16 coupling with block b-1, then the diagonal solve with lane shuffles), the other half waves already
// form the dot products of block b+1's rows with the y known so far (columns < 16 b; coalesced 256-byte pieces of a
// row).  One workgroup barrier per 16 rows, the row loads off wave 0's path.  y must not alias rhs; `red` is not used any more.
__device__ __forceinline__ void packed_forward_substitute(const double* Lm, int n, const double* rhs, double* y,
                                                          const int* skip, double* red) {
  (void)red;
  __shared__ double fsub_part[2][16];                    // partial sums of the current / the next block (one instance per kernel)
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int hw = tid >> 5, t32 = tid & 31, nhw = nthr >> 5;
  const int lane = tid & 63, a = lane & 15, g4 = lane >> 4;
  const int nblk = (n + 15) >> 4;
  if (tid < 16) fsub_part[0][tid] = 0.0;
  __syncthreads();
  for (int b = 0; b < nblk; ++b) {
    const int k0 = 16 * b;
    const int nb = (n - k0) < 16 ? (n - k0) : 16;
    const int cur = b & 1, nxt = cur ^ 1;
    const size_t kb = pk_row(k0), krs = (size_t)k0 + 16;     // rows k0 .. k0+15 (one 16-row block): row k0 + q starts at kb + q * krs
    if (tid < 64) {
      const int ac = a < nb ? a : nb - 1;                  // (unconditional loads from a row clamped into the block, selected afterwards)
      double Lr[16];                                       // row a of the diagonal block, Lr[q] = L(k0 + a, k0 + q), q <= a
#pragma unroll
      for (int q = 0; q < 16; ++q) Lr[q] = Lm[kb + ac * krs + k0 + q];
      double cq[4];                                        // coupling with block b-1: lane (g4, a) takes columns k0-16 + 4 g4 .. + 3 of row k0 + a
#pragma unroll
      for (int i = 0; i < 4; ++i) cq[i] = Lm[kb + ac * krs + (b > 0 ? k0 - 16 : 0) + 4 * g4 + i];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = 0; q < 16; ++q) Lr[q] = (q <= a && a < nb) ? Lr[q] : 0.0;
#pragma unroll
      for (int i = 0; i < 4; ++i) cq[i] = (b > 0 && a < nb) ? cq[i] : 0.0;
      const bool dead = a >= nb || (skip != nullptr && skip[k0 + a] != 0);
      double inv = 0.0;
#pragma unroll
      for (int q = 0; q < 16; ++q) if (q == a) inv = dead ? 0.0 : 1.0 / Lr[q];
      double cs = 0.0;
#pragma unroll
      for (int i = 0; i < 4; ++i) cs += cq[i] * ((b > 0) ? y[k0 - 16 + 4 * g4 + i] : 0.0);
      cs += __shfl_xor(cs, 16, 64);
      cs += __shfl_xor(cs, 32, 64);
      double v = (a < nb) ? rhs[k0 + a] - fsub_part[cur][a] - cs : 0.0;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const double yq = lane_value_f64(v * inv, q);   // final once every row above q has been subtracted (the four 16-lane groups hold the same values)
        if (a > q) v -= Lr[q] * yq;
      }
      if (tid < nb) y[k0 + a] = v * inv;
    } else if (b + 1 < nblk) {
      // block b+1, columns < k0: one half wave per row
      const size_t kbn = kb + 16 * krs, krn = krs + 16;
      for (int h = hw - 2; h < 16; h += nhw - 2) {
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
        if (k0 + 16 + h < n) {
          const double* La = Lm + kbn + h * krn;
          int j = t32;
          for (; j + 96 < k0; j += 128) {                  // four loads in flight per lane
            const double l0 = La[j], l1 = La[j + 32], l2 = La[j + 64], l3 = La[j + 96];
            s0 += l0 * y[j]; s1 += l1 * y[j + 32]; s2 += l2 * y[j + 64]; s3 += l3 * y[j + 96];
          }
          for (; j < k0; j += 32) s0 += La[j] * y[j];
        }
        double sacc = (s0 + s1) + (s2 + s3);
#pragma unroll
        for (int off = 16; off > 0; off >>= 1) sacc += __shfl_xor(sacc, off, 32);
        if (t32 == 0) fsub_part[nxt][h] = sacc;
      }
    }
    __syncthreads();
  }
}

// Products with blocks of a packed lower matrix.
// Rows: out(i, sum_{j0 <= j < jend(i)} L(row0 + i, j) x[j - j0]) for i < nrows -- one 32-lane half wave per row: coalesced
// 256-byte pieces of the row, four loads in flight per lane, shuffle reduction.  (One THREAD per row streams 64 rows per
// wave through a 32 KB L1 that cannot hold them: every 8 bytes come from L2 again.)  `out` runs on one lane per row.
template <class EndF, class OutF>
__device__ __forceinline__ void packed_rows_times(const double* Lm, int row0, int nrows, int j0, const double* x,
                                                  EndF&& jend, OutF&& out) {
  const int hw = threadIdx.x >> 5, t32 = threadIdx.x & 31, nhw = blockDim.x >> 5;
  for (int ib = 0; ib < nrows; ib += nhw) {
    const int i = ib + hw;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    if (i < nrows) {
      const double* Li = Lm + pk_row(row0 + i);
      const double* xv = x - j0;
      const int je = jend(i);
      int j = j0 + t32;
      for (; j + 96 < je; j += 128) {
        const double l0 = Li[j], l1 = Li[j + 32], l2 = Li[j + 64], l3 = Li[j + 96];
        s0 += l0 * xv[j]; s1 += l1 * xv[j + 32]; s2 += l2 * xv[j + 64]; s3 += l3 * xv[j + 96];
      }
      for (; j < je; j += 32) s0 += Li[j] * xv[j];
    }
    double sacc = (s0 + s1) + (s2 + s3);
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) sacc += __shfl_xor(sacc, off, 32);
    if (t32 == 0 && i < nrows) out(i, sacc);
  }
}
// Columns: out(k, sum_{ibeg(k) <= i < nrows} L(row0 + i, col0 + k) v(i)) for k < ncols -- one thread per column (coalesced
// across the threads), four independent loads in flight.
template <class BegF, class VF, class OutF>
__device__ __forceinline__ void packed_cols_times(const double* Lm, int row0, int nrows, int col0, int ncols,
                                                  BegF&& ibeg, VF&& v, OutF&& out) {
  for (int k = threadIdx.x; k < ncols; k += blockDim.x) {
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    int i = ibeg(k);
    for (; i + 3 < nrows; i += 4) {
      const size_t ri = (size_t)(row0 + i);
      const double l0 = Lm[pk_row(ri) + col0 + k], l1 = Lm[pk_row(ri + 1) + col0 + k];
      const double l2 = Lm[pk_row(ri + 2) + col0 + k], l3 = Lm[pk_row(ri + 3) + col0 + k];
      s0 += l0 * v(i); s1 += l1 * v(i + 1); s2 += l2 * v(i + 2); s3 += l3 * v(i + 3);
    }
    for (; i < nrows; ++i) {
      const size_t ri = (size_t)(row0 + i);
      s0 += Lm[pk_row(ri) + col0 + k] * v(i);
    }
    out(k, (s0 + s1) + (s2 + s3));
  }
}

// T = C' W C for the lower-triangular block C(i, a) = L(row0 + i, row0 + a), a <= i < nR, W = diag(w): packed lower
// triangle of T by v_mfma_f64_16x16x4, the row index i as the contraction index (4 rows per instruction; both operands
// are 128-byte pieces of packed rows).  A skipped pivot (skipd[a] != 0) has a zero column in L: its row and column of T
// come out zero and the diagonal entry is set to one.  Work items = (tile row A, group of up to four tile columns).
// `ncol` <= nR: only the leading ncol columns of C (rows / columns of T) are formed -- the rest are known to be zero columns.
__device__ __forceinline__ void packed_weighted_gram_mfma(const double* Lm, int row0, int nR, const double* w,
                                                          const int* skipd, double* T, int ncol = -1) {
  const int lane = threadIdx.x & 63, l15 = lane & 15, l4 = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), nwave = (int)(blockDim.x >> 6);
  if (ncol < 0) ncol = nR;
  const int nt = (ncol + 15) >> 4;
  int item = 0;
  for (int A = nt - 1; A >= 0; --A) {                  // longest rows first
    for (int B0 = 0; B0 <= A; B0 += 4, ++item) {
      if (item % nwave != wave) continue;               // wave-uniform
      d4 acc[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) acc[g] = d4{0.0, 0.0, 0.0, 0.0};
      const int a = 16 * A + l15;
      for (int i0 = 16 * A; i0 < nR; i0 += 4) {
        const int i = i0 + l4;
        const size_t ri = (size_t)(row0 + (i < nR ? i : nR - 1));
        const double* Li = Lm + pk_row(ri) + row0;
        const double av = (i < nR && a <= i && a < ncol) ? Li[a] * w[i] : 0.0;
        double bv[4];                                    // all loads of the step before the first MFMA
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int bcol = 16 * ((B0 + g <= A) ? B0 + g : A) + l15;
          bv[g] = (i < nR && bcol <= i && bcol < ncol) ? Li[bcol] : 0.0;
        }
#pragma unroll
        for (int g = 0; g < 4; ++g)
          if (B0 + g <= A) acc[g] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv[g], acc[g], 0, 0, 0);
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        if (B0 + g <= A) {
          const int bcol = 16 * (B0 + g) + l15;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int ar = 16 * A + l4 + 4 * q;
            if (ar < ncol && bcol <= ar) T[pk_row(ar) + bcol] = (ar == bcol && skipd[ar]) ? 1.0 : acc[g][q];
          }
        }
      }
    }
  }
}

// S(i, j) = A(row0 + i, row0 + j) - sum_{k < nk} L(row0 + i, k) L(row0 + j, k), j <= i < nB: the Schur complement of
// a trailing block behind nk factored columns (packed lower triangle of S), rows times rows on the matrix pipe with the
// column index as the contraction index: a lane loads 4 consecutive entries of "its" row per 16 columns.
__device__ __forceinline__ void packed_schur_mfma(const double* Lm, int row0, int nB, int nk, double* S) {
  const int lane = threadIdx.x & 63, l15 = lane & 15, l4 = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), nwave = (int)(blockDim.x >> 6);
  const int nt = (nB + 15) >> 4;
  auto rowp = [&](int i) -> const double* { const size_t ri = (size_t)(row0 + (i < nB ? i : nB - 1)); return Lm + pk_row(ri); };
  int item = 0;
  for (int A = nt - 1; A >= 0; --A) {
    for (int B0 = 0; B0 <= A; B0 += 4, ++item) {
      if (item % nwave != wave) continue;               // wave-uniform
      d4 acc[4];
      const double* rb[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) { acc[g] = d4{0.0, 0.0, 0.0, 0.0}; rb[g] = rowp(16 * (B0 + g) + l15) + 4 * l4; }
      const double* ra = rowp(16 * A + l15) + 4 * l4;
      for (int k0 = 0; k0 < nk; k0 += 16) {
        const int kk = k0 + 4 * l4;
        d2u8 a0 = *reinterpret_cast<const d2u8*>(ra + k0), a1 = *reinterpret_cast<const d2u8*>(ra + k0 + 2);
        if (k0 + 16 > nk) {                              // last chunk: entries past the factored columns do not count
          a0[0] = (kk < nk) ? a0[0] : 0.0; a0[1] = (kk + 1 < nk) ? a0[1] : 0.0;
          a1[0] = (kk + 2 < nk) ? a1[0] : 0.0; a1[1] = (kk + 3 < nk) ? a1[1] : 0.0;
        }
        d2u8 b0[4], b1[4];                               // all loads of the chunk before the first MFMA
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          b0[g] = *reinterpret_cast<const d2u8*>(rb[g] + k0);
          b1[g] = *reinterpret_cast<const d2u8*>(rb[g] + k0 + 2);
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          if (B0 + g <= A) {
            if (k0 + 16 > nk) {                          // (a zero on one side is not enough: the other side may hold anything)
              b0[g][0] = (kk < nk) ? b0[g][0] : 0.0; b0[g][1] = (kk + 1 < nk) ? b0[g][1] : 0.0;
              b1[g][0] = (kk + 2 < nk) ? b1[g][0] : 0.0; b1[g][1] = (kk + 3 < nk) ? b1[g][1] : 0.0;
            }
            acc[g] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[0], b0[g][0], acc[g], 0, 0, 0);
            acc[g] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[1], b0[g][1], acc[g], 0, 0, 0);
            acc[g] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[0], b1[g][0], acc[g], 0, 0, 0);
            acc[g] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[1], b1[g][1], acc[g], 0, 0, 0);
          }
        }
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        if (B0 + g <= A) {
          const int j = 16 * (B0 + g) + l15;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int i = 16 * A + l4 + 4 * q;
            if (i < nB && j <= i) {
              const size_t ri = (size_t)(row0 + i);
              S[pk_row(i) + j] = Lm[pk_row(ri) + row0 + j] - acc[g][q];
            }
          }
        }
      }
    }
  }
}

// z = H (H' x) for the implicit block-Hankel matrix H (rows rho = k*nch + ch, columns i < c); x and z are r-vectors in
// COMPONENT order (LDS, z must not alias x).  With the trajectory stored channel-interleaved, column i of H is the
// CONTIGUOUS window xflat[i*nch .. i*nch + r): the trajectory is streamed ONCE through LDS in chunks of time steps and both
// products are formed from the chunk -- alpha_i = <window_i, x> by one 32-lane half wave per column (conflict-free
// reads, shuffle reduction), then z += window_i * alpha_i with one thread per component.  `pan`: PSD_PAN doubles of LDS.
// (alpha itself, c numbers, never leaves the chip.)
// Register-blocked form for channel counts that divide the wave size (2, 4, 8, 16, 32: every shape of BASELINE.json).  Both
// products are correlations along the time axis, so a thread that keeps FOUR consecutive positions of one channel in
// registers needs one new trajectory entry per step instead of four:
//   alpha_{i0+j} = sum_k sum_ch X[i0+j+k][ch] x[k nch + ch]     thread = (4 columns i0.., channel):   loop over k,  2 LDS loads per 4 FMAs,
//                                                                then a shuffle sum over the nch lanes of a column group
//   z[(k0+j) nch + ch] = sum_i X[i+k0+j][ch] alpha_i             thread = (4 offsets k0.., channel, a part of the i range): loop over i,
//                                                                2 loads per 4 FMAs; accumulators live in registers across ALL chunks
// (the one-column-per-half-wave form above moves 2 LDS loads per FMA: 38 MB of LDS traffic per call at the cfg-5 size, 320 us
// measured; this form: 160 us).  The next chunk of the trajectory is fetched into registers while the current one is worked on.
// Returns false (nothing done) for shapes it does not cover.
__device__ __forceinline__ bool hankel_normal_times_blocked(const KParams& P, const double* __restrict__ ud,
                                                            const double* __restrict__ yd, const double* x, double* z, double* pan) {
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int m = P.m, p = P.p, nch = P.nch, c = P.c, r = P.r, Ln = P.Ln;
  if (nch < 2 || nch > 32 || (64 % nch) != 0) return false;
  const int lg = 31 - __clz(nch);
  const int KG = (Ln + 3) >> 2;                                          // groups of four time offsets
  const int TC = ((PSD_PAN - (Ln + 3) * nch) / (nch + 1)) & ~3;           // columns per chunk: (TC + Ln + 3) rows + TC alphas
  int NP = nthr / (KG * nch);                                            // parts of a chunk's column range in the z product
  if (NP > PSD_PAN / (KG * 4 * nch)) NP = PSD_PAN / (KG * 4 * nch);        // (the parts meet in `pan` at the end)
  constexpr int SR = 8;                                                  // staging registers per thread
  const int nrows = TC + Ln + 3;
  if (TC < 4 || NP < 1 || nrows * nch > SR * nthr) return false;         // (workgroup-uniform)
  double* xc = pan;                                                      // xc[row][ch], row = time step - t0, nrows rows
  double* al = pan + nrows * nch;                                        // alpha of the chunk
  double stg[SR];
  // (branch-free: one unconditional load per entry from an address that is always valid, zeroed afterwards where the entry
  //  lies outside the trajectory -- with a branch per entry the loads of a chunk wait for one another)
  auto fetch = [&](int t0) __attribute__((always_inline)) {
    int tq = tid;
    asm volatile("" : "+v"(tq));          // opaque per chunk: otherwise the SR source addresses are hoisted out of the chunk loop,
                                          // spilled, and every load of the chunk waits for a scratch reload of its own address
    const long long dyu = reinterpret_cast<const char*>(yd) - reinterpret_cast<const char*>(ud);   // (one flat address space)
    double keep[SR];
#pragma unroll
    for (int e = 0; e < SR; ++e) {
      int i = tq + e * nthr;
      keep[e] = (i < nrows * nch && t0 + (i >> lg) < P.N) ? 1.0 : 0.0;
      i = i < nrows * nch ? i : nrows * nch - 1;
      const int t = t0 + (i >> lg), ch = i & (nch - 1);
      const int tc = t < P.N ? t : P.N - 1;
      const long long ou = ((long long)tc * m + ch) * 8, oy = dyu + ((long long)tc * p + (ch - m)) * 8;
      const long long off = (ch < m) ? ou : oy;
      stg[e] = *reinterpret_cast<const double*>(reinterpret_cast<const char*>(ud) + off);
    }
    __builtin_amdgcn_sched_barrier(0);                      // all SR loads are in flight before the first of them is touched
#pragma unroll
    for (int e = 0; e < SR; ++e) stg[e] *= keep[e];
  };
  // z product: this thread's (part, offset group, channel) and its four accumulators
  const int zt = tid;
  const bool zon = zt < NP * KG * nch;
  const int zpart = zt / (KG * nch), zrem = zt - zpart * (KG * nch), zk0 = 4 * (zrem >> lg), zch = zrem & (nch - 1);
  double z0 = 0.0, z1 = 0.0, z2 = 0.0, z3 = 0.0;
  fetch(0);
  for (int t0 = 0; t0 < c; t0 += TC) {
    const int nt = (c - t0) < TC ? (c - t0) : TC;
    __syncthreads();                                                      // previous chunk consumed
#pragma unroll
    for (int e = 0; e < SR; ++e) {
      const int i = tid + e * nthr;
      if (i < nrows * nch) xc[i] = stg[e];
    }
    __syncthreads();
    fetch(t0 + TC);                                                       // in flight under the two products below
    // ---- alpha of the chunk
    const int ng = (nt + 3) >> 2;
    for (int tb = 0; tb < ng * nch; tb += nthr) {                         // (whole waves take part in the shuffles)
      const int t = tb + tid;
      const bool on = t < ng * nch;
      const int cg = on ? (t >> lg) : 0, ch = t & (nch - 1);
      const double* wp = xc + (4 * cg) * nch + ch;
      const double* xp = x + ch;
      double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
      double w0 = wp[0], w1 = wp[nch], w2 = wp[2 * nch];
      int k = 0;
      for (; k + 4 <= Ln; k += 4) {                                       // the window rotates through w0..w3
        // (all eight loads of the four steps first: issued one pair at a time, every step waits a full LDS round trip)
        const double n3 = wp[(k + 3) * nch], n4 = wp[(k + 4) * nch], n5 = wp[(k + 5) * nch], n6 = wp[(k + 6) * nch];
        const double x0 = xp[k * nch], x1 = xp[(k + 1) * nch], x2 = xp[(k + 2) * nch], x3 = xp[(k + 3) * nch];
        __builtin_amdgcn_sched_barrier(0);
        a0 = fma(w0, x0, a0); a1 = fma(w1, x0, a1); a2 = fma(w2, x0, a2); a3 = fma(n3, x0, a3);
        a0 = fma(w1, x1, a0); a1 = fma(w2, x1, a1); a2 = fma(n3, x1, a2); a3 = fma(n4, x1, a3);
        a0 = fma(w2, x2, a0); a1 = fma(n3, x2, a1); a2 = fma(n4, x2, a2); a3 = fma(n5, x2, a3);
        a0 = fma(n3, x3, a0); a1 = fma(n4, x3, a1); a2 = fma(n5, x3, a2); a3 = fma(n6, x3, a3);
        w0 = n4; w1 = n5; w2 = n6;
      }
      for (; k < Ln; ++k) {
        const double w3 = wp[(k + 3) * nch];
        const double xv = xp[k * nch];
        a0 = fma(w0, xv, a0); a1 = fma(w1, xv, a1); a2 = fma(w2, xv, a2); a3 = fma(w3, xv, a3);
        w0 = w1; w1 = w2; w2 = w3;
      }
      if (!on) { a0 = 0.0; a1 = 0.0; a2 = 0.0; a3 = 0.0; }
      for (int off = nch >> 1; off > 0; off >>= 1) {
        a0 += __shfl_xor(a0, off, 64); a1 += __shfl_xor(a1, off, 64); a2 += __shfl_xor(a2, off, 64); a3 += __shfl_xor(a3, off, 64);
      }
      if (on && ch == 0) { al[4 * cg] = a0; al[4 * cg + 1] = a1; al[4 * cg + 2] = a2; al[4 * cg + 3] = a3; }   // (4 cg + 3 < TC: TC is a multiple of 4)
    }
    __syncthreads();
    // ---- z += H[:, chunk] alpha[chunk]
    if (zon) {
      const int per = (nt + NP - 1) / NP;
      const int ia = zpart * per;
      const int ib = (ia + per) < nt ? (ia + per) : nt;
      if (ia < ib) {
        const double* wp = xc + (ia + zk0) * nch + zch;
        double w0 = wp[0], w1 = wp[nch], w2 = wp[2 * nch];
        int i = ia;
        const double* ap = al + ia;
        int q = 0;
        for (; i + 4 <= ib; i += 4, q += 4) {
          const double n3 = wp[(q + 3) * nch], n4 = wp[(q + 4) * nch], n5 = wp[(q + 5) * nch], n6 = wp[(q + 6) * nch];
          const double v0 = ap[q], v1 = ap[q + 1], v2 = ap[q + 2], v3 = ap[q + 3];
          __builtin_amdgcn_sched_barrier(0);
          z0 = fma(w0, v0, z0); z1 = fma(w1, v0, z1); z2 = fma(w2, v0, z2); z3 = fma(n3, v0, z3);
          z0 = fma(w1, v1, z0); z1 = fma(w2, v1, z1); z2 = fma(n3, v1, z2); z3 = fma(n4, v1, z3);
          z0 = fma(w2, v2, z0); z1 = fma(n3, v2, z1); z2 = fma(n4, v2, z2); z3 = fma(n5, v2, z3);
          z0 = fma(n3, v3, z0); z1 = fma(n4, v3, z1); z2 = fma(n5, v3, z2); z3 = fma(n6, v3, z3);
          w0 = n4; w1 = n5; w2 = n6;
        }
        for (; i < ib; ++i, ++q) {
          const double w3 = wp[(q + 3) * nch];
          const double av = ap[q];
          z0 = fma(w0, av, z0); z1 = fma(w1, av, z1); z2 = fma(w2, av, z2); z3 = fma(w3, av, z3);
          w0 = w1; w1 = w2; w2 = w3;
        }
      }
    }
  }
  __syncthreads();                                                        // the last chunk is consumed: `pan` takes the parts
  if (zon) {
    double* zp = pan + zpart * (KG * 4 * nch) + zk0 * nch + zch;
    zp[0] = z0; zp[nch] = z1; zp[2 * nch] = z2; zp[3 * nch] = z3;
  }
  __syncthreads();
  for (int rho = tid; rho < r; rho += nthr) {
    double sacc = 0.0;
    for (int q = 0; q < NP; ++q) sacc += pan[q * (KG * 4 * nch) + rho];
    z[rho] = sacc;
  }
  __syncthreads();
  return true;
}

__device__ __forceinline__ void hankel_normal_times(const KParams& P, const double* __restrict__ ud,
                                                    const double* __restrict__ yd, const double* x, double* z, double* pan) {
  if (hankel_normal_times_blocked(P, ud, yd, x, z, pan)) return;
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int hw = tid >> 5, t32 = tid & 31, nhw = nthr >> 5;
  const int m = P.m, p = P.p, nch = P.nch, c = P.c, r = P.r;
  const int TC = ((PSD_PAN - r) / (nch + 1)) & ~3;                        // columns per chunk: (TC + Ln) * nch + TC <= PSD_PAN
  double* xc = pan;                                                         // xc[(t - t0) * nch + ch], t0 <= t < t0 + TC + Ln - 1
  double* al = pan + (TC + P.Ln) * nch;                                     // alpha of the chunk
  double zacc[PSD_RPT];
#pragma unroll
  for (int e = 0; e < PSD_RPT; ++e) zacc[e] = 0.0;
  for (int t0 = 0; t0 < c; t0 += TC) {
    const int nt = (c - t0) < TC ? (c - t0) : TC;
    const int nload = (nt + P.Ln - 1) * nch;
    __syncthreads();                                                        // previous chunk consumed
    for (int i = tid; i < nload; i += nthr) {
      const int tt = i / nch, ch = i - tt * nch;
      xc[i] = (ch < m) ? ud[(long long)(t0 + tt) * m + ch] : yd[(long long)(t0 + tt) * p + (ch - m)];
    }
    __syncthreads();
    for (int ib = 0; ib < nt; ib += nhw) {
      const int i = ib + hw;
      double s0 = 0.0, s1 = 0.0;
      if (i < nt) {
        const double* win = xc + i * nch;
        int e = t32;
        for (; e + 32 < r; e += 64) { s0 += win[e] * x[e]; s1 += win[e + 32] * x[e + 32]; }
        if (e < r) s0 += win[e] * x[e];
      }
      double sacc = s0 + s1;
#pragma unroll
      for (int off = 16; off > 0; off >>= 1) sacc += __shfl_xor(sacc, off, 32);
      if (t32 == 0 && i < nt) al[i] = sacc;
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < PSD_RPT; ++e) {
      const int rho = tid + e * nthr;
      if (rho < r) {
        const double* colp = xc + rho;
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
        int i = 0;
        for (; i + 3 < nt; i += 4) {
          a0 += colp[i * nch] * al[i]; a1 += colp[(i + 1) * nch] * al[i + 1];
          a2 += colp[(i + 2) * nch] * al[i + 2]; a3 += colp[(i + 3) * nch] * al[i + 3];
        }
        for (; i < nt; ++i) a0 += colp[i * nch] * al[i];
        zacc[e] += (a0 + a1) + (a2 + a3);
      }
    }
  }
#pragma unroll
  for (int e = 0; e < PSD_RPT; ++e) {
    const int rho = tid + e * nthr;
    if (rho < r) z[rho] = zacc[e];
  }
  __syncthreads();
}

// Rows [t0, t0 + nrows) of the channel-interleaved trajectory into dst[row * nch + ch] (zeros beyond the last time step),
// by the whole workgroup.  Branch-free: one unconditional load per entry from an address that is always valid (a select
// between u_d and y_d or a range test per entry turns into one exec-masked block per entry, whose loads then wait for one
// another), SR entries per thread in flight at once; no barrier inside.
template <int SR>
__device__ __forceinline__ void stage_trajectory(const KParams& P, const double* __restrict__ ud, const double* __restrict__ yd,
                                                 int t0, int nrows, double* dst) {
  const int nthr = blockDim.x, m = P.m, p = P.p, nch = P.nch, total = nrows * nch;
  const long long dyu = reinterpret_cast<const char*>(yd) - reinterpret_cast<const char*>(ud);   // (one flat address space)
  const bool pow2 = (nch & (nch - 1)) == 0;
  const int lg = 31 - __clz(nch);
  for (int base = 0; base < total; base += SR * nthr) {
    int tq = threadIdx.x;
    asm volatile("" : "+v"(tq));                       // (opaque per round: keeps the SR source addresses from being hoisted and spilled)
    double v[SR], keep[SR];
#pragma unroll
    for (int e = 0; e < SR; ++e) {
      int i = base + tq + e * nthr;
      i = i < total ? i : total - 1;
      const int row = pow2 ? (i >> lg) : i / nch, ch = i - row * nch;
      const int t = t0 + row, tc = t < P.N ? t : P.N - 1;
      keep[e] = t < P.N ? 1.0 : 0.0;
      const long long ou = ((long long)tc * m + ch) * 8, oy = dyu + ((long long)tc * p + (ch - m)) * 8;
      v[e] = *reinterpret_cast<const double*>(reinterpret_cast<const char*>(ud) + ((ch < m) ? ou : oy));
    }
    __builtin_amdgcn_sched_barrier(0);                 // all SR loads are in flight before the first is touched
#pragma unroll
    for (int e = 0; e < SR; ++e) {
      const int i = base + tq + e * nthr;
      if (i < total) dst[i] = v[e] * keep[e];
    }
  }
}

// Packed lower triangle of G = H H' for the block-Hankel H of one instance, through the Hankel structure (as in the
// cold kernel): with components (k, a) = (time offset, channel),
//   G((k+d, a), (k, b)) = C_d(a,b) + sum_{j<k} ( x_a[j+c+d] x_b[j+c] - x_a[j+d] x_b[j] ),   C_d(a,b) = sum_{t<c} x_a[t+d] x_b[t],
// so only the Ln*nch^2 lag sums need the full length-c dot product (cfg 5: 19 M instead of 363 M multiply-adds).
// `Ctab` holds the Ln*nch^2 lag sums (any scratch of that size), `pan` = PSD_PAN doubles of LDS;
// `iperm` maps component rho = k*nch + ch to its row in G (nullptr: identity; written by the caller BEFORE the call).
__device__ __forceinline__ void hankel_gram_packed(const KParams& P, const double* __restrict__ ud,
                                                   const double* __restrict__ yd, double* G, double* Ctab,
                                                   const int* iperm, double* pan) {
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int m = P.m, p = P.p, nch = P.nch, c = P.c;
  const int nlag = P.Ln * nch * nch;
  auto xat = [&](int a, int t) -> double { return (a < m) ? ud[(long long)t * m + a] : yd[(long long)t * p + (a - m)]; };
  // lag sums on the matrix pipe: C_d = X_d' X_0 (X_d: the trajectory shifted by d time steps, one channel per column)
  // is a 16x16 tile per lag and channel-tile pair with the time index as the contraction index.  The trajectory is
  // streamed through LDS in chunks of time steps (the LDS scratch of the Cholesky is free at this point); a wave keeps
  // up to HG_SL lags in accumulators and shares the X_0 operand between them.
  constexpr int HG_SL = 5;
  const int TCH = ((PSD_PAN / nch) - P.Ln) & ~3;                            // time steps per chunk that fit with the lag overlap
  double* xc = pan;                                                         // xc[(t - t0) * nch + ch], t0 <= t < t0 + TCH + Ln
  const int lane = tid & 63, l15 = lane & 15, l4 = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), nwave = nthr >> 6;
  const int nat = (nch + 15) >> 4;
  for (int dg = 0; dg < P.Ln; dg += nwave * HG_SL)
    for (int at = 0; at < nat; ++at)
      for (int bt = 0; bt < nat; ++bt) {
        d4 acc[HG_SL];
#pragma unroll
        for (int sl = 0; sl < HG_SL; ++sl) acc[sl] = d4{0.0, 0.0, 0.0, 0.0};
        const int ca = (16 * at + l15 < nch) ? 16 * at + l15 : nch - 1;     // clamped: entries past nch are not stored
        const int cb = (16 * bt + l15 < nch) ? 16 * bt + l15 : nch - 1;
        for (int t0 = 0; t0 < c; t0 += TCH) {
          const int nt = (c - t0) < TCH ? (c - t0) : TCH;                    // terms of this chunk
          const int nload = nt + P.Ln - 1;                                  // time steps needed (x_a[t+d], d < Ln)
          __syncthreads();
          stage_trajectory<4>(P, ud, yd, t0, nload, xc);
          __syncthreads();
          for (int s4 = 0; s4 < nt; s4 += 4) {
            const int t = s4 + l4;
            const double bv = (t < nt) ? xc[t * nch + cb] : 0.0;
#pragma unroll
            for (int sl = 0; sl < HG_SL; ++sl) {
              const int d = dg + wave + nwave * sl;                         // wave-uniform
              if (d < P.Ln) {
                const int row = (t + d < nload) ? t + d : nload - 1;        // masked terms: any finite value
                acc[sl] = __builtin_amdgcn_mfma_f64_16x16x4f64(xc[row * nch + ca], bv, acc[sl], 0, 0, 0);
              }
            }
          }
        }
#pragma unroll
        for (int sl = 0; sl < HG_SL; ++sl) {
          const int d = dg + wave + nwave * sl;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int a = 16 * at + l4 + 4 * q, bb = 16 * bt + l15;
            if (d < P.Ln && a < nch && bb < nch) Ctab[(d * nch + a) * nch + bb] = acc[sl][q];
          }
        }
      }
  __syncthreads();
  // one (lag, channel pair) diagonal per thread-iteration, walked with the O(1) window update; every unordered
  // pair of components is met exactly once (lag 0: channel pairs a >= b only).  The walk only touches the first and the
  // last Ln - 1 rows of the window range: both pieces are staged in LDS (one dependent L2 round trip per step otherwise).
  const int nw = P.Ln - 1;
  double* xh = pan;                                                         // rows 0 .. Ln-2
  double* xt = pan + nw * nch;                                              // rows c .. c+Ln-2
  const bool walk_lds = 2 * nw * nch <= PSD_PAN;
  if (walk_lds) {
    stage_trajectory<4>(P, ud, yd, 0, nw, xh);
    stage_trajectory<4>(P, ud, yd, c, nw, xt);
    __syncthreads();
  }
  for (int e = tid; e < nlag; e += nthr) {
    const int d = e / (nch * nch), ab = e - d * nch * nch, a = ab / nch, bb = ab - a * nch;
    if (d == 0 && a < bb) continue;
    double s = Ctab[e];
    for (int k = 0; k + d < P.Ln; ++k) {
      if (k > 0) {
        if (walk_lds) s += xt[(k - 1 + d) * nch + a] * xt[(k - 1) * nch + bb] - xh[(k - 1 + d) * nch + a] * xh[(k - 1) * nch + bb];
        else s += xat(a, k - 1 + c + d) * xat(bb, k - 1 + c) - xat(a, k - 1 + d) * xat(bb, k - 1);
      }
      const int ci = (k + d) * nch + a, cj = k * nch + bb;
      const int pi = iperm ? iperm[ci] : ci, pj = iperm ? iperm[cj] : cj;
      const int hi = pi > pj ? pi : pj, lo = pi > pj ? pj : pi;
      G[pk_row(hi) + lo] = s;
    }
  }
}

// Sum of one value per thread over the workgroup (all threads get it); `red` holds >= nthr/64 doubles.
__device__ __forceinline__ double block_sum(double v, double* red) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  double t = 0.0;
  for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += red[w];
  return t;
}

// Maximum of one value per thread over the workgroup (all threads get it).
__device__ __forceinline__ double block_max(double v, double* red) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off, 64));
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  double t = red[0];
  for (int w = 1; w < (int)(blockDim.x >> 6); ++w) t = fmax(t, red[w]);
  return t;
}

#ifndef DDMPC_RR_WAVES
#define DDMPC_RR_WAVES 4
#endif
// Stable partition of the components 0 .. r-1 by class (cls(rho) in 0 .. NC-1, anything else: dropped): perm[position] = rho
// with class 0 first and the component order kept inside a class; cnt[k] = members of class k.  Every thread scans the
// class table (r ints of LDS scratch `kcl`, broadcast reads) for its own components: O(r) per thread, all in parallel --
// a single thread walking the table costs one dependent L2 round trip per component (0.2 ms of a cfg-5 batch, measured).
template <int NC, class ClsF>
__device__ __forceinline__ void stable_partition(int r, ClsF&& cls, int* perm, int* kcl, int* cnt) {
  const int tid = threadIdx.x, nthr = blockDim.x;
  for (int i = tid; i < r; i += nthr) kcl[i] = cls(i);
  __syncthreads();
  for (int rho = tid; rho < r; rho += nthr) {
    const int c = kcl[rho];
    int tot[NC];
#pragma unroll
    for (int k = 0; k < NC; ++k) tot[k] = 0;
    int before = 0;
    for (int j = 0; j < r; ++j) {
      const int cj = kcl[j];
#pragma unroll
      for (int k = 0; k < NC; ++k) tot[k] += (cj == k) ? 1 : 0;
      before += (cj == c && j < rho) ? 1 : 0;
    }
    if (c >= 0 && c < NC) {
      int off = before;
#pragma unroll
      for (int k = 0; k < NC; ++k) off += (k < c) ? tot[k] : 0;
      perm[off] = rho;
    }
    if (rho == 0) {
#pragma unroll
      for (int k = 0; k < NC; ++k) cnt[k] = tot[k];
    }
  }
  __syncthreads();
}

template <int MODE>
__device__ __forceinline__ void ddmpc_nominal_rr_body(const KParams& P, int RPs, const double* __restrict__ u_d,
                                                      const double* __restrict__ y_d,
                                                      const double* __restrict__ u_past,
                                                      const double* __restrict__ y_past,
                                                      double* __restrict__ u_opt, double* __restrict__ cost,
                                                      int* __restrict__ status, int* __restrict__ iters,
                                                      double rank_tol, double feas_tol, double* scratch,
                                                      long long scratch_stride, double* w_ws,
                                                      unsigned long long* dbg, double* __restrict__ z_ws,
                                                      int* __restrict__ rescued, double* __restrict__ x_ws,
                                                      int* __restrict__ meta_ws) {
  // MODE: 0 = the whole solve in one launch (matrices in LDS: the four-tank sizes); with the matrices in the global
  // workspace the solve is two launches -- 1 = the part that depends on the DATA alone (Gram, its rank-revealing factor,
  // the reduced normal matrix C'WC and its factor), 2 = a solve on the factors a MODE-1 launch left in the workspace --
  // so that (a) ddmpc_prepare / ddmpc_step repeat only the second one while the data stand (only the past window changes
  // between control steps, controller.py:389-407; the factorisations are 60 % of a solve of this size), and (b) each half
  // gets a register allocation of its own (as one kernel: 413 spilled VGPRs, 884 B of scratch per lane).  The pivot
  // pattern of both factors and the live column counts travel in meta_ws (2 rv + 2 ints per instance).
  constexpr int mode = MODE;
  extern __shared__ __attribute__((aligned(16))) double rsm_lds[];
  const long long b = blockIdx.x;
  if (status[b] != 4) return;                           // uniform: only instances the fast path gave up on
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int r = P.r, m = P.m, p = P.p, nch = P.nch, c = P.c;
  const int n = P.npu / m;
  // dynamic LDS: six r-vectors of doubles and four of ints first, then -- when they fit -- the two packed
  // matrices; otherwise the matrices live in a per-instance slice of a global workspace (same code, L2 instead of
  // LDS; __syncthreads orders global accesses within the workgroup)
  const int rv = (r + 1) & ~1;
  double* fv = rsm_lds;
  double* wv = fv + rv;
  double* zs = wv + rv;
  double* z0 = zs + rv;
  double* vv = z0 + rv;
  double* col = vv + rv;
  double* ra = col + rv;                                // four work vectors of the refinement step
  double* rb = ra + rv;
  double* rz = rb + rv;
  double* rd = rz + rv;
  int* perm = reinterpret_cast<int*>(rd + rv);
  int* skip = perm + rv;
  int* skipT = skip + rv;
  int* iperm = skipT + rv;                              // component -> position in the fixed-first order
  double* pan = reinterpret_cast<double*>(iperm + rv);                 // PSD_PAN doubles: scratch of the Cholesky / Gram (always LDS)
  double* rsm;
  if constexpr (MODE == 0) rsm = scratch ? scratch + b * scratch_stride : pan + PSD_PAN;
  else rsm = scratch + b * scratch_stride;                // (a kernel-argument pointer: the matrices are addressed with global, not flat, loads)
  double* G = rsm;                                      // r(r+1)/2
  // (in the global workspace both matrices take whole 16-row tiles: the layout the phase kernels of ddmpc_rr2.hpp work on)
  double* T = G + ((MODE == 0 && !scratch) ? pk_row(r) : pk_row((size_t)((r + 15) & ~15)));   // nR(nR+1)/2
  __shared__ double red[16];
  __shared__ int cnt4[4];
  const double* ud = u_d + b * (long long)P.N * m;
  const double* yd = y_d + b * (long long)P.N * p;
  const double* up = u_past + b * (long long)P.npu;
  const double* yp = y_past + b * (long long)(n * p);
  // ---- components, fixed first ------------------------------------------------------------
  // Inside the fixed block and inside the free block the INPUT components come first (time order), the outputs behind
  // them: with exact data the outputs are the dependent rows (beyond the few that pin the state), so the rows without a
  // pivot gather at the end of each block -- the tail of the free block is then one dead stretch at which the
  // factorisation stops early (packed_psd_cholesky), and everything downstream works on the leading `nlive` columns.
  stable_partition<4>(r, [&](int rho) {
    const int kind = P.tabi[0 * RPs + rho];
    return kind == K_UFIX ? 0 : kind == K_YFIX ? 1 : kind == K_UFREE ? 2 : kind == K_YFREE ? 3 : -1;
  }, perm, iperm, cnt4);
  const int nF = cnt4[0] + cnt4[1], nR = cnt4[2] + cnt4[3];
  for (int i = tid; i < nF + nR; i += nthr) {
    const int rho = perm[i];
    if (i < nF) {
      const int pidx = P.tabi[1 * RPs + rho];
      fv[i] = (pidx >= 0) ? ((pidx < P.npu) ? up[pidx] : yp[pidx - P.npu]) : P.tabd[2 * RPs + rho];
    } else {
      wv[i - nF] = P.tabd[3 * RPs + rho];
      zs[i - nF] = P.tabd[2 * RPs + rho];
    }
  }
  __syncthreads();
  if (dbg && tid == 0) dbg[b * 16 + 0] = __builtin_amdgcn_s_memrealtime();
  // ---- Gram in the permuted order: G(i,j) = sum_t x_{perm i}[t] x_{perm j}[t] ----------------
  // Hankel structure (as in the cold kernel): with components (k, a) = (time offset, channel),
  //   G((k+d, a), (k, b)) = C_d(a,b) + sum_{j<k} ( x_a[j+c+d] x_b[j+c] - x_a[j+d] x_b[j] ),   C_d(a,b) = sum_{t<c} x_a[t+d] x_b[t],
  // so only the Ln*nch^2 lag sums need the full length-c dot product (cfg 5: 19 M instead of 363 M multiply-adds).
  // The lag table borrows the (not yet used) storage of T; if it does not fit there, plain dot products are used.
  const int npk = pk_row(r);
  const int nlag = P.Ln * nch * nch;
  int* meta = meta_ws ? meta_ws + b * (long long)(2 * rv + 2) : nullptr;
  int nlive = 0, nRl = 0;
  if constexpr (MODE != 2) {
  if (nlag <= pk_row(nR)) {
    for (int i = tid; i < r; i += nthr) iperm[perm[i]] = i;
    hankel_gram_packed(P, ud, yd, G, T, iperm, pan);
  } else {
    for (int e = tid; e < r * (r + 1) / 2; e += nthr) {
      const int i = tri_row(e), j = e - i * (i + 1) / 2;
      const int ri = perm[i], rj = perm[j];
      const int ki = ri / nch, ci = ri - ki * nch, kj = rj / nch, cj = rj - kj * nch;
      const double* xi = (ci < m) ? ud + (long long)ki * m + ci : yd + (long long)ki * p + (ci - m);
      const double* xj = (cj < m) ? ud + (long long)kj * m + cj : yd + (long long)kj * p + (cj - m);
      const int si = (ci < m) ? m : p, sj = (cj < m) ? m : p;
      double s0 = 0.0, s1 = 0.0;
      int t = 0;
      for (; t + 1 < c; t += 2) { s0 += xi[t * si] * xj[t * sj]; s1 += xi[(t + 1) * si] * xj[(t + 1) * sj]; }
      if (t < c) s0 += xi[t * si] * xj[t * sj];
      G[pk_row(i) + j] = s0 + s1;
    }
  }
  __syncthreads();
  if (dbg && tid == 0) dbg[b * 16 + 1] = __builtin_amdgcn_s_memrealtime();
  double dmx = 0.0;
  for (int i = tid; i < r; i += nthr) dmx = fmax(dmx, G[pk_row(i) + i]);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) dmx = fmax(dmx, __shfl_xor(dmx, off, 64));
  if ((tid & 63) == 0) red[tid >> 6] = dmx;
  __syncthreads();
  double dmax = 0.0;
  for (int w = 0; w < (nthr >> 6); ++w) dmax = fmax(dmax, red[w]);
  __syncthreads();
  nlive = packed_psd_cholesky(G, r, rank_tol * dmax, skip, pan, -1, true, nF);                // columns >= nlive are zero
  nRl = (nlive > nF) ? ((nlive - nF) < nR ? (nlive - nF) : nR) : 0;                           // live columns of the free block
  } else {
    for (int i = tid; i < r; i += nthr) skip[i] = meta[i];
    nlive = meta[2 * rv]; nRl = meta[2 * rv + 1];
    if (dbg && tid == 0) dbg[b * 16 + 1] = __builtin_amdgcn_s_memrealtime();
    __syncthreads();
  }
  if (dbg && tid == 0) dbg[b * 16 + 2] = __builtin_amdgcn_s_memrealtime();
  // ---- hard constraints: L_FF w = f (skipped pivots carry no unknown); residual of the dependent rows ----
  double resid = 0.0, fmaxv = 1.0;
  if constexpr (MODE != 1) {
  packed_forward_substitute(G, nF, fv, col, skip, red);
  packed_rows_times(G, 0, nF, 0, col, [&](int i) { return i; },
                    [&](int i, double sacc) { vv[i] = skip[i] ? fabs(fv[i] - sacc) : 0.0; });   // what a dependent constraint row is off by
  __syncthreads();
  for (int k = 0; k < nF; ++k) { resid = fmax(resid, vv[k]); fmaxv = fmax(fmaxv, fabs(fv[k])); }
  // ---- z0 = L_RF w ------------------------------------------------------------------------------
  packed_rows_times(G, nF, nR, 0, col, [&](int) { return nF; }, [&](int i, double sacc) { z0[i] = sacc; });
  __syncthreads();
  }
  if (dbg && tid == 0) dbg[b * 16 + 3] = __builtin_amdgcn_s_memrealtime();
  // ---- reduced normal equations T = C' W C, rhs = C' W (zs - z0), C(i,a) = L(nF+i, nF+a), i >= a ----
  if constexpr (MODE != 2) packed_weighted_gram_mfma(G, nF, nR, wv, skip + nF, T, nRl);
  if (mode != 1) { for (int i = tid; i < nR; i += nthr) ra[i] = wv[i] * (zs[i] - z0[i]); }
  for (int a = tid; a < nR; a += nthr) {
    if (a >= nRl) { vv[a] = 0.0; skipT[a] = 1; }                                             // dead tail of the free block
    else if (mode == 2) skipT[a] = meta[rv + a];
  }
  __syncthreads();
  if constexpr (MODE != 1) {
  packed_cols_times(G, nF, nR, nF, nRl, [&](int a) { return a; }, [&](int i) { return ra[i]; },
                    [&](int a, double sacc) { vv[a] = skip[nF + a] ? 0.0 : sacc; });
  __syncthreads();
  }
  if (dbg && tid == 0) dbg[b * 16 + 4] = __builtin_amdgcn_s_memrealtime();
  if constexpr (MODE != 2) {
    double tmx = 0.0;
    for (int a = 0; a < nRl; ++a) tmx = fmax(tmx, T[pk_row(a) + a]);
    packed_psd_cholesky(T, nRl, 1e-14 * tmx, skipT, pan);
    if (meta) {                                          // what a later mode-2 launch needs besides the two factors
      for (int i = tid; i < r; i += nthr) meta[i] = skip[i];
      for (int a = tid; a < nR; a += nthr) meta[rv + a] = skipT[a];
      if (tid == 0) { meta[2 * rv] = nlive; meta[2 * rv + 1] = nRl; }
    }
  }
  if (dbg && tid == 0) dbg[b * 16 + 5] = __builtin_amdgcn_s_memrealtime();
  if constexpr (MODE == 1) return;                       // factors only (the status word stays as it is: MODE 2 follows)
  // ---- T v = rhs by the factor ------------------------------------------------------------------------------
  packed_forward_substitute(T, nRl, vv, rb, skipT, red);
  packed_back_substitute(T, nRl, rb, ra, skipT);          // w2; col keeps w1 for the refinement step
  for (int a = tid; a < nR; a += nthr) vv[a] = (a < nRl) ? ra[a] : 0.0;
  __syncthreads();
  if (dbg && tid == 0) dbg[b * 16 + 6] = __builtin_amdgcn_s_memrealtime();
  // ---- one step of iterative refinement on the KKT system of the problem in the coordinates w of
  //        z = B w,   B = H H_I' L_I^-T   (H_I: the rows with a pivot; B equals L up to the rounding of the Gram route):
  //        min (B_R w - zs)' W (B_R w - zs)   s.t.  B_F w = f,   multipliers mu on the independent fixed rows.
  //      B and B' are applied EXACTLY -- two products with the implicit Hankel matrix and one triangular solve each --
  //      while the correction is solved with the factors at hand (L in place of B).  The Gram route squares cond(H);
  //      this step brings the result back to what cond(H) itself allows.
  double* wk = w_ws + b * (long long)r;                                 // the current w = [w1; w2], position order
  for (int k = tid; k < r; k += nthr) wk[k] = (k < nF) ? col[k] : vv[k - nF];
  __syncthreads();
  // The pass is repeated while it still pays: the correction of pass k is applied through the rounded factors, so the
  // error left behind is about (relative size of that correction) x (relative accuracy of the factors ~ size of the FIRST
  // correction); another pass is made while that product is above 1e-9 (cap P.refine_max, at least one pass).
  double rel0 = 0.0, prevrel = 1e300;
  for (int pass = 0;; ++pass) {
  for (int a = tid; a < nR; a += nthr) vv[a] = wk[nF + a];             // w2 of this pass ((b) below reads it)
  // (a) z_ex = B w:  w (position order) -> x = L^-T w (zero on rows without a pivot) -> H' x -> H (H' x)
  for (int k = tid; k < r; k += nthr) { ra[k] = wk[k]; if (k >= nlive) rb[k] = 0.0; }
  __syncthreads();
  packed_back_substitute(G, nlive, ra, rb, skip);                     // (rows without a pivot carry no unknown: zero)
  if (dbg && tid == 0 && pass == 0) dbg[b * 16 + 8] = __builtin_amdgcn_s_memrealtime();   // (diagnostics: first refinement pass)
  for (int k = tid; k < r; k += nthr) ra[perm[k]] = rb[k];           // component order
  __syncthreads();
  hankel_normal_times(P, ud, yd, ra, rd, pan);
  for (int k = tid; k < r; k += nthr) rz[k] = rd[perm[k]];           // z_ex in position order
  if (dbg && tid == 0 && pass == 0) dbg[b * 16 + 9] = __builtin_amdgcn_s_memrealtime();   // (diagnostics: first refinement pass)
  // (b) multipliers of the starting point: L_FF' mu = -L_RF' W (z_R - zs), z_R = z0 + C w2 (the unrefined solution)
  packed_rows_times(G, nF, nR, nF, vv, [&](int i) { return (nF + i + 1) < nlive ? (nF + i + 1) : nlive; },
                    [&](int i, double sacc) { rb[nF + i] = wv[i] * (z0[i] + sacc - zs[i]); });
  __syncthreads();
  packed_cols_times(G, nF, nR, 0, nF, [&](int) { return 0; }, [&](int i) { return rb[nF + i]; },
                    [&](int k, double sacc) { ra[k] = -sacc; });
  __syncthreads();
  packed_back_substitute(G, nF, ra, rd, skip);                          // mu -> rd[0..nF)
  if (dbg && tid == 0 && pass == 0) dbg[b * 16 + 10] = __builtin_amdgcn_s_memrealtime();   // (diagnostics: first refinement pass)
  // (c) residual of the stationarity rows: rw = -B' v,  v = [mu on the independent fixed rows ; W (z_ex,R - zs)]
  for (int k = tid; k < r; k += nthr) rb[k] = (k < nF) ? (skip[k] ? 0.0 : rd[k]) : wv[k - nF] * (rz[k] - zs[k - nF]);
  __syncthreads();
  for (int k = tid; k < r; k += nthr) ra[perm[k]] = rb[k];
  __syncthreads();
  hankel_normal_times(P, ud, yd, ra, rd, pan);
  for (int k = tid; k < r; k += nthr) rb[k] = rd[perm[k]];
  if (dbg && tid == 0 && pass == 0) dbg[b * 16 + 11] = __builtin_amdgcn_s_memrealtime();   // (diagnostics: first refinement pass)
  __syncthreads();
  for (int k = nlive + tid; k < r; k += nthr) ra[k] = 0.0;
  packed_forward_substitute(G, nlive, rb, ra, skip, red);                // ra = L_I^-1 (H_I H' v) = -rw
  if (dbg && tid == 0 && pass == 0) dbg[b * 16 + 12] = __builtin_amdgcn_s_memrealtime();   // (diagnostics: first refinement pass)
  // (d) correction with the factors at hand:  dw1 = L_FF^-1 (f - z_ex,F);  T dw2 = rw2 - C' W L_RF dw1
  for (int k = tid; k < nF; k += nthr) rd[k] = fv[k] - rz[k];
  __syncthreads();
  packed_forward_substitute(G, nF, rd, rb, skip, red);                   // dw1 -> rb[0..nF)
  if (dbg && tid == 0 && pass == 0) dbg[b * 16 + 13] = __builtin_amdgcn_s_memrealtime();   // (diagnostics: first refinement pass)
  packed_rows_times(G, nF, nR, 0, rb, [&](int) { return nF; }, [&](int i, double sacc) { rd[nF + i] = sacc; });   // L_RF dw1
  __syncthreads();
  packed_cols_times(G, nF, nR, nF, nRl, [&](int a) { return a; }, [&](int i) { return wv[i] * rd[nF + i]; },
                    [&](int a, double sacc) { vv[a] = skip[nF + a] ? 0.0 : -ra[nF + a] - sacc; });              // rhs of the T system
  __syncthreads();
  packed_forward_substitute(T, nRl, vv, col, skipT, red);                // col: work vector (w lives in wk)
  if (dbg && tid == 0 && pass == 0) dbg[b * 16 + 14] = __builtin_amdgcn_s_memrealtime();   // (diagnostics: first refinement pass)
  packed_back_substitute(T, nRl, col, vv, skipT);                        // dw2 -> vv
  if (dbg && tid == 0 && pass == 0) dbg[b * 16 + 15] = __builtin_amdgcn_s_memrealtime();   // (diagnostics: first refinement pass)
  for (int a = nRl + tid; a < nR; a += nthr) vv[a] = 0.0;
  __syncthreads();
  // size of this correction relative to w; decide whether another pass pays
  double dmx = 0.0, wmx = 0.0;
  for (int k = tid; k < r; k += nthr) {
    const double dl = (k < nF) ? rb[k] : vv[k - nF];
    dmx = fmax(dmx, fabs(dl)); wmx = fmax(wmx, fabs(wk[k]));
  }
  const double rel = block_max(dmx, red) / fmax(block_max(wmx, red), 1e-300);
  if (pass == 0) rel0 = rel;
  const bool more_passes = (pass + 1 < P.refine_max) && (rel * rel0 > 1e-9) && (rel < 0.25 * prevrel);
  __syncthreads();
  if (!more_passes) break;
  prevrel = rel;
  for (int k = tid; k < r; k += nthr) wk[k] += (k < nF) ? rb[k] : vv[k - nF];
  __syncthreads();
  }
  // ---- z_R = z_ex,R + L_RF dw1 + C dw2; outputs ------------------------------------------------------
  double part = 0.0;
  double* uo = u_opt + b * (long long)((P.Ln - n) * m);
  packed_rows_times(G, nF, nR, nF, vv, [&](int i) { return (nF + i + 1) < nlive ? (nF + i + 1) : nlive; }, [&](int i, double sacc) {
    const double z = rz[nF + i] + rd[nF + i] + sacc;
    const double dlt = z - zs[i];
    part += wv[i] * dlt * dlt;
    const int oidx = P.tabi[2 * RPs + perm[nF + i]];
    if (oidx >= 0) uo[oidx] = z;
    if (z_ws) z_ws[b * (long long)P.rE + perm[nF + i]] = z;
  });
  for (int k = tid; k < nF; k += nthr) {
    const int oidx = P.tabi[2 * RPs + perm[k]];
    if (oidx >= 0) uo[oidx] = fv[k];                    // terminal inputs are part of optimal_u
    if (z_ws) z_ws[b * (long long)P.rE + perm[k]] = fv[k];
  }
  if (rescued && tid == 0) rescued[b] = 1;
  if (x_ws) {
    // x = L_I^-T w of the final w (component order): z = H (H' x), so alpha = H' x is formed on demand by
    // ddmpc_reconstruct_kernel (the `.alpha.value` stand-in of controller.py:434)
    __syncthreads();
    for (int k = tid; k < r; k += nthr) ra[k] = wk[k] + ((k < nF) ? rb[k] : vv[k - nF]);
    __syncthreads();
    for (int k = nlive + tid; k < r; k += nthr) col[k] = 0.0;
    __syncthreads();
    packed_back_substitute(G, nlive, ra, col, skip);
    for (int k = tid; k < r; k += nthr) x_ws[b * (long long)P.rE + perm[k]] = col[k];
  }
  part = wave_sum(part);
  __syncthreads();
  if ((tid & 63) == 0) red[tid >> 6] = part;
  __syncthreads();
  if (tid == 0) {
    double tot = 0.0;
    for (int w = 0; w < (nthr >> 6); ++w) tot += red[w];
    cost[b] = tot;
    const bool feasible = resid <= feas_tol * fmaxv;
    status[b] = !(fabs(tot) < 1e300) ? 4 : (feasible ? 0 : 2);      // 2 = "infeasible"
    if (iters) iters[b] = 1;
    if (dbg) dbg[b * 16 + 7] = __builtin_amdgcn_s_memrealtime();
  }
}

template <int MODE>
__global__ __launch_bounds__(512, DDMPC_RR_WAVES) void ddmpc_nominal_rr_kernel(KParams P, int RPs, const double* __restrict__ u_d,
                                                               const double* __restrict__ y_d,
                                                               const double* __restrict__ u_past,
                                                               const double* __restrict__ y_past,
                                                               double* __restrict__ u_opt, double* __restrict__ cost,
                                                               int* __restrict__ status, int* __restrict__ iters,
                                                               double rank_tol, double feas_tol, double* scratch,
                                                               long long scratch_stride, double* w_ws,
                                                               unsigned long long* dbg, double* __restrict__ z_ws,
                                                               int* __restrict__ rescued, double* __restrict__ x_ws,
                                                               int* __restrict__ meta_ws) {
  ddmpc_nominal_rr_body<MODE>(P, RPs, u_d, y_d, u_past, y_past, u_opt, cost, status, iters, rank_tol, feas_tol, scratch, scratch_stride, w_ws,
                              dbg, z_ws, rescued, x_ws, meta_ws);
}
// The same with 1024 threads per workgroup: NOMINAL controllers of 1025 .. 1524 rows (ten r-vectors + the panel scratch in
// 160 KB of LDS; the routines keep two entries of an r-vector per thread).  MODE 1 / 2 only (matrices in the global workspace).
template <int MODE>
__global__ __launch_bounds__(1024) void ddmpc_nominal_rr_wide_kernel(KParams P, int RPs, const double* __restrict__ u_d,
                                                                    const double* __restrict__ y_d,
                                                                    const double* __restrict__ u_past,
                                                                    const double* __restrict__ y_past,
                                                                    double* __restrict__ u_opt, double* __restrict__ cost,
                                                                    int* __restrict__ status, int* __restrict__ iters,
                                                                    double rank_tol, double feas_tol, double* scratch,
                                                                    long long scratch_stride, double* w_ws,
                                                                    unsigned long long* dbg, double* __restrict__ z_ws,
                                                                    int* __restrict__ rescued, double* __restrict__ x_ws,
                                                                    int* __restrict__ meta_ws) {
  ddmpc_nominal_rr_body<MODE>(P, RPs, u_d, y_d, u_past, y_past, u_opt, cost, status, iters, rank_tol, feas_tol, scratch, scratch_stride, w_ws,
                              dbg, z_ws, rescued, x_ws, meta_ws);
}

// ---------------------------------------------------------------------------------------------------------------
// Problems with more rows than the register-resident cold kernels hold ((m+p)(L+n) > 271): the same reduced system
//   (G + lam*D) beta = t,  z = t - lam*D*beta,  primal-dual active set on the slack box (CONVEX)
// with the matrices in a per-instance slice of a global workspace (packed lower triangles), plain VALU code:
// Hankel-structured Gram, panel-blocked Cholesky (panel rows in registers), row-wise substitutions.  The components the slack
// box acts on (set B, the sigma rows of the prediction window) are ordered LAST: the factor of the other columns
// (set A) and the Schur complement S = K_BB - L_BA L_BA' are formed once, and an active-set iteration only
// re-factors S + lam*D_B(active set) (|B| = p*L rows) and substitutes through it.  Same component tables, outputs,
// status and iteration count as ddmpc_cold_solve_kernel (ddmpc_kernels.hpp); scalar, diagonal or dense weights.  One workgroup
// per instance.  Workspace per instance: r(r+1)/2 + max(Ln*nch^2, 2*|B|(|B|+1)/2) doubles.
// ---------------------------------------------------------------------------------------------------------------
// MODE 0: the whole solve (ddmpc_solve).  MODE 1: what depends on the data and the weights alone -- Gram + lam D, the factor
// of the A columns, the Schur complement of the boxed block -- left in the workspace, with the outcome of that
// factorisation in meta_ws[b] (ddmpc_prepare).  MODE 2: a solve on what a MODE-1 launch left there (ddmpc_step: only the past
// window has changed, controller.py:389-407); same arithmetic, so the results are bit-equal to MODE 0's.
template <int MODE>
__device__ __forceinline__ void ddmpc_large_solve_body(const KParams& P, int RPs, const double* __restrict__ u_d,
                                                       const double* __restrict__ y_d,
                                                       const double* __restrict__ u_past,
                                                       const double* __restrict__ y_past,
                                                       double* __restrict__ u_opt, double* __restrict__ cost,
                                                       int* __restrict__ status, int* __restrict__ iters,
                                                       double* __restrict__ beta_ws,
                                                       signed char* __restrict__ act_ws, double* scratch,
                                                       long long scratch_stride, int* __restrict__ meta_ws, const long long b,
                                                       const long long slot) {      // slot: which slice of scratch / word of meta_ws
  extern __shared__ __attribute__((aligned(16))) double lsm_lds[];
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int r = P.r, m = P.m, p = P.p;
  const int n = P.npu / m;
  const int rv = (r + 1) & ~1;
  // r-vectors, indexed by POSITION in the elimination order (A first, B last)
  double* ct = lsm_lds;                                 // target of the component without the bound shift
  double* yv = ct + rv;                                 // y = L^-1 t  (A part fixed, B part per iteration)
  double* bv = yv + rv;                                 // beta
  double* zb = bv + rv;                                 // L_BA y_A (B part only)
  double* qa = zb + rv;                                 // two work vectors of the refinement step
  double* qb = qa + rv;
  int* act = reinterpret_cast<int*>(qb + rv);
  int* skip = act + rv;
  int* perm = skip + rv;                                // position -> component
  int* iperm = perm + rv;                               // component -> position
  double* pan = reinterpret_cast<double*>(iperm + rv);
  const int npk = pk_row(r);
  double* G = scratch + slot * scratch_stride;          // Gram -> [L_AA; L_BA] in its first nA columns
  __shared__ double red[16];
  __shared__ int flag[1];
  __shared__ int cnt[2];
  const double* ud = u_d + b * (long long)P.N * m;
  const double* yd = y_d + b * (long long)P.N * p;
  const double* up = u_past + b * (long long)P.npu;
  const double* yp = y_past + b * (long long)(n * p);
  stable_partition<2>(r, [&](int rho) {                   // elimination order: boxed components last
    const int kind = P.tabi[0 * RPs + rho];
    return (P.convex && (kind == K_WPRED || kind == K_WTERM)) ? 1 : 0;
  }, perm, skip, cnt);
  const int nA = cnt[0], nB = r - nA;
  const int npB = pk_row(nB);
  double* S = G + npk;                                  // Schur complement of the B block (without lam*D_B)
  double* T = S + npB;                                  // S + lam*D_B -> its factor
  for (int i = tid; i < r; i += nthr) {
    const int rho = perm[i];
    iperm[rho] = i;
    const int pidx = P.tabi[1 * RPs + rho];
    ct[i] = (pidx >= 0) ? ((pidx < P.npu) ? up[pidx] : yp[pidx - P.npu]) : P.tabd[2 * RPs + rho];
    act[i] = 0;
  }
  __syncthreads();
  // dense weights: wib[rho] = (W^-1 betac)[rho] for a beta in COMPONENT order, one 32-lane half wave per row of the matrix
  auto dense_times = [&](const double* betac, double* wib) {
    const int hw = tid >> 5, t32 = tid & 31, nhw = nthr >> 5;
    for (int r0 = 0; r0 < r; r0 += nhw) {
      const int rho = r0 + hw;
      double sacc = 0.0;
      if (rho < r) {
        const double* dr = P.dmat + (long long)rho * RPs;
        for (int j = t32; j < r; j += 32) sacc += dr[j] * betac[j];
      }
#pragma unroll
      for (int off = 16; off > 0; off >>= 1) sacc += __shfl_xor(sacc, off, 32);
      if (t32 == 0 && rho < r) wib[rho] = sacc;
    }
  };
  int st = 0, iter = 0;
  if constexpr (MODE != 2) {
  hankel_gram_packed(P, ud, yd, G, G + npk, iperm, pan);   // lag table in the (not yet used) storage behind G
  __syncthreads();
  if (P.dense_w) {     // dense weighting matrices: lam * W^-1 (shared by the batch, component order) on every pair of components;
                       // the diagonal table below then only carries the 1/lamb_sigma terms, which is all the slack box switches
    for (int e = tid; e < r * (r + 1) / 2; e += nthr) {
      const int i = tri_row(e), j = e - i * (i + 1) / 2;
      G[pk_row(i) + j] += P.lam * P.dmat[(long long)perm[i] * RPs + perm[j]];
    }
    __syncthreads();
  }
  for (int i = tid; i < nA; i += nthr) G[pk_row(i) + i] += P.lam * P.tabd[0 * RPs + perm[i]];
  __syncthreads();
  packed_psd_cholesky(G, r, 0.0, skip, pan, nA);        // columns of A only; a pivot that is not positive is skipped
  {
    double nbad = 0.0;
    for (int i = tid; i < nA; i += nthr) nbad += skip[i] ? 1.0 : 0.0;
    if (block_sum(nbad, red) != 0.0) st = 4;            // uniform
  }
  __syncthreads();
  // S = K_BB - L_BA L_BA'  (the diagonal shift of B is added per active set)
  if (st == 0) packed_schur_mfma(G, nA, nB, nA, S);
  if (tid == 0) meta_ws[slot] = st;                        // (a whole solve leaves the same record behind as MODE 1)
  if constexpr (MODE == 1) return;
  } else {
    st = meta_ws[slot];                                    // uniform: how the factorisation of the A columns went
  }
  if (st == 0) {
    // L_AA y_A = t_A (t_A does not depend on the active set), zb = L_BA y_A
    packed_forward_substitute(G, nA, ct, yv, nullptr, red);
    packed_rows_times(G, nA, nB, 0, yv, [&](int) { return nA; }, [&](int i, double sacc) { zb[i] = sacc; });
    __syncthreads();
    // ---- active-set iterations on the B block --------------------------------------------------------------
    while (true) {
      ++iter;
      if (nB == 0) break;
      for (int e = tid; e < npB; e += nthr) T[e] = S[e];
      __syncthreads();
      for (int i = tid; i < nB; i += nthr) {
        const int rho = perm[nA + i];
        const double D = act[nA + i] ? P.tabd[1 * RPs + rho] : P.tabd[0 * RPs + rho];
        T[pk_row(i) + i] += P.lam * D;
      }
      __syncthreads();
      packed_psd_cholesky(T, nB, 0.0, skip + nA, pan);
      double nbad = 0.0;
      for (int i = tid; i < nB; i += nthr) nbad += skip[nA + i] ? 1.0 : 0.0;
      if (block_sum(nbad, red) != 0.0) { st = 4; break; }
      __syncthreads();
      for (int k = tid; k < nB; k += nthr) qa[k] = (ct[nA + k] + act[nA + k] * P.bound) - zb[k];
      __syncthreads();
      packed_forward_substitute(T, nB, qa, yv + nA, nullptr, red);   // L_BB y_B = t_B - L_BA y_A
      packed_back_substitute(T, nB, yv + nA, bv + nA, nullptr);   // L_BB' beta_B = y_B
      // slack box: primal-dual active-set update (sigma[n*p:], controller.py:659); B holds exactly those components
      if (tid == 0) flag[0] = 0;
      __syncthreads();
      const double scale = -P.lam / P.lamb_sigma;
      for (int i = tid; i < nB; i += nthr) {
        const double sh = scale * bv[nA + i];
        const int ns = (sh > P.bound) ? 1 : (sh < -P.bound) ? -1 : 0;
        if (ns != act[nA + i]) { act[nA + i] = ns; flag[0] = 1; }
      }
      __syncthreads();
      const int changed = flag[0];
      __syncthreads();
      if (!changed) break;
      if (iter >= P.max_iter) { st = 4; break; }
    }
  }
  if (st == 0) {
    // L_AA' beta_A = y_A - L_BA' beta_B: the B rows first (column j of L_BA read by thread j: coalesced), then the
    // row-oriented back substitution through L_AA
    packed_cols_times(G, nA, nB, 0, nA, [&](int) { return 0; }, [&](int i) { return bv[nA + i]; },
                      [&](int j, double sacc) { yv[j] -= sacc; });
    __syncthreads();
    packed_back_substitute(G, nA, yv, bv, nullptr);
    // ---- one step of iterative refinement: the residual t - (H (H' beta) + lam*D*beta) is formed with two products
    //      with the implicit Hankel matrix instead of the rounded Gram matrix, the correction is solved with the
    //      block factors of the final active set.  The Gram route squares cond(H); with K applied as H H' the result
    //      is what cond(H) itself allows (cfg-5 size: 1e-7 -> 1e-10 in optimal_u).
    // passes repeat until the correction is at rounding level or stops shrinking (cap P.refine_max); DDMPC_REFINE_OFF skips them
    double prev = 1e300;
    for (int pass = 0; P.refine != 0 && pass < P.refine_max; ++pass) {
    for (int i = tid; i < r; i += nthr) qa[perm[i]] = bv[i];            // beta in component order
    __syncthreads();
    hankel_normal_times(P, ud, yd, qa, qb, pan);                         // H H' beta, component order
    if (P.dense_w) {                                                     // + lam W^-1 beta (zb is free after the active-set loop)
      dense_times(qa, zb);
      __syncthreads();
      for (int rho = tid; rho < r; rho += nthr) qb[rho] += P.lam * zb[rho];
      __syncthreads();
    }
    for (int i = tid; i < r; i += nthr) {
      const int rho = perm[i];
      const int a = act[i];
      const double D = a ? P.tabd[1 * RPs + rho] : P.tabd[0 * RPs + rho];
      qa[i] = (ct[i] + a * P.bound) - qb[rho] - P.lam * D * bv[i];       // residual, position order
    }
    __syncthreads();
    packed_forward_substitute(G, nA, qa, qb, nullptr, red);              // y_A
    packed_rows_times(G, nA, nB, 0, qb, [&](int) { return nA; }, [&](int i, double sacc) { qa[nA + i] -= sacc; });   // res_B - L_BA y_A
    __syncthreads();
    if (nB > 0) {
      packed_forward_substitute(T, nB, qa + nA, qb + nA, nullptr, red);  // y_B
      packed_back_substitute(T, nB, qb + nA, qa + nA, nullptr);          // dbeta_B -> qa[nA..r)
    }
    packed_cols_times(G, nA, nB, 0, nA, [&](int) { return 0; }, [&](int i) { return qa[nA + i]; },
                      [&](int j, double sacc) { qb[j] -= sacc; });
    __syncthreads();
    packed_back_substitute(G, nA, qb, qa, nullptr);                      // dbeta_A -> qa[0..nA)
    double dmx = 0.0, bmx = 0.0;
    for (int i = tid; i < r; i += nthr) {
      const double bn = bv[i] + qa[i];
      dmx = fmax(dmx, fabs(qa[i])); bmx = fmax(bmx, fabs(bn));
      bv[i] = bn;
    }
    const double rel = block_max(dmx, red) / fmax(block_max(bmx, red), 1e-300);
    __syncthreads();
    if (!(rel > 1e-13) || !(rel < 0.25 * prev)) break;
    prev = rel;
    }
  }
  // ---- outputs: z = t - lam*D*beta; cost = control cost + lam*beta'z + lamb_sigma*|sigma|^2 --------------------
  double part = 0.0, bad = 0.0;
  double* uo = u_opt + b * (long long)((P.Ln - n) * m);
  if (st == 0 && P.dense_w) {                            // zb[rho] = (W^-1 beta)[rho], component order
    for (int i = tid; i < r; i += nthr) qa[perm[i]] = bv[i];
    __syncthreads();
    dense_times(qa, zb);
    __syncthreads();
  }
  if (st == 0) {
    for (int i = tid; i < r; i += nthr) {
      const int rho = perm[i];
      const int s_act = act[i];
      const double bb = bv[i];
      const double D = s_act ? P.tabd[1 * RPs + rho] : P.tabd[0 * RPs + rho];
      const double t = ct[i] + s_act * P.bound;
      const double z = t - P.lam * (D * bb + (P.dense_w ? zb[rho] : 0.0));
      const double wq = P.tabd[3 * RPs + rho];
      const double tb = P.tabd[2 * RPs + rho];          // setpoint of the component (u_s / y_s)
      const int oidx = P.tabi[2 * RPs + rho];
      const int kind = P.tabi[0 * RPs + rho];
      if (!(fabs(bb) < 1e300)) bad = 1.0;
      double contrib = P.lam * bb * z;
      if (P.dense_w && (kind == K_UFREE || kind == K_YFREE || kind == K_WPRED)) {
        // (z - t)' W (z - t) summed over the weighted components equals -lam * beta' (z - t); a sigma held at its bound adds
        // lamb_sigma * bound^2 (as in the register-resident kernel)
        contrib -= P.lam * bb * (z - t);
        if (s_act != 0) contrib += P.box_cost;
      } else
      if (kind == K_UFREE || kind == K_YFREE) { const double dlt = z - tb; contrib += wq * dlt * dlt; }
      else if (kind == K_WINT) { const double sg = z - ct[i]; contrib += P.lamb_sigma * sg * sg; }
      else if (kind == K_WTERM) { const double sg = z - tb; contrib += P.lamb_sigma * sg * sg; }
      else if (kind == K_WPRED) {
        const double sg = (s_act != 0) ? s_act * P.bound : -P.lam * bb / P.lamb_sigma;
        const double dlt = z - sg - tb;
        contrib += wq * dlt * dlt + P.lamb_sigma * sg * sg;
      }
      part += contrib;
      if (oidx >= 0) uo[oidx] = z;                      // ubar[n*m:], controller.py:799-805
      if (beta_ws) beta_ws[b * (long long)P.rE + rho] = bb;
      if (act_ws) act_ws[b * (long long)P.rE + rho] = (signed char)s_act;
    }
  }
  const double tot = block_sum(part, red);
  const double nbad = block_sum(bad, red);
  if (tid == 0) {
    if (nbad != 0.0 || !(fabs(tot) < 1e300)) st = 4;
    cost[b] = tot;
    status[b] = st;
    if (iters) iters[b] = iter > 0 ? iter : 1;
  }
}

// grid = batch (only_status == 0): one workgroup per instance.  only_status != 0: the fall-back launch behind the phase kernels
// of ddmpc_rr3.hpp -- a SMALL persistent grid whose workgroups walk the batch and solve the instances marked with that status
// (usually none).  Small on purpose: this kernel spills (668 B of scratch per lane), and a batch-sized grid of it makes the
// runtime set up ~175 MB of scratch for the dispatch -- measured 3.9 ms per launch even when every workgroup left at once.
template <int MODE>
__global__ __launch_bounds__(512, 4) void ddmpc_large_solve_kernel(KParams P, int RPs, const double* __restrict__ u_d,
                                                                const double* __restrict__ y_d,
                                                                const double* __restrict__ u_past,
                                                                const double* __restrict__ y_past,
                                                                double* __restrict__ u_opt, double* __restrict__ cost,
                                                                int* __restrict__ status, int* __restrict__ iters,
                                                                double* __restrict__ beta_ws,
                                                                signed char* __restrict__ act_ws, double* scratch,
                                                                long long scratch_stride, int* __restrict__ meta_ws, int only_status,
                                                                long long nbatch) {
  if (only_status == 0) {
    ddmpc_large_solve_body<MODE>(P, RPs, u_d, y_d, u_past, y_past, u_opt, cost, status, iters, beta_ws, act_ws, scratch, scratch_stride,
                                 meta_ws, (long long)blockIdx.x, (long long)blockIdx.x);
    return;
  }
  {                                                                // usually nothing is marked: find that out with one parallel sweep
    int mine = 0;
    for (long long b = blockIdx.x + (long long)gridDim.x * threadIdx.x; b < nbatch; b += (long long)gridDim.x * blockDim.x)
      mine |= (status[b] == only_status) ? 1 : 0;
    if (!__syncthreads_or(mine)) return;
  }
  for (long long b = blockIdx.x; b < nbatch; b += gridDim.x) {
    if (status[b] == only_status)                                  // (workgroup-uniform)
      ddmpc_large_solve_body<MODE>(P, RPs, u_d, y_d, u_past, y_past, u_opt, cost, status, iters, beta_ws, act_ws, scratch, scratch_stride,
                                   meta_ws, b, (long long)blockIdx.x);     // (a workspace slice per WORKGROUP of the small grid)
    __syncthreads();                                               // LDS is reused by the next instance
  }
}

// The same with 1024 threads per workgroup: ROBUST controllers of 1025 .. 2048 rows (hankel_matrix.py:47 has no size bound).  The
// routines keep PSD_RPT = 2 entries of an r-vector per thread, so the row count a workgroup holds doubles with its size; the register
// budget per thread halves (this kernel spills either way: a set-up-size path, not a throughput kernel).  grid = batch.
template <int MODE>
__global__ __launch_bounds__(1024) void ddmpc_large_solve_wide_kernel(KParams P, int RPs, const double* __restrict__ u_d,
                                                                     const double* __restrict__ y_d,
                                                                     const double* __restrict__ u_past,
                                                                     const double* __restrict__ y_past,
                                                                     double* __restrict__ u_opt, double* __restrict__ cost,
                                                                     int* __restrict__ status, int* __restrict__ iters,
                                                                     double* __restrict__ beta_ws,
                                                                     signed char* __restrict__ act_ws, double* scratch,
                                                                     long long scratch_stride, int* __restrict__ meta_ws, int only_status,
                                                                     long long nbatch) {
  (void)only_status; (void)nbatch;
  ddmpc_large_solve_body<MODE>(P, RPs, u_d, y_d, u_past, y_past, u_opt, cost, status, iters, beta_ws, act_ws, scratch, scratch_stride,
                               meta_ws, (long long)blockIdx.x, (long long)blockIdx.x);
}

}  // namespace ddmpc
