// ddmpc_rr2.hpp -- problems beyond the register-resident kernels ((m+p)(L+n) > 271, BASELINE configs[4]): the part of a
// solve that depends on the DATA alone, as PHASE KERNELS that run the whole batch in lock step.
//
// ddmpc_nominal_rr_kernel<1> (ddmpc_workspace_kernels.hpp) does all of this inside ONE workgroup per instance: a chain of dependent
// steps on a workspace in HBM, one register allocation for every phase (219 spilled VGPRs), two co-resident workgroups that
// queue on the same memory path.  Here every phase is a kernel of its own -- its own register budget, no scratch -- and an
// instance is worked on by as many workgroups as the phase has independent tiles; the dependency between panel steps is the
// kernel boundary (~2 us each, ~35 of them per factorisation chain):
//
//   rr2_gram_kernel          G = H H' in the fixed-first component order (controller.py:506-538 through hankel_matrix.py:5-53):
//                            lag sums C_d = X_d' X_0 on the matrix pipe, then every wave walks "its" lags down the block
//                            diagonals with one rank-2 MFMA per tile (the Hankel window update) and stores the tiles
//   rr2_chol_update_kernel   left-looking update of one 64-column panel with every live column in front of it (MFMA tiles,
//                            both operands 128-byte pieces of packed rows, dead 16-column chunks skipped by a bit mask)
//   rr2_chol_panel_kernel    the panel's 64 x 64 diagonal block factored in LDS with the skipped-pivot rule, the rows below
//                            it solved against it (MFMA), pivot flags and the live-chunk mask recorded
//   rr2_meta_kernel          live column counts from the pivot pattern
//   rr2_cwc_kernel           T = C' W C, the reduced normal matrix of the weighted components (MFMA)
//   (the two Cholesky kernels again for T, with per-instance sizes)
//
// The workspace they leave -- the factor of G in place, the factor of T behind it, pivot flags and live counts in `meta` -- is
// exactly what ddmpc_nominal_rr_kernel<2> (the solve on the factors) expects.
#pragma once
#include "ddmpc_workspace_kernels.hpp"

namespace ddmpc {

constexpr int RR2_NB = 64;        // panel width of the lock-step Cholesky (four 16-column tiles)
constexpr int RR2_SL = 5;         // lags per wave in the Gram kernel
constexpr int RR2_XCAP = 4096;    // doubles of LDS the Gram kernel stages trajectory chunks in
constexpr int RR2_TLD = 17;       // doubles per row of a 16 x 16 tile in LDS (16 + 1: a column of a tile spreads over the banks;
                                  // with 16, 81 % of the panel kernel's LDS cycles were bank conflicts)
constexpr int RR2_TSZ = 16 * RR2_TLD;
constexpr double RR2_RETIRE = 1e-12;   // Rr2Chol::res
constexpr int RR2_UT = 3;         // row tiles per wave in the Cholesky update kernel

// D[a][b] += sum_k A[a][k] B[k][b] on v_mfma_f64_16x16x4: lane (l15, l4) passes A[a = l15][k = l4] and B[k = l4][b = l15];
// register q of the accumulator holds D[l4 + 4q][l15].
__device__ __forceinline__ d4 rr2_mfma(double a, double b, d4 acc) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0); }

// One packed factorisation of the batch (the Gram matrix G, or the reduced normal matrix T behind it).
struct Rr2Chol {
  double* ws;                       // workspace, instance b at ws + b * stride
  long long stride;
  long long off;                    // offset of this matrix inside an instance's slice
  int n16;                          // rows (a multiple of 16), the same for every instance ...
  const int* n_inst;                // ... unless this is set: rows of instance b = n_inst[b * n_stride] rounded up to 16
  long long n_stride;
  const unsigned long long* dmax;   // per instance: largest diagonal entry (bit pattern of a non-negative double)
  long long d_stride;
  double tol_rel;                   // pivots <= tol_rel * dmax are skipped
  int* skip;                        // pivot flags out: skip[b * s_stride + i], i < nflag
  long long s_stride;
  int nflag;
  unsigned long long* live;         // per instance: bit j = 16-column chunk j holds at least one pivot
  long long l_stride;
  double* m64;                      // per instance: Minv = S L~^-1 of every 64 x 64 diagonal block, row-major, block k at m64 + 4096 k
  long long m64_stride;
  // Early retirement of dependent rows (lock-step pipeline only; nullptr: off).  res[i] = G(i,i) - sum_{j done} L(i,j)^2, the
  // residual diagonal of row i below the current block, kept up to date by the update launches.  It can only shrink, and
  // (positive semi-definite matrix) the squares of the rest of row i of L sum to at most res[i]: once res[i] <= RR2_RETIRE * dmax
  // for all 16 rows of a tile the tile is retired -- its remaining columns are written as zeros, later update launches pass
  // over it and the panel step treats it as zero rows.  With exact data (BASELINE configs[4]: rank 312 of 608, the 176 output
  // rows of the free steps all dependent) that is a third of the rows.
  // RR2_RETIRE is NOT the pivot tolerance (1e-8): zeroing what is left of a row perturbs it by up to sqrt(res[i]), and a
  // dependent row can sit at a genuine 1e-10 * dmax for a few panels before the column that completes it arrives (a SISO plant
  // of the fuzz set: retiring at 1e-8 cost it four digits, 7e-13 -> 1.4e-8 in u).  At 1e-12 only rows at the rounding floor of
  // their own residue go (numpy, configs[4] and that plant: nothing lies between 1e-14 and 1e-12), rows above it are simply
  // carried to their pivot as before.
  // dead: two words per instance, bit t = row tile t retired; the launches of block k read word k & 1 and write the other one
  // (every workgroup of a launch sees the same set).
  double* res;
  long long res_stride;
  unsigned long long* dead;
  long long dead_stride;
  // The pivot candidates as they were met (nullptr: not recorded): cand[i] = the diagonal entry column i was decided on -- the
  // pivot where it was accepted, the residue where it was skipped (0 for padding and retired rows).  The rank decision of an
  // exact-data problem is judged from this sequence afterwards (rr2_rank_margin_kernel).
  double* cand;
  long long cand_stride;
  const double* tol_inst;           // per-instance pivot tolerance (relative), nullptr: tol_rel for everybody (second pass of the
                                    // rank decision, rr2_rank_margin_kernel)
};

__device__ __forceinline__ double rr2_bits_to_double(unsigned long long v) { return __longlong_as_double((long long)v); }

// ---------------------------------------------------------------------------------------------------------------
// Gram matrix in the permuted (fixed-first) order, packed lower triangle with rows on 128-byte boundaries (pk_row).
// grid = (ceil(Ln / (4 RR2_SL)), batch), 256 threads: wave w of workgroup g owns the lags g*4*RR2_SL + w + 4*sl.
//   G((k+d, a), (k, b)) = C_d(a,b) + sum_{j<k} ( x_a[j+c+d] x_b[j+c] - x_a[j+d] x_b[j] ),   C_d(a,b) = sum_{t<c} x_a[t+d] x_b[t]
// The lag sums stay in the accumulators they were formed in; the window update of a whole 16x16 block is ONE MFMA with the
// contraction index (tail term, -head term, 0, 0).
// ---------------------------------------------------------------------------------------------------------------
// PACKED (nch <= 8, one channel tile): the 16 rows of the A operand carry LP = 16 / nch consecutive lags of the nch channels --
// row l = (lag d0 + l / nch, channel l % nch) -- so one MFMA forms LP lag blocks instead of one in a corner of the tile (a
// four-channel plant: 4 x fewer matrix instructions; what a trajectory of thousands of steps is bound by).  A "lag" of the
// distribution over waves and workgroups is then a group of LP lags.
// TILES: G goes straight into the accumulator-tile layout the register-resident cold-solve kernel loads through KParams::gpre
// (tile (I, J), I >= J, at (I (I + 1) / 2 + J) * 256; double 4 * lane + j = register j of lane (l4, l15) = K[16 J + l4 + 4 j][16 I + l15];
// diagonal tiles filled on both sides; rows / columns r .. n16 - 1, n16 = 16 NT: zero) in the identity component order (iperm, dmaxbits
// unused): the structured Gram of plants with other than two or four channels, and of trajectories beyond that kernel's LDS.
template <bool PACKED, bool TILES>
__device__ __forceinline__ void rr2_gram_body(const KParams& P, const double* __restrict__ u_d, const double* __restrict__ y_d,
                                              const int* __restrict__ iperm, double* __restrict__ ws, long long stride,
                                              int n16, unsigned long long* __restrict__ dmaxbits) {
  __shared__ __attribute__((aligned(16))) double xc[RR2_XCAP];
  __shared__ int ipl[1024];                                                 // component -> row of G (a global load per stored entry
                                                                            // put one memory round trip into every step of the walk)
  const long long b = blockIdx.y;
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int m = P.m, p = P.p, nch = P.nch, c = P.c, Ln = P.Ln, r = P.r;
  __shared__ int prl[1024];                                                 // pk_row(i): start of row i of G
  if constexpr (TILES) {                                                    // entry K[x][y], x <= y, at prl[x] + ipl[y]
    for (int i = tid; i < n16; i += nthr) {
      prl[i] = (i >> 4) * 256 + 4 * ((i & 3) * 16) + ((i & 15) >> 2);
      ipl[i] = ((i >> 4) * ((i >> 4) + 1) / 2) * 256 + 4 * (i & 15);
    }
  } else {
    for (int i = tid; i < r; i += nthr) { ipl[i] = iperm[i]; prl[i] = (int)pk_row((size_t)i); }
  }
  for (int i = tid; i < RR2_XCAP; i += nthr) xc[i] = 0.0;                   // (masked MFMA terms multiply whatever lies behind a chunk by zero: finite)
  const double* ud = u_d + b * (long long)P.N * m;
  const double* yd = y_d + b * (long long)P.N * p;
  double* G = ws + b * stride;
  const int lane = tid & 63, l15 = lane & 15, l4 = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), nwave = nthr >> 6;
  const int dg = blockIdx.x * nwave * RR2_SL;
  const int nat = PACKED ? 1 : (nch + 15) >> 4;
  const int LP = PACKED ? 16 / nch : 1;                                     // lags per tile
  const int djA = PACKED ? ((l15 / nch < LP) ? l15 / nch : LP - 1) : 0;    // the A-operand row of this lane: lag offset ...
  int djq[4], aq[4];                                                        // ... and of the accumulator rows l4 + 4q
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    djq[q] = PACKED ? (l4 + 4 * q) / nch : 0;
    aq[q] = PACKED ? (l4 + 4 * q) - djq[q] * nch : l4 + 4 * q;
  }
  const int TCH = ((RR2_XCAP / nch) - Ln - 3) & ~3;                        // time steps per chunk next to the lag overlap (+ 3 rows of slack)
  const int nw = Ln - 1;
  double* xh = xc;                                                          // walk: rows 0 .. Ln-2
  double* xt = xc + nw * nch;                                               //       rows c .. c+Ln-2
  double dmx = 0.0;
  if (blockIdx.x == 0) {                                                    // padding rows r .. n16-1: zero (their pivots are skipped)
    if constexpr (TILES) {
      __syncthreads();                                                      // (the tables)
      for (int i = r; i < n16; ++i)
        for (int j = tid; j <= i; j += nthr) {
          G[prl[j] + ipl[i]] = 0.0;
          if ((j >> 4) == (i >> 4)) G[prl[i] + ipl[j]] = 0.0;
        }
    } else {
      for (int i = r; i < n16; ++i)
        for (int j = tid; j <= i; j += nthr) G[pk_row(i) + j] = 0.0;
    }
  }
  for (int at = 0; at < nat; ++at)
    for (int bt = 0; bt < nat; ++bt) {
      d4 acc[RR2_SL];
#pragma unroll
      for (int sl = 0; sl < RR2_SL; ++sl) acc[sl] = d4{0.0, 0.0, 0.0, 0.0};
      const int al = PACKED ? l15 - djA * nch : 16 * at + l15;
      const int ca = (al < nch) ? al : nch - 1;                             // clamped: entries past nch are not stored
      const int cb = (16 * bt + l15 < nch) ? 16 * bt + l15 : nch - 1;
      // (Staging with all of a thread's 16 loads of a chunk in flight at once, or in two rounds of eight, or issued ahead of
      //  the previous chunk's MFMA loop: 562 / 562 / 710 us against 565 as it stands -- the first two spill 28 - 120 B, the
      //  last 228 B: with the accumulators the kernel sits at 113 of the 128 registers four waves per SIMD leave.  The staging
      //  is not what the launch waits for: knocked out, the lag sums alone take 413 us = 72 % of the matrix pipe at ~2.0 GHz.)
      for (int t0 = 0; t0 < c; t0 += TCH) {
        const int nt = (c - t0) < TCH ? (c - t0) : TCH;
        const int nload = nt + Ln - 1;
        __syncthreads();
        stage_trajectory<4>(P, ud, yd, t0, nload, xc);
        __syncthreads();
        // lane (l15, l4): B operand x_b[t], A operand x_a[t + d], t = s4 + l4; terms t >= nt are masked by a zero in B (what
        // A reads there -- rows up to nload + 2 -- is finite: the region was zero-filled, later chunks leave older data)
        const double* pb = xc + l4 * nch + cb;
        if constexpr (PACKED) {
          const double* pas[RR2_SL];                                        // (a lane's lag clamped to Ln - 1: inside what was staged)
#pragma unroll
          for (int sl = 0; sl < RR2_SL; ++sl) {
            const int dl = LP * (dg + wave + nwave * sl) + djA;
            pas[sl] = xc + (l4 + (dl < Ln ? dl : Ln - 1)) * nch + ca;
          }
          for (int s4 = 0; s4 < nt; s4 += 4) {
            const double bv = (s4 + l4 < nt) ? pb[s4 * nch] : 0.0;
#pragma unroll
            for (int sl = 0; sl < RR2_SL; ++sl)
              if (LP * (dg + wave + nwave * sl) < Ln) acc[sl] = rr2_mfma(pas[sl][s4 * nch], bv, acc[sl]);   // (wave-uniform)
          }
        } else {
          const double* pa = xc + (l4 + dg + wave) * nch + ca;
          const int lstep = nwave * nch;
          for (int s4 = 0; s4 < nt; s4 += 4) {
            const double bv = (s4 + l4 < nt) ? pb[s4 * nch] : 0.0;
#pragma unroll
            for (int sl = 0; sl < RR2_SL; ++sl)
              if (dg + wave + nwave * sl < Ln) acc[sl] = rr2_mfma(pa[s4 * nch + sl * lstep], bv, acc[sl]);   // (wave-uniform)
          }
        }
      }
      __syncthreads();
      stage_trajectory<4>(P, ud, yd, 0, nw, xh);
      stage_trajectory<4>(P, ud, yd, c, nw, xt);
      __syncthreads();
      const double* xsel = (l4 == 0) ? xt : xh;
      const double sga = (l4 == 0) ? 1.0 : (l4 == 1 ? -1.0 : 0.0), sgb = (l4 < 2) ? 1.0 : 0.0;
#pragma unroll
      for (int sl = 0; sl < RR2_SL; ++sl) {
        const int d0 = LP * (dg + wave + nwave * sl);                       // (first) lag of the tile
        if (d0 >= Ln) continue;
        for (int k = 0; k + d0 < Ln; ++k) {
          if (k > 0) {
            const int ra = k - 1 + d0 + djA;                                // (rows past the walk: entries that are no longer stored)
            acc[sl] = rr2_mfma(sga * xsel[(ra < nw ? ra : nw - 1) * nch + ca], sgb * xsel[(k - 1) * nch + cb], acc[sl]);
          }
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int a = 16 * at + aq[q], bb = 16 * bt + l15, d = d0 + djq[q];
            if (a < nch && bb < nch && djq[q] < LP && k + d < Ln && (d > 0 || a >= bb)) {
              if constexpr (TILES) {
                const int hi = (k + d) * nch + a, lo = k * nch + bb;        // (hi >= lo)
                G[prl[lo] + ipl[hi]] = acc[sl][q];
                if ((hi >> 4) == (lo >> 4)) G[prl[hi] + ipl[lo]] = acc[sl][q];
              } else {
                const int pi = ipl[(k + d) * nch + a], pj = ipl[k * nch + bb];
                const int hi = pi > pj ? pi : pj, lo = pi > pj ? pj : pi;
                G[prl[hi] + lo] = acc[sl][q];
                if (d == 0 && a == bb) dmx = fmax(dmx, acc[sl][q]);
              }
            }
          }
        }
      }
    }
  if (!TILES && dg + wave == 0) {                                           // the wave that owns lag 0 has met every diagonal entry
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) dmx = fmax(dmx, __shfl_xor(dmx, off, 64));
    if (lane == 0) atomicMax(dmaxbits + 4 * b, (unsigned long long)__double_as_longlong(fmax(dmx, 0.0)));   // (four words per instance)
  }
}
__global__ __launch_bounds__(256, 4) void rr2_gram_kernel(KParams P, const double* __restrict__ u_d, const double* __restrict__ y_d,
                                                       const int* __restrict__ iperm, double* __restrict__ ws, long long stride,
                                                       int n16, unsigned long long* __restrict__ dmaxbits) {
  rr2_gram_body<false, false>(P, u_d, y_d, iperm, ws, stride, n16, dmaxbits);
}
__global__ __launch_bounds__(256, 4) void rr2_gram_packed_kernel(KParams P, const double* __restrict__ u_d, const double* __restrict__ y_d,
                                                              const int* __restrict__ iperm, double* __restrict__ ws, long long stride,
                                                              int n16, unsigned long long* __restrict__ dmaxbits) {
  rr2_gram_body<true, false>(P, u_d, y_d, iperm, ws, stride, n16, dmaxbits);
}
// ... into the tiles of the register-resident cold-solve kernel (n16 = 16 NT <= 1024)
__global__ __launch_bounds__(256, 4) void rr2_gram_tiles_kernel(KParams P, const double* __restrict__ u_d, const double* __restrict__ y_d,
                                                             double* __restrict__ gpre, long long gstride, int n16) {
  rr2_gram_body<false, true>(P, u_d, y_d, nullptr, gpre, gstride, n16, nullptr);
}
__global__ __launch_bounds__(256, 4) void rr2_gram_tiles_packed_kernel(KParams P, const double* __restrict__ u_d, const double* __restrict__ y_d,
                                                                    double* __restrict__ gpre, long long gstride, int n16) {
  rr2_gram_body<true, true>(P, u_d, y_d, nullptr, gpre, gstride, n16, nullptr);
}
// ---------------------------------------------------------------------------------------------------------------
// The same for FOUR channels (the four-tank plant of the reference's example) on v_mfma_f64_4x4x4: four independent 4 x 4 x 4
// products per instruction -- block blk of lane group l15 = 4 blk + i is the lag block C_d, d = 4 g + blk: A = x_i[t + k + d],
// B = x_j[t + k], D_blk[i][j] -- at a quarter of the cycles of a 16x16x4 whose tile holds the same four lags (64 clocks on gfx950).
// One accumulator register per group of four lags; the walk is the same instruction with the contraction slots (tail term,
// -head term, 0, 0); every lane stores its one entry per step.  Writes the cold-solve kernel's tiles (see TILES above): what a
// trajectory beyond that kernel's LDS takes.  grid = (ceil(ceil(Ln / 4) / (4 RR2_C4_SL)), batch), 256 threads.
// ---------------------------------------------------------------------------------------------------------------
constexpr int RR2_C4_SL = 4;        // groups of four lags per wave
__global__ __launch_bounds__(256, 4) void rr2_gram_tiles_c4_kernel(KParams P, const double* __restrict__ u_d, const double* __restrict__ y_d,
                                                                double* __restrict__ gpre, long long gstride, int n16) {
  __shared__ __attribute__((aligned(16))) double xc[RR2_XCAP];
  __shared__ int ipl[1024], prl[1024];                                      // entry K[x][y], x <= y, at prl[x] + ipl[y]
  const long long b = blockIdx.y;
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int m = P.m, p = P.p, c = P.c, Ln = P.Ln, r = P.r;
  constexpr int nch = 4;
  for (int i = tid; i < n16; i += nthr) {
    prl[i] = (i >> 4) * 256 + 4 * ((i & 3) * 16) + ((i & 15) >> 2);
    ipl[i] = ((i >> 4) * ((i >> 4) + 1) / 2) * 256 + 4 * (i & 15);
  }
  for (int i = tid; i < RR2_XCAP; i += nthr) xc[i] = 0.0;                   // (masked terms multiply whatever lies behind a chunk by zero: finite)
  const double* ud = u_d + b * (long long)P.N * m;
  const double* yd = y_d + b * (long long)P.N * p;
  double* G = gpre + b * gstride;
  const int lane = tid & 63, l15 = lane & 15, l4 = lane >> 4, blk = l15 >> 2, ij = l15 & 3;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), nwave = nthr >> 6;
  const int gtot = (Ln + 3) >> 2;
  const int dg = blockIdx.x * nwave * RR2_C4_SL;
  const int TCH = ((RR2_XCAP / nch) - Ln - 3) & ~3;
  const int nw = Ln - 1;
  double* xh = xc;                                                          // walk: rows 0 .. Ln-2
  double* xt = xc + nw * nch;                                               //       rows c .. c+Ln-2
  __syncthreads();                                                          // (the tables)
  if (blockIdx.x == 0)                                                      // padding rows r .. n16-1: zero
    for (int i = r; i < n16; ++i)
      for (int j = tid; j <= i; j += nthr) {
        G[prl[j] + ipl[i]] = 0.0;
        if ((j >> 4) == (i >> 4)) G[prl[i] + ipl[j]] = 0.0;
      }
  double acc[RR2_C4_SL];
  int dl[RR2_C4_SL];                                                        // this lane's lag in group sl, clamped to what is staged
#pragma unroll
  for (int sl = 0; sl < RR2_C4_SL; ++sl) {
    acc[sl] = 0.0;
    const int d = 4 * (dg + wave + nwave * sl) + blk;
    dl[sl] = d < Ln ? d : Ln - 1;
  }
  for (int t0 = 0; t0 < c; t0 += TCH) {
    const int nt = (c - t0) < TCH ? (c - t0) : TCH;
    __syncthreads();
    stage_trajectory<4>(P, ud, yd, t0, nt + Ln - 1, xc);
    __syncthreads();
    const double* pb = xc + l4 * nch + ij;
    for (int s4 = 0; s4 < nt; s4 += 4) {
      const double bv = (s4 + l4 < nt) ? pb[s4 * nch] : 0.0;
#pragma unroll
      for (int sl = 0; sl < RR2_C4_SL; ++sl)
        if (dg + wave + nwave * sl < gtot)                                  // (wave-uniform)
          acc[sl] = __builtin_amdgcn_mfma_f64_4x4x4f64(xc[(s4 + l4 + dl[sl]) * nch + ij], bv, acc[sl], 0, 0, 0);
    }
  }
  __syncthreads();
  stage_trajectory<4>(P, ud, yd, 0, nw, xh);
  stage_trajectory<4>(P, ud, yd, c, nw, xt);
  __syncthreads();
  // accumulator: lane (l4 = a, l15 = 4 blk + bb) holds C_d(a, bb), d = 4 g + blk
  const double* xsel = (l4 == 0) ? xt : xh;
  const double sga = (l4 == 0) ? 1.0 : (l4 == 1 ? -1.0 : 0.0), sgb = (l4 < 2) ? 1.0 : 0.0;
#pragma unroll
  for (int sl = 0; sl < RR2_C4_SL; ++sl) {
    const int g = dg + wave + nwave * sl;
    if (g >= gtot) continue;                                                // (wave-uniform)
    const int d = 4 * g + blk, a = l4, bb = ij;
    for (int k = 0; k + 4 * g < Ln; ++k) {
      if (k > 0) {
        const int ra = k - 1 + dl[sl];                                      // (rows past the walk: entries that are no longer stored)
        acc[sl] = __builtin_amdgcn_mfma_f64_4x4x4f64(sga * xsel[(ra < nw ? ra : nw - 1) * nch + ij], sgb * xsel[(k - 1) * nch + ij], acc[sl], 0, 0, 0);
      }
      if (k + d < Ln && (d > 0 || a >= bb)) {
        const int hi = (k + d) * nch + a, lo = k * nch + bb;
        G[prl[lo] + ipl[hi]] = acc[sl];
        if ((hi >> 4) == (lo >> 4)) G[prl[hi] + ipl[lo]] = acc[sl];
      }
    }
  }
}

// grid.x of the two (a workgroup's four waves own RR2_SL lags -- groups of 16 / nch lags -- each)
inline unsigned rr2_gram_grid(int Ln, int nch) {
  const int groups = nch <= 8 ? (Ln + 16 / nch - 1) / (16 / nch) : Ln;
  return (unsigned)((groups + 4 * RR2_SL - 1) / (4 * RR2_SL));
}

// ---------------------------------------------------------------------------------------------------------------
// Rows below the diagonal block of the panel at column c0 (RR2_NB columns), in ONE pass over the panel:
//   P(i, c) = A(i, c) - sum_{j < c0, j live} L(i, j) L(c, j)        left-looking update with every live column in front of it
//   X(i, :) = P(i, :) Minv'                                          Minv = inverse of the block's factor (panel kernel)
// grid = (row groups, batch), 256 threads: a wave owns RT row tiles x the four column tiles.  Accumulators transposed
// (register q of lane (l4, l15) = entry [panel column l4 + 4q][row l15]), which makes a finished P tile directly the B operand
// of the multiplication with Minv.  Per live 16-column chunk the panel's own 64 rows -- the operand every wave of every
// workgroup of the instance needs -- are staged ONCE per workgroup in LDS (double-buffered, one barrier per chunk), the waves'
// own rows are 32-byte pieces of packed rows loaded two chunks ahead of the MFMAs that consume them.  (As two kernels -- update
// with a store of P, then a solve that read P back -- the panel went through HBM four times per step instead of twice.)
// ---------------------------------------------------------------------------------------------------------------
constexpr int RR2_PLD = 18;         // doubles per staged panel row (16 + 2: the 16 rows of a tile land in different banks)
// `group`: which 4 RT row tiles below the block this workgroup takes; `mi_ready`: Minv of the block is already in Mi (the fused
// kernel below), else it is read from F.m64.  pl: 3 x 64 x RR2_PLD doubles, Mi: 10 tiles of RR2_TSZ (LDS).  All 256 threads.
template <int RT, bool TRACK>
__device__ __forceinline__ void rr2_update_body(const Rr2Chol& F, long long b, int c0, int n16, int group, bool mi_ready,
                                                double (*pl)[64 * RR2_PLD], double (*Mi)[RR2_TSZ]) {
  if (c0 + RR2_NB >= n16) return;                                           // (workgroup-uniform) no row below the block
  double* A = F.ws + b * F.stride + F.off;
  const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, l4 = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tp = c0 >> 4, nt = n16 >> 4;
  const int tb = tp + 4;                                                    // first row tile below the block (<= 63)
  // the row tiles below the block that are not retired, dealt RT at a time to the waves of the groups
  constexpr bool track = TRACK;                                             // (Rr2Chol::res)
  const int par = (c0 / RR2_NB) & 1;
  unsigned long long* deadw = track ? F.dead + b * F.dead_stride : nullptr;
  const unsigned long long dead = track ? deadw[par] : 0ull;
  if (track && group == 0 && tid == 0 && dead != 0ull) atomicOr(deadw + (par ^ 1), dead);      // the set only grows
  unsigned long long cand = ((nt >= 64) ? ~0ull : ((1ull << nt) - 1ull)) & ~((1ull << tb) - 1ull) & ~dead;
  if (__builtin_popcountll(cand) <= group * 4 * RT) return;                 // (workgroup-uniform: no row tile left for this group)
  for (int q = (group * 4 + wave) * RT; q > 0 && cand != 0ull; --q) cand &= cand - 1ull;       // (wave-uniform)
  int Ts[RT];
  unsigned long long live = F.live[b * F.l_stride] & ((1ull << tp) - 1ull);   // chunks in front of the panel (tp <= 63)
  // Column order inside a 16-column tile of the accumulators: lane l15 feeds panel row pi(l15) = 4 (l15 % 4) + l15 / 4 as the
  // A operand, so that register q of lane (l4, l15) holds panel column 4 l4 + q (not l4 + 4 q) of the tile: the FOUR registers
  // of a lane are four consecutive entries of a packed row -- one 32-byte access per lane and tile in the epilogue instead of
  // four 8-byte ones at a stride of four (with those the epilogue alone took ~100 us of a step: 117 us at 4 chunks of update).
  const int pil = 4 * (l15 & 3) + (l15 >> 2);
  // staging role of this thread: 32 bytes of panel row (tid >> 2), columns 4 (tid & 3) .. + 3 of the chunk
  const int srow = tid >> 2, spart = tid & 3;
  const double* sptr = A + pk_row((size_t)(c0 + srow)) + 4 * spart;
  const int soff = srow * RR2_PLD + 4 * spart;
  const double* mrow[RT];
  bool on[RT];
#pragma unroll
  for (int s = 0; s < RT; ++s) {
    on[s] = cand != 0ull;
    const int T = on[s] ? __builtin_ctzll(cand) : nt - 1;
    if (on[s]) cand &= cand - 1ull;
    Ts[s] = T;
    mrow[s] = A + (size_t)128 * T * (T + 1) + (size_t)l15 * 16 * (T + 1) + 4 * l4;
  }
  d4 acc[RT][4];
#pragma unroll
  for (int s = 0; s < RT; ++s)
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) acc[s][ct] = d4{0.0, 0.0, 0.0, 0.0};
  if (!mi_ready) {   // Minv of the block (written by the panel kernel of this step): ten 8-byte loads per thread
    const double* m64 = F.m64 + b * F.m64_stride + (size_t)(c0 / RR2_NB) * (RR2_NB * RR2_NB);
    double mv[10];
    const int rr = (tid >> 4) & 15, cc = tid & 15;
#pragma unroll
    for (int ti = 0; ti < 10; ++ti) {
      const int t = ti < 1 ? 0 : (ti < 3 ? 1 : (ti < 6 ? 2 : 3)), u = ti - t * (t + 1) / 2;
      mv[ti] = m64[(16 * t + rr) * RR2_NB + 16 * u + cc];
    }
#pragma unroll
    for (int ti = 0; ti < 10; ++ti) Mi[ti][rr * RR2_TLD + cc] = mv[ti];
  }
  // operands two live chunks ahead of the MFMAs that consume them (one chunk ahead every chunk waited out a memory round trip:
  // 64 % of the wave-cycles parked, the matrix pipe 26 % busy)
  d4 xa[3][RT], sreg[3];
  auto load = [&](int buf) __attribute__((always_inline)) {                 // next live chunk -> buffer `buf`; false when there is none
    if (live == 0ull) return false;
    const int jc = __builtin_ctzll(live);
    live &= live - 1ull;
    sreg[buf] = *reinterpret_cast<const d4*>(sptr + 16 * jc);
#pragma unroll
    for (int s = 0; s < RT; ++s) xa[buf][s] = *reinterpret_cast<const d4*>(mrow[s] + 16 * jc);
    return true;
  };
  auto stage = [&](int buf) __attribute__((always_inline)) {
    *reinterpret_cast<d2*>(&pl[buf][soff]) = d2{sreg[buf][0], sreg[buf][1]};
    *reinterpret_cast<d2*>(&pl[buf][soff + 2]) = d2{sreg[buf][2], sreg[buf][3]};
  };
  auto fma_chunk = [&](int buf) __attribute__((always_inline)) {
    if (!on[0]) return;                                                     // (wave-uniform) an idle wave only stages
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
      const double* pp = &pl[buf][(16 * ct + pil) * RR2_PLD + 4 * l4];
      const d2 p0 = *reinterpret_cast<const d2*>(pp), p1 = *reinterpret_cast<const d2*>(pp + 2);
#pragma unroll
      for (int s = 0; s < RT; ++s)
        if (on[s]) {                                                        // (wave-uniform)
          acc[s][ct] = rr2_mfma(p0[0], xa[buf][s][0], acc[s][ct]);
          acc[s][ct] = rr2_mfma(p0[1], xa[buf][s][1], acc[s][ct]);
          acc[s][ct] = rr2_mfma(p1[0], xa[buf][s][2], acc[s][ct]);
          acc[s][ct] = rr2_mfma(p1[1], xa[buf][s][3], acc[s][ct]);
        }
    }
  };
  {
    int have = 0;                                                           // chunks loaded and not yet consumed (0 .. 2 here)
    if (load(0)) have = 1;
    if (have == 1 && load(1)) have = 2;
    // steady state: buffer i % 3 is consumed while (i + 1) % 3 is in flight and (i + 2) % 3 is issued
    while (have > 0) {
      if (load(2)) ++have;
      stage(0); __syncthreads(); fma_chunk(0); --have;
      if (have == 0) break;
      if (load(0)) ++have;
      stage(1); __syncthreads(); fma_chunk(1); --have;
      if (have == 0) break;
      if (load(1)) ++have;
      stage(2); __syncthreads(); fma_chunk(2); --have;
    }
  }
  __syncthreads();                                                          // Mi is in LDS (and every wave is through with pl)
  // P' = A' - acc, then X' = Minv P' tile row by tile row, in place.  In the permuted column order both the contraction index
  // (register e of lane (l4, .) = column 4 l4 + e of tile u) and the output index (row a of the MFMA = column pi(a) of tile ct)
  // of the multiplication with Minv follow: the A operand is Minv(ct,u)[pi(l15)][4 l4 + e].
#pragma unroll
  for (int s = 0; s < RT; ++s) {
    if (!on[s]) continue;
    const int i = 16 * Ts[s] + l15;
    double* Ai = A + pk_row((size_t)i) + c0 + 4 * l4;
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) acc[s][ct] = *reinterpret_cast<const d4*>(Ai + 16 * ct) - acc[s][ct];
    d4 x[4];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) x[ct] = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
          if (ct >= u) x[ct] = rr2_mfma(Mi[ct * (ct + 1) / 2 + u][pil * RR2_TLD + 4 * l4 + e], acc[s][u][e], x[ct]);
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) *reinterpret_cast<d4*>(Ai + 16 * ct) = x[ct];
    if (track) {                                                            // residual diagonal of these 16 rows; retire the tile?
      double ss = 0.0;
#pragma unroll
      for (int ct = 0; ct < 4; ++ct)
#pragma unroll
        for (int e = 0; e < 4; ++e) ss = fma(x[ct][e], x[ct][e], ss);
      ss += __shfl_xor(ss, 16, 64);
      ss += __shfl_xor(ss, 32, 64);
      double* res = F.res + b * F.res_stride;
      const double d = ((c0 == 0) ? A[pk_row((size_t)i) + i] : res[i]) - ss;
      if (l4 == 0) res[i] = d;
      const double tol = RR2_RETIRE * rr2_bits_to_double(F.dmax[b * F.d_stride]);
      if (__builtin_amdgcn_ballot_w64(d > tol) == 0ull) {                    // (wave-uniform) every row of the tile is at its rounding floor
        const int rowlen = 16 * (Ts[s] + 1);
        double* Ar = A + pk_row((size_t)i);
        for (int col = c0 + RR2_NB + 4 * l4; col < rowlen; col += 16) *reinterpret_cast<d4*>(Ar + col) = d4{0.0, 0.0, 0.0, 0.0};
        if (lane == 0) atomicOr(deadw + (par ^ 1), 1ull << Ts[s]);
      }
    }
  }
}

template <int RT>
__global__ __launch_bounds__(256, 2) void rr2_chol_update_kernel(Rr2Chol F, int c0) {
  __shared__ __attribute__((aligned(16))) double pl[3][64 * RR2_PLD];
  __shared__ __attribute__((aligned(16))) double Mi[10][RR2_TSZ];           // tile (t, u), u <= t, of Minv at index t (t + 1) / 2 + u, row-major (RR2_TLD)
  const long long b = blockIdx.y;
  const int n16 = F.n_inst ? ((F.n_inst[b * F.n_stride] + 15) & ~15) : F.n16;
  if (F.res != nullptr) rr2_update_body<RT, true>(F, b, c0, n16, (int)blockIdx.x, false, pl, Mi);
  else rr2_update_body<RT, false>(F, b, c0, n16, (int)blockIdx.x, false, pl, Mi);
}

// Left-looking update of up to three tiles of the diagonal block by one wave (the tiles (S0,T0), (S1,T1), (S2,T2) of the block,
// S >= T, -1: none; chosen so that a wave touches few distinct tile rows): P(s,t) = A(s,t) - sum_{j < c0, j live} L(s rows, j)
// L(t rows, j)', result into the LDS tile (strict upper triangle of a diagonal tile: zero).
template <int S0, int T0, int S1, int T1, int S2, int T2>
__device__ __forceinline__ void rr2_diag_update(const double* A, double (*Dt)[RR2_TSZ], unsigned long long livemask, int c0, int n4) {
  constexpr int SS[3] = {S0, S1, S2};
  constexpr int TT[3] = {T0, T1, T2};
  const int lane = threadIdx.x & 63, l15 = lane & 15, l4 = lane >> 4;
  const int tp = c0 >> 4;
  unsigned long long live = livemask & ((1ull << tp) - 1ull);
  const double* rowp[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int T = (t < n4) ? tp + t : tp + n4 - 1;
    rowp[t] = A + (size_t)128 * T * (T + 1) + (size_t)l15 * 16 * (T + 1) + 4 * l4;
  }
  d4 acc[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) acc[k] = d4{0.0, 0.0, 0.0, 0.0};
  // six live chunks per round, every load of the round in flight before its first MFMA (one chunk ahead of the MFMAs the
  // loads of a chunk waited a full memory round trip for 12 MFMAs: 30 us of a late panel)
  constexpr int GC = 6;
  while (live != 0ull) {
    d4 xr[GC][4];
    int jj[GC];
#pragma unroll
    for (int u = 0; u < GC; ++u) {
      const bool ok = live != 0ull;
      const int jc = ok ? __builtin_ctzll(live) : 0;
      if (ok) live &= live - 1ull;
      jj[u] = ok ? jc : -1;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        bool need = false;
#pragma unroll
        for (int k = 0; k < 3; ++k) need = need || (SS[k] == t) || (SS[k] >= 0 && TT[k] == t);
        if (need) xr[u][t] = *reinterpret_cast<const d4*>(rowp[t] + 16 * jc);
      }
    }
#pragma unroll
    for (int u = 0; u < GC; ++u) {
      if (jj[u] < 0) continue;                                              // (wave-uniform)
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        if (SS[k] < 0) continue;
        if (SS[k] >= n4) continue;                                          // (wave-uniform) tile past the end of the matrix
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[k] = rr2_mfma(xr[u][TT[k]][e], xr[u][SS[k]][e], acc[k]);
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    if (SS[k] < 0) continue;
    if (SS[k] >= n4) continue;
    const int i = c0 + 16 * SS[k] + l15;
    const double* Ai = A + pk_row((size_t)i) + c0 + 16 * TT[k];
    double* Dst = Dt[SS[k] * (SS[k] + 1) / 2 + TT[k]];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int cc = l4 + 4 * q;
      // a diagonal tile is kept FULL and symmetric (rr2_tile_factor works on the whole tile): the entry above the diagonal is
      // the stored one mirrored (the accumulator is bitwise symmetric: both entries sum the same products in the same order)
      const bool lower = (SS[k] > TT[k]) || (cc <= l15);
      const double av = lower ? Ai[cc] : A[pk_row((size_t)(c0 + 16 * SS[k] + cc)) + c0 + 16 * TT[k] + l15];
      Dst[l15 * RR2_TLD + cc] = av - acc[k][q];
    }
  }
}

// One wave factors a full symmetric 16 x 16 tile held row-major in LDS, right-looking with the tile in ONE accumulator
// (register q of lane (l4, l15) = D[l4 + 4q][l15]): per column the pivot comes out of the accumulator by v_readlane, and the
// rank-1 update of the whole tile is a single MFMA whose operands are row c of the tile itself -- it sits in register c / 4 of
// the lanes with l4 = c % 4, exactly the lanes that feed contraction slot c % 4 -- scaled by 1 / sqrt(pivot).  No LDS round
// trip on the chain (psd_tile_factor: two per column, ~550 cycles; here ~200).  Pivots <= tol are skipped (column zeroed).
// The inverse rides along in a second accumulator R that starts as the identity: row c of Mt = S L~^-1 is row c of R over
// the pivot's square root, and the rows below lose u (x) that row -- one more MFMA per column, independent of the first.
// (Until round 4's end Mt came from a forward substitution by 16 lanes afterwards: 120 dependent LDS reads + FMAs per tile,
// about as long as the factorisation itself.)
// Leaves L in the lower triangle of Dg (zeros above), Ms[k][m] = Mt[m][k] (rows of skipped pivots zero), Dinv, the pivot flags.
__device__ __forceinline__ void rr2_tile_factor(double* Dg, double* Ms, double* Dinv, double tol, int* skipout, double* candout) {
  const int lane = threadIdx.x & 63, l15 = lane & 15, l4 = lane >> 4;
  d4 acc, rr;
#pragma unroll
  for (int q = 0; q < 4; ++q) { acc[q] = Dg[(l4 + 4 * q) * RR2_TLD + l15]; rr[q] = (l4 + 4 * q == l15) ? 1.0 : 0.0; }
  static_for<16>([&](auto cc) __attribute__((always_inline)) {
    constexpr int c = cc();
    constexpr int q = c >> 2, slot = c & 3;
    const double dk = lane_value_f64(acc[q], slot * 16 + c);               // D[c][c], the same in every lane
    const bool sk = !(dk > tol);
    const double dks = sk ? 1.0 : dk;
    const double y0 = __builtin_amdgcn_rsq(dks);
    const double ye = fma(-dks * y0, y0, 1.0);
    const double y1 = fma(0.5 * y0, ye, y0);
    const double ye1 = fma(-dks * y1, y1, 1.0);
    const double inv = sk ? 0.0 : fma(0.5 * y1, ye1, y1);
    const bool mine = l4 == slot;
    const double u = (mine && l15 >= c) ? acc[q] * inv : 0.0;                // u_i = D[c][i] / sqrt(pivot), i = l15 >= c
    const double y = mine ? rr[q] * inv : 0.0;                               // Mt[c][j] = R[c][j] / sqrt(pivot), j = l15
    if (mine) { Dg[l15 * RR2_TLD + c] = u; Ms[l15 * RR2_TLD + c] = y; }     // L(i, c), zeros above the diagonal; Ms[k = j][m = c]
    acc = rr2_mfma(-u, u, acc);
    rr = rr2_mfma(-u, y, rr);
    if (lane == 0) { skipout[c] = sk ? 1 : 0; Dinv[c] = inv; candout[c] = dk; }
  });
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");                     // in-wave hand-off through LDS
}

// ---------------------------------------------------------------------------------------------------------------
// Panel step, part 1: the (up to) 64 x 64 diagonal block at column c0 is brought up to date (the left-looking update with the
// live columns in front of it, its ten lower tiles dealt to the four waves; operands one chunk ahead) and factored in LDS --
// 16 x 16 tiles, each diagonal tile by one wave with the skipped-pivot rule (psd_tile_factor, which also yields Mt = S L~^-1), the tiles below it and the
// trailing tiles of the block by MFMA.  Stores the block's factor in place, the inverse Minv of the block in `m64` (for part 2
// and for the substitutions of the solve), the pivot flags and the live-chunk bits.  grid = (1, batch), 256 threads: ONE workgroup per instance (the block is factored in place).
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void rr2_panel_body(const Rr2Chol& F, long long b, int c0, int n16, double (*Dt)[RR2_TSZ],
                                               double (*Ms)[RR2_TSZ], double (*Mi)[RR2_TSZ], double* Dinv, int* skipl, double* candl) {
  if (c0 >= n16) return;
  double* A = F.ws + b * F.stride + F.off;
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int lane = tid & 63, l15 = lane & 15, l4 = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), nwave = nthr >> 6;
  const int tp = c0 >> 4, nt = n16 >> 4;
  const int n4 = (nt - tp) < 4 ? (nt - tp) : 4;                             // tiles across the panel
  const double tol = (F.tol_inst ? F.tol_inst[b] : F.tol_rel) * rr2_bits_to_double(F.dmax[b * F.d_stride]);
  // retired row tiles of the block (Rr2Chol::res): their rows hold zeros from the column they were retired at and count as zero
  const unsigned dmask = F.res ? (unsigned)((F.dead[b * F.dead_stride + ((c0 / RR2_NB) & 1)] >> tp) & ((1ull << n4) - 1ull)) : 0u;
  if (dmask == (1u << n4) - 1u) {                                           // (workgroup-uniform) nothing left to factor
    if (tid < 16 * n4 && c0 + tid < F.nflag) F.skip[b * F.s_stride + c0 + tid] = 1;
    if (F.cand && tid < 16 * n4 && c0 + tid < F.nflag) F.cand[b * F.cand_stride + c0 + tid] = 0.0;
    double* m64 = F.m64 + b * F.m64_stride + (size_t)(c0 / RR2_NB) * (RR2_NB * RR2_NB);
    for (int e = tid; e < RR2_NB * RR2_NB; e += nthr) m64[e] = 0.0;
    return;
  }
#ifdef RR2_PANEL_PROBE
  long long tp_[12]; int np_ = 0;
#define RR2_STAMP() do { if (tid == 0 && np_ < 12) tp_[np_++] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
#else
#define RR2_STAMP() do {} while (0)
#endif
  RR2_STAMP();
  // ---- the diagonal block, updated, into LDS (strict upper triangle of the diagonal tiles zero, absent tiles zero)
  for (int e = tid; e < 10 * RR2_TSZ; e += nthr) Dt[e / RR2_TSZ][e % RR2_TSZ] = 0.0;
  __syncthreads();
  if (wave == 0) rr2_diag_update<0, 0, 1, 0, 1, 1>(A, Dt, F.live[b * F.l_stride], c0, n4);
  else if (wave == 1) rr2_diag_update<2, 0, 2, 1, 2, 2>(A, Dt, F.live[b * F.l_stride], c0, n4);
  else if (wave == 2) rr2_diag_update<3, 0, 3, 1, -1, -1>(A, Dt, F.live[b * F.l_stride], c0, n4);
  else rr2_diag_update<3, 2, 3, 3, -1, -1>(A, Dt, F.live[b * F.l_stride], c0, n4);
  for (int e = tid; e < 4 * RR2_TSZ; e += nthr) Ms[e / RR2_TSZ][e % RR2_TSZ] = 0.0;
  __syncthreads();
  if (dmask != 0u) {                                                        // (workgroup-uniform)
    for (int e = tid; e < 10 * RR2_TSZ; e += nthr) {
      const int ti = e / RR2_TSZ;
      int s = 0;
      while ((s + 1) * (s + 2) / 2 <= ti) ++s;
      const int t = ti - s * (s + 1) / 2;
      if (((dmask >> s) | (dmask >> t)) & 1u) Dt[ti][e % RR2_TSZ] = 0.0;
    }
    __syncthreads();
  }
  RR2_STAMP();     // 1: block in LDS
  for (int t = 0; t < n4; ++t) {
    if (wave == 0) rr2_tile_factor(Dt[t * (t + 1) / 2 + t], Ms[t], Dinv + 16 * t, tol, skipl + 16 * t, candl + 16 * t);
    __syncthreads();
    if (t == 0) RR2_STAMP();   // 2: first tile factored
    // tiles below the diagonal tile: X(s,t) = P(s,t) Mt'   (X'[m][i] = sum_k Mt[m][k] P'[k][i])
    if (wave >= 1 && t + wave < n4) {
      double* Pst = Dt[(t + wave) * (t + wave + 1) / 2 + t];
      double pv[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) pv[q] = Pst[l15 * RR2_TLD + l4 + 4 * q];
      d4 x = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int q = 0; q < 4; ++q) x = rr2_mfma(Ms[t][(l4 + 4 * q) * RR2_TLD + l15], pv[q], x);
#pragma unroll
      for (int q = 0; q < 4; ++q) Pst[l15 * RR2_TLD + l4 + 4 * q] = x[q];
    }
    __syncthreads();
    // trailing tiles of the block: P(s,u) -= X(s,t) X(u,t)',  t < u <= s
    {
      int idx = 0;
      for (int s = t + 1; s < n4; ++s)
        for (int u = t + 1; u <= s; ++u, ++idx) {
          if (idx % nwave != wave) continue;                                // wave-uniform
          const double* Xs = Dt[s * (s + 1) / 2 + t];
          const double* Xu = Dt[u * (u + 1) / 2 + t];
          d4 acc = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
          for (int e = 0; e < 4; ++e) acc = rr2_mfma(Xu[l15 * RR2_TLD + l4 + 4 * e], Xs[l15 * RR2_TLD + l4 + 4 * e], acc);
          double* Psu = Dt[s * (s + 1) / 2 + u];
#pragma unroll
          for (int q = 0; q < 4; ++q) Psu[l15 * RR2_TLD + l4 + 4 * q] -= acc[q];
        }
    }
    __syncthreads();
  }
  // ---- Minv = S L~^-1 of the whole block (L~: the block's factor with unit diagonal where a pivot was skipped), 16 x 16 tiles:
  //        Minv(t,t) = Mt_t,   Minv(s,t) = -Mt_s sum_{t <= u < s} L(s,u) Minv(u,t)     (by distance s - t from the diagonal)
  // With it the rows below the block are ONE multiplication X = P Minv' of independent MFMA chains (part 2) instead of a
  // forward substitution through four dependent tile solves, and a substitution with the finished factor advances 64 rows
  // per step (ddmpc_rr2_solve.hpp).  The Dt tiles of the diagonal (their factor is no longer needed in LDS form) are reused.
  RR2_STAMP();     // 3: block factored
  for (int e = tid; e < 4 * 256; e += nthr) {
    const int t = e >> 8, a = (e >> 4) & 15, bb = e & 15;
    Mi[t * (t + 1) / 2 + t][a * RR2_TLD + bb] = Ms[t][bb * RR2_TLD + a];
  }
  __syncthreads();
  for (int dist = 1; dist < n4; ++dist) {
    const int t = wave, s = wave + dist;                                    // (wave-uniform) one tile per wave and distance
    if (s < n4) {
      d4 w = d4{0.0, 0.0, 0.0, 0.0};
      for (int u = t; u < s; ++u) {
        const double* Lsu = Dt[s * (s + 1) / 2 + u];
        const double* Mut = Mi[u * (u + 1) / 2 + t];
#pragma unroll
        for (int e = 0; e < 4; ++e) w = rr2_mfma(Lsu[l15 * RR2_TLD + l4 + 4 * e], Mut[(l4 + 4 * e) * RR2_TLD + l15], w);
      }
      d4 x = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int e = 0; e < 4; ++e) x = rr2_mfma(Ms[s][(l4 + 4 * e) * RR2_TLD + l15], w[e], x);
      double* Mst = Mi[s * (s + 1) / 2 + t];
#pragma unroll
      for (int q = 0; q < 4; ++q) Mst[(l4 + 4 * q) * RR2_TLD + l15] = -x[q];
    }
    __syncthreads();
  }
  RR2_STAMP();     // 4: Minv formed
  // ---- the block's own factor, Minv, pivot flags, live-chunk bits
  for (int e = tid; e < 10 * 256; e += nthr) {
    const int ti = e >> 8, rr = (e >> 4) & 15, cc = e & 15;
    int s = 0;
    while ((s + 1) * (s + 2) / 2 <= ti) ++s;
    const int t = ti - s * (s + 1) / 2;
    if (s < n4) {
      const int i = c0 + 16 * s + rr, j = c0 + 16 * t + cc;
      if (j <= i) A[pk_row((size_t)i) + j] = Dt[ti][rr * RR2_TLD + cc];
    }
  }
  double* m64 = F.m64 + b * F.m64_stride + (size_t)(c0 / RR2_NB) * (RR2_NB * RR2_NB);
  for (int e = tid; e < RR2_NB * RR2_NB; e += nthr) {                       // 64 x 64, row-major, zero above the diagonal tiles
    const int i = e >> 6, j = e & 63, s = i >> 4, t = j >> 4;
    m64[e] = (t <= s && s < n4) ? Mi[s * (s + 1) / 2 + t][(i & 15) * RR2_TLD + (j & 15)] : 0.0;
  }
  if (tid < 16 * n4 && c0 + tid < F.nflag) F.skip[b * F.s_stride + c0 + tid] = skipl[tid];
  if (F.cand && tid < 16 * n4 && c0 + tid < F.nflag) F.cand[b * F.cand_stride + c0 + tid] = ((dmask >> (tid >> 4)) & 1u) ? 0.0 : candl[tid];
  if (tid == 0) {
    unsigned long long bits = 0ull;
    for (int t = 0; t < n4; ++t) {
      bool any = false;
      for (int q = 0; q < 16; ++q) any = any || (skipl[16 * t + q] == 0);
      if (any) bits |= 1ull << (tp + t);
    }
    F.live[b * F.l_stride] |= bits;
  }
#ifdef RR2_PANEL_PROBE
  RR2_STAMP();     // 5: stores issued
  if (tid == 0 && b == 300) printf("panel c0=%d: to LDS %lld, tile0 %lld, rest of block %lld, Minv %lld, stores %lld (x10 ns)\n", c0,
                                   tp_[1] - tp_[0], tp_[2] - tp_[1], tp_[3] - tp_[2], tp_[4] - tp_[3], tp_[5] - tp_[4]);
#endif
}

__global__ __launch_bounds__(256) void rr2_chol_panel_kernel(Rr2Chol F, int c0) {
  __shared__ __attribute__((aligned(16))) double Dt[10][RR2_TSZ];           // tile (s, t), s >= t, at index s (s + 1) / 2 + t, row-major
  __shared__ __attribute__((aligned(16))) double Ms[4][RR2_TSZ];            // Mt of the diagonal tiles, k-major
  __shared__ __attribute__((aligned(16))) double Mi[10][RR2_TSZ];           // Minv of the block, tile (s, t) row-major
  __shared__ double Dinv[64];
  __shared__ double candl[64];
  __shared__ int skipl[64];
  const long long b = blockIdx.y;
  const int n16 = F.n_inst ? ((F.n_inst[b * F.n_stride] + 15) & ~15) : F.n16;
  rr2_panel_body(F, b, c0, n16, Dt, Ms, Mi, Dinv, skipl, candl);
}

// The whole factorisation of a SMALL matrix (the reduced normal matrix T: a few panels) in one launch: one workgroup per
// instance walks the panels -- panel step, then the rows below it, group by group -- with workgroup barriers where the
// lock-step pipeline has kernel boundaries.  (As separate launches the 168-row T of cfg 5 cost 270 us: three panel launches of
// ~50 us each, two update launches, six launches that found nothing to do.)  The staging buffer of the update aliases the
// panel step's tiles.  grid = batch, 256 threads.
template <int RT>
__global__ __launch_bounds__(256, 2) void rr2_chol_small_kernel(Rr2Chol F) {
  __shared__ __attribute__((aligned(16))) double DtMs[14][RR2_TSZ];         // panel step: Dt (10 tiles) | Ms (4 tiles); update: the staging ring
  __shared__ __attribute__((aligned(16))) double Mi[10][RR2_TSZ];
  __shared__ double Dinv[64];
  __shared__ double candl[64];
  __shared__ int skipl[64];
  static_assert(sizeof(double) * 14 * RR2_TSZ >= sizeof(double) * 3 * 64 * RR2_PLD, "the staging ring fits into the panel step's tiles");
  const long long b = blockIdx.x;
  const int n16 = F.n_inst ? ((F.n_inst[b * F.n_stride] + 15) & ~15) : F.n16;
  for (int c0 = 0; c0 < n16; c0 += RR2_NB) {
    rr2_panel_body(F, b, c0, n16, DtMs, DtMs + 10, Mi, Dinv, skipl, candl);
    __syncthreads();                                                        // the block's factor, Minv (LDS) and the live bits are in place
    const int nbelow = (n16 >> 4) - (c0 >> 4) - 4;
    for (int g = 0; g * 4 * RT < nbelow; ++g) {
      rr2_update_body<RT, false>(F, b, c0, n16, g, true, reinterpret_cast<double (*)[64 * RR2_PLD]>(&DtMs[0][0]), Mi);
      __syncthreads();
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Live column counts from the pivot pattern of G's factor: nlive = 1 + the last column with a pivot (every column behind it
// is zero), nRl = the part of it inside the free block.  Also presets the pivot flags of T (the dead tail counts as skipped).
// meta per instance: [skip (rv) | skipT (rv) | nlive | nRl].  grid = batch, 256 threads.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rr2_meta_kernel(int* __restrict__ meta, long long mstride, int rv, int r, int nF, int nR) {
  __shared__ int red[4];
  int* mt = meta + blockIdx.x * mstride;
  int last = -1;
  for (int i = threadIdx.x; i < r; i += blockDim.x) if (mt[i] == 0) last = i;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { const int o = __shfl_xor(last, off, 64); last = o > last ? o : last; }
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = last;
  __syncthreads();
  int all = red[0];
  for (int w = 1; w < (int)(blockDim.x >> 6); ++w) all = red[w] > all ? red[w] : all;
  const int nlive = all + 1;
  int nRl = nlive - nF;
  nRl = nRl < 0 ? 0 : (nRl > nR ? nR : nRl);
  if (threadIdx.x == 0) { mt[2 * rv] = nlive; mt[2 * rv + 1] = nRl; }
  for (int a = threadIdx.x; a < nR; a += blockDim.x) mt[rv + a] = (a >= nRl) ? 1 : 0;
}

// ---------------------------------------------------------------------------------------------------------------
// Dense weighting matrices: Y = W C for the lower-triangular block C(j, c) = L(nF + j, nF + c), c <= j < nR, leading ncol columns;
// W (nR x nR, row-major, position order) is shared by the batch.  Y: [16 ceil(nR / 16)][ldy] doubles per instance, row-major.
// grid = (row tiles of 16, batch), 256 threads: the four waves of a workgroup take the column tiles in turn; per tile one MFMA
// chain over j from the tile's first column on (C is zero above its diagonal), both operands straight from global memory / L2.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rr2_wc_kernel(const double* __restrict__ Wd, const double* __restrict__ ws, long long stride,
                                                     const int* __restrict__ meta, long long mstride, int rv, int nF, int nR,
                                                     double* __restrict__ Y, long long ystride, int ldy) {
  const long long b = blockIdx.y;
  const int ncol = meta[b * mstride + 2 * rv + 1];
  const int nct = (ncol + 15) >> 4;
  const double* Lm = ws + b * stride;
  double* Yb = Y + b * ystride;
  const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, l4 = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), nwave = blockDim.x >> 6;
  const int i0 = 16 * (int)blockIdx.x;
  const int ia = (i0 + l15 < nR) ? i0 + l15 : nR - 1;                       // row of W this lane feeds (clamped: rows past nR are not stored)
  const double* wr = Wd + (long long)ia * nR;
  for (int ct = wave; ct < nct; ct += nwave) {
    const int c0 = 16 * ct, cc = c0 + l15;
    d4 acc = d4{0.0, 0.0, 0.0, 0.0};
    for (int j0 = c0; j0 < nR; j0 += 4) {
      const int j = j0 + l4;
      const bool jok = j < nR;
      const double a = jok ? wr[j] : 0.0;
      const double bq = (jok && cc <= j && cc < ncol) ? Lm[pk_row((size_t)(nF + j)) + nF + cc] : 0.0;
      acc = rr2_mfma(a, bq, acc);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int i = i0 + l4 + 4 * q;
      Yb[(long long)i * ldy + cc] = (i < nR && cc < ncol) ? acc[q] : 0.0;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// T = C' W C for the lower-triangular block C(i, a) = L(nF + i, nF + a), a <= i < nR, W = diag(w) (the cost weights of the free
// components in the permuted order: tabd[3][perm[nF + i]]), leading nRl columns only; a skipped pivot has a zero column in L:
// its row and column of T come out zero and the diagonal entry is set to one.  The rows [nRl, 16 ceil(nRl / 16)) are written
// as zeros (the lock-step Cholesky works on whole tiles).
// One workgroup (1024 threads) per instance: C is streamed ONCE through LDS, 16 rows at a time (coalesced pieces of packed
// rows), and every wave keeps up to RR2_CW_TPW tiles of T in accumulators -- the row index is the contraction index, 4 MFMAs
// per tile and row block, both operands from the block in LDS.  (A first version dealt (tile row, four tile columns) items to
// the waves of several workgroups, each streaming its own rows from L2: 7.5 x the traffic, 1.9 ms per 512 instances.)
// More tiles than 16 RR2_CW_TPW: further passes over C.  LDS: 16 rows of `ldw` doubles (ldw = 16 mod 32) + 16 weights.
// ---------------------------------------------------------------------------------------------------------------
constexpr int RR2_CW_TPW = 5;
__global__ __launch_bounds__(1024) void rr2_cwc_kernel(KParams P, int RPs, const int* __restrict__ perm, double* __restrict__ ws,
                                                         long long stride, long long toff, const int* __restrict__ meta, long long mstride,
                                                         int rv, int nF, int nR, unsigned long long* __restrict__ tmaxbits, int ldw,
                                                         const double* __restrict__ Y, long long ystride, int ldy) {
  // Y != nullptr: dense weighting matrices -- T = C' Y with Y = W C (rr2_wc_kernel, [nR16][ldy] per instance) instead of C' diag(w) C;
  // a second staging buffer for the rows of Y
  extern __shared__ __attribute__((aligned(16))) double rr2_cb[];
  double* cb = rr2_cb;                                   // [16][ldw]
  double* wb = rr2_cb + 16 * ldw;                        // [16]
  double* yb = wb + 16;                                  // [16][ldw] (dense only)
  const double* Yb = Y ? Y + blockIdx.x * ystride : nullptr;
  const long long b = blockIdx.x;
  const int* mt = meta + b * mstride;
  const int ncol = mt[2 * rv + 1];
  const int nt = (ncol + 15) >> 4;
  if (nt == 0) return;
  const double* Lm = ws + b * stride;
  double* T = ws + b * stride + toff;
  const int* skipd = mt + nF;
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int lane = tid & 63, l15 = lane & 15, l4 = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), nwave = nthr >> 6;
  const int ntile = nt * (nt + 1) / 2;
  double tmx = 0.0;
  for (int t0 = 0; t0 < ntile; t0 += nwave * RR2_CW_TPW) {
    d4 acc[RR2_CW_TPW];
    int tA[RR2_CW_TPW], tB[RR2_CW_TPW];
#pragma unroll
    for (int s = 0; s < RR2_CW_TPW; ++s) {
      acc[s] = d4{0.0, 0.0, 0.0, 0.0};
      const int id = t0 + wave + nwave * s;               // wave-uniform
      int A = (int)((sqrtf(8.0f * (float)id + 1.0f) - 1.0f) * 0.5f);
      while ((A + 1) * (A + 2) / 2 <= id) ++A;
      while (A * (A + 1) / 2 > id) --A;
      tA[s] = (id < ntile) ? A : -1;
      tB[s] = id - A * (A + 1) / 2;
    }
    for (int i0 = 0; i0 < nR; i0 += 16) {
      const int cw = (i0 + 16) < 16 * nt ? (i0 + 16) : 16 * nt;      // columns that can be non-zero in this row block
      __syncthreads();                                     // the previous block has been consumed
      for (int e = tid; e < 16 * cw; e += nthr) {
        const int ii = e / cw, cc = e - ii * cw;
        const int i = i0 + ii;
        double v = 0.0;
        if (i < nR && cc <= i && cc < ncol) v = Lm[pk_row((size_t)(nF + i)) + nF + cc];
        cb[ii * ldw + cc] = v;
      }
      if (tid < 16) wb[tid] = Y ? 1.0 : ((i0 + tid < nR) ? P.tabd[3 * RPs + perm[nF + i0 + tid]] : 0.0);
      if (Y)
        for (int e = tid; e < 16 * cw; e += nthr) {
          const int ii = e / cw, cc = e - ii * cw;
          yb[ii * ldw + cc] = (i0 + ii < nR && cc < ncol) ? Yb[(long long)(i0 + ii) * ldy + cc] : 0.0;
        }
      __syncthreads();
#pragma unroll
      for (int s = 0; s < RR2_CW_TPW; ++s) {
        if (tA[s] < 0 || 16 * tA[s] > i0 + 15) continue;   // (wave-uniform) rows of this block have no entry in tile column A
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int k = l4 + 4 * e;
          acc[s] = rr2_mfma(cb[k * ldw + 16 * tA[s] + l15] * wb[k], (Y ? yb : cb)[k * ldw + 16 * tB[s] + l15], acc[s]);
        }
      }
    }
#pragma unroll
    for (int s = 0; s < RR2_CW_TPW; ++s) {
      if (tA[s] < 0) continue;
      const int bcol = 16 * tB[s] + l15;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int ar = 16 * tA[s] + l4 + 4 * q;
        if (bcol <= ar) {
          double v = 0.0;
          if (ar < ncol) v = (ar == bcol && skipd[ar]) ? 1.0 : acc[s][q];
          T[pk_row((size_t)ar) + bcol] = v;
          if (ar == bcol) tmx = fmax(tmx, v);
        }
      }
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) tmx = fmax(tmx, __shfl_xor(tmx, off, 64));
  if (lane == 0 && tmx > 0.0) atomicMax(tmaxbits + 4 * b, (unsigned long long)__double_as_longlong(tmx));   // (four words per instance)
}

// ---------------------------------------------------------------------------------------------------------------
// The rank decision of an exact-data NOMINAL problem, judged from the pivot candidates AFTER the factorisation (round 5).
// The reference decides ranks with a data-scaled tolerance on singular values (hankel_matrix.py:82); here a pivot below
// tol_rel x (largest diagonal entry) counts as zero, and that fixed tolerance has a window that moves with the plant: dependent
// rows leave rounding residues of ~1e-10 at configs[4] and up to 1.4e-8 on a 9-channel plant of the tests, genuine pivots start
// at 1.6e-7 resp. 1.1e-5 (tools/pivot_gap_study.py).  Per instance, from the sorted candidates (relative to the largest):
//   * more accepted pivots than rank H can be -- m (L + n) + n for exact data of an LTI system of order <= n (Willems' lemma; the
//     reference's own N_min rests on it, controller.py:275) -- means noise was taken for a pivot: a new tolerance is set in the
//     middle (geometric mean) of the gap behind the largest m (L + n) + n candidates and the instance is factored again (flag 2);
//   * a decision without a clear margin -- smallest accepted candidate / largest skipped one below `safe` -- is reported, not
//     hidden: flag bit 1, which the solve turns into the status "optimal_inaccurate".
// rec per instance: [flag, -] ints; tol_out[b]: the tolerance of the next pass.  counter: number of instances with flag 2.
// grid = batch, 256 threads, r <= 1024.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rr2_rank_margin_kernel(const double* __restrict__ cand, long long cand_stride, int r, int bound,
                                                              double tol_rel, const double* __restrict__ tol_in, double safe, int allow_redo,
                                                              double* __restrict__ tol_out, int* __restrict__ rec, int* __restrict__ counter,
                                                              double* __restrict__ noise_out) {
  // noise_out[b]: the largest candidate that counts as zero, relative to the largest: the level to which a dependent row is
  // reproduced by the accepted ones at best, sqrt of it in the rows themselves (the feasibility test of the solve scales with it)
  __shared__ double v[1024];
  const long long b = blockIdx.x;
  const int tid = threadIdx.x;
  const double* c = cand + b * cand_stride;
  double mx = 0.0;
  for (int i = tid; i < 1024; i += 256) { const double x = (i < r) ? c[i] : 0.0; v[i] = x > 0.0 ? x : 0.0; mx = fmax(mx, v[i]); }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) mx = fmax(mx, __shfl_xor(mx, off, 64));
  __shared__ double red[4];
  if ((tid & 63) == 0) red[tid >> 6] = mx;
  __syncthreads();
  mx = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
  const double inv = mx > 0.0 ? 1.0 / mx : 0.0;
  for (int i = tid; i < 1024; i += 256) v[i] *= inv;
  __syncthreads();
  // the common case needs no order statistics: count the accepted candidates, the smallest of them and the largest skipped one
  const double tol = tol_in ? tol_in[b] : tol_rel;
  int cnt = 0;
  double amin = 2.0, smax = 0.0;
  for (int i = tid; i < 1024; i += 256) {
    const double x = v[i];
    if (x > tol) { ++cnt; amin = fmin(amin, x); } else smax = fmax(smax, x);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    cnt += __shfl_xor(cnt, off, 64); amin = fmin(amin, __shfl_xor(amin, off, 64)); smax = fmax(smax, __shfl_xor(smax, off, 64));
  }
  __shared__ double ra[4], rs[4];
  __shared__ int rc[4];
  if ((tid & 63) == 0) { rc[tid >> 6] = cnt; ra[tid >> 6] = amin; rs[tid >> 6] = smax; }
  __syncthreads();
  const int K = rc[0] + rc[1] + rc[2] + rc[3];
  amin = fmin(fmin(ra[0], ra[1]), fmin(ra[2], ra[3]));
  smax = fmax(fmax(rs[0], rs[1]), fmax(rs[2], rs[3]));
  const bool over = K > bound && bound >= 1;                                // (workgroup-uniform) more pivots than rank H can have
  if (over) {
    for (int k = 2; k <= 1024; k <<= 1)                                     // bitonic sort, descending: v[bound - 1], v[bound] are needed
      for (int j = k >> 1; j > 0; j >>= 1) {
        for (int i = tid; i < 1024; i += 256) {
          const int l = i ^ j;
          if (l > i) {
            const bool up = (i & k) == 0;
            const double a = v[i], bq = v[l];
            if (up ? (a < bq) : (a > bq)) { v[i] = bq; v[l] = a; }
          }
        }
        __syncthreads();
      }
  }
  if (tid == 0) {
    int flag = 0;
    double tnew = tol, hi, lo;
    if (over) {
      hi = v[bound - 1]; lo = v[bound];
      if (allow_redo) { tnew = sqrt(hi * fmax(lo, 1e-300)); flag = 2; } else flag = 1;
    } else {
      hi = K > 0 ? amin : 1.0; lo = K < r ? smax : 0.0;
    }
    if (lo > 0.0 && hi < safe * lo) flag |= 1;
    tol_out[b] = tnew;
    if (noise_out) noise_out[b] = lo;
    rec[2 * b] = flag; rec[2 * b + 1] = K;
    if (flag & 2) atomicAdd(counter, 1);
  }
}

}  // namespace ddmpc
