// ddmpc_rr3.hpp -- ROBUST controllers beyond the register-resident kernels ((m+p)(L+n) > 271) on the phase kernels (round 5).
//
// Until round 4 this size ran on ddmpc_large_solve_kernel (ddmpc_workspace_kernels.hpp): ONE 512-thread workgroup per instance, every
// phase inside one 128-VGPR allocation with 524 B of scratch per lane, the slack box re-factoring a |B| x |B| Schur block per
// active-set iteration.  Here the same reduced system (controller.py:506-547, 631-677, 679-722)
//
//     (G + lam D(act)) beta = t(act),   G = H H',   z = t - lam D beta,   primal-dual active set on the slack box
//
// runs on the lock-step pipeline the NOMINAL controllers got in round 4 (ddmpc_rr2.hpp):
//
//   per data set  rr2_gram_kernel (components in the order "slack-box components LAST") -> rr3_shift_kernel (+ lam D0 on the
//                 diagonal) -> the lock-step Cholesky of ddmpc_rr2.hpp (rr2_chol_panel_kernel / rr2_chol_update_kernel per 64
//                 columns; pivot tolerance 0: the matrix is positive definite) -- K0 = L L', the factor of the EMPTY active set,
//                 with Minv of every 64 x 64 diagonal block next to it
//   per solve     rr3_solve_kernel: y = L^-1 t (64 rows per step, rr2_trsv_fwd); the active-set iterations on the TRAILING block
//                 only; beta = L^-T y'.  The slack box switches D_ii (and the target) of k components, all of them in the
//                 trailing block T = rows n0.. (n0 = the multiple of 64 below the first boxed component):
//                     K(act) = K0 - E diag(d) E',   d_s = lam (D0_s - D1_s) > 0,     t(act) = t0 + bound E sgn
//                     beta = L^-T ( y' + W (diag(1/d) - W'W)^-1 W' y' ),   W = L^-1 E,   y' = y + bound W sgn          (Woodbury)
//                 and W = [0; L_TT^-1 E_T] because L is lower triangular and the unit vectors vanish above row n0: every
//                 iteration works on the 240 .. 300 trailing rows -- W for ALL k columns in ONE pass over L_TT on the matrix pipe
//                 (rr3_w_forward; W lives in global memory / L2, row-major), W'W on the matrix pipe, a k x k Cholesky by one
//                 wave -- and the full factor is streamed twice per solve (forward, backward) whatever the number of
//                 iterations.  Always relative to the factor of the empty set: nothing is ever re-factored.
//                 (First version: 8 columns per VALU pass over L_TT with W in LDS -- 178 us of a 700 us solve went into three
//                 such passes per iteration, and the 126 KB of LDS left one workgroup per CU.)
//                 rr2_hankel_mfma_kernel (H (H' beta), exact products) -> rr3_refine_kernel: one pass of iterative refinement
//                 on the system of the FINAL active set, solved with the same factor + Woodbury data, then the output stage
//                 (same formulas as ddmpc_cold_solve_kernel2).
//   more than RR3_KMAX = 64 switched components (configs[4]'s size: 6 .. 36 in 96 instances), or a pivot that fails: the instance
//   is marked and ddmpc_large_solve_kernel finishes it (host: launch_rr3_solve).
#pragma once
#include "ddmpc_rr2_solve.hpp"

namespace ddmpc {

constexpr int RR3_KMAX = 64;        // columns of W at most (k x k system by one wave: lane = row)

struct Rr3 {
  const double* ws; long long stride;                 // factor of K0 (packed, rows on 128-byte boundaries), instance b at ws + b stride
  const double* m64; long long m64_stride;            // Minv of its 64 x 64 diagonal blocks
  const int* skip; long long s_stride;                // pivot flags of the factorisation
  const unsigned long long* dd;                       // per instance [max diag | - | live chunks | -]
  const int* perm;                                    // position -> component (rv entries), then component -> position
  int rv, r, nA, n0;                                  // nA: first boxed position (r without the box); n0 = nA rounded DOWN to 64
  double* V; long long vstride; int VL;               // vectors kept between the launches: R3_X | R3_BETA | R3_T0 (VL each)
  double* ZP;                                         // Hankel partial sums [batch][RR2_NG][VL]
  double* Wg; long long wstride; int ldw;             // per instance: W (ldw rows x RR3_KMAX, row-major), then the factor of the k x k system (RR3_KMAX x (RR3_KMAX + 1))
  int* kq; long long kstride;                         // per instance ints: [0] k  [1] state (0 ok, 4 failed, 5 -> fallback)  [2] iters  [3] -
                                                      //   [4 .. 4 + RR3_KMAX) the switched positions  [4 + RR3_KMAX ..) the active set (rv)
};
enum : int { R3_X = 0 /* beta, component order: what the Hankel kernel reads */, R3_BETA /* position order */, R3_T0, R3_NV };

// ---------------------------------------------------------------------------------------------------------------
// + lam D0 on the diagonal of the permuted Gram matrix (the Gram kernel wrote G itself); dense weighting matrices
// (controller.py:708-710): + lam W^-1 on every pair of components (P.dmat, component order, shared by the batch -- L2).
// grid = batch, 256 threads.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rr3_shift_kernel(KParams P, int RPs, const int* __restrict__ perm, double* __restrict__ ws, long long stride) {
  double* G = ws + blockIdx.x * stride;
  for (int i = threadIdx.x; i < P.r; i += blockDim.x) G[pk_row((size_t)i) + i] += P.lam * P.tabd[0 * RPs + perm[i]];
  if (P.dense_w) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nwave = blockDim.x >> 6;
    for (int i = wave; i < P.r; i += nwave) {                               // a wave per row: its lanes on consecutive columns
      const double* dr = P.dmat + (long long)perm[i] * RPs;
      double* Gi = G + pk_row((size_t)i);
      for (int j = lane; j <= i; j += 64) Gi[j] += P.lam * dr[perm[j]];
    }
  }
}

// Dense weighting matrices: out[rho] = (W^-1 x)[rho] for x in COMPONENT order (LDS), a wave per row with its lanes on consecutive
// columns of P.dmat (coalesced, from L2: the matrix is shared by the batch).  All threads of the workgroup; no barrier inside.
__device__ __forceinline__ void rr3_dense_times(const KParams& P, int RPs, const double* xc, double* out) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nwave = blockDim.x >> 6, r = P.r;
  for (int rho = wave; rho < r; rho += nwave) {
    const double* dr = P.dmat + (long long)rho * RPs;
    double s0 = 0.0, s1 = 0.0;
    int c = lane;
    for (; c + 64 < r; c += 128) { s0 = fma(dr[c], xc[c], s0); s1 = fma(dr[c + 64], xc[c + 64], s1); }
    if (c < r) s0 = fma(dr[c], xc[c], s0);
    double sacc = s0 + s1;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sacc += __shfl_xor(sacc, off, 64);
    if (lane == 0) out[rho] = sacc;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// y = L^-1 rhs on the rows [64 b_lo, n) of a packed factor for NR right-hand sides at once, WITHOUT the columns in front of
// 64 b_lo (the callers' right-hand sides vanish there, or already carry those terms): rr2_trsv_fwd with a block range.
// rhs(q, i): value of right-hand side q at row i; yq(q): LDS vector of solution q, indexed by the absolute row; tmp: NR x 64.
// ---------------------------------------------------------------------------------------------------------------
template <int NR, class RhsF, class YF>
__device__ __forceinline__ void rr3_trsv_fwd(const double* __restrict__ Lm, const double* __restrict__ m64, int n, unsigned long long live,
                                             int b_lo, RhsF&& rhs, YF&& yq, double* tmp) {
  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));     // opaque per call: what is derived from it is not kept alive across the kernel's phases
  const int row = tid >> 3, part = tid & 3, half = (tid >> 2) & 1, sub = tid & 7;
  const int nb = (n + 63) >> 6;
  const unsigned long long front = (b_lo > 0) ? ((1ull << (4 * b_lo)) - 1ull) : 0ull;
  for (int b = b_lo; b < nb; ++b) {
    const int k0 = 64 * b, i = k0 + row;
    const bool rok = i < n;
    const double* Li = Lm + pk_row((size_t)(rok ? i : n - 1)) + 4 * part;
    const double* Mr = m64 + (size_t)b * 4096 + row * 64 + 8 * sub;
    const d4 m0 = *reinterpret_cast<const d4*>(Mr), m1 = *reinterpret_cast<const d4*>(Mr + 4);
    unsigned long long lv = live & ((1ull << (4 * b)) - 1ull) & ~front;
    double s[NR];
#pragma unroll
    for (int q = 0; q < NR; ++q) s[q] = 0.0;
    constexpr int NU = 4;                                       // 32-byte pieces in flight per thread
    while (lv != 0ull) {
      d4 v[NU];
      int jj[NU];
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        int jc[2];
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
          const bool ok = lv != 0ull;
          jc[hh] = ok ? __builtin_ctzll(lv) : -1;
          if (ok) lv &= lv - 1ull;
        }
        const int mine = half ? jc[1] : jc[0];
        jj[u] = mine;
        v[u] = *reinterpret_cast<const d4*>(Li + 16 * (mine >= 0 ? mine : 0));
      }
#pragma unroll
      for (int u = 0; u < NU; ++u)
        if (jj[u] >= 0) {
#pragma unroll
          for (int q = 0; q < NR; ++q) {
            const double* yy = yq(q) + 16 * jj[u] + 4 * part;
            s[q] += (v[u][0] * yy[0] + v[u][1] * yy[1]) + (v[u][2] * yy[2] + v[u][3] * yy[3]);
          }
        }
    }
#pragma unroll
    for (int q = 0; q < NR; ++q) {
      double sq = s[q];
      sq += __shfl_xor(sq, 1, 64);
      sq += __shfl_xor(sq, 2, 64);
      sq += __shfl_xor(sq, 4, 64);
      if (sub == 0) tmp[q * 64 + row] = rok ? rhs(q, i) - sq : 0.0;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < NR; ++q) {
      double t = 0.0;
#pragma unroll
      for (int e = 0; e < 4; ++e) t += m0[e] * tmp[q * 64 + 8 * sub + e] + m1[e] * tmp[q * 64 + 8 * sub + 4 + e];
      t += __shfl_xor(t, 1, 64);
      t += __shfl_xor(t, 2, 64);
      t += __shfl_xor(t, 4, 64);
      if (sub == 0) yq(q)[i] = rok ? t : 0.0;
    }
    __syncthreads();
  }
}

// x = L^-T yv on the block rows b_hi-1 .. b_lo of the leading n x n block; x of the rows behind 64 b_hi must already be in place
// (the terms of the columns below).  rr2_trsv_bwd with a block range; red: 32 x 64 doubles, tmp: 64.
__device__ __forceinline__ void rr3_trsv_bwd(const double* __restrict__ Lm, const double* __restrict__ m64, int n, unsigned long long live,
                                             int b_hi, int b_lo, const double* yv, double* x, double* red, double* tmp) {
  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));     // (opaque per call, see rr3_trsv_fwd)
  const int cq = tid & 15, rg = tid >> 4;
  const int c = tid & 63, pr = tid >> 6;
  for (int b = b_hi - 1; b >= b_lo; --b) {
    const int k0 = 64 * b;
    const double* Mb = m64 + (size_t)b * 4096;
    double mc[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) mc[q] = Mb[(8 * pr + q) * 64 + c];
    d4 acc = d4{0.0, 0.0, 0.0, 0.0};
    constexpr int NU = 4;                                       // rows in flight per thread (two workgroups per CU: 128 VGPRs)
    for (int i0 = k0 + 64 + rg; i0 < n; i0 += 32 * NU) {
      d4 v[NU];
      double xi[NU];
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        const int i = i0 + 32 * u;
        const bool ok = i < n && ((live >> (i >> 4)) & 1ull) != 0ull;
        v[u] = *reinterpret_cast<const d4*>(Lm + pk_row((size_t)(i < n ? i : n - 1)) + k0 + 4 * cq);
        xi[u] = ok ? x[i] : 0.0;
      }
#pragma unroll
      for (int u = 0; u < NU; ++u) { acc[0] += v[u][0] * xi[u]; acc[1] += v[u][1] * xi[u]; acc[2] += v[u][2] * xi[u]; acc[3] += v[u][3] * xi[u]; }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) red[rg * 64 + 4 * cq + e] = acc[e];
    __syncthreads();
    if (tid < 64) {
      double s = 0.0;
#pragma unroll
      for (int g = 0; g < 32; ++g) s += red[g * 64 + tid];
      tmp[tid] = (k0 + tid < n) ? yv[k0 + tid] - s : 0.0;
    }
    __syncthreads();
    double t = 0.0;
#pragma unroll
    for (int q = 0; q < 8; ++q) t += mc[q] * tmp[8 * pr + q];
    red[pr * 64 + c] = t;
    __syncthreads();
    if (tid < 64) {
      double s = 0.0;
#pragma unroll
      for (int g = 0; g < 8; ++g) s += red[g * 64 + tid];
      x[k0 + tid] = (k0 + tid < n) ? s : 0.0;
    }
    __syncthreads();
  }
}

// LDS carve-up of the two one-workgroup kernels (doubles).  `big` is one region with three tenants that never overlap in time:
// the 64 x 64 tile exchange of rr3_w_forward, the k x k system (RR3_KMAX x (RR3_KMAX + 1)), the reduction buffer of the
// backward substitution (32 x 64).
struct Rr3Lds {
  int VL, tv, yv, y2, bv, act, big, tmp, hv, list, total;
  __host__ __device__ static Rr3Lds make(int r) {
    Rr3Lds L;
    L.VL = (r + 63) & ~63;
    L.tv = 0; L.yv = L.tv + L.VL; L.y2 = L.yv + L.VL; L.bv = L.y2 + L.VL;
    L.act = L.bv + L.VL;                                  // ints, VL of them
    L.big = L.act + L.VL / 2;
    L.tmp = L.big + RR3_KMAX * (RR3_KMAX + 1);            // 64 (+ 8 x 64 partial sums of W'y)
    L.hv = L.tmp + 64 + 8 * 64;                           // 2 x RR3_KMAX: h, ev
    L.list = L.hv + 2 * RR3_KMAX;                         // RR3_KMAX ints + a few flags
    L.total = (L.list + RR3_KMAX / 2 + 8 + 1) & ~1;
    return L;
  }
};
static_assert(RR3_KMAX * (RR3_KMAX + 1) >= 64 * 64 && RR3_KMAX * (RR3_KMAX + 1) >= 32 * 64, "the three tenants of Rr3Lds::big");

// ---------------------------------------------------------------------------------------------------------------
// W = L_TT^-1 E_T for all k <= 64 columns in ONE pass over the trailing block of the factor, on the matrix pipe.
// Per 64-row block b (rows i0 = 64 b ..): P = E_b - L(b, <b) W(<b) as 16 x 16 accumulator tiles -- wave (rt = wave % 4, ch =
// wave / 4) owns row tile rt and the column tiles ch, ch + 2; per 16-column chunk of L in front of the block one 32-byte load
// of its packed row per lane (A operand: the four entries feed the four contraction steps, the B operand follows the same
// permutation) and the rows of W already finished, from global memory / L2 (B operand: 16 lanes = 128 consecutive bytes) --
// then the block's Minv (64 x 64, lower triangular) times P through LDS, rows of W stored.  Two workgroup barriers per block.
// Wg: row-major [row - n0][RR3_KMAX]; list[j]: position of column j's unit vector; Pb: 64 x 64 doubles of LDS.
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void rr3_w_forward(const double* __restrict__ Lm, const double* __restrict__ m64, int n, unsigned long long live,
                                              int b_lo, int n0, const int* list, int k, double* Wg, double* Pb) {
  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));     // (opaque per call, see rr3_trsv_fwd)
  const int lane = tid & 63, l15 = lane & 15, l4 = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int rt = wave & 3, ch = wave >> 2;
  const int nb = (n + 63) >> 6;
  const int nct = (k + 15) >> 4;
  const unsigned long long front = (b_lo > 0) ? ((1ull << (4 * b_lo)) - 1ull) : 0ull;
  int lcol[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) { const int col = 16 * (ch + 2 * t) + l15; lcol[t] = (col < k) ? list[col] : -1; }
  for (int b = b_lo; b < nb; ++b) {
    const int i0 = 64 * b;
    const int irow = i0 + 16 * rt + l15;                                     // A operand: this lane's row of the factor
    const double* Li = Lm + pk_row((size_t)(irow < n ? irow : n - 1)) + 4 * l4;
    const double az = (irow < n) ? 1.0 : 0.0;
    d4 acc[2];
    acc[0] = d4{0.0, 0.0, 0.0, 0.0}; acc[1] = d4{0.0, 0.0, 0.0, 0.0};
    unsigned long long lv = live & ((1ull << (4 * b)) - 1ull) & ~front;
    while (lv != 0ull) {
      const int jc = __builtin_ctzll(lv);
      lv &= lv - 1ull;
      const d4 la = *reinterpret_cast<const d4*>(Li + 16 * jc);
      const double* wr = Wg + (size_t)(16 * jc - n0 + 4 * l4) * RR3_KMAX + l15;   // rows 16 jc + 4 l4 + e of W
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        if (ch + 2 * t < nct) {                                              // (wave-uniform)
          double wv[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) wv[e] = wr[(size_t)e * RR3_KMAX + 16 * (ch + 2 * t)];
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[t] = rr2_mfma(la[e] * az, wv[e], acc[t]);
        }
      }
    }
    // P = E - acc into LDS: Pb[row (64)][col (64)], register q of lane (l4, l15) = entry [16 rt + l4 + 4 q][16 ct + l15]
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      if (ch + 2 * t < nct) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int rloc = 16 * rt + l4 + 4 * q;
          Pb[rloc * 65 + 16 * (ch + 2 * t) + l15] = ((lcol[t] == i0 + rloc) ? 1.0 : 0.0) - acc[t][q];
        }
      }
    }
    __syncthreads();
    // X = Minv_b P: tile (rt, ct) = sum_{u <= rt} Minv(rt, u) P(u, ct)
    const double* Mb = m64 + (size_t)b * 4096 + (size_t)(16 * rt + l15) * 64 + 4 * l4;
    d4 x[2];
    x[0] = d4{0.0, 0.0, 0.0, 0.0}; x[1] = d4{0.0, 0.0, 0.0, 0.0};
    for (int u = 0; u <= rt; ++u) {
      const d4 mv = *reinterpret_cast<const d4*>(Mb + 16 * u);
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        if (ch + 2 * t < nct) {
#pragma unroll
          for (int e = 0; e < 4; ++e) x[t] = rr2_mfma(mv[e], Pb[(16 * u + 4 * l4 + e) * 65 + 16 * (ch + 2 * t) + l15], x[t]);
        }
      }
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      if (ch + 2 * t < nct) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int i = i0 + 16 * rt + l4 + 4 * q;
          Wg[(size_t)(i - n0) * RR3_KMAX + 16 * (ch + 2 * t) + l15] = (i < n) ? x[t][q] : 0.0;
        }
      }
    }
    __syncthreads();                                                         // the rows of W of this block are visible (workgroup scope)
  }
}

// Sm = -W'W (lower triangle, k x k, row stride RR3_KMAX + 1) on the matrix pipe: the row index of W is the contraction index,
// both operands 128-byte pieces of rows of W (global memory / L2).  Tiles dealt to the eight waves.
__device__ __forceinline__ void rr3_w_gram(const double* Wg, int nrows, int k, double* Sm) {
  constexpr int LD = RR3_KMAX + 1;
  const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, l4 = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), nwave = (int)(blockDim.x >> 6);
  const int nct = (k + 15) >> 4;
  int idx = 0;
  for (int a = 0; a < nct; ++a)
    for (int c = 0; c <= a; ++c, ++idx) {
      if (idx % nwave != wave) continue;                                     // (wave-uniform)
      d4 acc0 = d4{0.0, 0.0, 0.0, 0.0}, acc1 = d4{0.0, 0.0, 0.0, 0.0};
      const double* pa = Wg + (size_t)l4 * RR3_KMAX + 16 * a + l15;
      const double* pc = Wg + (size_t)l4 * RR3_KMAX + 16 * c + l15;
      int i = 0;
      for (; i + 8 <= nrows; i += 8) {
        acc0 = rr2_mfma(pa[(size_t)i * RR3_KMAX], pc[(size_t)i * RR3_KMAX], acc0);
        acc1 = rr2_mfma(pa[(size_t)(i + 4) * RR3_KMAX], pc[(size_t)(i + 4) * RR3_KMAX], acc1);
      }
      for (; i < nrows; i += 4) {
        const bool ok = i + l4 < nrows;
        const double va = ok ? pa[(size_t)i * RR3_KMAX] : 0.0, vc = ok ? pc[(size_t)i * RR3_KMAX] : 0.0;
        acc0 = rr2_mfma(va, vc, acc0);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int x = 16 * a + l4 + 4 * q, y = 16 * c + l15;                 // D[x'][y'] = sum_i W[i][16 a + x'] W[i][16 c + y']
        if (x < k && y <= x) Sm[x * LD + y] = -(acc0[q] + acc1[q]);
      }
    }
}

// ---- the Woodbury pieces shared by the solve and the refinement kernel ------------------------------------------------
// k x k system  Sm = diag(1/d) - W'W  in LDS (row-major, row stride RR3_KMAX + 1), its Cholesky factor in place by wave 0
// (lane = row; reciprocal pivots on the diagonal).  Returns false on a non-positive pivot (every thread gets the same answer
// through flagw, an LDS word).
__device__ __forceinline__ void rr3_small_cholesky(double* Sm, int k, int* flagw) {
  constexpr int LD = RR3_KMAX + 1;
  const int tid = threadIdx.x;
  if (tid < 64) {
    int x = tid;
    asm volatile("" : "+v"(x));     // (opaque: its LDS row address is formed here, not kept alive across the kernel)
    bool ok = true;
    for (int j = 0; j < k; ++j) {
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      double d = Sm[j * LD + j];
      for (int q = 0; q < j; ++q) d = fma(-Sm[j * LD + q], Sm[j * LD + q], d);
      ok = ok && (d > 0.0);
      const double il = 1.0 / sqrt(d);
      double v = 0.0;
      if (x > j && x < k) {
        v = Sm[x * LD + j];
        for (int q = 0; q < j; ++q) v = fma(-Sm[x * LD + q], Sm[j * LD + q], v);
        v *= il;
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      if (x > j && x < k) Sm[x * LD + j] = v;
      if (x == j) Sm[j * LD + j] = il;
    }
    if (tid == 0) *flagw = ok ? 0 : 1;
  }
}
// ev = Sm^-1 hv by wave 0 (lane = row; column-oriented: the unknown just found is broadcast and every later row takes its
// term), result in ev (LDS).  k <= RR3_KMAX <= 64.
__device__ __forceinline__ void rr3_small_solve(const double* Sm, int k, const double* hv, double* ev) {
  constexpr int LD = RR3_KMAX + 1;
  if (threadIdx.x < 64) {
    int x = threadIdx.x;
    asm volatile("" : "+v"(x));
    double h = (x < k) ? hv[x] : 0.0;
    const double il = (x < k) ? Sm[x * LD + x] : 0.0;
    for (int i = 0; i < k; ++i) {                               // L w = h
      const double e = __shfl(h * il, i, 64);
      if (x == i) h = e;
      else if (x > i && x < k) h = fma(-Sm[x * LD + i], e, h);
    }
    for (int i = k - 1; i >= 0; --i) {                          // L' ev = w
      const double e = __shfl(h * il, i, 64);
      if (x == i) h = e;
      else if (x < i) h = fma(-Sm[i * LD + x], e, h);
    }
    if (x < k) ev[x] = h;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// The solve on the kept factor.  grid = batch, RR2_TS threads, dynamic LDS (Rr3Lds + W).  refine != 0: beta and everything the
// refinement launch needs go to global memory and the outputs are left to rr3_refine_kernel; else the outputs are written here.
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void rr3_outputs(const KParams& P, int RPs, const Rr3& S, long long b, const int* perm, const double* tv, const double* bv,
                                            const int* act, int st, int iter, double* red, double* __restrict__ u_opt, double* __restrict__ cost,
                                            int* __restrict__ status, int* __restrict__ iters, double* __restrict__ beta_ws,
                                            signed char* __restrict__ act_ws, double* wa, double* wb) {
  // wa, wb: two LDS vectors of r doubles that are free by now (dense weighting matrices: beta in component order, W^-1 beta)
  const int tid = threadIdx.x, nthr = blockDim.x, r = P.r;
  const int n = P.npu / P.m;
  double part = 0.0, bad = 0.0;
  double* uo = u_opt + b * (long long)((P.Ln - n) * P.m);
  if (st == 0 && P.dense_w) {                                               // (kernel-uniform)
    __syncthreads();
    for (int i = tid; i < r; i += nthr) wa[perm[i]] = bv[i];
    __syncthreads();
    rr3_dense_times(P, RPs, wa, wb);
    __syncthreads();
  }
  if (st == 0) {
    for (int i = tid; i < r; i += nthr) {
      const int rho = perm[i];
      const int s_act = act[i];
      const double bb = bv[i];
      const double D = s_act ? P.tabd[1 * RPs + rho] : P.tabd[0 * RPs + rho];
      const double t = tv[i] + s_act * P.bound;
      const double z = t - P.lam * (D * bb + (P.dense_w ? wb[rho] : 0.0));
      const double wq = P.tabd[3 * RPs + rho];
      const double tb = P.tabd[2 * RPs + rho];
      const int oidx = P.tabi[2 * RPs + rho];
      const int kind = P.tabi[0 * RPs + rho];
      if (!(fabs(bb) < 1e300)) bad = 1.0;
      double contrib = P.lam * bb * z;
      if (P.dense_w && (kind == K_UFREE || kind == K_YFREE || kind == K_WPRED)) {
        // (z - t)' W (z - t) summed over the weighted components equals -lam beta' (z - t); a sigma held at its bound adds
        // lamb_sigma bound^2 (as in ddmpc_large_solve_kernel and the register-resident kernels)
        contrib -= P.lam * bb * (z - t);
        if (s_act != 0) contrib += P.box_cost;
      } else
      if (kind == K_UFREE || kind == K_YFREE) { const double dlt = z - tb; contrib += wq * dlt * dlt; }
      else if (kind == K_WINT) { const double sg = z - tv[i]; contrib += P.lamb_sigma * sg * sg; }
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

template <bool REFINE_PASS>
__global__ __launch_bounds__(RR2_TS, 4) void rr3_solve_kernel(Rr3 S, KParams P, int RPs, const double* __restrict__ u_past,
                                                           const double* __restrict__ y_past, double* __restrict__ u_opt,
                                                           double* __restrict__ cost, int* __restrict__ status, int* __restrict__ iters,
                                                           double* __restrict__ beta_ws, signed char* __restrict__ act_ws, int defer_outputs) {
  extern __shared__ __attribute__((aligned(16))) double r3_lds[];
  const long long b = blockIdx.x;
  const int tid = threadIdx.x, nthr = blockDim.x, lane = tid & 63;
  const int r = S.r, nA = S.nA, n0 = S.n0, m = P.m, p = P.p;
  const int n = P.npu / m;
  const Rr3Lds LD = Rr3Lds::make(r);
  double* tv = r3_lds + LD.tv; double* yv = r3_lds + LD.yv; double* y2 = r3_lds + LD.y2; double* bv = r3_lds + LD.bv;
  int* act = reinterpret_cast<int*>(r3_lds + LD.act);
  double* big = r3_lds + LD.big;                             // tile exchange of rr3_w_forward | k x k system | reduction buffer of the backward substitution
  double* red = big; double* Sm = big;
  double* tmp = r3_lds + LD.tmp; double* hpart = tmp + 64;
  double* hv = r3_lds + LD.hv; double* ev = hv + RR3_KMAX;
  int* list = reinterpret_cast<int*>(r3_lds + LD.list);
  int* flagw = list + RR3_KMAX;                              // [0] changed  [1] k  [2] small-system failure
  constexpr int LDS_ = RR3_KMAX + 1;
  const int* perm = S.perm;
  const double* G = S.ws + b * S.stride;
  const double* m64 = S.m64 + b * S.m64_stride;
  const unsigned long long live = S.dd[4 * b + 2];
  const int nb = (r + 63) >> 6, b0 = n0 >> 6, nT = r - n0;
  double* Vb = S.V + b * S.vstride;                          // [R3_X | R3_BETA | R3_T0]
  double* Wg = S.Wg + b * S.wstride;                         // W, row-major [row - n0][RR3_KMAX]
  double* Lg = Wg + (size_t)S.ldw * RR3_KMAX;                // factor of the k x k system of the final active set
  int* kq = S.kq + b * S.kstride;
  const double* up = u_past + b * (long long)P.npu;
  const double* yp = y_past + b * (long long)(n * p);
  int st = 0, iter = 0, k = 0;
  const long long t_begin = (long long)__builtin_amdgcn_s_memrealtime();   // diagnostics (kq[.. + rv ..]): when this workgroup ran
#ifdef RR3_PROBE
  long long tp_[16]; int np_ = 0;
#define RR3_STAMP() do { if (np_ < 16) tp_[np_++] = (long long)__builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define RR3_STAMP() do {} while (0)
#endif
  // h = W'v on the trailing block (v: LDS, absolute rows), into hv[0 .. k): column x by thread x of each of the 8 row parts
  auto w_times = [&](const double* v) {
    const int x = tid & 63, part = tid >> 6;
    const int per = (nT + 7) >> 3;
    double h = 0.0;
    if (x < k) {
      const int i1 = (part + 1) * per < nT ? (part + 1) * per : nT;
      for (int i = part * per; i < i1; ++i) h = fma(Wg[(size_t)i * RR3_KMAX + x], v[n0 + i], h);
    }
    hpart[part * 64 + x] = h;
    __syncthreads();
    if (tid < k) {
      double t = 0.0;
#pragma unroll
      for (int g = 0; g < 8; ++g) t += hpart[g * 64 + tid];
      hv[tid] = t;
    }
    __syncthreads();
  };
  // v_T += W e on the trailing block (one thread per row; the row of W is contiguous)
  auto w_apply = [&](const double* src, double* dst, const double* e) {
    for (int i = n0 + tid; i < r; i += nthr) {
      double v = src[i];
      const double* wr = Wg + (size_t)(i - n0) * RR3_KMAX;
      for (int x = 0; x < k; ++x) v = fma(wr[x], e[x], v);
      dst[i] = v;
    }
    __syncthreads();
  };

  if constexpr (!REFINE_PASS) {
    RR3_STAMP();   // 0
    double nbad = 0.0;
    for (int i = tid; i < LD.VL; i += nthr) {
      double t = 0.0;
      if (i < r) {
        const int rho = perm[i];
        const int pidx = P.tabi[1 * RPs + rho];
        t = (pidx >= 0) ? ((pidx < P.npu) ? up[pidx] : yp[pidx - P.npu]) : P.tabd[2 * RPs + rho];
        nbad += S.skip[b * S.s_stride + i] ? 1.0 : 0.0;
      }
      tv[i] = t; act[i] = 0; y2[i] = 0.0; bv[i] = 0.0;
    }
    if (block_sum(nbad, tmp) != 0.0) st = 4;                 // (uniform) a pivot of K0 failed: not positive definite numerically
    __syncthreads();
    if (st == 0) {
      RR3_STAMP();   // 1: staged
      rr3_trsv_fwd<1>(G, m64, r, live, 0, [&](int, int i) { return tv[i]; }, [&](int) { return yv; }, tmp);
      RR3_STAMP();   // 2: forward
      if (!P.convex || nA >= r) {
        rr3_trsv_bwd(G, m64, r, live, nb, 0, yv, bv, red, tmp);
        iter = 1;
      } else {
        for (int i = tid; i < LD.VL; i += nthr) y2[i] = yv[i];
        __syncthreads();
        rr3_trsv_bwd(G, m64, r, live, nb, b0, y2, bv, red, tmp);         // beta on the trailing block: the empty active set
        RR3_STAMP();   // 3: trailing backward
        for (;;) {
          ++iter;
          if (tid == 0) flagw[0] = 0;
          __syncthreads();
          for (int i = nA + tid; i < r; i += nthr) {                     // slack box (sigma[n*p:], controller.py:659): positions nA .. r-1
            const double sh = P.sig_scale * bv[i];                        // (host-computed -lam / lamb_sigma: a kernel argument, not a live register)
            const int ns = (sh > P.bound) ? 1 : (sh < -P.bound) ? -1 : 0;
            if (ns != act[i]) { act[i] = ns; flagw[0] = 1; }
          }
          __syncthreads();
          if (flagw[0] == 0) break;
          if (iter >= P.max_iter) { st = 4; break; }
          // the switched components, ascending (deterministic column order)
          if (tid < 64) {
            int base = 0;
            for (int i0 = nA; i0 < r; i0 += 64) {
              const int i = i0 + lane;
              const int a = (i < r) ? act[i] : 0;
              const unsigned long long mk = __ballot(a != 0);
              const int slot = base + __popcll(mk & ((1ull << lane) - 1ull));
              if (a != 0 && slot < RR3_KMAX) list[slot] = i;
              base += __popcll(mk);
            }
            if (lane == 0) flagw[1] = base;
          }
          __syncthreads();
          k = flagw[1];
          if (k > RR3_KMAX) { st = 5; break; }                            // more columns than W holds: ddmpc_large_solve_kernel finishes it
          rr3_w_forward(G, m64, r, live, b0, n0, list, k, Wg, big);       // W = L_TT^-1 E_T
          if (iter == 1) RR3_STAMP();   // 4: W
          rr3_w_gram(Wg, nT, k, Sm);                                      // Sm = -W'W (big: the tile exchange is done with)
          w_times(yv);                                                    // hv = W'y
          bool dok = true;
          if (tid < k) {
            const int rho = perm[list[tid]];
            const double dd = P.lam * (P.tabd[0 * RPs + rho] - P.tabd[1 * RPs + rho]);
            dok = dd > 1e-300;
            double h = hv[tid];
            for (int yy = 0; yy < k; ++yy) {                              // + (W'W) (bound sgn)
              const double g = -(yy <= tid ? Sm[tid * LDS_ + yy] : Sm[yy * LDS_ + tid]);
              h = fma(g, (double)act[list[yy]] * P.bound, h);
            }
            hv[tid] = h;
          }
          const int anybad = __syncthreads_or(dok ? 0 : 1);
          if (tid < k) {
            const int rho = perm[list[tid]];
            Sm[tid * LDS_ + tid] += 1.0 / (P.lam * (P.tabd[0 * RPs + rho] - P.tabd[1 * RPs + rho]));
          }
          __syncthreads();
          if (iter == 1) RR3_STAMP();   // 5: Gram + h
          rr3_small_cholesky(Sm, k, flagw + 2);
          __syncthreads();
          if (iter == 1) RR3_STAMP();   // 6: k x k Cholesky
          if (anybad || flagw[2] != 0) { st = 5; break; }
          rr3_small_solve(Sm, k, hv, ev);
          if (defer_outputs) {                                            // (the reduction buffer of the backward substitution takes Sm's place)
            for (int e = tid; e < k * LDS_; e += nthr) Lg[e] = Sm[e];
          }
          __syncthreads();
          if (tid < k) ev[tid] += (double)act[list[tid]] * P.bound;       // y' = y + W (bound sgn + cv) on the trailing block
          __syncthreads();
          w_apply(yv, y2, ev);
          if (iter == 1) RR3_STAMP();   // 7: k x k solve + y'
          rr3_trsv_bwd(G, m64, r, live, nb, b0, y2, bv, red, tmp);
          if (iter == 1) RR3_STAMP();   // 8: trailing backward
        }
        RR3_STAMP();     // iterations done
        if (st == 0) rr3_trsv_bwd(G, m64, r, live, b0, 0, y2, bv, red, tmp);   // the rows in front of the trailing block
        RR3_STAMP();     // leading backward
#ifdef RR3_PROBE
        if (tid == 0 && b == 5) {
          printf("rr3 b=%d k=%d iters=%d:", (int)b, k, iter);
          for (int q = 1; q < np_; ++q) printf(" %lld", tp_[q] - tp_[q - 1]);
          printf(" (x10 ns)\n");
        }
#endif
      }
    }
    if (defer_outputs && st == 0) {
      // what the refinement launch needs: beta (both orders), t0, the active set, the switched positions (W and the k x k
      // factor are in global memory already)
      double* vx = Vb + (size_t)R3_X * S.VL;
      double* vbeta = Vb + (size_t)R3_BETA * S.VL;
      double* vt0 = Vb + (size_t)R3_T0 * S.VL;
      for (int i = tid; i < r; i += nthr) { vx[perm[i]] = bv[i]; vbeta[i] = bv[i]; vt0[i] = tv[i]; kq[4 + RR3_KMAX + i] = act[i]; }
      if (tid < k) kq[4 + tid] = list[tid];
    }
    if (tid == 0) {
      kq[0] = (st == 0) ? k : 0; kq[1] = st; kq[2] = iter; kq[3] = k;
      const long long t_end = (long long)__builtin_amdgcn_s_memrealtime();
      kq[4 + RR3_KMAX + S.rv] = (int)(t_begin & 0x7fffffff); kq[4 + RR3_KMAX + S.rv + 1] = (int)(t_end - t_begin);   // (100 MHz ticks)
    }
    if (!defer_outputs || st != 0) {
      __syncthreads();
      rr3_outputs(P, RPs, S, b, perm, tv, bv, act, st == 5 ? 4 : st, iter, tmp, u_opt, cost, status, iters, beta_ws, act_ws, y2, yv);
      __syncthreads();
      if (tid == 0 && st == 5) status[b] = 5;                             // (host: finished by ddmpc_large_solve_kernel)
    }
  } else {
    // ---- one pass of iterative refinement on the system of the final active set, then the outputs.
    //      rho = t(act) - (H (H' beta) + lam D(act) beta), exact Hankel products (rr2_hankel_mfma_kernel left the partial sums in ZP);
    //      delta = L^-T ( y_rho + W Sm^-1 W' y_rho );  beta += delta
    k = kq[0]; st = kq[1]; iter = kq[2];
    if (st != 0) return;                                                  // (uniform) the solve launch wrote the outputs itself
    const double* vbeta = Vb + (size_t)R3_BETA * S.VL;
    const double* vt0 = Vb + (size_t)R3_T0 * S.VL;
    const double* zp = S.ZP + b * RR2_NG * (long long)S.VL;
    if (P.dense_w) {                                                      // (kernel-uniform) yv := W^-1 beta, component order
      for (int i = tid; i < r; i += nthr) y2[perm[i]] = vbeta[i];
      __syncthreads();
      rr3_dense_times(P, RPs, y2, yv);
      __syncthreads();
    }
    for (int i = tid; i < LD.VL; i += nthr) {
      double bb = 0.0, t0 = 0.0, rho_v = 0.0;
      int a = 0;
      if (i < r) {
        const int rho = perm[i];
        bb = vbeta[i]; t0 = vt0[i]; a = kq[4 + RR3_KMAX + i];
        double hz = 0.0;
#pragma unroll
        for (int g = 0; g < RR2_NG; ++g) hz += zp[g * (long long)S.VL + rho];
        const double D = a ? P.tabd[1 * RPs + rho] : P.tabd[0 * RPs + rho];
        rho_v = (t0 + a * P.bound) - hz - P.lam * (D * bb + (P.dense_w ? yv[rho] : 0.0));
      }
      tv[i] = t0; bv[i] = bb; act[i] = a; y2[i] = rho_v;                  // (dense: y2 = beta in component order has been consumed)
    }
    __syncthreads();
    rr3_trsv_fwd<1>(G, m64, r, live, 0, [&](int, int i) { return y2[i]; }, [&](int) { return yv; }, tmp);     // y_rho
    if (k > 0) {
      for (int e = tid; e < k * LDS_; e += nthr) Sm[e] = Lg[e];
      w_times(yv);
      rr3_small_solve(Sm, k, hv, ev);
      __syncthreads();
      w_apply(yv, yv, ev);
    }
    rr3_trsv_bwd(G, m64, r, live, nb, 0, yv, y2, red, tmp);                                                   // delta
    for (int i = tid; i < r; i += nthr) bv[i] += y2[i];
    __syncthreads();
    rr3_outputs(P, RPs, S, b, perm, tv, bv, act, 0, iter, tmp, u_opt, cost, status, iters, beta_ws, act_ws, y2, yv);
  }
}

}  // namespace ddmpc
