// ddmpc_kernels.hpp -- shared device-side types of the batched Data-Driven MPC QP engine for gfx950 (CDNA4):
// kernel parameters (KParams), component kinds, compile-time loops and small wave-level helpers.  The cold-solve
// kernel itself lives in ddmpc_cold2.hpp, the other kernels in ddmpc_aux_kernels.hpp.
//
// One workgroup solves one controller instance end to end and keeps the whole problem on chip:
//
//   trajectory (u_d,y_d) --coalesced--> LDS  (channel-interleaved "xflat")
//   G = H H'   : never materialises H.  H[rho][i] = xflat[i*nch + rho], so
//                * structured mode (nch == 4): G(k,l) depends on (k-l, l) only through a
//                  sliding-window recurrence; lag blocks C_d = G(d,0) by v_mfma_f64_4x4x4, then every
//                  wave walks its own tile diagonals in registers with 2 MFMAs per tile (no barriers);
//                * dense mode: fp64 MFMA (v_mfma_f64_16x16x4_f64) over the implicit operand.
//   K = G + lam*D, rhs t: diagonal / extra-column fix-up in registers (dense weights: lam*W^-1 from L2)
//   K = L L'   : blocked Cholesky on register tiles (MFMA accumulators), 16-wide panels (ddmpc_cold2.hpp)
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
  int gram_dense;   // 1: dense MFMA Gram, 0: structured (in the kernel for nch == 4, else from `gpre`)
  int xs_len;       // doubles reserved for xflat in LDS
  double lam;       // lamb_alpha * eps_max (0 for nominal)
  double lamb_sigma;
  double bound;     // c * eps_max
  double sig_scale; // -lam / lamb_sigma: sigma_hat = sig_scale * beta on the boxed components (host-computed: a uniform
  double box_cost;  // lamb_sigma * bound^2     fp64 value formed in the kernel would sit in a VGPR pair for its whole run)
  const double* tabd;
  const int* tabi;
  int refine;           // iterative refinement with exact Hankel products: 0 off, 1 auto (residual test), 2 always
  int refine_max;       // cap on refinement passes per factorisation
  double refine_res;    // auto: refine when |t - (H (H' beta) + lam D beta)|_inf / |t|_inf (exact Hankel products) exceeds this
  int res_fits;         // auto: 1 when alpha and a 16-row chunk of the residual check fit the LDS scratch region (else: refine)
  int epoch;            // launch counter of the handle (AUTO refinement: flags / last-flagged stamp carry it, nothing is cleared)
  int dense_w;          // 1: dense weighting matrices -> lam * W^-1 is the full [RP][RP] matrix `dmat`
  const double* dmat;   //    (shared by the batch, zero outside the weighted components), tabd D0 = D1 = 0
  // Structured Gram for channel counts other than 4: G = H H' of every instance comes from ddmpc_gram_tiles_kernel
  // (ddmpc_aux_kernels.hpp) in the accumulator layout of the cold-solve kernel -- tile (I, J), I >= J, at
  // gpre + b * gpre_stride + (I (I + 1) / 2 + J) * 256, double 4 * lane + j of it = register j of that lane -- and the
  // Gram phase of the kernel is one 32-byte load per lane and tile.  nullptr: the kernel forms G itself.
  const double* gpre;
  long long gpre_stride;
  int stage_xs;         // 1: the trajectory is staged in LDS (default).  0: it is longer than LDS holds (round 5): G comes from a
                        // streaming launch through `gpre`, nothing in the kernel touches xs, and what needs the trajectory -- the
                        // exact-Hankel residual check of AUTO refinement, the refining variant -- is not available (host: launch_cold)
};

template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (N > 0) {
    static_for<N - 1>(f);
    f(std::integral_constant<int, N - 1>{});
  }
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

}  // namespace ddmpc
