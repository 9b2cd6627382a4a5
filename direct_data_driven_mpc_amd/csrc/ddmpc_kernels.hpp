// ddmpc_kernels.hpp -- gfx950 (CDNA4) device code of the batched Data-Driven MPC QP engine.
//
// One workgroup solves one controller instance end to end and keeps the whole
// problem on chip:
//
//   trajectory (u_d,y_d) --coalesced--> LDS  (channel-interleaved "xflat")
//   G = H H'            : fp64 MFMA (v_mfma_f64_16x16x4_f64) straight into the
//                         accumulator registers; the Hankel matrix is never
//                         materialised: H[rho][i] = xflat[i*nch + rho]
//   K = G + lam*D, rhs t: diagonal/extra-row fix-up in registers
//   K = L L'            : right-looking blocked Cholesky, 4-wide panels through
//                         LDS, rank-4 trailing updates by MFMA on the register tiles
//   L y = t             : free -- t rides along as an extra matrix row
//   L' beta = y         : column-oriented back substitution, 4 rows per step
//   slack box           : primal-dual active set around the above (CONVEX only)
//   outputs             : optimal_u, cost, status (+ beta / active set workspace)
//
// Maths: see DESIGN.md section 3 (reduced r x r system).  Reference formulation:
// direct_data_driven_mpc_controller.py:409-445 (variables), :506-677 (constraints),
// :679-722 (cost), :780-808 (extraction).
//
// Internal component order ("rho"): time-major with the nch = m+p channels of one
// time step adjacent, rho = k*nch + ch, k = 0..L+n-1, ch < m -> ubar, else ybar(+sigma).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

namespace ddmpc {

typedef double d4 __attribute__((ext_vector_type(4)));

struct KParams {
  int m, p, n, L, N;
  int nch;          // m + p
  int Ln;           // L + n
  int r;            // nch * Ln           rows of H = [Hu; Hy]
  int rE;           // r rounded up to 4  (dummy identity rows in between)
  int c;            // N - Ln + 1         Hankel columns = len(alpha)
  int robust, convex, tec, weight_diag;
  double q_scalar, r_scalar;
  const double* qdiag;   // device, p*L (diag weights) or null
  const double* rdiag;   // device, m*L or null
  const double* u_s;     // device, [m]
  const double* y_s;     // device, [p]
  double lam;            // lamb_alpha * eps_max (0 for nominal)
  double lamb_sigma;
  double bound;          // c * eps_max
  int max_iter;
  int xs_len;            // doubles reserved for xflat in LDS
};

// component kinds
enum : int { K_UFIX = 0, K_UFREE = 1, K_WINT = 2, K_WTERM = 3, K_WPRED = 4,
             K_YFIX = 5, K_YFREE = 6, K_PAD = 7 };

struct Comp { double D, t, wq; int kind; };   // wq = cost weight (r or q) of the component

template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (N > 0) {
    static_for<N - 1>(f);
    f(std::integral_constant<int, N - 1>{});
  }
}

// Lower-triangular 16x16 tile map: column-major enumeration dealt cyclically to
// the W waves, so that every trailing sub-matrix is balanced over the waves.
template <int NT, int W>
struct TileMap {
  static constexpr int T = NT * (NT + 1) / 2;
  static constexpr int MAXS = (T + W - 1) / W;
  static constexpr int idx(int I, int J) { return J * NT - J * (J - 1) / 2 + (I - J); }
  static constexpr int wave(int I, int J) { return idx(I, J) % W; }
  static constexpr int slot(int I, int J) { return idx(I, J) / W; }
};

// Per-component tables of the reduced system.  `s_act` is the
// signed active-set state of a boxed slack component.
__device__ __forceinline__ Comp comp_of(const KParams& P, int rho, int s_act,
                                        const double* __restrict__ up, const double* __restrict__ yp) {
  Comp cmp;
  if (rho >= P.r) { cmp.D = 0.0; cmp.t = 0.0; cmp.wq = 0.0; cmp.kind = K_PAD; return cmp; }
  const int k = rho / P.nch, ch = rho - k * P.nch;
  const int kp = k - P.n;                         // prediction index (<0: internal window)
  const bool is_int = kp < 0;
  const bool is_term = P.tec && kp >= P.L - P.n;  // controller.py:612-616
  if (ch < P.m) {
    cmp.wq = 0.0;
    if (is_int) { cmp.D = 0.0; cmp.t = up[k * P.m + ch]; cmp.kind = K_UFIX; }        // :577
    else if (is_term) { cmp.D = 0.0; cmp.t = P.u_s[ch]; cmp.kind = K_UFIX; }         // :612,620
    else {
      const double rw = P.weight_diag ? P.rdiag[kp * P.m + ch] : P.r_scalar;        // :709
      cmp.D = 1.0 / rw; cmp.t = P.u_s[ch]; cmp.wq = rw; cmp.kind = K_UFREE;
    }
    return cmp;
  }
  const int cy = ch - P.m;
  const double qw = is_int ? 0.0 : (P.weight_diag ? P.qdiag[kp * P.p + cy] : P.q_scalar);   // :710
  cmp.wq = qw;
  if (!P.robust) {
    if (is_int) { cmp.D = 0.0; cmp.t = yp[k * P.p + cy]; cmp.kind = K_YFIX; }        // :578
    else if (is_term) { cmp.D = 0.0; cmp.t = P.y_s[cy]; cmp.kind = K_YFIX; }         // :615,621
    else { cmp.D = 1.0 / qw; cmp.t = P.y_s[cy]; cmp.kind = K_YFREE; }
    return cmp;
  }
  const double ils = 1.0 / P.lamb_sigma;
  if (is_int) { cmp.D = ils; cmp.t = yp[k * P.p + cy]; cmp.kind = K_WINT; return cmp; }
  if (is_term) {
    cmp.kind = K_WTERM;
    if (s_act == 0) { cmp.D = ils; cmp.t = P.y_s[cy]; }
    else { cmp.D = 0.0; cmp.t = P.y_s[cy] + s_act * P.bound; }
    return cmp;
  }
  cmp.kind = K_WPRED;
  if (s_act == 0) { cmp.D = 1.0 / qw + ils; cmp.t = P.y_s[cy]; }
  else { cmp.D = 1.0 / qw; cmp.t = P.y_s[cy] + s_act * P.bound; }
  return cmp;
}

// By-value select: keeps ternaries over captured variables from turning into
// pointer selects (which would pin the accumulators in scratch memory).
__device__ __forceinline__ double sel4(int k, double a, double b, double c, double d) {
  return (k == 0) ? a : (k == 1) ? b : (k == 2) ? c : d;
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// LDS carve-up (doubles).  Everything lives in ONE dynamic array (16-B aligned).
template <int NT>
struct Lds {
  static constexpr int RP = 16 * NT;
  int xs, PT, LT, LROW, dvec, tvec, beta, invd, yc, red, ints, total;
  __host__ __device__ explicit Lds(int xs_len) {
    int o = 0;
    xs = o;   o += xs_len;
    PT = o;   o += 4 * RP;
    LT = o;   o += 2 * 4 * RP;
    LROW = o; o += 2 * 4 * RP;
    dvec = o; o += RP;
    tvec = o; o += RP;
    beta = o; o += RP;
    invd = o; o += RP;
    yc = o;   o += 8;
    red = o;  o += 32;
    ints = o; o += (RP + 8 + 1) / 2 + 1;   // int act[RP], int flags[8]
    total = (o + 1) & ~1;
  }
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
                                          double* __restrict__ beta_ws, signed char* __restrict__ act_ws) {
  using TM = TileMap<NT, W>;
  constexpr int RP = 16 * NT;
  constexpr int NTHR = 64 * W;
  const Lds<NT> lds(P.xs_len);
  double* xs = sm + lds.xs;
  double* PT = sm + lds.PT;
  double* LT = sm + lds.LT;
  double* LROW = sm + lds.LROW;
  double* dvec = sm + lds.dvec;
  double* tvec = sm + lds.tvec;
  double* beta = sm + lds.beta;
  double* invd = sm + lds.invd;
  double* yc = sm + lds.yc;
  double* red = sm + lds.red;
  int* act = reinterpret_cast<int*>(sm + lds.ints);
  int* flags = act + RP;            // [0] fail, [1] active set changed

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int l15 = lane & 15, l4 = lane >> 4;
  const int r = P.r, rE = P.rE, nch = P.nch;
  const int NS = rE >> 2;           // panel steps
  const int IR = rE >> 4;           // tile row holding the rhs row (row index rE)
  const int rr = rE & 15;

  d4 acc[TM::MAXS];

  for (int i = tid; i < RP; i += NTHR) act[i] = 0;
  if (tid < 8) flags[tid] = 0;

  int iter = 0;
  int status = 0;
  for (;;) {
    ++iter;
    __syncthreads();
    // ---- component tables for the current active set -----------------------
    for (int rho = tid; rho < RP; rho += NTHR) {
      Comp cmp = comp_of(P, rho, act[rho], up, yp);
      dvec[rho] = cmp.D;
      tvec[rho] = cmp.t;
    }
    if (tid == 0) flags[1] = 0;

    // ---- G = H H' by fp64 MFMA over the implicit Hankel operand --------------
    static_for<TM::MAXS>([&](auto S) __attribute__((always_inline)) { acc[S] = d4{0.0, 0.0, 0.0, 0.0}; });
    {
      // Rows >= r of the padded operand read live trajectory data; the garbage they
      // produce lands only in padded rows/cols of G and is cleared in the fix-up below,
      // so the main loop carries no masks.  Only the last partial k-step is masked.
      const int c = P.c;
      const int cfull = c & ~3;
      const double* xp = xs + l4 * nch + l15;
      for (int i0 = 0; i0 < cfull; i0 += 4) {
        double op[NT];
        static_for<NT>([&](auto I) __attribute__((always_inline)) { op[I] = xp[16 * I]; });
        xp += 4 * nch;
        static_for<NT>([&](auto J) __attribute__((always_inline)) {
          static_for<NT>([&](auto I) __attribute__((always_inline)) {
            if constexpr (I >= J && TM::wave(I, J) == WAVE) {
              acc[TM::slot(I, J)] =
                  __builtin_amdgcn_mfma_f64_16x16x4f64(op[I], op[J], acc[TM::slot(I, J)], 0, 0, 0);
            }
          });
        });
      }
      if (cfull < c) {
        const bool kok = (cfull + l4) < c;
        double op[NT];
        static_for<NT>([&](auto I) __attribute__((always_inline)) { const double v = xp[16 * I]; op[I] = kok ? v : 0.0; });
        static_for<NT>([&](auto J) __attribute__((always_inline)) {
          static_for<NT>([&](auto I) __attribute__((always_inline)) {
            if constexpr (I >= J && TM::wave(I, J) == WAVE) {
              acc[TM::slot(I, J)] =
                  __builtin_amdgcn_mfma_f64_16x16x4f64(op[I], op[J], acc[TM::slot(I, J)], 0, 0, 0);
            }
          });
        });
      }
    }
    __syncthreads();   // dvec/tvec visible

    // ---- K = G + lam*D (diagonal), identity on dummy rows, rhs row ---------
    static_for<NT>([&](auto J) __attribute__((always_inline)) {
      static_for<NT>([&](auto I) __attribute__((always_inline)) {
        if constexpr (I >= J && TM::wave(I, J) == WAVE) {
          constexpr int S = TM::slot(I, J);
          const int col = 16 * J + l15;
          static_for<4>([&](auto j) __attribute__((always_inline)) {
            const int row = 16 * I + l4 + 4 * j;
            if (16 * I + 15 >= r && (row >= r || col >= r)) acc[S][j()] = 0.0;
            if (I == J && row == col) {
              if (row < r) acc[S][j()] += P.lam * dvec[row];
              else if (row < rE) acc[S][j()] = 1.0;
            }
            if (row == rE) acc[S][j()] = (col < r) ? tvec[col] : 0.0;
          });
        }
      });
    });

    // ---- blocked Cholesky, 4-wide panels ------------------------------------
    // Outer loop over tile columns is compile-time (all tile predicates fold);
    // inner loop over the four 4-wide sub-panels of a tile column is a runtime loop.
    static_for<NT>([&](auto JB) __attribute__((always_inline)) {
      constexpr int Jb = JB;
      const int qend = (NS - 4 * Jb) < 4 ? (NS - 4 * Jb) : 4;      // <=0 past the last panel
      for (int q = 0; q < qend; ++q) {
        const int s = 4 * Jb + q;
        const int c0 = 4 * s;
        double* LTb = LT + (s & 1) * 4 * RP;
        const bool mycols = (l15 >> 2) == q;
        // (1) panel columns -> LDS
        static_for<NT>([&](auto I) __attribute__((always_inline)) {
          if constexpr (I >= Jb && TM::wave(I, Jb) == WAVE) {
            if (mycols) {
              constexpr int S = TM::slot(I, Jb);
              static_for<4>([&](auto j) __attribute__((always_inline)) {
                PT[(lane & 3) * RP + 16 * I + l4 + 4 * j] = acc[S][j()];
              });
            }
          }
        });
        __syncthreads();
        // (2) factor the 4x4 diagonal block (redundantly) and solve one row per thread
        {
          const double* Pd = PT + c0;
          const double p00 = Pd[0 * RP + 0];
          const double p10 = Pd[0 * RP + 1], p11 = Pd[1 * RP + 1];
          const double p20 = Pd[0 * RP + 2], p21 = Pd[1 * RP + 2], p22 = Pd[2 * RP + 2];
          const double p30 = Pd[0 * RP + 3], p31 = Pd[1 * RP + 3], p32 = Pd[2 * RP + 3], p33 = Pd[3 * RP + 3];
          const double d0 = p00;
          const double i0 = rsqrt(d0);
          const double l10 = p10 * i0, l20 = p20 * i0, l30 = p30 * i0;
          const double d1 = p11 - l10 * l10;
          const double i1 = rsqrt(d1);
          const double l21 = (p21 - l20 * l10) * i1, l31 = (p31 - l30 * l10) * i1;
          const double d2 = p22 - l20 * l20 - l21 * l21;
          const double i2 = rsqrt(d2);
          const double l32 = (p32 - l30 * l20 - l31 * l21) * i2;
          const double d3 = p33 - l30 * l30 - l31 * l31 - l32 * l32;
          const double i3 = rsqrt(d3);
          if (tid == 0) {
            const bool ok = (d0 > 0.0) && (d1 > 0.0) && (d2 > 0.0) && (d3 > 0.0) &&
                            (i0 < 1e150) && (i1 < 1e150) && (i2 < 1e150) && (i3 < 1e150);
            if (!ok) flags[0] = 1;
            invd[c0 + 0] = i0; invd[c0 + 1] = i1; invd[c0 + 2] = i2; invd[c0 + 3] = i3;
          }
          for (int row = tid; row < RP; row += NTHR) {
            double x0, x1, x2, x3;
            if (row < c0) { x0 = x1 = x2 = x3 = 0.0; }
            else if (row >= c0 + 4) {
              x0 = PT[0 * RP + row] * i0;
              x1 = (PT[1 * RP + row] - x0 * l10) * i1;
              x2 = (PT[2 * RP + row] - x0 * l20 - x1 * l21) * i2;
              x3 = (PT[3 * RP + row] - x0 * l30 - x1 * l31 - x2 * l32) * i3;
            } else {
              const int i = row - c0;
              x0 = sel4(i, d0 * i0, l10, l20, l30);
              x1 = sel4(i, 0.0, d1 * i1, l21, l31);
              x2 = sel4(i, 0.0, 0.0, d2 * i2, l32);
              x3 = sel4(i, 0.0, 0.0, 0.0, d3 * i3);
            }
            LTb[0 * RP + row] = x0; LTb[1 * RP + row] = x1; LTb[2 * RP + row] = x2; LTb[3 * RP + row] = x3;
          }
        }
        __syncthreads();
        // (3) rank-4 trailing update on the register tiles + keep L in the panel columns
        {
          const int lim = c0 + 4;                 // rows/cols below this are final
          double op[NT];
          static_for<NT>([&](auto I) __attribute__((always_inline)) {
            if constexpr (I >= Jb) {
              const double v = LTb[l4 * RP + 16 * I + l15];
              if constexpr (I == Jb) op[I] = (16 * I + l15 >= lim) ? v : 0.0;
              else op[I] = v;                      // every row/col of a later tile is >= lim
            }
          });
          static_for<NT>([&](auto J) __attribute__((always_inline)) {
            static_for<NT>([&](auto I) __attribute__((always_inline)) {
              if constexpr (J >= Jb && I >= J && TM::wave(I, J) == WAVE) {
                constexpr int S = TM::slot(I, J);
                if constexpr (J > Jb) {
                  acc[S] = __builtin_amdgcn_mfma_f64_16x16x4f64(-op[I], op[J], acc[S], 0, 0, 0);
                } else {
                  if (q < 3) acc[S] = __builtin_amdgcn_mfma_f64_16x16x4f64(-op[I], op[J], acc[S], 0, 0, 0);
                  if (mycols) {
                    static_for<4>([&](auto j) __attribute__((always_inline)) {
                      acc[S][j()] = LTb[(lane & 3) * RP + 16 * I + l4 + 4 * j];
                    });
                  }
                }
              }
            });
          });
        }
      }
    });

    // ---- y = L^-1 t sits in row rE of the tiles; hand it to the row-owner threads
    static_for<NT>([&](auto J) __attribute__((always_inline)) {
      static_for<NT>([&](auto I) __attribute__((always_inline)) {
        if constexpr (I >= J && TM::wave(I, J) == WAVE) {
          if (I == IR) {
            constexpr int S = TM::slot(I, J);
            static_for<4>([&](auto j) __attribute__((always_inline)) {
              if (l4 + 4 * j == rr) beta[16 * J + l15] = acc[S][j()];   // beta[] doubles as y buffer
            });
          }
        }
      });
    });
    __syncthreads();
    // ---- back substitution L' beta = y, one 4-row block per step ----------------
    // Tile rows from the bottom up (compile-time), four 4-row blocks each (runtime).
    {
      constexpr int NE = (RP + NTHR - 1) / NTHR;
      double yv[NE];
      static_for<NE>([&](auto e) __attribute__((always_inline)) {
        const int i = tid + e * NTHR;
        yv[e()] = (i < rE) ? beta[i] : 0.0;
      });
      static_for<NT>([&](auto IREV) __attribute__((always_inline)) {
        constexpr int Is = NT - 1 - IREV;
        const int qtop = (NS - 4 * Is) < 4 ? (NS - 4 * Is) : 4;
        for (int q = qtop - 1; q >= 0; --q) {
          const int s = 4 * Is + q;
          const int c0 = 4 * s;
          // rows c0..c0+3 of L (all columns) -> LROW; their y values -> yc
          static_for<NT>([&](auto J) __attribute__((always_inline)) {
            if constexpr (J <= Is && TM::wave(Is, J) == WAVE) {
              constexpr int S = TM::slot(Is, J);
              const d4 av = acc[S];
              const double v = sel4(q, av[0], av[1], av[2], av[3]);
              LROW[l4 * RP + 16 * J + l15] = v;
            }
          });
          static_for<NE>([&](auto e) __attribute__((always_inline)) {
            const int i = tid + e * NTHR;
            if (i >= c0 && i < c0 + 4) yc[i - c0] = yv[e()];
          });
          __syncthreads();
          const double* LR = LROW;
          const double i0 = invd[c0], i1 = invd[c0 + 1], i2 = invd[c0 + 2], i3 = invd[c0 + 3];
          const double L10 = LR[1 * RP + c0], L20 = LR[2 * RP + c0], L21 = LR[2 * RP + c0 + 1];
          const double L30 = LR[3 * RP + c0], L31 = LR[3 * RP + c0 + 1], L32 = LR[3 * RP + c0 + 2];
          const double b3 = yc[3] * i3;
          const double b2 = (yc[2] - L32 * b3) * i2;
          const double b1 = (yc[1] - L21 * b2 - L31 * b3) * i1;
          const double b0 = (yc[0] - L10 * b1 - L20 * b2 - L30 * b3) * i0;
          static_for<NE>([&](auto e) __attribute__((always_inline)) {
            const int i = tid + e * NTHR;
            if (i < c0) {
              yv[e()] -= LR[0 * RP + i] * b0 + LR[1 * RP + i] * b1 + LR[2 * RP + i] * b2 + LR[3 * RP + i] * b3;
            } else if (i < c0 + 4) {
              beta[i] = sel4(i - c0, b0, b1, b2, b3);
            }
          });
          __syncthreads();
        }
      });
    }

    // ---- slack box: primal-dual active-set update --------------------------------
    bool again = false;
    if (P.convex) {
      const double scale = -P.lam / P.lamb_sigma;
      for (int rho = tid; rho < r; rho += NTHR) {
        const int k = rho / nch, ch = rho - k * nch;
        if (ch >= P.m && k >= P.n) {               // sigma[n*p:], controller.py:659
          const double sh = scale * beta[rho];
          const int ns = (sh > P.bound) ? 1 : (sh < -P.bound) ? -1 : 0;
          if (ns != act[rho]) { act[rho] = ns; flags[1] = 1; }
        }
      }
      __syncthreads();
      again = (flags[1] != 0) && (flags[0] == 0);
      if (again && iter >= P.max_iter) { again = false; status = 4; }
    }
    if (!again) break;
  }
  if (flags[0] != 0) status = 4;

  // ---- outputs --------------------------------------------------------------------
  // z = t - lam*D*beta; cost = control cost + lam*beta'z + lamb_sigma*|sigma|^2
  double part = 0.0;
  bool finite = true;
  for (int rho = tid; rho < r; rho += NTHR) {
    const int s_act = act[rho];
    const Comp cmp = comp_of(P, rho, s_act, up, yp);
    const double b = beta[rho];
    const double z = cmp.t - P.lam * cmp.D * b;
    finite = finite && (fabs(b) < 1e300);
    const int k = rho / nch, ch = rho - k * nch;
    const int kp = k - P.n;
    double contrib = P.lam * b * z;
    switch (cmp.kind) {
      case K_UFREE: { const double d = z - cmp.t; contrib += cmp.wq * d * d; } break;
      case K_YFREE: { const double d = z - cmp.t; contrib += cmp.wq * d * d; } break;
      case K_WINT: { const double sg = z - cmp.t; contrib += P.lamb_sigma * sg * sg; } break;
      case K_WTERM: { const double sg = z - P.y_s[ch - P.m]; contrib += P.lamb_sigma * sg * sg; } break;
      case K_WPRED: {
        const double sg = (s_act != 0) ? s_act * P.bound : -P.lam * b / P.lamb_sigma;
        const double d = z - sg - P.y_s[ch - P.m];
        contrib += cmp.wq * d * d + P.lamb_sigma * sg * sg;
      } break;
      default: break;
    }
    part += contrib;
    if (ch < P.m && kp >= 0) u_opt[kp * P.m + ch] = z;      // ubar[n*m:], controller.py:799-805
    if (beta_ws) beta_ws[rho] = b;
    if (act_ws) act_ws[rho] = (signed char)s_act;
  }
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
  }
}

// --------------------------------------------------------------------------
// Cold-solve kernel: grid = batch, block = 64*W threads.
// --------------------------------------------------------------------------
template <int NT, int W>
__global__ __launch_bounds__(64 * W, (W <= 4 ? 2 : 1)) void ddmpc_cold_solve_kernel(
    KParams P, const double* __restrict__ u_d, const double* __restrict__ y_d,
    const double* __restrict__ u_past, const double* __restrict__ y_past, double* __restrict__ u_opt,
    double* __restrict__ cost, int* __restrict__ status, int* __restrict__ iters,
    double* __restrict__ beta_ws, signed char* __restrict__ act_ws) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const long long b = blockIdx.x;
  const int tid = threadIdx.x;
  constexpr int NTHR = 64 * W;
  const Lds<NT> lds(P.xs_len);
  double* xs = sm + lds.xs;
  // ---- stage the instance's trajectory, channel-interleaved: xs[t*nch + ch] ----
  {
    const double* ud = u_d + b * (long long)P.N * P.m;
    const double* yd = y_d + b * (long long)P.N * P.p;
    const int nu = P.N * P.m, ny = P.N * P.p;
    for (int i = tid; i < nu; i += NTHR) {
      const int t = i / P.m, ch = i - t * P.m;
      xs[t * P.nch + ch] = ud[i];
    }
    for (int i = tid; i < ny; i += NTHR) {
      const int t = i / P.p, ch = i - t * P.p;
      xs[t * P.nch + P.m + ch] = yd[i];
    }
    for (int i = P.N * P.nch + tid; i < P.xs_len; i += NTHR) xs[i] = 0.0;
  }
  const double* up = u_past + b * (long long)(P.n * P.m);
  const double* yp = y_past + b * (long long)(P.n * P.p);
  double* uo = u_opt + b * (long long)(P.L * P.m);
  double* bw = beta_ws ? beta_ws + b * (long long)P.rE : nullptr;
  signed char* aw = act_ws ? act_ws + b * (long long)P.rE : nullptr;
  int* it = iters ? iters + b : nullptr;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  static_for<W>([&](auto WV) __attribute__((always_inline)) {
    if (wave == WV) wave_body<NT, W, WV>(P, sm, up, yp, uo, cost + b, status + b, it, bw, aw);
  });
}

#ifdef DDMPC_WITH_AUX_KERNELS   // non-template kernels: defined once, in the API translation unit
// --------------------------------------------------------------------------
// hankel_matrix for a batch: H[b][k*nch+ch][i] = X[b][i+k][ch]
// (direct_data_driven_mpc/utilities/hankel_matrix.py:47-51)
// --------------------------------------------------------------------------
__global__ void ddmpc_hankel_kernel(const double* __restrict__ X, double* __restrict__ H, int N, int nch,
                                    int L, long long batch) {
  const int cols = N - L + 1;
  const long long per = (long long)L * nch * cols;
  const long long total = per * batch;
  for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < total;
       g += (long long)gridDim.x * blockDim.x) {
    const long long b = g / per;
    const long long e = g - b * per;
    const int row = (int)(e / cols), i = (int)(e - (long long)row * cols);
    const int k = row / nch, ch = row - k * nch;
    H[g] = X[b * (long long)N * nch + (long long)(i + k) * nch + ch];
  }
}

// --------------------------------------------------------------------------
// Variable reconstruction for ddmpc_get_solution (the `.value` stand-ins of
// controller.py:434-445) from the beta / active-set workspace of the last solve.
// what: 0 alpha, 1 ubar, 2 ybar, 3 sigma.  One workgroup per instance.
// --------------------------------------------------------------------------
__global__ void ddmpc_reconstruct_kernel(KParams P, int what, const double* __restrict__ u_d,
                                         const double* __restrict__ y_d, const double* __restrict__ u_past,
                                         const double* __restrict__ y_past, const double* __restrict__ beta_ws,
                                         const signed char* __restrict__ act_ws, double* __restrict__ out) {
  const long long b = blockIdx.x;
  const double* bw = beta_ws + b * (long long)P.rE;
  const signed char* aw = act_ws + b * (long long)P.rE;
  const double* up = u_past + b * (long long)(P.n * P.m);
  const double* yp = y_past + b * (long long)(P.n * P.p);
  if (what == 0) {                       // alpha = H' beta
    const double* ud = u_d + b * (long long)P.N * P.m;
    const double* yd = y_d + b * (long long)P.N * P.p;
    double* o = out + b * (long long)P.c;
    for (int i = threadIdx.x; i < P.c; i += blockDim.x) {
      double s = 0.0;
      for (int k = 0; k < P.Ln; ++k) {
        for (int ch = 0; ch < P.m; ++ch) s += ud[(i + k) * P.m + ch] * bw[k * P.nch + ch];
        for (int ch = 0; ch < P.p; ++ch) s += yd[(i + k) * P.p + ch] * bw[k * P.nch + P.m + ch];
      }
      o[i] = s;
    }
    return;
  }
  for (int rho = threadIdx.x; rho < P.r; rho += blockDim.x) {
    const int k = rho / P.nch, ch = rho - k * P.nch;
    const int s_act = aw[rho];
    const Comp cmp = comp_of(P, rho, s_act, up, yp);
    const double bb = bw[rho];
    const double z = cmp.t - P.lam * cmp.D * bb;
    if (ch < P.m) {
      if (what == 1) out[b * (long long)(P.Ln * P.m) + k * P.m + ch] = z;
      continue;
    }
    const int cy = ch - P.m;
    double sg = 0.0;
    switch (cmp.kind) {
      case K_WINT: sg = z - cmp.t; break;
      case K_WTERM: sg = z - P.y_s[cy]; break;
      case K_WPRED: sg = (s_act != 0) ? s_act * P.bound : -P.lam * bb / P.lamb_sigma; break;
      default: sg = 0.0; break;
    }
    if (what == 2) out[b * (long long)(P.Ln * P.p) + k * P.p + cy] = z - sg;
    if (what == 3) out[b * (long long)(P.Ln * P.p) + k * P.p + cy] = sg;
  }
}

#endif  // DDMPC_WITH_AUX_KERNELS

}  // namespace ddmpc
