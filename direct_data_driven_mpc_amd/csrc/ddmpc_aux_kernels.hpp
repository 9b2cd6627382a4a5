// ddmpc_aux_kernels.hpp -- the small, non-template gfx950 kernels around the cold-solve kernel:
// Hankel gather, variable reconstruction, plant/FIFO step, the warm path (gain, warm step, fused
// closed loop) and the persistent-excitation guard.  Included by the API translation unit only, so
// editing it does not rebuild the (slow to compile) cold-solve instantiations.
#pragma once
#include "ddmpc_cold2.hpp"

namespace ddmpc {

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
// `RPs` is the row stride of the component tables.
// --------------------------------------------------------------------------
__global__ void ddmpc_reconstruct_kernel(KParams P, int RPs, int what, const double* __restrict__ u_d,
                                         const double* __restrict__ y_d, const double* __restrict__ u_past,
                                         const double* __restrict__ y_past, const double* __restrict__ beta_ws,
                                         const signed char* __restrict__ act_ws, double* __restrict__ out,
                                         const double* __restrict__ z_ws, const int* __restrict__ rescued,
                                         const double* __restrict__ x_ws) {
  const long long b = blockIdx.x;
  const int n = P.npu / P.m;
  const bool resc = rescued != nullptr && rescued[b] != 0;
  if (resc && what != 0) {
    // NOMINAL instance solved by the rank-revealing kernel (exact data): it exports z = [ubar; ybar] per component and
    // the vector x = L^-T w with z = H (H' x): alpha = H' x below, the minimum-norm alpha with H alpha = z (controller.py:434)
    for (int rho = threadIdx.x; rho < P.r; rho += blockDim.x) {
      const int k = rho / P.nch, ch = rho - k * P.nch;
      const double z = z_ws[b * (long long)P.rE + rho];
      if (ch < P.m) { if (what == 1) out[b * (long long)(P.Ln * P.m) + k * P.m + ch] = z; }
      else if (what == 2) out[b * (long long)(P.Ln * P.p) + k * P.p + (ch - P.m)] = z;
    }
    return;
  }
  const double* bw = (resc ? x_ws : beta_ws) + b * (long long)P.rE;
  const signed char* aw = act_ws + b * (long long)P.rE;
  const double* up = u_past + b * (long long)P.npu;
  const double* yp = y_past + b * (long long)(n * P.p);
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
    const int kind = P.tabi[0 * RPs + rho];
    const int pidx = P.tabi[1 * RPs + rho];
    const double tb = P.tabd[2 * RPs + rho];
    const double D = s_act ? P.tabd[1 * RPs + rho] : P.tabd[0 * RPs + rho];
    const double tpast = (pidx >= 0) ? ((pidx < P.npu) ? up[pidx] : yp[pidx - P.npu]) : tb;
    const double t = tpast + s_act * P.bound;
    const double bb = bw[rho];
    double z = t - P.lam * D * bb;
    if (P.dense_w) {
      const double* dr = P.dmat + (long long)rho * RPs;
      double sdb = 0.0;
      for (int j = 0; j < P.r; ++j) sdb += dr[j] * bw[j];
      z = t - P.lam * (D * bb + sdb);
    }
    if (ch < P.m) {
      if (what == 1) out[b * (long long)(P.Ln * P.m) + k * P.m + ch] = z;
      continue;
    }
    const int cy = ch - P.m;
    double sg = 0.0;
    if (kind == K_WINT) sg = z - tpast;
    else if (kind == K_WTERM) sg = z - tb;
    else if (kind == K_WPRED) sg = (s_act != 0) ? s_act * P.bound : -P.lam * bb / P.lamb_sigma;
    if (what == 2) out[b * (long long)(P.Ln * P.p) + k * P.p + cy] = z - sg;
    if (what == 3) out[b * (long long)(P.Ln * P.p) + k * P.p + cy] = sg;
  }
}
// --------------------------------------------------------------------------
// Closed-loop glue: apply up to `nsub` inputs of the last solve to the plant, record the
// trajectories, push (u,y) into the past windows.  One thread per instance (tiny matvecs).
//   pl: [A (ns*ns) | B (ns*m) | C (p*ns) | D (p*m)] row-major.
// utilities/controller/controller_operation.py:278-305, utilities/model_simulation.py:93-98,
// direct_data_driven_mpc_controller.py:893-895.
// --------------------------------------------------------------------------
__global__ void ddmpc_plant_kernel(long long batch, int ns, int m, int p, int n, int Lm, const double* __restrict__ pl,
                                   int t0, int nsub, int n_steps, const double* __restrict__ u_opt,
                                   const int* __restrict__ st_step, int* __restrict__ st_acc,
                                   double* __restrict__ x, double* __restrict__ u_past, double* __restrict__ y_past,
                                   const double* __restrict__ w, double* __restrict__ u_sys, double* __restrict__ y_sys) {
  const long long b = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= batch) return;
  const double* A = pl;
  const double* Bm = A + ns * ns;
  const double* C = Bm + ns * m;
  const double* D = C + p * ns;
  int acc = st_acc[b];
  if (st_step[b] > acc) acc = st_step[b];
  st_acc[b] = acc;
  double* xb = x + b * ns;
  double* up = u_past + b * (long long)(n * m);
  double* yp = y_past + b * (long long)(n * p);
  const double* uo = u_opt + b * (long long)Lm;
  const double nanv = __longlong_as_double(0x7ff8000000000000LL);
  for (int j = 0; j < nsub; ++j) {
    const int k = t0 + j;
    double* us = u_sys + (b * n_steps + k) * m;
    double* ys = y_sys + (b * n_steps + k) * p;
    if (acc > 1) {                               // not optimal / optimal_inaccurate: the reference raises here
      for (int i = 0; i < m; ++i) us[i] = nanv;
      for (int i = 0; i < p; ++i) ys[i] = nanv;
      continue;
    }
    const double* uk = uo + j * m;
    const double* wk = w + (b * n_steps + k) * p;
    for (int i = 0; i < p; ++i) {                // y = C x + D u + w, with the state BEFORE the update
      double s = wk[i];
      for (int q = 0; q < ns; ++q) s += C[i * ns + q] * xb[q];
      for (int q = 0; q < m; ++q) s += D[i * m + q] * uk[q];
      ys[i] = s;
    }
    double xn[16];
    for (int i = 0; i < ns; ++i) {
      double s = 0.0;
      for (int q = 0; q < ns; ++q) s += A[i * ns + q] * xb[q];
      for (int q = 0; q < m; ++q) s += Bm[i * m + q] * uk[q];
      xn[i] = s;
    }
    for (int i = 0; i < ns; ++i) xb[i] = xn[i];
    for (int i = 0; i < m; ++i) us[i] = uk[i];
    for (int i = 0; i < (n - 1) * m; ++i) up[i] = up[i + m];       // FIFO shift
    for (int i = 0; i < m; ++i) up[(n - 1) * m + i] = uk[i];
    for (int i = 0; i < (n - 1) * p; ++i) yp[i] = yp[i + p];
    for (int i = 0; i < p; ++i) yp[(n - 1) * p + i] = ys[i];
  }
}
// --------------------------------------------------------------------------
// Warm path (ddmpc_prepare / ddmpc_step), slack NONE and nominal controllers.
// Without inequality constraints the QP is equality-constrained, so beta is an affine
// function of the past window:  beta = A^-1 t,  t = t0 + sum_f past[f] e_{rho(f)}  with the
// step-invariant A = G + lam*D0 (controller.py:404-407 re-solves with only u_past/y_past
// changed, :577-581).  ddmpc_gain_kernel solves the nf+1 right-hand sides
// [t0 | e_rho(0) .. e_rho(nf-1)] once per instance against the exported Cholesky factor;
// a warm step is then one [r x (nf+1)] matrix-vector product and the usual output stage.
//
// gain layout: gain[b][j][rho], j = 0 (offset) .. nf, rho < r (stride r): lanes = rho coalesce.
// One wave per instance, lane = right-hand side, the column being solved lives in LDS
// (Y[i*CH + lane]); L entries are wave-uniform (scalar loads).  Setup-time only.
// --------------------------------------------------------------------------
__device__ __forceinline__ const double* lfac_row(const double* __restrict__ Lb, int i, int J) {
  const int I = i >> 4;
  return Lb + ((I * (I + 1) / 2 + J) << 8) + ((i & 15) << 4);
}

// Sum over the 16 lanes of a DPP row (all 16 lanes receive the total): four rotate-and-add steps.
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double row16_sum(double v) {
  v += dpp_f64<0x128>(v);   // row_ror:8
  v += dpp_f64<0x124>(v);   // row_ror:4
  v += dpp_f64<0x122>(v);   // row_ror:2
  v += dpp_f64<0x121>(v);   // row_ror:1
  return v;
}

// Gain kernel: grid = batch, block = 256 = 4 waves x 4 groups of 16 lanes.  One group solves one
// right-hand side e_rho(f): lane kk of the group owns the unknowns k = 16 J + kk in registers, and a row's
// dot product is <= NT multiply-adds per lane plus a 16-lane DPP sum.  The factor streams through LDS one
// tile row at a time (L from `lfac` going forward, L' from the tile-transposed copy going back): the whole
// workgroup copies the next tile row (coalesced, <= NT x 2 KB) into registers while the 16 rows of the
// current one are processed, so HBM latency is paid once per 16 rows; the 16 reciprocal pivots of a tile
// row are formed once, by lane.  Column 0 of the gain (the offset A^-1 t0) is the beta of the cold solve
// that exported the factor (zero past window).
template <int NT>
__global__ __launch_bounds__(256) void ddmpc_gain_kernel(KParams P, int RPs, int nf, const double* __restrict__ lfac,
                                                         const double* __restrict__ lfacT,
                                                         const double* __restrict__ beta0, double* __restrict__ gain) {
  extern __shared__ __attribute__((aligned(16))) double gsm[];      // 2 x NT x 256 (tile rows) + 2 x 16 (reciprocal pivots)
  auto Lbuf = [&](int buf) __attribute__((always_inline)) -> double* { return gsm + buf * (NT * 256); };
  auto rinv = [&](int buf) __attribute__((always_inline)) -> double* { return gsm + 2 * NT * 256 + buf * 16; };
  const long long b = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, grp = lane >> 4, kk = lane & 15, wave = tid >> 6;
  const int r = P.r, nrhs = nf + 1;
  const double* __restrict__ Lb = lfac + b * (long long)(NT * (NT + 1) / 2 * 256);
  const double* __restrict__ Tb = lfacT + b * (long long)(NT * (NT + 1) / 2 * 256);
  for (int rho = tid; rho < r; rho += 256) gain[(b * nrhs) * (long long)r + rho] = beta0[b * (long long)P.rE + rho];
  auto tile = [](int I, int J) { return (I * (I + 1) / 2 + J) << 8; };
  const int ntr = (r + 15) >> 4;                      // tile rows that hold real rows
  double stage[NT];
  // forward: tile row I = tiles (I, 0..I), contiguous in lfac; slot J of the buffer = tile (I, J)
  auto fetch_fwd = [&](int I) __attribute__((always_inline)) {
    static_for<NT>([&](auto J) __attribute__((always_inline)) { if (J <= I) stage[J] = Lb[tile(I, J) + tid]; });
  };
  // backward: tile column Ji of the transposed copy = tiles (I2, Ji), I2 >= Ji; slot I2 of the buffer
  auto fetch_bwd = [&](int Ji) __attribute__((always_inline)) {
    static_for<NT>([&](auto I2) __attribute__((always_inline)) { if (I2 >= Ji && I2 < ntr) stage[I2] = Tb[tile(I2, Ji) + tid]; });
  };
  auto commit = [&](int buf, int lo, int hi) __attribute__((always_inline)) {      // slots lo..hi of `stage` -> LDS
    static_for<NT>([&](auto J) __attribute__((always_inline)) { if (J >= lo && J <= hi) Lbuf(buf)[J * 256 + tid] = stage[J]; });
  };
  for (int f0 = 0; f0 < nf; f0 += 16) {
    const int f = f0 + wave * 4 + grp;                 // this group's right-hand side (idle groups ride along)
    int rho_f = -1;
    for (int rho = kk; rho < r; rho += 16) rho_f = (P.tabi[1 * RPs + rho] == f) ? rho : rho_f;
    {                                                  // max over the group (exactly one lane found it)
      double t = (double)rho_f;
      t = fmax(t, dpp_f64<0x128>(t)); t = fmax(t, dpp_f64<0x124>(t));
      t = fmax(t, dpp_f64<0x122>(t)); t = fmax(t, dpp_f64<0x121>(t));
      rho_f = (int)t;
    }
    double y[NT];
    static_for<NT>([&](auto J) __attribute__((always_inline)) { y[J] = 0.0; });
    // ---- forward substitution L y = e ------------------------------------------------------
    __syncthreads();                                   // buffers free (previous pass done)
    fetch_fwd(0);
    commit(0, 0, 0);
    for (int I = 0; I < ntr; ++I) {
      const int buf = I & 1;
      __syncthreads();                                 // tile row I visible; the other buffer is free
      if (I + 1 < ntr) fetch_fwd(I + 1);               // in flight while this tile row is processed
      if (wave == 0 && lane < 16) rinv(buf)[lane] = 1.0 / Lbuf(buf)[I * 256 + lane * 17];
      __syncthreads();
      const double* Lr = Lbuf(buf) + kk;
      const int nrow = (r - 16 * I) < 16 ? (r - 16 * I) : 16;
      for (int ii = 0; ii < nrow; ++ii) {
        double s = 0.0;
        static_for<NT>([&](auto J) __attribute__((always_inline)) {
          if (J < I) s += Lr[J * 256 + ii * 16] * y[J];
          if (J == I) s += (kk < ii) ? Lr[J * 256 + ii * 16] * y[J] : 0.0;
        });
        s = row16_sum(s);
        const double yi = ((16 * I + ii == rho_f ? 1.0 : 0.0) - s) * rinv(buf)[ii];
        static_for<NT>([&](auto J) __attribute__((always_inline)) { if (J == I && kk == ii) y[J] = yi; });
      }
      if (I + 1 < ntr) commit(buf ^ 1, 0, I + 1);
    }
    // ---- back substitution L' x = y: x_i = (y_i - sum_{k>i} L[k][i] x_k) / L[i][i] -------------
    __syncthreads();
    fetch_bwd(ntr - 1);
    commit(0, ntr - 1, ntr - 1);
    for (int Ji = ntr - 1; Ji >= 0; --Ji) {
      const int buf = (ntr - 1 - Ji) & 1;
      __syncthreads();
      if (Ji > 0) fetch_bwd(Ji - 1);
      if (wave == 0 && lane < 16) rinv(buf)[lane] = 1.0 / Lbuf(buf)[Ji * 256 + lane * 17];
      __syncthreads();
      const double* Tr = Lbuf(buf) + kk;
      const int nrow = (r - 16 * Ji) < 16 ? (r - 16 * Ji) : 16;
      for (int ii = nrow - 1; ii >= 0; --ii) {
        const int i = 16 * Ji + ii;
        double s = 0.0;
        static_for<NT>([&](auto I2) __attribute__((always_inline)) {
          if (I2 >= Ji && I2 < ntr) {
            const int k = 16 * I2 + kk;
            if (k > i && k < r) s += Tr[I2 * 256 + ii * 16] * y[I2];
            if (k == i) s -= y[I2];                    // the owner lane folds y_i into the sum: no broadcast needed
          }
        });
        s = row16_sum(s);
        const double xi = -s * rinv(buf)[ii];
        static_for<NT>([&](auto J) __attribute__((always_inline)) { if (J == Ji && kk == ii) y[J] = xi; });
      }
      if (Ji > 0) commit(buf ^ 1, Ji - 1, ntr - 1);
    }
    if (f < nf) {
      double* g = gain + (b * nrhs + 1 + f) * (long long)r;
      static_for<NT>([&](auto J) __attribute__((always_inline)) { if (16 * J + kk < r) g[16 * J + kk] = y[J]; });
    }
  }
}

// Refined affine law (ddmpc_prepare with DDMPC_REFINE_ALWAYS, or _AUTO for flagged instances): the columns of the gain are obtained from full refining cold
// solves instead of substitutions through the unrefined factor -- beta is affine in the past window, so column 1 + f is
// beta(past = e_f) - beta(past = 0).  Two helpers: the unit past window e_f for the whole batch (f < 0: all zero), and
// the difference of a solve's beta with the offset column.
__global__ void ddmpc_unit_past_kernel(long long batch, int npu, int npy, int f, double* __restrict__ up, double* __restrict__ yp) {
  const long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  const int np = npu + npy;
  if (idx >= batch * np) return;
  const long long b = idx / np;
  const int i = (int)(idx - b * np);
  const double v = (i == f) ? 1.0 : 0.0;
  if (i < npu) up[b * npu + i] = v; else yp[b * npy + (i - npu)] = v;
}
// The past window a controller starts from: the last n steps of its own data (controller.py:184-185), for the whole batch.
// ddmpc_prepare (AUTO) probes the conditioning of every data set with a solve at THIS window: the factor-export solve runs
// at the zero window, whose right-hand side vanishes altogether for zero setpoints (beta = 0, nothing to flag).
__global__ void ddmpc_tail_past_kernel(long long batch, int N, int m, int p, int n, const double* __restrict__ u_d,
                                       const double* __restrict__ y_d, double* __restrict__ up, double* __restrict__ yp) {
  const long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  const int npu = n * m, npy = n * p, np = npu + npy;
  if (idx >= batch * np) return;
  const long long b = idx / np;
  const int i = (int)(idx - b * np);
  if (i < npu) up[b * npu + i] = u_d[(b * N + (N - n)) * m + i];
  else yp[b * npy + (i - npu)] = y_d[(b * N + (N - n)) * p + (i - npu)];
}
__host__ __device__ inline size_t gram_tiles_lds_doubles(int xs_len, int r, int nch, int NT) {   // LDS of ddmpc_gram_tiles_kernel
  return (size_t)xs_len + (size_t)r * nch + 2 + (size_t)16 * NT;           // xs | ctab | offA, offB
}
// ---------------------------------------------------------------------------------------------------------------
// G = H H' of every instance for the register-resident kernels, through the Hankel structure, for ANY channel count
// (hankel_matrix.py:5-53 is generic in the channel count; the in-kernel lag-block + tile-walk Gram of ddmpc_cold2.hpp needs
// m + p == 4: its walk moves a whole 16 x 16 tile by 16 rows = 4 time steps, and for a channel count that does not divide 16
// a shift by 16 rows is not a shift in time).  Rows in the kernel's order rho = (time, channel), x[rho + i nch] = H[rho][i]:
//     K[rho + nch][sig + nch] = K[rho][sig] + x[rho + c nch] x[sig + c nch] - x[rho] x[sig]                (window slides by one)
// so only the entries with sig < nch need the full length-c sum -- the first nch rows of tile column 0 of the dense product --
// and every other entry is two multiply-adds on top of its neighbour one time step up the diagonal.  One workgroup per
// instance: trajectory into LDS as the cold-solve kernel holds it, those rows on the matrix pipe (tile I on wave I mod 4),
// then the chains.  Output in the accumulator layout of the cold-solve kernel (KParams::gpre): register j of lane (l4, l15) of
// tile (I, J), I >= J, holds K[16 J + l4 + 4 j][16 I + l15]; diagonal tiles are filled on both sides; rows past r are not written
// (the kernel's fix-up overwrites them).  LDS: gram_tiles_lds_doubles().  grid = batch, 256 threads.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ddmpc_gram_tiles_kernel(KParams P, int NT, const double* __restrict__ u_d,
                                                               const double* __restrict__ y_d, double* __restrict__ gpre,
                                                               long long gstride) {
  extern __shared__ __attribute__((aligned(16))) double gt_lds[];
  double* xs = gt_lds;
  double* ctab = gt_lds + P.xs_len;                                         // ctab[rho * nch + b] = K[rho][b], b < nch
  const long long b = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, l4 = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nch = P.nch, c = P.c, r = P.r, Ln = P.Ln;
  {
    const double* ud = u_d + b * (long long)P.N * P.m;
    const double* yd = y_d + b * (long long)P.N * P.p;
    const int nu = P.N * P.m, ny = P.N * P.p;
    for (int i = tid; i < nu; i += 256) {
      const int t = i / P.m, ch = i - t * P.m;
      xs[t * nch + ch] = ud[i];
    }
    for (int i = tid; i < ny; i += 256) {
      const int t = i / P.p, ch = i - t * P.p;
      xs[t * nch + P.m + ch] = yd[i];
    }
    for (int i = P.N * nch + tid; i < P.xs_len; i += 256) xs[i] = 0.0;
  }
  __syncthreads();
  {
    // First nch rows of tile column 0 on v_mfma_f64_4x4x4 (four independent 4x4x4 products per instruction, ~16 clk: a
    // quarter of a 16x16x4 -- of whose 16 output rows only nch would be used, 2 of 16 for a SISO plant).  Block blk of the
    // instruction for (row group sg, tile t): D[i][j] = K[4 sg + i][16 t + 4 blk + j]; lane = 16 k + 4 blk + i holds
    // A[i][k] = H[4 sg + i][i0 + k], lane = 16 k + 4 blk + j holds B[k][j] = H[16 t + 4 blk + j][i0 + k] (the same A in all four
    // blocks); D[i][j] comes back in lane 16 i + 4 blk + j (layout probed in tools/mfma_f64_4x4_probe.hip).
    constexpr int TW = 5, SG = 4;                                           // tiles per wave (NT <= 17 on four waves), row groups per pass
    const int nsg = (nch + 3) >> 2;
    const int kq = lane >> 4, ij = lane & 3;
    const int cfull = c & ~3;
    // (more than 16 channels: further passes of four row groups over the trajectory -- until round 5 only the first 16
    //  channels' lag sums were formed and wider plants read unwritten LDS)
#pragma unroll 1
    for (int sg0 = 0; sg0 < nsg; sg0 += SG) {
      double acc[SG][TW];
#pragma unroll
      for (int g = 0; g < SG; ++g)
#pragma unroll
        for (int t = 0; t < TW; ++t) acc[g][t] = 0.0;
      const double* pa = xs + kq * nch + ij + 4 * sg0;                      // + 4 g: A of row group sg0 + g
      const double* pb = xs + kq * nch + 16 * wave + (lane & 15);           // + 64 t: B of tile wave + 4 t
      auto kstep = [&](bool kok) __attribute__((always_inline)) {
        double av[SG], bv[TW];
#pragma unroll
        for (int g = 0; g < SG; ++g) av[g] = (sg0 + g < nsg && kok) ? pa[4 * g] : 0.0;
#pragma unroll
        for (int t = 0; t < TW; ++t) bv[t] = (wave + 4 * t < NT && kok) ? pb[64 * t] : 0.0;
#pragma unroll
        for (int g = 0; g < SG; ++g)
#pragma unroll
          for (int t = 0; t < TW; ++t)
            if (sg0 + g < nsg && wave + 4 * t < NT) acc[g][t] = __builtin_amdgcn_mfma_f64_4x4x4f64(av[g], bv[t], acc[g][t], 0, 0, 0);   // (wave-uniform)
      };
      for (int i0 = 0; i0 < cfull; i0 += 4, pa += 4 * nch, pb += 4 * nch) kstep(true);
      if (cfull < c) kstep(cfull + kq < c);
#pragma unroll
      for (int g = 0; g < SG; ++g)
#pragma unroll
        for (int t = 0; t < TW; ++t) {
          const int sig = 4 * (sg0 + g) + (lane >> 4), rho = 16 * (wave + 4 * t) + (lane & 15);
          if (sg0 + g < nsg && wave + 4 * t < NT && sig < nch && rho < r) ctab[rho * nch + sig] = acc[g][t];
        }
    }
  }
  // The chains: chain (rho0, b), rho0 = d nch + a, runs through the entries (sig, rho) = (b + k nch, rho0 + k nch), rho < r.
  // The lanes of a wave take consecutive rho0 of one b: at a given step they read broadcast values x[sig], x[sig + c nch] and
  // consecutive x[rho], x[rho + c nch], and their 16 stores into a tile of the output fall into four cache lines.  A thread
  // takes the chains rho0 = t and r - 1 - t of its b: Ln + 1 entries whatever t is.
  // Measured per 4096 instances of ~136 rows (SISO / 3x2 / 2x1 / 4x4 plants; phases knocked out one at a time): the launch
  // 197 / 307 / 199 / 327 us, of which the lag sums 63 / 66 / 46 / 124, the chains 90 / 150 / 86 / 133 (their stores 22 / 78 / 24 /
  // 58), staging and launch 44 / 91 / 67 / 70 -- no single bound; the bytes alone are 84 us.  Variants that lost: one chain
  // per thread (idle threads), tile rows gathered in LDS and written as whole lines (300 - 330 us: more instructions than
  // the write transactions they save), a row-major output (the four 8-byte loads per tile it needs in the cold-solve kernel
  // cost that kernel 116 B of scratch), 16x16x4 MFMAs for the lag sums (2 of 16 output rows used for a SISO plant).
  // Offset of entry (sig, rho) in the output = offA[rho] + offB[sig]: tile (rho / 16, sig / 16) at (I (I + 1) / 2 + J) * 256,
  // inside it double 4 * (16 * (sig % 4) + rho % 16) + (sig % 16) / 4.
  int* offA = reinterpret_cast<int*>(ctab + ((r * nch + 1) & ~1));
  int* offB = offA + 16 * NT;
  for (int i = tid; i < 16 * NT; i += 256) {
    const int T = i >> 4, l = i & 15;
    offA[i] = (T * (T + 1) / 2) * 256 + 4 * l;
    offB[i] = T * 256 + ((l & 3) << 6) + (l >> 2);
  }
  __syncthreads();
  const int cn = c * nch, half = (r + 1) >> 1;
  double* G = gpre + b * gstride;
  for (int e = tid; e < half * nch; e += 256) {
    const int bb = e / half, t = e - bb * half;
#pragma unroll 1
    for (int part = 0; part < 2; ++part) {
      int rho = part == 0 ? t : r - 1 - t;
      if (part == 1 && rho == t) break;                                     // (odd r: the middle chain once)
      int sig = bb;
      if (rho < sig) continue;                                              // (lag 0, a < b: the pair (b, a) covers it)
      double s = ctab[rho * nch + bb];
      if (rho - sig >= 16) {                                                // never in a diagonal tile
        for (; rho < r; rho += nch, sig += nch) {
          G[offA[rho] + offB[sig]] = s;
          s += xs[rho + cn] * xs[sig + cn] - xs[rho] * xs[sig];
        }
      } else {
        for (; rho < r; rho += nch, sig += nch) {
          G[offA[rho] + offB[sig]] = s;
          if ((rho >> 4) == (sig >> 4) && rho != sig) G[offA[sig] + offB[rho]] = s;      // diagonal tile: both sides
          s += xs[rho + cn] * xs[sig + cn] - xs[rho] * xs[sig];
        }
      }
    }
  }
}

// flags of the probe solve (their own buffer, same stamp) joined into the flags of the factor-export solve; word `batch` of
// both buffers is the largest stamp that flagged anything
__global__ void ddmpc_or_flags_kernel(long long batch, int epoch, const int* __restrict__ probe, int* __restrict__ flags) {
  const long long b = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (b >= batch || probe[b] != epoch) return;
  flags[b] = epoch;
  atomicMax(flags + batch, epoch);
}
// Trajectories beyond the LDS (KParams::stage_xs = 0): the plain kernel cannot run its exact residual check, it flags every
// instance its a-priori residual bound does not dismiss.  Here the check is made up for with the exact-Hankel residual formed by a
// streaming launch (H (H' beta) from global memory: `zp`, RR2-style partial sums [batch][ng][VL], component order):
//   res = |t - (H (H' beta) + lam D beta)|_inf / |t|_inf <= refine_res   clears the flag;
// what stays flagged is solved again by the refining variant (windowed trajectory, ddmpc_cold2.hpp).  One workgroup per
// instance; only flagged instances are looked at.
__global__ __launch_bounds__(256) void ddmpc_flag_inaccurate_kernel(KParams P, int RPs, int epoch, int* __restrict__ flags,
                                                                    const double* __restrict__ u_past, const double* __restrict__ y_past,
                                                                    const double* __restrict__ beta, const signed char* __restrict__ act,
                                                                    const double* __restrict__ zp, int ng, int VL, int* __restrict__ status) {
  const long long b = blockIdx.x;
  if (flags[b] != epoch) return;                                           // (workgroup-uniform)
  if (status[b] != 0) {                                                    // (nothing to refine: the solve itself failed)
    if (threadIdx.x == 0) flags[b] = 0;
    return;
  }
  __shared__ double red[8];
  const int tid = threadIdx.x, n = P.npu / P.m;
  const double* up = u_past + b * (long long)P.npu;
  const double* yp = y_past + b * (long long)(n * P.p);
  double rmx = 0.0, tmx = 0.0;
  for (int rho = tid; rho < P.r; rho += blockDim.x) {
    const int pidx = P.tabi[1 * RPs + rho];
    const int a = act[b * (long long)P.rE + rho];
    const double t = ((pidx >= 0) ? ((pidx < P.npu) ? up[pidx] : yp[pidx - P.npu]) : P.tabd[2 * RPs + rho]) + a * P.bound;
    const double D = a ? P.tabd[1 * RPs + rho] : P.tabd[0 * RPs + rho];
    double hz = 0.0;
    for (int g = 0; g < ng; ++g) hz += zp[(b * ng + g) * (long long)VL + rho];
    const double rv = t - hz - P.lam * D * beta[b * (long long)P.rE + rho];
    rmx = fmax(rmx, (rv == rv) ? fabs(rv) : 1e300);
    tmx = fmax(tmx, fabs(t));
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { rmx = fmax(rmx, __shfl_xor(rmx, off, 64)); tmx = fmax(tmx, __shfl_xor(tmx, off, 64)); }
  if ((tid & 63) == 0) { red[tid >> 6] = rmx; red[4 + (tid >> 6)] = tmx; }
  __syncthreads();
  if (tid == 0) {
    double r_ = 0.0, t_ = 0.0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) { r_ = fmax(r_, red[w]); t_ = fmax(t_, red[4 + w]); }
    if (r_ / fmax(t_, 1e-300) <= P.refine_res) flags[b] = 0;
  }
}
__global__ void ddmpc_gain_column_kernel(long long batch, int r, int rE, int nrhs, int j, const double* __restrict__ beta,
                                         double* __restrict__ gain, const int* __restrict__ flags, int epoch) {
  const long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (idx >= batch * r) return;
  const long long b = idx / r;
  if (flags != nullptr && flags[b] != epoch) return;      // AUTO: only the instances the factor-export launch flagged
  const int rho = (int)(idx - b * r);
  const double v = beta[b * rE + rho];
  double* g0 = gain + (b * nrhs) * (long long)r + rho;
  if (j == 0) *g0 = v; else g0[(long long)j * r] = v - *g0;
}

// Output stage shared by the warm kernels: z, cost contribution and optimal_u of one component
// (same formulas as the cold kernel's output stage, active set empty).
__device__ __forceinline__ double warm_component(const KParams& P, int RPs, int rho, double beta, const double* pv,
                                                 const double* bvec, double* z_out) {
  const int kind = P.tabi[0 * RPs + rho];
  const int pidx = P.tabi[1 * RPs + rho];
  const double D = P.tabd[0 * RPs + rho];
  const double tb = P.tabd[2 * RPs + rho];
  const double wq = P.tabd[3 * RPs + rho];
  const double t = (pidx >= 0) ? pv[pidx] : tb;
  double z = t - P.lam * D * beta;
  if (P.dense_w) {                                  // dense weighting matrices: z = t - lam (W^-1 beta)
    const double* dr = P.dmat + (long long)rho * RPs;
    double sdb = 0.0;
    for (int j = 0; j < P.r; ++j) sdb += dr[j] * bvec[j];
    z = t - P.lam * (D * beta + sdb);
  }
  double contrib = P.lam * beta * z;
  if (P.dense_w && (kind == K_UFREE || kind == K_YFREE || kind == K_WPRED)) contrib -= P.lam * beta * (z - tb);
  else
  if (kind == K_UFREE || kind == K_YFREE) { const double dlt = z - tb; contrib += wq * dlt * dlt; }
  else if (kind == K_WINT) { const double sg = z - t; contrib += P.lamb_sigma * sg * sg; }
  else if (kind == K_WTERM) { const double sg = z - tb; contrib += P.lamb_sigma * sg * sg; }
  else if (kind == K_WPRED) {
    const double sg = -P.lam * beta / P.lamb_sigma;
    const double dlt = z - sg - tb;
    contrib += wq * dlt * dlt + P.lamb_sigma * sg * sg;
  }
  *z_out = z;
  return contrib;
}

constexpr int WARM_MAX_NF = 256;
constexpr int WARM_MAX_R = 288;     // >= 16 * 17 + a little: rows of the largest cold-solve instance

// One warm step for the batch: grid = batch, block = r rounded up to 64.
// Replaces update_and_solve_data_driven_mpc (controller.py:389-407) once the data are fixed.
__global__ void ddmpc_warm_step_kernel(KParams P, int RPs, int nf, const double* __restrict__ gain,
                                       const int* __restrict__ prep_status, const double* __restrict__ u_past,
                                       const double* __restrict__ y_past, double* __restrict__ u_opt,
                                       double* __restrict__ cost, int* __restrict__ status, int* __restrict__ iters,
                                       double* __restrict__ beta_ws, signed char* __restrict__ act_ws,
                                       int* __restrict__ need_cold) {
  __shared__ double pv[WARM_MAX_NF];
  __shared__ double red[32];
  __shared__ double bsh[WARM_MAX_R];
  __shared__ int viol;
  const long long b = blockIdx.x;
  const int tid = threadIdx.x, r = P.r, nrhs = nf + 1;
  const int nyp = nf - P.npu;
  for (int f = tid; f < nf; f += blockDim.x)
    pv[f] = (f < P.npu) ? u_past[b * P.npu + f] : y_past[b * nyp + (f - P.npu)];
  __syncthreads();
  double part = 0.0;
  bool finite = true;
  const double* g = gain + b * (long long)nrhs * r;
  for (int rho = tid; rho < r; rho += blockDim.x) {
    // all loads of a chunk of 8 columns are issued before they are consumed (HBM-bound kernel: keep bytes in flight)
    double beta = g[rho];
    const double* gc = g + r + rho;
    int f = 0;
    for (; f + 8 <= nf; f += 8) {
      double v[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] = gc[(long long)(f + q) * r];
#pragma unroll
      for (int q = 0; q < 8; ++q) beta += pv[f + q] * v[q];
    }
    for (; f < nf; ++f) beta += pv[f] * gc[(long long)f * r];
    bsh[rho] = beta;
  }
  if (tid == 0) viol = 0;
  __syncthreads();
  if (need_cold != nullptr) {
    // slack box (controller.py:659,674): the affine law is the first primal-dual active-set iterate (empty
    // active set).  It is optimal iff no boxed sigma leaves the box; otherwise the instance is handed to
    // the cold kernel, which runs the full active-set iteration.
    const double scale = -P.lam / P.lamb_sigma;
    for (int rho = tid; rho < r; rho += blockDim.x) {
      const int kind = P.tabi[0 * RPs + rho];
      if ((kind == K_WPRED || kind == K_WTERM) && fabs(scale * bsh[rho]) > P.bound) viol = 1;
    }
    __syncthreads();
    if (tid == 0) need_cold[b] = viol;
    if (viol) return;                                 // uniform: the cold kernel produces this instance's outputs
  }
  for (int rho = tid; rho < r; rho += blockDim.x) {
    const double beta = bsh[rho];
    double z;
    part += warm_component(P, RPs, rho, beta, pv, bsh, &z);
    finite = finite && (fabs(beta) < 1e300);
    const int oidx = P.tabi[2 * RPs + rho];
    if (oidx >= 0) u_opt[b * (long long)((P.Ln - P.npu / P.m) * P.m) + oidx] = z;
    if (beta_ws) beta_ws[b * (long long)P.rE + rho] = beta;
    if (act_ws) act_ws[b * (long long)P.rE + rho] = 0;
  }
  part = wave_sum(part);
  const unsigned long long okmask = __ballot(finite);
  if ((tid & 63) == 0) { red[tid >> 6] = part; red[16 + (tid >> 6)] = (okmask == ~0ull) ? 0.0 : 1.0; }
  __syncthreads();
  if (tid == 0) {
    double tot = 0.0, bad = 0.0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) { tot += red[w]; bad += red[16 + w]; }
    int st = prep_status[b];
    if (bad != 0.0 || !(fabs(tot) < 1e300)) st = 4;
    cost[b] = tot;
    status[b] = st;
    if (iters) iters[b] = 1;
  }
}

// Whole closed loop of one instance in one workgroup (warm path): per solve the affine law gives
// optimal_u, then the plant/FIFO steps of ddmpc_plant_kernel.  Same loop order as
// utilities/controller/controller_operation.py:263-305.
__global__ void ddmpc_closed_loop_warm_kernel(KParams P, int RPs, int nf, const double* __restrict__ gain,
                                              const int* __restrict__ prep_status, int ns, const double* __restrict__ pl,
                                              int n_steps, int n_mpc_step, double* __restrict__ x,
                                              double* __restrict__ u_past, double* __restrict__ y_past,
                                              const double* __restrict__ w, double* __restrict__ u_sys,
                                              double* __restrict__ y_sys, int* __restrict__ status_out,
                                              double* __restrict__ beta_ws, signed char* __restrict__ act_ws) {
  __shared__ double pv[WARM_MAX_NF];
  __shared__ double uo[WARM_MAX_NF];      // the first n_mpc_step*m entries of optimal_u
  __shared__ double xs[16];
  __shared__ double bsh[WARM_MAX_R];
  const long long b = blockIdx.x;
  const int tid = threadIdx.x, r = P.r, nrhs = nf + 1, m = P.m, p = P.p;
  const int n = P.npu / m, nyp = nf - P.npu;
  const double* A = pl;
  const double* Bm = A + ns * ns;
  const double* C = Bm + ns * m;
  const double* Dm = C + p * ns;
  for (int f = tid; f < nf; f += blockDim.x)
    pv[f] = (f < P.npu) ? u_past[b * P.npu + f] : y_past[b * nyp + (f - P.npu)];
  if (tid < ns) xs[tid] = x[b * ns + tid];
  const int st = prep_status[b];
  const double nanv = __longlong_as_double(0x7ff8000000000000LL);
  const double* g = gain + b * (long long)nrhs * r;
  const int nuse = n_mpc_step * m;
  double* up = pv;
  double* yp = pv + P.npu;
  for (int t0 = 0; t0 < n_steps; t0 += n_mpc_step) {
    __syncthreads();
    const bool last = (t0 + n_mpc_step >= n_steps);
    const bool all_rows = last || P.dense_w;          // dense weights: z needs the whole beta vector
    for (int rho = tid; rho < r; rho += blockDim.x) {
      const int oidx = P.tabi[2 * RPs + rho];
      if ((oidx >= 0 && oidx < nuse) || all_rows) {
        double beta = g[rho];
        for (int f = 0; f < nf; ++f) beta += pv[f] * g[(long long)(1 + f) * r + rho];
        bsh[rho] = beta;
      }
    }
    __syncthreads();
    for (int rho = tid; rho < r; rho += blockDim.x) {
      const int oidx = P.tabi[2 * RPs + rho];
      if (oidx >= 0 && oidx < nuse) {
        double z;
        (void)warm_component(P, RPs, rho, bsh[rho], pv, bsh, &z);
        uo[oidx] = z;
      }
      if (last && beta_ws) { beta_ws[b * (long long)P.rE + rho] = bsh[rho]; act_ws[b * (long long)P.rE + rho] = 0; }
    }
    __syncthreads();
    if (tid == 0) {
      const int nsub = (t0 + n_mpc_step <= n_steps) ? n_mpc_step : n_steps - t0;
      for (int j = 0; j < nsub; ++j) {
        const int k = t0 + j;
        double* us = u_sys + (b * n_steps + k) * m;
        double* ys = y_sys + (b * n_steps + k) * p;
        if (st > 1) {
          for (int i = 0; i < m; ++i) us[i] = nanv;
          for (int i = 0; i < p; ++i) ys[i] = nanv;
          continue;
        }
        const double* uk = uo + j * m;
        const double* wk = w + (b * n_steps + k) * p;
        double yv[16], xn[16];
        for (int i = 0; i < p; ++i) {              // y = C x + D u + w with the state BEFORE the update
          double s = wk[i];
          for (int q = 0; q < ns; ++q) s += C[i * ns + q] * xs[q];
          for (int q = 0; q < m; ++q) s += Dm[i * m + q] * uk[q];
          yv[i] = s;
          ys[i] = s;
        }
        for (int i = 0; i < ns; ++i) {
          double s = 0.0;
          for (int q = 0; q < ns; ++q) s += A[i * ns + q] * xs[q];
          for (int q = 0; q < m; ++q) s += Bm[i * m + q] * uk[q];
          xn[i] = s;
        }
        for (int i = 0; i < ns; ++i) xs[i] = xn[i];
        for (int i = 0; i < m; ++i) us[i] = uk[i];
        for (int i = 0; i < (n - 1) * m; ++i) up[i] = up[i + m];       // FIFO shift
        for (int i = 0; i < m; ++i) up[(n - 1) * m + i] = uk[i];
        for (int i = 0; i < (n - 1) * p; ++i) yp[i] = yp[i + p];
        for (int i = 0; i < p; ++i) yp[(n - 1) * p + i] = yv[i];
      }
    }
  }
  __syncthreads();
  for (int f = tid; f < nf; f += blockDim.x) {
    if (f < P.npu) u_past[b * P.npu + f] = pv[f]; else y_past[b * nyp + (f - P.npu)] = pv[f];
  }
  if (tid < ns) x[b * ns + tid] = xs[tid];
  if (tid == 0) status_out[b] = st;
}
// --------------------------------------------------------------------------
// Persistent-excitation guard (controller.py:275-296, hankel_matrix.py:55-87) for a batch.
// The reference tests rank(H_order(u_d)) == m*order with an SVD.  Here one workgroup per instance
// forms G = H H' (r = m*order rows), factors G = L L' and inverts L in place, all in LDS, and
// returns a rigorous LOWER bound of sigma_min/sigma_max of H:
//     sigma_min^2 = lambda_min(G) >= 1 / trace(G^-1) = 1 / |L^-1|_F^2,   sigma_max^2 <= trace(G).
// A bound above a threshold far from both the SVD tolerance (max(M,N)*eps) and the rounding floor of
// the Gram (~sqrt(r*c*eps)) certifies full rank; anything else (incl. a failed pivot) reports 0 and
// is left to the exact SVD test on the host.  Packed lower storage: (i,j) at i(i+1)/2 + j.
// --------------------------------------------------------------------------
__global__ void ddmpc_pe_guard_kernel(const double* __restrict__ X, int N, int m, int order,
                                      double* __restrict__ ratio_lb, double* scratch, long long scratch_stride) {
  extern __shared__ __attribute__((aligned(16))) double sm_lds[];
  const long long b = blockIdx.x;
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int r = m * order, c = N - order + 1;
  const int npk = r * (r + 1) / 2;
  // the packed matrix lives in LDS when it fits, else in a per-instance slice of a global workspace
  double* sm = scratch ? scratch + b * scratch_stride : sm_lds;
  double* G = sm;                 // packed lower, r(r+1)/2
  double* col = sm + npk;         // r
  double* red = col + r;          // 64
  __shared__ int bad;
  const double* x = X + b * (long long)N * m;   // H[i][t] = x[t*m + i]  (i = k*m + ch)
  if (tid == 0) bad = 0;
  // ---- Gram: one entry per thread-iteration, operands straight from global/L2 (setup-time kernel)
  for (int e = tid; e < npk; e += nthr) {
    int i = (int)((sqrt(8.0 * e + 1.0) - 1.0) * 0.5);
    while ((i + 1) * (i + 2) / 2 <= e) ++i;
    while (i * (i + 1) / 2 > e) --i;
    const int j = e - i * (i + 1) / 2;
    const double* xi = x + i;
    const double* xj = x + j;
    double s0 = 0.0, s1 = 0.0;
    int t = 0;
    for (; t + 1 < c; t += 2) { s0 += xi[t * m] * xj[t * m]; s1 += xi[(t + 1) * m] * xj[(t + 1) * m]; }
    if (t < c) s0 += xi[t * m] * xj[t * m];
    G[e] = s0 + s1;
  }
  __syncthreads();
  double tr = 0.0;
  for (int i = tid; i < r; i += nthr) tr += G[i * (i + 1) / 2 + i];
  tr = wave_sum(tr);
  if ((tid & 63) == 0) red[tid >> 6] = tr;
  __syncthreads();
  double trace = 0.0;
  for (int w = 0; w < (nthr >> 6); ++w) trace += red[w];
  __syncthreads();
  // ---- right-looking Cholesky, column by column
  for (int k = 0; k < r; ++k) {
    const double dk = G[k * (k + 1) / 2 + k];
    if (!(dk > 1e-13 * trace)) { if (tid == 0) bad = 1; break; }     // uniform: every thread reads the same dk
    const double inv = 1.0 / sqrt(dk);
    for (int i = k + tid; i < r; i += nthr) col[i] = G[i * (i + 1) / 2 + k] * inv;
    __syncthreads();
    for (int i = k + tid; i < r; i += nthr) G[i * (i + 1) / 2 + k] = col[i];
    // trailing update of the packed lower triangle: entry (i,j), k < j <= i
    const int nt = r - k - 1;
    const int ne = nt * (nt + 1) / 2;
    for (int e = tid; e < ne; e += nthr) {
      int ii = (int)((sqrt(8.0 * e + 1.0) - 1.0) * 0.5);
      while ((ii + 1) * (ii + 2) / 2 <= e) ++ii;
      while (ii * (ii + 1) / 2 > e) --ii;
      const int jj = e - ii * (ii + 1) / 2;
      const int i = k + 1 + ii, j = k + 1 + jj;
      G[i * (i + 1) / 2 + j] -= col[i] * col[j];
    }
    __syncthreads();
  }
  __syncthreads();
  if (bad) { if (tid == 0) ratio_lb[b] = 0.0; return; }
  // ---- in-place inverse of L (unblocked, last column first): T = L^-1,
  //      T[j][j] = 1/L[j][j],  T[i][j] = -T[j][j] * sum_{k=j+1..i} T[i][k] L[k][j]
  for (int j = r - 1; j >= 0; --j) {
    const double tjj = 1.0 / G[j * (j + 1) / 2 + j];
    for (int i = j + 1 + tid; i < r; i += nthr) col[i] = G[i * (i + 1) / 2 + j];
    __syncthreads();
    for (int i = j + 1 + tid; i < r; i += nthr) {
      const double* Ti = G + i * (i + 1) / 2;
      double s = 0.0;
      for (int k = j + 1; k <= i; ++k) s += Ti[k] * col[k];
      G[i * (i + 1) / 2 + j] = -tjj * s;
    }
    if (tid == 0) G[j * (j + 1) / 2 + j] = tjj;
    __syncthreads();
  }
  double fs = 0.0;
  for (int e = tid; e < npk; e += nthr) fs += G[e] * G[e];
  fs = wave_sum(fs);
  if ((tid & 63) == 0) red[tid >> 6] = fs;
  __syncthreads();
  if (tid == 0) {
    double f2 = 0.0;
    for (int w = 0; w < (nthr >> 6); ++w) f2 += red[w];
    const double lb = 1.0 / (f2 * trace);                 // lambda_min_lb / lambda_max_ub
    ratio_lb[b] = (lb > 0.0 && lb < 1e300) ? sqrt(lb) : 0.0;
  }
}

}  // namespace ddmpc
