/* ddmpc_oracle_c.c -- compiled fp64 CPU restatement of the Data-Driven MPC cold QP solve.
 *
 * TEST / MEASUREMENT INFRASTRUCTURE ONLY.  Nothing in the product package links, loads or calls
 * this file; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may.
 *
 * What it restates (citations relative to /root/reference):
 *   - Hankel matrices H_{L+n}(u_d), H_{L+n}(y_d): direct_data_driven_mpc/utilities/hankel_matrix.py:5-53,
 *     built at direct_data_driven_mpc_controller.py:376-377 (never materialised here).
 *   - The QP of direct_data_driven_mpc_controller.py:409-445 (variables), :506-547 (dynamics),
 *     :549-583 (internal state), :585-629 (terminal), :631-677 (slack box), :679-722 (cost), solved
 *     through the reduced r x r system derived in oracle/reduced_form.py (same algebra, natural stacking
 *     z = [ubar; ybar + sigma]); extraction optimal_u = ubar[n*m:] (:799-805).
 * This is SURVEY 8(d)'s "compiled fp64 CPU oracle executing the same cold definition":
 * Hankel (implicit) -> Gram -> reduced KKT -> Cholesky -> solve (-> active-set iterations if CONVEX).
 *
 * Parity status: pinned to the numpy full-space oracle (oracle/ddmpc_oracle.py) in tests/test_oracle_c.py,
 * which is itself "QP parity unpinned" against CVXPY (see that file's header and DESIGN.md section 2).
 *
 * Build: see oracle/build_c.py (gcc -O3 -march=x86-64-v3 -fopenmp -shared -fPIC).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct {
  int n, m, p, L, N;
  int robust;       /* DataDrivenMPCType.ROBUST */
  int convex;       /* SlackVarConstraintTypes.CONVEX (else NONE) */
  int tec;          /* use_terminal_constraint */
  int max_iter;
  double eps_max, lamb_alpha, lamb_sigma, c;
  const double* qdiag; /* [p*L] diagonal of Q */
  const double* rdiag; /* [m*L] diagonal of R */
  const double* u_s;   /* [m] */
  const double* y_s;   /* [p] */
} ddmpc_oracle_spec;

/* Gram G = H H' in the natural stacking (rows: ubar (k,ch) = k*m+ch, then w (k,ch) = Ln*m + k*p+ch).
 * structured = 1: Hankel sliding-window recurrence per channel pair,
 *   S_ab(k+1,l+1) = S_ab(k,l) - x_a[k] x_b[l] + x_a[c+k] x_b[c+l];   first row/column by direct sums.
 * structured = 0: plain dot products over the implicit Hankel rows (r^2 c / 2 multiply-adds). */
static void gram(const ddmpc_oracle_spec* s, const double* u_d, const double* y_d, double* G, int structured,
                 double* x /* scratch [nch*N] */) {
  const int m = s->m, p = s->p, Ln = s->L + s->n, N = s->N, nch = m + p;
  const int r = nch * Ln, c = N - Ln + 1;
  /* channel-major copy: x[a][t] */
  for (int t = 0; t < N; ++t) {
    for (int a = 0; a < m; ++a) x[(size_t)a * N + t] = u_d[(size_t)t * m + a];
    for (int a = 0; a < p; ++a) x[(size_t)(m + a) * N + t] = y_d[(size_t)t * p + a];
  }
#define ROW(k, a) ((a) < m ? (k) * m + (a) : Ln * m + (k) * p + ((a) - m))
  if (!structured) {
    for (int a = 0; a < nch; ++a)
      for (int k = 0; k < Ln; ++k) {
        const double* xa = x + (size_t)a * N + k;
        const int i = ROW(k, a);
        for (int b = 0; b < nch; ++b)
          for (int l = 0; l < Ln; ++l) {
            const int j = ROW(l, b);
            if (j > i) continue;
            const double* xb = x + (size_t)b * N + l;
            double acc = 0.0;
            for (int t = 0; t < c; ++t) acc += xa[t] * xb[t];
            G[(size_t)i * r + j] = acc;
            G[(size_t)j * r + i] = acc;
          }
      }
  } else {
    for (int a = 0; a < nch; ++a)
      for (int b = 0; b < nch; ++b) {
        const double* xa = x + (size_t)a * N;
        const double* xb = x + (size_t)b * N;
        /* first column (k,0) and first row (0,l) */
        for (int k = 0; k < Ln; ++k) {
          double acc = 0.0;
          for (int t = 0; t < c; ++t) acc += xa[t + k] * xb[t];
          G[(size_t)ROW(k, a) * r + ROW(0, b)] = acc;
        }
        for (int l = 1; l < Ln; ++l) {
          double acc = 0.0;
          for (int t = 0; t < c; ++t) acc += xa[t] * xb[t + l];
          G[(size_t)ROW(0, a) * r + ROW(l, b)] = acc;
        }
        for (int k = 1; k < Ln; ++k)
          for (int l = 1; l < Ln; ++l)
            G[(size_t)ROW(k, a) * r + ROW(l, b)] = G[(size_t)ROW(k - 1, a) * r + ROW(l - 1, b)] -
                                                   xa[k - 1] * xb[l - 1] + xa[c + k - 1] * xb[c + l - 1];
      }
  }
#undef ROW
}

/* per-component (D_ii, t_i) for the current active set: oracle/reduced_form.py:component_tables */
static void tables(const ddmpc_oracle_spec* s, const double* u_past, const double* y_past, const int* act, double* D,
                   double* t) {
  const int n = s->n, m = s->m, p = s->p, L = s->L, Ln = L + n;
  const double bound = s->c * s->eps_max;
  for (int k = 0; k < Ln; ++k) {
    const int kp = k - n;
    const int is_int = kp < 0, is_term = s->tec && kp >= L - n;
    for (int ch = 0; ch < m; ++ch) {
      const int i = k * m + ch;
      if (is_int) { D[i] = 0.0; t[i] = u_past[k * m + ch]; }
      else if (is_term) { D[i] = 0.0; t[i] = s->u_s[ch]; }
      else { D[i] = 1.0 / s->rdiag[kp * m + ch]; t[i] = s->u_s[ch]; }
    }
    for (int ch = 0; ch < p; ++ch) {
      const int i = Ln * m + k * p + ch;
      if (!s->robust) {
        if (is_int) { D[i] = 0.0; t[i] = y_past[k * p + ch]; }
        else if (is_term) { D[i] = 0.0; t[i] = s->y_s[ch]; }
        else { D[i] = 1.0 / s->qdiag[kp * p + ch]; t[i] = s->y_s[ch]; }
        continue;
      }
      const double ls = s->lamb_sigma;
      if (is_int) { D[i] = 1.0 / ls; t[i] = y_past[k * p + ch]; continue; }
      const int sa = act[kp * p + ch];
      if (is_term) {
        if (sa == 0) { D[i] = 1.0 / ls; t[i] = s->y_s[ch]; }
        else { D[i] = 0.0; t[i] = s->y_s[ch] + sa * bound; }
      } else {
        const double q = s->qdiag[kp * p + ch];
        if (sa == 0) { D[i] = 1.0 / q + 1.0 / ls; t[i] = s->y_s[ch]; }
        else { D[i] = 1.0 / q; t[i] = s->y_s[ch] + sa * bound; }
      }
    }
  }
}

/* in-place lower Cholesky (row-major, right part untouched); returns 0 on a non-positive pivot */
static int cholesky(double* K, int r) {
  for (int j = 0; j < r; ++j) {
    double* Kj = K + (size_t)j * r;
    double d = Kj[j];
    for (int k = 0; k < j; ++k) d -= Kj[k] * Kj[k];
    if (!(d > 0.0)) return 0;
    d = sqrt(d);
    Kj[j] = d;
    const double inv = 1.0 / d;
    for (int i = j + 1; i < r; ++i) {
      double* Ki = K + (size_t)i * r;
      double v = Ki[j];
      for (int k = 0; k < j; ++k) v -= Ki[k] * Kj[k];
      Ki[j] = v * inv;
    }
  }
  return 1;
}

/* status codes follow include/ddmpc.h: 0 optimal, 4 solver_error */
static void solve_one(const ddmpc_oracle_spec* s, const double* u_d, const double* y_d, const double* u_past,
                      const double* y_past, double* u_opt, double* cost, int* status, int* iters, int structured,
                      double* G, double* K, double* work) {
  const int n = s->n, m = s->m, p = s->p, L = s->L, Ln = L + n, nch = m + p, r = nch * Ln;
  const double lam = s->robust ? s->lamb_alpha * s->eps_max : 0.0;
  const int boxed = s->robust && s->convex;
  const double bound = s->c * s->eps_max;
  double* D = work;                                   /* work: [4r] + [L*p ints] + [nch*N] */
  double *t = D + r, *beta = D + 2 * r, *z = D + 3 * r;
  int* act = (int*)(work + 4 * (size_t)r);
  memset(act, 0, sizeof(int) * ((size_t)L * p + 1));
  int st = 0, it = 0;
  gram(s, u_d, y_d, G, structured, work + 4 * (size_t)r + ((size_t)L * p + 2) / 2 + 1);
  for (;;) {
    ++it;
    tables(s, u_past, y_past, act, D, t);
    for (int i = 0; i < r; ++i) {
      memcpy(K + (size_t)i * r, G + (size_t)i * r, sizeof(double) * (size_t)(i + 1));
      K[(size_t)i * r + i] += lam * D[i];
    }
    if (!cholesky(K, r)) { st = 4; break; }
    for (int i = 0; i < r; ++i) {          /* L y = t */
      double v = t[i];
      const double* Ki = K + (size_t)i * r;
      for (int k = 0; k < i; ++k) v -= Ki[k] * beta[k];
      beta[i] = v / Ki[i];
    }
    for (int i = r - 1; i >= 0; --i) {     /* L' beta = y */
      double v = beta[i];
      for (int k = i + 1; k < r; ++k) v -= K[(size_t)k * r + i] * beta[k];
      beta[i] = v / K[(size_t)i * r + i];
    }
    if (!boxed) break;
    int changed = 0;
    for (int j = 0; j < L * p; ++j) {      /* sigma[n*p:], controller.py:659,674 */
      const double sh = -lam * beta[Ln * m + n * p + j] / s->lamb_sigma;
      const int ns = sh > bound ? 1 : (sh < -bound ? -1 : 0);
      if (ns != act[j]) { act[j] = ns; changed = 1; }
    }
    if (!changed) break;
    if (it >= s->max_iter) { st = 4; break; }
  }
  double cst = 0.0;
  if (st == 0) {
    tables(s, u_past, y_past, act, D, t);
    for (int i = 0; i < r; ++i) z[i] = t[i] - lam * D[i] * beta[i];
    /* cost: control cost + lam*beta'G beta + lamb_sigma*|sigma|^2  (controller.py:708-716) */
    for (int k = 0; k < L; ++k) {
      for (int ch = 0; ch < m; ++ch) {
        const double du = z[(k + n) * m + ch] - s->u_s[ch];
        cst += s->rdiag[k * m + ch] * du * du;
        u_opt[k * m + ch] = z[(k + n) * m + ch];
      }
    }
    if (!s->robust) {
      for (int k = 0; k < L; ++k)
        for (int ch = 0; ch < p; ++ch) {
          const double dy = z[Ln * m + (k + n) * p + ch] - s->y_s[ch];
          cst += s->qdiag[k * p + ch] * dy * dy;
        }
    } else {
      /* beta' G beta = beta' (t - lam D beta - ... ) : use G directly for independence */
      double bgb = 0.0;
      for (int i = 0; i < r; ++i) {
        double gi = 0.0;
        const double* Gi = G + (size_t)i * r;
        for (int j = 0; j < r; ++j) gi += Gi[j] * beta[j];
        bgb += beta[i] * gi;
      }
      cst += lam * bgb;
      for (int k = 0; k < Ln; ++k)
        for (int ch = 0; ch < p; ++ch) {
          const int i = Ln * m + k * p + ch;
          const int kp = k - n;
          double sg;
          if (kp < 0) sg = z[i] - y_past[k * p + ch];
          else if (s->tec && kp >= L - n) sg = z[i] - s->y_s[ch];
          else {
            const int sa = act[kp * p + ch];
            sg = (boxed && sa != 0) ? sa * bound : -lam * beta[i] / s->lamb_sigma;
          }
          cst += s->lamb_sigma * sg * sg;
          if (kp >= 0) {
            const double dy = z[i] - sg - s->y_s[ch];
            cst += s->qdiag[kp * p + ch] * dy * dy;
          }
        }
    }
    if (!(fabs(cst) < 1e300)) st = 4;
  }
  *cost = cst;
  *status = st;
  if (iters) *iters = it;
}

/* Solve `batch` independent instances with `nthreads` threads (one instance per thread at a time).
 * u_d [batch,N,m], y_d [batch,N,p], u_past [batch,n*m], y_past [batch,n*p] -> u_opt [batch,L*m], cost, status, iters. */
int ddmpc_oracle_c_solve_batch(const ddmpc_oracle_spec* s, int batch, const double* u_d, const double* y_d,
                               const double* u_past, const double* y_past, double* u_opt, double* cost, int* status,
                               int* iters, int nthreads, int structured) {
  const int r = (s->m + s->p) * (s->L + s->n);
  if (s->N - s->L - s->n + 1 <= 0 || batch < 0) return -1;
  if (nthreads < 1) nthreads = 1;
#pragma omp parallel num_threads(nthreads)
  {
    const size_t nwork = 4 * (size_t)r + ((size_t)s->L * s->p + 2) / 2 + 1 + (size_t)(s->m + s->p) * s->N + 64;
    double* G = (double*)malloc(sizeof(double) * (2 * (size_t)r * r + nwork));
    double* K = G + (size_t)r * r;
    double* work = K + (size_t)r * r;
#pragma omp for schedule(dynamic, 1)
    for (int b = 0; b < batch; ++b)
      solve_one(s, u_d + (size_t)b * s->N * s->m, y_d + (size_t)b * s->N * s->p, u_past + (size_t)b * s->n * s->m,
                y_past + (size_t)b * s->n * s->p, u_opt + (size_t)b * s->L * s->m, cost + b, status + b,
                iters ? iters + b : 0, structured, G, K, work);
    free(G);
  }
  return 0;
}

int ddmpc_oracle_c_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
