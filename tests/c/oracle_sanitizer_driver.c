/* CPU-only hygiene driver (sanitizer builds of the native CPU-side code): the compiled C restatement
 * oracle/ddmpc_oracle_c.c, compiled INTO this program with -fsanitize=address,undefined, run on a batch read from a binary
 * file written by tests/test_oracle_c.py.  Test infrastructure only.
 *
 *   oracle_sanitizer_driver <in.bin> <out.bin>
 * in.bin : int32 B, N, m, p, n, L, convex, tec, structured; doubles eps_max, lamb_alpha, lamb_sigma, c; qdiag[p*L], rdiag[m*L],
 *          u_s[m], y_s[p]; u_d[B*N*m], y_d[B*N*p], u_past[B*n*m], y_past[B*n*p]
 * out.bin: doubles u_opt[B*L*m], cost[B]; int32 status[B], iters[B]
 */
#include <stdint.h>
#include <stdio.h>
#include "../../oracle/ddmpc_oracle_c.c"

static double* rd(FILE* f, size_t n) {
  double* p = (double*)malloc((n ? n : 1) * sizeof(double));
  if (!p || fread(p, sizeof(double), n, f) != n) { fprintf(stderr, "short read\n"); exit(3); }
  return p;
}

int main(int argc, char** argv) {
  if (argc != 3) return 1;
  FILE* f = fopen(argv[1], "rb");
  if (!f) return 1;
  int32_t h[9];
  double s4[4];
  if (fread(h, sizeof(int32_t), 9, f) != 9 || fread(s4, sizeof(double), 4, f) != 4) return 3;
  const int B = h[0], N = h[1], m = h[2], p = h[3], n = h[4], L = h[5];
  double* qd = rd(f, (size_t)p * L); double* rdg = rd(f, (size_t)m * L);
  double* us = rd(f, (size_t)m); double* ys = rd(f, (size_t)p);
  double* u_d = rd(f, (size_t)B * N * m); double* y_d = rd(f, (size_t)B * N * p);
  double* up = rd(f, (size_t)B * n * m); double* yp = rd(f, (size_t)B * n * p);
  fclose(f);
  ddmpc_oracle_spec s;
  s.n = n; s.m = m; s.p = p; s.L = L; s.N = N; s.robust = 1; s.convex = h[6]; s.tec = h[7]; s.max_iter = 50;
  s.eps_max = s4[0]; s.lamb_alpha = s4[1]; s.lamb_sigma = s4[2]; s.c = s4[3];
  s.qdiag = qd; s.rdiag = rdg; s.u_s = us; s.y_s = ys;
  double* uo = (double*)malloc((size_t)B * L * m * sizeof(double));
  double* co = (double*)malloc((size_t)B * sizeof(double));
  int* st = (int*)malloc((size_t)B * sizeof(int));
  int* it = (int*)malloc((size_t)B * sizeof(int));
  const int rc = ddmpc_oracle_c_solve_batch(&s, B, u_d, y_d, up, yp, uo, co, st, it, 1, h[8]);
  if (rc != 0) { fprintf(stderr, "solve_batch -> %d\n", rc); return 2; }
  FILE* g = fopen(argv[2], "wb");
  if (!g) return 1;
  fwrite(uo, sizeof(double), (size_t)B * L * m, g); fwrite(co, sizeof(double), (size_t)B, g);
  fwrite(st, sizeof(int), (size_t)B, g); fwrite(it, sizeof(int), (size_t)B, g);
  fclose(g);
  free(qd); free(rdg); free(us); free(ys); free(u_d); free(y_d); free(up); free(yp); free(uo); free(co); free(st); free(it);
  return 0;
}
