/* A plain C99 caller of the C ABI (include/ddmpc.h), the way a non-Python host would use the engine:
 * reads a batch of trajectories and past windows from a binary file written by the test, creates a
 * robust four-tank-shaped controller batch, runs ddmpc_solve and ddmpc_step from HOST buffers, reads
 * `sigma` back, and writes the results to a binary file the test compares with the Python layer's.
 *
 *   capi_caller <in.bin> <out.bin>
 * in.bin : int32 B, N, m, p, n, L, slack; then doubles u_d[B*N*m], y_d[B*N*p], u_past[B*n*m], y_past[B*n*p]
 * out.bin: doubles u_opt[B*L*m], cost[B], u_step[B*L*m], cost_step[B], sigma[B*(L+n)*p]; int32 status[B], iters[B]
 */
#include <stdio.h>
#include <stdlib.h>
#include "ddmpc.h"

#define CHECK(call)                                                                  \
  do {                                                                               \
    int rc__ = (call);                                                               \
    if (rc__ != DDMPC_OK) {                                                          \
      fprintf(stderr, "%s -> %d: %s\n", #call, rc__, ddmpc_last_error());            \
      return 2;                                                                      \
    }                                                                                \
  } while (0)

static double* read_doubles(FILE* f, size_t n) {
  double* p = (double*)malloc(n * sizeof(double));
  if (!p || fread(p, sizeof(double), n, f) != n) { fprintf(stderr, "short read\n"); exit(3); }
  return p;
}

int main(int argc, char** argv) {
  if (argc != 3) return 1;
  FILE* f = fopen(argv[1], "rb");
  if (!f) return 1;
  int32_t hdr[7];
  if (fread(hdr, sizeof(int32_t), 7, f) != 7) return 3;
  const int B = hdr[0], N = hdr[1], m = hdr[2], p = hdr[3], n = hdr[4], L = hdr[5], slack = hdr[6];
  double* u_d = read_doubles(f, (size_t)B * N * m);
  double* y_d = read_doubles(f, (size_t)B * N * p);
  double* u_past = read_doubles(f, (size_t)B * n * m);
  double* y_past = read_doubles(f, (size_t)B * n * p);
  fclose(f);

  if (ddmpc_version() != DDMPC_ABI_VERSION) return 4;
  if (ddmpc_device_count() <= 0) { fprintf(stderr, "no HIP device\n"); return 5; }

  const double Q = 3.0, R = 1e-4, u_s[2] = {1.0, 1.0}, y_s[2] = {0.65, 0.77};
  ddmpc_params prm;
  prm.struct_size = (int32_t)sizeof(prm);
  prm.m = m; prm.p = p; prm.n = n; prm.L = L; prm.N = N;
  prm.controller_type = DDMPC_ROBUST;
  prm.slack_type = slack;
  prm.use_terminal_constraint = 1;
  prm.weight_kind = DDMPC_WEIGHT_SCALAR;
  prm.Q = &Q; prm.R = &R;
  prm.eps_max = 0.002; prm.lamb_alpha = 50.0; prm.lamb_sigma = 1000.0; prm.c = 1.0;
  prm.u_s = u_s; prm.y_s = y_s;
  prm.max_iter = 0;
  prm.gram_mode = DDMPC_GRAM_AUTO;

  ddmpc_handle* h = NULL;
  CHECK(ddmpc_create(&prm, B, 0, &h));
  CHECK(ddmpc_set_data(h, u_d, y_d, DDMPC_MEM_HOST));
  const size_t nu = (size_t)B * L * m, ns = (size_t)B * (L + n) * p;
  double* u_opt = (double*)malloc(nu * sizeof(double));
  double* u_stp = (double*)malloc(nu * sizeof(double));
  double* cost = (double*)malloc(B * sizeof(double));
  double* cost_stp = (double*)malloc(B * sizeof(double));
  double* sigma = (double*)malloc(ns * sizeof(double));
  int32_t* status = (int32_t*)malloc(B * sizeof(int32_t));
  int32_t* iters = (int32_t*)malloc(B * sizeof(int32_t));
  int32_t* status2 = (int32_t*)malloc(B * sizeof(int32_t));
  CHECK(ddmpc_solve(h, u_past, y_past, u_opt, cost, status, iters, DDMPC_MEM_HOST));
  CHECK(ddmpc_get_solution(h, DDMPC_SOL_SIGMA, sigma, DDMPC_MEM_HOST));
  CHECK(ddmpc_prepare(h));
  CHECK(ddmpc_step(h, u_past, y_past, u_stp, cost_stp, status2, NULL, DDMPC_MEM_HOST));
  /* misuse must come back as an error code with a message, never crash */
  if (ddmpc_solve(h, NULL, y_past, u_opt, cost, status, iters, DDMPC_MEM_HOST) != DDMPC_ERR_INVALID) return 6;
  if (ddmpc_last_error()[0] == '\0') return 7;
  CHECK(ddmpc_destroy(h));

  f = fopen(argv[2], "wb");
  if (!f) return 1;
  fwrite(u_opt, sizeof(double), nu, f);
  fwrite(cost, sizeof(double), (size_t)B, f);
  fwrite(u_stp, sizeof(double), nu, f);
  fwrite(cost_stp, sizeof(double), (size_t)B, f);
  fwrite(sigma, sizeof(double), ns, f);
  fwrite(status, sizeof(int32_t), (size_t)B, f);
  fwrite(iters, sizeof(int32_t), (size_t)B, f);
  fclose(f);
  printf("capi_caller: %d instances, status[0] = %d, cost[0] = %.9f\n", B, status[0], cost[0]);
  return 0;
}
