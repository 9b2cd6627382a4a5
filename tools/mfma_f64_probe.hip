// Probe: v_mfma_f64_16x16x4_f64 operand/accumulator lane layout and issue rate on gfx950.
// Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_f64_probe.hip -o tools/mfma_f64_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>
typedef double d4 __attribute__((ext_vector_type(4)));
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)

__global__ void layout_kernel(const double* A /*16x4 row-major*/, const double* B /*4x16 row-major*/, double* D /*16x16*/) {
  int l = threadIdx.x;
  double a = A[(l & 15) * 4 + (l >> 4)];      // hypothesis: A[i=l&15][k=l>>4]
  double b = B[(l >> 4) * 16 + (l & 15)];     // hypothesis: B[k=l>>4][j=l&15]
  d4 acc = {0, 0, 0, 0};
  acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
  for (int j = 0; j < 4; j++) D[((l >> 4) + 4 * j) * 16 + (l & 15)] = acc[j];   // hypothesis: row=(l>>4)+4j, col=l&15
}

template <int NACC>
__global__ void rate_kernel(double* out, int iters, long long* cyc) {
  d4 acc[NACC];
  for (int i = 0; i < NACC; i++) acc[i] = d4{0, 0, 0, 0};
  double a = 1.0 + threadIdx.x * 1e-3, b = 0.5 - threadIdx.x * 1e-3;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < NACC; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

__global__ void fma_rate_kernel(double* out, int iters, long long* cyc) {
  double x[16];
  for (int i = 0; i < 16; i++) x[i] = threadIdx.x * 1e-3 + i;
  double a = 1.0000001, b = 1e-9;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 16; i++) x[i] = fma(x[i], a, b);
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  double s = 0;
  for (int i = 0; i < 16; i++) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

int main() {
  // ---- layout
  std::vector<double> A(64), B(64), D(256), Dref(256, 0.0);
  for (int i = 0; i < 16; i++) for (int k = 0; k < 4; k++) A[i * 4 + k] = 1 + i * 7 + k * 3;
  for (int k = 0; k < 4; k++) for (int j = 0; j < 16; j++) B[k * 16 + j] = 2 + k * 11 + j * 5 + (j * j % 7);
  for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++) for (int k = 0; k < 4; k++) Dref[i * 16 + j] += A[i * 4 + k] * B[k * 16 + j];
  double *dA, *dB, *dD; long long* dC;
  CK(hipMalloc(&dA, 64 * 8)); CK(hipMalloc(&dB, 64 * 8)); CK(hipMalloc(&dD, 256 * 8)); CK(hipMalloc(&dC, 8));
  CK(hipMemcpy(dA, A.data(), 64 * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), 64 * 8, hipMemcpyHostToDevice));
  layout_kernel<<<1, 64>>>(dA, dB, dD);
  CK(hipMemcpy(D.data(), dD, 256 * 8, hipMemcpyDeviceToHost));
  int bad = 0; for (int i = 0; i < 256; i++) if (D[i] != Dref[i]) bad++;
  printf("LAYOUT mismatches=%d (0 means A[l&15][l>>4], B[l>>4][l&15], D row=(l>>4)+4j col=l&15)\n", bad);
  // ---- rates
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  printf("device %s CUs=%d clock=%d kHz\n", prop.name, prop.multiProcessorCount, prop.clockRate);
  double* dout; CK(hipMalloc(&dout, 1024 * 256 * 64 * 8));
  long long cyc; int iters = 2000;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); float ms;
#define RUN(NACC, BLK, THR) \
  rate_kernel<NACC><<<BLK, THR>>>(dout, 10, dC); CK(hipDeviceSynchronize()); \
  CK(hipEventRecord(e0)); rate_kernel<NACC><<<BLK, THR>>>(dout, iters, dC); CK(hipEventRecord(e1)); CK(hipDeviceSynchronize()); \
  CK(hipEventElapsedTime(&ms, e0, e1)); CK(hipMemcpy(&cyc, dC, 8, hipMemcpyDeviceToHost)); \
  printf("MFMA f64 16x16x4: nacc=%d blocks=%d thr=%d  memtime-ticks/mfma(wave0)=%.2f  wall %.3f ms  => %.2f TFLOP/s\n", NACC, BLK, THR, \
         (double)cyc / ((double)iters * NACC), ms, 2048.0 * iters * NACC * (double)BLK * (THR / 64) / (ms * 1e-3) / 1e12);
  RUN(1, 1, 64) RUN(4, 1, 64) RUN(8, 1, 64)
  RUN(4, 256, 256) RUN(8, 256, 256) RUN(4, 1024, 256) RUN(8, 2048, 256) RUN(4, 2048, 512)
  fma_rate_kernel<<<2048, 256>>>(dout, 10, dC); CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0)); fma_rate_kernel<<<2048, 256>>>(dout, iters, dC); CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
  CK(hipEventElapsedTime(&ms, e0, e1)); CK(hipMemcpy(&cyc, dC, 8, hipMemcpyDeviceToHost));
  printf("VALU f64 fma: ticks/fma(wave0)=%.2f wall %.3f ms => %.2f TFLOP/s\n", (double)cyc / (iters * 16.0), ms,
         2.0 * 64 * 16 * iters * 2048.0 * 4 / (ms * 1e-3) / 1e12);
  return bad != 0;
}
