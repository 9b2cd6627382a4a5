// Probe: v_mfma_f64_4x4x4_4b_f64 layout + issue rate (is it 4x cheaper than 16x16x4?)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)
__global__ void layout(const double* A, const double* B, double* D) {   // A,B: 64 values each (per lane), D: 64
  int l = threadIdx.x;
  double acc = 0.0;
  acc = __builtin_amdgcn_mfma_f64_4x4x4f64(A[l], B[l], acc, 0, 0, 0);
  D[l] = acc;
}
template <int NACC> __global__ void rate(double* out, int iters, long long* cyc) {
  double acc[NACC];
  for (int i = 0; i < NACC; i++) acc[i] = 0;
  double a = 1.0 + threadIdx.x * 1e-3, b = 0.5 - threadIdx.x * 1e-3;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0; for (int i = 0; i < NACC; i++) s += acc[i];
  long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
int main() {
  std::vector<double> A(64), B(64), D(64);
  for (int l = 0; l < 64; l++) { A[l] = 1 + l; B[l] = 100 + 3 * l; }
  double *dA, *dB, *dD; long long* dC;
  CK(hipMalloc(&dA, 512)); CK(hipMalloc(&dB, 512)); CK(hipMalloc(&dD, 512)); CK(hipMalloc(&dC, 8));
  CK(hipMemcpy(dA, A.data(), 512, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), 512, hipMemcpyHostToDevice));
  layout<<<1, 64>>>(dA, dB, dD); CK(hipMemcpy(D.data(), dD, 512, hipMemcpyDeviceToHost));
  // try hypothesis: lane l -> block = l>>4, A[i = l&3 ... ]: brute force search of a consistent mapping
  // print raw so the layout can be decoded offline
  printf("D:"); for (int l = 0; l < 64; l++) printf(" %.0f", D[l]); printf("\n");
  // hypothesis H1: block b = l / 16; within block: A elem (i = l%4, k = (l%16)/4); B elem (k = (l%16)/4, j = l%4); D (i = (l%16)/4?...)
  for (int hyp = 0; hyp < 2; hyp++) {
    int bad = 0;
    for (int l = 0; l < 64; l++) {
      int b = l / 16, x = l % 4, y = (l % 16) / 4;       // D index within block: (row,col) = hyp? (x,y) : (y,x)
      int i = hyp ? x : y, j = hyp ? y : x;
      double s = 0;
      for (int k = 0; k < 4; k++) {
        // A[i][k] held by lane with (l%4 == i, (l%16)/4 == k) in block b; B[k][j] by lane (l%4 == j, (l%16)/4 == k)
        double a = A[b * 16 + k * 4 + i], bb = B[b * 16 + k * 4 + j];
        s += a * bb;
      }
      if (s != D[l]) bad++;
    }
    printf("hypothesis %d mismatches %d\n", hyp, bad);
  }
  double* dout; CK(hipMalloc(&dout, 1 << 22)); long long c; int iters = 2000; hipEvent_t e0, e1; float ms; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
#define RUN(NACC, BLK, THR) rate<NACC><<<BLK, THR>>>(dout, 10, dC); CK(hipDeviceSynchronize()); CK(hipEventRecord(e0)); rate<NACC><<<BLK, THR>>>(dout, iters, dC); CK(hipEventRecord(e1)); CK(hipDeviceSynchronize()); \
  CK(hipEventElapsedTime(&ms, e0, e1)); CK(hipMemcpy(&c, dC, 8, hipMemcpyDeviceToHost)); printf("mfma_f64_4x4x4 nacc=%d blocks=%d thr=%d: %.2f ticks/instr; %.2f TFLOP/s\n", NACC, BLK, THR, (double)c / iters / NACC, 512.0 * iters * NACC * BLK * (THR / 64) / (ms * 1e-3) / 1e12);
  RUN(1, 1, 64) RUN(4, 1, 64) RUN(8, 1, 64) RUN(8, 2048, 256)
  return 0;
}
