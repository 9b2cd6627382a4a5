// Probe: fp64 VALU issue rate / dependent latency, v_rsq_f64, LDS round trips, s_barrier, for ONE wave
// (and for 2 waves on one SIMD).  Build: hipcc --offload-arch=gfx950 -O3 tools/valu_f64_probe.hip -o tools/valu_f64_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)

template <int MODE>
__global__ void k(double* out, long long* cyc, int iters) {
  __shared__ double lds[4096];
  const int t = threadIdx.x;
  lds[t] = t * 0.5; lds[t + 1024] = 1.0; lds[t + 2048] = 2.0;
  __syncthreads();
  double x[16];
  for (int i = 0; i < 16; i++) x[i] = 1.0 + t * 1e-3 + i;
  double a = 1.0000001, b = 1e-9;
  int idx = t;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; it++) {
    if (MODE == 0) {          // 16 independent FMAs
#pragma unroll
      for (int i = 0; i < 16; i++) x[i] = fma(x[i], a, b);
    } else if (MODE == 1) {   // 16 dependent FMAs
#pragma unroll
      for (int i = 0; i < 16; i++) x[0] = fma(x[0], a, b);
    } else if (MODE == 2) {   // 4 dependent rsqrt (library)
#pragma unroll
      for (int i = 0; i < 4; i++) x[0] = rsqrt(x[0] + 1.5);
    } else if (MODE == 3) {   // 4 dependent raw v_rsq_f64
#pragma unroll
      for (int i = 0; i < 4; i++) x[0] = __builtin_amdgcn_rsq(x[0] + 1.5);
    } else if (MODE == 4) {   // dependent LDS read chain (pointer chasing through values)
#pragma unroll
      for (int i = 0; i < 4; i++) { idx = (int)lds[idx & 1023] & 1023; }
    } else if (MODE == 5) {   // LDS write -> barrier -> read (one round trip + barrier)
      lds[2048 + t] = x[0];
      __syncthreads();
      x[0] += lds[2048 + ((t + 1) & (blockDim.x - 1))];
      __syncthreads();
    } else if (MODE == 6) {   // 16 independent f64 multiplies by cndmask-select (v_cndmask pairs)
#pragma unroll
      for (int i = 0; i < 16; i++) x[i] = (t & (1 << (i & 3))) ? x[i] : x[(i + 1) & 15];
    } else if (MODE == 7) {   // 8 independent ds_read_b64
      double s = 0;
#pragma unroll
      for (int i = 0; i < 8; i++) s += lds[(t + 64 * i) & 4095];
      x[0] += s;
    }
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  double s = idx;
  for (int i = 0; i < 16; i++) s += x[i];
  out[blockIdx.x * blockDim.x + t] = s;
  if (t == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

int main() {
  double* dout; long long* dc; long long c;
  CK(hipMalloc(&dout, 1 << 20)); CK(hipMalloc(&dc, 8));
  const int iters = 2000;
  const char* names[] = {"16 indep fma_f64", "16 dep fma_f64", "4 dep rsqrt(lib)", "4 dep v_rsq_f64", "4 dep ds_read chain",
                         "lds write+barrier+read+barrier", "16 select(f64)", "8 indep ds_read_b64 + adds"};
  const int per[] = {16, 16, 4, 4, 4, 1, 16, 8};
#define RUN(M, THR) k<M><<<1, THR>>>(dout, dc, 10); CK(hipDeviceSynchronize()); k<M><<<1, THR>>>(dout, dc, iters); CK(hipDeviceSynchronize()); \
  CK(hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost)); printf("%-34s threads=%4d : %.1f cycles per op (%.0f per iteration)\n", names[M], THR, (double)c / iters / per[M], (double)c / iters);
  RUN(0, 64) RUN(0, 256) RUN(0, 512) RUN(1, 64) RUN(1, 256) RUN(2, 64) RUN(3, 64) RUN(4, 64) RUN(5, 64) RUN(5, 256) RUN(5, 512) RUN(6, 64) RUN(7, 64) RUN(7, 256)
  return 0;
}
