// Probe: cycles per column step of a row-per-lane 16x16 in-wave elimination (one wave), and of its ingredients.
//   hipcc --offload-arch=gfx950 -O3 -o tools/intile_probe tools/intile_probe.hip && tools/intile_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <type_traits>

template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (N > 0) { static_for<N - 1>(f); f(std::integral_constant<int, N - 1>{}); }
}
__device__ __forceinline__ double rsq_n2(double x) {
  const double y0 = __builtin_amdgcn_rsq(x);
  const double e = fma(-x * y0, y0, 1.0);
  return fma(0.5 * y0, e, y0);
}
template <int L>
__device__ __forceinline__ double readlane_f64(double v) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), L);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), L);
  return __hiloint2double(hi, lo);
}

// MODE 0: full column steps (readlane pivot, rsq, scale, all updates through readlane scalars)
// MODE 1: chain only (pivot readlane -> rsq_n2 -> scale -> readlane -> fma of the next entry), no other updates
// MODE 2: like 1 without the readlanes (lane-local pivot)
// MODE 3: like 2 with a plain multiply instead of rsq_n2 (dependent fp64 ops only)
// MODE 4: like 1 with DPP-free ds_bpermute broadcasts (__shfl) instead of readlane
template <int MODE>
__global__ void probe(double* out, long long* cyc, int reps) {
  const int lane = threadIdx.x & 63;
  double ar[16];
  double accum = 0.0;
  long long t0 = 0, t1 = 0;
  for (int r = 0; r < reps + 1; ++r) {
    static_for<16>([&](auto c) { ar[c()] = ((lane & 15) == c()) ? 20.0 + c() : 1.0 / (1 + ((lane + c()) & 7)); });
    if (r == 1) t0 = __builtin_amdgcn_s_memtime();
    static_for<16>([&](auto K) {
      constexpr int k = K;
      double piv;
      if constexpr (MODE == 0 || MODE == 1) piv = readlane_f64<k>(ar[k]);
      else if constexpr (MODE == 4) piv = __shfl(ar[k], k, 64);
      else piv = ar[k] + 16.0;
      double inv;
      if constexpr (MODE == 3) inv = piv * 0.25; else inv = rsq_n2(piv);
      const double l = ar[k] * inv;
      if constexpr (k + 1 < 16) {
        double s1;
        if constexpr (MODE == 0 || MODE == 1) s1 = readlane_f64<(k + 1 < 16 ? k + 1 : 15)>(l);
        else if constexpr (MODE == 4) s1 = __shfl(l, k + 1, 64);
        else s1 = l;
        ar[k + 1] = fma(-l, s1, ar[k + 1]);
      }
      if constexpr (MODE == 0) {
        static_for<16>([&](auto C) {
          constexpr int c = C;
          if constexpr (c >= k + 2) {
            const double sc = readlane_f64<c>(l);
            ar[c] = fma(-l, sc, ar[c]);
          }
        });
      }
      accum += l;
    });
  }
  t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * 64 + lane] = accum;
  if (lane == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char* name, int blocks) {
  double* out; long long* cyc;
  hipMalloc(&out, sizeof(double) * 64 * blocks); hipMalloc(&cyc, sizeof(long long) * blocks);
  const int reps = 200;
  probe<MODE><<<blocks, 64>>>(out, cyc, reps);
  hipDeviceSynchronize();
  long long h[4096]; hipMemcpy(h, cyc, sizeof(long long) * blocks, hipMemcpyDeviceToHost);
  double o0; hipMemcpy(&o0, out, sizeof(double), hipMemcpyDeviceToHost);
  printf("%-62s blocks %4d: %7.1f clk (s_memtime ticks) per column step   [check %g]\n", name, blocks, (double)h[0] / reps / 16.0, o0);
  hipFree(out); hipFree(cyc);
}

int main() {
  for (int blocks : {1, 1024}) {
    run<0>("full column steps (readlane scalars for every update)", blocks);
    run<1>("chain only: readlane pivot, rsq+Newton, scale, readlane, fma", blocks);
    run<2>("chain without readlanes", blocks);
    run<3>("chain without readlanes and without rsq (dependent fp64 ops)", blocks);
    run<4>("chain with ds_bpermute broadcasts instead of readlane", blocks);
  }
  return 0;
}
