// Probe: on which SIMD of its CU does wave w of a 4-wave (or 8-wave) workgroup land, with three workgroups co-resident per CU
// (the cold kernel's launch shape)?  Reads HW_ID (gfx9: wave_id[3:0] simd_id[5:4] pipe_id[7:6] cu_id[11:8] sh_id[12]
// se_id[15:13]) and XCC_ID; every wave spins long enough for the whole grid's first round to be resident together.
//   hipcc --offload-arch=gfx950 -O3 -o tools/simd_placement_probe tools/simd_placement_probe.hip && tools/simd_placement_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ void __launch_bounds__(512) probe(unsigned* out, int spin) {
  extern __shared__ double sm[];
  const int w = threadIdx.x >> 6;
  const unsigned hw = __builtin_amdgcn_s_getreg((15 << 11) | (0 << 6) | 4);      // HW_ID bits 15:0
  const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);     // XCC_ID bits 3:0
  const long long t0 = __builtin_amdgcn_s_memtime();
  double v = threadIdx.x;
  for (int i = 0; i < spin; ++i) v = fma(v, 1.0000001, 1e-9);
  sm[threadIdx.x] = v;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) {
    out[(blockIdx.x * 8 + w) * 2] = hw | (xcc << 16);
    out[(blockIdx.x * 8 + w) * 2 + 1] = (unsigned)(t0 >> 8) + (sm[0] == 12345.0);
  }
}

int main(int argc, char** argv) {
  const int W = argc > 1 ? atoi(argv[1]) : 4;
  const int nwg = argc > 2 ? atoi(argv[2]) : 4096;
  const int lds = argc > 3 ? atoi(argv[3]) : 53248;
  unsigned* d;
  hipMalloc(&d, nwg * 8 * 2 * sizeof(unsigned));
  hipMemset(d, 0, nwg * 8 * 2 * sizeof(unsigned));
  hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  probe<<<nwg, 64 * W, lds>>>(d, 20000);
  hipDeviceSynchronize();
  std::vector<unsigned> h(nwg * 16);
  hipMemcpy(h.data(), d, h.size() * sizeof(unsigned), hipMemcpyDeviceToHost);
  // histogram: SIMD of wave w; and for wave 0, the SIMD pattern of the workgroups sharing a CU in the first round
  int hist[8][4] = {};
  for (int b = 0; b < nwg; ++b)
    for (int w = 0; w < W; ++w) hist[w][(h[(b * 8 + w) * 2] >> 4) & 3]++;
  for (int w = 0; w < W; ++w) printf("wave %d: simd0 %d simd1 %d simd2 %d simd3 %d\n", w, hist[w][0], hist[w][1], hist[w][2], hist[w][3]);
  printf("first 24 workgroups: (xcc se sh cu | simd of waves)\n");
  for (int b = 0; b < 24 && b < nwg; ++b) {
    const unsigned hw = h[(b * 8) * 2];
    printf("  wg %4d xcc %u se %u sh %u cu %2u t0 %u |", b, hw >> 16, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 15, h[b * 16 + 1]);
    for (int w = 0; w < W; ++w) printf(" %u", (h[(b * 8 + w) * 2] >> 4) & 3);
    printf("\n");
  }
  // per CU (xcc, se, sh, cu): how many distinct SIMDs host a wave 0 among the workgroups placed there
  int same = 0, total = 0;
  for (int a = 0; a < nwg; ++a)
    for (int b = a + 1; b < nwg && b < a + 4096; ++b) {
      const unsigned ha = h[a * 16], hb = h[b * 16];
      if ((ha >> 8) == (hb >> 8) && abs((int)h[a * 16 + 1] - (int)h[b * 16 + 1]) < 50) { ++total; same += ((ha >> 4) & 3) == ((hb >> 4) & 3); }
    }
  printf("pairs of workgroups on one CU starting together: %d, of which wave 0 on the same SIMD: %d\n", total, same);
  return 0;
}
