// Dev probe: which lanes v_permlane16_swap / v_permlane32_swap exchange on gfx950.
//   hipcc --offload-arch=gfx950 -O2 tools/permlane_probe.hip -o tools/permlane_probe && tools/permlane_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out) {
  const unsigned lane = threadIdx.x;
  unsigned a = lane, b = 100 + lane;
  auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  out[lane] = r[0]; out[64 + lane] = r[1];
  auto s = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  out[128 + lane] = s[0]; out[192 + lane] = s[1];
}
int main() {
  unsigned* d; unsigned h[256];
  hipMalloc(&d, sizeof(h));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const char* names[4] = {"permlane16_swap vdst (a = lane)", "permlane16_swap src  (b = 100 + lane)", "permlane32_swap vdst", "permlane32_swap src "};
  for (int v = 0; v < 4; ++v) {
    printf("%s:\n", names[v]);
    for (int row = 0; row < 4; ++row) { printf("  row %d:", row); for (int i = 0; i < 16; ++i) printf(" %3u", h[v * 64 + row * 16 + i]); printf("\n"); }
  }
  return 0;
}
