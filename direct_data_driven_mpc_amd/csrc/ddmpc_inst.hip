// One explicit instantiation of the cold-solve kernels per translation unit:
//   hipcc -DDDMPC_INST_NT=9 -DDDMPC_INST_W=4 [-DDDMPC_INST_REF=true | -DDDMPC_INST_V1] -c ddmpc_inst.hip
// Default: the 16-wide-panel kernel (ddmpc_cold2.hpp).  -DDDMPC_INST_V1: the first-generation kernel
// (ddmpc_kernels.hpp), kept selectable (DDMPC_KERNEL=1) for A/B measurements.
#include "ddmpc_cold2.hpp"
namespace ddmpc {
#ifdef DDMPC_INST_V1
template __global__ void ddmpc_cold_solve_kernel<DDMPC_INST_NT, DDMPC_INST_W>(
    KParams, const double*, const double*, const double*, const double*, double*, double*, int*, int*,
    double*, signed char*, unsigned long long*, double*, const int*);
#else
#ifndef DDMPC_INST_REF
#define DDMPC_INST_REF false
#endif
template __global__ void ddmpc_cold_solve_kernel2<DDMPC_INST_NT, DDMPC_INST_W, DDMPC_INST_REF>(
    KParams, const double*, const double*, const double*, const double*, double*, double*, int*, int*,
    double*, signed char*, unsigned long long*, double*, double*, int*, const int*, long long, int*);
#endif
}
