// One explicit instantiation of the cold-solve kernel per translation unit:
//   hipcc -DDDMPC_INST_NT=9 -DDDMPC_INST_W=4 -c ddmpc_inst.hip
#include "ddmpc_kernels.hpp"
namespace ddmpc {
template __global__ void ddmpc_cold_solve_kernel<DDMPC_INST_NT, DDMPC_INST_W>(
    KParams, const double*, const double*, const double*, const double*, double*, double*, int*, int*,
    double*, signed char*, unsigned long long*, double*, const int*);
}
