// One explicit instantiation of the cold-solve kernel per translation unit:
//   hipcc -DDDMPC_INST_NT=9 -DDDMPC_INST_W=4 [-DDDMPC_INST_REF=true] -c ddmpc_inst.hip
// DDMPC_INST_REF=true: the variant with the iterative-refinement loop compiled in (ddmpc_cold2.hpp).
// DDMPC_INST_CVX=true: the plain variant with the rank-k treatment of the slack box compiled in (controllers with the CONVEX box).
#include "ddmpc_cold2.hpp"
namespace ddmpc {
#ifndef DDMPC_INST_REF
#define DDMPC_INST_REF false
#endif
#ifndef DDMPC_INST_CVX
#define DDMPC_INST_CVX false
#endif
template __global__ void ddmpc_cold_solve_kernel2<DDMPC_INST_NT, DDMPC_INST_W, DDMPC_INST_REF, DDMPC_INST_CVX>(
    KParams, const double*, const double*, const double*, const double*, double*, double*, int*, int*,
    double*, signed char*, unsigned long long*, double*, double*, int*, const int*, long long, int*);
}
