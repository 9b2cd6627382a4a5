// Probe: cycles per 4-wide sub-step of the in-tile factorisation of ddmpc_cold2.hpp (one wave alone on a CU), and what the
// pieces of its dependency chain cost.  The sub-step is re-stated here with switches:
//   bit 0: pivot block through v_mfma_f64_4x4x4 + v_readlane (the shipped form); else: carried in registers (no hand-off at all)
//   bit 1: the two 16x16x4 MFMAs (rank-4 update of the tile and of the identity block)
//   bit 2: panel columns through LDS (write from the accumulators, read per row); else: taken from registers
//   bit 3: the substitution of the wave's rows (x0..x3), permlane operands
//   hipcc --offload-arch=gfx950 -O3 -o tools/substep_probe tools/substep_probe.hip && tools/substep_probe
#include <hip/hip_runtime.h>
#include <cstdio>

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

template <int L>
__device__ __forceinline__ double readlane_f64(double v) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), L);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), L);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ d2 permlane16_swap_f64(double a, double b) {
  const auto lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
  const auto hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
  return d2{__hiloint2double((int)hi[0], (int)lo[0]), __hiloint2double((int)hi[1], (int)lo[1])};
}
__device__ __forceinline__ double rsq_n2(double x) {
  const double y0 = __builtin_amdgcn_rsq(x);
  const double e = fma(-x * y0, y0, 1.0);
  return fma(0.5 * y0, e, y0);
}

template <int MODE>
__global__ void probe(double* out, long long* cyc, int reps) {
  __shared__ double PT2[32 * 4];
  const int lane = threadIdx.x & 63, l15 = lane & 15, l4 = lane >> 4, l3 = lane & 3, lo = l15 >> 2;
  const int x = lane & 31;
  d4 Ad, Et;
  for (int j = 0; j < 4; ++j) { Ad[j] = (l4 + 4 * j == l15) ? -40.0 : -0.01 * ((lane + j) & 7); Et[j] = (l4 + 4 * j == l15) ? -1.0 : 0.0; }
  double pdg = (lo == 0) ? Ad[0] : (lo == 1) ? Ad[1] : (lo == 2) ? Ad[2] : Ad[3];
  double accum = 0.0;
  double oAp = 0.0, oEp = 0.0;                   // operands of the previous sub-step (MODE & 256: its identity-block MFMA is issued late)
  long long t0 = 0, t1 = 0;
  for (int r = 0; r < reps + 1; ++r) {
    if (r == 1) t0 = __builtin_amdgcn_s_memtime();
    double p00, p10, p11, p20, p21, p22, p30, p31, p32, p33;
    if constexpr (MODE & 1) {
      p00 = -readlane_f64<0>(pdg);
      p10 = -readlane_f64<16>(pdg); p11 = -readlane_f64<17>(pdg);
      p20 = -readlane_f64<32>(pdg); p21 = -readlane_f64<33>(pdg); p22 = -readlane_f64<34>(pdg);
      p30 = -readlane_f64<48>(pdg); p31 = -readlane_f64<49>(pdg); p32 = -readlane_f64<50>(pdg); p33 = -readlane_f64<51>(pdg);
    } else if constexpr (MODE & 128) {
      // (filled in below from the handed-over columns)
      p00 = p10 = p11 = p20 = p21 = p22 = p30 = p31 = p32 = p33 = 0.0;
    } else {
      p00 = 40.0 - pdg * 1e-30; p10 = 0.01; p11 = 40.0; p20 = 0.02; p21 = 0.01; p22 = 40.0; p30 = 0.03; p31 = 0.02; p32 = 0.01; p33 = 40.0;
    }
    double r0, r1, r2, r3;
    if constexpr ((MODE & 129) == 128) {          // pivot block = rows 0..3 of the handed-over columns
      const d2 s1 = permlane16_swap_f64(Ad[0], Et[0]);
      const auto sw32 = [](double v, double& a, double& b) __attribute__((always_inline)) {
        const auto lo2 = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(v), (unsigned)__double2loint(v), false, false);
        const auto hi2 = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(v), (unsigned)__double2hiint(v), false, false);
        a = __hiloint2double((int)hi2[0], (int)lo2[0]); b = __hiloint2double((int)hi2[1], (int)lo2[1]);
      };
      sw32(s1[0], r0, r2); sw32(s1[1], r1, r3);
      p00 = -readlane_f64<0>(r0);
      p10 = -readlane_f64<1>(r0); p11 = -readlane_f64<1>(r1);
      p20 = -readlane_f64<2>(r0); p21 = -readlane_f64<2>(r1); p22 = -readlane_f64<2>(r2);
      p30 = -readlane_f64<3>(r0); p31 = -readlane_f64<3>(r1); p32 = -readlane_f64<3>(r2); p33 = -readlane_f64<3>(r3);
    }
    double i0 = rsq_n2(p00);
    if constexpr (!(MODE & 512)) asm volatile("" : "+v"(i0));
    if constexpr (MODE & 256) Et = __builtin_amdgcn_mfma_f64_16x16x4f64(oAp, oEp, Et, 0, 0, 0);   // behind the first pivot: the matrix pipe is free again
    auto panel = [&]() __attribute__((always_inline)) {
      if constexpr (MODE & 128) {
        // panel columns straight out of the accumulators: register 0 of lane (v, i) is entry (i, v) of the tile (symmetric) and of
        // the TRANSPOSED identity block; two levels of row swaps bring column v to every lane of the tile rows / identity rows
        const d2 s1 = permlane16_swap_f64(Ad[0], Et[0]);              // [A.r0 E.r0 A.r2 E.r2], [A.r1 E.r1 A.r3 E.r3]
        const auto sw32 = [](double v, double& a, double& b) __attribute__((always_inline)) {
          const auto lo2 = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(v), (unsigned)__double2loint(v), false, false);
          const auto hi2 = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(v), (unsigned)__double2hiint(v), false, false);
          a = __hiloint2double((int)hi2[0], (int)lo2[0]); b = __hiloint2double((int)hi2[1], (int)lo2[1]);
        };
        sw32(s1[0], r0, r2); sw32(s1[1], r1, r3);
      } else
      if constexpr (MODE & 4) {
        if (lo == 0) {
          for (int j = 0; j < 4; ++j) { PT2[(l4 + 4 * j) * 4 + l3] = Ad[j]; PT2[(16 + l4 + 4 * j) * 4 + l3] = Et[j]; }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        r0 = PT2[x * 4 + 0]; r1 = PT2[x * 4 + 1]; r2 = PT2[x * 4 + 2]; r3 = PT2[x * 4 + 3];
        __builtin_amdgcn_sched_barrier(0);
      } else {
        r0 = Ad[0]; r1 = Ad[1]; r2 = Et[2]; r3 = Et[3];
      }
    };
    if constexpr ((MODE & 48) == 0 && (MODE & 129) != 128) panel();
    const double l10 = p10 * i0, l20 = p20 * i0, l30 = p30 * i0;
    double i1 = rsq_n2(p11 - l10 * l10);
    if constexpr ((MODE & 48) == 16) { asm volatile("" : "+v"(i1)); panel(); }
    const double l21 = (p21 - l20 * l10) * i1, l31 = (p31 - l30 * l10) * i1;
    double i2 = rsq_n2(p22 - l20 * l20 - l21 * l21);
    if constexpr ((MODE & 48) == 32) { asm volatile("" : "+v"(i2)); panel(); }
    const double l32 = (p32 - l30 * l20 - l31 * l21) * i2;
    double i3 = rsq_n2(p33 - l30 * l30 - l31 * l31 - l32 * l32);
    if constexpr (MODE & 64) { asm volatile("" : "+v"(i3), "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3)); }   // the wait for the LDS data stays behind the pivots
    double opA, opE;
    if constexpr (MODE & 8) {
      const double x0 = -r0 * i0;
      const double x1 = -(r1 + x0 * l10) * i1;
      const double x2 = -(r2 + x0 * l20 + x1 * l21) * i2;
      const double x3 = -(r3 + x0 * l30 + x1 * l31 + x2 * l32) * i3;
      const d2 s01a = permlane16_swap_f64(x0, x1), s23a = permlane16_swap_f64(x2, x3);
      const bool lowhalf = lane < 32;
      opA = (lowhalf ? s01a[0] : s23a[0]) * 1e-3;
      opE = (lowhalf ? s01a[1] : s23a[1]) * 1e-3;
    } else {
      opA = i3 * 1e-3 + r0 * 1e-9; opE = i3 * 1e-3 + r3 * 1e-9;
    }
    if constexpr (MODE & 1) {
      pdg = __builtin_amdgcn_mfma_f64_4x4x4f64(opA, opA, pdg, 0, 0, 0);
      if constexpr (!(MODE & 512)) __builtin_amdgcn_sched_barrier(0);
    } else {
      pdg = pdg + opA * 1e-30;
    }
    if constexpr (MODE & 2) {
      Ad = __builtin_amdgcn_mfma_f64_16x16x4f64(opA, opA, Ad, 0, 0, 0);
      if constexpr (MODE & 256) { oAp = opA; oEp = opE; }
      else if constexpr (MODE & 128) Et = __builtin_amdgcn_mfma_f64_16x16x4f64(opA, opE, Et, 0, 0, 0);
      else Et = __builtin_amdgcn_mfma_f64_16x16x4f64(opE, opA, Et, 0, 0, 0);
      if constexpr (MODE & 128) { PT2[lane] = opE; }                  // the one store left: a row block of M for the other waves
    } else {
      Ad[0] += opA * 1e-30; Et[3] += opE * 1e-30;
    }
    accum += i3;
  }
  t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) cyc[blockIdx.x] = t1 - t0;
  out[blockIdx.x * 64 + lane] = accum + Ad[0] + Ad[1] + Ad[2] + Ad[3] + Et[0] + Et[1] + Et[2] + Et[3] + pdg;
}

// Two instances per instruction stream (round-5 experiment): lanes 0..31 = the 16 tile rows + 16 identity rows of
// instance A, lanes 32..63 = those of instance B.  The pivot chain, the substitution of the rows and the LDS hand-off are ONE
// set of instructions for both (each half wave works on its own 4 x 4 block and its own rows); what cannot be shared is the
// matrix pipe: a v_mfma_f64_16x16x4 takes its operand from all 64 lanes, so every instance needs its own pair of MFMAs, with
// operands that first have to be spread from its half wave over the whole wave (v_permlane32_swap in front of the
// v_permlane16_swap pair of the shipped sub-step).
__global__ void probe_pair(double* out, long long* cyc, int reps) {
  __shared__ double PT2[2 * 32 * 4];
  const int lane = threadIdx.x & 63, l15 = lane & 15, l4 = lane >> 4, l3 = lane & 3, lo = l15 >> 2;
  const int x = lane & 31, inst = lane >> 5;
  d4 Ad[2], Et[2];
  for (int t = 0; t < 2; ++t)
    for (int j = 0; j < 4; ++j) { Ad[t][j] = (l4 + 4 * j == l15) ? -40.0 - t : -0.01 * ((lane + j + t) & 7); Et[t][j] = (l4 + 4 * j == l15) ? -1.0 : 0.0; }
  double accum = 0.0;
  long long t0 = 0, t1 = 0;
  double* P = PT2 + inst * 128;                      // this half wave's panel columns
  for (int r = 0; r < reps + 1; ++r) {
    if (r == 1) t0 = __builtin_amdgcn_s_memtime();
    // panel columns of both instances out of their accumulators (one store instruction per register pair, as shipped)
    if (lo == 0) {
      for (int t = 0; t < 2; ++t)
        for (int j = 0; j < 4; ++j) { PT2[t * 128 + (l4 + 4 * j) * 4 + l3] = Ad[t][j]; PT2[t * 128 + (16 + l4 + 4 * j) * 4 + l3] = Et[t][j]; }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    const double p00 = -P[0];
    const double p10 = -P[4], p11 = -P[5];
    const double p20 = -P[8], p21 = -P[9], p22 = -P[10];
    const double p30 = -P[12], p31 = -P[13], p32 = -P[14], p33 = -P[15];
    const double r0 = P[x * 4 + 0], r1 = P[x * 4 + 1], r2 = P[x * 4 + 2], r3 = P[x * 4 + 3];
    const double i0 = rsq_n2(p00);
    const double l10 = p10 * i0, l20 = p20 * i0, l30 = p30 * i0;
    const double i1 = rsq_n2(p11 - l10 * l10);
    const double l21 = (p21 - l20 * l10) * i1, l31 = (p31 - l30 * l10) * i1;
    const double i2 = rsq_n2(p22 - l20 * l20 - l21 * l21);
    const double l32 = (p32 - l30 * l20 - l31 * l21) * i2;
    const double i3 = rsq_n2(p33 - l30 * l30 - l31 * l31 - l32 * l32);
    const double x0 = -r0 * i0;
    const double x1 = -(r1 + x0 * l10) * i1;
    const double x2 = -(r2 + x0 * l20 + x1 * l21) * i2;
    const double x3 = -(r3 + x0 * l30 + x1 * l31 + x2 * l32) * i3;
    // operands: for instance t the values of ITS half wave in both halves (permlane32 swap with itself), then the shipped swaps
    auto spread = [&](double v, int t) __attribute__((always_inline)) -> double {
      const auto lo2 = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(v), (unsigned)__double2loint(v), false, false);
      const auto hi2 = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(v), (unsigned)__double2hiint(v), false, false);
      // swap(v, v): result[0] = [v.lower half, v.lower half], result[1] = [v.upper half, v.upper half]
      return t == 0 ? __hiloint2double((int)hi2[0], (int)lo2[0]) : __hiloint2double((int)hi2[1], (int)lo2[1]);
    };
    for (int t = 0; t < 2; ++t) {
      const double y0 = spread(x0, t), y1 = spread(x1, t), y2 = spread(x2, t), y3 = spread(x3, t);
      const d2 s01a = permlane16_swap_f64(y0, y1), s23a = permlane16_swap_f64(y2, y3);
      const bool lowhalf = lane < 32;
      const double opA = (lowhalf ? s01a[0] : s23a[0]) * 1e-3;
      const double opE = (lowhalf ? s01a[1] : s23a[1]) * 1e-3;
      Ad[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(opA, opA, Ad[t], 0, 0, 0);
      Et[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(opE, opA, Et[t], 0, 0, 0);
    }
    accum += i3;
  }
  t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) cyc[blockIdx.x] = t1 - t0;
  out[blockIdx.x * 64 + lane] = accum + Ad[0][0] + Ad[1][1] + Et[0][2] + Et[1][3];
}

void run_pair() {
  double* out; long long* cyc;
  hipMalloc(&out, 64 * 8 * sizeof(double)); hipMalloc(&cyc, 8 * sizeof(long long));
  const int reps = 2000;
  probe_pair<<<1, 64>>>(out, cyc, reps);
  hipDeviceSynchronize();
  long long h; hipMemcpy(&h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  printf("pair     %-90s %7.1f cycles per sub-step of TWO instances = %.1f per instance\n",
         "whole sub-step, two instances per instruction stream (half wave each, 4 MFMAs, operands spread by permlane32)", (double)h / reps, (double)h / reps / 2);
  hipFree(out); hipFree(cyc);
}

template <int MODE>
void run(const char* what) {
  double* out; long long* cyc;
  hipMalloc(&out, 64 * 8 * sizeof(double)); hipMalloc(&cyc, 8 * sizeof(long long));
  const int reps = 2000;
  probe<MODE><<<1, 64>>>(out, cyc, reps);
  hipDeviceSynchronize();
  long long h; hipMemcpy(&h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  printf("mode %2d  %-90s %7.1f cycles per sub-step\n", MODE, what, (double)h / reps);
  hipFree(out); hipFree(cyc);
}

int main() {
  run<0>("pivot chain alone (four dependent rsq pivots, block carried in registers)");
  run<1>("+ pivot block through v_mfma_f64_4x4x4 + 20 v_readlane");
  run<8>("pivot chain + substitution of the rows + permlane operands");
  run<9>("pivot block through 4x4x4 MFMA + readlane, substitution, permlane operands");
  run<11>("... + the two 16x16x4 MFMAs");
  run<13>("... + panel columns through LDS, without the 16x16x4 MFMAs");
  run<15>("the whole sub-step");
  run<14>("whole sub-step but the pivot block carried in registers (no hand-off of the pivots)");
  run<15 + 16>("whole sub-step, panel columns (LDS write + read) issued behind the SECOND pivot");
  run<15 + 32>("whole sub-step, panel columns (LDS write + read) issued behind the THIRD pivot");
  run<13 + 16>("... behind the second pivot, without the 16x16x4 MFMAs");
  run<15 + 64>("whole sub-step, hand-off behind the first pivot, the wait for its data pinned behind the last pivot");
  run<13 + 64>("... without the 16x16x4 MFMAs");
  run<15 + 16 + 64>("whole sub-step, hand-off behind the second pivot, wait pinned behind the last pivot");
  run<14 + 64>("pivot block in registers, wait pinned behind the last pivot");
  run<11 + 128>("register hand-off of the panel columns (permlane swaps), one LDS store per sub-step, pivots via 4x4x4 + readlane");
  run<11 + 128 + 16>("... hand-off issued behind the second pivot");
  run<11 + 128 + 32>("... hand-off issued behind the third pivot");
  run<11 + 128 + 512>("register hand-off, no scheduling pins at all (one basic block, the compiler interleaves)");
  run<10 + 128 + 512>("... and the pivot block taken from the handed-over columns by v_readlane (no 4x4x4 MFMA)");
  run<11 + 128 + 256 + 16>("register hand-off behind the second pivot, identity-block MFMA of the previous sub-step issued behind the first pivot");
  run<11 + 128 + 256 + 32>("... hand-off behind the third pivot");
  run_pair();
  return 0;
}
