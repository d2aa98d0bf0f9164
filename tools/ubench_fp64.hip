// FP64 VALU micro-benchmarks on gfx950: establishes the practical ceilings the geodesic kernel is
// priced against (v_fma_f64 issue rate, IEEE division / sqrt expansions, v_rcp_f64/v_rsq_f64, cv_sincos).
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I curvis_amd/csrc tools/ubench_fp64.hip -o build/ubench_fp64
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "cv_math.h"

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int OP, int ILP>
__global__ __launch_bounds__(256) void k(double *out, int iters, double b, double c) {
  double a[ILP];
  for (int i = 0; i < ILP; ++i) a[i] = 1.0 + 1e-3 * (threadIdx.x + i);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < ILP; ++i) {
      if (OP == 0) a[i] = __builtin_fma(a[i], b, c);
      else if (OP == 1) a[i] = a[i] * b;
      else if (OP == 2) a[i] = a[i] + c;
      else if (OP == 3) a[i] = c / a[i] + b;            // IEEE div + add
      else if (OP == 4) a[i] = __builtin_sqrt(a[i]) + b;  // IEEE sqrt + add
      else if (OP == 5) a[i] = __builtin_amdgcn_rcp(a[i]) + b;
      else if (OP == 6) a[i] = __builtin_amdgcn_rsq(a[i]) + b;
      else if (OP == 7) { double s, cs; cv_sincos(a[i], &s, &cs); a[i] = s + cs + b; }
      else if (OP == 8) a[i] = cv_atan(a[i]) + b;
      else if (OP == 9) a[i] = cv_log(a[i]) + b;
      else if (OP == 10) a[i] = __builtin_amdgcn_frexp_mant(a[i]) + b;                 // v_frexp_mant_f64 + add
      else if (OP == 11) a[i] = c * (double)__builtin_amdgcn_frexp_exp(a[i]) + b;      // v_frexp_exp_i32_f64 + v_cvt_f64_i32 + fma
      else if (OP == 12) a[i] = cv_from_bits((cv_bits(a[i]) & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL) + b;  // v_and + v_or + add
      else if (OP == 13) a[i] = c * (double)(int)((cv_hi(a[i]) >> 20) - 0x3ffu) + b;   // v_lshr + v_add + v_cvt_f64_i32 + fma
    }
  }
  double s = 0;
  for (int i = 0; i < ILP; ++i) s += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int OP, int ILP>
int run(const char *name, double flop_per_op, int blocks, int iters, double *d_out) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL((k<OP, ILP>), dim3(blocks), dim3(256), 0, 0, d_out, 16, 1.0000001, 1e-9);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL((k<OP, ILP>), dim3(blocks), dim3(256), 0, 0, d_out, iters, 1.0000001, 1e-9);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  double ops = (double)blocks * 256.0 * iters * ILP;
  double gops = ops / (ms * 1e-3) / 1e9;
  // cycles per wave-instruction per SIMD at 2.4 GHz nominal: 1024 SIMDs
  double wave_ops_per_s = ops / 64.0 / (ms * 1e-3);
  double cyc = 1024.0 * 2.4e9 / wave_ops_per_s;
  printf("%-28s ILP=%d blocks=%d  %8.3f ms  %10.1f Gop/s  %8.2f TFLOP/s-equiv  %7.1f cyc/wave-op/SIMD@2.4GHz\n", name, ILP, blocks, ms, gops,
         gops * flop_per_op / 1e3, cyc);
  return 0;
}

int main() {
  hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
  printf("device %s %s CUs=%d clock=%d MHz\n", p.name, p.gcnArchName, p.multiProcessorCount, p.clockRate / 1000);
  double *d_out; CHECK(hipMalloc(&d_out, sizeof(double) * 256 * 8192));
  const int B = p.multiProcessorCount * 8;
  for (int occ = 1; occ <= 8; occ *= 2) {
    int blocks = p.multiProcessorCount * occ;
    run<0, 1>("fma chain (dependent)", 2, blocks, 1 << 16, d_out);
  }
  run<0, 4>("fma", 2, B, 1 << 15, d_out);
  run<0, 8>("fma", 2, B, 1 << 15, d_out);
  run<1, 8>("mul", 1, B, 1 << 15, d_out);
  run<2, 8>("add", 1, B, 1 << 15, d_out);
  run<3, 4>("ieee div (+add)", 1, B, 1 << 13, d_out);
  run<4, 4>("ieee sqrt (+add)", 1, B, 1 << 13, d_out);
  run<5, 4>("v_rcp_f64 (+add)", 1, B, 1 << 14, d_out);
  run<6, 4>("v_rsq_f64 (+add)", 1, B, 1 << 14, d_out);
  run<7, 2>("cv_sincos (+2 add)", 1, B, 1 << 12, d_out);
  run<8, 2>("cv_atan (+add)", 1, B, 1 << 12, d_out);
  run<9, 2>("cv_log (+add)", 1, B, 1 << 12, d_out);
  run<10, 4>("v_frexp_mant_f64 (+add)", 1, B, 1 << 14, d_out);
  run<11, 4>("frexp_exp+cvt (+fma)", 1, B, 1 << 14, d_out);
  run<12, 4>("and+or (+add)", 1, B, 1 << 14, d_out);
  run<13, 4>("lshr+add+cvt (+fma)", 1, B, 1 << 14, d_out);
  for (int occ = 1; occ <= 2; occ *= 2) {   // what ONE wave per SIMD can issue: independent vs dependent instructions
    run<0, 4>("fma, few waves", 2, p.multiProcessorCount * occ, 1 << 15, d_out);
    run<0, 8>("fma, few waves", 2, p.multiProcessorCount * occ, 1 << 15, d_out);
  }
  for (int occ = 1; occ <= 8; occ *= 2) run<3, 1>("ieee div dependent", 1, p.multiProcessorCount * occ, 1 << 13, d_out);
  return 0;
}
