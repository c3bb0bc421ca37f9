// Micro-benchmark 2: (a) v_fma_f64 rate vs ILP/occupancy, (b) do MFMA-f64 and VALU-f64 overlap?
#include <hip/hip_runtime.h>
#include <cstdio>

template <int NACC>
__global__ void k_fma(double* out, double a, double b, int iters) {
  double acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = threadIdx.x * 1e-3 + i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = fma(acc[i], a, b);
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// NM mfma + NF fma per iteration, all independent chains
template <int NM, int NF>
__global__ void k_mix(double* out, double a, double b, int iters) {
  double macc[NM > 0 ? NM : 1];
  double facc[NF > 0 ? NF : 1];
#pragma unroll
  for (int i = 0; i < NM; ++i) macc[i] = 0;
#pragma unroll
  for (int i = 0; i < NF; ++i) facc[i] = threadIdx.x * 1e-3 + i;
  double av = a + threadIdx.x * 1e-6, bv = b;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < (NM > NF ? NM : NF); ++i) {
      if (i < NM) macc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(av, bv, macc[i], 0, 0, 0);
      if (i < NF) facc[i] = fma(facc[i], a, b);
    }
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < NM; ++i) s += macc[i];
#pragma unroll
  for (int i = 0; i < NF; ++i) s += facc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename F>
float timeit(F launch) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  launch();
  hipDeviceSynchronize();
  hipEventRecord(e0);
  launch();
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

#define RUN_FMA(NACC, WPC)                                                                                         \
  {                                                                                                                \
    float ms = timeit([&] { hipLaunchKernelGGL(k_fma<NACC>, dim3(256), dim3(64 * WPC), 0, 0, out, 1.0000001, 1e-9, iters); }); \
    printf("fma acc=%2d waves/CU=%2d : %7.2f TF\n", NACC, WPC, 2.0 * NACC * iters * 64.0 * WPC * 256 / ms * 1e-9); \
  }
#define RUN_MIX(NM, NF, WPC)                                                                                       \
  {                                                                                                                \
    float ms = timeit([&] { hipLaunchKernelGGL((k_mix<NM, NF>), dim3(256), dim3(64 * WPC), 0, 0, out, 1.0000001, 1e-9, iters); }); \
    double fm = 512.0 * NM * iters * WPC * 256, ff = 2.0 * NF * iters * 64.0 * WPC * 256;                          \
    printf("mix mfma=%2d fma=%2d waves/CU=%2d : mfma %7.2f TF + fma %7.2f TF = %7.2f TF (%.3f ms)\n", NM, NF, WPC, \
           fm / ms * 1e-9, ff / ms * 1e-9, (fm + ff) / ms * 1e-9, ms);                                             \
  }

int main() {
  double* out;
  hipMalloc(&out, sizeof(double) * 256 * 1024);
  const int iters = 10000;
  RUN_FMA(8, 4) RUN_FMA(16, 4) RUN_FMA(32, 4) RUN_FMA(64, 4)
  RUN_FMA(8, 8) RUN_FMA(16, 8) RUN_FMA(32, 8) RUN_FMA(64, 8)
  RUN_FMA(16, 16) RUN_FMA(32, 16)
  RUN_FMA(8, 32) RUN_FMA(16, 32)
  RUN_MIX(8, 0, 8) RUN_MIX(0, 32, 8) RUN_MIX(8, 8, 8) RUN_MIX(8, 16, 8) RUN_MIX(8, 32, 8) RUN_MIX(8, 64, 8)
  RUN_MIX(8, 32, 4) RUN_MIX(8, 32, 16) RUN_MIX(4, 32, 8)
  return 0;
}
