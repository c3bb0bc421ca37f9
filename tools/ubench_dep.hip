// Micro-benchmark: latency of DEPENDENT fp64 MFMA chains on gfx950 (one wave per SIMD, NACC independent
// accumulators per wave): cycles per instruction at 1 wave/SIMD = max(issue, dependent latency / NACC).
// hipcc --offload-arch=gfx950 -O3 -o ubench_dep ubench_dep.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ void k_mfma16(double* out, double a, double b, int iters) {
  d4 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = (d4){0, 0, 0, 0};
  double av = a + threadIdx.x * 1e-6, bv = b;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc[i], 0, 0, 0);
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
__global__ void k_mfma4(double* out, double a, double b, int iters) {
  double acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = 0;
  double av = a + threadIdx.x * 1e-6, bv = b;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(av, bv, acc[i], 0, 0, 0);
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <typename F>
float timeit(F launch) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) launch();
  hipDeviceSynchronize();
  hipEventRecord(e0);
  launch();
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms;
}
int main() {
  double* out;
  hipMalloc(&out, 256 * 1024 * sizeof(double));
  const int iters = 200000;
  const int grid = 256;  // one 4-wave workgroup per CU -> one wave per SIMD
#define RUN(K, N, name)                                                                                         \
  {                                                                                                             \
    float ms = timeit([&] { hipLaunchKernelGGL(K<N>, dim3(grid), dim3(256), 0, 0, out, 1.0000001, 1e-9, iters); }); \
    printf("%s NACC=%d : %.1f cycles per instruction per SIMD @2.4 GHz (%.3f ms)\n", name, N, ms * 1e-3 * 2.4e9 / ((double)iters * N), ms); \
  }
  RUN(k_mfma16, 1, "mfma_f64_16x16x4 ");
  RUN(k_mfma16, 2, "mfma_f64_16x16x4 ");
  RUN(k_mfma16, 4, "mfma_f64_16x16x4 ");
  RUN(k_mfma4, 1, "mfma_f64_4x4x4_4b");
  RUN(k_mfma4, 2, "mfma_f64_4x4x4_4b");
  RUN(k_mfma4, 3, "mfma_f64_4x4x4_4b");
  RUN(k_mfma4, 4, "mfma_f64_4x4x4_4b");
  RUN(k_mfma4, 8, "mfma_f64_4x4x4_4b");
  return 0;
}
