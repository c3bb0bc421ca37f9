// Micro-benchmark: fp64 issue rates on gfx950 (v_fma_f64, v_mfma_f64_16x16x4, v_mfma_f64_4x4x4_4b)
// hipcc --offload-arch=gfx950 -O3 -o ubench_f64 ubench_f64.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));

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

int main() {
  double* out;
  hipMalloc(&out, sizeof(double) * 256 * 1024 * 64);
  const int iters = 20000;
  for (int wpc : {4, 8, 16}) {  // waves per CU
    const int threads = 64 * wpc / 4 * 4;  // one block per CU of wpc waves
    const int grid = 256;
    {
      float ms = timeit([&] { hipLaunchKernelGGL(k_fma<8>, dim3(grid), dim3(64 * wpc), 0, 0, out, 1.0000001, 1e-9, iters); });
      double flops = 2.0 * 8 * iters * 64.0 * wpc * grid;
      printf("v_fma_f64        waves/CU=%2d : %8.2f TFLOP/s (%.3f ms)\n", wpc, flops / ms * 1e-9, ms);
    }
    {
      float ms = timeit([&] { hipLaunchKernelGGL(k_mfma16<4>, dim3(grid), dim3(64 * wpc), 0, 0, out, 1.0000001, 1e-9, iters); });
      double flops = 2.0 * 16 * 16 * 4 * 4 * iters * (double)wpc * grid;
      printf("mfma_f64_16x16x4 waves/CU=%2d : %8.2f TFLOP/s (%.3f ms)  cycles/instr/SIMD @2.4GHz=%.1f\n", wpc, flops / ms * 1e-9, ms,
             ms * 1e-3 * 2.4e9 / (4.0 * iters * wpc / 4.0));
    }
    {
      float ms = timeit([&] { hipLaunchKernelGGL(k_mfma4<8>, dim3(grid), dim3(64 * wpc), 0, 0, out, 1.0000001, 1e-9, iters); });
      double flops = 2.0 * 4 * 4 * 4 * 4 * 8 * iters * (double)wpc * grid;
      printf("mfma_f64_4x4x4_4b waves/CU=%2d : %8.2f TFLOP/s (%.3f ms)  cycles/instr/SIMD @2.4GHz=%.1f\n", wpc, flops / ms * 1e-9, ms,
             ms * 1e-3 * 2.4e9 / (8.0 * iters * wpc / 4.0));
    }
    (void)threads;
  }
  return 0;
}
