// Micro-benchmark (round 6): issue cost of the fp64 vector instructions the small-D kernels use, 2 waves per SIMD, 32 independent chains per wave
// (cycles per wave instruction on one SIMD at 2.4 GHz; v_mfma_f64_4x4x4_4b measured the same way for the clock reference)
// build + run:  hipcc --offload-arch=gfx950 -O3 tools/ubench_valu64.hip -o tools/probe/ubench_valu64 && tools/probe/ubench_valu64
#include <hip/hip_runtime.h>
#include <cstdio>

template <int KIND>
__global__ void k(double* out, double a, double b, int iters) {
  constexpr int N = 32;
  double x[N];
#pragma unroll
  for (int i = 0; i < N; ++i) x[i] = threadIdx.x * 1e-3 + i;
  double av = a + threadIdx.x * 1e-9;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < N; ++i) {
      if (KIND == 0) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x[i]) : "v"(av), "v"(b));
      if (KIND == 1) asm volatile("v_add_f64 %0, %0, %1" : "+v"(x[i]) : "v"(b));
      if (KIND == 2) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x[i]) : "v"(av));
      if (KIND == 3) asm volatile("v_fma_f64 %0, %0, 1.0, %1" : "+v"(x[i]) : "v"(b));
      if (KIND == 4) asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(x[i]) : "v"(av), "v"(b));
      if (KIND == 5) asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0" : "+v"(x[i]) : "v"(av), "v"(b));
      if (KIND == 6) asm volatile("v_add_f64 %0, %0, %0" : "+v"(x[i]));
      if (KIND == 7) asm volatile("v_fma_f64 %0, %0, %0, %0" : "+v"(x[i]));
      if (KIND == 8) asm volatile("v_mov_b64 %0, %1" : "=v"(x[i]) : "v"(x[(i + 1) % N]));
      if (KIND == 9) asm volatile("v_add_f64 %0, -%0, %1" : "+v"(x[i]) : "v"(b));
      if (KIND == 10) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(x[i]) : "v"(av));
      if (KIND == 11) asm volatile("v_add_f64 %0, %1, %2" : "=v"(x[i]) : "v"(x[(i + 1) % N]), "v"(x[(i + 2) % N]));
    }
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < N; ++i) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
static const char* names[] = {"v_fma_f64 x,a,b (3 regs)", "v_add_f64 x,b", "v_mul_f64 x,a", "v_fma_f64 x,1.0,b", "v_fmac_f64 a,b", "v_mfma_f64_4x4x4_4b", "v_add_f64 x,x", "v_fma_f64 x,x,x", "v_mov_b64", "v_add_f64 -x,b", "v_fma_f64 x,a,x", "v_add_f64 y,z (3 regs)"};
template <int KIND>
void run(double* out, int wpc) {
  const int iters = 20000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(64 * wpc), 0, 0, out, 1.0000001, 1e-9, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(64 * wpc), 0, 0, out, 1.0000001, 1e-9, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  printf("%-28s waves/SIMD=%d : %6.2f cycles per wave instruction\n", names[KIND], wpc / 4, ms * 1e-3 * 2.4e9 / iters / 32 / (wpc / 4));
}
int main() {
  double* out;
  hipMalloc(&out, sizeof(double) * 256 * 1024);
  run<5>(out, 8); run<5>(out, 8);
  run<0>(out, 8); run<1>(out, 8); run<2>(out, 8); run<3>(out, 8); run<4>(out, 8); run<6>(out, 8); run<7>(out, 8); run<8>(out, 8); run<9>(out, 8); run<10>(out, 8); run<11>(out, 8);
  run<0>(out, 4); run<1>(out, 4); run<4>(out, 4);
  run<0>(out, 16); run<1>(out, 16); run<4>(out, 16);
  return 0;
}
