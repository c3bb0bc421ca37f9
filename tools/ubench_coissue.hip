// Micro-benchmark (round 6): what overlaps with the fp64 matrix instruction on one SIMD?
// NM x v_mfma_f64_4x4x4_4b + NO x "other" instruction per iteration, 2 waves per SIMD (8 per CU), independent chains.
// other = 0: v_fma_f64, 1: v_mov_b32 dpp (quad_perm), 2: v_xor_b32 (integer), 3: v_add_f64, 4: ds_bpermute_b32, 5: v_cndmask/v_mov pair (v_mov_b32)
// Printed: cycles per iteration per SIMD at 2.4 GHz next to the serial sum NM*16 + NO*4 (x2 waves) and the max of the two.
// build + run:  hipcc --offload-arch=gfx950 -O3 tools/ubench_coissue.hip -o tools/probe/ubench_coissue && tools/probe/ubench_coissue
#include <hip/hip_runtime.h>
#include <cstdio>

template <int NM, int NO, int KIND>
__global__ void k_mix(double* out, double a, double b, int iters) {
  double macc[NM > 0 ? NM : 1];
  double facc[NO > 0 ? NO : 1];
  int iacc[NO > 0 ? NO : 1];
#pragma unroll
  for (int i = 0; i < NM; ++i) macc[i] = 0;
#pragma unroll
  for (int i = 0; i < NO; ++i) facc[i] = threadIdx.x * 1e-3 + i, iacc[i] = threadIdx.x + i;
  double av = a + threadIdx.x * 1e-6, bv = b;
  const int perm = ((threadIdx.x & 63) ^ 17) << 2;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < (NM > NO ? NM : NO); ++i) {
      if (i < NM) macc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(av, bv, macc[i], 0, 0, 0);
      if (i < NO) {
        if (KIND == 0) facc[i] = fma(facc[i], a, b);
        if (KIND == 1) iacc[i] = __builtin_amdgcn_mov_dpp(iacc[i], 0xB1, 0xf, 0xf, true);
        if (KIND == 2) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(iacc[i]) : "v"(perm));
        if (KIND == 3) facc[i] = facc[i] + b;
        if (KIND == 4) iacc[i] = __builtin_amdgcn_ds_bpermute(perm, iacc[i]);
        if (KIND == 5) asm volatile("v_mov_b32 %0, %1" : "=v"(iacc[i]) : "v"(iacc[(i + 1) % NO]));
      }
    }
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < NM; ++i) s += macc[i];
#pragma unroll
  for (int i = 0; i < NO; ++i) s += facc[i] + iacc[i];
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

static const char* kinds[] = {"v_fma_f64", "v_mov_dpp", "v_xor_b32", "v_add_f64", "ds_bpermute", "v_mov_b32"};
#define RUN(NM, NO, KIND, WPC)                                                                                              \
  {                                                                                                                         \
    float ms = timeit([&] { hipLaunchKernelGGL((k_mix<NM, NO, KIND>), dim3(256), dim3(64 * WPC), 0, 0, out, 1.0000001, 1e-9, iters); }); \
    double cyc = ms * 1e-3 * 2.4e9 / iters;                                                                                 \
    int wps = WPC / 4;                                                                                                      \
    printf("mfma=%2d + %2d x %-11s waves/SIMD=%d : %7.1f cycles/iter/SIMD  (serial sum %5d, max %5d)\n", NM, NO, kinds[KIND], wps, cyc, \
           wps * (NM * 16 + NO * 4), wps * (NM * 16 > NO * 4 ? NM * 16 : NO * 4));                                          \
  }

int main() {
  double* out;
  hipMalloc(&out, sizeof(double) * 256 * 1024);
  const int iters = 20000;
  { float ms = timeit([&] { hipLaunchKernelGGL((k_mix<8, 0, 0>), dim3(256), dim3(512), 0, 0, out, 1.0000001, 1e-9, 200000); }); (void)ms; }  // clock ramp
  RUN(8, 0, 0, 8)
  RUN(0, 32, 0, 8) RUN(0, 32, 1, 8) RUN(0, 32, 2, 8) RUN(0, 32, 3, 8) RUN(0, 32, 4, 8) RUN(0, 32, 5, 8)
  RUN(8, 8, 0, 8) RUN(8, 16, 0, 8) RUN(8, 32, 0, 8)
  RUN(8, 8, 1, 8) RUN(8, 16, 1, 8) RUN(8, 32, 1, 8)
  RUN(8, 8, 2, 8) RUN(8, 16, 2, 8) RUN(8, 32, 2, 8)
  RUN(8, 8, 3, 8) RUN(8, 16, 3, 8) RUN(8, 32, 3, 8)
  RUN(8, 8, 4, 8) RUN(8, 16, 4, 8) RUN(8, 32, 4, 8)
  RUN(8, 8, 5, 8) RUN(8, 16, 5, 8) RUN(8, 32, 5, 8)
  RUN(8, 32, 1, 4) RUN(8, 32, 2, 4) RUN(8, 32, 0, 4)
  RUN(8, 32, 1, 12) RUN(8, 32, 0, 12)
  return 0;
}
