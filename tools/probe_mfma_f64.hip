// Derive the lane layout of v_mfma_f64_4x4x4_4b_f64 and v_mfma_f64_16x16x4_f64 empirically.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void probe4(double* out) {  // out[la][lb][lane]
  const int lane = threadIdx.x;
  for (int la = 0; la < 64; ++la)
    for (int lb = 0; lb < 64; ++lb) {
      double a = lane == la ? 1.0 : 0.0, b = lane == lb ? 1.0 : 0.0;
      double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
      out[(la * 64 + lb) * 64 + lane] = d;
    }
}
__global__ void probe16(double* out) {  // out[la][lb][lane][4]
  const int lane = threadIdx.x;
  for (int la = 0; la < 64; ++la)
    for (int lb = 0; lb < 64; ++lb) {
      double a = lane == la ? 1.0 : 0.0, b = lane == lb ? 1.0 : 0.0;
      d4 z = {0, 0, 0, 0};
      d4 d = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, z, 0, 0, 0);
      for (int r = 0; r < 4; ++r) out[((la * 64 + lb) * 64 + lane) * 4 + r] = d[r];
    }
}
int main() {
  double* d;
  hipMalloc(&d, sizeof(double) * 64 * 64 * 64 * 4);
  std::vector<double> h(64 * 64 * 64 * 4);
  hipLaunchKernelGGL(probe4, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h.data(), d, sizeof(double) * 64 * 64 * 64, hipMemcpyDeviceToHost);
  printf("== mfma_f64_4x4x4_4b: for each A-lane la: list of (lb -> out lane)\n");
  for (int la = 0; la < 64; ++la) {
    printf("la=%2d:", la);
    for (int lb = 0; lb < 64; ++lb)
      for (int l = 0; l < 64; ++l)
        if (h[(la * 64 + lb) * 64 + l] != 0.0) printf(" (%d->%d)", lb, l);
    printf("\n");
  }
  hipLaunchKernelGGL(probe16, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h.data(), d, sizeof(double) * 64 * 64 * 64 * 4, hipMemcpyDeviceToHost);
  printf("== mfma_f64_16x16x4: la in {0,1,16,17,32,48}: (lb -> lane.reg)\n");
  for (int la : {0, 1, 16, 17, 32, 48}) {
    printf("la=%2d:", la);
    for (int lb = 0; lb < 64; ++lb)
      for (int l = 0; l < 64; ++l)
        for (int r = 0; r < 4; ++r)
          if (h[((la * 64 + lb) * 64 + l) * 4 + r] != 0.0) printf(" (%d->%d.%d)", lb, l, r);
    printf("\n");
  }
  return 0;
}
