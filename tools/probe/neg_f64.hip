// Probe: the neg modifiers of v_mfma_f64_4x4x4_4b (blgp bits 0 / 1 / 2 of the builtin negate A / B / C on gfx950).
// hipcc --offload-arch=gfx950 -O2 tools/probe/neg_f64.hip -o tools/probe/neg_f64
#include <hip/hip_runtime.h>
#include <cstdio>
template <int BL>
__global__ void k(const double* a, const double* b, double* o) {
  o[threadIdx.x] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[threadIdx.x], b[threadIdx.x], 1.0, 0, 0, BL);
}
int main() {
  double ha[64], hb[64], ho[64]; for (int i = 0; i < 64; ++i) ha[i] = 0.3 + 0.01 * i, hb[i] = 0.7 - 0.02 * i;
  double *da, *db, *dout; hipMalloc(&da, 512); hipMalloc(&db, 512); hipMalloc(&dout, 512);
  hipMemcpy(da, ha, 512, hipMemcpyHostToDevice); hipMemcpy(db, hb, 512, hipMemcpyHostToDevice);
  double ref[8];
  auto run = [&](int bl) {
    if (bl == 0) hipLaunchKernelGGL(k<0>, dim3(1), dim3(64), 0, 0, da, db, dout);
    if (bl == 1) hipLaunchKernelGGL(k<1>, dim3(1), dim3(64), 0, 0, da, db, dout);
    if (bl == 2) hipLaunchKernelGGL(k<2>, dim3(1), dim3(64), 0, 0, da, db, dout);
    if (bl == 4) hipLaunchKernelGGL(k<4>, dim3(1), dim3(64), 0, 0, da, db, dout);
    hipMemcpy(ho, dout, 512, hipMemcpyDeviceToHost);
    printf("blgp=%d: out[0]=%.6f out[5]=%.6f (c=1: plain = 1 + ab)\n", bl, ho[0], ho[5]);
  };
  run(0); run(1); run(2); run(4);
  return 0;
}
