// Micro-benchmark of the inner K-step of c3p_regd.hip: 75 independent v_mfma_f64_4x4x4_4b per step (5 x 5 tiles x 3
// real products), one wave per SIMD, with the operand traffic of the real loop switched on piece by piece.
// hipcc --offload-arch=gfx950 -O3 -o ubench_regd_loop ubench_regd_loop.hip
#include <hip/hip_runtime.h>
#include <cstdio>

extern __shared__ __attribute__((aligned(16))) double lds[];

template <int BP>
__device__ __forceinline__ double bc(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_ds_swizzle(lo, 0x13 | (BP << 7));
  hi = __builtin_amdgcn_ds_swizzle(hi, 0x13 | (BP << 7));
  return __hiloint2double(hi, lo);
}

// MODE bit 0: B operands through ds_swizzle each step; bit 1: A operands through ds_read_b128 each step;
// bit 2: the sums (as, bs) by v_add_f64 each step
template <int MODE>
__global__ void __launch_bounds__(256, 1) k_loop(double* out, long long* cyc, double a0, double b0, int iters) {
  constexpr int N = 5;
  double aP[N][N], aQ[N][N], aR[N][N];
  double Rr[N], Ri[N];
  const int lane = threadIdx.x & 63;
  const double2* img = reinterpret_cast<const double2*>(lds);
  for (int e = threadIdx.x; e < 81 * 82 * 2; e += 256) lds[e] = 1e-3 * e;
  __syncthreads();
#pragma unroll
  for (int i = 0; i < N; ++i) {
    Rr[i] = b0 + 1e-7 * (lane + i);
    Ri[i] = b0 - 1e-7 * (lane + i);
#pragma unroll
    for (int j = 0; j < N; ++j) aP[i][j] = aQ[i][j] = aR[i][j] = 0.0;
  }
  const int q = lane >> 4, b = (lane >> 2) & 3, p = lane & 3;
  const double2* pa = img + (4 * b + p) * 82 + q;
  double2 aC[N];
#pragma unroll
  for (int i = 0; i < N; ++i) aC[i] = (MODE & 2) ? pa[16 * i * 82] : make_double2(a0 + 1e-7 * i, a0 - 1e-7 * i);
  double br[N], bi[N];
#pragma unroll
  for (int j = 0; j < N; ++j) {
    br[j] = Rr[j];
    bi[j] = Ri[j];
  }
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
    double2 aN[N];
    double brN[N], biN[N];
#pragma unroll
    for (int i = 0; i < N; ++i) aN[i] = (MODE & 2) ? pa[16 * i * 82 + 4 * ((it & 15) + 1)] : aC[i];
#pragma unroll
    for (int j = 0; j < N; ++j) {
      brN[j] = (MODE & 1) ? bc<1>(Rr[j]) : br[j];
      biN[j] = (MODE & 1) ? bc<1>(Ri[j]) : bi[j];
    }
    double bs[N], as[N];
#pragma unroll
    for (int j = 0; j < N; ++j) bs[j] = (MODE & 4) ? br[j] + bi[j] : br[j];
#pragma unroll
    for (int i = 0; i < N; ++i) {
      as[i] = (MODE & 4) ? aC[i].x + aC[i].y : aC[i].x;
#pragma unroll
      for (int j = 0; j < N; ++j) {
        aP[i][j] = __builtin_amdgcn_mfma_f64_4x4x4f64(aC[i].x, br[j], aP[i][j], 0, 0, 0);
        aQ[i][j] = __builtin_amdgcn_mfma_f64_4x4x4f64(aC[i].y, bi[j], aQ[i][j], 0, 0, 0);
        aR[i][j] = __builtin_amdgcn_mfma_f64_4x4x4f64(as[i], bs[j], aR[i][j], 0, 0, 0);
      }
    }
#pragma unroll
    for (int i = 0; i < N; ++i) aC[i] = aN[i];
#pragma unroll
    for (int j = 0; j < N; ++j) {
      br[j] = brN[j];
      bi[j] = biN[j];
    }
  }
  const long long t1 = clock64();
  double s = 0;
#pragma unroll
  for (int i = 0; i < N; ++i)
#pragma unroll
    for (int j = 0; j < N; ++j) s += aP[i][j] + aQ[i][j] + aR[i][j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <typename F>
float timeit(F launch) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int i = 0; i < 2; ++i) launch();
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
  long long* cyc;
  hipMalloc(&out, 256 * 256 * sizeof(double));
  hipMalloc(&cyc, 8);
  const int iters = 20000;
  const size_t ldsb = 81 * 82 * 16;
#define RUN(M)                                                                                                      \
  {                                                                                                                 \
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_loop<M>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb); \
    float ms = timeit([&] { hipLaunchKernelGGL(k_loop<M>, dim3(256), dim3(256), ldsb, 0, out, cyc, 1.0000001, 1e-9, iters); }); \
    long long c;                                                                                                    \
    hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);                                                                   \
    printf("mode %d: %.3f ms, %.2f clock64 ticks per MFMA, %.2f ns per MFMA (x 2.4 = %.2f cycles @2.4GHz)\n", M, ms, \
           (double)c / (75.0 * iters), ms * 1e6 / (75.0 * iters), ms * 1e6 / (75.0 * iters) * 2.4);                 \
  }
  RUN(0);
  RUN(1);
  RUN(2);
  RUN(3);
  RUN(4);
  RUN(7);
  return 0;
}
