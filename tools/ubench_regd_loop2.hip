// Micro-benchmark 2 of the inner K-step of c3p_regd.hip: the pinned issue order (3 MFMAs, 1 preparation piece), pieces switched
// on one by one.  hipcc --offload-arch=gfx950 -O3 -w -o ubench_regd_loop2 ubench_regd_loop2.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <utility>
extern __shared__ __attribute__((aligned(16))) double lds[];
template <typename F, int... Is>
__device__ __forceinline__ void sf_impl(F&& f, std::integer_sequence<int, Is...>) { (f(std::integral_constant<int, Is>{}), ...); }
template <int N, typename F>
__device__ __forceinline__ void sf(F&& f) { sf_impl(f, std::make_integer_sequence<int, N>{}); }
template <int S>
__device__ __forceinline__ double rot(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_mov_dpp(lo, 0x120 + 4 * S, 0xf, 0xf, false);
  hi = __builtin_amdgcn_mov_dpp(hi, 0x120 + 4 * S, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
// MODE bit0: DPP rotations, bit1: ds_read_b128 A prefetch, bit2: v_add_f64 sums, bit3: no sched_barrier pinning
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
  double2 aC[N], aN[N];
#pragma unroll
  for (int i = 0; i < N; ++i) aC[i] = aN[i] = make_double2(a0 + 1e-7 * i, a0 - 1e-7 * i);
  double br[N], bi[N], bs[N];
#pragma unroll
  for (int j = 0; j < N; ++j) {
    br[j] = Rr[j];
    bi[j] = Ri[j];
    bs[j] = br[j] + bi[j];
  }
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
    double as[N], brN[N], biN[N], bsN[N];
#pragma unroll
    for (int j = 0; j < N; ++j) {
      brN[j] = br[j];
      biN[j] = bi[j];
      bsN[j] = bs[j];
    }
    const int koff = 4 * (it & 15);
    sf<N>([&](auto i_) {
      constexpr int i = decltype(i_)::value;
      as[i] = (MODE & 4) ? aC[i].x + aC[i].y : aC[i].x;
      sf<N>([&](auto j_) {
        constexpr int j = decltype(j_)::value;
        constexpr int u = i * N + j;
        aP[i][j] = __builtin_amdgcn_mfma_f64_4x4x4f64(aC[i].x, br[j], aP[i][j], 0, 0, 0);
        aQ[i][j] = __builtin_amdgcn_mfma_f64_4x4x4f64(aC[i].y, bi[j], aQ[i][j], 0, 0, 0);
        aR[i][j] = __builtin_amdgcn_mfma_f64_4x4x4f64(as[i], bs[j], aR[i][j], 0, 0, 0);
        if constexpr (u < N) {
          if (MODE & 2) aN[u] = pa[16 * u * 82 + koff];
        } else if constexpr (u < 3 * N) {
          constexpr int j2 = (u - N) >> 1;
          if (MODE & 1) {
            if constexpr (((u - N) & 1) == 0) brN[j2] = rot<1>(Rr[j2]);
            else biN[j2] = rot<1>(Ri[j2]);
          }
        } else if constexpr (u < 4 * N) {
          constexpr int j2 = u - 3 * N;
          if (MODE & 4) bsN[j2] = brN[j2] + biN[j2];
        }
        if (!(MODE & 8)) __builtin_amdgcn_sched_barrier(0);
      });
    });
#pragma unroll
    for (int i = 0; i < N; ++i) aC[i] = aN[i];
#pragma unroll
    for (int j = 0; j < N; ++j) {
      br[j] = brN[j];
      bi[j] = biN[j];
      bs[j] = bsN[j];
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
    printf("mode %2d: %.3f ms, %.2f clock64 ticks per MFMA, %.2f ns per MFMA\n", M, ms, (double)c / (75.0 * iters), ms * 1e6 / (75.0 * iters)); \
  }
  RUN(0);
  RUN(1);
  RUN(2);
  RUN(4);
  RUN(7);
  RUN(8);
  RUN(15);
  return 0;
}
