// Micro-benchmark 3: the HYBRID K-step considered for c3p_regd.hip -- per wave 16 columns as one v_mfma_f64_16x16x4 tile
// (its D layout is a valid B operand, no data movement) + 4 columns as one v_mfma_f64_4x4x4_4b block (B broadcast by
// ds_swizzle): 15 big + 15 small MFMAs (= 75 small ones in matrix-pipe time), 5 ds_read_b128, 4 swizzles, 7 adds per step.
// hipcc --offload-arch=gfx950 -O3 -w -o ubench_regd_loop3 ubench_regd_loop3.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <utility>
extern __shared__ __attribute__((aligned(16))) double lds[];
typedef double d4 __attribute__((ext_vector_type(4)));
template <typename F, int... Is>
__device__ __forceinline__ void sf_impl(F&& f, std::integer_sequence<int, Is...>) { (f(std::integral_constant<int, Is>{}), ...); }
template <int N, typename F>
__device__ __forceinline__ void sf(F&& f) { sf_impl(f, std::make_integer_sequence<int, N>{}); }
template <int BP>
__device__ __forceinline__ double bc(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_ds_swizzle(lo, 0x13 | (BP << 7));
  hi = __builtin_amdgcn_ds_swizzle(hi, 0x13 | (BP << 7));
  return __hiloint2double(hi, lo);
}
// MODE bit0: operand traffic on (ds_read A frags, swizzles, adds); bit1: pinned order
template <int MODE>
__global__ void __launch_bounds__(256, 1) k_loop(double* out, long long* cyc, double a0, double b0, int iters) {
  constexpr int N = 5;
  d4 bP[N], bQ[N], bR[N];
  double sP[N], sQ[N], sR[N];
  const int lane = threadIdx.x & 63;
  const double2* img = reinterpret_cast<const double2*>(lds);
  for (int e = threadIdx.x; e < 81 * 82 * 2; e += 256) lds[e] = 1e-3 * e;
  __syncthreads();
#pragma unroll
  for (int i = 0; i < N; ++i) {
    bP[i] = bQ[i] = bR[i] = (d4){0, 0, 0, 0};
    sP[i] = sQ[i] = sR[i] = 0.0;
  }
  // right operand: big tile rows (4 regs per row group) + small tile (1 reg per row group); one row group used per step
  d4 Rbr = {b0, b0 + 1e-7, b0 + 2e-7, b0 + 3e-7}, Rbi = {b0, b0 - 1e-7, b0 - 2e-7, b0 - 3e-7};
  double Rsr = b0 + 1e-8 * lane, Rsi = b0 - 1e-8 * lane;
  const int q = lane >> 4;
  const double2* pa = img + (lane & 15) * 82 + q;
  double2 aC[N], aN[N];
#pragma unroll
  for (int i = 0; i < N; ++i) aC[i] = aN[i] = make_double2(a0 + 1e-7 * i, a0 - 1e-7 * i);
  double brs = Rsr, bis = Rsi, bss = brs + bis;
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
    sf<4>([&](auto v_) {
      constexpr int v = decltype(v_)::value;
      const double bbr = Rbr[v], bbi = Rbi[v], bbs = (MODE & 1) ? bbr + bbi : bbr;
      double brsN = brs, bisN = bis, bssN = bss;
      sf<N>([&](auto i_) {
        constexpr int i = decltype(i_)::value;
        const double as = (MODE & 1) ? aC[i].x + aC[i].y : aC[i].x;
        if (MODE & 4) {  // grouped: the three big ones, then the three small ones
          bP[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(aC[i].x, bbr, bP[i], 0, 0, 0);
          bQ[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(aC[i].y, bbi, bQ[i], 0, 0, 0);
          bR[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(as, bbs, bR[i], 0, 0, 0);
          sP[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(aC[i].x, brs, sP[i], 0, 0, 0);
          sQ[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(aC[i].y, bis, sQ[i], 0, 0, 0);
          sR[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(as, bss, sR[i], 0, 0, 0);
        } else {
        bP[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(aC[i].x, bbr, bP[i], 0, 0, 0);
        sP[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(aC[i].x, brs, sP[i], 0, 0, 0);
        bQ[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(aC[i].y, bbi, bQ[i], 0, 0, 0);
        sQ[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(aC[i].y, bis, sQ[i], 0, 0, 0);
        bR[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(as, bbs, bR[i], 0, 0, 0);
        sR[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(as, bss, sR[i], 0, 0, 0);
        }
        if (MODE & 1) {
          aN[i] = pa[16 * i * 82 + 4 * ((it * 4 + v) & 15)];
          if constexpr (i == 1) brsN = bc<(v + 1) & 3>(Rsr);
          if constexpr (i == 2) bisN = bc<(v + 1) & 3>(Rsi);
          if constexpr (i == 3) bssN = brsN + bisN;
        }
        if (MODE & 2) __builtin_amdgcn_sched_barrier(0);
      });
#pragma unroll
      for (int i = 0; i < N; ++i) aC[i] = aN[i];
      brs = brsN;
      bis = bisN;
      bss = bssN;
    });
  }
  const long long t1 = clock64();
  double s = 0;
#pragma unroll
  for (int i = 0; i < N; ++i) s += bP[i][0] + bQ[i][1] + bR[i][2] + bP[i][3] + sP[i] + sQ[i] + sR[i];
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
  const int iters = 5000;
  const size_t ldsb = 81 * 82 * 16;
#define RUN(M)                                                                                                      \
  {                                                                                                                 \
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_loop<M>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb); \
    float ms = timeit([&] { hipLaunchKernelGGL(k_loop<M>, dim3(256), dim3(256), ldsb, 0, out, cyc, 1.0000001, 1e-9, iters); }); \
    long long c;                                                                                                    \
    hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);                                                                   \
    printf("mode %d: %.3f ms, %.2f clock64 ticks per small-MFMA equivalent (4 steps x 75 per iteration)\n", M, ms, (double)c / (300.0 * iters)); \
  }
  RUN(0);
  RUN(1);
  RUN(2);
  RUN(3);
  RUN(6);
  RUN(7);
  return 0;
}
