// Micro-benchmark + functional probe: v_fmac_f64_dpp with row_newbcast on gfx950 (the broadcast-fused FMA
// the ODE kernels are built on).  hipcc --offload-arch=gfx950 -O3 -o ubench_dpp ubench_dpp.hip
//  (1) semantics: D = bcast_row(S0, lane J) * S1 + D for every J, neg modifier on S1;
//  (2) issue rate against plain v_fma_f64, 1 / 2 / 4 waves per SIMD;
//  (3) hazard probe: a VALU write of the DPP source directly in front of the DPP read, with and without s_nop.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>

template <int J>
__device__ __forceinline__ void fmac_bc(double& acc, double src, double own) {
  asm("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(own), "n"(J));
}
template <int J>
__device__ __forceinline__ void fmac_bc_neg(double& acc, double src, double own) {
  asm("s_nop 1\n\tv_fmac_f64_dpp %0, %1, -%2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(own), "n"(J));
}

template <int J>
__device__ void probe_one(const double* src, const double* own, double* out, int l) {
  double acc = 0.5, accn = 0.25;
  fmac_bc<J>(acc, src[l], own[l]);
  fmac_bc_neg<J>(accn, src[l], own[l]);
  out[J * 64 + l] = acc;
  out[(16 + J) * 64 + l] = accn;
}
__global__ void k_probe(const double* src, const double* own, double* out) {
  const int l = threadIdx.x;
  probe_one<0>(src, own, out, l); probe_one<1>(src, own, out, l); probe_one<2>(src, own, out, l); probe_one<3>(src, own, out, l);
  probe_one<4>(src, own, out, l); probe_one<5>(src, own, out, l); probe_one<6>(src, own, out, l); probe_one<7>(src, own, out, l);
  probe_one<8>(src, own, out, l); probe_one<9>(src, own, out, l); probe_one<10>(src, own, out, l); probe_one<11>(src, own, out, l);
  probe_one<12>(src, own, out, l); probe_one<13>(src, own, out, l); probe_one<14>(src, own, out, l); probe_one<15>(src, own, out, l);
}

// hazard probe: the source is produced by a VALU instruction immediately before the DPP read
template <int NOPS>
__global__ void k_hazard(const double* src, const double* own, double* out, int iters) {
  const int l = threadIdx.x;
  double s = src[l], o = own[l], acc = 0.0;
  for (int it = 0; it < iters; ++it) {
    if constexpr (NOPS == 0)
      asm volatile("v_add_f64 %1, %1, 1.0\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "+v"(acc), "+v"(s) : "v"(o));
    else if constexpr (NOPS == 1)
      asm volatile("v_add_f64 %1, %1, 1.0\n\ts_nop 0\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "+v"(acc), "+v"(s) : "v"(o));
    else
      asm volatile("v_add_f64 %1, %1, 1.0\n\ts_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "+v"(acc), "+v"(s) : "v"(o));
  }
  out[blockIdx.x * 64 + l] = acc;
}

template <bool DPP>
__global__ void k_rate(double* out, double a, int iters) {
  double acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = threadIdx.x * 1e-3 + i;
  double s = a + threadIdx.x * 1e-9, o = 1e-9;
  for (int it = 0; it < iters; ++it) {
    if constexpr (DPP) {
      asm volatile(
          "v_fmac_f64_dpp %0, %8, %9 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %1, %8, %9 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %2, %8, %9 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %3, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %4, %8, %9 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %5, %8, %9 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %6, %8, %9 row_newbcast:6 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f64_dpp %7, %8, %9 row_newbcast:7 row_mask:0xf bank_mask:0xf"
          : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7])
          : "v"(s), "v"(o));
    } else {
      asm volatile(
          "v_fmac_f64_e32 %0, %8, %9\n\tv_fmac_f64_e32 %1, %8, %9\n\tv_fmac_f64_e32 %2, %8, %9\n\tv_fmac_f64_e32 %3, %8, %9\n\t"
          "v_fmac_f64_e32 %4, %8, %9\n\tv_fmac_f64_e32 %5, %8, %9\n\tv_fmac_f64_e32 %6, %8, %9\n\tv_fmac_f64_e32 %7, %8, %9"
          : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7])
          : "v"(s), "v"(o));
    }
  }
  double t = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) t += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = t;
}

// dependent chain: one accumulator
template <bool DPP>
__global__ void k_dep(double* out, double a, int iters) {
  double acc = threadIdx.x * 1e-3;
  double s = a + threadIdx.x * 1e-9, o = 1e-9;
  for (int it = 0; it < iters; ++it) {
    if constexpr (DPP)
      asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
                   "v_fmac_f64_dpp %0, %1, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:6 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(s), "v"(o));
    else
      asm volatile("v_fmac_f64_e32 %0, %1, %2\n\tv_fmac_f64_e32 %0, %1, %2\n\tv_fmac_f64_e32 %0, %1, %2\n\tv_fmac_f64_e32 %0, %1, %2" : "+v"(acc) : "v"(s), "v"(o));
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
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
  double *src, *own, *out;
  hipMalloc(&src, 64 * 8);
  hipMalloc(&own, 64 * 8);
  hipMalloc(&out, sizeof(double) * 256 * 1024 * 16);
  std::vector<double> hs(64), ho(64), hout(32 * 64);
  for (int l = 0; l < 64; ++l) { hs[l] = 1.0 + l; ho[l] = 100.0 + 0.5 * l; }
  hipMemcpy(src, hs.data(), 64 * 8, hipMemcpyHostToDevice);
  hipMemcpy(own, ho.data(), 64 * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, 0, src, own, out);
  hipMemcpy(hout.data(), out, 32 * 64 * 8, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int J = 0; J < 16; ++J)
    for (int l = 0; l < 64; ++l) {
      const double bc = hs[(l / 16) * 16 + J];
      if (hout[J * 64 + l] != 0.5 + bc * ho[l]) ++bad;
      if (hout[(16 + J) * 64 + l] != 0.25 - bc * ho[l]) ++bad;
    }
  printf("probe: v_fmac_f64_dpp row_newbcast semantics mismatches = %d of 2048\n", bad);

  // hazard probe
  for (int nops = 0; nops < 3; ++nops) {
    const int iters = 1000;
    if (nops == 0) hipLaunchKernelGGL(k_hazard<0>, dim3(256), dim3(64), 0, 0, src, own, out, iters);
    if (nops == 1) hipLaunchKernelGGL(k_hazard<1>, dim3(256), dim3(64), 0, 0, src, own, out, iters);
    if (nops == 2) hipLaunchKernelGGL(k_hazard<2>, dim3(256), dim3(64), 0, 0, src, own, out, iters);
    std::vector<double> h(256 * 64);
    hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);
    int wrong = 0;
    for (int blk = 0; blk < 256; ++blk)
      for (int l = 0; l < 64; ++l) {
        double s = hs[(l / 16) * 16 + 5], acc = 0;
        for (int it = 0; it < iters; ++it) { s += 1.0; acc = fma(s, ho[l], acc); }
        if (h[blk * 64 + l] != acc) ++wrong;
      }
    printf("hazard probe: VALU write -> %d wait states -> DPP read: wrong lanes = %d of %d\n", nops == 0 ? 0 : nops, wrong, 256 * 64);
  }

  const int iters = 20000;
  for (int wpc : {4, 8, 16}) {
    const int grid = 256;
    float ms = timeit([&] { hipLaunchKernelGGL(k_rate<false>, dim3(grid), dim3(64 * wpc), 0, 0, out, 1.0000001, iters); });
    double flops = 2.0 * 8 * iters * 64.0 * wpc * grid;
    printf("v_fmac_f64_e32   waves/CU=%2d : %8.2f TFLOP/s (%.3f ms)\n", wpc, flops / ms * 1e-9, ms);
    ms = timeit([&] { hipLaunchKernelGGL(k_rate<true>, dim3(grid), dim3(64 * wpc), 0, 0, out, 1.0000001, iters); });
    printf("v_fmac_f64_dpp   waves/CU=%2d : %8.2f TFLOP/s (%.3f ms)\n", wpc, flops / ms * 1e-9, ms);
  }
  {
    const int grid = 256, wpc = 4;
    float ms = timeit([&] { hipLaunchKernelGGL(k_dep<false>, dim3(grid), dim3(64 * wpc), 0, 0, out, 1.0000001, iters); });
    // cycles per dependent instruction at ~2.4 GHz
    printf("dependent v_fmac_f64_e32 : %.2f ns per instruction (one wave per SIMD)\n", ms * 1e6 / (4.0 * iters));
    ms = timeit([&] { hipLaunchKernelGGL(k_dep<true>, dim3(grid), dim3(64 * wpc), 0, 0, out, 1.0000001, iters); });
    printf("dependent v_fmac_f64_dpp : %.2f ns per instruction (one wave per SIMD)\n", ms * 1e6 / (4.0 * iters));
  }
  return 0;
}
