// A/B of the "8 + 1 split" for the 9 x 9 real symmetric products of the small-D real path: one
// dependent chain of squarings W <- W W per MFMA block (= per chain, four per wave), in the register layout of
// smalld_chain_kernel<9> (c3p_smalld.hip: mm_sym + sym_fill), in two forms:
//   FORM 0 (production): 12 x 12 zero-padded tiles -- the 6 tiles on and above the diagonal x 2 K-steps = 12 MFMAs
//            (the tiles (0,2), (1,2), (2,2) carry 4, 4 and 1 of 16 valid elements), the k = 8 term as a rank-1 update on the
//            vector unit (6 FMAs, 3 quad broadcasts, 3 lane swaps), one mirrored lower tile;
//   FORM 1 (8 + 1 split): the 8 x 8 core on the matrix cores (3 tiles x 2 K-steps = 6 MFMAs, rank-1 term 3 FMAs), the
//            border column as two replicated 8-vectors and the corner as a scalar on the vector unit: column = core x vector
//            (4 FMAs + a quad reduction) + vector x corner, its transposed replica by one lane swap per tile row, corner =
//            dot product (2 FMAs + a quad reduction).
// Both forms compute the same numbers (checked), run R dependent products, and are timed per wave with wall_clock64 at
// 1 and 2 waves per SIMD.        hipcc --offload-arch=gfx950 -O3 tools/ubench_sym9.hip -o tools/ubench_sym9 && tools/ubench_sym9
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

__device__ __forceinline__ double mfma4(double a, double b, double c) { return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0); }
template <int CTRL>
__device__ __forceinline__ double dpp(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xf, 0xf, true);
  hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double quad_sum(double v) {
  v += dpp<0xB1>(v);  // quad_perm [1,0,3,2]
  v += dpp<0x4E>(v);  // quad_perm [2,3,0,1]
  return v;
}

// W[ch][9][9] row-major in, W^(2^R) * scale^(2^R - 1) out; cycles[wave] = wall-clock ticks of the loop
template <int FORM>
__global__ void __launch_bounds__(64) sym9_kernel(const double* Win, double* Wout, long long* ticks, int R, double scale) {
  const int lane = threadIdx.x;
  const int c = lane & 3, blk = (lane >> 2) & 3, r = lane >> 4;
  const long ch = (long)blockIdx.x * 4 + blk;
  const double* W = Win + ch * 81;
  const int swap_lane = 16 * c + 4 * blk + r;  // lane (c, r) of the same block
  const int col0_lane = 16 * c + 4 * blk;      // lane (c, 0)
  auto elem = [&](int i, int j) -> double { return (i < 9 && j < 9) ? W[i * 9 + j] : 0.0; };
  long long t0 = 0, t1 = 0;
  if constexpr (FORM == 0) {
    double m[3][3];
#pragma unroll
    for (int I = 0; I < 3; ++I)
#pragma unroll
      for (int J = 0; J < 3; ++J) m[I][J] = elem(4 * I + r, 4 * J + c);
    t0 = wall_clock64();
    for (int it = 0; it < R; ++it) {
      double acc[3][3];
#pragma unroll
      for (int I = 0; I < 3; ++I)
#pragma unroll
        for (int J = I; J < 3; ++J) acc[I][J] = 0.0;
#pragma unroll
      for (int K = 0; K < 2; ++K)
#pragma unroll
        for (int I = 0; I < 3; ++I)
#pragma unroll
          for (int J = I; J < 3; ++J) acc[I][J] = mfma4(m[K][I], m[K][J], acc[I][J]);
      double av[3], bv[3];
#pragma unroll
      for (int I = 0; I < 3; ++I) {
        av[I] = dpp<0x00>(m[I][2]);  // quad broadcast of column 0
        bv[I] = __shfl(m[I][2], col0_lane);
      }
#pragma unroll
      for (int I = 0; I < 3; ++I)
#pragma unroll
        for (int J = I; J < 3; ++J) m[I][J] = scale * fma(av[I], bv[J], acc[I][J]);
      m[1][0] = __shfl(m[0][1], swap_lane);
    }
    t1 = wall_clock64();
    m[2][0] = __shfl(m[0][2], swap_lane);
    m[2][1] = __shfl(m[1][2], swap_lane);
#pragma unroll
    for (int I = 0; I < 3; ++I)
#pragma unroll
      for (int J = 0; J < 3; ++J)
        if (4 * I + r < 9 && 4 * J + c < 9) Wout[ch * 81 + (4 * I + r) * 9 + 4 * J + c] = m[I][J];
  } else {
    double m[2][2], vr[2], vc[2], s;
#pragma unroll
    for (int I = 0; I < 2; ++I) {
#pragma unroll
      for (int J = 0; J < 2; ++J) m[I][J] = elem(4 * I + r, 4 * J + c);
      vr[I] = elem(4 * I + r, 8);
      vc[I] = elem(4 * I + c, 8);
    }
    s = elem(8, 8);
    t0 = wall_clock64();
    for (int it = 0; it < R; ++it) {
      double acc[2][2];
      acc[0][0] = acc[0][1] = acc[1][1] = 0.0;
#pragma unroll
      for (int K = 0; K < 2; ++K) {
        acc[0][0] = mfma4(m[K][0], m[K][0], acc[0][0]);
        acc[0][1] = mfma4(m[K][0], m[K][1], acc[0][1]);
        acc[1][1] = mfma4(m[K][1], m[K][1], acc[1][1]);
      }
      // border column (rows r of tile row I, replicated over c) and corner
      double t[2], cs;
#pragma unroll
      for (int I = 0; I < 2; ++I) t[I] = quad_sum(fma(m[I][1], vc[1], m[I][0] * vc[0]));
      cs = quad_sum(fma(vc[1], vc[1], vc[0] * vc[0]));
      cs = scale * fma(s, s, cs);
#pragma unroll
      for (int I = 0; I < 2; ++I) t[I] = scale * fma(vr[I], s, t[I]);
      // core: the k = 8 term
      const double n00 = scale * fma(vr[0], vc[0], acc[0][0]);
      const double n01 = scale * fma(vr[0], vc[1], acc[0][1]);
      const double n11 = scale * fma(vr[1], vc[1], acc[1][1]);
      m[0][0] = n00, m[0][1] = n01, m[1][1] = n11;
      m[1][0] = __shfl(n01, swap_lane);
      vr[0] = t[0], vr[1] = t[1];
      vc[0] = __shfl(t[0], col0_lane);  // lane (r, c) <- lane (c, .): the value of row 4 J + c
      vc[1] = __shfl(t[1], col0_lane);
      s = cs;
    }
    t1 = wall_clock64();
#pragma unroll
    for (int I = 0; I < 2; ++I) {
#pragma unroll
      for (int J = 0; J < 2; ++J) Wout[ch * 81 + (4 * I + r) * 9 + 4 * J + c] = m[I][J];
      if (c == 0) {
        Wout[ch * 81 + (4 * I + r) * 9 + 8] = vr[I];
        Wout[ch * 81 + 8 * 9 + 4 * I + r] = vr[I];
      }
    }
    if (r == 0 && c == 0) Wout[ch * 81 + 80] = s;
  }
  if (lane == 0) ticks[blockIdx.x] = t1 - t0;
}

int main() {
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount;
  const int maxw = 2;
  const long nblk_max = (long)cus * 4 * maxw;
  const long nch = nblk_max * 4;
  std::vector<double> h(nch * 81);
  srand(7);
  for (long ch = 0; ch < nch; ++ch)
    for (int i = 0; i < 9; ++i)
      for (int j = i; j < 9; ++j) {
        const double v = (rand() / (double)RAND_MAX - 0.5) * 0.5;
        h[ch * 81 + i * 9 + j] = h[ch * 81 + j * 9 + i] = v;
      }
  double *din, *dout0, *dout1;
  long long* dt;
  hipMalloc(&din, h.size() * 8);
  hipMalloc(&dout0, h.size() * 8);
  hipMalloc(&dout1, h.size() * 8);
  hipMalloc(&dt, nblk_max * 8);
  hipMemcpy(din, h.data(), h.size() * 8, hipMemcpyHostToDevice);
  // correctness: 5 squarings, both forms against a host evaluation
  {
    const int R = 5;
    const double scale = 1.3;
    hipLaunchKernelGGL(sym9_kernel<0>, dim3(64), dim3(64), 0, 0, din, dout0, dt, R, scale);
    hipLaunchKernelGGL(sym9_kernel<1>, dim3(64), dim3(64), 0, 0, din, dout1, dt, R, scale);
    hipDeviceSynchronize();
    std::vector<double> o0(256 * 81), o1(256 * 81);
    hipMemcpy(o0.data(), dout0, o0.size() * 8, hipMemcpyDeviceToHost);
    hipMemcpy(o1.data(), dout1, o1.size() * 8, hipMemcpyDeviceToHost);
    double e0 = 0, e1 = 0, mag = 0;
    for (int ch = 0; ch < 256; ++ch) {
      double a[81], b[81];
      for (int e = 0; e < 81; ++e) a[e] = h[ch * 81 + e];
      for (int it = 0; it < R; ++it) {
        for (int i = 0; i < 9; ++i)
          for (int j = 0; j < 9; ++j) {
            double sacc = 0;
            for (int k = 0; k < 9; ++k) sacc += a[i * 9 + k] * a[k * 9 + j];
            b[i * 9 + j] = scale * sacc;
          }
        for (int e = 0; e < 81; ++e) a[e] = b[e];
      }
      for (int e = 0; e < 81; ++e) {
        mag = fmax(mag, fabs(a[e]));
        e0 = fmax(e0, fabs(o0[ch * 81 + e] - a[e]));
        e1 = fmax(e1, fabs(o1[ch * 81 + e] - a[e]));
      }
    }
    printf("check (5 squarings, 256 chains): max |form0 - host| = %.3e, max |form1 - host| = %.3e, max |result| = %.3e\n", e0, e1, mag);
  }
  // timing input: Householder reflections W = 1 - 2 v v^T / v^T v (W W = 1: the chain stays at non-trivial, bounded values)
  for (long ch = 0; ch < nch; ++ch) {
    double v[9], vv = 0;
    for (int i = 0; i < 9; ++i) v[i] = rand() / (double)RAND_MAX - 0.5, vv += v[i] * v[i];
    for (int i = 0; i < 9; ++i)
      for (int j = 0; j < 9; ++j) h[ch * 81 + i * 9 + j] = (i == j ? 1.0 : 0.0) - 2.0 * v[i] * v[j] / vv;
  }
  hipMemcpy(din, h.data(), h.size() * 8, hipMemcpyHostToDevice);
  const int R = 20001;
  for (int w = 1; w <= maxw; ++w) {
    const long nblk = (long)cus * 4 * w;
    for (int form = 0; form < 2; ++form) {
      double best = 1e30;
      for (int rep = 0; rep < 5; ++rep) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        hipEventRecord(e0, 0);
        if (form == 0) hipLaunchKernelGGL(sym9_kernel<0>, dim3(nblk), dim3(64), 0, 0, din, dout0, dt, R, 1.0);
        else hipLaunchKernelGGL(sym9_kernel<1>, dim3(nblk), dim3(64), 0, 0, din, dout1, dt, R, 1.0);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        best = fmin(best, (double)ms);
      }
      std::vector<long long> tk(nblk);
      hipMemcpy(tk.data(), dt, nblk * 8, hipMemcpyDeviceToHost);
      double avg = 0;
      for (long i = 0; i < nblk; ++i) avg += (double)tk[i];
      avg /= (double)nblk;
      // wall_clock64 ticks at 100 MHz; kernel time gives ns per product per wave
      const double ns_per_prod = best * 1e6 / R;
      printf("waves/SIMD %d  form %d (%s): %.1f ns per dependent product and wave (kernel %.3f ms, R = %d; wall_clock ticks per product %.3f)  => %.2f products/us per SIMD\n",
             w, form, form == 0 ? "12 x 12 tiles, 12 MFMA + rank-1 tail" : "8 + 1 split, 6 MFMA + border on the vector unit", ns_per_prod, best, R,
             avg / R, w * 1e3 / ns_per_prod);
    }
  }
  return 0;
}
