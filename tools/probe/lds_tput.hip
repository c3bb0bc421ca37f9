// LDS read throughput on gfx950 with a full CU (12 waves / CU as in the mid-D real kernel, 3 workgroups of 4 waves):
// LDS cycles per wave-instruction for the lane -> address maps of the pinwheel deal.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <functional>
template <int MODE>  // 0: ds_read_b64, 1: ds_read_b128, 2: ds_write_b64
__global__ void __launch_bounds__(256, 3) k(const int* addr, int iters, double* out) {
  __shared__ double lds[5120];
  for (int i = threadIdx.x; i < 5120; i += 256) lds[i] = i;
  __syncthreads();
  const unsigned a = (unsigned)(size_t)(lds) + 8u * addr[threadIdx.x & 63];
  double acc = 0;
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
      double v0, v1, v2, v3, v4, v5, v6, v7;
      asm volatile("ds_read_b64 %0, %8\n ds_read_b64 %1, %8 offset:1024\n ds_read_b64 %2, %8 offset:2048\n ds_read_b64 %3, %8 offset:3072\n"
                   "ds_read_b64 %4, %8 offset:4096\n ds_read_b64 %5, %8 offset:5120\n ds_read_b64 %6, %8 offset:6144\n ds_read_b64 %7, %8 offset:7168\n s_waitcnt lgkmcnt(0)"
                   : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3), "=&v"(v4), "=&v"(v5), "=&v"(v6), "=&v"(v7) : "v"(a) : "memory");
      acc += v0 + v7;
    } else if (MODE == 1) {
      typedef double d2 __attribute__((ext_vector_type(2)));
      d2 v0, v1, v2, v3, v4, v5, v6, v7;
      asm volatile("ds_read_b128 %0, %8\n ds_read_b128 %1, %8 offset:1024\n ds_read_b128 %2, %8 offset:2048\n ds_read_b128 %3, %8 offset:3072\n"
                   "ds_read_b128 %4, %8 offset:4096\n ds_read_b128 %5, %8 offset:5120\n ds_read_b128 %6, %8 offset:6144\n ds_read_b128 %7, %8 offset:7168\n s_waitcnt lgkmcnt(0)"
                   : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3), "=&v"(v4), "=&v"(v5), "=&v"(v6), "=&v"(v7) : "v"(a) : "memory");
      acc += v0.x + v7.y;
    } else {
      asm volatile("ds_write_b64 %0, %1\n ds_write_b64 %0, %1 offset:1024\n ds_write_b64 %0, %1 offset:2048\n ds_write_b64 %0, %1 offset:3072\n"
                   "ds_write_b64 %0, %1 offset:4096\n ds_write_b64 %0, %1 offset:5120\n ds_write_b64 %0, %1 offset:6144\n ds_write_b64 %0, %1 offset:7168\n s_waitcnt lgkmcnt(0)"
                   : : "v"(a), "v"(acc) : "memory");
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}
int main() {
  struct P { const char* name; std::function<int(int)> f; int mode; };
  auto R = [](int l) { return l >> 4; }; auto Bq = [](int l) { return (l >> 2) & 3; }; auto C = [](int l) { return l & 3; };
  std::vector<P> ps = {
      {"b64 consecutive", [](int l) { return l; }, 0},
      {"b64 F16 swizzled (row r, 16 columns)", [=](int l) { int r = R(l); return 32 * r + ((4 * Bq(l) + C(l)) ^ (16 * (r & 1))); }, 0},
      {"b64 F16 at column 12", [=](int l) { int r = R(l); return 32 * r + ((12 + 4 * Bq(l) + C(l)) ^ (16 * (r & 1))); }, 0},
      {"b64 F4 broadcast (row r, column c)", [=](int l) { int r = R(l); return 32 * r + (C(l) ^ (16 * (r & 1))); }, 0},
      {"b64 centre K-packed (row 4b + r, column 12 + c)", [=](int l) { int r = R(l); return 32 * (4 * Bq(l) + r) + ((12 + C(l)) ^ (16 * (r & 1))); }, 0},
      {"b64 one address", [](int) { return 5; }, 0},
      {"b128 consecutive", [](int l) { return 2 * l; }, 1},
      {"b128 broadcast 16 addresses", [=](int l) { return 2 * (4 * R(l) + C(l)); }, 1},
      {"b128 32 addresses (2 x F4)", [=](int l) { return 2 * (8 * R(l) + 4 * (Bq(l) & 1) + C(l)); }, 1},
      {"W=37 B big read: 37 r + 4b + c", [=](int l) { return 37 * R(l) + 4 * Bq(l) + C(l); }, 0},
      {"W=37 A slab read: 37 (4b + c) + r", [=](int l) { return 37 * (4 * Bq(l) + C(l)) + R(l); }, 0},
      {"W=37 small B block: 37 r + c", [=](int l) { return 37 * R(l) + C(l); }, 0},
      {"W=37 wide A (rows 32..35): 37 c + r", [=](int l) { return 37 * C(l) + R(l); }, 0},
      {"W=40 rotated B big read: 40 r + (4b + c + 8 (r&1))", [=](int l) { int r = R(l); return 40 * r + (4 * Bq(l) + C(l) + 8 * (r & 1)); }, 0},
      {"W=40 rotated small B block", [=](int l) { int r = R(l); return 40 * r + (C(l) + 8 * (r & 1)); }, 0},
      {"w64 W=37 big element: 37 r + 4b + c", [=](int l) { return 37 * R(l) + 4 * Bq(l) + C(l); }, 2},
      {"w64 W=37 small element: 37 (4b + r) + c", [=](int l) { return 37 * (4 * Bq(l) + R(l)) + C(l); }, 2},
      {"w64 W=40 rotated big element", [=](int l) { int r = R(l); return 40 * r + 4 * Bq(l) + C(l) + 8 * (r & 1); }, 2},
      {"w64 W=40 rotated small element: 40 (4b + r) + c + 8 (r & 1)", [=](int l) { int r = R(l); return 40 * (4 * Bq(l) + r) + C(l) + 8 * (r & 1); }, 2},
      {"w64 consecutive", [](int l) { return l; }, 2},
      {"w64 wide element (row r, 16 columns swizzled)", [=](int l) { int r = R(l); return 32 * r + ((4 * Bq(l) + C(l)) ^ (16 * (r & 1))); }, 2},
      {"w64 tall element (row 4b + r, column c)", [=](int l) { int r = R(l); return 32 * (4 * Bq(l) + r) + (C(l) ^ (16 * (r & 1))); }, 2},
      {"w64 tall element, column ^ 4 (row >> 2)", [=](int l) { int r = R(l), row = 4 * Bq(l) + r; return 32 * row + ((C(l) ^ (16 * (r & 1))) ^ (4 * ((row >> 2) & 3))); }, 2},
  };
  int* da; double* dout;
  hipMalloc(&da, 64 * 4); hipMalloc(&dout, 768 * 256 * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000, nwg = 768;
  for (auto& p : ps) {
    int h[64]; for (int l = 0; l < 64; ++l) h[l] = p.f(l);
    hipMemcpy(da, h, sizeof(h), hipMemcpyHostToDevice);
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      if (p.mode == 0) hipLaunchKernelGGL(k<0>, dim3(nwg), dim3(256), 0, 0, da, iters, dout);
      else if (p.mode == 1) hipLaunchKernelGGL(k<1>, dim3(nwg), dim3(256), 0, 0, da, iters, dout);
      else hipLaunchKernelGGL(k<2>, dim3(nwg), dim3(256), 0, 0, da, iters, dout);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    // per CU: 12 waves x iters x 8 instructions
    const double cyc = best * 1e-3 * 2.4e9 / (12.0 * iters * 8);
    printf("%-55s %8.3f ms  %.2f cycles per wave-instruction per CU (@2.4 GHz)\n", p.name, best, cyc);
  }
  return 0;
}
