// LDS bank-conflict model of gfx950 for 64-bit accesses: cycles per ds_read_b64 / ds_write_b64 for lane -> address maps
// taken from the mid-D kernels.  hipcc --offload-arch=gfx950 -O3 tools/ubench_lds.hip -o tools/ubench_lds && tools/ubench_lds
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void __launch_bounds__(64) k(const int* addr, int iters, int write, double* out, long long* cyc) {
  __shared__ double lds[8192];
  const int lane = threadIdx.x;
  for (int i = lane; i < 8192; i += 64) lds[i] = i;
  __syncthreads();
  const int a = addr[lane];
  double acc = 0.0;
  long long t0 = clock64();
  if (write) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        lds[a + 64 * (u & 7)] = acc + u;
        asm volatile("" ::: "memory");
      }
    }
  } else {
    for (int it = 0; it < iters; ++it) {
      double v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        v[u] = lds[a + 64 * (u & 7)];
        asm volatile("" : "+v"(v[u])::"memory");
      }
#pragma unroll
      for (int u = 0; u < 16; ++u) acc += v[u];
    }
  }
  __syncthreads();
  long long t1 = clock64();
  out[blockIdx.x * 64 + lane] = acc + lds[lane];
  if (lane == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
  struct P { const char* name; int (*f)(int); };
  P ps[] = {
      {"consecutive: addr = lane", [](int l) { return l; }},
      {"B big read, stride 37: (l&15) + 37 (l>>4)", [](int l) { return (l & 15) + 37 * (l >> 4); }},
      {"B big read, stride 38: (l&15) + 38 (l>>4)", [](int l) { return (l & 15) + 38 * (l >> 4); }},
      {"B big read, stride 32: (l&15) + 32 (l>>4)", [](int l) { return (l & 15) + 32 * (l >> 4); }},
      {"B big read, swizzled 32: ((l&15) ^ 16 (r&1)) + 32 r", [](int l) { int r = l >> 4; return ((l & 15) ^ (16 * (r & 1))) + 32 * r; }},
      {"A slab read, stride 37: 37 (l&15) + (l>>4)", [](int l) { return 37 * (l & 15) + (l >> 4); }},
      {"A slab read, stride 38: 38 (l&15) + (l>>4)", [](int l) { return 38 * (l & 15) + (l >> 4); }},
      {"A slab read, stride 32: 32 (l&15) + (l>>4)", [](int l) { return 32 * (l & 15) + (l >> 4); }},
      {"A slab read, stride 33", [](int l) { return 33 * (l & 15) + (l >> 4); }},
      {"small B block (broadcast over b), stride 37: 37 r + c", [](int l) { return 37 * (l >> 4) + (l & 3); }},
      {"small tile element, stride 37: 37 (4b + r) + c", [](int l) { return 37 * (4 * ((l >> 2) & 3) + (l >> 4)) + (l & 3); }},
      {"small tile element, stride 38", [](int l) { return 38 * (4 * ((l >> 2) & 3) + (l >> 4)) + (l & 3); }},
      {"small tile element, stride 32 swizzled", [](int l) { int row = 4 * ((l >> 2) & 3) + (l >> 4); return 32 * row + ((l & 3) ^ (16 * (row & 1))); }},
      {"mirror read 32-wide, f = x & 15", [](int l) { int cp = l & 15, rp = l >> 4; return 32 * cp + (rp ^ (cp & 15)); }},
      {"mirror read 32-wide, operand swizzle", [](int l) { int cp = l & 15, rp = l >> 4; return 32 * cp + (rp ^ (16 * (cp & 1))); }},
      {"all lanes one address", [](int) { return 5; }},
      {"2 addresses per 16 lanes (8-way broadcast)", [](int l) { return (l & 1) + 2 * (l >> 4); }},
      {"stride 2: 2 lane", [](int l) { return 2 * l; }},
      {"stride 16: 16 (l&15) + (l>>4)", [](int l) { return 16 * (l & 15) + (l >> 4); }},
      {"stride 8", [](int l) { return 8 * (l & 15) + (l >> 4); }},
      {"stride 4", [](int l) { return 4 * (l & 15) + (l >> 4); }},
  };
  int* da; double* dout; long long* dc;
  hipMalloc(&da, 64 * sizeof(int)); hipMalloc(&dout, 64 * 8); hipMalloc(&dc, 8);
  const int iters = 2000;
  for (auto& p : ps) {
    std::vector<int> a(64);
    for (int l = 0; l < 64; ++l) a[l] = p.f(l);
    hipMemcpy(da, a.data(), 64 * sizeof(int), hipMemcpyHostToDevice);
    double res[2];
    for (int w = 0; w < 2; ++w) {
      hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, iters, w, dout, dc);
      hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, iters, w, dout, dc);
      hipDeviceSynchronize();
      long long c; hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
      res[w] = (double)c / (iters * 16.0);
    }
    printf("%-62s read %6.2f  write %6.2f  (clock64 ticks per instruction)\n", p.name, res[0], res[1]);
  }
  return 0;
}
