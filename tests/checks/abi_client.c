/* A plain C99 consumer of include/c3prop.h: the drop-in boundary is a C ABI -- no C++ types, no torch, no HIP header on the
 * caller's side.  Built by tests/test_abi_and_host.py with `gcc -std=c99 -pedantic -Wall -Werror` against the in-tree
 * libc3prop.so (loaded with dlopen, as a foreign-language FFI would).
 *
 *   abi_client <libc3prop.so> host     version, option table, tape arithmetic, error path without a device (CPU test)
 *   abi_client <libc3prop.so> gpu      one batch through c3p_pwc_unitary with host pointers: U unitary, and equal to the
 *                                      product of the slice propagators c3p_pwc_unitary returns in dUs_out (GPU test)
 */
#include <dlfcn.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "c3prop.h"

#define LOAD(name)                                                  \
  do {                                                              \
    *(void**)(&p_##name) = dlsym(h, #name);                         \
    if (!p_##name) {                                                \
      fprintf(stderr, "missing symbol %s\n", #name);                \
      return 2;                                                     \
    }                                                               \
  } while (0)

static int (*p_c3p_version)(void);
static int (*p_c3p_device_count)(void);
static const char* (*p_c3p_last_error)(void);
static int (*p_c3p_last_kernel)(void);
static int (*p_c3p_set_option)(const char*, const char*);
static long (*p_c3p_get_option)(const char*);
static size_t (*p_c3p_pwc_lindblad_tape_bytes)(int, int, int, int, int*);
static int (*p_c3p_pwc_unitary)(const void*, int64_t, const void*, int64_t, const double*, double, int, int, int, int, int, const double*,
                                void*, void*, void*);
static void (*p_c3p_shutdown)(void);

#define CHECK(cond)                                                         \
  do {                                                                      \
    if (!(cond)) {                                                          \
      fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond);     \
      return 1;                                                             \
    }                                                                       \
  } while (0)

/* c = a b, complex D x D row-major interleaved */
static void cmatmul(const double* a, const double* b, double* c, int D) {
  int i, j, k;
  for (i = 0; i < D; ++i)
    for (j = 0; j < D; ++j) {
      double re = 0.0, im = 0.0;
      for (k = 0; k < D; ++k) {
        const double ar = a[2 * (i * D + k)], ai = a[2 * (i * D + k) + 1], br = b[2 * (k * D + j)], bi = b[2 * (k * D + j) + 1];
        re += ar * br - ai * bi;
        im += ar * bi + ai * br;
      }
      c[2 * (i * D + j)] = re;
      c[2 * (i * D + j) + 1] = im;
    }
}

int main(int argc, char** argv) {
  void* h;
  if (argc < 3) {
    fprintf(stderr, "usage: abi_client <libc3prop.so> host|gpu\n");
    return 2;
  }
  h = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL);
  if (!h) {
    fprintf(stderr, "dlopen: %s\n", dlerror());
    return 2;
  }
  LOAD(c3p_version);
  LOAD(c3p_device_count);
  LOAD(c3p_last_error);
  LOAD(c3p_last_kernel);
  LOAD(c3p_set_option);
  LOAD(c3p_get_option);
  LOAD(c3p_pwc_lindblad_tape_bytes);
  LOAD(c3p_pwc_unitary);
  LOAD(c3p_shutdown);

  if (strcmp(argv[2], "host") == 0) {
    int seg = -1;
    size_t n;
    CHECK(p_c3p_version() >= 1);
    CHECK(p_c3p_get_option("no_such_option") == -2);
    CHECK(p_c3p_set_option("no_such_option", "1") != 0 && strlen(p_c3p_last_error()) > 0);
    CHECK(p_c3p_set_option("segments", "3") == 0 && p_c3p_get_option("segments") == 3);
    CHECK(p_c3p_set_option("segments", NULL) == 0 && p_c3p_get_option("segments") == -1);
    n = p_c3p_pwc_lindblad_tape_bytes(64, 2, 1000, 9, &seg);
    CHECK(seg == 4 && n >= (size_t)64 * 1000 * 81 * 81 * 8);
    CHECK(p_c3p_pwc_lindblad_tape_bytes(4, 2, 100, 5, &seg) == 0 && seg == 0);
    if (p_c3p_device_count() == 0) {
      /* no device: every compute entry point fails loudly, with a message */
      double h0[2 * 4] = {1, 0, 0, 0, 0, 0, -1, 0}, hk[2 * 4] = {0, 0, 1, 0, 1, 0, 0, 0}, sig[3] = {0.1, 0.2, 0.3}, U[2 * 4];
      const int rc = p_c3p_pwc_unitary(h0, 0, hk, 0, sig, 0.1, 1, 1, 3, 2, C3P_HOST_PTRS, NULL, U, NULL, NULL);
      CHECK(rc != 0 && strlen(p_c3p_last_error()) > 0);
    }
    printf("ABI_CLIENT_HOST_OK version=%d devices=%d\n", p_c3p_version(), p_c3p_device_count());
    return 0;
  }

  {
    /* three-level system, two control lines, 4 samples x 24 slices, host pointers */
    enum { D = 3, K = 2, N = 24, B = 4 };
    double h0[2 * D * D] = {0}, hks[2 * K * D * D] = {0}, sig[B * K * N], phase[B * D];
    double* U = (double*)malloc(sizeof(double) * 2 * B * D * D);
    double* dUs = (double*)malloc(sizeof(double) * 2 * B * N * D * D);
    double acc[2 * D * D], tmp[2 * D * D];
    int b, n, i, j, k, rc;
    double worst_unit = 0.0, worst_prod = 0.0;
    CHECK(p_c3p_device_count() > 0);
    h0[2 * (1 * D + 1)] = 1.0;
    h0[2 * (2 * D + 2)] = 1.9;
    /* a + a^dagger, and i (a^dagger - a) (a complex control operator: the complex path) */
    hks[2 * (0 * D + 1)] = hks[2 * (1 * D + 0)] = 1.0;
    hks[2 * (1 * D + 2)] = hks[2 * (2 * D + 1)] = sqrt(2.0);
    hks[2 * D * D + 2 * (0 * D + 1) + 1] = -1.0;
    hks[2 * D * D + 2 * (1 * D + 0) + 1] = 1.0;
    hks[2 * D * D + 2 * (1 * D + 2) + 1] = -sqrt(2.0);
    hks[2 * D * D + 2 * (2 * D + 1) + 1] = sqrt(2.0);
    for (i = 0; i < B * K * N; ++i) sig[i] = 0.8 * sin(0.37 * i) + 0.1;
    for (i = 0; i < B * D; ++i) phase[i] = 0.0;
    rc = p_c3p_pwc_unitary(h0, 0, hks, 0, sig, 0.05, B, K, N, D, C3P_HOST_PTRS, phase, U, dUs, NULL);
    if (rc != 0) fprintf(stderr, "c3p_pwc_unitary: %s\n", p_c3p_last_error());
    CHECK(rc == 0);
    for (b = 0; b < B; ++b) {
      const double* Ub = U + 2 * b * D * D;
      /* U^dagger U = 1 */
      for (i = 0; i < D; ++i)
        for (j = 0; j < D; ++j) {
          double re = 0.0, im = 0.0;
          for (k = 0; k < D; ++k) {
            const double ar = Ub[2 * (k * D + i)], ai = -Ub[2 * (k * D + i) + 1], br = Ub[2 * (k * D + j)], bi = Ub[2 * (k * D + j) + 1];
            re += ar * br - ai * bi;
            im += ar * bi + ai * br;
          }
          re -= (i == j) ? 1.0 : 0.0;
          if (fabs(re) > worst_unit) worst_unit = fabs(re);
          if (fabs(im) > worst_unit) worst_unit = fabs(im);
        }
      /* U = dU_{N-1} ... dU_0 (tf_matmul_n's order, c3/utils/tf_utils.py:144-163) */
      memcpy(acc, dUs + 2 * ((size_t)b * N) * D * D, sizeof(acc));
      for (n = 1; n < N; ++n) {
        cmatmul(dUs + 2 * ((size_t)b * N + n) * D * D, acc, tmp, D);
        memcpy(acc, tmp, sizeof(acc));
      }
      for (i = 0; i < 2 * D * D; ++i)
        if (fabs(acc[i] - Ub[i]) > worst_prod) worst_prod = fabs(acc[i] - Ub[i]);
    }
    CHECK(worst_unit < 1e-12);
    CHECK(worst_prod < 1e-12);
    CHECK(p_c3p_last_kernel() != C3P_KERNEL_NONE);
    printf("ABI_CLIENT_GPU_OK kernel=%d unitarity=%.2e product=%.2e\n", p_c3p_last_kernel(), worst_unit, worst_prod);
    free(U);
    free(dUs);
    p_c3p_shutdown();
  }
  return 0;
}
