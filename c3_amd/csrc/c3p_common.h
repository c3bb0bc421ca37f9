// Shared device/host helpers for libc3prop (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef double2 cplx;  // interleaved complex128, same bytes as numpy / TF complex128

__host__ __device__ __forceinline__ cplx cmake(double re, double im) { return make_double2(re, im); }
__device__ __forceinline__ cplx cadd(cplx a, cplx b) { return cmake(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ cplx csub(cplx a, cplx b) { return cmake(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ cplx cmul(cplx a, cplx b) {
  return cmake(fma(a.x, b.x, -a.y * b.y), fma(a.x, b.y, a.y * b.x));
}
__device__ __forceinline__ cplx cscale(cplx a, double s) { return cmake(a.x * s, a.y * s); }
__device__ __forceinline__ cplx cconj(cplx a) { return cmake(a.x, -a.y); }
// acc += a*b  (4 real FMAs)
__device__ __forceinline__ void cfma(cplx& acc, cplx a, cplx b) {
  acc.x = fma(a.x, b.x, acc.x);
  acc.x = fma(-a.y, b.y, acc.x);
  acc.y = fma(a.x, b.y, acc.y);
  acc.y = fma(a.y, b.x, acc.y);
}
__device__ __forceinline__ double cabs1(cplx a) { return hypot(a.x, a.y); }

// Running sum of the imaginary trace shifts, kept in [-pi, pi] (two-constant reduction):
// the phase e^{i sum mu_n} needs ABSOLUTE accuracy in the angle, and an unreduced sum over
// thousands of slices (|sum| ~ 1e4 rad) would lose ~1e-12 per addition.
__device__ __forceinline__ double c3p_phase_add(double acc, double inc) {
  const double two_pi_hi = 6.283185307179586, two_pi_lo = 2.4492935982947064e-16;
  double s = acc + inc;
  const double k = rint(s * 0.15915494309189535);
  s = fma(-k, two_pi_hi, s);
  s = fma(-k, two_pi_lo, s);
  return s;
}

// ---------------------------------------------------------------------------
// Truncated-Taylor / Paterson-Stockmeyer plan for exp(X), ||X||_1 = norm.
//
// The reference exponentiates with tf.linalg.expm (Higham 2005 Pade + solve).
// On the GPU the LU solve is the awkward part (pivoting, divergent), so the
// kernels use the mathematically equivalent scaled Taylor polynomial evaluated
// with Paterson-Stockmeyer: only matrix products.  Degrees are the PS-optimal
// ones; theta_m are the double-precision Taylor backward-error bounds of
// Al-Mohy & Higham 2011 (SIAM J. Sci. Comput. 33, table 3.1) -- with
// ||X||_1 <= theta_m the truncated series is exp(X + dX), ||dX|| <= 2^-53 ||X||.
// ---------------------------------------------------------------------------
struct TaylorPlan {
  int m;  // polynomial degree
  int q;  // block size: powers X^2..X^q are formed explicitly
  int r;  // number of blocks, m = q*r
  int s;  // squarings
};

#define C3P_NPLANS 7
__host__ __device__ __forceinline__ TaylorPlan c3p_pick_plan(double norm) {
  const int pm[C3P_NPLANS] = {2, 4, 6, 9, 12, 16, 20};
  const int pq[C3P_NPLANS] = {2, 2, 2, 3, 3, 4, 4};
  const int pr[C3P_NPLANS] = {1, 2, 3, 3, 4, 4, 5};
  const double th[C3P_NPLANS] = {2.58e-8, 3.40e-4, 9.07e-3, 8.96e-2, 3.00e-1, 7.81e-1, 1.44};
  int best = C3P_NPLANS - 1, best_cost = 1 << 30, best_s = 0;
  for (int i = 0; i < C3P_NPLANS; ++i) {
    int s = 0;
    if (norm > th[i]) {
      double ratio = norm / th[i];
      // s = ceil(log2(ratio)) without libm differences between host and device
      s = 0;
      double p = 1.0;
      while (p < ratio && s < 60) {
        p *= 2.0;
        ++s;
      }
    }
    int cost = (pq[i] - 1) + (pr[i] - 1) + s;
    if (cost < best_cost || (cost == best_cost && s <= best_s)) {
      best = i;
      best_cost = cost;
      best_s = s;
    }
  }
  TaylorPlan p;
  p.m = pm[best];
  p.q = pq[best];
  p.r = pr[best];
  p.s = best_s;
  return p;
}

// q = 4 only (powers X..X^4, Horner in X^4): degree 4r, r = 1..5.  Thresholds are the Taylor
// backward-error bounds for unit roundoff 2^-52 (theta_m * 2^(1/m)).
__host__ __device__ __forceinline__ TaylorPlan c3p_pick_plan_q4(double nrm) {
  const double th[5] = {4.0e-4, 5.45e-2, 3.18e-1, 8.16e-1, 1.49};
  int best_r = 5, best_s = 0, best_cost = 1 << 30;
  for (int i = 0; i < 5; ++i) {
    int si = 0;
    double p = th[i];
    while (p < nrm && si < 40) {
      p *= 2.0;
      ++si;
    }
    const int cost = i + si;
    if (cost < best_cost || (cost == best_cost && si <= best_s)) {
      best_cost = cost;
      best_r = i + 1;
      best_s = si;
    }
  }
  TaylorPlan p;
  p.q = 4;
  p.r = best_r;
  p.m = 4 * best_r;
  p.s = best_s;
  return p;
}

// Degree-18 Taylor polynomial in 5 products (Bader, Blanes & Casas, "Computing the matrix
// exponential with an optimized Taylor polynomial approximation", Mathematics 7 (2019) 1174):
//   A2 = A^2, A3 = A2 A, A6 = A3^2,
//   B1 = a11 A + a21 A2 + a31 A3,                  B5 = b24 A2 + b34 A3 + b64 A6,
//   B4 = b03 I + b13 A + b23 A2 + b33 A3 + b63 A6, A9 = B1 B5 + B4,
//   B3 = b02 I + b12 A + b22 A2 + b32 A3 + b62 A6, B2 = b11 A + b21 A2 + b31 A3 + b61 A6,
//   T18 = B2 + (B3 + A9) A9.
// Expanding it reproduces 1/k!, k <= 18, to 1e-15 (tests/test_abi_and_host.py).  Backward-error
// bound theta_18 = 1.09 (u = 2^-53); 1.13 for 2^-52.
#define C3P_T18_THETA 1.13
#define C3P_T18_A11 (-0.10036558103014462001)
#define C3P_T18_A21 (-0.00802924648241156960)
#define C3P_T18_A31 (-0.00089213849804572995)
#define C3P_T18_B11 (0.39784974949964507614)
#define C3P_T18_B21 (1.36783778460411719922)
#define C3P_T18_B31 (0.49828962252538267755)
#define C3P_T18_B61 (-0.00063789819459472330)
#define C3P_T18_B02 (-10.9676396052962062593)
#define C3P_T18_B12 (1.68015813878906197182)
#define C3P_T18_B22 (0.05717798464788655127)
#define C3P_T18_B32 (-0.00698210122488052084)
#define C3P_T18_B62 (0.00003349750170860705)
#define C3P_T18_B03 (-0.09043168323908105619)
#define C3P_T18_B13 (-0.06764045190713819075)
#define C3P_T18_B23 (0.06759613017704596460)
#define C3P_T18_B33 (0.02955525704293155274)
#define C3P_T18_B63 (-0.00001391802575160607)
#define C3P_T18_B24 (-0.09233646193671185927)
#define C3P_T18_B34 (-0.01693649390020817171)
#define C3P_T18_B64 (-0.00001400867981820361)

// T18 for NORMAL generators with an imaginary spectrum (round 6; tools/gen_t18_normal.py: the 20 parameters re-solved in 60-digit
// arithmetic for the Chebyshev-economised degree-18 polynomial of e^z on [-i theta, i theta]).  For X = -i dt (H - tr H / D) with H
// Hermitian, and for a Lindblad generator with Hermitian H up to a small dissipator, ||p(X) - exp(X)||_2 is the scalar error on the
// spectrum, so the radius is 2.0 instead of the 1.13 of the Taylor parameters (any matrix): cfg4 (1.6) needs no squaring.  Error
// bound of the economisation: even part 1.6e-18, odd part 6.3e-17; with the parameters rounded to double 3.4e-16 on the interval;
// measured against scipy's expm the no-squaring evaluation is as accurate as Taylor-T18 + one squaring also with a dissipator of
// norm 0.1 next to a skew part of norm 1.6 (tests/test_abi_and_host.py::test_t18_for_normal_generators).
// Row 0 = the published Taylor parameters, row 1 = the economised ones; order: a11, a21, a31, b11, b21, b31, b61, b02, b12, b22, b32, b62, b03, b13, b23, b33, b63, b24, b34, b64.
#define C3P_T18N_THETA 2.0
#define C3P_T18N_MAX_NONNORMAL 0.25  /* 1-norm of the symmetric (non-skew) part of a real generator the economised table accepts */
enum { C3P_I_A11, C3P_I_A21, C3P_I_A31, C3P_I_B11, C3P_I_B21, C3P_I_B31, C3P_I_B61, C3P_I_B02, C3P_I_B12, C3P_I_B22, C3P_I_B32, C3P_I_B62, C3P_I_B03, C3P_I_B13, C3P_I_B23, C3P_I_B33, C3P_I_B63, C3P_I_B24, C3P_I_B34, C3P_I_B64 };
__constant__ double c3p_t18_tab[2][20] = {
    {C3P_T18_A11, C3P_T18_A21, C3P_T18_A31, C3P_T18_B11, C3P_T18_B21, C3P_T18_B31, C3P_T18_B61, C3P_T18_B02, C3P_T18_B12, C3P_T18_B22, C3P_T18_B32, C3P_T18_B62, C3P_T18_B03, C3P_T18_B13, C3P_T18_B23, C3P_T18_B33, C3P_T18_B63, C3P_T18_B24, C3P_T18_B34, C3P_T18_B64},
    {-0x1.eb38ce5c91740p-4, -0x1.25636bcf4b5b5p-7, -0x1.04c9aedbe699cp-10, 0x1.6a80b11b0bcabp-3, 0x1.1dc21d43eb4cfp+0, 0x1.420dbe5746422p-2, -0x1.289454f80f910p-11, -0x1.3deb71e2be061p+2, 0x1.b130d3faac326p+0, 0x1.21f0bba48bb5dp-4, -0x1.f3a48235292fdp-9, 0x1.1ada67e7564b6p-15, -0x1.8ccda6e1cc3c0p-3, -0x1.b823d05f2b313p-3, 0x1.b4116a8ca5541p-5, 0x1.83665c777fc72p-6, -0x1.5ac92f9e0ac99p-17, -0x1.d79745bcca89dp-4, -0x1.7f4469c9ea034p-7, -0x1.9a9222987bdffp-17},
};

// FOUR-product evaluation of exp for normal generators with an imaginary spectrum (round 6; tools/gen_t16n4.py): the 16-parameter
// degree-16 scheme
//     A2 = A^2,   y0 = A2 (e0 A2 + e1 A),   y1 = (y0 + e2 A2 + e3 A)(y0 + e4 A2) + e5 y0 + e6 A2,
//     exp ~ y2 = (y1 + e7 A2 + e8 A)(y1 + e9 y0 + e10 A) + e11 y1 + e12 y0 + e13 A2 + e14 A + e15 I
// with its parameters solved (70-digit Gauss-Newton) for the Chebyshev-economised target on [-1.35 i, 1.35 i]: error bound 7.9e-17
// (even part) / 1.4e-17 (odd part), + 1.1e-16 for the unit constant term; with the parameters in double 6e-16 on the interval, and in
// double-precision matrix tests MORE accurate than the published T18 (5 products, radius 1.13) at every norm <= 1.35, also next to a
// dissipator of norm 0.25 (tests/test_abi_and_host.py::test_four_product_scheme_for_normal_generators).  Used where T18N is: Hermitian
// Hamiltonians on the complex small-D / mid-D loops, nearly skew-symmetric real Lindblad generators.
#define C3P_E4N_THETA 1.35
// (compile-time constants: the kernels index them with literals, so they are instruction operands, not loads)
constexpr double c3p_e4n[16] = {-0x1.a2dcd9f317fb5p-12, -0x1.7eb9c3a1416fep-9, 0x1.001119dfac460p-7, -0x1.9f18356154345p-2, -0x1.0a6d848c42c9ep-5, 0x1.ca82628a0fca7p-5, 0x1.0a698401e0e35p-2, -0x1.e6fe451dff289p-3, -0x1.9f4a7dec6c165p-6, -0x1.70af6b378b1a2p+2, 0x1.17d3c4bc46d68p+1, 0x1.44da69372b3bap+3, 0x1.579d596d5062bp+1, -0x1.0af90f8857611p+1, 0x1.0000000000000p+0, 0x1.0000000000000p+0};

// Plan for the MFMA kernels: either the q = 4 Paterson-Stockmeyer polynomial (degree 4r,
// 3 + (r-1) products) or T18 (5 products), plus s squarings; fewest products wins, ties go
// to Paterson-Stockmeyer (less element-wise work).
struct MfmaPlan {
  int t18;  // 1: use T18; 2: the four-product scheme c3p_e4n (normal generators only)
  int r;    // q = 4 degree 4r (when !t18)
  int s;    // squarings
};
// (theta18: C3P_T18_THETA for any matrix, C3P_T18N_THETA when the generator is known to be normal with an imaginary spectrum
// and the economised parameters of c3p_t18_tab row 1 are used)
// (normal: also consider the four-product scheme, 4 + s products; it wins whenever T18N would need a squaring that it does not, and
// ties with 5 + (s - 1) go to T18N's smaller truncation error)
__host__ __device__ __forceinline__ MfmaPlan c3p_pick_plan_mfma(double nrm, double theta18 = C3P_T18_THETA, bool normal = false) {
  const TaylorPlan q = c3p_pick_plan_q4(nrm);
  int s18 = 0;
  double p = theta18;
  while (p < nrm && s18 < 40) {
    p *= 2.0;
    ++s18;
  }
  MfmaPlan m;
  const int cost_q = 3 + (q.r - 1) + q.s, cost_t = 5 + s18;
  if (cost_t < cost_q) {
    m.t18 = 1;
    m.r = 0;
    m.s = s18;
  } else {
    m.t18 = 0;
    m.r = q.r;
    m.s = q.s;
  }
  if (normal) {
    int s4 = 0;
    double p4 = C3P_E4N_THETA;
    while (p4 < nrm && s4 < 40) {
      p4 *= 2.0;
      ++s4;
    }
    const int cost = m.t18 ? 5 + m.s : 3 + (m.r - 1) + m.s;
    if (4 + s4 < cost) {
      m.t18 = 2;
      m.r = 0;
      m.s = s4;
    }
  }
  return m;
}

// 1/k!, k = 0..21
__constant__ double c3p_inv_fact[22] = {
    1.0,
    1.0,
    0.5,
    1.0 / 6.0,
    1.0 / 24.0,
    1.0 / 120.0,
    1.0 / 720.0,
    1.0 / 5040.0,
    1.0 / 40320.0,
    1.0 / 362880.0,
    1.0 / 3628800.0,
    1.0 / 39916800.0,
    1.0 / 479001600.0,
    1.0 / 6227020800.0,
    1.0 / 87178291200.0,
    1.0 / 1307674368000.0,
    1.0 / 20922789888000.0,
    1.0 / 355687428096000.0,
    1.0 / 6402373705728000.0,
    1.0 / 121645100408832000.0,
    1.0 / 2432902008176640000.0,
    1.0 / 51090942171709440000.0,
};

// Economised cos / sin polynomials of the real path (round 6; tools/gen_minimax_cossin.py, exact rational arithmetic): for a real
// symmetric Y with ||Y|| <= theta the spectrum of W = Y^2 lies in [0, theta^2] and ||p(W) - f(W)||_2 is the scalar error on that
// interval (Y is normal), so cos(sqrt w) and sin(sqrt w) / sqrt w are replaced by the Chebyshev-economised degree-6 / degree-7
// polynomials on [0, theta^2]: the same accuracy as the degree-8 Taylor polynomials at theta_16 = 0.816 with ONE PRODUCT LESS
// (degree 6: powers to W^3), and theta = 1.30 instead of 1.13 for the degree-7 pair (powers to W^4, no W^4 term in the factor).
// (constant terms set to 1 exactly, so that exp(0) = I: idle slices, zero Hamiltonians; the bounds include that shift)
// degree 6, ||Y|| <= 0.83: error bound cos 2.0e-16, sin / Y 1.4e-17
#define C3P_MM6_THETA 0.83
__constant__ double c3p_mm6_cos[7] = {0x1.0000000000000p+0, -0x1.ffffffffffefbp-2, 0x1.5555555549783p-5, -0x1.6c16c15f2ae1ep-10, 0x1.a019f4383856bp-16, -0x1.27ddd6f90d4c9p-22, 0x1.1b266723fe85ap-29};
__constant__ double c3p_mm6_sinc[7] = {0x1.0000000000000p+0, -0x1.5555555555532p-3, 0x1.111111110de63p-7, -0x1.a01a019933c31p-13, 0x1.71de332d191bcp-19, -0x1.ae5cb6388e6d3p-26, 0x1.5d1c004547c53p-33};
// degree 7, ||Y|| <= 1.3: error bound cos 1.9e-16, sin / Y 1.1e-17
#define C3P_MM7_THETA 1.3
__constant__ double c3p_mm7_cos[8] = {0x1.0000000000000p+0, -0x1.fffffffffff7fp-2, 0x1.5555555552310p-5, -0x1.6c16c16a3a84dp-10, 0x1.a01a008b1794dp-16, -0x1.27e4a3ee8ce57p-22, 0x1.1ecee25c752dfp-29, -0x1.885f3ce855c89p-37};
__constant__ double c3p_mm7_sinc[8] = {0x1.0000000000000p+0, -0x1.5555555555546p-3, 0x1.1111111110536p-7, -0x1.a01a019f395fdp-13, 0x1.71de39d2c6922p-19, -0x1.ae6403f077f9dp-26, 0x1.610774ce48dc0p-33, -0x1.a3ec628c31277p-41};
// degree 8, ||Y|| <= 1.85: error bound cos 1.5e-16, sin / Y 7.8e-18 -- the product structure of the degree-8 Taylor pair (powers to W^4,
// theta_16 = 0.816) at more than twice its radius: cfg3 (1.32) and cfg5 (1.47) need no squaring with it
#define C3P_MM8_THETA 1.85
__constant__ double c3p_mm8_cos[9] = {0x1.0000000000000p+0, -0x1.fffffffffffc1p-2, 0x1.555555555460bp-5, -0x1.6c16c16bbaf0ap-10, 0x1.a01a017d9204bp-16, -0x1.27e4f42b216adp-22, 0x1.1eebbeaf2fd74p-29, -0x1.931094adecb53p-37, 0x1.99564aa7ce940p-45};
__constant__ double c3p_mm8_sinc[9] = {0x1.0000000000000p+0, -0x1.555555555554fp-3, 0x1.1111111110dd6p-7, -0x1.a01a019ff3325p-13, 0x1.71de3a46d43f1p-19, -0x1.ae6450543683bp-26, 0x1.6122d8e56c0d7p-33, -0x1.ae0d75ffe9679p-41, 0x1.83508886d45cep-49};

// Launch log of the current API call of this thread (c3p_last_kernel_detail; INTEGRATION.md's dispatch table is generated from
// it by tools/dispatch_table.py): every kernel launch of the library goes through C3P_LAUNCH, which notes the kernel's host
// function and the source file before handing over to hipLaunchKernelGGL.  A pointer store per launch; names are resolved
// (hipKernelNameRefByPtr + demangling) only when the log is read.
void c3p_note_launch(const void* host_fn, const char* file);
#define C3P_LAUNCH(kern, ...)                                                   \
  do {                                                                          \
    c3p_note_launch(reinterpret_cast<const void*>(&*(kern)), __FILE__);         \
    hipLaunchKernelGGL(kern, __VA_ARGS__);                                      \
  } while (0)

