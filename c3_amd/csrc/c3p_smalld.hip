// Small-D propagator chains on the f64 matrix cores (D <= 12: cfg1 D=3, cfg2 D=9, the
// 9x9 Lindblad superoperator of a qutrit, ...).
//
// Mapping (gfx950): v_mfma_f64_4x4x4_4b_f64 performs FOUR independent 4x4x4 real
// products per instruction at the full fp64 rate (16 cycles; measured 72.8 TFLOP/s,
// tools/ubench_f64.hip).  Block b of the instruction is given to chain b: one
// wavefront owns four independent (sample, time-segment) chains, 16 lanes each, and
// every matrix of a chain lives in registers as 4x4 real blocks.
//
// Complex arithmetic is carried by the real 2x2 representation z=a+ib -> [[a,-b],[b,a]]:
//   R(A) (2D x 2D) times the "half" image Bh = first column of each 2x2 (2D x D)
//   gives Ch = half image of C = A B.  No flop is redundant.
//   * D-layout (MFMA C/D and B operand): register zh[I][J], lane (r=l/16, c=l%4) holds
//       Zh[4I+r][4J+c] = (r even ? Re : Im) Z[2I + r/2][4J + c]
//   * A-layout (MFMA A operand): register ra[I][K], lane (r,c) holds R(A)[4I+c][4K+r].
//     Obtained from the D-layout through a per-chain LDS image (15 ds_write_b64 +
//     25 ds_read_b64 + sign xor for D=9); row stride 4*NJ+1 doubles keeps both the
//     16-lane writes and the 32-lane reads bank-conflict free.
//
// Per slice (reference: propagation.py:426-440 + tf_utils.py:144-193):
//   X = G0 + sum_k c_k(n) G_k           (tables prepared by smalld_prep: G = -i dt (h - tr/D))
//   E = exp(X): Taylor, Paterson-Stockmeyer with q=4 (powers X..X^4, Horner in X^4),
//       degree 4r in {4,8,12,16,20} and s squarings chosen from a 1-norm bound, wave-uniform
//   U <- E U ;  the scalar factors e^{mu_n} are summed and applied once per segment.
#include <type_traits>

#include <cstdlib>
#include "c3p_common.h"
#include "c3p_kernels.h"
#include "c3p_smalld.h"

// Built as two translation units (__graft_entry__.build: -DC3P_SMALLD_PART=1 | 2): everything but the real-Hamiltonian
// backward sweep, and that sweep alone -- it is compiled with -mllvm -amdgpu-mfma-vgpr-form (see smalld_grad_real_kernel).
// Without the macro everything lands in one unit.
#ifdef C3P_SMALLD_PART
#define C3P_SMALLD_HAS(p) (C3P_SMALLD_PART == (p))
#else
#define C3P_SMALLD_HAS(p) 1
#endif

extern __shared__ __attribute__((aligned(16))) double c3p_sd_lds[];

namespace {

template <int D>
struct SD {
  static constexpr int NBI = (D + 1) / 2;  // 4-row blocks of the 2D-row half image
  static constexpr int NJ = (D + 3) / 4;   // 4-column blocks
  static constexpr int W = 4 * NJ + 1;     // LDS row stride in doubles
  static constexpr int MAT = 4 * NBI * W;  // doubles per matrix image
  // per-chain image buffer of the forward kernel: one complex half image.  IMG = 12 (mod 32) doubles keeps the
  // LDS patterns bank-conflict free under the gfx950 rules (ds_read_b64: 32-lane groups on 64 dword banks;
  // ds_write_b64: 16-lane groups on 32): image writes b IMG + r W + c, A-fragment reads b IMG + rho W.
  static constexpr int IMG = ((MAT - 12 + 31) / 32) * 32 + 12;
};

__device__ __forceinline__ double mfma4(double a, double b, double c) {
  return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ double readfirstlane_f64(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_readfirstlane(lo);
  hi = __builtin_amdgcn_readfirstlane(hi);
  return __hiloint2double(hi, lo);
}

// LDS read that the backend may not pair into ds_read2_b64: a wave64 ds_read_b64 costs 2 LDS cycles
// (256 B/clk), a ds_read2_b64 8 for two (128 B/clk), and the image strides are laid out for the bank
// rule of the single form (32-lane groups, 64 dword banks).  With eight waves per CU the LDS pipe, not
// the matrix cores, was the busiest unit of the real fast path before this.
typedef __attribute__((address_space(3))) const volatile double c3p_lds_cvd;
__device__ __forceinline__ double lds_ld(const double* p) { return *(c3p_lds_cvd*)p; }

__device__ __forceinline__ double flip_sign(double v, unsigned mask_hi) {
  unsigned long long u = __double_as_longlong(v);
  u ^= ((unsigned long long)mask_hi) << 32;
  return __longlong_as_double(u);
}

// Wave-level ordering point for the per-chain LDS images.  The workgroup is a single
// wavefront and the LDS services one wave's instructions in issue order, so no s_barrier
// is needed; the fence only stops the compiler from reordering image writes and reads.
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// D-layout registers -> the chain's LDS image (plain half image Zh, row stride W)
template <int D>
__device__ __forceinline__ void write_image(const double (&zh)[SD<D>::NBI][SD<D>::NJ], double* img, int woff) {
  using C = SD<D>;
  wave_sync();
#pragma unroll
  for (int I = 0; I < C::NBI; ++I)
#pragma unroll
    for (int J = 0; J < C::NJ; ++J) img[woff + I * 4 * C::W + J * 4] = zh[I][J];
  wave_sync();
}

// acc += M * B, where the left operand M is read from its LDS image as A-layout fragments
// (one K-column of 4x4 blocks at a time: NBI ds_read_b64 + sign xor per 4*NBI*NJ... MFMAs)
template <int D>
__device__ __forceinline__ void mm_img(const double* img, int roff, unsigned negmask,
                                       const double (&zb)[SD<D>::NBI][SD<D>::NJ],
                                       double (&acc)[SD<D>::NBI][SD<D>::NJ]) {
  using C = SD<D>;
#ifdef C3P_SD_ABSKIP
  // TIMING-ONLY builds (wrong results; tools/ab_complex_border.sh): what a core + border form of the complex products could
  // save at most.  1: no matrix instructions for the last column block (D = 9: the block that holds column 8 alone);
  // 2: none for the last row block and the last K-step either (the 8 x 8 core alone, borders for free)
  constexpr int SKJ = (D % 4 == 1 && D > 4) ? 1 : 0, SKI = (C3P_SD_ABSKIP >= 2 && D % 4 == 1 && D > 4) ? 1 : 0;
#else
  constexpr int SKJ = 0, SKI = 0;
#endif
  // software pipelined: the A fragments of step K+1 are in flight while step K's MFMAs issue
  double ra[2][C::NBI];
#pragma unroll
  for (int I = 0; I < C::NBI - SKI; ++I) ra[0][I] = flip_sign(lds_ld(img + roff + I * 4 * C::W), negmask);
#pragma unroll
  for (int K = 0; K < C::NBI - SKI; ++K) {
    if (K + 1 < C::NBI - SKI) {
#pragma unroll
      for (int I = 0; I < C::NBI - SKI; ++I) ra[(K + 1) & 1][I] = flip_sign(lds_ld(img + roff + I * 4 * C::W + (K + 1) * 2), negmask);
    }
#pragma unroll
    for (int I = 0; I < C::NBI - SKI; ++I)
#pragma unroll
      for (int J = 0; J < C::NJ - SKJ; ++J) acc[I][J] = mfma4(ra[K & 1][I], zb[K][J], acc[I][J]);
    __builtin_amdgcn_sched_barrier(0);
  }
}

struct LanePos {
  int r, b, c;
  int idx16;  // position among the chain's 16 lanes
};

// out = c1 X + c2 A2 + c3 A3 (+ c0 on the diagonal)
template <int D>
__device__ __forceinline__ void poly_block(double (&out)[SD<D>::NBI][SD<D>::NJ], double c0, double c1,
                                           double c2, double c3, const double (&X)[SD<D>::NBI][SD<D>::NJ],
                                           const double (&A2)[SD<D>::NBI][SD<D>::NJ],
                                           const double (&A3)[SD<D>::NBI][SD<D>::NJ], int ddelta,
                                           int rhalf) {
  using C = SD<D>;
#pragma unroll
  for (int I = 0; I < C::NBI; ++I)
#pragma unroll
    for (int J = 0; J < C::NJ; ++J) {
      double v = c1 * X[I][J];
      v = fma(c2, A2[I][J], v);
      v = fma(c3, A3[I][J], v);
      // diagonal: row 2I + r/2 (real part, r even) == column 4J + c, inside the D x D matrix
      if (2 * I - 4 * J >= -1 && 2 * I - 4 * J <= 3) {
        const bool on = (ddelta == 4 * J - 2 * I) && (2 * I + rhalf < D);
        v += on ? c0 : 0.0;
      }
      out[I][J] = v;
    }
}

// out = cx X + c2 A2 + c3 A3 + c6 A6 (+ c0 on the diagonal) (+ add)
template <int D>
__device__ __forceinline__ void lincomb6(double (&out)[SD<D>::NBI][SD<D>::NJ], double c0, double cx, double c2,
                                         double c3, double c6, const double (&X)[SD<D>::NBI][SD<D>::NJ],
                                         const double (&A2)[SD<D>::NBI][SD<D>::NJ],
                                         const double (&A3)[SD<D>::NBI][SD<D>::NJ],
                                         const double (&A6)[SD<D>::NBI][SD<D>::NJ], int ddelta, int rhalf) {
  using C = SD<D>;
#pragma unroll
  for (int I = 0; I < C::NBI; ++I)
#pragma unroll
    for (int J = 0; J < C::NJ; ++J) {
      double v = cx * X[I][J];
      v = fma(c2, A2[I][J], v);
      v = fma(c3, A3[I][J], v);
      if (c6 != 0.0) v = fma(c6, A6[I][J], v);
      if (c0 != 0.0 && 2 * I - 4 * J >= -1 && 2 * I - 4 * J <= 3) {
        const bool on = (ddelta == 4 * J - 2 * I) && (2 * I + rhalf < D);
        v += on ? c0 : 0.0;
      }
      out[I][J] = v;
    }
}

// ---------------------------------------------------------------------------------------------
// Real-Hamiltonian fast path.  When h0 and every hk are real (lab-frame transmon models in the dressed
// basis are: model.py:453-534 dresses with the real eigenvectors of a real symmetric matrix),
// X = -i Y with Y = dt (H - tr H / D) real symmetric and
//     exp(X) = cos Y - i sin Y,   cos Y = sum_j (-1)^j W^j/(2j)!,  sin Y = Y sum_j (-1)^j W^j/(2j+1)!,  W = Y^2,
// i.e. the even and odd parts of the SAME degree-18 Taylor polynomial the complex path evaluates (T18), so
// the truncation / scaling rule (theta = 1.13) is unchanged.  Everything up to the chain product is REAL
// D x D arithmetic: 8 real products (27 MFMAs each at D = 9) instead of 5 complex ones (75 each).
// Real matrices live in registers as NB x NB tiles of 4x4 (lane (r,c) of the chain's block holds
// M[4I+r][4J+c]); the left operand is read from a row-major LDS image (A layout: M[4I+c][4K+r]).
// ---------------------------------------------------------------------------------------------
template <int D>
struct RD {
  static constexpr int NB = (D + 3) / 4;
};

// Every matrix of the real path is a polynomial in the real SYMMETRIC Y, hence symmetric: products only
// compute the tiles on and above the diagonal (6 of 9 at D = 9) and sym_fill mirrors them.
// lane (r, c) of tile (I, J) holds M[4I + r][4J + c]; its mirror element M[4J + c][4I + r] is held by lane
// (c, r) of tile (J, I): one in-chain lane swap per lower tile (ds_bpermute).
// Only the tile rows a product reads as operands need their lower tiles: operand tiles are za[K][.] / zb[K][.] with
// K < KM, and with the rank-1 tail (D = 1 mod 4) the last row of tiles is never an operand -- at D = 9 ONE lower tile,
// (1,0), instead of three.  FULL mirrors everything (matrices that leave the symmetric stage: dU output, U <- E).
template <int D>
struct SymTail;
template <int D, bool FULL = false>
__device__ __forceinline__ void sym_fill(double (&m)[RD<D>::NB][RD<D>::NB], int swap_lane);

// The left operand of every real product is SYMMETRIC, and for a symmetric M the A-layout fragment of tile
// (I, K) -- lane (r, c) <- M[4I + c][4K + r] -- equals M[4K + r][4I + c], the D-layout register of tile (K, I)
// the same lane already holds: the real stage needs NO LDS image, no fragment loads, no round trips.
//   acc[I][J] += sum_K A(I,K) B(K,J) = sum_K mfma(Areg[K][I], Breg[K][J])     (tiles J >= I, mirrored afterwards)
// broadcast of the quad's lane 0 (column c = 0 of a tile) to its four lanes: DPP quad_perm [0,0,0,0], no LDS
__device__ __forceinline__ double quad_bcast0(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_mov_dpp(lo, 0x00, 0xf, 0xf, true);
  hi = __builtin_amdgcn_mov_dpp(hi, 0x00, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}

// When D = 1 (mod 4) the last K-step of a product carries ONE valid k (k = D-1; D = 9: a third of the MFMAs doing
// a quarter of their work).  It is applied as a rank-1 update on the vector unit instead:
//   acc(I,J) += a_I (x) b_J,   a_I[lane (r,c)] = A[4I+r][D-1]   (tile (I,NB-1), column 0, quad broadcast)
//                              b_J[lane (r,c)] = B[D-1][4J+c] = B[4J+c][D-1]   (tile (J,NB-1) of the lane (c,0))
// Both vectors come from tiles on or above the diagonal, i.e. they do not wait for the mirror.
template <int D>
struct SymTail {
  static constexpr bool ON = (D % 4 == 1) && (RD<D>::NB > 1);
  static constexpr int KM = ON ? RD<D>::NB - 1 : RD<D>::NB;
};

// first tile column of row I among the tiles kept for a symmetric matrix (operand rows in full, else upper tiles)
template <int D>
__device__ __forceinline__ constexpr int sym_j0(int I) {
  return I >= SymTail<D>::KM ? I : 0;
}
template <int D, bool FULL>
__device__ __forceinline__ void sym_fill(double (&m)[RD<D>::NB][RD<D>::NB], int swap_lane) {
  constexpr int NB = RD<D>::NB;
  constexpr int ROWS = FULL ? NB : SymTail<D>::KM;
#pragma unroll
  for (int I = 1; I < ROWS; ++I)
#pragma unroll
    for (int J = 0; J < I; ++J) m[I][J] = __shfl(m[J][I], swap_lane);
}
// the lower tiles sym_fill<D, false> left out
template <int D>
__device__ __forceinline__ void sym_fill_rest(double (&m)[RD<D>::NB][RD<D>::NB], int swap_lane) {
  constexpr int NB = RD<D>::NB;
#pragma unroll
  for (int I = SymTail<D>::KM; I < NB; ++I)
#pragma unroll
    for (int J = 0; J < I; ++J) m[I][J] = __shfl(m[J][I], swap_lane);
}

template <int D>
__device__ __forceinline__ void mm_sym(const double (&za)[RD<D>::NB][RD<D>::NB], const double (&zb)[RD<D>::NB][RD<D>::NB],
                                       double (&acc)[RD<D>::NB][RD<D>::NB], int tail_lane) {
  constexpr int NB = RD<D>::NB;
#pragma unroll
  for (int K = 0; K < SymTail<D>::KM; ++K)
#pragma unroll
    for (int I = 0; I < NB; ++I)
#pragma unroll
      for (int J = I; J < NB; ++J) acc[I][J] = mfma4(za[K][I], zb[K][J], acc[I][J]);
  if constexpr (SymTail<D>::ON) {
    double av[NB], bv[NB];
#pragma unroll
    for (int I = 0; I < NB; ++I) {
      av[I] = quad_bcast0(za[I][NB - 1]);
      bv[I] = __shfl(zb[I][NB - 1], tail_lane);
    }
#pragma unroll
    for (int I = 0; I < NB; ++I)
#pragma unroll
      for (int J = I; J < NB; ++J) acc[I][J] = fma(av[I], bv[J], acc[I][J]);
  }
}
template <int D>
__device__ __forceinline__ void mm_sym2(const double (&za)[RD<D>::NB][RD<D>::NB], const double (&zb1)[RD<D>::NB][RD<D>::NB],
                                        double (&acc1)[RD<D>::NB][RD<D>::NB], const double (&zb2)[RD<D>::NB][RD<D>::NB],
                                        double (&acc2)[RD<D>::NB][RD<D>::NB], int tail_lane) {
  constexpr int NB = RD<D>::NB;
#pragma unroll
  for (int K = 0; K < SymTail<D>::KM; ++K)
#pragma unroll
    for (int I = 0; I < NB; ++I)
#pragma unroll
      for (int J = I; J < NB; ++J) {
        acc1[I][J] = mfma4(za[K][I], zb1[K][J], acc1[I][J]);
        acc2[I][J] = mfma4(za[K][I], zb2[K][J], acc2[I][J]);
      }
  if constexpr (SymTail<D>::ON) {
    double av[NB], bv1[NB], bv2[NB];
#pragma unroll
    for (int I = 0; I < NB; ++I) {
      av[I] = quad_bcast0(za[I][NB - 1]);
      bv1[I] = __shfl(zb1[I][NB - 1], tail_lane);
      bv2[I] = __shfl(zb2[I][NB - 1], tail_lane);
    }
#pragma unroll
    for (int I = 0; I < NB; ++I)
#pragma unroll
      for (int J = I; J < NB; ++J) {
        acc1[I][J] = fma(av[I], bv1[J], acc1[I][J]);
        acc2[I][J] = fma(av[I], bv2[J], acc2[I][J]);
      }
  }
}

// acc += A B for a SYMMETRIC left operand A (A fragments = registers, as in mm_sym) and a general right operand
// B (all tiles): the chain product of the real path, E U with E = C - iS split into real blocks.
// The single-k tail takes the row D-1 of B from the lanes r = 0 (broadcast over r by one lane swap).
template <int D>
__device__ __forceinline__ void mm_symA(const double (&za)[RD<D>::NB][RD<D>::NB], const double (&zb)[RD<D>::NB][RD<D>::NB],
                                        double (&acc)[RD<D>::NB][RD<D>::NB], int row0_lane, double asign) {
  constexpr int NB = RD<D>::NB;
#pragma unroll
  for (int K = 0; K < SymTail<D>::KM; ++K)
#pragma unroll
    for (int I = 0; I < NB; ++I)
#pragma unroll
      for (int J = 0; J < NB; ++J) acc[I][J] = mfma4(za[K][I], zb[K][J], acc[I][J]);
  if constexpr (SymTail<D>::ON) {
    double av[NB], bv[NB];
#pragma unroll
    for (int I = 0; I < NB; ++I) {
      av[I] = asign * quad_bcast0(za[I][NB - 1]);
      bv[I] = __shfl(zb[NB - 1][I], row0_lane);
    }
#pragma unroll
    for (int I = 0; I < NB; ++I)
#pragma unroll
      for (int J = 0; J < NB; ++J) acc[I][J] = fma(av[I], bv[J], acc[I][J]);
  }
}

// acc (ALL tiles) += A B for symmetric A and B (operand tiles filled): the full product whose mirror gives
// A B + B A (backward sweep).  The single-k tail takes column D-1 of both from their last tile column.
template <int D>
__device__ __forceinline__ void mm_symfull(const double (&za)[RD<D>::NB][RD<D>::NB], const double (&zb)[RD<D>::NB][RD<D>::NB],
                                           double (&acc)[RD<D>::NB][RD<D>::NB], int tail_lane) {
  constexpr int NB = RD<D>::NB;
#pragma unroll
  for (int K = 0; K < SymTail<D>::KM; ++K)
#pragma unroll
    for (int I = 0; I < NB; ++I)
#pragma unroll
      for (int J = 0; J < NB; ++J) acc[I][J] = mfma4(za[K][I], zb[K][J], acc[I][J]);
  if constexpr (SymTail<D>::ON) {
    double av[NB], bv[NB];
#pragma unroll
    for (int I = 0; I < NB; ++I) {
      av[I] = quad_bcast0(za[I][NB - 1]);
      bv[I] = __shfl(zb[I][NB - 1], tail_lane);
    }
#pragma unroll
    for (int I = 0; I < NB; ++I)
#pragma unroll
      for (int J = 0; J < NB; ++J) acc[I][J] = fma(av[I], bv[J], acc[I][J]);
  }
}

// out = c0 I + c1 W + c2 W2 (+ c3 W3); UPPER: tiles J >= I only (accumulators of a symmetric product, mirrored
// afterwards); otherwise the tiles a product reads as operands (rows I < KM in full, upper tiles of the other rows)
template <int D, bool WITH3, bool UPPER = false>
__device__ __forceinline__ void rcomb(double (&out)[RD<D>::NB][RD<D>::NB], double c0, double c1, double c2, double c3,
                                      const double (&W1)[RD<D>::NB][RD<D>::NB], const double (&W2)[RD<D>::NB][RD<D>::NB],
                                      const double (&W3)[RD<D>::NB][RD<D>::NB], const LanePos& lp) {
  constexpr int NB = RD<D>::NB;
#pragma unroll
  for (int I = 0; I < NB; ++I)
#pragma unroll
    for (int J = (UPPER || I >= SymTail<D>::KM) ? I : 0; J < NB; ++J) {
      double v = c1 * W1[I][J];
      v = fma(c2, W2[I][J], v);
      if constexpr (WITH3) v = fma(c3, W3[I][J], v);
      if (I == J) v += (lp.r == lp.c && 4 * I + lp.r < D) ? c0 : 0.0;
      out[I][J] = v;
    }
}


// ---------------------------------------------------------------------------------------------
// "Core + border" form of the real path for D = 4 NC + 1 (D = 9: the 8 + 1 split).  A 9 x 9 matrix in
// 12 x 12 tiles spends 6 of the 12 MFMAs of a symmetric product (10 of the 18 of a chain product) on the tiles of its last
// row and column, which carry 4, 4 and 1 of 16 elements.  Here the (D-1) x (D-1) core tiles exactly (NC x NC tiles, no padding
// at all) and the last row / column / corner are carried separately, per chain, in the lanes of its MFMA block:
//   symmetric matrix (SMat): core tiles (upper, lower mirrored for operands), the border column twice -- vr[I] = M[4I+r][D-1]
//     (replicated over c) and vc[J] = M[4J+c][D-1] (replicated over r) --, the corner s (all 16 lanes);
//   general matrix (GMat, the chain state): core, column border cr (r form) / cc (c form), row border rc[J] = M[D-1][4J+c], s.
// Border of a product: column = core x vector on the vector unit (one FMA per tile and lane + a quad reduction by DPP),
// its c form by one lane swap; corner = dot product (quad reduction); the ROW border of a chain product is a vector-matrix
// product whose sum runs over the r lanes (16 apart: no DPP reach) -- it stays on the matrix cores as ONE A tile per K-step
// whose only row is the border (4 MFMAs instead of the 10 of the padded tiles).  Measured in isolation
// (tools/ubench_sym9.hip, profiles/r04/ubench_sym9.txt): a dependent symmetric product 284 ns against 315 ns per wave at two
// waves per SIMD.
// ---------------------------------------------------------------------------------------------
template <int NC>
struct SMat {
  double m[NC][NC];
  double vr[NC], vc[NC];
  double s;
};
template <int NC>
struct GMat {
  double m[NC][NC];
  double cr[NC], cc[NC], rc[NC];
  double s;
};

template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xf, 0xf, true);
  hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
// sum over the four lanes of a quad (the column index c), result in all four
__device__ __forceinline__ double quad_sum(double v) {
#if defined(C3P_SD_ABL) && (C3P_SD_ABL & 16)
  return v * 4.0;  // TIMING-ONLY (wrong results): no quad reductions
#endif
  v += dpp_f64<0xB1>(v);  // quad_perm [1,0,3,2]
  v += dpp_f64<0x4E>(v);  // quad_perm [2,3,0,1]
  return v;
}

template <int NC>
__device__ __forceinline__ void s8_zero(SMat<NC>& a) {
#pragma unroll
  for (int I = 0; I < NC; ++I) {
#pragma unroll
    for (int J = 0; J < NC; ++J) a.m[I][J] = 0.0;
    a.vr[I] = a.vc[I] = 0.0;
  }
  a.s = 0.0;
}

// Border sums on the matrix cores (round 6).  On gfx950 a wave's vector instructions and the fp64 matrix instructions of its
// SIMD do not overlap (tools/ubench_coissue.hip: 8 MFMAs + 32 v_mov_dpp take the SUM of their issue times), a quad reduction of a
// double costs four v_mov_b32_dpp + two v_add_f64 = ~29 cycles, one v_mfma_f64_4x4x4_4b 16.7 -- and the instruction contracts over
// the lane's r index for free.  So the partial products of a border are formed with the sum index ON r (for a symmetric left
// operand: its transposed tile, the register the lane holds anyway, times the r form of the vector) and reduced by ONE matrix
// instruction against a tile of ones: A = partials, B = 1 -> D[i][j] = sum_r t(r, c = i), i.e. the r form of the result in every lane,
// with the rest of the border update (a.vr b.s + c.vr) riding in as the accumulator.  -DC3P_SD_QUADSUM builds the DPP reductions.
#ifndef C3P_SD_QUADSUM
// C += A B for symmetric commuting A, B (upper core tiles, vr, s of C; s8_finish completes the operand form of C)
template <int NC>
__device__ __forceinline__ void mm_s8(const SMat<NC>& a, const SMat<NC>& b, SMat<NC>& c) {
  // (the rank-1 term of the border goes into the accumulators BEFORE the matrix instructions: a vector instruction that reads a
  // matrix result waits for the whole instruction, ~45 cycles; the other way round costs two wait states)
#pragma unroll
  for (int I = 0; I < NC; ++I)
#pragma unroll
    for (int J = I; J < NC; ++J) c.m[I][J] = fma(a.vr[I], b.vc[J], c.m[I][J]);
#pragma unroll
  for (int K = 0; K < NC; ++K)
#pragma unroll
    for (int I = 0; I < NC; ++I)
#pragma unroll
      for (int J = I; J < NC; ++J) c.m[I][J] = mfma4(a.m[K][I], b.m[K][J], c.m[I][J]);
  double t[NC], cs = 0.0;
#pragma unroll
  for (int I = 0; I < NC; ++I) {
    t[I] = 0.0;
#pragma unroll
    for (int K = 0; K < NC; ++K) t[I] = fma(a.m[K][I], b.vr[K], t[I]);  // lane (r, c): A[4I+c][4K+r] b[4K+r]
    cs = fma(a.vr[I], b.vr[I], cs);
  }
#pragma unroll
  for (int I = 0; I < NC; ++I) c.vr[I] = mfma4(t[I], 1.0, fma(a.vr[I], b.s, c.vr[I]));
  c.s = mfma4(cs, 1.0, fma(a.s, b.s, c.s));
}
// two products with one left operand, interleaved (C1 += A B1, C2 += A B2)
template <int NC>
__device__ __forceinline__ void mm_s8x2(const SMat<NC>& a, const SMat<NC>& b1, SMat<NC>& c1, const SMat<NC>& b2, SMat<NC>& c2) {
#pragma unroll
  for (int I = 0; I < NC; ++I)
#pragma unroll
    for (int J = I; J < NC; ++J) {
      c1.m[I][J] = fma(a.vr[I], b1.vc[J], c1.m[I][J]);
      c2.m[I][J] = fma(a.vr[I], b2.vc[J], c2.m[I][J]);
    }
#pragma unroll
  for (int K = 0; K < NC; ++K)
#pragma unroll
    for (int I = 0; I < NC; ++I)
#pragma unroll
      for (int J = I; J < NC; ++J) {
        c1.m[I][J] = mfma4(a.m[K][I], b1.m[K][J], c1.m[I][J]);
        c2.m[I][J] = mfma4(a.m[K][I], b2.m[K][J], c2.m[I][J]);
      }
  double t1[NC], t2[NC], cs1 = 0.0, cs2 = 0.0;
#pragma unroll
  for (int I = 0; I < NC; ++I) {
    t1[I] = t2[I] = 0.0;
#pragma unroll
    for (int K = 0; K < NC; ++K) {
      t1[I] = fma(a.m[K][I], b1.vr[K], t1[I]);
      t2[I] = fma(a.m[K][I], b2.vr[K], t2[I]);
    }
    cs1 = fma(a.vr[I], b1.vr[I], cs1);
    cs2 = fma(a.vr[I], b2.vr[I], cs2);
  }
#pragma unroll
  for (int I = 0; I < NC; ++I) {
    c1.vr[I] = mfma4(t1[I], 1.0, fma(a.vr[I], b1.s, c1.vr[I]));
    c2.vr[I] = mfma4(t2[I], 1.0, fma(a.vr[I], b2.s, c2.vr[I]));
  }
  c1.s = mfma4(cs1, 1.0, fma(a.s, b1.s, c1.s));
  c2.s = mfma4(cs2, 1.0, fma(a.s, b2.s, c2.s));
}
#else
// C += A B for symmetric commuting A, B (upper core tiles, vr, s of C; s8_finish completes the operand form of C)
template <int NC>
__device__ __forceinline__ void mm_s8(const SMat<NC>& a, const SMat<NC>& b, SMat<NC>& c) {
#pragma unroll
  for (int K = 0; K < NC; ++K)
#pragma unroll
    for (int I = 0; I < NC; ++I)
#pragma unroll
      for (int J = I; J < NC; ++J) c.m[I][J] = mfma4(a.m[K][I], b.m[K][J], c.m[I][J]);
  double t[NC], cs = 0.0;
#pragma unroll
  for (int I = 0; I < NC; ++I) {
    t[I] = 0.0;
#pragma unroll
    for (int K = 0; K < NC; ++K) t[I] = fma(a.m[I][K], b.vc[K], t[I]);
    cs = fma(a.vc[I], b.vc[I], cs);
  }
#pragma unroll
  for (int I = 0; I < NC; ++I) c.vr[I] += fma(a.vr[I], b.s, quad_sum(t[I]));
  c.s += fma(a.s, b.s, quad_sum(cs));
#pragma unroll
  for (int I = 0; I < NC; ++I)
#pragma unroll
    for (int J = I; J < NC; ++J) c.m[I][J] = fma(a.vr[I], b.vc[J], c.m[I][J]);
}
// two products with one left operand, interleaved (C1 += A B1, C2 += A B2)
template <int NC>
__device__ __forceinline__ void mm_s8x2(const SMat<NC>& a, const SMat<NC>& b1, SMat<NC>& c1, const SMat<NC>& b2, SMat<NC>& c2) {
#pragma unroll
  for (int K = 0; K < NC; ++K)
#pragma unroll
    for (int I = 0; I < NC; ++I)
#pragma unroll
      for (int J = I; J < NC; ++J) {
        c1.m[I][J] = mfma4(a.m[K][I], b1.m[K][J], c1.m[I][J]);
        c2.m[I][J] = mfma4(a.m[K][I], b2.m[K][J], c2.m[I][J]);
      }
  double t1[NC], t2[NC], cs1 = 0.0, cs2 = 0.0;
#pragma unroll
  for (int I = 0; I < NC; ++I) {
    t1[I] = t2[I] = 0.0;
#pragma unroll
    for (int K = 0; K < NC; ++K) {
      t1[I] = fma(a.m[I][K], b1.vc[K], t1[I]);
      t2[I] = fma(a.m[I][K], b2.vc[K], t2[I]);
    }
    cs1 = fma(a.vc[I], b1.vc[I], cs1);
    cs2 = fma(a.vc[I], b2.vc[I], cs2);
  }
#pragma unroll
  for (int I = 0; I < NC; ++I) {
    c1.vr[I] += fma(a.vr[I], b1.s, quad_sum(t1[I]));
    c2.vr[I] += fma(a.vr[I], b2.s, quad_sum(t2[I]));
  }
  c1.s += fma(a.s, b1.s, quad_sum(cs1));
  c2.s += fma(a.s, b2.s, quad_sum(cs2));
#pragma unroll
  for (int I = 0; I < NC; ++I)
#pragma unroll
    for (int J = I; J < NC; ++J) {
      c1.m[I][J] = fma(a.vr[I], b1.vc[J], c1.m[I][J]);
      c2.m[I][J] = fma(a.vr[I], b2.vc[J], c2.m[I][J]);
    }
}
#endif
// operand form of a product: lower core tiles by the in-chain lane swap, the c form of the border column
template <int NC>
__device__ __forceinline__ void s8_finish(SMat<NC>& a, int swap_lane, int tail_lane) {
#if defined(C3P_SD_ABL) && (C3P_SD_ABL & 1)
  // TIMING-ONLY (wrong results): no lane swaps in the symmetric stage
#pragma unroll
  for (int I = 1; I < NC; ++I)
#pragma unroll
    for (int J = 0; J < I; ++J) a.m[I][J] = a.m[J][I];
#pragma unroll
  for (int J = 0; J < NC; ++J) a.vc[J] = a.vr[J];
  return;
#endif
#pragma unroll
  for (int I = 1; I < NC; ++I)
#pragma unroll
    for (int J = 0; J < I; ++J) a.m[I][J] = __shfl(a.m[J][I], swap_lane);
#pragma unroll
  for (int J = 0; J < NC; ++J) a.vc[J] = __shfl(a.vr[J], tail_lane);
}
// LEFT-operand form of a product (core tiles complete, vr, s): the lower core tiles alone -- a symmetric left operand is read
// through its tiles, its r-form border and its corner only (mm_s8, chain_step8), the c form is a right operand's business
template <int NC>
__device__ __forceinline__ void s8_finish_lower(SMat<NC>& a, int swap_lane) {
#pragma unroll
  for (int I = 1; I < NC; ++I)
#pragma unroll
    for (int J = 0; J < I; ++J) a.m[I][J] = __shfl(a.m[J][I], swap_lane);
}
template <int NC>
__device__ __forceinline__ void s8_finish_vc(SMat<NC>& a, int tail_lane) {
#pragma unroll
  for (int J = 0; J < NC; ++J) a.vc[J] = __shfl(a.vr[J], tail_lane);
}
#ifndef C3P_SD_QUADSUM
// two products with one RIGHT operand, interleaved (C1 += A1 B, C2 += A2 B): A1, A2 in left-operand form
template <int NC>
__device__ __forceinline__ void mm_s8x2r(const SMat<NC>& a1, const SMat<NC>& a2, const SMat<NC>& b, SMat<NC>& c1, SMat<NC>& c2) {
#pragma unroll
  for (int I = 0; I < NC; ++I)
#pragma unroll
    for (int J = I; J < NC; ++J) {
      c1.m[I][J] = fma(a1.vr[I], b.vc[J], c1.m[I][J]);
      c2.m[I][J] = fma(a2.vr[I], b.vc[J], c2.m[I][J]);
    }
#pragma unroll
  for (int K = 0; K < NC; ++K)
#pragma unroll
    for (int I = 0; I < NC; ++I)
#pragma unroll
      for (int J = I; J < NC; ++J) {
        c1.m[I][J] = mfma4(a1.m[K][I], b.m[K][J], c1.m[I][J]);
        c2.m[I][J] = mfma4(a2.m[K][I], b.m[K][J], c2.m[I][J]);
      }
  double t1[NC], t2[NC], cs1 = 0.0, cs2 = 0.0;
#pragma unroll
  for (int I = 0; I < NC; ++I) {
    t1[I] = t2[I] = 0.0;
#pragma unroll
    for (int K = 0; K < NC; ++K) {
      t1[I] = fma(a1.m[K][I], b.vr[K], t1[I]);
      t2[I] = fma(a2.m[K][I], b.vr[K], t2[I]);
    }
    cs1 = fma(a1.vr[I], b.vr[I], cs1);
    cs2 = fma(a2.vr[I], b.vr[I], cs2);
  }
#pragma unroll
  for (int I = 0; I < NC; ++I) {
    c1.vr[I] = mfma4(t1[I], 1.0, fma(a1.vr[I], b.s, c1.vr[I]));
    c2.vr[I] = mfma4(t2[I], 1.0, fma(a2.vr[I], b.s, c2.vr[I]));
  }
  c1.s = mfma4(cs1, 1.0, fma(a1.s, b.s, c1.s));
  c2.s = mfma4(cs2, 1.0, fma(a2.s, b.s, c2.s));
}
#endif
// out = c0 I + c1 W1 + c2 W2 (+ c3 W3), every part (operands are complete: the combination is too); ACCUM: only what a
// product accumulates into (upper core tiles, vr, s) -- the initial value of an accumulation that s8_finish completes later
template <int NC, bool WITH3, bool ACCUM = false>
__device__ __forceinline__ void s8_comb(SMat<NC>& out, double c0, double c1, double c2, double c3, const SMat<NC>& W1, const SMat<NC>& W2,
                                        const SMat<NC>& W3, const LanePos& lp) {
  // (the c0 I term rides in as the addend of the first multiply-add: no separate add on the diagonal tiles and the corner)
  auto lc = [&](double x1, double x2, double x3, double add) {
    double v = fma(c1, x1, add);
    v = fma(c2, x2, v);
    if constexpr (WITH3) v = fma(c3, x3, v);
    return v;
  };
  const double dg = (lp.r == lp.c) ? c0 : 0.0;
#pragma unroll
  for (int I = 0; I < NC; ++I) {
#pragma unroll
    for (int J = ACCUM ? I : 0; J < NC; ++J) out.m[I][J] = lc(W1.m[I][J], W2.m[I][J], W3.m[I][J], I == J ? dg : 0.0);
    out.vr[I] = lc(W1.vr[I], W2.vr[I], W3.vr[I], 0.0);
    if constexpr (!ACCUM) out.vc[I] = lc(W1.vc[I], W2.vc[I], W3.vc[I], 0.0);
  }
  out.s = lc(W1.s, W2.s, W3.s, c0);
}


// chain step in real blocks for the core + border form: (Ur + i Ui) <- (C - i S)(Ur + i Ui) with three real products
// (T1 = C Ur, T2 = S Ui, T3 = (C - S)(Ur + Ui): Re = T1 + T2, Im = T3 - T1 + T2), C and S symmetric (complete operand form).
// Round 6: no lane swaps and no DPP reductions.  The column border and the corner are reduced over r by a matrix instruction
// against ones (see mm_s8; the partial products use the TRANSPOSED tile of the symmetric left operand and the r form `cr` of the
// state's column border, so the c form `cc` is not carried any more); the row border comes out of its matrix instructions already
// replicated over r, because the A tile holds the border row of the left operand in EVERY row instead of row 0 alone.
#ifndef C3P_SD_QUADSUM
template <int NC>
__device__ __forceinline__ void chain_step8(const SMat<NC>& Cc, const SMat<NC>& Sc, GMat<NC>& Ur, GMat<NC>& Ui, const LanePos& lp,
                                            int tail_lane, int row0_lane) {
  (void)lp, (void)tail_lane, (void)row0_lane;
  SMat<NC> Dm;
  GMat<NC> Us;
#pragma unroll
  for (int I = 0; I < NC; ++I) {
#pragma unroll
    for (int J = 0; J < NC; ++J) {
      Dm.m[I][J] = Cc.m[I][J] - Sc.m[I][J];
      Us.m[I][J] = Ur.m[I][J] + Ui.m[I][J];
    }
    Dm.vr[I] = Cc.vr[I] - Sc.vr[I];
    Us.cr[I] = Ur.cr[I] + Ui.cr[I];
    Us.rc[I] = Ur.rc[I] + Ui.rc[I];
  }
  Dm.s = Cc.s - Sc.s;
  Us.s = Ur.s + Ui.s;
  double T1[NC][NC], T2[NC][NC], T3[NC][NC], R1[NC], R2[NC], R3[NC];
#pragma unroll
  for (int I = 0; I < NC; ++I) {
#pragma unroll
    for (int J = 0; J < NC; ++J) T1[I][J] = T2[I][J] = T3[I][J] = 0.0;
    R1[I] = R2[I] = R3[I] = 0.0;
  }
#pragma unroll
  for (int K = 0; K < NC; ++K) {
#pragma unroll
    for (int I = 0; I < NC; ++I)
#pragma unroll
      for (int J = 0; J < NC; ++J) {
        T1[I][J] = mfma4(Cc.m[K][I], Ur.m[K][J], T1[I][J]);
        T2[I][J] = mfma4(Sc.m[K][I], Ui.m[K][J], T2[I][J]);
        T3[I][J] = mfma4(Dm.m[K][I], Us.m[K][J], T3[I][J]);
      }
    // row border: A tile = the border row of the left operand in every row (vr[K]: lane (r, c) holds M[4K+r][D-1] = M[D-1][4K+r])
#pragma unroll
    for (int J = 0; J < NC; ++J) {
      R1[J] = mfma4(Cc.vr[K], Ur.m[K][J], R1[J]);
      R2[J] = mfma4(Sc.vr[K], Ui.m[K][J], R2[J]);
      R3[J] = mfma4(Dm.vr[K], Us.m[K][J], R3[J]);
    }
  }
  // column border and corner: partial products with the sum index on r, the three products combined before the reduction
  double pr[NC], pi[NC], cr_ = 0.0, ci_ = 0.0;
#pragma unroll
  for (int I = 0; I < NC; ++I) {
    double p1 = 0.0, p2 = 0.0, p3 = 0.0;
#pragma unroll
    for (int K = 0; K < NC; ++K) {
      p1 = fma(Cc.m[K][I], Ur.cr[K], p1);
      p2 = fma(Sc.m[K][I], Ui.cr[K], p2);
      p3 = fma(Dm.m[K][I], Us.cr[K], p3);
    }
    pr[I] = p1 + p2;
    pi[I] = (p3 - p1) + p2;
    const double c1 = Cc.vr[I] * Ur.cr[I], c2 = Sc.vr[I] * Ui.cr[I], c3 = Dm.vr[I] * Us.cr[I];
    cr_ += c1 + c2;
    ci_ += (c3 - c1) + c2;
  }
  const double e1 = Cc.s * Ur.s, e2 = Sc.s * Ui.s, e3 = Dm.s * Us.s;
  const double sr = mfma4(cr_, 1.0, e1 + e2), si = mfma4(ci_, 1.0, (e3 - e1) + e2);
  double colr[NC], coli[NC], rowr[NC], rowi[NC];
#pragma unroll
  for (int I = 0; I < NC; ++I) {
    const double q1 = Cc.vr[I] * Ur.s, q2 = Sc.vr[I] * Ui.s, q3 = Dm.vr[I] * Us.s;
    colr[I] = mfma4(pr[I], 1.0, q1 + q2);
    coli[I] = mfma4(pi[I], 1.0, (q3 - q1) + q2);
    // row border: the k = D-1 term on top of the matrix-core sums (every lane holds the row)
    const double r1 = fma(Cc.s, Ur.rc[I], R1[I]), r2 = fma(Sc.s, Ui.rc[I], R2[I]), r3 = fma(Dm.s, Us.rc[I], R3[I]);
    rowr[I] = r1 + r2;
    rowi[I] = (r3 - r1) + r2;
  }
#pragma unroll
  for (int I = 0; I < NC; ++I)
#pragma unroll
    for (int J = 0; J < NC; ++J) {
      const double t1 = fma(Cc.vr[I], Ur.rc[J], T1[I][J]), t2 = fma(Sc.vr[I], Ui.rc[J], T2[I][J]), t3 = fma(Dm.vr[I], Us.rc[J], T3[I][J]);
      Ur.m[I][J] = t1 + t2;
      Ui.m[I][J] = (t3 - t1) + t2;
    }
#pragma unroll
  for (int I = 0; I < NC; ++I) {
    Ur.cr[I] = colr[I];
    Ui.cr[I] = coli[I];
    Ur.rc[I] = rowr[I];
    Ui.rc[I] = rowi[I];
  }
  Ur.s = sr;
  Ui.s = si;
}
#else
template <int NC>
__device__ __forceinline__ void chain_step8(const SMat<NC>& Cc, const SMat<NC>& Sc, GMat<NC>& Ur, GMat<NC>& Ui, const LanePos& lp,
                                            int tail_lane, int row0_lane) {
  SMat<NC> Dm;
  GMat<NC> Us;
#pragma unroll
  for (int I = 0; I < NC; ++I) {
#pragma unroll
    for (int J = 0; J < NC; ++J) {
      Dm.m[I][J] = Cc.m[I][J] - Sc.m[I][J];
      Us.m[I][J] = Ur.m[I][J] + Ui.m[I][J];
    }
    Dm.vr[I] = Cc.vr[I] - Sc.vr[I];
    Dm.vc[I] = Cc.vc[I] - Sc.vc[I];
    Us.cc[I] = Ur.cc[I] + Ui.cc[I];
    Us.rc[I] = Ur.rc[I] + Ui.rc[I];
  }
  Dm.s = Cc.s - Sc.s;
  Us.s = Ur.s + Ui.s;
  double T1[NC][NC], T2[NC][NC], T3[NC][NC], R1[NC], R2[NC], R3[NC];
  double aC[NC], aS[NC], aD[NC];
#pragma unroll
  for (int I = 0; I < NC; ++I) {
#pragma unroll
    for (int J = 0; J < NC; ++J) T1[I][J] = T2[I][J] = T3[I][J] = 0.0;
    R1[I] = R2[I] = R3[I] = 0.0;
    // A tile whose only row (i = 0: lanes c = 0) is the border row of the symmetric left operand
    aC[I] = lp.c == 0 ? Cc.vr[I] : 0.0;
    aS[I] = lp.c == 0 ? Sc.vr[I] : 0.0;
    aD[I] = lp.c == 0 ? Dm.vr[I] : 0.0;
  }
#pragma unroll
  for (int K = 0; K < NC; ++K) {
#pragma unroll
    for (int I = 0; I < NC; ++I)
#pragma unroll
      for (int J = 0; J < NC; ++J) {
        T1[I][J] = mfma4(Cc.m[K][I], Ur.m[K][J], T1[I][J]);
        T2[I][J] = mfma4(Sc.m[K][I], Ui.m[K][J], T2[I][J]);
        T3[I][J] = mfma4(Dm.m[K][I], Us.m[K][J], T3[I][J]);
      }
#pragma unroll
    for (int J = 0; J < NC; ++J) {
      R1[J] = mfma4(aC[K], Ur.m[K][J], R1[J]);
      R2[J] = mfma4(aS[K], Ui.m[K][J], R2[J]);
      R3[J] = mfma4(aD[K], Us.m[K][J], R3[J]);
    }
  }
  // column border and corner: partial sums over the lane's column index, reduced after the three products are combined
  double pr[NC], pi[NC], cr_ = 0.0, ci_ = 0.0;
#pragma unroll
  for (int I = 0; I < NC; ++I) {
    double p1 = 0.0, p2 = 0.0, p3 = 0.0;
#pragma unroll
    for (int K = 0; K < NC; ++K) {
      p1 = fma(Cc.m[I][K], Ur.cc[K], p1);
      p2 = fma(Sc.m[I][K], Ui.cc[K], p2);
      p3 = fma(Dm.m[I][K], Us.cc[K], p3);
    }
    pr[I] = p1 + p2;
    pi[I] = (p3 - p1) + p2;
    const double c1 = Cc.vc[I] * Ur.cc[I], c2 = Sc.vc[I] * Ui.cc[I], c3 = Dm.vc[I] * Us.cc[I];
    cr_ += c1 + c2;
    ci_ += (c3 - c1) + c2;
  }
  const double e1 = Cc.s * Ur.s, e2 = Sc.s * Ui.s, e3 = Dm.s * Us.s;
  const double sr = quad_sum(cr_) + (e1 + e2), si = quad_sum(ci_) + ((e3 - e1) + e2);
  double colr[NC], coli[NC], rowr[NC], rowi[NC];
#pragma unroll
  for (int I = 0; I < NC; ++I) {
    const double q1 = Cc.vr[I] * Ur.s, q2 = Sc.vr[I] * Ui.s, q3 = Dm.vr[I] * Us.s;
    colr[I] = quad_sum(pr[I]) + (q1 + q2);
    coli[I] = quad_sum(pi[I]) + ((q3 - q1) + q2);
    // row border: the k = D-1 term on top of the matrix-core sums (lanes r = 0 hold the row)
    const double r1 = fma(Cc.s, Ur.rc[I], R1[I]), r2 = fma(Sc.s, Ui.rc[I], R2[I]), r3 = fma(Dm.s, Us.rc[I], R3[I]);
    rowr[I] = r1 + r2;
    rowi[I] = (r3 - r1) + r2;
  }
#pragma unroll
  for (int I = 0; I < NC; ++I)
#pragma unroll
    for (int J = 0; J < NC; ++J) {
      const double t1 = fma(Cc.vr[I], Ur.rc[J], T1[I][J]), t2 = fma(Sc.vr[I], Ui.rc[J], T2[I][J]), t3 = fma(Dm.vr[I], Us.rc[J], T3[I][J]);
      Ur.m[I][J] = t1 + t2;
      Ui.m[I][J] = (t3 - t1) + t2;
    }
#pragma unroll
  for (int I = 0; I < NC; ++I) {
    Ur.cr[I] = colr[I];
    Ui.cr[I] = coli[I];
#if defined(C3P_SD_ABL) && (C3P_SD_ABL & 8)
    Ur.cc[I] = colr[I], Ui.cc[I] = coli[I], Ur.rc[I] = rowr[I], Ui.rc[I] = rowi[I];  // TIMING-ONLY: no lane swaps in the chain step
#else
    Ur.cc[I] = __shfl(colr[I], tail_lane);
    Ui.cc[I] = __shfl(coli[I], tail_lane);
    Ur.rc[I] = __shfl(rowr[I], row0_lane);
    Ui.rc[I] = __shfl(rowi[I], row0_lane);
#endif
  }
  Ur.s = sr;
  Ui.s = si;
}

#endif

// q = 4 plan: degree 4r, s squarings, from a bound on ||X||_1
__device__ __forceinline__ void plan_q4(double nrm, int& r, int& s) {
  // Taylor backward-error bounds for unit roundoff 2^-52 (theta_m of Al-Mohy & Higham scaled by 2^(1/m))
  const double th[5] = {4.0e-4, 5.45e-2, 3.18e-1, 8.16e-1, 1.49};
  int best_r = 5, best_s = 0, best_cost = 1 << 30;
#pragma unroll
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
  r = best_r;
  s = best_s;
}

// Write a D-layout matrix times the complex scalar (sr + i si), with optional row phases,
// to a plain complex [D][D] array.
// cos / sin of the row phases of the lane's rows (store_plain takes them instead of evaluating sincos in the store)
template <int D>
__device__ __forceinline__ void row_phase_factors(const double* row_phase, const LanePos& lp, double (&pc)[SD<D>::NBI],
                                                  double (&ps)[SD<D>::NBI]) {
#pragma unroll
  for (int I = 0; I < SD<D>::NBI; ++I) {
    const int row = 2 * I + (lp.r >> 1);
    pc[I] = 1.0, ps[I] = 0.0;
    if (row_phase != nullptr && row < D) sincos(row_phase[row], &ps[I], &pc[I]);
  }
}

template <int D, bool WRITE_THROUGH = false>
__device__ __forceinline__ void store_plain(const double (&zh)[SD<D>::NBI][SD<D>::NJ], double* dst,
                                            double sr, double si, const double* row_phase,
                                            const LanePos& lp, bool active, const double* pre_c = nullptr,
                                            const double* pre_s = nullptr) {
  using C = SD<D>;
#pragma unroll
  for (int I = 0; I < C::NBI; ++I) {
    const int row = 2 * I + (lp.r >> 1);
    double pr = sr, pi = si;
    if (pre_c != nullptr) {
      pr = sr * pre_c[I] - si * pre_s[I];
      pi = sr * pre_s[I] + si * pre_c[I];
    } else if (row_phase != nullptr && row < D) {
      double sn, cs;
      sincos(row_phase[row], &sn, &cs);
      pr = sr * cs - si * sn;
      pi = sr * sn + si * cs;
    }
#pragma unroll
    for (int J = 0; J < C::NJ; ++J) {
      const double mine = zh[I][J];
      const double other = __shfl_xor(mine, 16);
      // r even: mine = Re, other = Im ; r odd: mine = Im, other = Re
      const double outv = (lp.r & 1) ? fma(pr, mine, pi * other) : fma(pr, mine, -pi * other);
      const int col = 4 * J + lp.c;
      if (active && row < D && col < D) {
        double* p = dst + (row * D + col) * 2 + (lp.r & 1);
        if constexpr (WRITE_THROUGH) {
          // sc1 (write-through) store: visible to other CUs/XCDs without a release fence
          __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), (unsigned long long)__double_as_longlong(outv),
                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
          *p = outv;
        }
      }
    }
  }
}

__device__ __forceinline__ double wave_sum64(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ double wave_max64(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o));
  return v;
}

// The wave builds its own tables in LDS (unitary mode): G = -i dt (h - tr h / D) as half images plus
// {Re mu, Im mu, ||G||_1, max |Re G|} -- what smalld_prep_kernel writes, without a dependent launch in
// front of the chain kernel (~12 us of a 250 us batch).
template <int D>
__device__ __forceinline__ void build_tables(const SmallArgs& A, int sample, double* tab, int lane, int first = 0, int step = 1) {
  using C = SD<D>;
  constexpr int MAT = C::MAT, W = C::W;
  constexpr int NE = (D * D + 63) / 64;
  for (int ti = first; ti <= A.K; ti += step) {  // (a workgroup of several waves deals the tables to its waves)
    double* out = tab + ti * (MAT + 4);
    const cplx* h = (ti == 0) ? A.h0 + (long)sample * A.h0_bstride
                              : A.hks + (long)sample * A.hks_bstride + (long)(ti - 1) * D * D;
    for (int e = lane; e < MAT; e += 64) out[e] = 0.0;
    double gr[NE], gi[NE];
    double tim = 0.0;
#pragma unroll
    for (int q = 0; q < NE; ++q) {
      const int e = lane + 64 * q;
      gr[q] = gi[q] = 0.0;
      if (e < D * D) {
        const cplx x = h[e];
        gr[q] = x.y * A.dt;  // -i dt h
        gi[q] = -x.x * A.dt;
        if (e / D == e % D) tim += gi[q];
      }
    }
    // (round 6) the shift is IMAGINARY only: a real shift (lossy Hamiltonians) makes the shifted chain product grow like
    // e^{|Re mu| n} while e^{sum mu} underflows -- inf * 0 over a long segment (tools/fuzz_r06.py found it on the Lindblad tables)
    const double mur = 0.0, mui = wave_sum64(tim) / D;
    wave_sync();
    double remax = 0.0;
#pragma unroll
    for (int q = 0; q < NE; ++q) {
      const int e = lane + 64 * q;
      if (e < D * D) {
        const int i = e / D, j = e - i * D;
        if (i == j) {
          gr[q] -= mur;
          gi[q] -= mui;
        }
        out[(2 * i) * W + j] = gr[q];
        out[(2 * i + 1) * W + j] = gi[q];
        remax = fmax(remax, fabs(gr[q]));
      }
    }
    // column sums of |G| (the 1-norm).  Round 6: plain sqrt(a^2 + b^2) on unrolled, independent LDS reads instead of D
    // dependent hypot() calls (the elements are dt x Hamiltonian entries: no overflow to guard) -- this function sits on the
    // critical path of every workgroup's prologue
    wave_sync();
    double cs = 0.0;
    if (lane < D) {
      double ar[D], ai[D];
#pragma unroll
      for (int i = 0; i < D; ++i) ar[i] = out[(2 * i) * W + lane], ai[i] = out[(2 * i + 1) * W + lane];
#pragma unroll
      for (int i = 0; i < D; ++i) cs += sqrt(fma(ar[i], ar[i], ai[i] * ai[i]));
    }
    const double nrm = wave_max64(cs);
    // the real fast path also needs Y = -Im G symmetric (Hermitian input).  Dressed operators V^T H V are
    // symmetric only to rounding (measured 5e-16 relative), so asymmetry below 1e-14 ||G|| counts as none;
    // anything larger is folded into the flag and sends the sample to the complex path.
    double asym = 0.0, rsk = 0.0;
#pragma unroll
    for (int q = 0; q < NE; ++q) {
      const int e = lane + 64 * q;
      if (e < D * D) {
        const int i = e / D, j = e - i * D;
        asym = fmax(asym, fabs(gi[q] - out[(2 * j + 1) * W + i]));
        rsk = fmax(rsk, fabs(gr[q] + out[(2 * j) * W + i]));  // G + G^H = 0 <=> Hermitian Hamiltonian (diagonal: 2 Re G_ii)
      }
    }
    asym = wave_max64(asym);
    rsk = wave_max64(rsk);
    if (asym > 1e-14 * nrm) remax = fmax(remax, asym);
    remax = wave_max64(remax);
    // (round 6) NEGATIVE: complex but skew-Hermitian to rounding -- a normal generator: T18 with the economised parameters
    if (remax > 0.0 && fmax(asym, rsk) <= 1e-14 * nrm) remax = -remax;
    if (lane == 0) {
      out[MAT + 0] = mur;
      out[MAT + 1] = mui;
      out[MAT + 2] = nrm;
      out[MAT + 3] = remax;
    }
  }
  wave_sync();
}

// XG: the generators are supplied per slice (branch B of pwc, propagation.py:295-308, and c3p_expm):
// X_n = coef * hs[b,n] - mu_n with (mu_n, ||X_n - mu_n||_1) from the hmeta pre-pass; no tables, no signals.
// MW (table modes, fused combine): ONE WORKGROUP PER SAMPLE, S / 4 waves of four chains each (cfg2: 8 waves = two per SIMD
// of one CU).  The tables are built once per workgroup and the waves' partial products meet in LDS behind a barrier,
// instead of once per wave and through global memory behind a ticket: the fixed cost of a cfg2 batch (29 us of 149:
// combine 12, table build 4, launch and prologue 13) is what this mode attacks.
template <int D, bool GIVEN, bool DUS, bool XG = false, bool MW = false, bool SPLIT = false>
__global__ void __launch_bounds__((MW ? 512 : 64), 2) smalld_chain_kernel(SmallArgs A) {
  using C = SD<D>;
  constexpr int NBI = C::NBI, NJ = C::NJ, W = C::W, MAT = C::MAT, IMG = C::IMG;
  const int lane = MW ? (threadIdx.x & 63) : threadIdx.x;
  const int wv = MW ? __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) : 0;
  const int nwv = MW ? (int)(blockDim.x >> 6) : 1;
#ifdef C3P_SD_TIMING
  long long tk0 = wall_clock64(), tk1 = 0, tk2 = 0, tk3 = 0, tk4 = 0, tk5 = 0, tk6 = 0, tk7 = 0, tk8 = 0, tk9 = 0;
#define SD_TICK(v) v = wall_clock64()
#else
#define SD_TICK(v)
#endif
  LanePos lp;
  lp.r = lane >> 4;
  lp.b = (lane >> 2) & 3;
  lp.c = lane & 3;
  lp.idx16 = lp.r * 4 + lp.c;
  const int K = A.K;
  double* tab = c3p_sd_lds;  // (1+K) images + scalars (table modes only; shared by the waves of a workgroup)
  const int SG = (K * A.Lmax) | 1;
  double* img = tab + ((GIVEN || XG) ? 0 : (1 + K) * (MAT + 4)) + wv * (4 * IMG + 4 * SG);  // this wave's 4 chain images
  double* sg = img + 4 * IMG;  // 4 chains x K x Lmax signals, odd chain stride (bank spread)
  double* part = tab + (1 + K) * (MAT + 4) + nwv * (4 * IMG + 4 * SG);  // MW: the waves' partial products [nwv][D][D][2]

  // XCD-aware block order: workgroups are dealt round-robin to the 8 XCDs, each with its own L2.  Logical block
  // (bid % 8) * (nb / 8) + bid / 8 keeps consecutive logical blocks -- the waves of one sample, which exchange their
  // segment partials in the fused combine -- on ONE XCD.
  long lblock = blockIdx.x;
  if constexpr (MW)
    lblock = (long)blockIdx.x * nwv + wv;  // workgroup = sample, wave = four consecutive segments
  else if ((gridDim.x & 7) == 0)
    lblock = (long)(blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  const long chain = lblock * 4 + lp.b;
  const long nchains = (long)A.B * A.S;
  const bool valid = chain < nchains;
  const long cc = valid ? chain : nchains - 1;
  const int sample = (int)(cc / A.S);
  const int seg = (int)(cc - (long)sample * A.S);
  int n0 = (int)(((long)seg * A.N) / A.S);
  int n1 = (int)(((long)(seg + 1) * A.N) / A.S);
  int len, tmax;
  // uneven segments (MW): the first S / 2 chains (waves 0 .. 3, the older wave of each SIMD) take `seg_long` slices each, the
  // others share the rest; the wave iterates over the longest of its four chains
  auto split_segments = [&](int seg_long) {
    if (MW && seg_long > 0) {
      const int h = A.S >> 1;
      const long rest = (long)A.N - (long)h * seg_long;
      n0 = seg < h ? seg * seg_long : h * seg_long + (int)(((long)(seg - h) * rest) / h);
      n1 = seg + 1 <= h ? (seg + 1) * seg_long : h * seg_long + (int)(((long)(seg + 1 - h) * rest) / h);
    }
    len = n1 - n0;
    tmax = A.Lmax;
    if (MW && seg_long > 0) {
      int m = len;
      m = max(m, __shfl_xor(m, 4));
      m = max(m, __shfl_xor(m, 8));
      tmax = __builtin_amdgcn_readfirstlane(m);
    }
  };
  split_segments(A.seg_long);

  // per-lane LDS offsets (doubles)
  const int woff = lp.b * IMG + lp.r * W + lp.c;
  const int roff = lp.b * IMG + (2 * (lp.c >> 1) + ((lp.c ^ lp.r) & 1)) * W + (lp.r >> 1);
  const unsigned negmask = (((lp.c & 1) == 0) && ((lp.r & 1) == 1)) ? 0x80000000u : 0u;
  const int toff = lp.r * W + lp.c;  // table read offset (D-layout)
  const int ddelta = (lp.r & 1) ? 1000 : ((lp.r >> 1) - lp.c);
  const int rhalf = lp.r >> 1;

  double U[NBI][NJ];
  double mus_r = 0.0, mus_i = 0.0;

  if constexpr (GIVEN) {
    // ---- ordered product of supplied matrices ----
    const double* base = reinterpret_cast<const double*>(A.mats) + ((long)sample * A.N + n0) * D * D * 2;
    for (int t = 0; t < tmax; ++t) {
      const bool act = valid && t < len;
      double P[NBI][NJ];
      const double* src = base + (long)(act ? t : 0) * D * D * 2;
#pragma unroll
      for (int I = 0; I < NBI; ++I)
#pragma unroll
        for (int J = 0; J < NJ; ++J) {
          const int row = 2 * I + (lp.r >> 1), col = 4 * J + lp.c;
          P[I][J] = (act && row < D && col < D) ? src[(row * D + col) * 2 + (lp.r & 1)] : 0.0;
        }
      if (t == 0) {
#pragma unroll
        for (int I = 0; I < NBI; ++I)
#pragma unroll
          for (int J = 0; J < NJ; ++J) U[I][J] = P[I][J];
      } else {
        double acc[NBI][NJ];
#pragma unroll
        for (int I = 0; I < NBI; ++I)
#pragma unroll
          for (int J = 0; J < NJ; ++J) acc[I][J] = 0.0;
        if (A.right_order) {
          write_image<D>(U, img, woff);
          mm_img<D>(img, roff, negmask, P, acc);
        } else {
          write_image<D>(P, img, woff);
          mm_img<D>(img, roff, negmask, U, acc);
        }
#pragma unroll
        for (int I = 0; I < NBI; ++I)
#pragma unroll
          for (int J = 0; J < NJ; ++J) U[I][J] = act ? acc[I][J] : U[I][J];
      }
    }
  } else {
    // ---- prologue: tables and this segment's control amplitudes into LDS ----
    // all four chains of a wave share the sample when tables are per sample (S % 4 == 0)
    double nrm = 0.0;
    if constexpr (XG) {
      const double* mt = A.meta + ((long)sample * A.N + n0) * 4;
      for (int t = lp.idx16; t < A.Lmax; t += 16)
        if (valid && t < len) nrm = fmax(nrm, mt[(long)t * 4 + 2]);
      nrm = fmax(nrm, __shfl_xor(nrm, 1));
      nrm = fmax(nrm, __shfl_xor(nrm, 2));
      nrm = fmax(nrm, __shfl_xor(nrm, 16));
      nrm = fmax(nrm, __shfl_xor(nrm, 32));
    } else {
    // (MW: table k is built by wave k mod nwv; the barrier comes after the waves have fetched their control amplitudes)
    // The waves that build a table issue the loads of their control amplitudes FIRST (K <= 2, segments up to 48 slices: six
    // registers) -- both are the kernel's first, cold, global accesses and were serialised (4.0 + 1.0 us on wave 0 at cfg2)
    constexpr int PF_K = 2, PF_T = 3;
    const bool prefetch = MW && A.inline_tables && K <= PF_K && A.Lmax <= 16 * PF_T;
    double pre[PF_K][PF_T];
    if (prefetch) {
#pragma unroll
      for (int k = 0; k < PF_K; ++k)
#pragma unroll
        for (int i = 0; i < PF_T; ++i) {
          const int t = lp.idx16 + 16 * i;
          const double* sk = A.signals + ((long)sample * K + (k < K ? k : 0)) * A.N + n0;
          pre[k][i] = (k < K && valid && t < len) ? sk[t] : 0.0;
        }
    }
    if (A.inline_tables) {
      build_tables<D>(A, __builtin_amdgcn_readfirstlane(sample), tab, lane, wv, nwv);
    } else {
      const double* gt0 = A.tables + (long)(A.tab_per_sample ? sample : 0) * (1 + K) * (MAT + 4);
      for (int e = (MW ? (int)threadIdx.x : lane); e < (1 + K) * (MAT + 4); e += 64 * nwv) tab[e] = gt0[e];
    }
    if constexpr (!MW) __syncthreads();
    // segment-wide bound on ||X||_1 <= ||G0|| + sum_k max_t |c_k(t)| ||G_k||  -> one plan per segment
    // (MW: the amplitudes are fetched while wave 0 builds the tables, their maxima taken from LDS after the barrier)
    auto seg_max = [&](int k, bool from_lds) {
      const double* s = A.signals + ((long)sample * K + k) * A.N + n0;
      double cmax = 0.0;
      for (int t = lp.idx16; t < A.Lmax; t += 16) {
        double v;
        if (from_lds) {
          v = sg[lp.b * SG + k * A.Lmax + t];
        } else {
          v = (valid && t < len) ? s[t] : 0.0;
          sg[lp.b * SG + k * A.Lmax + t] = v;
        }
        cmax = fmax(cmax, fabs(v));
      }
      cmax = fmax(cmax, __shfl_xor(cmax, 1));
      cmax = fmax(cmax, __shfl_xor(cmax, 2));
      cmax = fmax(cmax, __shfl_xor(cmax, 16));
      cmax = fmax(cmax, __shfl_xor(cmax, 32));
      return cmax;
    };
    SD_TICK(tk8);
    if constexpr (MW) {
      if (prefetch) {
#pragma unroll
        for (int k = 0; k < PF_K; ++k)
#pragma unroll
          for (int i = 0; i < PF_T; ++i) {
            const int t = lp.idx16 + 16 * i;
            if (k < K && t < A.Lmax) sg[lp.b * SG + k * A.Lmax + t] = pre[k][i];
          }
      } else {
        for (int k = 0; k < K; ++k) (void)seg_max(k, false);
      }
      SD_TICK(tk9);
      __syncthreads();  // the tables of wave 0 are in place
      // a sample on the COMPLEX loop has its own split (the two waves of a SIMD finish together at a different ratio there): the
      // table flags are known only now, so its waves fetch their amplitudes a second time (~1 us of a 285 us kernel)
      if (A.seg_long_c > 0 && A.seg_long_c != A.seg_long) {
        bool re = (A.mode == C3P_MODE_UNITARY);
        for (int k = 0; k <= K; ++k) re = re && (tab[k * (MAT + 4) + MAT + 3] == 0.0);
        if (__builtin_amdgcn_readfirstlane((int)re) == 0) {
          wave_sync();
          split_segments(A.seg_long_c);
          for (int k = 0; k < K; ++k) (void)seg_max(k, false);
          wave_sync();
        }
      }
    }
    nrm = tab[MAT + 2];
    for (int k = 0; k < K; ++k) nrm = fma(seg_max(k, MW), tab[(k + 1) * (MAT + 4) + MAT + 2], nrm);
    }
    nrm = fmax(nrm, __shfl_xor(nrm, 4));
    nrm = fmax(nrm, __shfl_xor(nrm, 8));
    nrm = readfirstlane_f64(nrm);
    // every table of this sample purely imaginary (real Hamiltonians) -> real fast path
    bool realH = !XG && (A.mode == C3P_MODE_UNITARY);
    if constexpr (!XG)
      for (int k = 0; k <= K; ++k) realH = realH && (tab[k * (MAT + 4) + MAT + 3] == 0.0);
    realH = __builtin_amdgcn_readfirstlane((int)realH) != 0;
    // plan of the complex loop (not evaluated on the real path: its threshold loops cost ~0.5 us of a workgroup's prologue).
    // (round 6) Hermitian Hamiltonians -- every table flagged skew-Hermitian -- take the schemes for normal generators
    MfmaPlan plan = {0, 0, 0};
    int t18n = 0;
    if (!realH) {
      bool normalG = !XG && (A.mode == C3P_MODE_UNITARY) && !(A.no_t18n & 1);
      if constexpr (!XG)
        for (int k = 0; k <= K; ++k) normalG = normalG && (tab[k * (MAT + 4) + MAT + 3] <= 0.0);  // negative: complex skew-Hermitian; zero: real symmetric Hamiltonian
      t18n = __builtin_amdgcn_readfirstlane((int)normalG);
      plan = c3p_pick_plan_mfma(nrm, t18n ? C3P_T18N_THETA : C3P_T18_THETA, t18n != 0 && !(A.no_t18n & 2));
    }
    const int pr = __builtin_amdgcn_readfirstlane(plan.r);
    const int ps = __builtin_amdgcn_readfirstlane(plan.s);
    const int t18 = __builtin_amdgcn_readfirstlane(plan.t18);
    const double scale = ldexp(1.0, -ps);
    __syncthreads();
    SD_TICK(tk1);

    if (realH) {
      constexpr int NB = RD<D>::NB;
      typedef double RMat[NB][NB];
      // Round 6: the 7-product variant of the padded-tile loop below evaluates the economised degree-8 pair (theta = 1.85,
      // c3p_common.h) and serves every norm; -DC3P_SD_QUADSUM keeps round 5's Taylor variants (theta_16 = 0.816 / 1.13)
#ifndef C3P_SD_QUADSUM
#define C3P_SD_CA(j) c3p_mm8_cos[j]
#define C3P_SD_SA(j) c3p_mm8_sinc[j]
      constexpr double theta_real = C3P_MM8_THETA, theta_deg16 = C3P_MM8_THETA;
#else
#define C3P_SD_CA(j) (((j) & 1) ? -c3p_inv_fact[2 * (j)] : c3p_inv_fact[2 * (j)])
#define C3P_SD_SA(j) (((j) & 1) ? -c3p_inv_fact[2 * (j) + 1] : c3p_inv_fact[2 * (j) + 1])
      constexpr double theta_real = C3P_T18_THETA, theta_deg16 = 8.16e-1;
#endif
      int ps18 = 0;
      {
        double p = theta_real;
        while (p < nrm && ps18 < 40) {
          p *= 2.0;
          ++ps18;
        }
      }
      ps18 = __builtin_amdgcn_readfirstlane(ps18);
      const double rscale = ldexp(1.0, -ps18);
      const int swap_lane = 16 * lp.c + 4 * lp.b + lp.r;  // (r, c) <-> (c, r) inside the chain's block
      const int tail_lane = 16 * lp.c + 4 * lp.b;         // lane (c, 0): source of the rank-1 tail's row vector
      const int row0_lane = 4 * lp.b + lp.c;               // lane (0, c): row D-1 of a general right operand
      RMat Ur, Ui;  // the chain state in real blocks
      int yo[NB];        // Im rows of the half-image tables hold -Y
      double ymask[NB];
#pragma unroll
      for (int I = 0; I < NB; ++I) {
        const bool ok = 4 * I + lp.r < D;
        yo[I] = (2 * (ok ? 4 * I + lp.r : 0) + 1) * W + lp.c;
        ymask[I] = ok ? 1.0 : 0.0;
      }
      // theta_16 (unit roundoff 2^-52) = 0.816: below it the degree-16 / 17 polynomials are exact to roundoff and the
      // shallower 7-product evaluation is used.  The variant is chosen per segment OUTSIDE the slice loop (the loop is
      // instantiated twice) so that neither variant's registers burden the other's schedule.
      const bool deg16 = __builtin_amdgcn_readfirstlane((int)(nrm * rscale <= theta_deg16)) != 0;
      if constexpr (SPLIT) {
        // ---- core + border form (D = 4 NC + 1; see SMat / GMat above): the same polynomial evaluation and chain step ----
        static_assert(!SPLIT || (D % 4 == 1 && D > 4 && !DUS), "core + border form: D = 5, 9 without slice output");
        constexpr int NC = (D - 1) / 4;
        typedef SMat<NC> SM;
        GMat<NC> Gr, Gi;  // the chain state
        int so[NC], sro[NC], sco[NC];
#pragma unroll
        for (int I = 0; I < NC; ++I) {
          so[I] = (2 * (4 * I + lp.r) + 1) * W + lp.c;
          sro[I] = (2 * (4 * I + lp.r) + 1) * W + (D - 1);
          sco[I] = (2 * (4 * I + lp.c) + 1) * W + (D - 1);
        }
        constexpr int sso = (2 * (D - 1) + 1) * W + (D - 1);
#ifndef C3P_SD_QUADSUM
        // Round 6: the ECONOMISED polynomials of c3p_common.h (Y is real symmetric, so the polynomial error on [0, theta^2] IS the
        // matrix error).  Scaling against theta_8 = 1.85; a scaled norm below theta_6 = 0.83 (cfg2: 0.81) takes the degree-6 pair:
        // W, W^2, W^3, ONE paired Horner step in W^3, sin = (sin Y / Y) Y -- 6 products at dependency depth 5 (the degree-8 Taylor
        // pair needed W^4: 7); above it the degree-8 pair with W^4 (7 products where the degree-9 / 8 Taylor pair, theta 1.13, took 8).
        int ps_s = 0;
        {
          double p = C3P_MM8_THETA;
          while (p < nrm && ps_s < 40) {
            p *= 2.0;
            ++ps_s;
          }
        }
        ps_s = __builtin_amdgcn_readfirstlane(ps_s);
        const double rscale_s = ldexp(1.0, -ps_s);
        const bool small_var = __builtin_amdgcn_readfirstlane((int)(nrm * rscale_s <= C3P_MM6_THETA)) != 0;
#else
        const int ps_s = ps18;
        const double rscale_s = rscale;
        const bool small_var = deg16;
#endif
        auto split_loop = [&](auto deg16_tag) {
          constexpr bool DEG16 = decltype(deg16_tag)::value;  // (round-6 build: true = the degree-6 pair, false = the degree-8 pair)
          for (int t = 0; t < tmax; ++t) {
            const bool act = valid && t < len;
            const double sc = act ? rscale_s : 0.0;
            const double muw = act ? 1.0 : 0.0;
            double mu_r = muw * tab[MAT + 0], mu_i = muw * tab[MAT + 1];
            SM Y;
            {
              const double f = -sc;
#pragma unroll
              for (int I = 0; I < NC; ++I) {
#pragma unroll
                for (int J = 0; J < NC; ++J) Y.m[I][J] = f * lds_ld(tab + so[I] + J * 4);
                Y.vr[I] = f * lds_ld(tab + sro[I]);
                Y.vc[I] = f * lds_ld(tab + sco[I]);
              }
              Y.s = f * lds_ld(tab + sso);
            }
            for (int k = 0; k < K; ++k) {
              const double c0 = sg[lp.b * SG + k * A.Lmax + t];
              const double f = -sc * c0;
              const double* tk = tab + (k + 1) * (MAT + 4);
              mu_r = fma(c0, tk[MAT + 0], mu_r);
              mu_i = fma(c0, tk[MAT + 1], mu_i);
#pragma unroll
              for (int I = 0; I < NC; ++I) {
#pragma unroll
                for (int J = 0; J < NC; ++J) Y.m[I][J] = fma(f, lds_ld(tk + so[I] + J * 4), Y.m[I][J]);
                Y.vr[I] = fma(f, lds_ld(tk + sro[I]), Y.vr[I]);
                Y.vc[I] = fma(f, lds_ld(tk + sco[I]), Y.vc[I]);
              }
              Y.s = fma(f, lds_ld(tk + sso), Y.s);
            }
            SM W1, W2, W3, Cm, Sp, acc, acs;
            s8_zero(W1), s8_zero(W2), s8_zero(W3);
            mm_s8(Y, Y, W1);  // W = Y^2
            s8_finish(W1, swap_lane, tail_lane);
            mm_s8(W1, W1, W2);  // W^2
            s8_finish(W2, swap_lane, tail_lane);
#ifdef C3P_SD_QUADSUM
            if constexpr (DEG16) {
              SM W4;
              s8_zero(W4);
              mm_s8(W1, W2, W3);
              mm_s8(W2, W2, W4);
              s8_finish(W3, swap_lane, tail_lane);
              s8_finish(W4, swap_lane, tail_lane);
              s8_comb<NC, true>(acc, c3p_inv_fact[8], -c3p_inv_fact[10], c3p_inv_fact[12], -c3p_inv_fact[14], W1, W2, W3, lp);
              s8_comb<NC, true>(acs, c3p_inv_fact[9], -c3p_inv_fact[11], c3p_inv_fact[13], -c3p_inv_fact[15], W1, W2, W3, lp);
#pragma unroll
              for (int I = 0; I < NC; ++I) {
#pragma unroll
                for (int J = 0; J < NC; ++J) {
                  acc.m[I][J] = fma(c3p_inv_fact[16], W4.m[I][J], acc.m[I][J]);
                  acs.m[I][J] = fma(c3p_inv_fact[17], W4.m[I][J], acs.m[I][J]);
                }
                acc.vr[I] = fma(c3p_inv_fact[16], W4.vr[I], acc.vr[I]);
                acs.vr[I] = fma(c3p_inv_fact[17], W4.vr[I], acs.vr[I]);
                acc.vc[I] = fma(c3p_inv_fact[16], W4.vc[I], acc.vc[I]);
                acs.vc[I] = fma(c3p_inv_fact[17], W4.vc[I], acs.vc[I]);
              }
              acc.s = fma(c3p_inv_fact[16], W4.s, acc.s);
              acs.s = fma(c3p_inv_fact[17], W4.s, acs.s);
              s8_comb<NC, true, true>(Cm, 1.0, -c3p_inv_fact[2], c3p_inv_fact[4], -c3p_inv_fact[6], W1, W2, W3, lp);
              s8_comb<NC, true, true>(Sp, 1.0, -c3p_inv_fact[3], c3p_inv_fact[5], -c3p_inv_fact[7], W1, W2, W3, lp);
              mm_s8x2(W4, acc, Cm, acs, Sp);  // Cm = cos Y, Sp = sin(Y) / Y
            } else {
              mm_s8(W1, W2, W3);  // W^3
              s8_finish(W3, swap_lane, tail_lane);
              s8_comb<NC, true>(Cm, c3p_inv_fact[12], -c3p_inv_fact[14], c3p_inv_fact[16], -c3p_inv_fact[18], W1, W2, W3, lp);
              s8_comb<NC, false>(Sp, c3p_inv_fact[13], -c3p_inv_fact[15], c3p_inv_fact[17], 0.0, W1, W2, W3, lp);
              s8_comb<NC, false, true>(acc, -c3p_inv_fact[6], c3p_inv_fact[8], -c3p_inv_fact[10], 0.0, W1, W2, W3, lp);
              s8_comb<NC, false, true>(acs, -c3p_inv_fact[7], c3p_inv_fact[9], -c3p_inv_fact[11], 0.0, W1, W2, W3, lp);
              mm_s8x2(W3, Cm, acc, Sp, acs);
              s8_finish(acc, swap_lane, tail_lane);
              s8_finish(acs, swap_lane, tail_lane);
              s8_comb<NC, false, true>(Cm, 1.0, -c3p_inv_fact[2], c3p_inv_fact[4], 0.0, W1, W2, W3, lp);
              s8_comb<NC, false, true>(Sp, 1.0, -c3p_inv_fact[3], c3p_inv_fact[5], 0.0, W1, W2, W3, lp);
              mm_s8x2(W3, acc, Cm, acs, Sp);  // Cm = cos Y, Sp = sin(Y) / Y
            }
#else
            // The Horner factors are the LEFT operands of the paired product (everything commutes): they are combined on the 6
            // registers a product accumulates into (upper core tiles, r-form border, corner) and completed by ONE lane swap each;
            // the shared right operand (W^3 / W^4) is completed in full; a power that is never a product operand is not completed.
            if constexpr (DEG16) {
              // degree 6: cos = (c0 + c1 W + c2 W^2) + W^3 (c3 + c4 W + c5 W^2 + c6 W^3)
              mm_s8(W1, W2, W3);
              s8_finish(W3, swap_lane, tail_lane);
              s8_comb<NC, true, true>(acc, c3p_mm6_cos[3], c3p_mm6_cos[4], c3p_mm6_cos[5], c3p_mm6_cos[6], W1, W2, W3, lp);
              s8_comb<NC, true, true>(acs, c3p_mm6_sinc[3], c3p_mm6_sinc[4], c3p_mm6_sinc[5], c3p_mm6_sinc[6], W1, W2, W3, lp);
              s8_finish_lower(acc, swap_lane);
              s8_finish_lower(acs, swap_lane);
              s8_comb<NC, false, true>(Cm, c3p_mm6_cos[0], c3p_mm6_cos[1], c3p_mm6_cos[2], 0.0, W1, W2, W3, lp);
              s8_comb<NC, false, true>(Sp, c3p_mm6_sinc[0], c3p_mm6_sinc[1], c3p_mm6_sinc[2], 0.0, W1, W2, W3, lp);
              mm_s8x2r(acc, acs, W3, Cm, Sp);  // Cm = cos Y, Sp = sin(Y) / Y
            } else {
              // degree 8: cos = (c0 + c1 W + c2 W^2 + c3 W^3) + W^4 (c4 + c5 W + c6 W^2 + c7 W^3 + c8 W^4); W^3 is not a product operand
              SM W4;
              s8_zero(W4);
              mm_s8(W1, W2, W3);
              mm_s8(W2, W2, W4);
              s8_finish(W4, swap_lane, tail_lane);
              s8_comb<NC, true, true>(acc, c3p_mm8_cos[4], c3p_mm8_cos[5], c3p_mm8_cos[6], c3p_mm8_cos[7], W1, W2, W3, lp);
              s8_comb<NC, true, true>(acs, c3p_mm8_sinc[4], c3p_mm8_sinc[5], c3p_mm8_sinc[6], c3p_mm8_sinc[7], W1, W2, W3, lp);
#pragma unroll
              for (int I = 0; I < NC; ++I) {
#pragma unroll
                for (int J = I; J < NC; ++J) {
                  acc.m[I][J] = fma(c3p_mm8_cos[8], W4.m[I][J], acc.m[I][J]);
                  acs.m[I][J] = fma(c3p_mm8_sinc[8], W4.m[I][J], acs.m[I][J]);
                }
                acc.vr[I] = fma(c3p_mm8_cos[8], W4.vr[I], acc.vr[I]);
                acs.vr[I] = fma(c3p_mm8_sinc[8], W4.vr[I], acs.vr[I]);
              }
              acc.s = fma(c3p_mm8_cos[8], W4.s, acc.s);
              acs.s = fma(c3p_mm8_sinc[8], W4.s, acs.s);
              s8_finish_lower(acc, swap_lane);
              s8_finish_lower(acs, swap_lane);
              s8_comb<NC, true, true>(Cm, c3p_mm8_cos[0], c3p_mm8_cos[1], c3p_mm8_cos[2], c3p_mm8_cos[3], W1, W2, W3, lp);
              s8_comb<NC, true, true>(Sp, c3p_mm8_sinc[0], c3p_mm8_sinc[1], c3p_mm8_sinc[2], c3p_mm8_sinc[3], W1, W2, W3, lp);
              mm_s8x2r(acc, acs, W4, Cm, Sp);  // Cm = cos Y, Sp = sin(Y) / Y
            }
#endif
#ifdef C3P_SD_QUADSUM
            s8_finish(Cm, swap_lane, tail_lane);
            s8_finish(Sp, swap_lane, tail_lane);
            s8_zero(acc);
            mm_s8(Y, Sp, acc);  // acc = sin Y
            s8_finish(acc, swap_lane, tail_lane);
#else
            // cos Y, sin Y / Y and sin Y are LEFT operands from here on (sin Y = (sin Y / Y) Y, the chain step): lower tiles only;
            // their c-form borders are only fetched for the squarings and the first slice of a segment
            s8_finish_lower(Cm, swap_lane);
            s8_finish_lower(Sp, swap_lane);
            s8_zero(acc);
            mm_s8(Sp, Y, acc);  // acc = sin Y
            s8_finish_lower(acc, swap_lane);
            if (ps_s > 0 || t == 0) {
              s8_finish_vc(Cm, tail_lane);
              s8_finish_vc(acc, tail_lane);
            }
#endif
            // squarings: cos 2Y = (C - S)(C + S), sin 2Y = 2 S C
            for (int it = 0; it < ps_s; ++it) {
              SM Dm, Sm, C2, SC;
#pragma unroll
              for (int I = 0; I < NC; ++I) {
#pragma unroll
                for (int J = 0; J < NC; ++J) {
                  Dm.m[I][J] = Cm.m[I][J] - acc.m[I][J];
                  Sm.m[I][J] = Cm.m[I][J] + acc.m[I][J];
                }
                Dm.vr[I] = Cm.vr[I] - acc.vr[I];
                Sm.vr[I] = Cm.vr[I] + acc.vr[I];
                Dm.vc[I] = Cm.vc[I] - acc.vc[I];
                Sm.vc[I] = Cm.vc[I] + acc.vc[I];
              }
              Dm.s = Cm.s - acc.s;
              Sm.s = Cm.s + acc.s;
              s8_zero(C2), s8_zero(SC);
              mm_s8(Dm, Sm, C2);
              mm_s8(acc, Cm, SC);
#pragma unroll
              for (int I = 0; I < NC; ++I) {
#pragma unroll
                for (int J = I; J < NC; ++J) {
                  Cm.m[I][J] = C2.m[I][J];
                  acc.m[I][J] = 2.0 * SC.m[I][J];
                }
                Cm.vr[I] = C2.vr[I];
                acc.vr[I] = 2.0 * SC.vr[I];
              }
              Cm.s = C2.s;
              acc.s = 2.0 * SC.s;
              s8_finish(Cm, swap_lane, tail_lane);
              s8_finish(acc, swap_lane, tail_lane);
            }
            if (t == 0) {  // U <- E = C - i S
#pragma unroll
              for (int I = 0; I < NC; ++I) {
#pragma unroll
                for (int J = 0; J < NC; ++J) {
                  Gr.m[I][J] = Cm.m[I][J];
                  Gi.m[I][J] = -acc.m[I][J];
                }
                Gr.cr[I] = Cm.vr[I], Gr.cc[I] = Cm.vc[I], Gr.rc[I] = Cm.vc[I];
                Gi.cr[I] = -acc.vr[I], Gi.cc[I] = -acc.vc[I], Gi.rc[I] = -acc.vc[I];
              }
              Gr.s = Cm.s;
              Gi.s = -acc.s;
              mus_r = mu_r;
              mus_i = c3p_phase_add(0.0, mu_i);
            } else {
#if defined(C3P_SD_ABL) && (C3P_SD_ABL & 2)
              // TIMING-ONLY (wrong results): no chain product
#pragma unroll
              for (int I = 0; I < NC; ++I) {
#pragma unroll
                for (int J = 0; J < NC; ++J) Gr.m[I][J] += Cm.m[I][J], Gi.m[I][J] -= acc.m[I][J];
                Gr.cr[I] += Cm.vr[I], Gr.cc[I] += Cm.vc[I], Gr.rc[I] += Cm.vc[I];
                Gi.cr[I] -= acc.vr[I], Gi.cc[I] -= acc.vc[I], Gi.rc[I] -= acc.vc[I];
              }
              Gr.s += Cm.s, Gi.s -= acc.s;
#else
              chain_step8(Cm, acc, Gr, Gi, lp, tail_lane, row0_lane);
#endif
              mus_r += mu_r;
              mus_i = c3p_phase_add(mus_i, mu_i);
            }
          }
        };
        if (small_var)
          split_loop(std::true_type{});
        else
          split_loop(std::false_type{});
        // back to the complex half-image layout of the epilogue (once per segment, through the chain's image)
        wave_sync();
        for (int e = lp.idx16; e < 4 * NBI * W; e += 16) img[lp.b * IMG + e] = 0.0;
        wave_sync();
#pragma unroll
        for (int I = 0; I < NC; ++I) {
          const int i = 4 * I + lp.r;
#pragma unroll
          for (int J = 0; J < NC; ++J) {
            img[lp.b * IMG + (2 * i) * W + 4 * J + lp.c] = Gr.m[I][J];
            img[lp.b * IMG + (2 * i + 1) * W + 4 * J + lp.c] = Gi.m[I][J];
          }
          if (lp.c == 0) {
            img[lp.b * IMG + (2 * i) * W + D - 1] = Gr.cr[I];
            img[lp.b * IMG + (2 * i + 1) * W + D - 1] = Gi.cr[I];
          }
          if (lp.r == 0) {
            img[lp.b * IMG + (2 * (D - 1)) * W + 4 * I + lp.c] = Gr.rc[I];
            img[lp.b * IMG + (2 * (D - 1) + 1) * W + 4 * I + lp.c] = Gi.rc[I];
          }
        }
        if (lp.idx16 == 0) {
          img[lp.b * IMG + (2 * (D - 1)) * W + D - 1] = Gr.s;
          img[lp.b * IMG + (2 * (D - 1) + 1) * W + D - 1] = Gi.s;
        }
        wave_sync();
#pragma unroll
        for (int I = 0; I < NBI; ++I)
#pragma unroll
          for (int J = 0; J < NJ; ++J) U[I][J] = img[woff + I * 4 * W + J * 4];
        wave_sync();
      } else {
      auto real_loop = [&](auto deg16_tag) {
      constexpr bool DEG16 = decltype(deg16_tag)::value;
      for (int t = 0; t < tmax; ++t) {
        // (Two waves share a SIMD and the arbiter serves the OLDER one first: wave w finishes its segment at ~64 % of the
        // kernel time and wave w + 4 then runs alone -- wall_clock64 probes, -DC3P_SD_TIMING.  Alternating s_setprio per
        // slice makes them finish together and changes nothing in the total: measured, not kept.)
        const bool act = valid && t < len;
        const double sc = act ? rscale : 0.0;
        const double muw = act ? 1.0 : 0.0;
        double mu_r = muw * tab[MAT + 0], mu_i = muw * tab[MAT + 1];
        // ---- Y = scale dt (H - tr H / D) = -Im(G0 + sum_k c_k G_k) ----
        // (rows past the table are clamped to a valid row and masked: no exec-mask branches around LDS loads)
        RMat Y;
#pragma unroll
        for (int I = 0; I < NB; ++I) {
          const double f = -sc * ymask[I];
#pragma unroll
          for (int J = 0; J < NB; ++J) Y[I][J] = f * lds_ld(tab + yo[I] + J * 4);
        }
        for (int k = 0; k < K; ++k) {
          const double c0 = sg[lp.b * SG + k * A.Lmax + t];
          const double ck = sc * c0;
          const double* tk = tab + (k + 1) * (MAT + 4);
          mu_r = fma(c0, tk[MAT + 0], mu_r);
          mu_i = fma(c0, tk[MAT + 1], mu_i);
#pragma unroll
          for (int I = 0; I < NB; ++I) {
            const double f = -ck * ymask[I];
#pragma unroll
            for (int J = 0; J < NB; ++J) Y[I][J] = fma(f, lds_ld(tk + yo[I] + J * 4), Y[I][J]);
          }
        }
        RMat W1, W2, W3, Cm, Sp, acc, acs;
#pragma unroll
        for (int I = 0; I < NB; ++I)
#pragma unroll
          for (int J = 0; J < NB; ++J) W1[I][J] = W2[I][J] = W3[I][J] = 0.0;
        mm_sym<D>(Y, Y, W1, tail_lane);  // W = Y^2
        sym_fill<D>(W1, swap_lane);
        mm_sym<D>(W1, W1, W2, tail_lane);  // W^2
        sym_fill<D>(W2, swap_lane);
        if constexpr (DEG16) {
          // cos to W^8, sin/Y to W^8 (Taylor degree 16 / 17), Paterson-Stockmeyer with q = 4: W^3 = W W^2 and
          // W^4 = W^2 W^2 are independent, then ONE paired Horner step  P = B0 + W^4 (B1 + c8 W^4):
          // 7 products, dependency depth 5 (Y^2, W^2, {W^3, W^4}, {cos, sin/Y}, sin) instead of 8 and 7.
          RMat W4;
#pragma unroll
          for (int I = 0; I < NB; ++I)
#pragma unroll
            for (int J = 0; J < NB; ++J) W4[I][J] = 0.0;
          mm_sym<D>(W1, W2, W3, tail_lane);
          mm_sym<D>(W2, W2, W4, tail_lane);
          sym_fill<D>(W3, swap_lane);
          sym_fill<D>(W4, swap_lane);
          rcomb<D, true>(acc, C3P_SD_CA(4), C3P_SD_CA(5), C3P_SD_CA(6), C3P_SD_CA(7), W1, W2, W3, lp);
          rcomb<D, true>(acs, C3P_SD_SA(4), C3P_SD_SA(5), C3P_SD_SA(6), C3P_SD_SA(7), W1, W2, W3, lp);
#pragma unroll
          for (int I = 0; I < NB; ++I)
#pragma unroll
            for (int J = sym_j0<D>(I); J < NB; ++J) {
              acc[I][J] = fma(C3P_SD_CA(8), W4[I][J], acc[I][J]);
              acs[I][J] = fma(C3P_SD_SA(8), W4[I][J], acs[I][J]);
            }
          rcomb<D, true, true>(Cm, C3P_SD_CA(0), C3P_SD_CA(1), C3P_SD_CA(2), C3P_SD_CA(3), W1, W2, W3, lp);
          rcomb<D, true, true>(Sp, C3P_SD_SA(0), C3P_SD_SA(1), C3P_SD_SA(2), C3P_SD_SA(3), W1, W2, W3, lp);
          mm_sym2<D>(W4, acc, Cm, acs, Sp, tail_lane);  // Cm = cos Y, Sp = sin(Y) / Y
        } else {
        mm_sym<D>(W1, W2, W3, tail_lane);  // W^3
        sym_fill<D>(W3, swap_lane);
        // cos: c_j = (-1)^j / (2j)!;  sin / Y: s_j = (-1)^j / (2j+1)!;  both by Horner in W^3, interleaved
        rcomb<D, true>(Cm, c3p_inv_fact[12], -c3p_inv_fact[14], c3p_inv_fact[16], -c3p_inv_fact[18], W1, W2, W3, lp);
        rcomb<D, false>(Sp, c3p_inv_fact[13], -c3p_inv_fact[15], c3p_inv_fact[17], 0.0, W1, W2, W3, lp);
        rcomb<D, false, true>(acc, -c3p_inv_fact[6], c3p_inv_fact[8], -c3p_inv_fact[10], 0.0, W1, W2, W3, lp);
        rcomb<D, false, true>(acs, -c3p_inv_fact[7], c3p_inv_fact[9], -c3p_inv_fact[11], 0.0, W1, W2, W3, lp);
        mm_sym2<D>(W3, Cm, acc, Sp, acs, tail_lane);
        sym_fill<D>(acc, swap_lane);
        sym_fill<D>(acs, swap_lane);
        rcomb<D, false, true>(Cm, 1.0, -c3p_inv_fact[2], c3p_inv_fact[4], 0.0, W1, W2, W3, lp);
        rcomb<D, false, true>(Sp, 1.0, -c3p_inv_fact[3], c3p_inv_fact[5], 0.0, W1, W2, W3, lp);
        mm_sym2<D>(W3, acc, Cm, acs, Sp, tail_lane);  // Cm = cos Y, Sp = sin(Y) / Y
        }
        sym_fill<D>(Cm, swap_lane);
        sym_fill<D>(Sp, swap_lane);
#pragma unroll
        for (int I = 0; I < NB; ++I)
#pragma unroll
          for (int J = 0; J < NB; ++J) acc[I][J] = 0.0;
        mm_sym<D>(Y, Sp, acc, tail_lane);  // acc = sin Y
        sym_fill<D>(acc, swap_lane);
        // ---- E = cos Y - i sin Y (Cm, acc) ; squarings in real form.  C and S are polynomials in Y and commute:
        //      cos 2Y = C^2 - S^2 = (C - S)(C + S) (ONE product, symmetric result), sin 2Y = 2 S C ----
        for (int it = 0; it < ps18; ++it) {
          RMat Dm, Sm, C2, SC;
#pragma unroll
          for (int I = 0; I < NB; ++I)
#pragma unroll
            for (int J = sym_j0<D>(I); J < NB; ++J) {
              Dm[I][J] = Cm[I][J] - acc[I][J];
              Sm[I][J] = Cm[I][J] + acc[I][J];
              C2[I][J] = SC[I][J] = 0.0;
            }
          mm_sym<D>(Dm, Sm, C2, tail_lane);
          mm_sym<D>(acc, Cm, SC, tail_lane);
#pragma unroll
          for (int I = 0; I < NB; ++I)
#pragma unroll
            for (int J = I; J < NB; ++J) {
              Cm[I][J] = C2[I][J];
              acc[I][J] = 2.0 * SC[I][J];
            }
          sym_fill<D>(Cm, swap_lane);
          sym_fill<D>(acc, swap_lane);
        }
        if (DUS || t == 0) {  // the whole matrices leave the symmetric stage: dU output / U <- E
          sym_fill_rest<D>(Cm, swap_lane);
          sym_fill_rest<D>(acc, swap_lane);
        }
        if constexpr (DUS) {
          // dU = e^{mu} (C - iS)
          double sn, cs;
          sincos(mu_i, &sn, &cs);
          const double er = exp(mu_r);
          const double pr = er * cs, pi = er * sn;
          double* dst = reinterpret_cast<double*>(A.dUs_out) + ((long)sample * A.N + n0 + (act ? t : 0)) * D * D * 2;
#pragma unroll
          for (int I = 0; I < NB; ++I)
#pragma unroll
            for (int J = 0; J < NB; ++J) {
              const int i = 4 * I + lp.r, j = 4 * J + lp.c;
              if (act && i < D && j < D) {
                dst[(i * D + j) * 2 + 0] = fma(pr, Cm[I][J], pi * acc[I][J]);
                dst[(i * D + j) * 2 + 1] = fma(pi, Cm[I][J], -pr * acc[I][J]);
              }
            }
        }
        // ---- chain in real blocks (no LDS): Ur' = C Ur + S Ui,  Ui' = C Ui - S Ur ----
        if (t == 0) {
#pragma unroll
          for (int I = 0; I < NB; ++I)
#pragma unroll
            for (int J = 0; J < NB; ++J) {
              Ur[I][J] = Cm[I][J];
              Ui[I][J] = -acc[I][J];
            }
          mus_r = mu_r;
          mus_i = c3p_phase_add(0.0, mu_i);
        } else {
          // (C - iS)(Ur + i Ui) with three real products: T1 = C Ur, T2 = S Ui, T3 = (C - S)(Ur + Ui);
          // Re = T1 + T2, Im = T3 - T1 + T2  (C - S is symmetric: its A fragments are registers, as for C and S)
          RMat Dm, Us, T1, T2, T3;
#pragma unroll
          for (int I = 0; I < NB; ++I)
#pragma unroll
            for (int J = 0; J < NB; ++J) {
              if (J >= sym_j0<D>(I)) Dm[I][J] = Cm[I][J] - acc[I][J];
              Us[I][J] = Ur[I][J] + Ui[I][J];
              T1[I][J] = T2[I][J] = T3[I][J] = 0.0;
            }
          mm_symA<D>(Cm, Ur, T1, row0_lane, 1.0);
          mm_symA<D>(acc, Ui, T2, row0_lane, 1.0);
          mm_symA<D>(Dm, Us, T3, row0_lane, 1.0);
#pragma unroll
          for (int I = 0; I < NB; ++I)
#pragma unroll
            for (int J = 0; J < NB; ++J) {
              Ur[I][J] = T1[I][J] + T2[I][J];
              Ui[I][J] = (T3[I][J] - T1[I][J]) + T2[I][J];
            }
          mus_r += mu_r;
          mus_i = c3p_phase_add(mus_i, mu_i);
        }
      }
      };
      if (deg16)
        real_loop(std::true_type{});
      else
        real_loop(std::false_type{});
      // back to the complex half-image layout of the epilogue (once per segment, through the chain's image)
      wave_sync();
#pragma unroll
      for (int I = 0; I < NB; ++I)
#pragma unroll
        for (int J = 0; J < NB; ++J) {
          const int i = 4 * I + lp.r, j = 4 * J + lp.c;
          if (2 * i + 1 < 4 * NBI) {
            img[lp.b * IMG + (2 * i) * W + j] = Ur[I][J];
            img[lp.b * IMG + (2 * i + 1) * W + j] = Ui[I][J];
          }
        }
      wave_sync();
#pragma unroll
      for (int I = 0; I < NBI; ++I)
#pragma unroll
        for (int J = 0; J < NJ; ++J) U[I][J] = img[woff + I * 4 * W + J * 4];
      wave_sync();
      }  // !SPLIT
    } else {
    // the slice loop is instantiated per plan (T18 / Paterson-Stockmeyer) with the branch outside, as on the real path
    // (the 20 parameters as wave-uniform values loaded ONCE: read through the table inside the slice loop they were re-fetched per slice)
    const double* tc = c3p_t18_tab[t18n];
    const double t18_a11 = tc[C3P_I_A11];
    const double t18_a21 = tc[C3P_I_A21];
    const double t18_a31 = tc[C3P_I_A31];
    const double t18_b02 = tc[C3P_I_B02];
    const double t18_b03 = tc[C3P_I_B03];
    const double t18_b11 = tc[C3P_I_B11];
    const double t18_b12 = tc[C3P_I_B12];
    const double t18_b13 = tc[C3P_I_B13];
    const double t18_b21 = tc[C3P_I_B21];
    const double t18_b22 = tc[C3P_I_B22];
    const double t18_b23 = tc[C3P_I_B23];
    const double t18_b24 = tc[C3P_I_B24];
    const double t18_b31 = tc[C3P_I_B31];
    const double t18_b32 = tc[C3P_I_B32];
    const double t18_b33 = tc[C3P_I_B33];
    const double t18_b34 = tc[C3P_I_B34];
    const double t18_b61 = tc[C3P_I_B61];
    const double t18_b62 = tc[C3P_I_B62];
    const double t18_b63 = tc[C3P_I_B63];
    const double t18_b64 = tc[C3P_I_B64];
    auto complex_loop = [&](auto t18_tag) {
    constexpr int VARIANT = decltype(t18_tag)::value;  // 0 Paterson-Stockmeyer, 1 T18, 2 the four-product scheme (normal generators)
    constexpr bool T18 = VARIANT == 1;
    for (int t = 0; t < tmax; ++t) {
      const bool act = valid && t < len;
      // ---- X = scale (G0 + sum_k c_k G_k) in D-layout; trace shift mu ----
      // chains past their segment end (lengths differ by at most one slice) take X = 0, E = I
      const double sc = act ? scale : 0.0;
      const double muw = act ? 1.0 : 0.0;
      double mu_r, mu_i;
      double X[NBI][NJ];
      if constexpr (XG) {
        const long m = (long)sample * A.N + n0 + (act ? t : 0);
        mu_r = muw * A.meta[m * 4 + 0];
        mu_i = muw * A.meta[m * 4 + 1];
        const double2* src = reinterpret_cast<const double2*>(A.hs) + (long)sample * A.hs_bstride + (long)(n0 + (act ? t : 0)) * D * D;
        const double mine = (lp.r & 1) ? mu_i : mu_r;
#pragma unroll
        for (int I = 0; I < NBI; ++I)
#pragma unroll
          for (int J = 0; J < NJ; ++J) {
            const int row = 2 * I + (lp.r >> 1), col = 4 * J + lp.c;
            const bool in = row < D && col < D;
            const double2 h = src[in ? row * D + col : 0];
            double v = (lp.r & 1) ? fma(A.coef_r, h.y, A.coef_i * h.x) : fma(A.coef_r, h.x, -A.coef_i * h.y);
            v -= (row == col) ? mine : 0.0;
            X[I][J] = in ? sc * v : 0.0;
          }
      } else {
      mu_r = muw * tab[MAT + 0];
      mu_i = muw * tab[MAT + 1];
#pragma unroll
      for (int I = 0; I < NBI; ++I)
#pragma unroll
        for (int J = 0; J < NJ; ++J) X[I][J] = sc * tab[toff + I * 4 * W + J * 4];
      for (int k = 0; k < K; ++k) {
        const double c0 = sg[lp.b * SG + k * A.Lmax + t];
        const double ck = sc * c0;
        const double* tk = tab + (k + 1) * (MAT + 4);
        mu_r = fma(c0, tk[MAT + 0], mu_r);  // c0 = 0 for inactive slices (sg is zero padded)
        mu_i = fma(c0, tk[MAT + 1], mu_i);
#pragma unroll
        for (int I = 0; I < NBI; ++I)
#pragma unroll
          for (int J = 0; J < NJ; ++J) X[I][J] = fma(ck, tk[toff + I * 4 * W + J * 4], X[I][J]);
      }
      }
      write_image<D>(X, img, woff);
      // ---- powers with the left operand X (read from its image) ----
      double A2[NBI][NJ], A3[NBI][NJ], P[NBI][NJ];
#pragma unroll
      for (int I = 0; I < NBI; ++I)
#pragma unroll
        for (int J = 0; J < NJ; ++J) A2[I][J] = A3[I][J] = 0.0;
      mm_img<D>(img, roff, negmask, X, A2);
      if constexpr (VARIANT == 2) {
        // ---- four products (c3p_common.h, c3p_e4n): A2 above; y0 = A2 (e0 A2 + e1 X);
        //      y1 = (y0 + e2 A2 + e3 X)(y0 + e4 A2) + e5 y0 + e6 A2;
        //      P = (y1 + e7 A2 + e8 X)(y1 + e9 y0 + e10 X) + e11 y1 + e12 y0 + e13 A2 + e14 X + e15 I ----
        // (A3 holds y0)
        // (element loops with exactly the terms of each combination: 15 vector operations per element and slice)
        double y1[NBI][NJ];
        {
          double R[NBI][NJ];
#pragma unroll
          for (int I = 0; I < NBI; ++I)
#pragma unroll
            for (int J = 0; J < NJ; ++J) R[I][J] = fma(c3p_e4n[0], A2[I][J], c3p_e4n[1] * X[I][J]);
          write_image<D>(A2, img, woff);
          mm_img<D>(img, roff, negmask, R, A3);
        }
        {
          double L[NBI][NJ], R[NBI][NJ];
#pragma unroll
          for (int I = 0; I < NBI; ++I)
#pragma unroll
            for (int J = 0; J < NJ; ++J) L[I][J] = fma(c3p_e4n[2], A2[I][J], fma(c3p_e4n[3], X[I][J], A3[I][J]));
          write_image<D>(L, img, woff);
#pragma unroll
          for (int I = 0; I < NBI; ++I)
#pragma unroll
            for (int J = 0; J < NJ; ++J) {
              R[I][J] = fma(c3p_e4n[4], A2[I][J], A3[I][J]);
              y1[I][J] = fma(c3p_e4n[5], A3[I][J], c3p_e4n[6] * A2[I][J]);
            }
          mm_img<D>(img, roff, negmask, R, y1);
        }
        {
          double L[NBI][NJ], R[NBI][NJ];
#pragma unroll
          for (int I = 0; I < NBI; ++I)
#pragma unroll
            for (int J = 0; J < NJ; ++J) L[I][J] = fma(c3p_e4n[7], A2[I][J], fma(c3p_e4n[8], X[I][J], y1[I][J]));
          write_image<D>(L, img, woff);
#pragma unroll
          for (int I = 0; I < NBI; ++I)
#pragma unroll
            for (int J = 0; J < NJ; ++J) {
              R[I][J] = fma(c3p_e4n[9], A3[I][J], fma(c3p_e4n[10], X[I][J], y1[I][J]));
              double v = fma(c3p_e4n[11], y1[I][J], c3p_e4n[12] * A3[I][J]);
              v = fma(c3p_e4n[13], A2[I][J], v);
              v = fma(c3p_e4n[14], X[I][J], v);
              if (2 * I - 4 * J >= -1 && 2 * I - 4 * J <= 3) {  // the real diagonal of the half image: + e15 I  (as lincomb6)
                const bool on = (ddelta == 4 * J - 2 * I) && (2 * I + rhalf < D);
                v += on ? c3p_e4n[15] : 0.0;
              }
              P[I][J] = v;
            }
          mm_img<D>(img, roff, negmask, R, P);
        }
      } else {
      mm_img<D>(img, roff, negmask, A2, A3);
      if constexpr (T18) {
        // ---- T18 (Bader-Blanes-Casas): 5 products in total ----
        double A6[NBI][NJ];
#pragma unroll
        for (int I = 0; I < NBI; ++I)
#pragma unroll
          for (int J = 0; J < NJ; ++J) A6[I][J] = 0.0;
        write_image<D>(A3, img, woff);
        mm_img<D>(img, roff, negmask, A3, A6);
        double acc[NBI][NJ];
        {
          double B1[NBI][NJ], B5[NBI][NJ];
          lincomb6<D>(B1, 0.0, t18_a11, t18_a21, t18_a31, 0.0, X, A2, A3, A6, ddelta, rhalf);
          write_image<D>(B1, img, woff);
          lincomb6<D>(B5, 0.0, 0.0, t18_b24, t18_b34, t18_b64, X, A2, A3, A6, ddelta, rhalf);
          lincomb6<D>(acc, t18_b03, t18_b13, t18_b23, t18_b33, t18_b63, X, A2, A3, A6, ddelta, rhalf);
          mm_img<D>(img, roff, negmask, B5, acc);  // acc = A9 = B1 B5 + B4
        }
        {
          double L[NBI][NJ];
          lincomb6<D>(L, t18_b02, t18_b12, t18_b22, t18_b32, t18_b62, X, A2, A3, A6, ddelta, rhalf);
#pragma unroll
          for (int I = 0; I < NBI; ++I)
#pragma unroll
            for (int J = 0; J < NJ; ++J) L[I][J] += acc[I][J];
          write_image<D>(L, img, woff);
        }
        lincomb6<D>(P, 0.0, t18_b11, t18_b21, t18_b31, t18_b61, X, A2, A3, A6, ddelta, rhalf);
        mm_img<D>(img, roff, negmask, acc, P);  // P = B2 + (B3 + A9) A9
      } else {
        {
          double A4[NBI][NJ];
  #pragma unroll
          for (int I = 0; I < NBI; ++I)
  #pragma unroll
            for (int J = 0; J < NJ; ++J) A4[I][J] = 0.0;
          mm_img<D>(img, roff, negmask, A3, A4);
          // ---- Horner in X^4: P = c_m X^4 + B_{r-1};  P = X^4 P + B_j ----
          const int j = pr - 1;
          poly_block<D>(P, c3p_inv_fact[4 * j], c3p_inv_fact[4 * j + 1], c3p_inv_fact[4 * j + 2],
                        c3p_inv_fact[4 * j + 3], X, A2, A3, ddelta, rhalf);
          const double cm = c3p_inv_fact[4 * pr];
  #pragma unroll
          for (int I = 0; I < NBI; ++I)
  #pragma unroll
            for (int J = 0; J < NJ; ++J) P[I][J] = fma(cm, A4[I][J], P[I][J]);
          if (pr > 1) write_image<D>(A4, img, woff);
        }
        for (int j = pr - 2; j >= 0; --j) {
          double acc[NBI][NJ];
          poly_block<D>(acc, c3p_inv_fact[4 * j], c3p_inv_fact[4 * j + 1], c3p_inv_fact[4 * j + 2],
                        c3p_inv_fact[4 * j + 3], X, A2, A3, ddelta, rhalf);
          mm_img<D>(img, roff, negmask, P, acc);
  #pragma unroll
          for (int I = 0; I < NBI; ++I)
  #pragma unroll
            for (int J = 0; J < NJ; ++J) P[I][J] = acc[I][J];
        }
      }
      }  // VARIANT != 2
      // ---- squarings ----
      for (int it = 0; it < ps; ++it) {
        write_image<D>(P, img, woff);
        double acc[NBI][NJ];
#pragma unroll
        for (int I = 0; I < NBI; ++I)
#pragma unroll
          for (int J = 0; J < NJ; ++J) acc[I][J] = 0.0;
        mm_img<D>(img, roff, negmask, P, acc);
#pragma unroll
        for (int I = 0; I < NBI; ++I)
#pragma unroll
          for (int J = 0; J < NJ; ++J) P[I][J] = acc[I][J];
      }
      // ---- partial propagator write-out ----
      if constexpr (DUS) {
        double sn, cs;
        sincos(mu_i, &sn, &cs);
        const double er = exp(mu_r);
        double* dst = reinterpret_cast<double*>(A.dUs_out) + ((long)sample * A.N + n0 + (act ? t : 0)) * D * D * 2;
        store_plain<D>(P, dst, er * cs, er * sn, nullptr, lp, act);
      }
      // ---- chain ----
      if (t == 0) {
#pragma unroll
        for (int I = 0; I < NBI; ++I)
#pragma unroll
          for (int J = 0; J < NJ; ++J) U[I][J] = P[I][J];
        mus_r = mu_r;
        mus_i = c3p_phase_add(0.0, mu_i);
      } else {
        write_image<D>(P, img, woff);
        double acc[NBI][NJ];
#pragma unroll
        for (int I = 0; I < NBI; ++I)
#pragma unroll
          for (int J = 0; J < NJ; ++J) acc[I][J] = 0.0;
        mm_img<D>(img, roff, negmask, U, acc);
#pragma unroll
        for (int I = 0; I < NBI; ++I)
#pragma unroll
          for (int J = 0; J < NJ; ++J) U[I][J] = acc[I][J];
        mus_r += mu_r;
        mus_i = c3p_phase_add(mus_i, mu_i);
      }
    }
    };
    if (t18 == 2)
      complex_loop(std::integral_constant<int, 2>{});
    else if (t18)
      complex_loop(std::integral_constant<int, 1>{});
    else
      complex_loop(std::integral_constant<int, 0>{});
    }  // complex path
  }
  // ---- segment result ----
  SD_TICK(tk2);
#ifdef C3P_SD_TIMING
  if (MW && blockIdx.x == 7 && lane == 0) printf("sd wave %d: loop ends at %lld (100 MHz ticks after the wave's start), prologue %lld, slices %d\n", wv, tk2 - tk0, tk1 - tk0, tmax);
#endif
  if constexpr (!GIVEN) {
    if (A.fuse) {
      // (1) fold the wave's four consecutive segments: W = U3 U2 U1 U0 (later segment on the left)
      double tr = mus_r, ti = mus_i;
      tr += __shfl_xor(tr, 4);
      tr += __shfl_xor(tr, 8);
      ti = c3p_phase_add(ti, __shfl_xor(ti, 4));
      ti = c3p_phase_add(ti, __shfl_xor(ti, 8));
      const int rrest = roff - lp.b * IMG;
      const int roff1 = ((lp.b + 1) & 3) * IMG + rrest;  // A fragments of the NEXT chain's image
      const int roff2 = ((lp.b + 2) & 3) * IMG + rrest;
      double V[NBI][NJ], Wt[NBI][NJ];
#pragma unroll
      for (int I = 0; I < NBI; ++I)
#pragma unroll
        for (int J = 0; J < NJ; ++J) V[I][J] = Wt[I][J] = 0.0;
      write_image<D>(U, img, woff);
      mm_img<D>(img, roff1, negmask, U, V);  // slots 0, 2: U_{b+1} U_b
      write_image<D>(V, img, woff);
      mm_img<D>(img, roff2, negmask, V, Wt);  // slot 0: (U3 U2)(U1 U0)
      const int nW = A.S >> 2;
      const int wq = seg >> 2;
      double sn, cs;
      sincos(ti, &sn, &cs);
      // (Hermitian Hamiltonians: the trace shifts are imaginary, tr = 0 exactly -- skip the double-precision exp of the fold)
      double er = 1.0;
      if (__ballot(tr != 0.0) != 0) er = exp(tr);
      const double* ph = A.fr_phase ? A.fr_phase + (long)sample * D : nullptr;
      if (nW == 1) {
        double* dst = reinterpret_cast<double*>(A.final_out) + (long)sample * D * D * 2;
        store_plain<D>(Wt, dst, er * cs, er * sn, ph, lp, valid && lp.b == 0);
        return;
      }
      const int lo = (lp.b * nW) >> 2, hi = ((lp.b + 1) * nW) >> 2;
      const int cnt = hi - lo;
      const int cmax = (nW + 3) >> 2;
      const double* base;
      double phc[NBI], phs[NBI];
      if constexpr (MW) {
        // (2) the waves of the sample are the waves of this workgroup: partials through LDS, wave 0 folds them
        store_plain<D>(Wt, part + (long)wq * D * D * 2, er * cs, er * sn, nullptr, lp, lp.b == 0);
        SD_TICK(tk3);
        // (wave 0 -- the older wave of its SIMD, through its loop first -- evaluates the row phases of the final store
        // while it waits for the others: five double-precision sincos, 1.5 us otherwise spent after the barrier)
        if (wv == 0) row_phase_factors<D>(ph, lp, phc, phs);
        __syncthreads();
        if (wv != 0) return;
        SD_TICK(tk5);
        base = part + (long)lo * D * D * 2;
      } else {
        {
          double* dst = reinterpret_cast<double*>(A.seg_out) + ((long)sample * nW + wq) * D * D * 2;
          store_plain<D, true>(Wt, dst, er * cs, er * sn, nullptr, lp, valid && lp.b == 0);
        }
        // (2) publish and arrive (cdna guide G16, form R1: write-through payload, drain, relaxed ticket)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        int old = 0;
        if (lane == 0)
          old = __hip_atomic_fetch_add(A.counters + sample, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        old = __builtin_amdgcn_readfirstlane(old);
        if (old != nW - 1) return;
        if (lane == 0) A.counters[sample] = 0;  // self-resetting: the next launch finds it zero
        // (3) last arriver of this sample: fold the nW partials (slot b takes a contiguous quarter)
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        base = reinterpret_cast<const double*>(A.seg_out) + ((long)sample * nW + lo) * D * D * 2;
      }
      // the next partial of the slot is always in flight while the current product runs, and the first two are
      // fetched together: ONE cross-CU memory round trip for S <= 32
      double Pn[NBI][NJ];
      auto fetch = [&](double (&P)[NBI][NJ], int t) {
        const bool act = t < cnt;
        const double* src = base + (long)(act ? t : 0) * D * D * 2;
#pragma unroll
        for (int I = 0; I < NBI; ++I)
#pragma unroll
          for (int J = 0; J < NJ; ++J) {
            const int row = 2 * I + (lp.r >> 1), col = 4 * J + lp.c;
            const bool in = row < D && col < D;
            const double idv = (in && row == col && (lp.r & 1) == 0) ? 1.0 : 0.0;  // identity when idle
            P[I][J] = (act && in) ? src[(row * D + col) * 2 + (lp.r & 1)] : idv;
          }
      };
      fetch(U, 0);
      if (cmax > 1) fetch(Pn, 1);
      for (int t = 1; t < cmax; ++t) {
        double P[NBI][NJ];
#pragma unroll
        for (int I = 0; I < NBI; ++I)
#pragma unroll
          for (int J = 0; J < NJ; ++J) P[I][J] = Pn[I][J];
        if (t + 1 < cmax) fetch(Pn, t + 1);
        {
          double acc[NBI][NJ];
#pragma unroll
          for (int I = 0; I < NBI; ++I)
#pragma unroll
            for (int J = 0; J < NJ; ++J) acc[I][J] = 0.0;
          write_image<D>(P, img, woff);
          mm_img<D>(img, roff, negmask, U, acc);
#pragma unroll
          for (int I = 0; I < NBI; ++I)
#pragma unroll
            for (int J = 0; J < NJ; ++J) U[I][J] = acc[I][J];
        }
      }
      SD_TICK(tk6);
#pragma unroll
      for (int I = 0; I < NBI; ++I)
#pragma unroll
        for (int J = 0; J < NJ; ++J) V[I][J] = Wt[I][J] = 0.0;
      write_image<D>(U, img, woff);
      mm_img<D>(img, roff1, negmask, U, V);
      write_image<D>(V, img, woff);
      mm_img<D>(img, roff2, negmask, V, Wt);
      SD_TICK(tk7);
      double* dst = reinterpret_cast<double*>(A.final_out) + (long)sample * D * D * 2;
      if constexpr (MW)
        store_plain<D>(Wt, dst, 1.0, 0.0, ph, lp, lp.b == 0, phc, phs);
      else
        store_plain<D>(Wt, dst, 1.0, 0.0, ph, lp, lp.b == 0);
#ifdef C3P_SD_TIMING
      SD_TICK(tk4);
      if (MW && blockIdx.x == 0 && threadIdx.x == 0)
        printf("sd timing (100 MHz ticks): tables %lld signals %lld prologue %lld loop %lld fold4+store %lld barrier %lld level1 %lld fold2 %lld store %lld total %lld\n", tk8 - tk0, tk9 - tk8, tk1 - tk0, tk2 - tk1, tk3 - tk2, tk5 - tk3, tk6 - tk5, tk7 - tk6, tk4 - tk7, tk4 - tk0);
#endif
      return;
    }
  }
  double sn, cs;
  sincos(mus_i, &sn, &cs);
  const double er = exp(mus_r);
  double* dst = reinterpret_cast<double*>(A.seg_out) + (long)cc * D * D * 2;
  const double* ph = A.fr_phase ? A.fr_phase + (long)sample * D : nullptr;
  store_plain<D>(U, dst, er * cs, er * sn, ph, lp, valid);
}

// Table preparation: one block per (sample, table index).  G = fac * h (unitary: fac = -i dt)
// or the Lindblad generator (table 0 carries the dissipator).  Output image: Zh layout,
// zero padded, followed by {Re mu, Im mu, ||G - mu||_1, 0}.
template <int D>
__global__ void __launch_bounds__(64) smalld_prep_kernel(PrepArgs P) {
  using C = SD<D>;
  constexpr int MAT = C::MAT, W = C::W;
  __shared__ double g[D * D * 2];
  __shared__ double colsum[D];
  __shared__ double mu[2];
  const int tid = threadIdx.x;
  const int ti = blockIdx.x % (1 + P.K);
  const int sample = blockIdx.x / (1 + P.K);
  if (P.counters != nullptr && blockIdx.x == 0)
    for (int e = tid; e < P.ncounters; e += 64) P.counters[e] = 0;
  const int Dh = P.Dh;  // Hilbert dimension of the inputs (D for unitary, sqrt(D) for Lindblad)
  const cplx* h = (ti == 0) ? P.h0 + (long)sample * P.h0_bstride
                            : P.hks + (long)sample * P.hks_bstride + (long)(ti - 1) * Dh * Dh;
  for (int e = tid; e < D * D; e += 64) {
    const int row = e / D, col = e - row * D;
    cplx v;
    if (!P.lindblad) {
      const cplx x = h[e];
      v = cmake(x.y * P.dt, -x.x * P.dt);  // -i dt h
    } else {
      const int i = row / Dh, j = row - i * Dh, k = col / Dh, l = col - k * Dh;
      v = (ti == 0) ? P.clp[e] : cmake(0, 0);
      if (j == l) {
        const cplx x = h[i * Dh + k];
        v.x += x.y;
        v.y -= x.x;
      }
      if (i == k) {
        const cplx x = h[l * Dh + j];
        v.x -= x.y;
        v.y += x.x;
      }
      v = cscale(v, P.dt);
    }
    const int eo = P.conjT ? col * D + row : e;  // G^H: element (col, row) = conj(G[row][col])
    g[2 * eo] = v.x;
    g[2 * eo + 1] = P.conjT ? -v.y : v.y;
  }
  __syncthreads();
  if (tid == 0) {
    double tr = 0, ti2 = 0;
    for (int i = 0; i < D; ++i) {
      tr += g[2 * (i * D + i)];
      ti2 += g[2 * (i * D + i) + 1];
    }
    mu[0] = 0.0;  // imaginary shift only (see build_tables)
    mu[1] = ti2 / D;
  }
  __syncthreads();
  if (tid < D) {
    g[2 * (tid * D + tid)] -= mu[0];
    g[2 * (tid * D + tid) + 1] -= mu[1];
  }
  __syncthreads();
  if (tid < D) {
    double s = 0;
    for (int i = 0; i < D; ++i) s += hypot(g[2 * (i * D + tid)], g[2 * (i * D + tid) + 1]);
    colsum[tid] = s;
  }
  __syncthreads();
  double* out = P.tables + ((long)sample * (1 + P.K) + ti) * (MAT + 4);
  for (int e = tid; e < MAT; e += 64) {
    const int rho = e / W, col = e - rho * W;
    const int i = rho >> 1, p = rho & 1;
    out[e] = (i < D && col < D) ? g[2 * (i * D + col) + p] : 0.0;
  }
  if (tid == 0) {
    double nrm = 0;
    for (int j = 0; j < D; ++j) nrm = fmax(nrm, colsum[j]);
    out[MAT + 0] = mu[0];
    out[MAT + 1] = mu[1];
    out[MAT + 2] = nrm;
    // 0 exactly when the Hamiltonian is real (the generator is purely imaginary): selects the real fast path
    double remax = 0, asym = 0, rsk = 0;  // real AND symmetric (to rounding: see build_tables)
    for (int e = 0; e < D * D; ++e) {
      const int i = e / D, j = e - i * D;
      remax = fmax(remax, fabs(g[2 * e]));
      asym = fmax(asym, fabs(g[2 * e + 1] - g[2 * (j * D + i) + 1]));
      rsk = fmax(rsk, fabs(g[2 * e] + g[2 * (j * D + i)]));
    }
    if (asym > 1e-14 * nrm) remax = fmax(remax, asym);
    if (remax > 0.0 && fmax(asym, rsk) <= 1e-14 * nrm) remax = -remax;  // skew-Hermitian: see build_tables
    out[MAT + 3] = P.lindblad ? 1.0 : remax;
  }
}

template <int D>
hipError_t launch_chain_t(const SmallArgs& A, hipStream_t st) {
  using C = SD<D>;
  const long nchains = (long)A.B * A.S;
  const unsigned grid = (unsigned)((nchains + 3) / 4);
  size_t lds = 0;
  if (A.mode == C3P_MODE_GIVEN || A.mode == C3P_MODE_EXPM)
    lds = (size_t)(4 * C::IMG) * sizeof(double);
  else
    lds = (size_t)((1 + A.K) * (C::MAT + 4) + 4 * C::IMG + 4 * ((A.K * A.Lmax) | 1)) * sizeof(double);
  if (lds > 60 * 1024) return hipErrorInvalidValue;
  if (A.mode == C3P_MODE_EXPM) {
    if (A.dUs_out)
      C3P_LAUNCH((smalld_chain_kernel<D, false, true, true>), dim3(grid), dim3(64), lds, st, A);
    else
      C3P_LAUNCH((smalld_chain_kernel<D, false, false, true>), dim3(grid), dim3(64), lds, st, A);
  } else if (A.mode == C3P_MODE_GIVEN)
    C3P_LAUNCH((smalld_chain_kernel<D, true, false>), dim3(grid), dim3(64), lds, st, A);
  else if (A.dUs_out)
    C3P_LAUNCH((smalld_chain_kernel<D, false, true>), dim3(grid), dim3(64), lds, st, A);
  else {
    // one workgroup per sample (MW) when the S / 4 waves of a sample fit a CU together with enough other samples for all
    // B workgroups to be resident at once (8 waves per CU at two per SIMD): B = 256, S = 32 -> 256 workgroups of 8 waves
    const int nW = A.S >> 2;
    const size_t wstride = (size_t)(4 * C::IMG + 4 * ((A.K * A.Lmax) | 1));
    const size_t lds_mw = (size_t)((1 + A.K) * (C::MAT + 4) + nW * wstride + (size_t)nW * D * D * 2) * sizeof(double);
    const bool mw = A.fuse && (A.S & 3) == 0 && (nW == 2 || nW == 4 || nW == 8) && (long)A.B * nW <= 2048 &&
                    lds_mw * (8 / nW) <= (size_t)156 * 1024 && !c3p_opt_on(C3P_OPT_no_mw);
    if (mw) {
      // Two waves share a SIMD (nW = 8) and the arbiter serves the older one first: with equal segments wave w leaves its
      // loop at 64 % of the kernel time and wave w + 4 runs the rest alone, at the (slower) one-wave rate.  Give the older
      // waves the longer segments so that both finish together: measured optimum near 2 : 1 (C3P_MW_SKEW = long share of
      // a pair's slices in per mille, default 640; 500 = equal segments).
      SmallArgs A2 = A;
      if (nW == 8 && A.N >= 4 * A.S) {
        // (measured optimum 640 with the padded tiles, 660 with the core + border form of D = 5, 9, 700 with its border sums on the
        // matrix cores -- round 6: tools/sweep_split81.py, profiles/r06/sweep_skew.txt; one slice of the long segments moves the
        // two waves of a SIMD by 7 us against each other, so the optimum is sharp)
        // (the complex loop of the same kernel keeps 640: profiles/r06/sweep_skew_complex.txt -- its samples re-split on the device)
        int skew = ((D == 9 || D == 5) && !c3p_opt_on(C3P_OPT_no_split81)) ? 700 : 640, skew_c = 640;
        if (c3p_opt(C3P_OPT_mw_skew) >= 0) skew = skew_c = (int)c3p_opt(C3P_OPT_mw_skew);
        if (skew > 500 && skew < 900) {
          const int h = A.S / 2;
          int lmax = 0;
          auto split = [&](int per_mille) {
            int La = (int)(((long)A.N * per_mille) / (500L * A.S));  // = per_mille / 1000 of the 2 N / S slices of a pair of chains
            if (La < 1) La = 1;
            if ((long)h * La > A.N - h) La = (A.N - h) / h;  // at least one slice for every short chain
            const long rest = (long)A.N - (long)h * La;
            const int Lb = (int)((rest + h - 1) / h);
            lmax = std::max(lmax, std::max(La, Lb));
            return La;
          };
          A2.seg_long = split(skew);
          A2.seg_long_c = (skew_c > 500 && skew_c < 900 && skew_c != skew) ? split(skew_c) : 0;
          A2.Lmax = lmax;
        }
      }
      const size_t wstride2 = (size_t)(4 * C::IMG + 4 * ((A2.K * A2.Lmax) | 1));
      const size_t lds_mw2 = (size_t)((1 + A2.K) * (C::MAT + 4) + nW * wstride2 + (size_t)nW * D * D * 2) * sizeof(double);
      if (lds_mw2 > (size_t)156 * 1024) A2 = A;  // (the longer segments need more LDS for their control amplitudes)
      const size_t lds_use = A2.seg_long > 0 ? lds_mw2 : lds_mw;
      auto kern = smalld_chain_kernel<D, false, false, false, true>;
      if constexpr (D == 9 || D == 5) {
        // core + border form of the real path (8 + 1 split at D = 9): no_split81 = 1 keeps the padded 12 x 12 tiles
        if (!c3p_opt_on(C3P_OPT_no_split81)) kern = smalld_chain_kernel<D, false, false, false, true, true>;
      }
      if (lds_use > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_use);
        if (e != hipSuccess) return e;
      }
      C3P_LAUNCH(kern, dim3((unsigned)A.B), dim3(64 * nW), lds_use, st, A2);
    } else {
      auto kern1 = smalld_chain_kernel<D, false, false>;
      if constexpr (D == 9 || D == 5) {
        if (!c3p_opt_on(C3P_OPT_no_split81)) kern1 = smalld_chain_kernel<D, false, false, false, false, true>;
      }
      C3P_LAUNCH(kern1, dim3(grid), dim3(64), lds, st, A);
    }
  }
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Backward sweep of the control gradient (method in c3p_grad.hip) on the matrix cores.
// Same mapping as the forward kernel (chain = MFMA block), but ONE wave per SIMD: the pair evaluation
// of T18 keeps ~13 matrices live (X, X^2, X^3, X^6 and their derivatives, accumulators), which fits
// the 512-register budget of a single-wave SIMD and not the 256 of two.  Two LDS images (value and
// derivative left operands); per slice 18 + 3s products:
//   (T, dT) = pair-T18(X~, M) ; Z = T^H dT ; grad[k] = <Z, G_k> ; M <- T^H (M T).
// ---------------------------------------------------------------------------------------------
template <int D>
__device__ __forceinline__ void write_image_H(const double (&zh)[SD<D>::NBI][SD<D>::NJ], double* img, const LanePos& lp) {
  using C = SD<D>;
  wave_sync();
#pragma unroll
  for (int I = 0; I < C::NBI; ++I)
#pragma unroll
    for (int J = 0; J < C::NJ; ++J) {
      // lane holds (r even ? Re : Im) Z[i][j], i = 2I + r/2, j = 4J + c; Y = Z^H: Yh[2j + p][i] = p ? -Im Z[i][j] : Re Z[i][j]
      const int trow = 2 * (4 * J + lp.c) + (lp.r & 1), tcol = 2 * I + (lp.r >> 1);
      const double v = (lp.r & 1) ? -zh[I][J] : zh[I][J];
      if (trow < 4 * C::NBI) img[lp.b * C::MAT + trow * C::W + tcol] = v;
    }
  wave_sync();
}

template <int D>
__device__ __forceinline__ void mat_zero(double (&m)[SD<D>::NBI][SD<D>::NJ]) {
#pragma unroll
  for (int I = 0; I < SD<D>::NBI; ++I)
#pragma unroll
    for (int J = 0; J < SD<D>::NJ; ++J) m[I][J] = 0.0;
}

// squarings of the real-Hamiltonian backward sweep (smalld_grad_real_kernel below): the economised degree-8 cos / sin pair of
// c3p_common.h (round 6: theta = 1.85 on the product structure of the degree-8 Taylor pair, theta_16 = 0.816)
constexpr int SDG_MAXS = 3;

template <int D>
__device__ __forceinline__ int sdg_real_squarings(double nrm) {
  int ps = 0;
  double p = C3P_MM8_THETA;
  while (p < nrm && ps < 40) {
    p *= 2.0;
    ++ps;
  }
  return ps;
}

template <int D>
__global__ void __launch_bounds__(64, 1) smalld_grad_kernel(SmallGradArgs A) {
  using C = SD<D>;
  constexpr int NBI = C::NBI, NJ = C::NJ, W = C::W, MAT = C::MAT;
  typedef double Mat[NBI][NJ];
  const int lane = threadIdx.x;
  LanePos lp;
  lp.r = lane >> 4;
  lp.b = (lane >> 2) & 3;
  lp.c = lane & 3;
  lp.idx16 = lp.r * 4 + lp.c;
  const int K = A.K;
  double* tab = c3p_sd_lds;
  double* img0 = tab + (1 + K) * (MAT + 4);
  double* img1 = img0 + 4 * MAT;
  double* sg = img1 + 4 * MAT;

  const long chain = (long)blockIdx.x * 4 + lp.b;
  const long nchains = (long)A.B * A.S;
  const bool valid = chain < nchains;
  const long cc = valid ? chain : nchains - 1;
  const int sample = (int)(cc / A.S);
  const int seg = (int)(cc - (long)sample * A.S);
  const int n0 = (int)(((long)seg * A.N) / A.S);
  const int n1 = (int)(((long)(seg + 1) * A.N) / A.S);
  const int len = n1 - n0;

  const int woff = lp.b * MAT + lp.r * W + lp.c;
  const int roff = lp.b * MAT + (2 * (lp.c >> 1) + ((lp.c ^ lp.r) & 1)) * W + (lp.r >> 1);
  const unsigned negmask = (((lp.c & 1) == 0) && ((lp.r & 1) == 1)) ? 0x80000000u : 0u;
  const int toff = lp.r * W + lp.c;
  const int ddelta = (lp.r & 1) ? 1000 : ((lp.r >> 1) - lp.c);
  const int rhalf = lp.r >> 1;

  const double* gt0 = A.tables + (long)(A.tab_per_sample ? sample : 0) * (1 + K) * (MAT + 4);
  for (int e = lane; e < (1 + K) * (MAT + 4); e += 64) tab[e] = gt0[e];
  __syncthreads();
  double nrm = tab[MAT + 2];
  for (int k = 0; k < K; ++k) {
    const double* s = A.signals + ((long)sample * K + k) * A.N + n0;
    double cmax = 0.0;
    for (int t = lp.idx16; t < A.Lmax; t += 16) {
      const double v = (valid && t < len) ? s[t] : 0.0;
      sg[(lp.b * K + k) * A.Lmax + t] = v;
      cmax = fmax(cmax, fabs(v));
    }
    cmax = fmax(cmax, __shfl_xor(cmax, 1));
    cmax = fmax(cmax, __shfl_xor(cmax, 2));
    cmax = fmax(cmax, __shfl_xor(cmax, 16));
    cmax = fmax(cmax, __shfl_xor(cmax, 32));
    nrm = fma(cmax, tab[(k + 1) * (MAT + 4) + MAT + 2], nrm);
  }
  nrm = fmax(nrm, __shfl_xor(nrm, 4));
  nrm = fmax(nrm, __shfl_xor(nrm, 8));
  nrm = readfirstlane_f64(nrm);
  int ps = 0;
  {
    double p = C3P_T18_THETA;
    while (p < nrm && ps < 40) {
      p *= 2.0;
      ++ps;
    }
  }
  ps = __builtin_amdgcn_readfirstlane(ps);
  if (A.skip_real) {  // the real-Hamiltonian sweep has taken this wave's chains (same tables, same norm bound: same decision)
    bool realH = true;
    for (int k = 0; k <= K; ++k) realH = realH && (tab[k * (MAT + 4) + MAT + 3] == 0.0);
    if (__builtin_amdgcn_readfirstlane((int)realH) != 0 && __builtin_amdgcn_readfirstlane(sdg_real_squarings<D>(nrm)) <= SDG_MAXS)
      return;
  }
  const double scale = ldexp(1.0, -ps);
  __syncthreads();

  // adjoint state at the end of this segment, D-layout
  Mat M;
  {
    const double* src = reinterpret_cast<const double*>(A.Mb) + cc * D * D * 2;
#pragma unroll
    for (int I = 0; I < NBI; ++I)
#pragma unroll
      for (int J = 0; J < NJ; ++J) {
        const int row = 2 * I + (lp.r >> 1), col = 4 * J + lp.c;
        M[I][J] = (valid && row < D && col < D) ? src[(row * D + col) * 2 + (lp.r & 1)] : 0.0;
      }
  }

  for (int t = A.Lmax - 1; t >= 0; --t) {
    const bool act = valid && t < len;
    const double sc = act ? scale : 0.0;
    Mat X, dX;
#pragma unroll
    for (int I = 0; I < NBI; ++I)
#pragma unroll
      for (int J = 0; J < NJ; ++J) {
        X[I][J] = sc * tab[toff + I * 4 * W + J * 4];
        dX[I][J] = scale * M[I][J];
      }
    for (int k = 0; k < K; ++k) {
      const double ck = sc * sg[(lp.b * K + k) * A.Lmax + t];
      const double* tk = tab + (k + 1) * (MAT + 4);
#pragma unroll
      for (int I = 0; I < NBI; ++I)
#pragma unroll
        for (int J = 0; J < NJ; ++J) X[I][J] = fma(ck, tk[toff + I * 4 * W + J * 4], X[I][J]);
    }
    write_image<D>(X, img0, woff);
    write_image<D>(dX, img1, woff);
    Mat A2, dA2, A3, dA3, A6, dA6;
    mat_zero<D>(A2), mat_zero<D>(dA2), mat_zero<D>(A3), mat_zero<D>(dA3), mat_zero<D>(A6), mat_zero<D>(dA6);
    mm_img<D>(img0, roff, negmask, X, A2);
    mm_img<D>(img0, roff, negmask, dX, dA2);
    mm_img<D>(img1, roff, negmask, X, dA2);
    mm_img<D>(img0, roff, negmask, A2, A3);
    mm_img<D>(img0, roff, negmask, dA2, dA3);
    mm_img<D>(img1, roff, negmask, A2, dA3);
    write_image<D>(A3, img0, woff);
    write_image<D>(dA3, img1, woff);
    mm_img<D>(img0, roff, negmask, A3, A6);
    mm_img<D>(img0, roff, negmask, dA3, dA6);
    mm_img<D>(img1, roff, negmask, A3, dA6);
    Mat A9, dA9;
    {
      Mat B1, dB1;
      lincomb6<D>(B1, 0.0, C3P_T18_A11, C3P_T18_A21, C3P_T18_A31, 0.0, X, A2, A3, A6, ddelta, rhalf);
      lincomb6<D>(dB1, 0.0, C3P_T18_A11, C3P_T18_A21, C3P_T18_A31, 0.0, dX, dA2, dA3, dA6, ddelta, rhalf);
      write_image<D>(B1, img0, woff);
      write_image<D>(dB1, img1, woff);
    }
    {
      Mat B5, dB5;
      lincomb6<D>(B5, 0.0, 0.0, C3P_T18_B24, C3P_T18_B34, C3P_T18_B64, X, A2, A3, A6, ddelta, rhalf);
      lincomb6<D>(dB5, 0.0, 0.0, C3P_T18_B24, C3P_T18_B34, C3P_T18_B64, dX, dA2, dA3, dA6, ddelta, rhalf);
      lincomb6<D>(A9, C3P_T18_B03, C3P_T18_B13, C3P_T18_B23, C3P_T18_B33, C3P_T18_B63, X, A2, A3, A6, ddelta, rhalf);
      lincomb6<D>(dA9, 0.0, C3P_T18_B13, C3P_T18_B23, C3P_T18_B33, C3P_T18_B63, dX, dA2, dA3, dA6, ddelta, rhalf);
      mm_img<D>(img0, roff, negmask, B5, A9);
      mm_img<D>(img0, roff, negmask, dB5, dA9);
      mm_img<D>(img1, roff, negmask, B5, dA9);
    }
    Mat T, dT;
    {
      Mat L, dL;
      lincomb6<D>(L, C3P_T18_B02, C3P_T18_B12, C3P_T18_B22, C3P_T18_B32, C3P_T18_B62, X, A2, A3, A6, ddelta, rhalf);
      lincomb6<D>(dL, 0.0, C3P_T18_B12, C3P_T18_B22, C3P_T18_B32, C3P_T18_B62, dX, dA2, dA3, dA6, ddelta, rhalf);
#pragma unroll
      for (int I = 0; I < NBI; ++I)
#pragma unroll
        for (int J = 0; J < NJ; ++J) {
          L[I][J] += A9[I][J];
          dL[I][J] += dA9[I][J];
        }
      write_image<D>(L, img0, woff);
      write_image<D>(dL, img1, woff);
    }
    lincomb6<D>(T, 0.0, C3P_T18_B11, C3P_T18_B21, C3P_T18_B31, C3P_T18_B61, X, A2, A3, A6, ddelta, rhalf);
    lincomb6<D>(dT, 0.0, C3P_T18_B11, C3P_T18_B21, C3P_T18_B31, C3P_T18_B61, dX, dA2, dA3, dA6, ddelta, rhalf);
    mm_img<D>(img0, roff, negmask, A9, T);
    mm_img<D>(img0, roff, negmask, dA9, dT);
    mm_img<D>(img1, roff, negmask, A9, dT);
    for (int it = 0; it < ps; ++it) {
      write_image<D>(T, img0, woff);
      write_image<D>(dT, img1, woff);
      Mat T2, dT2;
      mat_zero<D>(T2), mat_zero<D>(dT2);
      mm_img<D>(img0, roff, negmask, T, T2);
      mm_img<D>(img0, roff, negmask, dT, dT2);
      mm_img<D>(img1, roff, negmask, T, dT2);
#pragma unroll
      for (int I = 0; I < NBI; ++I)
#pragma unroll
        for (int J = 0; J < NJ; ++J) {
          T[I][J] = T2[I][J];
          dT[I][J] = dT2[I][J];
        }
    }
    // ---- Z = T^H dT, grad[k] = Re<Z, G_k> = sum Zh.G~h + Re(mu_k conj(tr Z)) ----
    write_image_H<D>(T, img0, lp);
    write_image<D>(M, img1, woff);
    {
      Mat Z;
      mat_zero<D>(Z);
      mm_img<D>(img0, roff, negmask, dT, Z);
      if (A.zout != nullptr) {  // per-slice cotangent of the generator G_n = -i dt H_n
        double* dst = reinterpret_cast<double*>(A.zout) + ((long)sample * A.N + n0 + (act ? t : 0)) * D * D * 2;
        store_plain<D>(Z, dst, 1.0, 0.0, nullptr, lp, act);
      }
      double trr = 0.0, tri = 0.0;
#pragma unroll
      for (int I = 0; I < NBI; ++I)
#pragma unroll
        for (int J = 0; J < NJ; ++J)
          if (2 * I - 4 * J >= -1 && 2 * I - 4 * J <= 3) {
            const bool on = (((lp.r >> 1) - lp.c) == 4 * J - 2 * I) && (2 * I + rhalf < D);
            const double v = on ? Z[I][J] : 0.0;
            if (lp.r & 1)
              tri += v;
            else
              trr += v;
          }
      for (int k = 0; k < K; ++k) {
        const double* tk = tab + (k + 1) * (MAT + 4);
        double part = fma(tk[MAT + 0], trr, tk[MAT + 1] * tri);
#pragma unroll
        for (int I = 0; I < NBI; ++I)
#pragma unroll
          for (int J = 0; J < NJ; ++J) part = fma(Z[I][J], tk[toff + I * 4 * W + J * 4], part);
        part += __shfl_xor(part, 1);
        part += __shfl_xor(part, 2);
        part += __shfl_xor(part, 16);
        part += __shfl_xor(part, 32);
        if (act && lp.idx16 == 0) A.grad[((long)sample * K + k) * A.N + n0 + t] = part;
      }
    }
    // ---- M <- T^H (M T) ----
    {
      Mat V;
      mat_zero<D>(V);
      mm_img<D>(img1, roff, negmask, T, V);
      mat_zero<D>(M);
      mm_img<D>(img0, roff, negmask, V, M);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Backward sweep for GENERAL generators (Lindblad superoperators up to 12 x 12, i.e. D <= 3): the slices are not unitary, so
// the adjoint state cannot be carried as M_n = dU_n^H M_{n+1} dU_n.  Method of c3p_grad.hip (general form):
//   grad[k,n] = Re <L(X_n^H; M_n), G_k>,  M_n = A_n P_n^H,  P_n = dU_{n-1} .. dU_0 (prefix),  A_n = S_n^H FR^H Ubar (left adjoint).
// Per chain: a forward walk over the segment, P <- dU_n P with the slice propagators of the forward pass (one product per
// slice), every P_n stored; then backwards, ONE pair evaluation of T18 per slice at Y = X_n^H (tables of G^H): the value
// exp(Y) = dU_n^H advances A, the derivative L(Y; M_n) is the generator's cotangent.  The trace shifts do not cancel here: A and
// the derivative carry them as ONE complex scalar per chain, applied to the complex inner product <dT, G_k> at the end.
// ---------------------------------------------------------------------------------------------
template <int D, bool XG = false>
__global__ void __launch_bounds__(64, 1) smalld_grad_general_kernel(SmallGradArgs A) {
  using C = SD<D>;
  constexpr int NBI = C::NBI, NJ = C::NJ, W = C::W, MAT = C::MAT;
  typedef double Mat[NBI][NJ];
  const int lane = threadIdx.x;
  LanePos lp;
  lp.r = lane >> 4;
  lp.b = (lane >> 2) & 3;
  lp.c = lane & 3;
  lp.idx16 = lp.r * 4 + lp.c;
  const int K = XG ? 0 : A.K;
  double* tab = c3p_sd_lds;                 // G tables: inner products
  double* tabh = tab + (1 + K) * (MAT + 4);  // G^H tables: the matrix that is exponentiated
  double* img0 = tabh + (1 + K) * (MAT + 4);
  double* img1 = img0 + 4 * MAT;
  double* sg = img1 + 4 * MAT;

  const long chain = (long)blockIdx.x * 4 + lp.b;
  const long nchains = (long)A.B * A.S;
  const bool valid = chain < nchains;
  const long cc = valid ? chain : nchains - 1;
  const int sample = (int)(cc / A.S);
  const int seg = (int)(cc - (long)sample * A.S);
  const int n0 = (int)(((long)seg * A.N) / A.S);
  const int n1 = (int)(((long)(seg + 1) * A.N) / A.S);
  const int len = n1 - n0;

  const int woff = lp.b * MAT + lp.r * W + lp.c;
  const int roff = lp.b * MAT + (2 * (lp.c >> 1) + ((lp.c ^ lp.r) & 1)) * W + (lp.r >> 1);
  const unsigned negmask = (((lp.c & 1) == 0) && ((lp.r & 1) == 1)) ? 0x80000000u : 0u;
  const int toff = lp.r * W + lp.c;
  const int poff = (lp.r ^ 1) * W + lp.c;  // the same element's other part (Re <-> Im) in a table
  const int ddelta = (lp.r & 1) ? 1000 : ((lp.r >> 1) - lp.c);
  const int rhalf = lp.r >> 1;

  const long tsz = (long)(1 + K) * (MAT + 4);
  double nrm = 0.0;
  if constexpr (XG) {
    // supplied generators: the matrix exponentiated is X_n^H, whose 1-norm is the row-sum norm of X_n (slot 3 of the hmeta pass)
    const double* mt = A.meta + ((long)sample * A.N + n0) * 4;
    for (int t = lp.idx16; t < A.Lmax; t += 16)
      if (valid && t < len) nrm = fmax(nrm, mt[(long)t * 4 + 3]);
    nrm = fmax(nrm, __shfl_xor(nrm, 1));
    nrm = fmax(nrm, __shfl_xor(nrm, 2));
    nrm = fmax(nrm, __shfl_xor(nrm, 16));
    nrm = fmax(nrm, __shfl_xor(nrm, 32));
  } else {
    const double* gt0 = A.tables + (long)(A.tab_per_sample ? sample : 0) * tsz;
    const double* gh0 = A.tables_h + (long)(A.tab_per_sample ? sample : 0) * tsz;
    for (int e = lane; e < (int)tsz; e += 64) {
      tab[e] = gt0[e];
      tabh[e] = gh0[e];
    }
    __syncthreads();
    nrm = tabh[MAT + 2];
  }
  for (int k = 0; k < K; ++k) {
    const double* s = A.signals + ((long)sample * K + k) * A.N + n0;
    double cmax = 0.0;
    for (int t = lp.idx16; t < A.Lmax; t += 16) {
      const double v = (valid && t < len) ? s[t] : 0.0;
      sg[(lp.b * K + k) * A.Lmax + t] = v;
      cmax = fmax(cmax, fabs(v));
    }
    cmax = fmax(cmax, __shfl_xor(cmax, 1));
    cmax = fmax(cmax, __shfl_xor(cmax, 2));
    cmax = fmax(cmax, __shfl_xor(cmax, 16));
    cmax = fmax(cmax, __shfl_xor(cmax, 32));
    nrm = fma(cmax, tabh[(k + 1) * (MAT + 4) + MAT + 2], nrm);
  }
  nrm = fmax(nrm, __shfl_xor(nrm, 4));
  nrm = fmax(nrm, __shfl_xor(nrm, 8));
  nrm = readfirstlane_f64(nrm);
  int ps = 0;
  {
    double p = C3P_T18_THETA;
    while (p < nrm && ps < 40) {
      p *= 2.0;
      ++ps;
    }
  }
  ps = __builtin_amdgcn_readfirstlane(ps);
  const double scale = ldexp(1.0, -ps);
  __syncthreads();

  // element (row, col), part r & 1 of a plain complex matrix in global memory -> D layout (TR: its conjugate transpose)
  auto load_plain = [&](Mat& m, const cplx* srcc, bool on, bool TR) {
    const double* src = reinterpret_cast<const double*>(srcc);
#pragma unroll
    for (int I = 0; I < NBI; ++I)
#pragma unroll
      for (int J = 0; J < NJ; ++J) {
        const int row = 2 * I + (lp.r >> 1), col = 4 * J + lp.c;
        double v = 0.0;
        if (on && row < D && col < D) {
          v = TR ? src[(col * D + row) * 2 + (lp.r & 1)] : src[(row * D + col) * 2 + (lp.r & 1)];
          if (TR && (lp.r & 1)) v = -v;
        }
        m[I][J] = v;
      }
  };
  auto set_identity = [&](Mat& m) {
#pragma unroll
    for (int I = 0; I < NBI; ++I)
#pragma unroll
      for (int J = 0; J < NJ; ++J) {
        const int row = 2 * I + (lp.r >> 1), col = 4 * J + lp.c;
        m[I][J] = (!(lp.r & 1) && row == col && row < D) ? 1.0 : 0.0;
      }
  };

  // ---- forward: the prefix in front of every slice of the segment ----
  cplx* pst = A.pstore + ((long)sample * A.N + n0) * D * D;
  {
    Mat P;
    load_plain(P, A.pre + cc * D * D, valid, false);
    const cplx* du = A.dUs + ((long)sample * A.N + n0) * D * D;
    for (int t = 0; t < A.Lmax; ++t) {
      const bool act = valid && t < len;
      store_plain<D>(P, reinterpret_cast<double*>(pst + (long)(act ? t : 0) * D * D), 1.0, 0.0, nullptr, lp, act);
      if (t + 1 == A.Lmax) break;
      Mat E, V;
      if (__builtin_amdgcn_readfirstlane((int)__any(act)) == 0) continue;
      load_plain(E, du + (long)(act ? t : 0) * D * D, act, false);
      if (!act) set_identity(E);
      write_image<D>(E, img0, woff);
      mat_zero<D>(V);
      mm_img<D>(img0, roff, negmask, P, V);
#pragma unroll
      for (int I = 0; I < NBI; ++I)
#pragma unroll
        for (int J = 0; J < NJ; ++J) P[I][J] = V[I][J];
    }
  }
  __threadfence_block();

  // ---- backward ----
  Mat Aa;  // left adjoint, WITHOUT the trace shifts of the slices behind it: they accumulate in (ams_r, ams_i)
  load_plain(Aa, A.Mb + cc * D * D, valid, false);
  double ams_r = 0.0, ams_i = 0.0;
  for (int t = A.Lmax - 1; t >= 0; --t) {
    const bool act = valid && t < len;
    const double sc = act ? scale : 0.0;
    Mat X, dX;
    {
      Mat PH, Mn;
      load_plain(PH, pst + (long)(act ? t : 0) * D * D, act, true);
      write_image<D>(Aa, img0, woff);
      mat_zero<D>(Mn);
      mm_img<D>(img0, roff, negmask, PH, Mn);  // M_n = A P_n^H
#pragma unroll
      for (int I = 0; I < NBI; ++I)
#pragma unroll
        for (int J = 0; J < NJ; ++J) dX[I][J] = scale * Mn[I][J];
    }
    double mu_r = 0.0, mu_i = 0.0;  // trace shift of Y = X_n^H
    if constexpr (XG) {
      // Y = conj(coef) hs[b,n]^H - conj(mu_n): element (row, col) = conj(coef hs[col][row])
      const long m = (long)sample * A.N + n0 + (act ? t : 0);
      mu_r = act ? A.meta[m * 4 + 0] : 0.0;
      mu_i = act ? -A.meta[m * 4 + 1] : 0.0;
      const double2* src = reinterpret_cast<const double2*>(A.hs) + (long)sample * A.hs_bstride + (long)(n0 + (act ? t : 0)) * D * D;
      const double mine = (lp.r & 1) ? mu_i : mu_r;
#pragma unroll
      for (int I = 0; I < NBI; ++I)
#pragma unroll
        for (int J = 0; J < NJ; ++J) {
          const int row = 2 * I + (lp.r >> 1), col = 4 * J + lp.c;
          const bool in = row < D && col < D;
          const double2 h = src[in ? col * D + row : 0];
          // coef h = (cr hx - ci hy) + i (cr hy + ci hx); conjugated for the transpose
          double v = (lp.r & 1) ? -fma(A.coef_r, h.y, A.coef_i * h.x) : fma(A.coef_r, h.x, -A.coef_i * h.y);
          v -= (row == col) ? mine : 0.0;
          X[I][J] = in ? sc * v : 0.0;
        }
    } else {
    mu_r = act ? tabh[MAT + 0] : 0.0;
    mu_i = act ? tabh[MAT + 1] : 0.0;
#pragma unroll
    for (int I = 0; I < NBI; ++I)
#pragma unroll
      for (int J = 0; J < NJ; ++J) X[I][J] = sc * tabh[toff + I * 4 * W + J * 4];
    }
    for (int k = 0; k < K; ++k) {
      const double c0 = sg[(lp.b * K + k) * A.Lmax + t];  // zero for inactive slices
      const double ck = sc * c0;
      const double* tk = tabh + (k + 1) * (MAT + 4);
      mu_r = fma(c0, tk[MAT + 0], mu_r);
      mu_i = fma(c0, tk[MAT + 1], mu_i);
#pragma unroll
      for (int I = 0; I < NBI; ++I)
#pragma unroll
        for (int J = 0; J < NJ; ++J) X[I][J] = fma(ck, tk[toff + I * 4 * W + J * 4], X[I][J]);
    }
    write_image<D>(X, img0, woff);
    write_image<D>(dX, img1, woff);
    Mat A2, dA2, A3, dA3, A6, dA6;
    mat_zero<D>(A2), mat_zero<D>(dA2), mat_zero<D>(A3), mat_zero<D>(dA3), mat_zero<D>(A6), mat_zero<D>(dA6);
    mm_img<D>(img0, roff, negmask, X, A2);
    mm_img<D>(img0, roff, negmask, dX, dA2);
    mm_img<D>(img1, roff, negmask, X, dA2);
    mm_img<D>(img0, roff, negmask, A2, A3);
    mm_img<D>(img0, roff, negmask, dA2, dA3);
    mm_img<D>(img1, roff, negmask, A2, dA3);
    write_image<D>(A3, img0, woff);
    write_image<D>(dA3, img1, woff);
    mm_img<D>(img0, roff, negmask, A3, A6);
    mm_img<D>(img0, roff, negmask, dA3, dA6);
    mm_img<D>(img1, roff, negmask, A3, dA6);
    Mat A9, dA9;
    {
      Mat B1, dB1;
      lincomb6<D>(B1, 0.0, C3P_T18_A11, C3P_T18_A21, C3P_T18_A31, 0.0, X, A2, A3, A6, ddelta, rhalf);
      lincomb6<D>(dB1, 0.0, C3P_T18_A11, C3P_T18_A21, C3P_T18_A31, 0.0, dX, dA2, dA3, dA6, ddelta, rhalf);
      write_image<D>(B1, img0, woff);
      write_image<D>(dB1, img1, woff);
    }
    {
      Mat B5, dB5;
      lincomb6<D>(B5, 0.0, 0.0, C3P_T18_B24, C3P_T18_B34, C3P_T18_B64, X, A2, A3, A6, ddelta, rhalf);
      lincomb6<D>(dB5, 0.0, 0.0, C3P_T18_B24, C3P_T18_B34, C3P_T18_B64, dX, dA2, dA3, dA6, ddelta, rhalf);
      lincomb6<D>(A9, C3P_T18_B03, C3P_T18_B13, C3P_T18_B23, C3P_T18_B33, C3P_T18_B63, X, A2, A3, A6, ddelta, rhalf);
      lincomb6<D>(dA9, 0.0, C3P_T18_B13, C3P_T18_B23, C3P_T18_B33, C3P_T18_B63, dX, dA2, dA3, dA6, ddelta, rhalf);
      mm_img<D>(img0, roff, negmask, B5, A9);
      mm_img<D>(img0, roff, negmask, dB5, dA9);
      mm_img<D>(img1, roff, negmask, B5, dA9);
    }
    Mat T, dT;
    {
      Mat L, dL;
      lincomb6<D>(L, C3P_T18_B02, C3P_T18_B12, C3P_T18_B22, C3P_T18_B32, C3P_T18_B62, X, A2, A3, A6, ddelta, rhalf);
      lincomb6<D>(dL, 0.0, C3P_T18_B12, C3P_T18_B22, C3P_T18_B32, C3P_T18_B62, dX, dA2, dA3, dA6, ddelta, rhalf);
#pragma unroll
      for (int I = 0; I < NBI; ++I)
#pragma unroll
        for (int J = 0; J < NJ; ++J) {
          L[I][J] += A9[I][J];
          dL[I][J] += dA9[I][J];
        }
      write_image<D>(L, img0, woff);
      write_image<D>(dL, img1, woff);
    }
    lincomb6<D>(T, 0.0, C3P_T18_B11, C3P_T18_B21, C3P_T18_B31, C3P_T18_B61, X, A2, A3, A6, ddelta, rhalf);
    lincomb6<D>(dT, 0.0, C3P_T18_B11, C3P_T18_B21, C3P_T18_B31, C3P_T18_B61, dX, dA2, dA3, dA6, ddelta, rhalf);
    mm_img<D>(img0, roff, negmask, A9, T);
    mm_img<D>(img0, roff, negmask, dA9, dT);
    mm_img<D>(img1, roff, negmask, A9, dT);
    for (int it = 0; it < ps; ++it) {
      write_image<D>(T, img0, woff);
      write_image<D>(dT, img1, woff);
      Mat T2, dT2;
      mat_zero<D>(T2), mat_zero<D>(dT2);
      mm_img<D>(img0, roff, negmask, T, T2);
      mm_img<D>(img0, roff, negmask, dT, dT2);
      mm_img<D>(img1, roff, negmask, T, dT2);
#pragma unroll
      for (int I = 0; I < NBI; ++I)
#pragma unroll
        for (int J = 0; J < NJ; ++J) {
          T[I][J] = T2[I][J];
          dT[I][J] = dT2[I][J];
        }
    }
    // ---- grad[k] = Re <e^{ams + mu} dT, G_k> = Re( conj(e^{ams + mu}) <dT, G_k> ), complex inner product <Z, G> = sum conj(Z) G ----
    {
      double sn, cs;
      sincos(ams_i + mu_i, &sn, &cs);
      const double er = exp(ams_r + mu_r);
      const double pr = er * cs, pi = er * sn;
      if (A.zout != nullptr) {  // Z_n = e^{shift} dT: the cotangent of the generator X_n
        double* dst = reinterpret_cast<double*>(A.zout) + ((long)sample * A.N + n0 + (act ? t : 0)) * D * D * 2;
        store_plain<D>(dT, dst, pr, pi, nullptr, lp, act);
      }
      double trr = 0.0, tri = 0.0;
#pragma unroll
      for (int I = 0; I < NBI; ++I)
#pragma unroll
        for (int J = 0; J < NJ; ++J)
          if (2 * I - 4 * J >= -1 && 2 * I - 4 * J <= 3) {
            const bool on = (((lp.r >> 1) - lp.c) == 4 * J - 2 * I) && (2 * I + rhalf < D);
            const double v = on ? dT[I][J] : 0.0;
            if (lp.r & 1)
              tri += v;
            else
              trr += v;
          }
      for (int k = 0; k < K; ++k) {
        const double* tk = tab + (k + 1) * (MAT + 4);
        // trace part: conj(tr Z) mu_k
        double re = fma(tk[MAT + 0], trr, tk[MAT + 1] * tri);
        double im = fma(tk[MAT + 1], trr, -tk[MAT + 0] * tri);
#pragma unroll
        for (int I = 0; I < NBI; ++I)
#pragma unroll
          for (int J = 0; J < NJ; ++J) {
            const double z = dT[I][J];
            re = fma(z, tk[toff + I * 4 * W + J * 4], re);
            const double go = tk[poff + I * 4 * W + J * 4];  // r even: Im G of my element; r odd: Re G
            im = (lp.r & 1) ? fma(-z, go, im) : fma(z, go, im);
          }
        double part = fma(pr, re, pi * im);
        part += __shfl_xor(part, 1);
        part += __shfl_xor(part, 2);
        part += __shfl_xor(part, 16);
        part += __shfl_xor(part, 32);
        if (act && lp.idx16 == 0) A.grad[((long)sample * K + k) * A.N + n0 + t] = part;
      }
    }
    // ---- A <- dU_n^H A = e^{mu} T A ----
    {
      Mat V;
      write_image<D>(T, img0, woff);
      mat_zero<D>(V);
      mm_img<D>(img0, roff, negmask, Aa, V);
#pragma unroll
      for (int I = 0; I < NBI; ++I)
#pragma unroll
        for (int J = 0; J < NJ; ++J) Aa[I][J] = V[I][J];
      ams_r += mu_r;
      ams_i = c3p_phase_add(ams_i, mu_i);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Backward sweep for REAL Hamiltonians (the case of every dressed lab-frame transmon model): reverse mode through the
// real cos / sin evaluation of the forward kernel instead of the forward-mode pair evaluation of complex T18 above.
// With X = -iY (Y real symmetric), dU = C - iS, and the adjoint state carried TRANSPOSED, N = M^T:
//   R = dU N = (M dU)^T  ->  the cotangents of C and S are REAL and may be symmetrised:  C_bar = sym Re R, S_bar = -sym Im R;
//   every adjoint of the symmetric stage is a symmetrised product of symmetric matrices, 2 sym(A_bar B) = A_bar B + B A_bar
//   (two upper-triangle products whose left operands come from registers, as in the forward kernel);
//   grad[k, n] = scale <Y_bar, Y_k> + Re(mu_k conj(tr N));   N <- R conj(dU)  (the only product with a general LEFT operand:
//   R goes through a real LDS image).
// ~600 MFMAs per slice at D = 9 against ~1500 of the complex pair evaluation.  Always the degree-16 / 17 polynomials
// (theta_16 = 0.816) with up to SDG_MAXS squarings; samples that are not real, or need more squarings, are left to the
// complex kernel (same norm bound, same decision).
// ---------------------------------------------------------------------------------------------
template <int D>
__global__ void __launch_bounds__(64, 1) smalld_grad_real_kernel(SmallGradArgs A) {
  using C = SD<D>;
  constexpr int W = C::W, MAT = C::MAT;
  constexpr int NB = RD<D>::NB, WR = 4 * NB + 1, RIMG = 4 * NB * WR;
  typedef double RMat[NB][NB];
  const int lane = threadIdx.x;
  LanePos lp;
  lp.r = lane >> 4;
  lp.b = (lane >> 2) & 3;
  lp.c = lane & 3;
  lp.idx16 = lp.r * 4 + lp.c;
  const int K = A.K;
  double* tab = c3p_sd_lds;
  double* rimg = tab + (1 + K) * (MAT + 4);  // per chain: Re R, Im R as real row-major images (row stride WR)
  double* sg = rimg + 4 * 2 * RIMG;

  const long chain = (long)blockIdx.x * 4 + lp.b;
  const long nchains = (long)A.B * A.S;
  const bool valid = chain < nchains;
  const long cc = valid ? chain : nchains - 1;
  const int sample = (int)(cc / A.S);
  const int seg = (int)(cc - (long)sample * A.S);
  const int n0 = (int)(((long)seg * A.N) / A.S);
  const int n1 = (int)(((long)(seg + 1) * A.N) / A.S);
  const int len = n1 - n0;

  const double* gt0 = A.tables + (long)(A.tab_per_sample ? sample : 0) * (1 + K) * (MAT + 4);
  for (int e = lane; e < (1 + K) * (MAT + 4); e += 64) tab[e] = gt0[e];
  __syncthreads();
  double nrm = tab[MAT + 2];
  for (int k = 0; k < K; ++k) {
    const double* s = A.signals + ((long)sample * K + k) * A.N + n0;
    double cmax = 0.0;
    for (int t = lp.idx16; t < A.Lmax; t += 16) {
      const double v = (valid && t < len) ? s[t] : 0.0;
      sg[(lp.b * K + k) * A.Lmax + t] = v;
      cmax = fmax(cmax, fabs(v));
    }
    cmax = fmax(cmax, __shfl_xor(cmax, 1));
    cmax = fmax(cmax, __shfl_xor(cmax, 2));
    cmax = fmax(cmax, __shfl_xor(cmax, 16));
    cmax = fmax(cmax, __shfl_xor(cmax, 32));
    nrm = fma(cmax, tab[(k + 1) * (MAT + 4) + MAT + 2], nrm);
  }
  nrm = fmax(nrm, __shfl_xor(nrm, 4));
  nrm = fmax(nrm, __shfl_xor(nrm, 8));
  nrm = readfirstlane_f64(nrm);
  bool realH = true;
  for (int k = 0; k <= K; ++k) realH = realH && (tab[k * (MAT + 4) + MAT + 3] == 0.0);
  // (without per-sample tables the four chains of a wave may belong to different samples, but then all samples share
  // the tables; with per-sample tables S % 4 == 0 and the wave has one sample)
  const int ps = __builtin_amdgcn_readfirstlane(sdg_real_squarings<D>(nrm));
  if (!(__builtin_amdgcn_readfirstlane((int)realH) != 0) || ps > SDG_MAXS) return;
  const double scale = ldexp(1.0, -ps);
  __syncthreads();

  const int swap_lane = 16 * lp.c + 4 * lp.b + lp.r;
  const int tail_lane = 16 * lp.c + 4 * lp.b;
  const int row0_lane = 4 * lp.b + lp.c;
  int yo[NB];
  double ymask[NB];
#pragma unroll
  for (int I = 0; I < NB; ++I) {
    const bool ok = 4 * I + lp.r < D;
    yo[I] = (2 * (ok ? 4 * I + lp.r : 0) + 1) * W + lp.c;
    ymask[I] = ok ? 1.0 : 0.0;
  }
  auto zero = [&](RMat& m) {
#pragma unroll
    for (int I = 0; I < NB; ++I)
#pragma unroll
      for (int J = 0; J < NB; ++J) m[I][J] = 0.0;
  };
  // acc (all tiles) += A B for symmetric A, B (operand tiles filled)
  auto prod = [&](const RMat& Am, const RMat& Bm, RMat& acc) { mm_symfull<D>(Am, Bm, acc, tail_lane); };
  // out = f (P + P^T) on the tiles a product reads as operands (rows I < KM in full, upper tiles of the others): element
  // (4J + c, 4I + r) of P sits in lane (c, r) of tile (J, I).  One round of lane swaps, no separate mirror fill.
  auto mirror = [&](const RMat& P, double f, RMat& out) {
#pragma unroll
    for (int I = 0; I < NB; ++I)
#pragma unroll
      for (int J = sym_j0<D>(I); J < NB; ++J) out[I][J] = f * (P[I][J] + __shfl(P[J][I], swap_lane));
  };

  // N = M^T at the end of this segment
  RMat Nr, Ni;
  {
    const double* src = reinterpret_cast<const double*>(A.Mb) + cc * D * D * 2;
#pragma unroll
    for (int I = 0; I < NB; ++I)
#pragma unroll
      for (int J = 0; J < NB; ++J) {
        const int row = 4 * I + lp.r, col = 4 * J + lp.c;
        const bool ok = valid && row < D && col < D;
        Nr[I][J] = ok ? src[(col * D + row) * 2 + 0] : 0.0;
        Ni[I][J] = ok ? src[(col * D + row) * 2 + 1] : 0.0;
      }
  }

  // Two instances of the slice loop, the branch outside (as in the forward kernel): MM6 = the economised degree-6 pair (scaled norm
  // <= 0.83: cfg2) -- Horner step in W^3, no W^4: one forward product and one adjoint product less per slice; otherwise the
  // degree-8 pair with W^4 (theta = 1.85).
  const bool small_var = __builtin_amdgcn_readfirstlane((int)(nrm * scale <= C3P_MM6_THETA)) != 0;
  auto sweep = [&](auto mm6_tag) {
  constexpr bool MM6 = decltype(mm6_tag)::value;
  for (int t = A.Lmax - 1; t >= 0; --t) {
    const bool act = valid && t < len;
    const double sc = act ? scale : 0.0;
    // ---- forward: Y, W = Y^2, ..., cos Y, sin Y (as in smalld_chain_kernel) ----
    RMat Y;
#pragma unroll
    for (int I = 0; I < NB; ++I) {
      const double f = -sc * ymask[I];
#pragma unroll
      for (int J = sym_j0<D>(I); J < NB; ++J) Y[I][J] = f * lds_ld(tab + yo[I] + J * 4);
    }
    for (int k = 0; k < K; ++k) {
      const double ck = sc * sg[(lp.b * K + k) * A.Lmax + t];
      const double* tk = tab + (k + 1) * (MAT + 4);
#pragma unroll
      for (int I = 0; I < NB; ++I) {
        const double f = -ck * ymask[I];
#pragma unroll
        for (int J = sym_j0<D>(I); J < NB; ++J) Y[I][J] = fma(f, lds_ld(tk + yo[I] + J * 4), Y[I][J]);
      }
    }
    RMat W1, W2, W3, W4, acc, acs, Cm, Sp, S0;
    zero(W1), zero(W2), zero(W3), zero(W4), zero(S0);
    mm_sym<D>(Y, Y, W1, tail_lane);
    sym_fill<D>(W1, swap_lane);
    mm_sym<D>(W1, W1, W2, tail_lane);
    sym_fill<D>(W2, swap_lane);
    mm_sym<D>(W1, W2, W3, tail_lane);
    if constexpr (!MM6) mm_sym<D>(W2, W2, W4, tail_lane);
    sym_fill<D>(W3, swap_lane);
    if constexpr (MM6) {
      // cos = (c0 + c1 W + c2 W^2) + W^3 (c3 + c4 W + c5 W^2 + c6 W^3)
      rcomb<D, true>(acc, c3p_mm6_cos[3], c3p_mm6_cos[4], c3p_mm6_cos[5], c3p_mm6_cos[6], W1, W2, W3, lp);
      rcomb<D, true>(acs, c3p_mm6_sinc[3], c3p_mm6_sinc[4], c3p_mm6_sinc[5], c3p_mm6_sinc[6], W1, W2, W3, lp);
      rcomb<D, false, true>(Cm, c3p_mm6_cos[0], c3p_mm6_cos[1], c3p_mm6_cos[2], 0.0, W1, W2, W3, lp);
      rcomb<D, false, true>(Sp, c3p_mm6_sinc[0], c3p_mm6_sinc[1], c3p_mm6_sinc[2], 0.0, W1, W2, W3, lp);
      mm_sym2<D>(W3, acc, Cm, acs, Sp, tail_lane);
    } else {
      sym_fill<D>(W4, swap_lane);
      rcomb<D, true>(acc, c3p_mm8_cos[4], c3p_mm8_cos[5], c3p_mm8_cos[6], c3p_mm8_cos[7], W1, W2, W3, lp);
      rcomb<D, true>(acs, c3p_mm8_sinc[4], c3p_mm8_sinc[5], c3p_mm8_sinc[6], c3p_mm8_sinc[7], W1, W2, W3, lp);
#pragma unroll
      for (int I = 0; I < NB; ++I)
#pragma unroll
        for (int J = sym_j0<D>(I); J < NB; ++J) {
          acc[I][J] = fma(c3p_mm8_cos[8], W4[I][J], acc[I][J]);
          acs[I][J] = fma(c3p_mm8_sinc[8], W4[I][J], acs[I][J]);
        }
      rcomb<D, true, true>(Cm, c3p_mm8_cos[0], c3p_mm8_cos[1], c3p_mm8_cos[2], c3p_mm8_cos[3], W1, W2, W3, lp);
      rcomb<D, true, true>(Sp, c3p_mm8_sinc[0], c3p_mm8_sinc[1], c3p_mm8_sinc[2], c3p_mm8_sinc[3], W1, W2, W3, lp);
      mm_sym2<D>(W4, acc, Cm, acs, Sp, tail_lane);
    }
    sym_fill<D>(Cm, swap_lane);
    sym_fill<D>(Sp, swap_lane);
    mm_sym<D>(Y, Sp, S0, tail_lane);
    sym_fill<D>(S0, swap_lane);
    // squarings, every level kept for the way back: C' = (C - S)(C + S), S' = 2 S C
    RMat Cl[SDG_MAXS], Sl[SDG_MAXS], Cf, Sf;
#pragma unroll
    for (int I = 0; I < NB; ++I)
#pragma unroll
      for (int J = sym_j0<D>(I); J < NB; ++J) {
        Cf[I][J] = Cm[I][J];
        Sf[I][J] = S0[I][J];
      }
#pragma unroll
    for (int j = 0; j < SDG_MAXS; ++j)
      if (j < ps) {
        RMat Dm, Sm, C2, SC;
#pragma unroll
        for (int I = 0; I < NB; ++I)
#pragma unroll
          for (int J = sym_j0<D>(I); J < NB; ++J) {
            Cl[j][I][J] = Cf[I][J];
            Sl[j][I][J] = Sf[I][J];
            Dm[I][J] = Cf[I][J] - Sf[I][J];
            Sm[I][J] = Cf[I][J] + Sf[I][J];
            C2[I][J] = SC[I][J] = 0.0;
          }
        mm_sym<D>(Dm, Sm, C2, tail_lane);
        mm_sym<D>(Sf, Cf, SC, tail_lane);
#pragma unroll
        for (int I = 0; I < NB; ++I)
#pragma unroll
          for (int J = I; J < NB; ++J) {
            Cf[I][J] = C2[I][J];
            Sf[I][J] = 2.0 * SC[I][J];
          }
        sym_fill<D>(Cf, swap_lane);
        sym_fill<D>(Sf, swap_lane);
      }
    sym_fill_rest<D>(Cf, swap_lane);  // the whole matrices: right operands of the update of N
    sym_fill_rest<D>(Sf, swap_lane);
    // ---- R = dU N = (C - iS)(Nr + i Ni) with three real products ----
    RMat Rr, Ri;
    {
      RMat Ps, Nd, P1, P2, P3;
#pragma unroll
      for (int I = 0; I < NB; ++I)
#pragma unroll
        for (int J = 0; J < NB; ++J) {
          if (J >= sym_j0<D>(I)) Ps[I][J] = Cf[I][J] + Sf[I][J];
          Nd[I][J] = Ni[I][J] - Nr[I][J];
          P1[I][J] = P2[I][J] = P3[I][J] = 0.0;
        }
      mm_symA<D>(Cf, Nr, P1, row0_lane, 1.0);
      mm_symA<D>(Sf, Ni, P2, row0_lane, 1.0);
      mm_symA<D>(Ps, Nd, P3, row0_lane, 1.0);
#pragma unroll
      for (int I = 0; I < NB; ++I)
#pragma unroll
        for (int J = 0; J < NB; ++J) {
          Rr[I][J] = P1[I][J] + P2[I][J];
          Ri[I][J] = (P3[I][J] + P1[I][J]) - P2[I][J];
        }
    }
    // tr N (= tr M): the trace-shift part of the gradient
    double trr = 0.0, tri = 0.0;
#pragma unroll
    for (int I = 0; I < NB; ++I) {
      const bool on = (lp.r == lp.c) && (4 * I + lp.r < D);
      trr += on ? Nr[I][I] : 0.0;
      tri += on ? Ni[I][I] : 0.0;
    }
    // the images of R for the update of N (written now, read at the end of the slice)
    wave_sync();
#pragma unroll
    for (int I = 0; I < NB; ++I)
#pragma unroll
      for (int J = 0; J < NB; ++J) {
        rimg[(lp.b * 2 + 0) * RIMG + (4 * I + lp.r) * WR + 4 * J + lp.c] = Rr[I][J];
        rimg[(lp.b * 2 + 1) * RIMG + (4 * I + lp.r) * WR + 4 * J + lp.c] = Ri[I][J];
      }
    wave_sync();
    // ---- cotangents of cos / sin: C_bar = sym Re R, S_bar = -sym Im R (upper tiles, then the operand tiles) ----
    RMat Cb, Sb;
    mirror(Rr, 0.5, Cb);
    mirror(Ri, -0.5, Sb);
    // Every cotangent below is A_bar B + B A_bar = P + P^T with ONE full product P = A_bar B (18 MFMAs + a 9-FMA tail
    // at D = 9) and the mirror by lane swaps -- against two upper-triangle products (24 + 12); products that feed the same
    // cotangent are summed (in the accumulators, or by one FMA per element) BEFORE the mirror.
    // ---- back through the squarings: C_bar = {C_bar', C} + {S_bar', S},  S_bar = {S_bar', C} - {C_bar', S} ----
#pragma unroll
    for (int j = SDG_MAXS - 1; j >= 0; --j)
      if (j < ps) {
        RMat Pc, Ps, Pt;
        zero(Pc), zero(Ps), zero(Pt);
        prod(Cb, Cl[j], Pc);
        prod(Sb, Sl[j], Pc);
        prod(Sb, Cl[j], Ps);
        prod(Cb, Sl[j], Pt);
#pragma unroll
        for (int I = 0; I < NB; ++I)
#pragma unroll
          for (int J = 0; J < NB; ++J) Ps[I][J] -= Pt[I][J];
        mirror(Pc, 1.0, Cb);
        mirror(Ps, 1.0, Sb);
      }
    // ---- back through S = Y Sp, Cm = Cm0 + W4 acc, Sp = Sp0 + W4 acs and the powers ----
    RMat Yb2, Spb, W4b, accb, acsb;
    {
      RMat Pa, Pb;
      zero(Pa), zero(Pb);
      prod(Sb, Sp, Pa);
      prod(Sb, Y, Pb);
      mirror(Pa, 1.0, Yb2);  // 2 x sym(S_bar Sp)
      mirror(Pb, 0.5, Spb);  // sym(S_bar Y)
    }
    {
      RMat Pw, Pd, Pf;
      zero(Pw), zero(Pd), zero(Pf);
      prod(Cb, acc, Pw);
      prod(Spb, acs, Pw);
      prod(Cb, MM6 ? W3 : W4, Pd);
      prod(Spb, MM6 ? W3 : W4, Pf);
      mirror(Pw, 0.5, W4b);   // sym(C_bar acc) + sym(Sp_bar acs): the cotangent of the Horner power (W^3 in the degree-6 variant)
      mirror(Pd, 0.5, accb);  // sym(W4 C_bar)
      mirror(Pf, 0.5, acsb);  // sym(W4 Sp_bar)
    }
    RMat W1b, W2b, W3b;
#pragma unroll
    for (int I = 0; I < NB; ++I)
#pragma unroll
      for (int J = sym_j0<D>(I); J < NB; ++J) {
        const double ab = accb[I][J], sb = acsb[I][J], cb = Cb[I][J], pb = Spb[I][J];
        if constexpr (MM6) {
          W1b[I][J] = c3p_mm6_cos[1] * cb + c3p_mm6_sinc[1] * pb + c3p_mm6_cos[4] * ab + c3p_mm6_sinc[4] * sb;
          W2b[I][J] = c3p_mm6_cos[2] * cb + c3p_mm6_sinc[2] * pb + c3p_mm6_cos[5] * ab + c3p_mm6_sinc[5] * sb;
          W3b[I][J] = W4b[I][J] + c3p_mm6_cos[6] * ab + c3p_mm6_sinc[6] * sb;  // (W4b holds the Horner power's cotangent)
        } else {
          W1b[I][J] = c3p_mm8_cos[1] * cb + c3p_mm8_sinc[1] * pb + c3p_mm8_cos[5] * ab + c3p_mm8_sinc[5] * sb;
          W2b[I][J] = c3p_mm8_cos[2] * cb + c3p_mm8_sinc[2] * pb + c3p_mm8_cos[6] * ab + c3p_mm8_sinc[6] * sb;
          W3b[I][J] = c3p_mm8_cos[3] * cb + c3p_mm8_sinc[3] * pb + c3p_mm8_cos[7] * ab + c3p_mm8_sinc[7] * sb;
          W4b[I][J] = W4b[I][J] + c3p_mm8_cos[8] * ab + c3p_mm8_sinc[8] * sb;
        }
      }
    {
      // W4 = W2^2, W3 = W W2:  W2_bar += {W4_bar, W2} + sym(W3_bar W),  W_bar += sym(W3_bar W2)   (degree 6: no W4)
      RMat Pi, Pg, Ph, q, h;
      zero(Pi), zero(Pg), zero(Ph);
      if constexpr (!MM6) prod(W4b, W2, Pi);
      prod(W3b, W1, Pg);
      prod(W3b, W2, Ph);
#pragma unroll
      for (int I = 0; I < NB; ++I)
#pragma unroll
        for (int J = 0; J < NB; ++J) Pi[I][J] = fma(0.5, Pg[I][J], Pi[I][J]);
      mirror(Pi, 1.0, q);
      mirror(Ph, 0.5, h);
#pragma unroll
      for (int I = 0; I < NB; ++I)
#pragma unroll
        for (int J = sym_j0<D>(I); J < NB; ++J) {
          W2b[I][J] += q[I][J];
          W1b[I][J] += h[I][J];
        }
    }
    {
      RMat Pj, q;  // W2 = W^2:  W_bar += {W2_bar, W}
      zero(Pj);
      prod(W2b, W1, Pj);
      mirror(Pj, 1.0, q);
#pragma unroll
      for (int I = 0; I < NB; ++I)
#pragma unroll
        for (int J = sym_j0<D>(I); J < NB; ++J) W1b[I][J] += q[I][J];
    }
    RMat Yb;
    {
      RMat Pk;  // W = Y^2:  Y_bar = sym(S_bar Sp) + {W_bar, Y}
      zero(Pk);
      prod(W1b, Y, Pk);
      mirror(Pk, 1.0, Yb);
#pragma unroll
      for (int I = 0; I < NB; ++I)
#pragma unroll
        for (int J = I; J < NB; ++J) Yb[I][J] = fma(0.5, Yb2[I][J], Yb[I][J]);
    }
    // ---- grad[k] = scale <Y_bar, Y_k> (full-matrix inner product of symmetric matrices) + Re(mu_k conj(tr N)) ----
    for (int k = 0; k < K; ++k) {
      const double* tk = tab + (k + 1) * (MAT + 4);
      double part = 0.0;
#pragma unroll
      for (int I = 0; I < NB; ++I)
#pragma unroll
        for (int J = I; J < NB; ++J) {
          const double yk = -ymask[I] * lds_ld(tk + yo[I] + J * 4);
          part = fma((I == J ? sc : 2.0 * sc) * Yb[I][J], yk, part);
        }
      part = fma(tk[MAT + 0], trr, fma(tk[MAT + 1], tri, part));
      part += __shfl_xor(part, 1);
      part += __shfl_xor(part, 2);
      part += __shfl_xor(part, 16);
      part += __shfl_xor(part, 32);
      if (act && lp.idx16 == 0) A.grad[((long)sample * K + k) * A.N + n0 + t] = part;
    }
    // ---- N <- R conj(dU) = (Rr + i Ri)(C + iS): general left operand from its LDS image, three real products ----
    {
      RMat Ps, Q1, Q2, Q3;
#pragma unroll
      for (int I = 0; I < NB; ++I)
#pragma unroll
        for (int J = 0; J < NB; ++J) {
          Ps[I][J] = Cf[I][J] + Sf[I][J];
          Q1[I][J] = Q2[I][J] = Q3[I][J] = 0.0;
        }
      const double* ir = rimg + (lp.b * 2 + 0) * RIMG;
      const double* ii = rimg + (lp.b * 2 + 1) * RIMG;
#pragma unroll
      for (int Kk = 0; Kk < NB; ++Kk)
#pragma unroll
        for (int I = 0; I < NB; ++I) {
          const double fr = lds_ld(ir + (4 * I + lp.c) * WR + 4 * Kk + lp.r);  // A layout: lane (r, c) <- R[4I + c][4K + r]
          const double fi = lds_ld(ii + (4 * I + lp.c) * WR + 4 * Kk + lp.r);
          const double fs = fr + fi;
#pragma unroll
          for (int J = 0; J < NB; ++J) {
            Q1[I][J] = mfma4(fr, Cf[Kk][J], Q1[I][J]);
            Q2[I][J] = mfma4(fi, Sf[Kk][J], Q2[I][J]);
            Q3[I][J] = mfma4(fs, Ps[Kk][J], Q3[I][J]);
          }
        }
#pragma unroll
      for (int I = 0; I < NB; ++I)
#pragma unroll
        for (int J = 0; J < NB; ++J) {
          Nr[I][J] = Q1[I][J] - Q2[I][J];
          Ni[I][J] = (Q3[I][J] - Q1[I][J]) - Q2[I][J];
        }
    }
  }
  };
  if (small_var)
    sweep(std::true_type{});
  else
    sweep(std::false_type{});
}

template <int D>
hipError_t launch_grad_real_t(const SmallGradArgs& A, hipStream_t st) {
  using C = SD<D>;
  constexpr int NB = RD<D>::NB, RIMG = 4 * NB * (4 * NB + 1);
  const long nchains = (long)A.B * A.S;
  const unsigned grid = (unsigned)((nchains + 3) / 4);
  const size_t lds = (size_t)((1 + A.K) * (C::MAT + 4) + 8 * RIMG + 4 * A.K * A.Lmax) * sizeof(double);
  if (lds > 60 * 1024) return hipErrorInvalidValue;
  C3P_LAUNCH(smalld_grad_real_kernel<D>, dim3(grid), dim3(64), lds, st, A);
  return hipGetLastError();
}

template <int D>
hipError_t launch_grad_t(const SmallGradArgs& A, hipStream_t st) {
  using C = SD<D>;
  const long nchains = (long)A.B * A.S;
  const unsigned grid = (unsigned)((nchains + 3) / 4);
  const size_t lds = (size_t)((1 + A.K) * (C::MAT + 4) + 8 * C::MAT + 4 * A.K * A.Lmax) * sizeof(double);
  if (lds > 60 * 1024) return hipErrorInvalidValue;
  C3P_LAUNCH(smalld_grad_kernel<D>, dim3(grid), dim3(64), lds, st, A);
  return hipGetLastError();
}

template <int D>
hipError_t launch_grad_general_t(const SmallGradArgs& A, hipStream_t st) {
  using C = SD<D>;
  const long nchains = (long)A.B * A.S;
  const unsigned grid = (unsigned)((nchains + 3) / 4);
  const size_t lds = (size_t)(2 * (1 + A.K) * (C::MAT + 4) + 8 * C::MAT + 4 * A.K * A.Lmax) * sizeof(double);
  if (lds > 60 * 1024) return hipErrorInvalidValue;
  if (A.hs != nullptr)
    C3P_LAUNCH((smalld_grad_general_kernel<D, true>), dim3(grid), dim3(64), lds, st, A);
  else
    C3P_LAUNCH((smalld_grad_general_kernel<D, false>), dim3(grid), dim3(64), lds, st, A);
  return hipGetLastError();
}

template <int D>
hipError_t launch_prep_t(const PrepArgs& P, int nsamp, hipStream_t st) {
  C3P_LAUNCH(smalld_prep_kernel<D>, dim3((unsigned)(nsamp * (1 + P.K))), dim3(64), 0, st, P);
  return hipGetLastError();
}

}  // namespace

#define SD_DISPATCH(FN, ...)                                   \
  switch (Dm) {                                                \
    case 2: return FN<2>(__VA_ARGS__);                         \
    case 3: return FN<3>(__VA_ARGS__);                         \
    case 4: return FN<4>(__VA_ARGS__);                         \
    case 5: return FN<5>(__VA_ARGS__);                         \
    case 6: return FN<6>(__VA_ARGS__);                         \
    case 7: return FN<7>(__VA_ARGS__);                         \
    case 8: return FN<8>(__VA_ARGS__);                         \
    case 9: return FN<9>(__VA_ARGS__);                         \
    case 10: return FN<10>(__VA_ARGS__);                       \
    case 11: return FN<11>(__VA_ARGS__);                       \
    case 12: return FN<12>(__VA_ARGS__);                       \
    default: return hipErrorInvalidValue;                      \
  }

#if C3P_SMALLD_HAS(2)
hipError_t c3p_launch_smalld_grad_real(const SmallGradArgs& A, hipStream_t st) {
  const int Dm = A.Dm;
  SD_DISPATCH(launch_grad_real_t, A, st)
}
#endif

#if C3P_SMALLD_HAS(1)
int c3p_smalld_mat_doubles(int Dm) {
  const int NBI = (Dm + 1) / 2, NJ = (Dm + 3) / 4;
  return 4 * NBI * (4 * NJ + 1);
}

int c3p_smalld_img_doubles(int Dm) {
  const int NBI = (Dm + 1) / 2, NJ = (Dm + 3) / 4;
  const int mat = 4 * NBI * (4 * NJ + 1);
  return ((mat - 12 + 31) / 32) * 32 + 12;
}

size_t c3p_smalld_table_doubles(int Dm, int K) { return (size_t)(1 + K) * (c3p_smalld_mat_doubles(Dm) + 4); }

bool c3p_smalld_supported(int Dm) { return Dm >= 2 && Dm <= C3P_SMALLD_MAX; }

hipError_t c3p_launch_smalld_chain(const SmallArgs& A, hipStream_t st) {
  const int Dm = A.Dm;
  SD_DISPATCH(launch_chain_t, A, st)
}

hipError_t c3p_launch_smalld_grad(const SmallGradArgs& A_, hipStream_t st) {
  const int Dm = A_.Dm;
  SmallGradArgs A = A_;
  A.skip_real = 0;
  // real Hamiltonians first (reverse mode through the cos / sin evaluation), then the general sweep for the rest; the
  // per-slice generator cotangents (zout) only exist in the general sweep
  if (A.zout == nullptr && !c3p_opt_on(C3P_OPT_no_real_grad)) {
    const hipError_t e = c3p_launch_smalld_grad_real(A, st);
    if (e != hipSuccess) return e;
    A.skip_real = 1;
  }
  SD_DISPATCH(launch_grad_t, A, st)
}

hipError_t c3p_launch_smalld_grad_general(const SmallGradArgs& A, hipStream_t st) {
  const int Dm = A.Dm;
  SD_DISPATCH(launch_grad_general_t, A, st)
}

hipError_t c3p_launch_smalld_prep(const PrepArgs& P, int Dm, int nsamp, hipStream_t st) {
  SD_DISPATCH(launch_prep_t, P, nsamp, st)
}
#endif
