// Mid-D propagator chains on the f64 matrix cores (13 <= D <= 40: cfg3 D=27, cfg5 D=36,
// the excitation-cut D=14/24 fixtures).
//
// One workgroup of 4 wavefronts (one per SIMD) owns one (sample, time-segment) chain.
// Every matrix of the chain lives in LDS as a "half image" Zh of the real 2x2
// representation (rows 2i+p = Re/Im of row i, D columns; see c3p_smalld.hip):
//     R(A) * Bh = Ch ,   C = A B complex,
// computed with v_mfma_f64_4x4x4_4b_f64.  The four blocks of an instruction are four
// consecutive 4-row blocks of the output ("I-group", 16 real rows) times one 4-column block:
//   A operand  : lane (r,b,c) <- R(A)[16 Ig + 4b + c][4K + r]   (16x4 slab, sign-fixed on read)
//   B operand  : lane (r,b,c) <- Bh[4K + r][4J + c]             (4x4 block, LDS-broadcast over b)
//   acc (tile) : lane (r,b,c) -> Ch[16 Ig + 4b + r][4J + c]
// The NIG x NJ output tiles are dealt to the 4 waves as contiguous runs of the flat index
// t = J*NIG + Ig, compile-time per wave, so each wave loads every A slab it needs once per
// K step and at most 3-4 B blocks: ~0.6 LDS reads per MFMA.
// Row stride W (doubles) = 4 NJ + 1, ODD: a 64-bit LDS access is served 16 lanes per cycle (32 banks x 4 B), and the 16
// lanes of a group of an A-slab read hold 16 different rows of one column -- i W mod 16 must be distinct over 16 rows
// (round 2: the even strides 4 NJ + 2 of round 1, chosen for a 64-bank model, made every A-slab read 2-way conflicted).
//
// Per slice: X (registers, from pre-shifted global tables) -> image; X^2, X^3, X^4; Horner in
// X^4; squarings; U <- E U.  Every wave keeps ITS tiles of X, X^2, X^3, P and U in registers
// (the block polynomials B_j are lane-local), so only three LDS images are live at any time:
// buf0 = X then P/E, buf1 = X^2 then X^3 then U, buf2 = X^4  (73 KB at D = 36 -> two
// workgroups = two waves per SIMD per CU).  Same plan logic as the small-D kernel.
#include <cstdlib>
#include <type_traits>

#include "c3p_common.h"
#include "c3p_kernels.h"
#include "c3p_midd.h"
#ifndef C3P_MDR_BIG_WGS
#define C3P_MDR_BIG_WGS 2  // workgroups per CU of the 48-row real classes (D = 33..40): 2 = 256 registers per wave
#endif

// The file is compiled as up to three translation units (__graft_entry__.build passes -DC3P_MIDD_PART=1|2|3: complex
// chain kernels + tables, real-Hamiltonian instance, gradient sweep) so that they build concurrently; without the
// macro everything lands in one unit.
#ifdef C3P_MIDD_PART
#define C3P_MIDD_HAS(p) (C3P_MIDD_PART == (p))
#else
#define C3P_MIDD_HAS(p) 1
#endif

// 32 + 4 row split of the 48-row real class with 9 column blocks (D = 33..36, cfg5) in the real-Hamiltonian instances: the
// forward chain kernel (translation unit 2) and the real backward sweep (unit 3), whose products all go through mm_real.
// -DC3P_MDR_NO_ROWSPLIT builds the padded deal for A/B (tools/ab_build.sh).
#if !defined(C3P_MDR_NO_ROWSPLIT)
#define C3P_MDR_ROWSPLIT 1
#else
#define C3P_MDR_ROWSPLIT 0
#endif
// Pinwheel deal of the 32-row real class with 7 column blocks (D = 25..28, cfg3): the 7 x 7 grid of 4 x 4 output blocks is
// cut into four 3 x 4 / 4 x 3 rectangles around the centre block, one rectangle per wave, every block on
// v_mfma_f64_4x4x4_4b (see Sched::PW).  -DC3P_MDR_NO_PINWHEEL builds the four padded 16 x 16 units for A/B.
#ifndef C3P_PW_PFP
#define C3P_PW_PFP 1  // K-step PAIRS the operand fetch of the pinwheel products runs ahead
#endif
#ifndef C3P_MMR_ILV
#define C3P_MMR_ILV 2  // mm_real of the 48-row real classes: reads per matrix instruction in the interleaved issue order (0 = grouped)
#endif
#ifndef C3P_MMR_PRIO
#define C3P_MMR_PRIO 0  // s_setprio inside the K loops of mm_real (the classes other than the pinwheel one)
#endif
#ifndef C3P_PW_SPLITK
#define C3P_PW_SPLITK 0  // single pinwheel products: odd K-steps in a second accumulator set (measured: -1.4 %)
#endif
#ifndef C3P_PW_ROT
#define C3P_PW_ROT 0  // wave roles of the pinwheel class rotated by blockIdx % this (0 = off)
#endif
#ifndef C3P_PW_PRIO
#define C3P_PW_PRIO 2  // s_setprio inside the K loops of the pinwheel products
#endif
#ifndef C3P_PW_ILV
#define C3P_PW_ILV 1  // operand reads of the next K-step pair interleaved with the matrix instructions of this one
#endif
#ifndef C3P_PW_CPL
#define C3P_PW_CPL 1  // K-step pair whose matrix instructions the centre block's operand reads go out with
#define C3P_PW_CPF 2  // ... and the pair its two K-packed instructions follow
#endif
#ifndef C3P_PW_TIMING
#define C3P_PW_TIMING 0  // 1: timing-only build (WRONG results): one 4-column fragment per K-step instead of three
#endif
#ifndef C3P_PW_WGS
#define C3P_PW_WGS 3  // workgroups per CU of the pinwheel class (forward kernel)
#endif
#if !defined(C3P_MDR_NO_PINWHEEL)
#define C3P_MDR_PINWHEEL 1
#else
#define C3P_MDR_PINWHEEL 0
#endif
extern __shared__ __attribute__((aligned(16))) double c3p_md_lds[];

namespace {

constexpr int NW = 4;  // wavefronts per workgroup
constexpr int SGC = 64;  // slices of control amplitudes staged in LDS at a time (chain kernel)

__device__ __forceinline__ double md_mfma4(double a, double b, double c) {
  return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0);
}
// c - a b (the A operand negated by the instruction's neg modifier: blgp bit 0 on the f64 forms)
__device__ __forceinline__ double md_mfma4n(double a, double b, double c) {
  return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 1);
}
__device__ __forceinline__ double md_flip(double v, unsigned mask_hi) {
  unsigned long long u = __double_as_longlong(v);
  u ^= ((unsigned long long)mask_hi) << 32;
  return __longlong_as_double(u);
}
__device__ __forceinline__ double md_rfl(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_readfirstlane(lo);
  hi = __builtin_amdgcn_readfirstlane(hi);
  return __hiloint2double(hi, lo);
}

typedef double d4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ d4 md_mfma16(double a, double b, d4 c) {
  return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ d4 md_mfma16n(double a, double b, d4 c) {  // c - a b (neg modifier on A)
  return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 1);
}

template <int NIG, int NJ>
struct MD {
  static constexpr int ROWS = 16 * NIG;
};

// Compile-time deal of the output tiles to the four wavefronts.
//  * "big" units: 16 x 16 tiles (one v_mfma_f64_16x16x4_f64, 64 cycles) = four adjacent column
//    blocks of one 16-row group; one instruction per K-step instead of four, and one
//    non-replicated B read instead of four broadcast ones.
//  * "small" units: the NJ % 4 leftover 16 x 4 tiles (v_mfma_f64_4x4x4_4b_f64, 16 cycles).
// Bigs are dealt round-robin (cost 4 each), smalls greedily to the least loaded wave.
template <int NIG, int NJ>
struct Sched {
  // The 32-row real class with 7 column blocks (D = 25..28, cfg3: only the real instance has NIG = 2 there) pads the three
  // leftover column blocks to a second 16-column unit (the 32-wide image has the zero columns): FOUR 16 x 16 units, one per
  // wave, instead of two plus six small ones -- the small units cost two waves five LDS reads per K-step for 48 matrix-pipe
  // cycles and their 16 x 4 stores are 8-way bank-conflicted; the kernel shares the LDS pipe of a CU between three
  // workgroups and was as LDS-bound (63 % busy) as MFMA-bound (62 %).  cfg3 +9 %; with 6 column blocks (D = 21..24) the
  // padding costs more matrix-pipe time than it saves LDS time (-4 %).
  // Round 5, the same class (PW): the padded deal multiplies 32 x 32 x 28 for a 27 x 27 x 27 product (tile utilisation 0.686).
  // Pinwheel: the 28 x 28 output is 7 x 7 blocks of 4 x 4; wave 0 owns block rows 0..3 x block columns 0..2 (three "tall"
  // 16 x 4 units), wave 1 block rows 0..2 x block columns 3..6 (three "wide" 4 x 16 units at column 12) plus the centre
  // block (3, 3), wave 2 block rows 3..6 x block columns 4..6 (tall units at row 12), wave 3 block rows 4..6 x block
  // columns 0..3 (wide units).  Every unit is ONE v_mfma_f64_4x4x4_4b per K-step: 3 x 16 matrix-pipe cycles per wave and
  // K-step instead of 64; the centre block is K-packed (the four blocks of an instruction are four K-steps of the SAME
  // output block, summed across the lane quads afterwards): 2 instructions per product instead of 7.  Per product
  // 21 (+ 2) instructions of 16 cycles against 7 of 64; tile utilisation 27^3 / (4 * 21 + 2) / 256 = 0.894.
  static constexpr bool PW = C3P_MDR_PINWHEEL != 0 && NIG == 2 && NJ == 7;
  static constexpr bool PAD = !PW && NIG == 2 && NJ == 7;
  static constexpr int NB16 = PAD ? 2 : (PW ? 0 : NJ / 4);
  static constexpr int JR = (PAD || PW) ? 0 : NJ % 4;
  // The 48-row real class with 9 column blocks (D = 33..36, cfg5; real instance only): six 16 x 16 units and three 16 x 4
  // ones deal out as 128 / 128 / 96 / 80 matrix-pipe cycles per K-step; with the last NSPLIT = 2 big units dealt as four
  // small ones each it is 112 / 112 / 112 / 96.  (Splitting wherever it lowers the maximum was measured: +2 - 3 % here,
  // 20 - 27 % SLOWER at D <= 24, where small units mean four broadcast B reads and four instructions instead of one.)
  // Round 4, the same class in the real forward instance (RS): its third row group holds 4 of 16 rows (36 = 32 + 4), so the two
  // 16 x 16 units there are dealt as "wide" units -- ONE v_mfma_f64_4x4x4_4b per K-step (16 cycles instead of 64) whose four
  // blocks are the four COLUMN blocks of the unit and whose A fragment is rows 32..35 broadcast to all blocks (one more,
  // conflict-free, LDS read); the result lands in accumulator component 0 of the 16 x 16 register layout, components 1..3 are
  // the zero padding rows 36..47.  Deal: one full unit per wave, the wide units to waves 2 / 3, the 16 x 4 units of column
  // block 8 to waves 0 / 1 / 2: 80 / 80 / 96 / 80 matrix-pipe cycles per K-step instead of 112 / 112 / 112 / 96.  Upper bound
  // measured first with a timing-only build (161.5 -> 141.7 ms per 1024-sample cfg5 batch).
  static constexpr bool RS = C3P_MDR_ROWSPLIT != 0 && NIG == 3 && NJ == 9;
  static constexpr int NSPLIT = (!RS && NIG == 3 && NJ == 9) ? 2 : 0;
  static constexpr int NBIG = NIG * NB16 - NSPLIT;
  static constexpr int NSMALL = NIG * JR + 4 * NSPLIT;
  static constexpr int nbig(int w) {
    if (RS) return w < 2 ? 1 : 2;
    return w < NBIG ? (NBIG - w + NW - 1) / NW : 0;
  }
  // small unit us -> (column block J, row group Ig): first the leftover column blocks, then the split big units
  static constexpr int small_j(int us) {
    if (us < NIG * JR) return 4 * NB16 + us / NIG;
    const int k = us - NIG * JR;
    return 4 * ((NBIG + k / 4) / NIG) + k % 4;
  }
  static constexpr int small_ig(int us) {
    if (us < NIG * JR) return us % NIG;
    return (NBIG + (us - NIG * JR) / 4) % NIG;
  }
  static constexpr int small_owner(int us) {
    if (RS) return us;  // (row group us of column block 8)
    int load[NW] = {4 * nbig(0), 4 * nbig(1), 4 * nbig(2), 4 * nbig(3)};
    int owner = 0;
    for (int u = 0; u <= us; ++u) {
      int best = 0;
      for (int w = 1; w < NW; ++w)
        if (load[w] < load[best]) best = w;
      load[best] += 1;
      owner = best;
    }
    return owner;
  }
  static constexpr int nsmall(int w) {
    int n = 0;
    for (int u = 0; u < NSMALL; ++u)
      if (small_owner(u) == w) ++n;
    return n;
  }
  static constexpr int small_us(int w, int i) {
    int n = 0;
    for (int u = 0; u < NSMALL; ++u)
      if (small_owner(u) == w) {
        if (n == i) return u;
        ++n;
      }
    return 0;
  }
};

// The units of wave WV and the geometry of every register element it owns.
template <int NIG, int NJ, int W, int WV>
struct WaveTiles {
  using S = Sched<NIG, NJ>;
  static constexpr int NBW = S::PW ? 0 : S::nbig(WV);
  static constexpr int NSW = S::PW ? (WV == 1 ? 4 : 3) : S::nsmall(WV);
  static constexpr int NE = 4 * NBW + NSW;  // doubles per matrix held by each lane of this wave
  // pinwheel deal (S::PW): element e < 3 is a tall unit (waves 0 / 2: rows R0 + 4b + r, column C0 + c) or a wide unit
  // (waves 1 / 3: row R0 + r, columns C0 + 4b + c); element 3 of wave 1 is the centre block (rows 12 + r, columns 12 + c),
  // valid in the lanes of quad b = 0 -- the other quads hold copies that are parked at never-read positions of the zero
  // rows 28..31 (columns 4 (b - 1) + c), outside the matrix for every mask.  The units at row / column 12 (waves 2 / 1)
  // put quad b on block (b + 1) & 3 of their 16 rows / columns (see md_pw_off).
  static constexpr bool pw_tall = (WV & 1) == 0;
  static constexpr int pw_r0(int e) { return WV == 0 ? 0 : WV == 2 ? 12 : WV == 1 ? 4 * e : 16 + 4 * e; }
  static constexpr int pw_c0(int e) { return WV == 0 ? 4 * e : WV == 2 ? 16 + 4 * e : WV == 1 ? 12 : 0; }
  // row / column of element e in lane (r, b, c)
  static __device__ __forceinline__ int lrow(int e, int r, int b, int c) {
    if (S::PW) {
      if (e == 3) return b == 0 ? 12 + r : 28 + r;
      return pw_r0(e) + (pw_tall ? 4 * (WV == 2 ? (b + 1) & 3 : b) + r : r);
    }
    return row0(e) + (is_big(e) ? r : 4 * b + r);
  }
  static __device__ __forceinline__ int lcol(int e, int r, int b, int c) {
    if (S::PW) {
      if (e == 3) return b == 0 ? 12 + c : 4 * (b - 1) + c;
      return pw_c0(e) + (pw_tall ? c : 4 * (WV == 1 ? (b + 1) & 3 : b) + c);
    }
    return col0(e) + (is_big(e) ? 4 * b + c : c);
  }
  // big unit i: ub = WV + 4 i -> (Jg, Ig)
  // (RS deal: wave 0 (0,0); wave 1 (1,0); wave 2 (0,1) + wide (2,0); wave 3 (1,1) + wide (2,1))
  static constexpr int bIg(int i) {
    if (S::RS) return WV < 2 ? WV : (i == 0 ? WV - 2 : 2);
    return (WV + NW * i) % NIG;
  }
  static constexpr int bJg(int i) {
    if (S::RS) return WV < 2 ? 0 : (i == 0 ? 1 : WV - 2);
    return (WV + NW * i) / NIG;
  }
  static constexpr bool is_wide(int i) { return S::RS && bIg(i) == NIG - 1; }
  static constexpr bool has_wide() {
    for (int i = 0; i < NBW; ++i)
      if (is_wide(i)) return true;
    return false;
  }
  // small unit i: us -> (J, Ig)
  static constexpr int sIg(int i) { return S::small_ig(S::small_us(WV, i)); }
  static constexpr int sJ(int i) { return S::small_j(S::small_us(WV, i)); }
  // element e: big tile i = e/4, accumulator register q = e%4 (rows 4q + r of the 16-row group),
  //            or small tile e - 4 NBW
  static constexpr bool is_big(int e) { return e < 4 * NBW; }
  static constexpr int row0(int e) { return is_big(e) ? 16 * bIg(e / 4) + 4 * (e % 4) : 16 * sIg(e - 4 * NBW); }
  static constexpr int col0(int e) { return is_big(e) ? 16 * bJg(e / 4) : 4 * sJ(e - 4 * NBW); }
  static constexpr int off0(int e) { return row0(e) * W + col0(e); }
  // which A slabs / B blocks does the wave read in a K-step
  static constexpr bool uses_ig(int Ig) {
    for (int i = 0; i < NBW; ++i)
      if (bIg(i) == Ig) return true;
    for (int i = 0; i < NSW; ++i)
      if (sIg(i) == Ig) return true;
    return false;
  }
  static constexpr bool uses_jg(int Jg) {
    for (int i = 0; i < NBW; ++i)
      if (bJg(i) == Jg) return true;
    return false;
  }
  static constexpr bool uses_j(int J) {
    for (int i = 0; i < NSW; ++i)
      if (sJ(i) == J) return true;
    return false;
  }
};

// Register tile set of one matrix for wave WV.
template <int NBW, int NSW>
struct TileRegs {
  d4 big[NBW > 0 ? NBW : 1];
  double sm[NSW > 0 ? NSW : 1];
  __device__ __forceinline__ double get(int e) const { return e < 4 * NBW ? big[e / 4][e % 4] : sm[e - 4 * NBW]; }
  __device__ __forceinline__ void set(int e, double v) {
    if (e < 4 * NBW)
      big[e / 4][e % 4] = v;
    else
      sm[e - 4 * NBW] = v;
  }
};

struct MidCommon {
  int lane, r, b, c;
  int D, nbk, K;
  int sample, n0, len;
  int aoff, boff, lbig, lsmall;
  int aoffR, nbkR, realH;  // real-Hamiltonian path: A-slab offset of a REAL image, its K-steps, the flag
  // real images: lane offsets of a 16x16 unit element / a 16x4 unit element / a 4x4 B block; index 0 for columns
  // 0..15, 1 for columns 16..31 (they differ only in the swizzled layout)
  int rbig[2], rsm[2], rblk[2];
  int rb12, rctr;  // pinwheel deal: 16 columns from column 12 of row r; the centre block's K-packed fragments (row 4b + r, column 12 + c)
  unsigned negmask;
  int pr, ps, t18;
  int t18n;  // T18 with the economised parameters for normal generators (c3p_t18_tab row 1)
  double scale;
  const double* tabs;
  double *buf0, *buf1, *buf2, *sg;
};

// acc += (A image) * (B image) for the units of wave WV
template <int NIG, int NJ, int W, int WV>
__device__ __forceinline__ void mm_tiles(const double* imgA, const double* imgB, const MidCommon& cm,
                                         TileRegs<WaveTiles<NIG, NJ, W, WV>::NBW, WaveTiles<NIG, NJ, W, WV>::NSW>& acc) {
  using T = WaveTiles<NIG, NJ, W, WV>;
  using S = Sched<NIG, NJ>;
  constexpr int NB16 = S::NB16 > 0 ? S::NB16 : 1;
  constexpr int JR = NJ;  // small B blocks are indexed by their column block J (only the ones the wave uses are loaded)
  // software pipelined over K: operands of step K+1 are in flight while step K's MFMAs issue
  double a0[NIG], a1[NIG], g0[NB16], g1[NB16], s0[JR], s1[JR];
  const double* pa = imgA + cm.aoff;
  const double* pg = imgB + cm.lbig;     // row r, columns 4b + c of a 16-column group
  const double* ps4 = imgB + cm.boff;    // row r, column c of a 4-column block (broadcast over b)
  auto load = [&](double (&a)[NIG], double (&g)[NB16], double (&sb)[JR], int K) {
#pragma unroll
    for (int Ig = 0; Ig < NIG; ++Ig) a[Ig] = T::uses_ig(Ig) ? md_flip(pa[Ig * 16 * W + 2 * K], cm.negmask) : 0.0;
#pragma unroll
    for (int Jg = 0; Jg < S::NB16; ++Jg) g[Jg] = T::uses_jg(Jg) ? pg[K * 4 * W + 16 * Jg] : 0.0;
#pragma unroll
    for (int J = 0; J < NJ; ++J) sb[J] = T::uses_j(J) ? ps4[K * 4 * W + 4 * J] : 0.0;
  };
  auto fmas = [&](const double (&a)[NIG], const double (&g)[NB16], const double (&sb)[JR]) {
#pragma unroll
    for (int i = 0; i < T::NBW; ++i) acc.big[i] = md_mfma16(a[T::bIg(i)], g[T::bJg(i)], acc.big[i]);
#pragma unroll
    for (int i = 0; i < T::NSW; ++i) acc.sm[i] = md_mfma4(a[T::sIg(i)], sb[T::sJ(i)], acc.sm[i]);
  };
  const int nbk = cm.nbk;
  load(a0, g0, s0, 0);
  for (int K = 0; K < nbk; K += 2) {
    const int K1 = (K + 1 < nbk) ? K + 1 : K;
    load(a1, g1, s1, K1);
    __builtin_amdgcn_sched_barrier(0);  // keep the prefetch ahead of the MFMA group
    fmas(a0, g0, s0);
    __builtin_amdgcn_sched_barrier(0);
    const int K2 = (K + 2 < nbk) ? K + 2 : K;
    load(a0, g0, s0, K2);
    __builtin_amdgcn_sched_barrier(0);
    if (K + 1 < nbk) fmas(a1, g1, s1);
    __builtin_amdgcn_sched_barrier(0);
  }
}

template <int I>
using IC = std::integral_constant<int, I>;

// compile-time loop: f(integral_constant<int, I>) for I in [B, E)
template <int B, int E, typename F>
__device__ __forceinline__ void md_unroll(F&& f) {
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    md_unroll<B + 1, E>(f);
  }
}

// mm_tiles with the K loop fully unrolled (2 NJ K-steps; the last one is skipped when 2 D <= 8 NJ - 4) and the operands
// fetched PF steps ahead in a ring of register stages -- for kernels that run ONE wave per SIMD (the gradient
// sweep), where nothing else hides the LDS round trip behind a one-step prefetch.
template <int NIG, int NJ, int W, int WV, int PF>
__device__ __forceinline__ void mm_tiles_pf(const double* imgA, const double* imgB, const MidCommon& cm,
                                            TileRegs<WaveTiles<NIG, NJ, W, WV>::NBW, WaveTiles<NIG, NJ, W, WV>::NSW>& acc) {
  using T = WaveTiles<NIG, NJ, W, WV>;
  using S = Sched<NIG, NJ>;
  constexpr int NB16 = S::NB16 > 0 ? S::NB16 : 1;
  constexpr int JR = NJ;  // small B blocks by column block J
  constexpr int NK = 2 * NJ, NS = PF + 1;
  double a[NS][NIG], g[NS][NB16], sb[NS][JR];
  const double* pa = imgA + cm.aoff;
  const double* pg = imgB + cm.lbig;
  const double* ps4 = imgB + cm.boff;
  const bool tail = NK - 1 < cm.nbk;  // is the last K-step inside the matrix?
#define C3P_MMT_LOAD(K)                                                                                             \
  {                                                                                                                 \
    constexpr int st_ = (K) % NS;                                                                                   \
    _Pragma("unroll") for (int Ig = 0; Ig < NIG; ++Ig)                                                              \
        a[st_][Ig] = T::uses_ig(Ig) ? md_flip(pa[Ig * 16 * W + 2 * (K)], cm.negmask) : 0.0;                         \
    _Pragma("unroll") for (int Jg = 0; Jg < S::NB16; ++Jg) g[st_][Jg] = T::uses_jg(Jg) ? pg[(K) * 4 * W + 16 * Jg] : 0.0; \
    _Pragma("unroll") for (int J = 0; J < NJ; ++J)                                                                  \
        sb[st_][J] = T::uses_j(J) ? ps4[(K) * 4 * W + 4 * J] : 0.0;                                                 \
  }
#define C3P_MMT_FMAS(K)                                                                                             \
  {                                                                                                                 \
    constexpr int st_ = (K) % NS;                                                                                   \
    _Pragma("unroll") for (int i = 0; i < T::NBW; ++i)                                                              \
        acc.big[i] = md_mfma16(a[st_][T::bIg(i)], g[st_][T::bJg(i)], acc.big[i]);                                   \
    _Pragma("unroll") for (int i = 0; i < T::NSW; ++i)                                                              \
        acc.sm[i] = md_mfma4(a[st_][T::sIg(i)], sb[st_][T::sJ(i)], acc.sm[i]);                                      \
  }
  md_unroll<0, PF>([&](auto Kc) { constexpr int K = decltype(Kc)::value; C3P_MMT_LOAD(K) });
  md_unroll<0, NK>([&](auto Kc) {
    constexpr int K = decltype(Kc)::value;
    if constexpr (K + PF < NK) C3P_MMT_LOAD(K + PF)
    __builtin_amdgcn_sched_barrier(0);
    if (K < NK - 1 || tail) C3P_MMT_FMAS(K)
    __builtin_amdgcn_sched_barrier(0);
  });
#undef C3P_MMT_LOAD
#undef C3P_MMT_FMAS
}

// Workgroup barrier that waits for the LDS traffic only (__syncthreads also drains the vector-memory counter,
// i.e. stalls on outstanding global stores of partial propagators).
__device__ __forceinline__ void md_bar() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Stage the control amplitudes of slices [t0, t0 + SGC) of the chain's segment in LDS (all four waves call this
// at the same t0).  Readers of the previous chunk are at least one barrier behind its last use.
template <int WV>
__device__ __forceinline__ void md_stage_signals(const MidArgs& A, const MidCommon& cm, int t0) {
  const int tid = WV * 64 + cm.lane;
  __syncthreads();
  for (int i = tid; i < cm.K * SGC; i += 256) {
    const int k = i / SGC, j = i - k * SGC;
    if (t0 + j < cm.len) cm.sg[i] = A.signals[((long)cm.sample * cm.K + k) * A.N + cm.n0 + t0 + j];
  }
  __syncthreads();
}

// ---------------------------------------------------------------------------------------------
// Real-Hamiltonian path (unitary mode, every table purely imaginary: H real).  X = -iY with Y real, so
//   exp(X) = cos Y - i sin Y,  cos Y = p_c(W),  sin Y = Y p_s(W),  W = Y^2  (Taylor degree 18 / 17, the
// T18 scaling rule), all in REAL D x D products: a real image has 16 NIGR rows (NIGR = ceil(NIG / 2)) and
// a real product runs ceil(D / 4) K-steps -- a quarter of the complex product's MFMA work.  Per slice
// 8 + 2 s + 4 real products (W, W^2, W^3, two paired Horner steps in W^3, Y p_s; squarings
// cos 2Y = 2 C^2 - I, sin 2Y = 2 S C; chain Ur' = C Ur + S Ui, Ui' = C Ui - S Ur) instead of
// (5 + s + 1) complex = 24 + 4 s real-equivalent ones.  Products that share an operand are issued
// together (MODE 1: shared left operand, MODE 2: shared right operand) and share its LDS reads.
// ---------------------------------------------------------------------------------------------
// v rotated by N lanes inside each row of 16 lanes (DPP row_ror)
template <int N>
__device__ __forceinline__ double md_row_ror(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, 0x120 + N, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(0, hi, 0x120 + N, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}

// Image layout of the pinwheel class (Sched::PW): 32 x 32 doubles, rows paired so that ONE ds_read_b128 fetches a
// fragment of two K-steps.  Element (row, col) lives in "pair row" p = 4 (row >> 3) + (row & 3), half h = (row >> 2) & 1
// (K-steps 2P and 2P + 1 = row blocks 2P, 2P + 1 share the pair rows 4P..4P + 3), 16-byte chunk col ^ X(p) with
// X(p) = 12 (p & 1) ^ 4 ((p >> 2) & 3):  double index (32 p + (col ^ X(p))) 2 + h.
//  * 12 (p & 1): a 128-bit read is served in lane groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ... -- quads 0 / 3 of pair row
//    4P + r and quads 1 / 2 of 4P + r + 1; chunk c and c + 16 share banks.  16-column fragments put chunk block b (mod 4) in quad
//    b, so the XOR (which swaps blocks 1 <-> 2, 0 <-> 3) keeps them apart; 4-column fragments (all quads one block) need it.
//  * 4 ((p >> 2) & 3) = 4 (P & 3): spreads the four row blocks of a tall unit's 64-bit stores over the banks.
__device__ __forceinline__ int md_pw_off(int row, int col) {
  const int p = 4 * (row >> 3) + (row & 3), h = (row >> 2) & 1;
  return (32 * p + (col ^ (12 * (p & 1)) ^ (4 * ((p >> 2) & 3)))) * 2 + h;
}

// mm_real for the pinwheel deal (Sched::PW; 7 K-steps = 4 K-step pairs).  Per PAIR a wave reads ONE 16-column fragment
// (pair row 4P + r, columns X + 4 q(b) + c: the left operand's rows for a tall wave, the right operand's columns for a wide
// one) and THREE 4-column fragments (column X' + c, the same in all four lane quads: the right operand's column blocks of a
// tall wave, the left operand's row blocks of a wide one), 128 bits each, and issues 2 x 3 v_mfma_f64_4x4x4_4b (the second
// K-step of the last pair is the zero rows 28..31: skipped).  Left operands are symmetric (or stored transposed), read as
// M[k][i] like everywhere on this path.
template <int WV, int MODE, int IA1, int IA2, int IB1, int IB2, typename Regs>
__device__ __forceinline__ void mm_real_pw(const MidCommon& cm, Regs& acc1, Regs& acc2) {
  typedef double d2 __attribute__((ext_vector_type(2)));
  constexpr int NJ = 7, NP = 4, IMGR = 16 * 2 * 32;
  // MODE 3 / 4: the four products of a complex-by-complex step in real blocks, one pass over the operands (two left, two
  // right fragments sets per K-step pair instead of two passes with one + two):
  //   3:  acc1 += A1 B1 + A2 B2,  acc2 += A1 B2 - A2 B1     (Ur' = C Ur + S Ui, Ui' = C Ui - S Ur)
  //   4:  acc1 += A1 B1 - A2 B2,  acc2 += A1 B2 + A2 B1     (Nr' = Rr C - Ri S, Ni' = Rr S + Ri C)
  // (the minus sign is the neg modifier of the matrix instruction)
  constexpr bool QUAD = MODE == 3 || MODE == 4;
  constexpr bool TWOA = MODE == 2 || QUAD, TWOB = MODE == 1 || QUAD;
  constexpr bool TALL = (WV & 1) == 0;
  constexpr int PFP = C3P_PW_PFP, NSP = PFP + 1;
  // 16-column fragment: waves 0 / 3 columns 4b + c; waves 1 / 2 the unit at 12 with quad b on columns 12 + 4 ((b + 1) & 3) + c
  // (chunk block b mod 4 in quad b); one lane base per P & 3
  const int cf = (WV == 0 || WV == 3) ? 4 * cm.b + cm.c : 12 + 4 * ((cm.b + 1) & 3) + cm.c;
  const int xr = 12 * (cm.r & 1);
  const double* qfm[4];
#pragma unroll
  for (int m = 0; m < 4; ++m) qfm[m] = c3p_md_lds + (32 * cm.r + (cf ^ xr ^ (4 * m))) * 2;
  // 4-column fragments: column block j of the aligned group X4 (compile-time j = block ^ P): one lane base per block
  constexpr int X4 = WV < 2 ? 0 : 16;
  const double* q4j[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) q4j[j] = c3p_md_lds + (32 * cm.r + ((X4 + 4 * j + cm.c) ^ xr)) * 2;
  constexpr int IF1 = TALL ? IA1 : IB1, IF2 = TALL ? IA2 : IB2;  // images of the 16-column fragments
  constexpr int I41 = TALL ? IB1 : IA1, I42 = TALL ? IB2 : IA2;  // images of the 4-column fragments
  constexpr bool TWOF = TALL ? TWOA : TWOB, TWO4 = TALL ? TWOB : TWOA;
  d2 f1[NSP], f2[NSP], s1[NSP][3], s2[NSP][3];
  // centre block (wave 1): quad b of instruction h multiplies K-step 4h + b (K-step 7 = the zero rows 28..31); fetched when the
  // ring of the main loop no longer refills (the registers of its drained stages are free)
  double ca1[2], ca2[2], cb1[2], cb2[2];
  auto load_centre = [&]() {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const double* qc = c3p_md_lds + md_pw_off(16 * h + 4 * cm.b + cm.r, 12 + cm.c);
      ca1[h] = qc[IA1 * IMGR];
      cb1[h] = qc[IB1 * IMGR];
      if constexpr (TWOA) ca2[h] = qc[IA2 * IMGR];
      if constexpr (TWOB) cb2[h] = qc[IB2 * IMGR];
    }
  };
#define C3P_MMP_LOAD(P)                                                                                    \
  {                                                                                                        \
    constexpr int st_ = (P) % NSP;                                                                         \
    f1[st_] = *reinterpret_cast<const d2*>(qfm[(P) & 3] + IF1 * IMGR + (P) * 256);                         \
    if constexpr (TWOF) f2[st_] = *reinterpret_cast<const d2*>(qfm[(P) & 3] + IF2 * IMGR + (P) * 256);     \
    _Pragma("unroll") for (int e = 0; e < 3; ++e) {                                                        \
      const int j_ = (C3P_PW_TIMING ? 0 : e) ^ ((P) & 3);  /* timing builds: one fragment, read once */                                                  \
      s1[st_][e] = *reinterpret_cast<const d2*>(q4j[j_] + I41 * IMGR + (P) * 256);                         \
      if constexpr (TWO4) s2[st_][e] = *reinterpret_cast<const d2*>(q4j[j_] + I42 * IMGR + (P) * 256);     \
      if constexpr (C3P_PW_TIMING == 2 && true) {  /* same registers as the real build, one LDS read */   \
        if (e > 0) {                                                                                       \
          asm volatile("" : "+v"(s1[st_][e]));                                                             \
          if constexpr (TWO4) asm volatile("" : "+v"(s2[st_][e]));                                         \
        }                                                                                                  \
      }                                                                                                    \
    }                                                                                                      \
  }
#define C3P_MMP_FMAS(P, H)                                                                                 \
  {                                                                                                        \
    constexpr int st_ = (P) % NSP;                                                                         \
    if constexpr (QUAD) {                                                                                  \
      /* tall: f = left (A1, A2), s = right (B1, B2); wide: s = left, f = right */                          \
      _Pragma("unroll") for (int e = 0; e < 3; ++e)                                                        \
        acc1.sm[e] = TALL ? md_mfma4(f1[st_][H], s1[st_][e][H], acc1.sm[e]) : md_mfma4(s1[st_][e][H], f1[st_][H], acc1.sm[e]); \
      _Pragma("unroll") for (int e = 0; e < 3; ++e)                                                        \
        acc2.sm[e] = TALL ? md_mfma4(f1[st_][H], s2[st_][e][H], acc2.sm[e]) : md_mfma4(s1[st_][e][H], f2[st_][H], acc2.sm[e]); \
      _Pragma("unroll") for (int e = 0; e < 3; ++e) {                                                      \
        const double a_ = TALL ? f2[st_][H] : s2[st_][e][H], b_ = TALL ? s2[st_][e][H] : f2[st_][H];       \
        acc1.sm[e] = MODE == 3 ? md_mfma4(a_, b_, acc1.sm[e]) : md_mfma4n(a_, b_, acc1.sm[e]);             \
      }                                                                                                    \
      _Pragma("unroll") for (int e = 0; e < 3; ++e) {                                                      \
        const double a_ = TALL ? f2[st_][H] : s2[st_][e][H], b_ = TALL ? s1[st_][e][H] : f1[st_][H];       \
        acc2.sm[e] = MODE == 3 ? md_mfma4n(a_, b_, acc2.sm[e]) : md_mfma4(a_, b_, acc2.sm[e]);             \
      }                                                                                                    \
    } else if constexpr (MODE == 0 && C3P_PW_SPLITK != 0 && (H) == 1) {                                    \
      _Pragma("unroll") for (int e = 0; e < 3; ++e)                                                        \
        odd[e] = TALL ? md_mfma4(f1[st_][H], s1[st_][e][H], odd[e]) : md_mfma4(s1[st_][e][H], f1[st_][H], odd[e]); \
    } else {                                                                                               \
    _Pragma("unroll") for (int e = 0; e < 3; ++e) {                                                        \
      if constexpr (TALL) {                                                                                \
        acc1.sm[e] = md_mfma4(f1[st_][H], s1[st_][e][H], acc1.sm[e]);                                      \
        if constexpr (TWOB) acc2.sm[e] = md_mfma4(f1[st_][H], s2[st_][e][H], acc2.sm[e]);                  \
        if constexpr (TWOA) acc2.sm[e] = md_mfma4(f2[st_][H], s1[st_][e][H], acc2.sm[e]);                  \
      } else {                                                                                             \
        acc1.sm[e] = md_mfma4(s1[st_][e][H], f1[st_][H], acc1.sm[e]);                                      \
        if constexpr (TWOB) acc2.sm[e] = md_mfma4(s1[st_][e][H], f2[st_][H], acc2.sm[e]);                  \
        if constexpr (TWOA) acc2.sm[e] = md_mfma4(s2[st_][e][H], f1[st_][H], acc2.sm[e]);                  \
      }                                                                                                    \
    }                                                                                                      \
    }                                                                                                      \
  }
  // centre block (wave 1): operands fetched with pair 1's, its two K-packed instructions (+ two of the paired product) go
  // out behind pair 2's, the quad sums after the last pair -- nothing of it is left for the end of the product, where the other
  // three waves would wait at the barrier
  constexpr int CP_LOAD = C3P_PW_CPL, CP_FMA = C3P_PW_CPF;
  double c1 = 0.0, c2 = 0.0;
  // single products: the odd K-steps accumulate in a second register set (three dependent chains of a 4 x 4 x 4 instruction
  // leave no slack: 45 cycles of latency against 3 x 16 of issue)
  double odd[3] = {0.0, 0.0, 0.0};
  md_unroll<0, PFP>([&](auto Pc) { constexpr int P = decltype(Pc)::value; C3P_MMP_LOAD(P) });
  if constexpr (C3P_PW_PRIO != 0) __builtin_amdgcn_s_setprio(C3P_PW_PRIO);
  md_unroll<0, NP>([&](auto Pc) {
    constexpr int P = decltype(Pc)::value;
    if constexpr (P + PFP < NP) C3P_MMP_LOAD(P + PFP)
    if constexpr (WV == 1 && P == CP_LOAD) load_centre();
    if constexpr (C3P_PW_ILV == 0) __builtin_amdgcn_sched_barrier(0);  // keep the prefetch ahead of the MFMA group
    C3P_MMP_FMAS(P, 0)
    if constexpr (2 * P + 1 < NJ) C3P_MMP_FMAS(P, 1)
    if constexpr (WV == 1 && P == CP_FMA) {
      // the incoming value counts once (quad 0); after the quad sums every quad holds the sum over the four K-slices
      c1 = cm.b == 0 ? acc1.sm[3] : 0.0;
      if constexpr (TWOA || TWOB) c2 = cm.b == 0 ? acc2.sm[3] : 0.0;
      if constexpr (QUAD) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          c1 = md_mfma4(ca1[h], cb1[h], c1);
          c2 = md_mfma4(ca1[h], cb2[h], c2);
          c1 = MODE == 3 ? md_mfma4(ca2[h], cb2[h], c1) : md_mfma4n(ca2[h], cb2[h], c1);
          c2 = MODE == 3 ? md_mfma4n(ca2[h], cb1[h], c2) : md_mfma4(ca2[h], cb1[h], c2);
        }
      } else {
        c1 = md_mfma4(ca1[0], cb1[0], c1);
        if constexpr (TWOA || TWOB) c2 = md_mfma4(TWOA ? ca2[0] : ca1[0], TWOB ? cb2[0] : cb1[0], c2);
        c1 = md_mfma4(ca1[1], cb1[1], c1);
        if constexpr (TWOA || TWOB) c2 = md_mfma4(TWOA ? ca2[1] : ca1[1], TWOB ? cb2[1] : cb1[1], c2);
      }
    }
    if constexpr (C3P_PW_ILV != 0) {
      // the reads of the next pair go out BETWEEN this pair's matrix instructions: a read that waits for the LDS queue then
      // holds up one instruction that has its predecessor still in the pipe, not the whole group
      constexpr int NCR = QUAD ? 8 : (MODE == 0 ? 4 : 6), NCM = QUAD ? 8 : (MODE == 0 ? 2 : 4);  // centre: reads, matrix instructions
      constexpr int NM = (QUAD ? 12 : (MODE == 0 ? 3 : 6)) * (2 * P + 1 < NJ ? 2 : 1) + ((WV == 1 && P == CP_FMA) ? NCM : 0);
      constexpr int NR = ((P + PFP < NP) ? (QUAD ? 8 : (MODE == 0 ? 4 : (TWOF ? 5 : 7))) : 0) + ((WV == 1 && P == CP_LOAD) ? NCR : 0);
      constexpr int STEP = 1;  // (one matrix instruction per read from the start of the group: +0.5 % over an even spread)
      md_unroll<0, NR>([&](auto) {
        __builtin_amdgcn_sched_group_barrier(0x008, STEP, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      });
      __builtin_amdgcn_sched_group_barrier(0x008, NM, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  });
#undef C3P_MMP_LOAD
#undef C3P_MMP_FMAS
  if constexpr (C3P_PW_PRIO != 0) __builtin_amdgcn_s_setprio(0);
  if constexpr (MODE == 0 && C3P_PW_SPLITK != 0) {
#pragma unroll
    for (int e = 0; e < 3; ++e) acc1.sm[e] += odd[e];
  }
  if constexpr (WV == 1) {
    c1 += md_row_ror<4>(c1);
    c1 += md_row_ror<8>(c1);
    acc1.sm[3] = c1;
    if constexpr (TWOA || TWOB) {
      c2 += md_row_ror<4>(c2);
      c2 += md_row_ror<8>(c2);
      acc2.sm[3] = c2;
    }
  }
}

template <int NIGR, int NJ, int W, int WV, int MODE, int IA1, int IA2, int IB1, int IB2>
__device__ __forceinline__ void mm_real(const MidCommon& cm,
                                        TileRegs<WaveTiles<NIGR, NJ, W, WV>::NBW, WaveTiles<NIGR, NJ, W, WV>::NSW>& acc1,
                                        TileRegs<WaveTiles<NIGR, NJ, W, WV>::NBW, WaveTiles<NIGR, NJ, W, WV>::NSW>& acc2) {
  constexpr bool SWZ = NIGR <= 2;  // MDR::SWZ
  constexpr int WI = SWZ ? 32 : W;
  using T = WaveTiles<NIGR, NJ, WI, WV>;
  using S = Sched<NIGR, NJ>;
  if constexpr (S::PW) {
    mm_real_pw<WV, MODE, IA1, IA2, IB1, IB2>(cm, acc1, acc2);
    return;
  }
  constexpr int NB16 = S::NB16 > 0 ? S::NB16 : 1;
  constexpr int JR = NJ;  // small B blocks by column block J
  constexpr bool QUAD = MODE == 3 || MODE == 4;  // four products in one pass over the operands, as in mm_real_pw
  constexpr bool TWOA = MODE == 2 || QUAD, TWOB = MODE == 1 || QUAD;
  // D is in (4 NJ - 4, 4 NJ]: a real product has exactly NJ K-steps.  The loop is fully unrolled with the operands
  // fetched PF steps ahead in a ring of PF + 1 register stages: the LDS round trip (~150 cycles) is more than two
  // K-steps of a single 16x16x4 instruction, and a one-step prefetch left the matrix pipe waiting on every step.
  constexpr int PF = NIGR >= 3 ? 2 : (NJ < 3 ? NJ : 3);  // the 48-row classes are register-bound: one stage less
  constexpr int NS = PF + 1;
  double a[NS][NIGR], x[NS][NIGR], g[NS][NB16], h[NS][NB16], sb[NS][JR], ub[NS][JR];
  double aw[NS], xw[NS];  // wide units: rows 32..35 of the left operand(s), the same fragment in all four blocks
  const double* qw = c3p_md_lds + (cm.c * WI + cm.r);  // plain layout, block 0 of the last row group
  // one base register per lane-offset kind; the image (a compile-time index) goes into the instruction offset
  constexpr int IMGR = 16 * NIGR * WI;
  const double* qa = c3p_md_lds + cm.aoffR;  // plain layout: 16 rows x 4 columns of the left operand
  const double* qg0 = c3p_md_lds + cm.rbig[0];
  const double* qg1 = c3p_md_lds + cm.rbig[1];
  const double* qs0 = c3p_md_lds + cm.rblk[0];
  const double* qs1 = c3p_md_lds + cm.rblk[1];
#define C3P_MMR_LOAD(K)                                                                                          \
  {                                                                                                              \
    constexpr int st_ = (K) % NS;                                                                                \
    _Pragma("unroll") for (int Ig = 0; Ig < NIGR; ++Ig) {                                                        \
      if constexpr (SWZ) {                                                                                       \
        const double* q_ = Ig == 0 ? qg0 : qg1;                                                                  \
        a[st_][Ig] = T::uses_ig(Ig) ? q_[IA1 * IMGR + (K) * 4 * WI + 16 * Ig] : 0.0;                             \
        x[st_][Ig] = (TWOA && T::uses_ig(Ig)) ? q_[IA2 * IMGR + (K) * 4 * WI + 16 * Ig] : 0.0;                   \
      } else {                                                                                                   \
        a[st_][Ig] = T::uses_ig(Ig) ? qa[IA1 * IMGR + Ig * 16 * WI + 4 * (K)] : 0.0;                             \
        x[st_][Ig] = (TWOA && T::uses_ig(Ig)) ? qa[IA2 * IMGR + Ig * 16 * WI + 4 * (K)] : 0.0;                   \
      }                                                                                                          \
    }                                                                                                            \
    if constexpr (T::has_wide()) {                                                                               \
      aw[st_] = qw[IA1 * IMGR + (NIGR - 1) * 16 * WI + 4 * (K)];                                                 \
      if constexpr (TWOA) xw[st_] = qw[IA2 * IMGR + (NIGR - 1) * 16 * WI + 4 * (K)];                             \
    }                                                                                                            \
    _Pragma("unroll") for (int Jg = 0; Jg < S::NB16; ++Jg) {                                                     \
      const double* q_ = Jg == 0 ? qg0 : qg1;                                                                    \
      g[st_][Jg] = T::uses_jg(Jg) ? q_[IB1 * IMGR + (K) * 4 * WI + 16 * Jg] : 0.0;                               \
      h[st_][Jg] = (TWOB && T::uses_jg(Jg)) ? q_[IB2 * IMGR + (K) * 4 * WI + 16 * Jg] : 0.0;                     \
    }                                                                                                            \
    _Pragma("unroll") for (int J = 0; J < NJ; ++J) {                                                             \
      const double* q_ = J < 4 ? qs0 : qs1;                                                                      \
      sb[st_][J] = T::uses_j(J) ? q_[IB1 * IMGR + (K) * 4 * WI + 4 * J] : 0.0;                                   \
      ub[st_][J] = (TWOB && T::uses_j(J)) ? q_[IB2 * IMGR + (K) * 4 * WI + 4 * J] : 0.0;                         \
    }                                                                                                            \
  }
#define C3P_MMR_FMAS(K)                                                                                          \
  if constexpr (QUAD) {                                                                                          \
    constexpr int st_ = (K) % NS;                                                                                \
    _Pragma("unroll") for (int i = 0; i < T::NBW; ++i) {                                                         \
      if (T::is_wide(i)) {                                                                                       \
        acc1.big[i][0] = md_mfma4(aw[st_], g[st_][T::bJg(i)], acc1.big[i][0]);                                   \
        acc2.big[i][0] = md_mfma4(aw[st_], h[st_][T::bJg(i)], acc2.big[i][0]);                                   \
        acc1.big[i][0] = MODE == 3 ? md_mfma4(xw[st_], h[st_][T::bJg(i)], acc1.big[i][0]) : md_mfma4n(xw[st_], h[st_][T::bJg(i)], acc1.big[i][0]); \
        acc2.big[i][0] = MODE == 3 ? md_mfma4n(xw[st_], g[st_][T::bJg(i)], acc2.big[i][0]) : md_mfma4(xw[st_], g[st_][T::bJg(i)], acc2.big[i][0]); \
        continue;                                                                                                \
      }                                                                                                          \
      acc1.big[i] = md_mfma16(a[st_][T::bIg(i)], g[st_][T::bJg(i)], acc1.big[i]);                                \
      acc2.big[i] = md_mfma16(a[st_][T::bIg(i)], h[st_][T::bJg(i)], acc2.big[i]);                                \
      acc1.big[i] = MODE == 3 ? md_mfma16(x[st_][T::bIg(i)], h[st_][T::bJg(i)], acc1.big[i]) : md_mfma16n(x[st_][T::bIg(i)], h[st_][T::bJg(i)], acc1.big[i]); \
      acc2.big[i] = MODE == 3 ? md_mfma16n(x[st_][T::bIg(i)], g[st_][T::bJg(i)], acc2.big[i]) : md_mfma16(x[st_][T::bIg(i)], g[st_][T::bJg(i)], acc2.big[i]); \
    }                                                                                                            \
    _Pragma("unroll") for (int i = 0; i < T::NSW; ++i) {                                                         \
      acc1.sm[i] = md_mfma4(a[st_][T::sIg(i)], sb[st_][T::sJ(i)], acc1.sm[i]);                                   \
      acc2.sm[i] = md_mfma4(a[st_][T::sIg(i)], ub[st_][T::sJ(i)], acc2.sm[i]);                                   \
      acc1.sm[i] = MODE == 3 ? md_mfma4(x[st_][T::sIg(i)], ub[st_][T::sJ(i)], acc1.sm[i]) : md_mfma4n(x[st_][T::sIg(i)], ub[st_][T::sJ(i)], acc1.sm[i]); \
      acc2.sm[i] = MODE == 3 ? md_mfma4n(x[st_][T::sIg(i)], sb[st_][T::sJ(i)], acc2.sm[i]) : md_mfma4(x[st_][T::sIg(i)], sb[st_][T::sJ(i)], acc2.sm[i]); \
    }                                                                                                            \
  } else                                                                                                         \
  {                                                                                                              \
    constexpr int st_ = (K) % NS;                                                                                \
    _Pragma("unroll") for (int i = 0; i < T::NBW; ++i) {                                                         \
      if (T::is_wide(i)) {                                                                                       \
        acc1.big[i][0] = md_mfma4(aw[st_], g[st_][T::bJg(i)], acc1.big[i][0]);                                   \
        if constexpr (TWOB) acc2.big[i][0] = md_mfma4(aw[st_], h[st_][T::bJg(i)], acc2.big[i][0]);               \
        if constexpr (TWOA) acc2.big[i][0] = md_mfma4(xw[st_], g[st_][T::bJg(i)], acc2.big[i][0]);               \
        continue;                                                                                                \
      }                                                                                                          \
      acc1.big[i] = md_mfma16(a[st_][T::bIg(i)], g[st_][T::bJg(i)], acc1.big[i]);                                \
      if constexpr (TWOB) acc2.big[i] = md_mfma16(a[st_][T::bIg(i)], h[st_][T::bJg(i)], acc2.big[i]);            \
      if constexpr (TWOA) acc2.big[i] = md_mfma16(x[st_][T::bIg(i)], g[st_][T::bJg(i)], acc2.big[i]);            \
    }                                                                                                            \
    _Pragma("unroll") for (int i = 0; i < T::NSW; ++i) {                                                         \
      acc1.sm[i] = md_mfma4(a[st_][T::sIg(i)], sb[st_][T::sJ(i)], acc1.sm[i]);                                   \
      if constexpr (TWOB) acc2.sm[i] = md_mfma4(a[st_][T::sIg(i)], ub[st_][T::sJ(i)], acc2.sm[i]);               \
      if constexpr (TWOA) acc2.sm[i] = md_mfma4(x[st_][T::sIg(i)], sb[st_][T::sJ(i)], acc2.sm[i]);               \
    }                                                                                                            \
  }
  md_unroll<0, PF>([&](auto Kc) { constexpr int K = decltype(Kc)::value; C3P_MMR_LOAD(K) });
  if constexpr (C3P_MMR_PRIO != 0) __builtin_amdgcn_s_setprio(C3P_MMR_PRIO);
  md_unroll<0, NJ>([&](auto Kc) {
    constexpr int K = decltype(Kc)::value;
    if constexpr (K + PF < NJ) C3P_MMR_LOAD(K + PF)
    // 48-row classes (D = 33..40): the reads of a later K-step go out BETWEEN this step's matrix instructions, as in
    // mm_real_pw (cfg5 +1.9 %, its gradient +3.8 %; D = 40 +1.5 %); the 16- / 32-row classes keep the grouped order (D = 24 lost 11 % with it)
    constexpr int ILV = NIGR >= 3 ? (C3P_MMR_ILV) : 0;
    if constexpr (ILV == 0) __builtin_amdgcn_sched_barrier(0);  // keep the prefetch ahead of the MFMA group
    C3P_MMR_FMAS(K)
    if constexpr (ILV != 0) {
      md_unroll<0, 12>([&](auto) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, ILV, 0);
      });
      __builtin_amdgcn_sched_group_barrier(0x008, 32, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  });
#undef C3P_MMR_LOAD
#undef C3P_MMR_FMAS
  if constexpr (C3P_MMR_PRIO != 0) __builtin_amdgcn_s_setprio(0);
}

template <int NIG, int W>
struct MDR {
  static constexpr int NIGR = (NIG + 1) / 2;
  static constexpr int NIMG = 5;                       // real images of the slice pipeline
  // D <= 32: 32-wide images with the column index XOR 16 on odd rows, and EVERY operand read in the B pattern
  // (row 4K + r, 16 consecutive columns) -- the left operands of the real path are all symmetric, so the A
  // fragment A[i][4K + r] is read as M[4K + r][i].  Rows 4K and 4K + 1 then sit in different bank halves: no
  // bank conflicts on any operand read (a W = 30 image had 2-way conflicts on every B read).
  static constexpr bool SWZ = NIGR <= 2;
  static constexpr int WI = SWZ ? 32 : W;              // image row stride
  static constexpr int AREA = NIMG * 16 * NIGR * WI;   // doubles
  static constexpr int KP = 3;                         // control lines whose tables stay in registers
  static constexpr int WGS = (AREA * 8 + 6144) * 3 <= 160 * 1024 ? 3 : (C3P_MDR_BIG_WGS);
};

// doubles of the real image area: the pinwheel class runs its unsquared degree-20 slices on SIX images (see PLAN6 below)
template <int NIG, int NJ, int W>
constexpr int mdr_area() {
  return MDR<NIG, W>::AREA + (Sched<MDR<NIG, W>::NIGR, NJ>::PW ? 16 * MDR<NIG, W>::NIGR * MDR<NIG, W>::WI : 0);
}

template <int NIG, int NJ, int W, bool DUS, int WV>
__device__ __forceinline__ void midd_real_body(const MidArgs& A, const MidCommon& cm, long chain) {
  constexpr int NIGR = MDR<NIG, W>::NIGR;
  constexpr int WI = MDR<NIG, W>::WI;         // image row stride (32, swizzled, for D <= 32)
  using T = WaveTiles<NIGR, NJ, WI, WV>;
  constexpr int IMG = MD<NIG, NJ>::ROWS * W;  // complex table image
  constexpr int IMGR = 16 * NIGR * WI, NE = T::NE;
  typedef TileRegs<T::NBW, T::NSW> Regs;
  const int D = cm.D, K = cm.K;
  const int lbig = cm.lbig, lsmall = cm.lsmall;
  const int rbig = cm.r, rsmall = 4 * cm.b + cm.r;
  const int cbig = 4 * cm.b + cm.c, csmall = cm.c;
  const double* tabs = cm.tabs;
  // image i adds i * IMGR: a compile-time instruction offset; [1] = units in columns 16..31 (swizzled layout)
  double* const qbig0 = c3p_md_lds + cm.rbig[0];
  double* const qbig1 = c3p_md_lds + cm.rbig[1];
  double* const qsm0 = c3p_md_lds + cm.rsm[0];
  double* const qsm1 = c3p_md_lds + cm.rsm[1];
  auto erow = [&](int e) -> int { return T::lrow(e, cm.r, cm.b, cm.c); };
  auto ecol = [&](int e) -> int { return T::lcol(e, cm.r, cm.b, cm.c); };
  // pinwheel deal: one lane offset per element (swizzled position of its row / column)
  constexpr bool PW = Sched<NIGR, NJ>::PW;
  int pwoff[NE > 0 ? NE : 1];
#pragma unroll
  for (int e = 0; e < NE; ++e) pwoff[e] = PW ? md_pw_off(erow(e), ecol(e)) : 0;
  auto store_tiles = [&](auto img, const Regs& v) {
    constexpr int I = decltype(img)::value;
#pragma unroll
    for (int e = 0; e < NE; ++e) {
      if constexpr (PW)
        c3p_md_lds[I * IMGR + pwoff[e]] = v.get(e);
      else
        (T::is_big(e) ? (T::col0(e) < 16 ? qbig0 : qbig1) : (T::col0(e) < 16 ? qsm0 : qsm1))[I * IMGR + T::off0(e)] = v.get(e);
    }
  };
  auto zero = [&](Regs& v) {
#pragma unroll
    for (int e = 0; e < NE; ++e) v.set(e, 0.0);
  };
  // bit e: element e of the lane's tile set lies on the diagonal (inside the matrix)
  unsigned dbits = 0;
#pragma unroll
  for (int e = 0; e < NE; ++e) {
    const int row = erow(e), col = ecol(e);
    dbits |= (row == col && col < D) ? (1u << e) : 0u;
  }
  auto dmask = [&](int e) -> double { return (dbits >> e) & 1u ? 1.0 : 0.0; };
  Regs Ur, Ui;
  double mus_r = 0.0, mus_i = 0.0;
  Regs dummy;
  // The lane's table elements do not depend on the slice: -scale Im(table) at the lane's positions stays in
  // registers for the whole segment (the first KP control lines; further ones are read per slice).  Im rows of the
  // half-image tables hold -Y;
  // positions outside the matrix are clamped and masked.
  constexpr int KP = MDR<NIG, W>::KP;
  // Pinwheel class (TABL): the slice loop is register-bound (three workgroups per CU: 168 registers), so the table elements
  // are NOT kept across the slice -- they are read again (L2) while the chain products of the previous slice run and are dead
  // after Y is formed: 8 NE registers less at the peak of the polynomial stage.
  constexpr bool TABL = false && PW;
  Regs Tab[KP + 1];
  double tmu_r[KP + 1], tmu_i[KP + 1];
  int tix[NE > 0 ? NE : 1];
  unsigned inb = 0;
#pragma unroll
  for (int e = 0; e < NE; ++e) {
    const int row = erow(e), col = ecol(e);
    const bool in = row < D && col < D;
    tix[e] = in ? (2 * row + 1) * W + col : W;
    inb |= in ? (1u << e) : 0u;
  }
  auto load_tabs = [&]() {
    const double* tb = tabs;
    if constexpr (TABL) asm volatile("" : "+s"(tb));  // opaque per call: the (loop-invariant) loads must not be hoisted
#pragma unroll
    for (int k = 0; k <= KP; ++k) {
      const double* tk = tb + (long)(k <= K ? k : 0) * (IMG + 4);
      const double on = k <= K ? -cm.scale : 0.0;
#pragma unroll
      for (int e = 0; e < NE; ++e) Tab[k].set(e, (inb >> e) & 1u ? on * tk[tix[e]] : 0.0);
    }
  };
  load_tabs();
#pragma unroll
  for (int k = 0; k <= KP; ++k) {
    const double* tk = tabs + (long)(k <= K ? k : 0) * (IMG + 4);
    tmu_r[k] = md_rfl(k <= K ? tk[IMG + 0] : 0.0);
    tmu_i[k] = md_rfl(k <= K ? tk[IMG + 1] : 0.0);
  }
  // instantiated per polynomial variant with the branch outside the loop (as in the small-D kernel): below
  // theta_16 = 0.816 the degree-16 / 17 polynomials are exact to roundoff and W^3, W^4 are one paired product
  auto real_loop = [&](auto var_tag, auto plan_tag) {
  constexpr int VAR = decltype(var_tag)::value;  // Taylor degree of cos: 16, 18 or 20
  constexpr bool DEG16 = VAR == 16, DEG20 = VAR == 20;
  // Pinwheel class, degree 20 without squarings (cfg3): six images.  Y keeps image 0 for the whole slice (W^5 goes to image 5),
  // so sin Y = Y (sin Y / Y) reads it there -- no second store of Y -- and the operands of the chain step (C, S, Ur, Ui ->
  // images 1, 3, 4, 5) are none of the two that product reads: the barrier in front of their stores goes (7 instead of 8
  // per slice, 12 tile-set stores instead of 13).
  constexpr bool PLAN6 = decltype(plan_tag)::value != 0;
  for (int t = 0; t < cm.len; ++t) {
    if ((t & (SGC - 1)) == 0) md_stage_signals<WV>(A, cm, t);
    double mu_r = 0.0, mu_i = 0.0;
    auto form_Y = [&](Regs& Yo) {
      mu_r = tmu_r[0], mu_i = tmu_i[0];
      Yo = Tab[0];
#pragma unroll
      for (int k = 0; k < KP; ++k) {
        const double c0 = k < K ? cm.sg[k * SGC + (t & (SGC - 1))] : 0.0;  // Tab[k + 1] = 0 beyond K
        mu_r = fma(c0, tmu_r[k + 1], mu_r);
        mu_i = fma(c0, tmu_i[k + 1], mu_i);
#pragma unroll
        for (int e = 0; e < NE; ++e) Yo.set(e, fma(c0, Tab[k + 1].get(e), Yo.get(e)));
      }
      for (int k = KP; k < K; ++k) {  // control lines beyond the register budget: their tables come from L2 every slice
        const double c0 = cm.sg[k * SGC + (t & (SGC - 1))];
        const double* tk = tabs + (long)(k + 1) * (IMG + 4);
        mu_r = fma(c0, md_rfl(tk[IMG + 0]), mu_r);
        mu_i = fma(c0, md_rfl(tk[IMG + 1]), mu_i);
        const double f = -cm.scale * c0;
#pragma unroll
        for (int e = 0; e < NE; ++e) {
          const int row = erow(e), col = ecol(e);
          const bool in = row < D && col < D;
          Yo.set(e, fma(f, in ? tk[(2 * row + 1) * W + col] : 0.0, Yo.get(e)));
        }
      }
    };
    Regs Y;
    form_Y(Y);
    store_tiles(IC<0>{}, Y);
    md_bar();
    Regs W1, W2, W3, Cm, Sp, acc, acs;
    zero(W1);
    mm_real<NIGR, NJ, W, WV, 0, 0, 0, 0, 0>(cm, W1, dummy);  // W = Y^2
    store_tiles(IC<1>{}, W1);
    md_bar();
    zero(W2);
    mm_real<NIGR, NJ, W, WV, 0, 1, 1, 1, 1>(cm, W2, dummy);  // W^2
    store_tiles(IC<2>{}, W2);
    md_bar();
    zero(W3);
    auto rc = [&](Regs& out, double c0, double c1, double c2, double c3) {
#pragma unroll
      for (int e = 0; e < NE; ++e) {
        double v = c1 * W1.get(e);
        v = fma(c2, W2.get(e), v);
        v = fma(c3, W3.get(e), v);
        out.set(e, fma(c0, dmask(e), v));
      }
    };
    Regs Sn;
    zero(Sn);
    // cos: c_j = (-1)^j / (2j)!;  sin / Y: s_j = (-1)^j / (2j+1)!
    if constexpr (DEG20) {
      // q = 5 (theta_20 = 1.49: cfg3 / cfg5 need no squaring): {W^3, W^4} as in the degree-16 variant, W^5 = W^2 W^3,
      // then ONE paired Horner step in W^5:  p = B0(W..W^4) + W^5 (B1(W..W^4) + c10 W^5)
      Regs W4, W5;
      zero(W4);
      zero(W5);
      mm_real<NIGR, NJ, W, WV, 2, 1, 2, 2, 2>(cm, W3, W4);
      store_tiles(IC<3>{}, W3);
      // everything W .. W^4 feed is formed BEFORE the W^5 product, whose operands come from the images: the four power
      // tile sets are dead while it runs and afterwards (16 -> 12 live tile sets at the peak; the slice loop is register-bound)
      rc(Cm, 1.0, -c3p_inv_fact[2], c3p_inv_fact[4], -c3p_inv_fact[6]);
      rc(Sp, 1.0, -c3p_inv_fact[3], c3p_inv_fact[5], -c3p_inv_fact[7]);
#pragma unroll
      for (int e = 0; e < NE; ++e) {
        Cm.set(e, fma(c3p_inv_fact[8], W4.get(e), Cm.get(e)));
        Sp.set(e, fma(c3p_inv_fact[9], W4.get(e), Sp.get(e)));
        double a = -c3p_inv_fact[10] * dmask(e), s = -c3p_inv_fact[11] * dmask(e);
        a = fma(c3p_inv_fact[12], W1.get(e), a), s = fma(c3p_inv_fact[13], W1.get(e), s);
        a = fma(-c3p_inv_fact[14], W2.get(e), a), s = fma(-c3p_inv_fact[15], W2.get(e), s);
        a = fma(c3p_inv_fact[16], W3.get(e), a), s = fma(c3p_inv_fact[17], W3.get(e), s);
        acc.set(e, fma(-c3p_inv_fact[18], W4.get(e), a));
        acs.set(e, fma(-c3p_inv_fact[19], W4.get(e), s));
      }
      md_bar();
      mm_real<NIGR, NJ, W, WV, 0, 2, 2, 3, 3>(cm, W5, dummy);
#pragma unroll
      for (int e = 0; e < NE; ++e) {
        acc.set(e, fma(c3p_inv_fact[20], W5.get(e), acc.get(e)));
        acs.set(e, fma(c3p_inv_fact[21], W5.get(e), acs.get(e)));
      }
      if constexpr (PLAN6) {
        store_tiles(IC<5>{}, W5);
        store_tiles(IC<1>{}, acc);
        store_tiles(IC<4>{}, acs);
        md_bar();
        mm_real<NIGR, NJ, W, WV, 1, 5, 5, 1, 4>(cm, Cm, Sp);  // Cm = cos Y, Sp = sin(Y) / Y
        store_tiles(IC<2>{}, Sp);
        md_bar();
        mm_real<NIGR, NJ, W, WV, 0, 0, 0, 2, 2>(cm, Sn, dummy);  // sin Y
      } else {
        store_tiles(IC<0>{}, W5);
        store_tiles(IC<1>{}, acc);
        store_tiles(IC<4>{}, acs);
        md_bar();
        mm_real<NIGR, NJ, W, WV, 1, 0, 0, 1, 4>(cm, Cm, Sp);  // Cm = cos Y, Sp = sin(Y) / Y
        store_tiles(IC<2>{}, Sp);
        store_tiles(IC<3>{}, Y);
        md_bar();
        mm_real<NIGR, NJ, W, WV, 0, 3, 3, 2, 2>(cm, Sn, dummy);  // sin Y
      }
    } else if constexpr (DEG16) {
      // q = 4: {W^3, W^4} = {W, W^2} W^2 as one paired product, then ONE paired Horner step in W^4
      Regs W4;
      zero(W4);
      mm_real<NIGR, NJ, W, WV, 2, 1, 2, 2, 2>(cm, W3, W4);
      // (round 6: the ECONOMISED degree-8 pair of c3p_common.h -- same products, valid to ||Y|| = 1.85 instead of 0.816)
      rc(acc, c3p_mm8_cos[4], c3p_mm8_cos[5], c3p_mm8_cos[6], c3p_mm8_cos[7]);
      rc(acs, c3p_mm8_sinc[4], c3p_mm8_sinc[5], c3p_mm8_sinc[6], c3p_mm8_sinc[7]);
#pragma unroll
      for (int e = 0; e < NE; ++e) {
        acc.set(e, fma(c3p_mm8_cos[8], W4.get(e), acc.get(e)));
        acs.set(e, fma(c3p_mm8_sinc[8], W4.get(e), acs.get(e)));
      }
      store_tiles(IC<3>{}, W4);
      store_tiles(IC<0>{}, acc);
      store_tiles(IC<4>{}, acs);
      md_bar();
      rc(Cm, c3p_mm8_cos[0], c3p_mm8_cos[1], c3p_mm8_cos[2], c3p_mm8_cos[3]);
      rc(Sp, c3p_mm8_sinc[0], c3p_mm8_sinc[1], c3p_mm8_sinc[2], c3p_mm8_sinc[3]);
      mm_real<NIGR, NJ, W, WV, 1, 3, 3, 0, 4>(cm, Cm, Sp);  // Cm = cos Y, Sp = sin(Y) / Y
      store_tiles(IC<1>{}, Sp);
      store_tiles(IC<2>{}, Y);
      md_bar();
      mm_real<NIGR, NJ, W, WV, 0, 2, 2, 1, 1>(cm, Sn, dummy);  // sin Y
    } else {
      // degree 18 / 17: two paired Horner steps in W^3
      mm_real<NIGR, NJ, W, WV, 0, 1, 1, 2, 2>(cm, W3, dummy);  // W^3
      rc(Cm, c3p_inv_fact[12], -c3p_inv_fact[14], c3p_inv_fact[16], -c3p_inv_fact[18]);
      rc(Sp, c3p_inv_fact[13], -c3p_inv_fact[15], c3p_inv_fact[17], 0.0);
      store_tiles(IC<3>{}, W3);
      store_tiles(IC<0>{}, Cm);
      store_tiles(IC<4>{}, Sp);
      md_bar();
      rc(acc, -c3p_inv_fact[6], c3p_inv_fact[8], -c3p_inv_fact[10], 0.0);
      rc(acs, -c3p_inv_fact[7], c3p_inv_fact[9], -c3p_inv_fact[11], 0.0);
      mm_real<NIGR, NJ, W, WV, 1, 3, 3, 0, 4>(cm, acc, acs);
      store_tiles(IC<1>{}, acc);
      store_tiles(IC<2>{}, acs);
      md_bar();
      rc(Cm, 1.0, -c3p_inv_fact[2], c3p_inv_fact[4], 0.0);
      rc(Sp, 1.0, -c3p_inv_fact[3], c3p_inv_fact[5], 0.0);
      mm_real<NIGR, NJ, W, WV, 1, 3, 3, 1, 2>(cm, Cm, Sp);  // Cm = cos Y, Sp = sin(Y) / Y
      store_tiles(IC<0>{}, Sp);
      store_tiles(IC<4>{}, Y);
      md_bar();
      mm_real<NIGR, NJ, W, WV, 0, 4, 4, 0, 0>(cm, Sn, dummy);  // sin Y
    }
    // ---- squarings in real form: cos 2Y = 2 C^2 - I, sin 2Y = 2 S C (image pairs alternate: no extra barrier) ----
    auto square = [&](auto ia, auto ib) {
      constexpr int IA = decltype(ia)::value, IB = decltype(ib)::value;
      store_tiles(ia, Cm);
      store_tiles(ib, Sn);
      md_bar();
      Regs C2, SC;
      zero(C2);
      zero(SC);
      mm_real<NIGR, NJ, W, WV, 2, IA, IB, IA, IA>(cm, C2, SC);
#pragma unroll
      for (int e = 0; e < NE; ++e) {
        Cm.set(e, fma(2.0, C2.get(e), -dmask(e)));
        Sn.set(e, 2.0 * SC.get(e));
      }
    };
    for (int it = 0; it < cm.ps; ++it) {
      // the image pair not read by the previous product (sin Y reads images 2, 1 / 4, 0 / 3, 2 at degree 16 / 18 / 20)
      if ((it & 1) != 0)
        DEG16 ? square(IC<1>{}, IC<2>{}) : square(IC<3>{}, IC<4>{});
      else if constexpr (DEG20)
        square(IC<0>{}, IC<1>{});
      else
        DEG16 ? square(IC<3>{}, IC<4>{}) : square(IC<1>{}, IC<2>{});
    }
    if constexpr (DUS) {
      // dU = e^{mu} (C - iS)
      double sn, cs;
      sincos(mu_i, &sn, &cs);
      const double er = exp(mu_r);
      const double pr = er * cs, pi = er * sn;
      double* dst = reinterpret_cast<double*>(A.dUs_out) + ((long)cm.sample * A.N + cm.n0 + t) * D * D * 2;
#pragma unroll
      for (int e = 0; e < NE; ++e) {
        const int row = erow(e), col = ecol(e);
        if (row < D && col < D) {
          dst[(row * D + col) * 2 + 0] = fma(pr, Cm.get(e), pi * Sn.get(e));
          dst[(row * D + col) * 2 + 1] = fma(pi, Cm.get(e), -pr * Sn.get(e));
        }
      }
    }
    // ---- chain in real blocks: Ur' = C Ur + S Ui,  Ui' = C Ui - S Ur ----
    if constexpr (TABL) load_tabs();  // for the next slice, in flight behind the chain products
    if (t == 0) {
#pragma unroll
      for (int e = 0; e < NE; ++e) {
        Ur.set(e, Cm.get(e));
        Ui.set(e, -Sn.get(e));
      }
      mus_r = mu_r;
      mus_i = c3p_phase_add(0.0, mu_i);
      md_bar();  // image 0 is rewritten by the next slice
    } else {
      if constexpr (!PLAN6) md_bar();  // the last product's operands are no longer read
      store_tiles(IC<1>{}, Cm);
      store_tiles(IC<PLAN6 ? 3 : 2>{}, Sn);
      store_tiles(IC<PLAN6 ? 4 : 3>{}, Ur);
      store_tiles(IC<PLAN6 ? 5 : 4>{}, Ui);
      md_bar();
      Regs Vr, Vi;
      zero(Vr);
      zero(Vi);
      // (the 48-row classes take the four-product pass of the pinwheel class too: cfg5 +0.6 %, D = 36 / 40 at 256 samples +1.5 - 2 %;
      // no gain on the 16- / 32-row classes)
      constexpr bool QUADC = PW || NIGR >= 3;
      if constexpr (PLAN6) {
        mm_real<NIGR, NJ, W, WV, 3, 1, 3, 4, 5>(cm, Vr, Vi);  // all four products in one pass over the operands
      } else if constexpr (QUADC) {
        mm_real<NIGR, NJ, W, WV, 3, 1, 2, 3, 4>(cm, Vr, Vi);
      } else {
        mm_real<NIGR, NJ, W, WV, 1, 2, 2, 4, 3>(cm, Vr, Vi);  // S Ui, S Ur
#pragma unroll
        for (int e = 0; e < NE; ++e) Vi.set(e, -Vi.get(e));
        mm_real<NIGR, NJ, W, WV, 1, 1, 1, 3, 4>(cm, Vr, Vi);  // + C Ur, + C Ui
      }
      Ur = Vr;
      Ui = Vi;
      mus_r += mu_r;
      mus_i = c3p_phase_add(mus_i, mu_i);
    }
  }
  };
  if (cm.t18 == 2) {  // (reused as the variant flag on the real path: 1 = degree 16, 0 = degree 18, 2 = degree 20)
    if (PW && cm.ps == 0)
      real_loop(IC<20>{}, IC<PW ? 1 : 0>{});
    else
      real_loop(IC<20>{}, IC<0>{});
  } else if (cm.t18 == 1) {
    real_loop(IC<16>{}, IC<0>{});
  } else {
    real_loop(IC<18>{}, IC<0>{});
  }
  // ---- segment result: e^{sum mu} (Ur + i Ui), optional row phases ----
  double sn, cs;
  sincos(mus_i, &sn, &cs);
  const double er = exp(mus_r);
  double* dst = reinterpret_cast<double*>(A.seg_out) + chain * D * D * 2;
  const double* ph = A.fr_phase ? A.fr_phase + (long)cm.sample * D : nullptr;
#pragma unroll
  for (int e = 0; e < NE; ++e) {
    const int row = erow(e), col = ecol(e);
    if (row < D && col < D) {
      double sr = er * cs, si = er * sn;
      if (ph != nullptr) {
        double s2, c2;
        sincos(ph[row], &s2, &c2);
        const double tr = sr * c2 - si * s2;
        si = sr * s2 + si * c2;
        sr = tr;
      }
      dst[(row * D + col) * 2 + 0] = fma(sr, Ur.get(e), -si * Ui.get(e));
      dst[(row * D + col) * 2 + 1] = fma(sr, Ui.get(e), si * Ur.get(e));
    }
  }
}

// The whole slice loop, specialised per wave so that every tile index is a compile-time
// constant (LDS offsets become instruction immediates; no per-tile predication).
template <int NIG, int NJ, int W, bool GIVEN, bool DUS, bool XG, int WV>
__device__ __forceinline__ void midd_body(const MidArgs& A, const MidCommon& cm, long chain) {
  using T = WaveTiles<NIG, NJ, W, WV>;
  constexpr int IMG = MD<NIG, NJ>::ROWS * W, NE = T::NE;
  typedef TileRegs<T::NBW, T::NSW> Regs;
  const int D = cm.D, K = cm.K, r = cm.r;
  // lane parts of an element's image offset / row / column
  const int lbig = cm.lbig, lsmall = cm.lsmall;
  const int rbig = cm.r, rsmall = 4 * cm.b + cm.r;   // row inside the 16-row group (+ 4q for bigs)
  const int cbig = 4 * cm.b + cm.c, csmall = cm.c;   // column inside the unit
  double mus_r = 0.0, mus_i = 0.0;
  const double* tabs = cm.tabs;
  Regs U;

  auto eoff = [&](int e) -> int { return T::off0(e) + (T::is_big(e) ? lbig : lsmall); };
  auto erow = [&](int e) -> int { return T::row0(e) + (T::is_big(e) ? rbig : rsmall); };
  auto ecol = [&](int e) -> int { return T::col0(e) + (T::is_big(e) ? cbig : csmall); };
  auto store_tiles = [&](double* img, const Regs& v) {
#pragma unroll
    for (int e = 0; e < NE; ++e) img[eoff(e)] = v.get(e);
  };
  auto zero = [&](Regs& v) {
#pragma unroll
    for (int e = 0; e < NE; ++e) v.set(e, 0.0);
  };
  auto product = [&](const double* imgA, const double* imgB, Regs& acc) {
    mm_tiles_pf<NIG, NJ, W, WV, 2>(imgA, imgB, cm, acc);
  };
  auto is_diag = [&](int e) -> bool {
    const int row = erow(e), col = ecol(e);
    return ((row & 1) == 0) && ((row >> 1) == col) && (col < D);
  };
  // store a tile set to a plain complex [D][D] array times the scalar (sr + i si) (row phases opt.)
  auto store_plain = [&](double* dst, const Regs& v, double sr0, double si0, const double* ph) {
#pragma unroll
    for (int e = 0; e < NE; ++e) {
      const int row = erow(e), col = ecol(e);
      const int ci = row >> 1;
      double sr = sr0, si = si0;
      if (ph != nullptr && ci < D) {
        double s2, c2;
        sincos(ph[ci], &s2, &c2);
        const double tr = sr * c2 - si * s2;
        si = sr * s2 + si * c2;
        sr = tr;
      }
      const double mine = v.get(e);
      const double other = __shfl_xor(mine, 16);  // partner row r^1 (Re <-> Im) in both unit kinds
      const double outv = (r & 1) ? fma(sr, mine, si * other) : fma(sr, mine, -si * other);
      if (ci < D && col < D) dst[(ci * D + col) * 2 + (r & 1)] = outv;
    }
  };

  // the slice loop is instantiated per plan (T18 / Paterson-Stockmeyer), branch outside: inside the loop the two
  // variants' live ranges merge
  auto slice_loop = [&](auto t18_tag) {
  constexpr int VARIANT = decltype(t18_tag)::value;  // 0 Paterson-Stockmeyer, 1 T18, 2 the four-product scheme (c3p_e4n, normal generators)
  constexpr bool T18 = VARIANT == 1;
  for (int t = 0; t < cm.len; ++t) {
    if constexpr (!GIVEN && !XG)
      if ((t & (SGC - 1)) == 0) md_stage_signals<WV>(A, cm, t);
    Regs P;
    zero(P);
    double mu_r = 0.0, mu_i = 0.0;
    if constexpr (GIVEN) {
      const double* src = reinterpret_cast<const double*>(A.mats) + ((long)cm.sample * A.N + cm.n0 + t) * D * D * 2;
#pragma unroll
      for (int e = 0; e < NE; ++e) {
        const int row = erow(e), col = ecol(e);
        const int ci = row >> 1;
        P.set(e, (ci < D && col < D) ? src[(ci * D + col) * 2 + (row & 1)] : 0.0);
      }
    } else {
      // ---- X = scale (G0 + sum_k c_k G_k) at the lane's element positions ----
      Regs X;
      if constexpr (XG) {
        // supplied generator: X = scale (coef * hs[b,n] - mu), mu from the hmeta pre-pass
        const long m = (long)cm.sample * A.N + cm.n0 + t;
        mu_r = A.meta[m * 4 + 0];
        mu_i = A.meta[m * 4 + 1];
        const double2* src = reinterpret_cast<const double2*>(A.hs) + (long)cm.sample * A.hs_bstride + (long)(cm.n0 + t) * D * D;
#pragma unroll
        for (int e = 0; e < NE; ++e) {
          const int row = erow(e), col = ecol(e);
          const int ci = row >> 1;
          const bool in = ci < D && col < D;
          const double2 h = src[in ? ci * D + col : 0];
          double v = (row & 1) ? fma(A.coef_r, h.y, A.coef_i * h.x) : fma(A.coef_r, h.x, -A.coef_i * h.y);
          v -= (ci == col) ? ((row & 1) ? mu_i : mu_r) : 0.0;
          X.set(e, in ? cm.scale * v : 0.0);
        }
      } else {
      mu_r = tabs[IMG + 0];
      mu_i = tabs[IMG + 1];
#pragma unroll
      for (int e = 0; e < NE; ++e) X.set(e, cm.scale * tabs[eoff(e)]);
      for (int k = 0; k < K; ++k) {
        const double c0 = cm.sg[k * SGC + (t & (SGC - 1))];
        const double ck = cm.scale * c0;
        const double* tk = tabs + (long)(k + 1) * (IMG + 4);
        mu_r = fma(c0, tk[IMG + 0], mu_r);
        mu_i = fma(c0, tk[IMG + 1], mu_i);
#pragma unroll
        for (int e = 0; e < NE; ++e) X.set(e, fma(ck, tk[eoff(e)], X.get(e)));
      }
      }
      store_tiles(cm.buf0, X);
      __syncthreads();
      // ---- powers: A2 = X X, A3 = X A2.  buf0 = X (left operand), buf1 = A2 then A3 (right
      //      operands); A2, A3 also stay in registers for the lane-local block polynomials ----
      Regs A2, A3, acc;
      zero(A2);
      product(cm.buf0, cm.buf0, A2);
      store_tiles(cm.buf1, A2);
      __syncthreads();
      if constexpr (VARIANT == 2) {
        // ---- four products (c3p_common.h): y0 = A2 (e0 A2 + e1 X); y1 = (y0 + e2 A2 + e3 X)(y0 + e4 A2) + e5 y0 + e6 A2;
        //      P = (y1 + e7 A2 + e8 X)(y1 + e9 y0 + e10 X) + e11 y1 + e12 y0 + e13 A2 + e14 X + e15 I.  A3 holds y0, acc y1 ----
        constexpr const double (&ec)[16] = c3p_e4n;
        zero(A3);
        Regs T1, T2;
#pragma unroll
        for (int e = 0; e < NE; ++e) T1.set(e, fma(ec[0], A2.get(e), ec[1] * X.get(e)));
        store_tiles(cm.buf2, T1);
        __syncthreads();
        product(cm.buf1, cm.buf2, A3);  // y0
#pragma unroll
        for (int e = 0; e < NE; ++e) {
          T1.set(e, fma(ec[2], A2.get(e), fma(ec[3], X.get(e), A3.get(e))));
          T2.set(e, fma(ec[4], A2.get(e), A3.get(e)));
        }
        __syncthreads();  // buf0 (X), buf2 no longer read
        store_tiles(cm.buf0, T1);
        store_tiles(cm.buf2, T2);
#pragma unroll
        for (int e = 0; e < NE; ++e) acc.set(e, fma(ec[5], A3.get(e), ec[6] * A2.get(e)));
        __syncthreads();
        product(cm.buf0, cm.buf2, acc);  // y1
#pragma unroll
        for (int e = 0; e < NE; ++e) {
          T1.set(e, fma(ec[7], A2.get(e), fma(ec[8], X.get(e), acc.get(e))));
          T2.set(e, fma(ec[9], A3.get(e), fma(ec[10], X.get(e), acc.get(e))));
          double v = fma(ec[11], acc.get(e), ec[12] * A3.get(e));
          v = fma(ec[13], A2.get(e), v);
          v = fma(ec[14], X.get(e), v);
          v += is_diag(e) ? ec[15] : 0.0;
          P.set(e, v);
        }
        __syncthreads();
        store_tiles(cm.buf0, T1);
        store_tiles(cm.buf2, T2);
        __syncthreads();
        product(cm.buf0, cm.buf2, P);
      } else {
      zero(A3);
      product(cm.buf0, cm.buf1, A3);
      __syncthreads();  // all waves done reading A2 from buf1
      store_tiles(cm.buf1, A3);
      __syncthreads();
      zero(acc);
      if constexpr (T18) {
        // ---- T18 (Bader-Blanes-Casas), 5 products: A6 = A3 A3; A9 = B1 B5 + B4; P = B2 + (B3 + A9) A9
        product(cm.buf1, cm.buf1, acc);  // acc = A6
        auto comb = [&](Regs& out, double c0, double cx, double c2, double c3, double c6) {
#pragma unroll
          for (int e = 0; e < NE; ++e) {
            double v = cx * X.get(e);
            v = fma(c2, A2.get(e), v);
            v = fma(c3, A3.get(e), v);
            v = fma(c6, acc.get(e), v);
            v += (c0 != 0.0 && is_diag(e)) ? c0 : 0.0;
            out.set(e, v);
          }
        };
        const double* tc = c3p_t18_tab[cm.t18n];
        Regs T1, T2;
        comb(T1, 0.0, tc[C3P_I_A11], tc[C3P_I_A21], tc[C3P_I_A31], 0.0);  // B1
        comb(T2, 0.0, 0.0, tc[C3P_I_B24], tc[C3P_I_B34], tc[C3P_I_B64]);  // B5
        __syncthreads();  // buf0 (X) no longer read
        store_tiles(cm.buf0, T1);
        store_tiles(cm.buf2, T2);
        comb(T1, tc[C3P_I_B03], tc[C3P_I_B13], tc[C3P_I_B23], tc[C3P_I_B33], tc[C3P_I_B63]);  // B4
        __syncthreads();
        product(cm.buf0, cm.buf2, T1);  // T1 = A9
        comb(T2, tc[C3P_I_B02], tc[C3P_I_B12], tc[C3P_I_B22], tc[C3P_I_B32], tc[C3P_I_B62]);  // B3
#pragma unroll
        for (int e = 0; e < NE; ++e) T2.set(e, T2.get(e) + T1.get(e));
        comb(P, 0.0, tc[C3P_I_B11], tc[C3P_I_B21], tc[C3P_I_B31], tc[C3P_I_B61]);  // B2
        __syncthreads();  // buf0 / buf2 no longer read
        store_tiles(cm.buf0, T2);
        store_tiles(cm.buf2, T1);
        __syncthreads();
        product(cm.buf0, cm.buf2, P);
      } else {
        product(cm.buf0, cm.buf1, acc);  // acc = X^4
        // ---- Horner init: P = c_m X^4 + B_{r-1} ----
        {
          const int j = cm.pr - 1;
          const double c0 = c3p_inv_fact[4 * j], c1 = c3p_inv_fact[4 * j + 1], c2 = c3p_inv_fact[4 * j + 2],
                       c3 = c3p_inv_fact[4 * j + 3], cmm = c3p_inv_fact[4 * cm.pr];
#pragma unroll
          for (int e = 0; e < NE; ++e) {
            double v = cmm * acc.get(e);
            v = fma(c1, X.get(e), v);
            v = fma(c2, A2.get(e), v);
            v = fma(c3, A3.get(e), v);
            v += is_diag(e) ? c0 : 0.0;
            P.set(e, v);
          }
          if (cm.pr > 1) store_tiles(cm.buf2, acc);  // buf2 = X^4, the Horner left operand
        }
        for (int j = cm.pr - 2; j >= 0; --j) {
          __syncthreads();  // buf0 (X, then the previous P) is no longer read by any wave
          store_tiles(cm.buf0, P);
          __syncthreads();
          const double c0 = c3p_inv_fact[4 * j], c1 = c3p_inv_fact[4 * j + 1], c2 = c3p_inv_fact[4 * j + 2],
                       c3 = c3p_inv_fact[4 * j + 3];
#pragma unroll
          for (int e = 0; e < NE; ++e) {
            double v = c1 * X.get(e);
            v = fma(c2, A2.get(e), v);
            v = fma(c3, A3.get(e), v);
            v += is_diag(e) ? c0 : 0.0;
            acc.set(e, v);
          }
          product(cm.buf2, cm.buf0, acc);
          P = acc;
        }
      }
      }  // VARIANT != 2
      // ---- squarings ----
      for (int it = 0; it < cm.ps; ++it) {
        __syncthreads();
        store_tiles(cm.buf0, P);
        __syncthreads();
        zero(acc);
        product(cm.buf0, cm.buf0, acc);
        P = acc;
      }
    }
    // ---- partial propagator write-out ----
    if constexpr (DUS) {
      double sn, cs;
      sincos(mu_i, &sn, &cs);
      const double er = exp(mu_r);
      double* dst = reinterpret_cast<double*>(A.dUs_out) + ((long)cm.sample * A.N + cm.n0 + t) * D * D * 2;
      store_plain(dst, P, er * cs, er * sn, nullptr);
    }
    // ---- chain: U <- E U (or U E for the right-ordered list product) ----
    if (t == 0) {
      U = P;
      mus_r = mu_r;
      mus_i = c3p_phase_add(0.0, mu_i);
      __syncthreads();  // buf0 is rewritten by the next slice
    } else {
      // buf0 <- E, buf1 <- U (both free now; make sure every wave has left the last product)
      __syncthreads();
      store_tiles(cm.buf0, P);
      store_tiles(cm.buf1, U);
      __syncthreads();
      Regs cacc;
      zero(cacc);
      if (GIVEN && A.right_order)
        product(cm.buf1, cm.buf0, cacc);
      else
        product(cm.buf0, cm.buf1, cacc);
      U = cacc;
      mus_r += mu_r;
      mus_i = c3p_phase_add(mus_i, mu_i);
      __syncthreads();  // buf0 / buf1 are rewritten by the next slice
    }
  }
  };
  if (cm.t18 == 2)
    slice_loop(std::integral_constant<int, 2>{});
  else if (cm.t18)
    slice_loop(std::integral_constant<int, 1>{});
  else
    slice_loop(std::integral_constant<int, 0>{});
  // ---- segment result: scalar e^{sum mu}, optional row phases ----
  double sn, cs;
  sincos(mus_i, &sn, &cs);
  const double er = exp(mus_r);
  double* dst = reinterpret_cast<double*>(A.seg_out) + chain * D * D * 2;
  const double* ph = A.fr_phase ? A.fr_phase + (long)cm.sample * D : nullptr;
  store_plain(dst, U, er * cs, er * sn, ph);
}

template <int NIG, int W>
struct MidOcc {
  static constexpr int IMG_BYTES = 16 * NIG * W * 8;
  static constexpr int WGS = (3 * IMG_BYTES + 6144) * 3 <= 160 * 1024 ? 3 : 2;
};

// REAL = true: the real-Hamiltonian kernel (own LDS layout, register budget and occupancy).  For unitary-mode
// calls both kernels are launched; a workgroup whose sample is (not) real leaves the (real) complex one at once.
template <int NIG, int NJ, int W, bool GIVEN, bool DUS, bool XG = false, bool REAL = false>
__global__ void __launch_bounds__(256, (REAL ? (Sched<MDR<NIG, W>::NIGR, NJ>::PW ? (C3P_PW_WGS) : MDR<NIG, W>::WGS) : MidOcc<NIG, W>::WGS))
    midd_chain_kernel(MidArgs A) {
  using C = MD<NIG, NJ>;
  constexpr int IMG = C::ROWS * W;
  const int tid = threadIdx.x;
  MidCommon cm;
  cm.lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  cm.r = cm.lane >> 4;
  cm.b = (cm.lane >> 2) & 3;
  cm.c = cm.lane & 3;
  cm.D = A.Dm;
  cm.nbk = (2 * cm.D + 3) / 4;
  cm.nbkR = (cm.D + 3) / 4;
  cm.K = A.K;
  const int K = A.K;
  constexpr int AREA = REAL ? mdr_area<NIG, NJ, W>() : 3 * IMG;  // image area: 3 complex images or the real pipeline's

  cm.buf0 = c3p_md_lds;
  cm.buf1 = cm.buf0 + IMG;
  cm.buf2 = cm.buf1 + IMG;
  cm.sg = cm.buf0 + AREA;  // K x SGC control amplitudes (a chunk of the segment)
  __shared__ double red[NW];

  const long chain = blockIdx.x;
  cm.sample = (int)(chain / A.S);
  cm.tabs = XG ? nullptr : A.tables + (long)(A.tab_per_sample ? cm.sample : 0) * (1 + K) * (IMG + 4);
  cm.realH = 0;
  if constexpr (!GIVEN && !XG) {
    // every table purely imaginary (real Hamiltonian, unitary mode): real cos / sin kernel
    bool realH = A.mode == C3P_MODE_UNITARY && !A.no_real;
    for (int k = 0; k <= K; ++k) realH = realH && (cm.tabs[(long)k * (IMG + 4) + IMG + 3] == 0.0);
    cm.realH = __builtin_amdgcn_readfirstlane((int)realH);
    if ((cm.realH != 0) != REAL) return;
  }
  const int seg = (int)(chain - (long)cm.sample * A.S);
  cm.n0 = (int)(((long)seg * A.N) / A.S);
  const int n1 = (int)(((long)(seg + 1) * A.N) / A.S);
  cm.len = n1 - cm.n0;

  // lane geometry
  cm.aoff = (4 * cm.b + (cm.c & ~1) + ((cm.c ^ cm.r) & 1)) * W + (cm.r >> 1);
  cm.aoffR = (4 * cm.b + cm.c) * W + cm.r;  // real image: row 4b + c of the 16-row group, column r of the K-step
  cm.boff = cm.r * W + cm.c;
  cm.lsmall = (4 * cm.b + cm.r) * W + cm.c;  // element of a 16x4 unit: row 4b + r, column c
  cm.lbig = cm.r * W + 4 * cm.b + cm.c;      // element of a 16x16 unit (register q adds 4q rows): row r, column 4b + c
  cm.negmask = (((cm.c & 1) == 0) && ((cm.r & 1) == 1)) ? 0x80000000u : 0u;
  if constexpr (REAL) {
    constexpr int WI = MDR<NIG, W>::WI;
    // swizzled layout: element (row, col) lives at row * 32 + (col ^ 16 (row & 1)); every unit's rows have the
    // parity of the lane's r, so the XOR is +16 for columns 0..15 and -16 for columns 16..31 on odd-r lanes
    const int sw = MDR<NIG, W>::SWZ ? 16 * (cm.r & 1) : 0;
    cm.aoffR = (4 * cm.b + cm.c) * WI + cm.r;
    cm.rbig[0] = cm.r * WI + 4 * cm.b + cm.c + sw;
    cm.rbig[1] = cm.r * WI + 4 * cm.b + cm.c - sw;
    cm.rsm[0] = (4 * cm.b + cm.r) * WI + cm.c + sw;
    cm.rsm[1] = (4 * cm.b + cm.r) * WI + cm.c - sw;
    cm.rblk[0] = cm.r * WI + cm.c + sw;
    cm.rblk[1] = cm.r * WI + cm.c - sw;
    cm.rb12 = cm.r * WI + ((12 + 4 * cm.b + cm.c) ^ sw);
    cm.rctr = (4 * cm.b + cm.r) * WI + ((12 + cm.c) ^ sw ^ (4 * cm.b));
  }

  // zero all images once (padding rows/columns must stay zero)
  for (int e = tid; e < AREA; e += 256) c3p_md_lds[e] = 0.0;
  __syncthreads();

  cm.pr = 1;
  cm.ps = 0;
  cm.t18 = 0;
  cm.t18n = 0;
  cm.scale = 1.0;
  if constexpr (!GIVEN) {
    // segment-wide plan from ||G0|| + sum_k max_t |c_k(t)| ||G_k||  (XG: max of the per-slice norms)
    double nrm = 0.0;
    if constexpr (XG) {
      const double* mt = A.meta + ((long)cm.sample * A.N + cm.n0) * 4;
      for (int t = tid; t < cm.len; t += 256) nrm = fmax(nrm, mt[(long)t * 4 + 2]);
      for (int o = 32; o >= 1; o >>= 1) nrm = fmax(nrm, __shfl_xor(nrm, o));
      if (cm.lane == 0) red[wave] = nrm;
      __syncthreads();
      nrm = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
      __syncthreads();
    } else {
      nrm = cm.tabs[IMG + 2];
    }
    for (int k = 0; k < (XG ? 0 : K); ++k) {
      const double* s = A.signals + ((long)cm.sample * K + k) * A.N + cm.n0;
      double cmax = 0.0;
      for (int t = tid; t < cm.len; t += 256) {
        const double v = s[t];
        cmax = fmax(cmax, fabs(v));
      }
      for (int o = 32; o >= 1; o >>= 1) cmax = fmax(cmax, __shfl_xor(cmax, o));
      if (cm.lane == 0) red[wave] = cmax;
      __syncthreads();
      cmax = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
      __syncthreads();
      nrm = fma(cmax, cm.tabs[(long)(k + 1) * (IMG + 4) + IMG + 2], nrm);
    }
    nrm = md_rfl(nrm);
    // round 6: Hermitian Hamiltonians (every table flagged skew-Hermitian by the prep kernel) -> T18 with the economised
    // parameters, radius 2.0 instead of 1.13
    bool normalG = !GIVEN && !XG && (A.mode == C3P_MODE_UNITARY) && !(A.no_t18n & 1);
    if constexpr (!GIVEN && !XG)
      for (int k = 0; k <= K; ++k) normalG = normalG && (cm.tabs[(long)k * (IMG + 4) + IMG + 3] <= 0.0);  // negative: complex skew-Hermitian; zero: real symmetric Hamiltonian
    cm.t18n = __builtin_amdgcn_readfirstlane((int)normalG);
    const MfmaPlan p = c3p_pick_plan_mfma(nrm, cm.t18n ? C3P_T18N_THETA : C3P_T18_THETA, cm.t18n != 0 && !(A.no_t18n & 2));
    cm.pr = __builtin_amdgcn_readfirstlane(p.r);
    cm.ps = __builtin_amdgcn_readfirstlane(p.s);
    cm.t18 = __builtin_amdgcn_readfirstlane(p.t18);
    if (A.no_t18 && cm.t18) {
      const TaylorPlan q = c3p_pick_plan_q4(nrm);
      cm.t18 = 0;
      cm.pr = __builtin_amdgcn_readfirstlane(q.r);
      cm.ps = __builtin_amdgcn_readfirstlane(q.s);
    }
    if constexpr (REAL) {
      // Taylor degree 16 / 18 / 20 of cos (theta = 0.816 / 1.13 / 1.49: backward error below 2^-53) with s squarings:
      // 7 / 8 / 8 + 2 s real products; the cheapest, the lower degree on ties (variant kept in t18: 1 / 0 / 2)
      auto squarings = [&](double theta) {
        int s = 0;
        while (theta < nrm && s < 40) {
          theta *= 2.0;
          ++s;
        }
        return s;
      };
      // Round 6: the 7-product variant evaluates the Chebyshev-economised degree-8 polynomials (theta = 1.85: Y is real symmetric,
      // so the scalar error on [0, theta^2] is the matrix error) -- it dominates the degree-18 / 20 Taylor variants below
      const int s16 = squarings(C3P_MM8_THETA), s18 = squarings(C3P_T18_THETA), s20 = squarings(1.49);
      int var = 1, s = s16, cost = 7 + 2 * s16;
      if (8 + 2 * s18 < cost) var = 0, s = s18, cost = 8 + 2 * s18;
      // (degree 20 only in the 16- and 32-row classes: at D >= 33 its extra dependent product and two more live tile sets
      // cost more than the squaring they save -- cfg5 measured 3 % slower with it, cfg3 3.6 % faster)
      if (MDR<NIG, W>::NIGR <= 2 && 8 + 2 * s20 < cost) var = 2, s = s20, cost = 8 + 2 * s20;
      cm.ps = __builtin_amdgcn_readfirstlane(s);
      cm.t18 = __builtin_amdgcn_readfirstlane(var);
    }
    cm.scale = ldexp(1.0, -cm.ps);
    __syncthreads();
  }
  if constexpr (REAL) {
    // (pinwheel class: the role that also carries the centre block moves with the workgroup, so that the workgroups of a CU
    // do not put it on one SIMD)
    const int role = (Sched<MDR<NIG, W>::NIGR, NJ>::PW && C3P_PW_ROT) ? ((wave + (int)(blockIdx.x % (C3P_PW_ROT ? C3P_PW_ROT : 1))) & 3) : wave;
    switch (role) {
      case 0: midd_real_body<NIG, NJ, W, DUS, 0>(A, cm, chain); break;
      case 1: midd_real_body<NIG, NJ, W, DUS, 1>(A, cm, chain); break;
      case 2: midd_real_body<NIG, NJ, W, DUS, 2>(A, cm, chain); break;
      default: midd_real_body<NIG, NJ, W, DUS, 3>(A, cm, chain); break;
    }
  } else {
    switch (wave) {
      case 0: midd_body<NIG, NJ, W, GIVEN, DUS, XG, 0>(A, cm, chain); break;
      case 1: midd_body<NIG, NJ, W, GIVEN, DUS, XG, 1>(A, cm, chain); break;
      case 2: midd_body<NIG, NJ, W, GIVEN, DUS, XG, 2>(A, cm, chain); break;
      default: midd_body<NIG, NJ, W, GIVEN, DUS, XG, 3>(A, cm, chain); break;
    }
  }
}

// Tables for the mid-D kernel: one workgroup per (sample, table).  Image = Zh layout of
// G - mu I (G = -i dt h, or the Lindblad generator pieces), ROWS x W doubles zero padded,
// followed by {Re mu, Im mu, ||G - mu I||_1, 0}.
__global__ void __launch_bounds__(256) midd_prep_kernel(MidPrepArgs P) {
  __shared__ double redr[256], redi[256], reda[256];
  __shared__ double mu[2];
  const int tid = threadIdx.x;
  const int ti = blockIdx.x % (1 + P.K);
  const int sample = blockIdx.x / (1 + P.K);
  const int D = P.Dm, Dh = P.Dh;
  const cplx* h = (ti == 0) ? P.h0 + (long)sample * P.h0_bstride
                            : P.hks + (long)sample * P.hks_bstride + (long)(ti - 1) * Dh * Dh;
  auto gelem = [&](int row_, int col_) -> cplx {
    const int row = P.conjT ? col_ : row_, col = P.conjT ? row_ : col_;  // G^H: element (r, c) = conj(G[c][r])
    cplx v;
    if (!P.lindblad) {
      const cplx x = h[row * D + col];
      v = cmake(x.y * P.dt, -x.x * P.dt);
    } else {
      const int i = row / Dh, j = row - i * Dh, k = col / Dh, l = col - k * Dh;
      v = (ti == 0) ? P.clp[(long)row * D + col] : cmake(0, 0);
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
    if (P.conjT) v.y = -v.y;
    return v;
  };
  double tr = 0, tim = 0;
  for (int i = tid; i < D; i += 256) {
    const cplx v = gelem(i, i);
    tr += v.x;
    tim += v.y;
  }
  redr[tid] = tr;
  redi[tid] = tim;
  __syncthreads();
  if (tid == 0) {
    double a = 0, bq = 0;
    for (int i = 0; i < 256; ++i) {
      a += redr[i];
      bq += redi[i];
    }
    mu[0] = 0.0;  // imaginary shift only: a real one overflows the shifted product of long, strongly damped segments (c3p_smalld.hip: build_tables)
    mu[1] = bq / D;
  }
  __syncthreads();
  double cs = 0, remax = 0, asym = 0, rsk = 0;
  for (int j = tid; j < D; j += 256) {
    double s = 0;
    for (int i = 0; i < D; ++i) {
      cplx v = gelem(i, j);
      if (i == j) {
        v.x -= mu[0];
        v.y -= mu[1];
      }
      s += hypot(v.x, v.y);
      remax = fmax(remax, fabs(v.x));
      if (i < j) {
        const cplx w = gelem(j, i);
        asym = fmax(asym, fabs(v.y - w.y));
        rsk = fmax(rsk, fabs(v.x + w.x));  // G + G^H: Re part antisymmetric, Im part symmetric <=> the Hamiltonian is Hermitian
      } else if (i == j) {
        rsk = fmax(rsk, fabs(v.x));
      }
    }
    cs = fmax(cs, s);
  }
  redr[tid] = cs;
  redi[tid] = remax;
  reda[tid] = asym;
  __shared__ double redk[256];
  redk[tid] = rsk;
  __syncthreads();
  const int IMG = P.tile_nig ? P.tile_nig * P.tile_nj * 64 : P.rows * P.W;
  double* out = P.tables + ((long)sample * (1 + P.K) + ti) * (IMG + 4);
  for (int e = tid; e < IMG; e += 256) {
    int rho, col;
    if (P.tile_nig) {
      // tile-major image (c3p_bigd.hip): tile (J,Ig) = 64 doubles in MFMA C/D lane order
      const int tile = e >> 6, l = e & 63;
      const int J = tile / P.tile_nig, Ig = tile - J * P.tile_nig;
      rho = 16 * Ig + 4 * ((l >> 2) & 3) + (l >> 4);
      col = 4 * J + (l & 3);
    } else {
      rho = e / P.W;
      col = e - rho * P.W;
    }
    const int i = rho >> 1, p = rho & 1;
    double v = 0.0;
    if (i < D && col < D) {
      cplx g = gelem(i, col);
      if (i == col) {
        g.x -= mu[0];
        g.y -= mu[1];
      }
      v = p ? g.y : g.x;
    }
    out[e] = v;
  }
  if (tid == 0) {
    double nrm = 0, re = 0;
    for (int i = 0; i < 256; ++i) {
      nrm = fmax(nrm, redr[i]);
      re = fmax(re, redi[i]);
    }
    // the real path reads its left operands through their transposes: the flag also demands a symmetric Im part
    // (to rounding -- dressed operators V^T H V are symmetric to ~5e-16 relative, as in the small-D kernel)
    double as = 0;
    for (int i = 0; i < 256; ++i) as = fmax(as, reda[i]);
    if (as > 1e-14 * nrm) re = fmax(re, as);
    double sk = as;
    for (int i = 0; i < 256; ++i) sk = fmax(sk, redk[i]);
    out[IMG + 0] = mu[0];
    out[IMG + 1] = mu[1];
    out[IMG + 2] = nrm;
    // 0: G purely imaginary (real symmetric Hamiltonian) -> real path of the mid-D kernel; NEGATIVE: complex but skew-Hermitian to
    // rounding (Hermitian Hamiltonian: a normal generator, the economised T18 parameters apply); positive: anything else
    out[IMG + 3] = P.lindblad ? 1.0 : ((re > 0.0 && sk <= 1e-14 * nrm && !P.conjT) ? -re : re);
  }
}

template <typename Kern>
hipError_t md_go(Kern kern, const MidArgs& A, size_t bytes, hipStream_t st) {
  if (bytes > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)bytes);
    if (e != hipSuccess) return e;
  }
  C3P_LAUNCH(kern, dim3((unsigned)((long)A.B * A.S)), dim3(256), bytes, st, A);
  return hipGetLastError();
}

#if C3P_MIDD_HAS(1)
template <int NIG, int NJ, int W>
hipError_t launch_t(const MidArgs& A, hipStream_t st) {
  constexpr int IMG = MD<NIG, NJ>::ROWS * W;
  const size_t lds = (size_t)(3 * IMG + (A.mode == C3P_MODE_GIVEN ? 0 : A.K * SGC)) * sizeof(double);
  if (A.mode == C3P_MODE_GIVEN) return md_go(midd_chain_kernel<NIG, NJ, W, true, false>, A, lds, st);
  if (A.mode == C3P_MODE_EXPM)
    return A.dUs_out ? md_go(midd_chain_kernel<NIG, NJ, W, false, true, true>, A, lds, st)
                     : md_go(midd_chain_kernel<NIG, NJ, W, false, false, true>, A, lds, st);
  if (A.mode == C3P_MODE_UNITARY && !A.no_real) {
    // samples with real Hamiltonians are taken by the real kernel, the others by the complex one
    hipError_t e = c3p_launch_midd_real(A, st);
    if (e != hipSuccess) return e;
  }
  if (A.dUs_out) return md_go(midd_chain_kernel<NIG, NJ, W, false, true>, A, lds, st);
  return md_go(midd_chain_kernel<NIG, NJ, W, false, false>, A, lds, st);
}
#endif

#if C3P_MIDD_HAS(2)
template <int NIG, int NJ, int W>
hipError_t launch_real_t(const MidArgs& A, hipStream_t st) {
  const size_t ldsr = (size_t)(mdr_area<NIG, NJ, W>() + A.K * SGC) * sizeof(double);
  return A.dUs_out ? md_go(midd_chain_kernel<NIG, NJ, W, false, true, false, true>, A, ldsr, st)
                   : md_go(midd_chain_kernel<NIG, NJ, W, false, false, false, true>, A, ldsr, st);
}
#endif

// ---------------------------------------------------------------------------------------------
// Backward sweep of the control gradient (method in c3p_grad.hip) in the mid-D layout.
// Four LDS images (value / derivative left and right operands of the pair products) and every tile
// set in registers: one workgroup per CU (1 wave per SIMD, up to 512 registers).  Per slice 18 + 3s
// products:  (T, dT) = pair-T18(X~, M);  Z = T^H dT;  grad[k] = <Z, G_k>;  M <- T^H (M T).
// ---------------------------------------------------------------------------------------------
template <int NIG, int NJ, int W, int WV>
__device__ __forceinline__ void midd_grad_body(const MidGradArgs& A, const MidCommon& cm, long chain, double* i0,
                                               double* i1, double* i2, double* i3, double* red) {
  using T = WaveTiles<NIG, NJ, W, WV>;
  constexpr int IMG = MD<NIG, NJ>::ROWS * W, NE = T::NE;
  typedef TileRegs<T::NBW, T::NSW> Regs;
  const int D = cm.D, K = cm.K;
  const int lbig = cm.lbig, lsmall = cm.lsmall;
  const int rbig = cm.r, rsmall = 4 * cm.b + cm.r;
  const int cbig = 4 * cm.b + cm.c, csmall = cm.c;
  const double* tabs = cm.tabs;
  auto eoff = [&](int e) -> int { return T::off0(e) + (T::is_big(e) ? lbig : lsmall); };
  auto erow = [&](int e) -> int { return T::row0(e) + (T::is_big(e) ? rbig : rsmall); };
  auto ecol = [&](int e) -> int { return T::col0(e) + (T::is_big(e) ? cbig : csmall); };
  auto store_tiles = [&](double* img, const Regs& v) {
#pragma unroll
    for (int e = 0; e < NE; ++e) img[eoff(e)] = v.get(e);
  };
  // image of the conjugate transpose: Yh[2j + p][i] = p ? -Im Z[i][j] : Re Z[i][j]
  auto store_tiles_H = [&](double* img, const Regs& v) {
#pragma unroll
    for (int e = 0; e < NE; ++e) {
      const int row = erow(e), col = ecol(e);
      const int ci = row >> 1, p = row & 1;
      if (ci < D && col < D) img[(2 * col + p) * W + ci] = p ? -v.get(e) : v.get(e);
    }
  };
  auto zero = [&](Regs& v) {
#pragma unroll
    for (int e = 0; e < NE; ++e) v.set(e, 0.0);
  };
  auto product = [&](const double* imgA, const double* imgB, Regs& acc) { mm_tiles_pf<NIG, NJ, W, WV, 3>(imgA, imgB, cm, acc); };
  auto is_diag = [&](int e) -> bool {
    const int row = erow(e), col = ecol(e);
    return ((row & 1) == 0) && ((row >> 1) == col) && (col < D);
  };
  auto comb = [&](Regs& out, double c0, double cx, double c2, double c3, double c6, const Regs& X, const Regs& A2,
                  const Regs& A3, const Regs& A6) {
#pragma unroll
    for (int e = 0; e < NE; ++e) {
      double v = cx * X.get(e);
      v = fma(c2, A2.get(e), v);
      v = fma(c3, A3.get(e), v);
      v = fma(c6, A6.get(e), v);
      v += (c0 != 0.0 && is_diag(e)) ? c0 : 0.0;
      out.set(e, v);
    }
  };

  Regs M;
  {
    const double* src = reinterpret_cast<const double*>(A.Mb) + chain * D * D * 2;
#pragma unroll
    for (int e = 0; e < NE; ++e) {
      const int row = erow(e), col = ecol(e);
      const int ci = row >> 1;
      M.set(e, (ci < D && col < D) ? src[(ci * D + col) * 2 + (row & 1)] : 0.0);
    }
  }
  for (int t = cm.len - 1; t >= 0; --t) {
    Regs X, dX;
#pragma unroll
    for (int e = 0; e < NE; ++e) {
      X.set(e, cm.scale * tabs[eoff(e)]);
      dX.set(e, cm.scale * M.get(e));
    }
    for (int k = 0; k < K; ++k) {
      const double ck = cm.scale * cm.sg[k * A.Lmax + t];
      const double* tk = tabs + (long)(k + 1) * (IMG + 4);
#pragma unroll
      for (int e = 0; e < NE; ++e) X.set(e, fma(ck, tk[eoff(e)], X.get(e)));
    }
    __syncthreads();  // the previous slice's last product has left the images
    store_tiles(i0, X);
    store_tiles(i1, dX);
    __syncthreads();
    Regs A2, dA2, A3, dA3, A6, dA6;
    zero(A2), zero(dA2), zero(A3), zero(dA3), zero(A6), zero(dA6);
    product(i0, i0, A2);
    product(i0, i1, dA2);
    product(i1, i0, dA2);
    store_tiles(i2, A2);
    store_tiles(i3, dA2);
    __syncthreads();
    product(i0, i2, A3);
    product(i0, i3, dA3);
    product(i1, i2, dA3);
    __syncthreads();
    store_tiles(i0, A3);
    store_tiles(i1, dA3);
    __syncthreads();
    product(i0, i0, A6);
    product(i0, i1, dA6);
    product(i1, i0, dA6);
    Regs A9, dA9;
    {
      Regs B1, dB1, B5, dB5;
      comb(B1, 0.0, C3P_T18_A11, C3P_T18_A21, C3P_T18_A31, 0.0, X, A2, A3, A6);
      comb(dB1, 0.0, C3P_T18_A11, C3P_T18_A21, C3P_T18_A31, 0.0, dX, dA2, dA3, dA6);
      comb(B5, 0.0, 0.0, C3P_T18_B24, C3P_T18_B34, C3P_T18_B64, X, A2, A3, A6);
      comb(dB5, 0.0, 0.0, C3P_T18_B24, C3P_T18_B34, C3P_T18_B64, dX, dA2, dA3, dA6);
      __syncthreads();
      store_tiles(i0, B1);
      store_tiles(i1, dB1);
      store_tiles(i2, B5);
      store_tiles(i3, dB5);
    }
    comb(A9, C3P_T18_B03, C3P_T18_B13, C3P_T18_B23, C3P_T18_B33, C3P_T18_B63, X, A2, A3, A6);
    comb(dA9, 0.0, C3P_T18_B13, C3P_T18_B23, C3P_T18_B33, C3P_T18_B63, dX, dA2, dA3, dA6);
    __syncthreads();
    product(i0, i2, A9);
    product(i0, i3, dA9);
    product(i1, i2, dA9);
    Regs Tm, dT;
    {
      Regs L, dL;
      comb(L, C3P_T18_B02, C3P_T18_B12, C3P_T18_B22, C3P_T18_B32, C3P_T18_B62, X, A2, A3, A6);
      comb(dL, 0.0, C3P_T18_B12, C3P_T18_B22, C3P_T18_B32, C3P_T18_B62, dX, dA2, dA3, dA6);
#pragma unroll
      for (int e = 0; e < NE; ++e) {
        L.set(e, L.get(e) + A9.get(e));
        dL.set(e, dL.get(e) + dA9.get(e));
      }
      __syncthreads();
      store_tiles(i0, L);
      store_tiles(i1, dL);
      store_tiles(i2, A9);
      store_tiles(i3, dA9);
    }
    comb(Tm, 0.0, C3P_T18_B11, C3P_T18_B21, C3P_T18_B31, C3P_T18_B61, X, A2, A3, A6);
    comb(dT, 0.0, C3P_T18_B11, C3P_T18_B21, C3P_T18_B31, C3P_T18_B61, dX, dA2, dA3, dA6);
    __syncthreads();
    product(i0, i2, Tm);
    product(i0, i3, dT);
    product(i1, i2, dT);
    for (int it = 0; it < cm.ps; ++it) {
      __syncthreads();
      store_tiles(i0, Tm);
      store_tiles(i1, dT);
      __syncthreads();
      Regs T2, dT2;
      zero(T2), zero(dT2);
      product(i0, i0, T2);
      product(i0, i1, dT2);
      product(i1, i0, dT2);
      Tm = T2;
      dT = dT2;
    }
    // ---- Z = T^H dT, V = M T ----
    __syncthreads();
    store_tiles_H(i0, Tm);
    store_tiles(i1, dT);
    store_tiles(i2, M);
    store_tiles(i3, Tm);
    __syncthreads();
    Regs Z, V;
    zero(Z), zero(V);
    product(i0, i1, Z);
    product(i2, i3, V);
    if (A.zout != nullptr) {  // per-slice cotangent of the generator G_n = -i dt H_n
      double* dst = reinterpret_cast<double*>(A.zout) + ((long)cm.sample * A.N + cm.n0 + t) * D * D * 2;
#pragma unroll
      for (int e = 0; e < NE; ++e) {
        const int row = erow(e), col = ecol(e);
        const int ci = row >> 1;
        if (ci < D && col < D) dst[(ci * D + col) * 2 + (row & 1)] = Z.get(e);
      }
    }
    // ---- grad[k] = sum Zh . G~_k h + Re(mu_k conj(tr Z)) ----
    {
      double trr = 0.0, tri = 0.0;
#pragma unroll
      for (int e = 0; e < NE; ++e) {
        const int row = erow(e), col = ecol(e);
        const bool dg = ((row >> 1) == col) && (col < D);
        const double v = dg ? Z.get(e) : 0.0;
        if (row & 1)
          tri += v;
        else
          trr += v;
      }
      for (int o = 32; o >= 1; o >>= 1) {
        trr += __shfl_xor(trr, o);
        tri += __shfl_xor(tri, o);
      }
      for (int k = 0; k < K; ++k) {
        const double* tk = tabs + (long)(k + 1) * (IMG + 4);
        double part = fma(tk[IMG + 0], trr, tk[IMG + 1] * tri);
        double dot = 0.0;
#pragma unroll
        for (int e = 0; e < NE; ++e) dot = fma(Z.get(e), tk[eoff(e)], dot);
        for (int o = 32; o >= 1; o >>= 1) dot += __shfl_xor(dot, o);
        if (cm.lane == 0) red[WV * 16 + k] = dot + part;
      }
    }
    __syncthreads();  // Z / V products done in every wave, partial sums visible
    if (WV == 0 && cm.lane < K)
      A.grad[((long)cm.sample * K + cm.lane) * A.N + cm.n0 + t] =
          red[cm.lane] + red[16 + cm.lane] + red[32 + cm.lane] + red[48 + cm.lane];
    if (t > 0) {
      store_tiles(i1, V);
      __syncthreads();
      zero(M);
      product(i0, i1, M);  // M <- T^H (M T)
    }
  }
}

// real-Hamiltonian backward sweep (midd_grad_real_kernel below): squarings kept, image slots, scaling rule (theta_16)
constexpr int MGR_SLOTS = 8;
// squaring levels kept in registers: two, one for the 48-row classes (D >= 33, register-bound)
template <int NIG>
struct MGR {
  static constexpr int MAXS = (NIG + 1) / 2 >= 3 ? 1 : 2;
};

__device__ __forceinline__ int mgr_squarings(double nrm) {
  int ps = 0;
  double p = C3P_MM8_THETA;  // round 6: the economised degree-8 pair (c3p_common.h) instead of theta_16 = 0.816
  while (p < nrm && ps < 40) {
    p *= 2.0;
    ++ps;
  }
  return ps;
}

template <int NIG, int NJ, int W>
__global__ void __launch_bounds__(256, 1) midd_grad_kernel(MidGradArgs A) {
  using C = MD<NIG, NJ>;
  constexpr int IMG = C::ROWS * W;
  const int tid = threadIdx.x;
  MidCommon cm;
  cm.lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  cm.r = cm.lane >> 4;
  cm.b = (cm.lane >> 2) & 3;
  cm.c = cm.lane & 3;
  cm.D = A.Dm;
  cm.nbk = (2 * cm.D + 3) / 4;
  cm.K = A.K;
  const int K = A.K;
  double* i0 = c3p_md_lds;
  double* i1 = i0 + IMG;
  double* i2 = i1 + IMG;
  double* i3 = i2 + IMG;
  cm.sg = i3 + IMG;
  __shared__ double red[64];
  __shared__ double redn[NW];
  const long chain = blockIdx.x;
  cm.sample = (int)(chain / A.S);
  const int seg = (int)(chain - (long)cm.sample * A.S);
  cm.n0 = (int)(((long)seg * A.N) / A.S);
  const int n1 = (int)(((long)(seg + 1) * A.N) / A.S);
  cm.len = n1 - cm.n0;
  cm.aoff = (4 * cm.b + (cm.c & ~1) + ((cm.c ^ cm.r) & 1)) * W + (cm.r >> 1);
  cm.boff = cm.r * W + cm.c;
  cm.lsmall = (4 * cm.b + cm.r) * W + cm.c;
  cm.lbig = cm.r * W + 4 * cm.b + cm.c;
  cm.negmask = (((cm.c & 1) == 0) && ((cm.r & 1) == 1)) ? 0x80000000u : 0u;
  for (int e = tid; e < 4 * IMG; e += 256) c3p_md_lds[e] = 0.0;
  __syncthreads();
  cm.tabs = A.tables + (long)(A.tab_per_sample ? cm.sample : 0) * (1 + K) * (IMG + 4);
  double nrm = cm.tabs[IMG + 2];
  for (int k = 0; k < K; ++k) {
    const double* s = A.signals + ((long)cm.sample * K + k) * A.N + cm.n0;
    double cmax = 0.0;
    for (int t = tid; t < cm.len; t += 256) {
      const double v = s[t];
      cm.sg[k * A.Lmax + t] = v;
      cmax = fmax(cmax, fabs(v));
    }
    for (int o = 32; o >= 1; o >>= 1) cmax = fmax(cmax, __shfl_xor(cmax, o));
    if (cm.lane == 0) redn[wave] = cmax;
    __syncthreads();
    cmax = fmax(fmax(redn[0], redn[1]), fmax(redn[2], redn[3]));
    __syncthreads();
    nrm = fma(cmax, cm.tabs[(long)(k + 1) * (IMG + 4) + IMG + 2], nrm);
  }
  nrm = md_rfl(nrm);
  int ps = 0;
  {
    double p = C3P_T18_THETA;
    while (p < nrm && ps < 40) {
      p *= 2.0;
      ++ps;
    }
  }
  cm.ps = __builtin_amdgcn_readfirstlane(ps);
  if (A.skip_real) {  // the real-Hamiltonian sweep has taken this chain (same tables, same norm bound: same decision)
    bool realH = K <= MDR<NIG, W>::KP;
    for (int k = 0; k <= K; ++k) realH = realH && (cm.tabs[(long)k * (IMG + 4) + IMG + 3] == 0.0);
    if (__builtin_amdgcn_readfirstlane((int)realH) != 0 && __builtin_amdgcn_readfirstlane(mgr_squarings(nrm)) <= MGR<NIG>::MAXS) return;
  }
  cm.pr = 0;
  cm.t18 = 1;
  cm.scale = ldexp(1.0, -cm.ps);
  cm.buf0 = i0;
  cm.buf1 = i1;
  cm.buf2 = i2;
  __syncthreads();
  switch (wave) {
    case 0: midd_grad_body<NIG, NJ, W, 0>(A, cm, chain, i0, i1, i2, i3, red); break;
    case 1: midd_grad_body<NIG, NJ, W, 1>(A, cm, chain, i0, i1, i2, i3, red); break;
    case 2: midd_grad_body<NIG, NJ, W, 2>(A, cm, chain, i0, i1, i2, i3, red); break;
    default: midd_grad_body<NIG, NJ, W, 3>(A, cm, chain, i0, i1, i2, i3, red); break;
  }
}

// ---------------------------------------------------------------------------------------------
// Backward sweep for GENERAL generators in the mid-D layout (Lindblad superoperators of two qubits, 16 x 16, and of
// D = 5 / 6): method of c3p_grad.hip's general form, layout of midd_grad_body.  Forward over the segment with the slice
// propagators of the forward pass (P <- dU_n P, every P_n stored), then backwards: M_n = A P_n^H, pair evaluation of T18 at
// Y = X_n^H (tables of G^H), grad[k] = Re(conj(e^{shift}) <dT, G_k>), A <- T A; the trace shifts ride along as one scalar.
// ---------------------------------------------------------------------------------------------
template <int NIG, int NJ, int W, int WV, bool XG>
__device__ __forceinline__ void midd_grad_general_body(const MidGradArgs& A, const MidCommon& cm, long chain, double* i0,
                                                       double* i1, double* i2, double* i3, double* red) {
  using T = WaveTiles<NIG, NJ, W, WV>;
  constexpr int IMG = MD<NIG, NJ>::ROWS * W, NE = T::NE;
  typedef TileRegs<T::NBW, T::NSW> Regs;
  const int D = cm.D, K = XG ? 0 : cm.K;
  const int lbig = cm.lbig, lsmall = cm.lsmall;
  const int rbig = cm.r, rsmall = 4 * cm.b + cm.r;
  const int cbig = 4 * cm.b + cm.c, csmall = cm.c;
  const double* tabs = cm.tabs;                                                                    // G^H: the exponent
  const double* tabg = XG ? nullptr : A.tables + (long)(A.tab_per_sample ? cm.sample : 0) * (1 + K) * (IMG + 4);  // G: inner products
  auto eoff = [&](int e) -> int { return T::off0(e) + (T::is_big(e) ? lbig : lsmall); };
  auto erow = [&](int e) -> int { return T::row0(e) + (T::is_big(e) ? rbig : rsmall); };
  auto ecol = [&](int e) -> int { return T::col0(e) + (T::is_big(e) ? cbig : csmall); };
  auto store_tiles = [&](double* img, const Regs& v) {
#pragma unroll
    for (int e = 0; e < NE; ++e) img[eoff(e)] = v.get(e);
  };
  auto store_tiles_H = [&](double* img, const Regs& v) {
#pragma unroll
    for (int e = 0; e < NE; ++e) {
      const int row = erow(e), col = ecol(e);
      const int ci = row >> 1, p = row & 1;
      if (ci < D && col < D) img[(2 * col + p) * W + ci] = p ? -v.get(e) : v.get(e);
    }
  };
  auto zero = [&](Regs& v) {
#pragma unroll
    for (int e = 0; e < NE; ++e) v.set(e, 0.0);
  };
  auto load_plain = [&](Regs& v, const cplx* srcc) {
    const double* src = reinterpret_cast<const double*>(srcc);
#pragma unroll
    for (int e = 0; e < NE; ++e) {
      const int row = erow(e), col = ecol(e);
      const int ci = row >> 1;
      v.set(e, (ci < D && col < D) ? src[(ci * D + col) * 2 + (row & 1)] : 0.0);
    }
  };
  auto store_plain = [&](cplx* dstc, const Regs& v) {
    double* dst = reinterpret_cast<double*>(dstc);
#pragma unroll
    for (int e = 0; e < NE; ++e) {
      const int row = erow(e), col = ecol(e);
      const int ci = row >> 1;
      if (ci < D && col < D) dst[(ci * D + col) * 2 + (row & 1)] = v.get(e);
    }
  };
  auto product = [&](const double* imgA, const double* imgB, Regs& acc) { mm_tiles_pf<NIG, NJ, W, WV, 3>(imgA, imgB, cm, acc); };
  auto is_diag = [&](int e) -> bool {
    const int row = erow(e), col = ecol(e);
    return ((row & 1) == 0) && ((row >> 1) == col) && (col < D);
  };
  auto comb = [&](Regs& out, double c0, double cx, double c2, double c3, double c6, const Regs& X, const Regs& A2,
                  const Regs& A3, const Regs& A6) {
#pragma unroll
    for (int e = 0; e < NE; ++e) {
      double v = cx * X.get(e);
      v = fma(c2, A2.get(e), v);
      v = fma(c3, A3.get(e), v);
      v = fma(c6, A6.get(e), v);
      v += (c0 != 0.0 && is_diag(e)) ? c0 : 0.0;
      out.set(e, v);
    }
  };

  // ---- forward: the prefix in front of every slice (full values: the slice propagators carry their trace shifts) ----
  cplx* pst = A.pstore + ((long)cm.sample * A.N + cm.n0) * D * D;
  {
    Regs P;
    load_plain(P, A.pre + chain * D * D);
    const cplx* du = A.dUs + ((long)cm.sample * A.N + cm.n0) * D * D;
    for (int t = 0; t < cm.len; ++t) {
      store_plain(pst + (long)t * D * D, P);
      if (t + 1 == cm.len) break;
      Regs E, V;
      load_plain(E, du + (long)t * D * D);
      __syncthreads();
      store_tiles(i0, E);
      store_tiles(i1, P);
      __syncthreads();
      zero(V);
      product(i0, i1, V);
      P = V;
    }
  }
  __threadfence_block();
  __syncthreads();

  // ---- backward ----
  Regs Aa;  // left adjoint without the trace shifts of the slices behind it: they accumulate in (ams_r, ams_i)
  load_plain(Aa, A.Mb + chain * D * D);
  double ams_r = 0.0, ams_i = 0.0;
  for (int t = cm.len - 1; t >= 0; --t) {
    Regs X, dX;
    {
      Regs Pn, Mn;
      load_plain(Pn, pst + (long)t * D * D);
      __syncthreads();  // the previous slice's last product has left the images
      store_tiles(i0, Aa);
      store_tiles_H(i1, Pn);
      __syncthreads();
      zero(Mn);
      product(i0, i1, Mn);  // M_n = A P_n^H
#pragma unroll
      for (int e = 0; e < NE; ++e) dX.set(e, cm.scale * Mn.get(e));
    }
    double mu_r, mu_i;  // trace shift of Y = X_n^H
    if constexpr (XG) {
      // supplied generator: Y = conj(coef) hs[b,n]^H - conj(mu_n); element (ci, col) = conj(coef hs[col][ci])
      const long m = (long)cm.sample * A.N + cm.n0 + t;
      mu_r = A.meta[m * 4 + 0];
      mu_i = -A.meta[m * 4 + 1];
      const double2* src = reinterpret_cast<const double2*>(A.hs) + (long)cm.sample * A.hs_bstride + (long)(cm.n0 + t) * D * D;
#pragma unroll
      for (int e = 0; e < NE; ++e) {
        const int row = erow(e), col = ecol(e);
        const int ci = row >> 1;
        const bool in = ci < D && col < D;
        const double2 h = src[in ? col * D + ci : 0];
        double v = (row & 1) ? -fma(A.coef_r, h.y, A.coef_i * h.x) : fma(A.coef_r, h.x, -A.coef_i * h.y);
        v -= (ci == col) ? ((row & 1) ? mu_i : mu_r) : 0.0;
        X.set(e, in ? cm.scale * v : 0.0);
      }
    } else {
      mu_r = tabs[IMG + 0];
      mu_i = tabs[IMG + 1];
#pragma unroll
      for (int e = 0; e < NE; ++e) X.set(e, cm.scale * tabs[eoff(e)]);
    }
    for (int k = 0; k < K; ++k) {
      const double c0 = cm.sg[k * A.Lmax + t];
      const double ck = cm.scale * c0;
      const double* tk = tabs + (long)(k + 1) * (IMG + 4);
      mu_r = fma(c0, tk[IMG + 0], mu_r);
      mu_i = fma(c0, tk[IMG + 1], mu_i);
#pragma unroll
      for (int e = 0; e < NE; ++e) X.set(e, fma(ck, tk[eoff(e)], X.get(e)));
    }
    __syncthreads();
    store_tiles(i0, X);
    store_tiles(i1, dX);
    __syncthreads();
    Regs A2, dA2, A3, dA3, A6, dA6;
    zero(A2), zero(dA2), zero(A3), zero(dA3), zero(A6), zero(dA6);
    product(i0, i0, A2);
    product(i0, i1, dA2);
    product(i1, i0, dA2);
    store_tiles(i2, A2);
    store_tiles(i3, dA2);
    __syncthreads();
    product(i0, i2, A3);
    product(i0, i3, dA3);
    product(i1, i2, dA3);
    __syncthreads();
    store_tiles(i0, A3);
    store_tiles(i1, dA3);
    __syncthreads();
    product(i0, i0, A6);
    product(i0, i1, dA6);
    product(i1, i0, dA6);
    Regs A9, dA9;
    {
      Regs B1, dB1, B5, dB5;
      comb(B1, 0.0, C3P_T18_A11, C3P_T18_A21, C3P_T18_A31, 0.0, X, A2, A3, A6);
      comb(dB1, 0.0, C3P_T18_A11, C3P_T18_A21, C3P_T18_A31, 0.0, dX, dA2, dA3, dA6);
      comb(B5, 0.0, 0.0, C3P_T18_B24, C3P_T18_B34, C3P_T18_B64, X, A2, A3, A6);
      comb(dB5, 0.0, 0.0, C3P_T18_B24, C3P_T18_B34, C3P_T18_B64, dX, dA2, dA3, dA6);
      __syncthreads();
      store_tiles(i0, B1);
      store_tiles(i1, dB1);
      store_tiles(i2, B5);
      store_tiles(i3, dB5);
    }
    comb(A9, C3P_T18_B03, C3P_T18_B13, C3P_T18_B23, C3P_T18_B33, C3P_T18_B63, X, A2, A3, A6);
    comb(dA9, 0.0, C3P_T18_B13, C3P_T18_B23, C3P_T18_B33, C3P_T18_B63, dX, dA2, dA3, dA6);
    __syncthreads();
    product(i0, i2, A9);
    product(i0, i3, dA9);
    product(i1, i2, dA9);
    Regs Tm, dT;
    {
      Regs L, dL;
      comb(L, C3P_T18_B02, C3P_T18_B12, C3P_T18_B22, C3P_T18_B32, C3P_T18_B62, X, A2, A3, A6);
      comb(dL, 0.0, C3P_T18_B12, C3P_T18_B22, C3P_T18_B32, C3P_T18_B62, dX, dA2, dA3, dA6);
#pragma unroll
      for (int e = 0; e < NE; ++e) {
        L.set(e, L.get(e) + A9.get(e));
        dL.set(e, dL.get(e) + dA9.get(e));
      }
      __syncthreads();
      store_tiles(i0, L);
      store_tiles(i1, dL);
      store_tiles(i2, A9);
      store_tiles(i3, dA9);
    }
    comb(Tm, 0.0, C3P_T18_B11, C3P_T18_B21, C3P_T18_B31, C3P_T18_B61, X, A2, A3, A6);
    comb(dT, 0.0, C3P_T18_B11, C3P_T18_B21, C3P_T18_B31, C3P_T18_B61, dX, dA2, dA3, dA6);
    __syncthreads();
    product(i0, i2, Tm);
    product(i0, i3, dT);
    product(i1, i2, dT);
    for (int it = 0; it < cm.ps; ++it) {
      __syncthreads();
      store_tiles(i0, Tm);
      store_tiles(i1, dT);
      __syncthreads();
      Regs T2, dT2;
      zero(T2), zero(dT2);
      product(i0, i0, T2);
      product(i0, i1, dT2);
      product(i1, i0, dT2);
      Tm = T2;
      dT = dT2;
    }
    // ---- A <- T A (issued first: its operands go to the images while the inner products run on registers) ----
    __syncthreads();
    store_tiles(i0, Tm);
    store_tiles(i1, Aa);
    __syncthreads();
    // ---- grad[k] = Re( conj(e^{ams + mu}) <dT, G_k> ), <Z, G> = sum conj(Z) G ----
    {
      double sn, cs;
      sincos(ams_i + mu_i, &sn, &cs);
      const double er = exp(ams_r + mu_r);
      const double pr = er * cs, pi = er * sn;
      if (A.zout != nullptr) {  // Z_n = e^{shift} dT, the cotangent of the generator X_n (Re / Im rows of an element: lanes l, l ^ 16)
        double* dst = reinterpret_cast<double*>(A.zout) + ((long)cm.sample * A.N + cm.n0 + t) * D * D * 2;
#pragma unroll
        for (int e = 0; e < NE; ++e) {
          const int row = erow(e), col = ecol(e);
          const int ci = row >> 1;
          const double mine = dT.get(e), other = __shfl_xor(mine, 16);
          const double outv = (row & 1) ? fma(pr, mine, pi * other) : fma(pr, mine, -pi * other);
          if (ci < D && col < D) dst[(ci * D + col) * 2 + (row & 1)] = outv;
        }
      }
      double trr = 0.0, tri = 0.0;
#pragma unroll
      for (int e = 0; e < NE; ++e) {
        const int row = erow(e), col = ecol(e);
        const bool dg = ((row >> 1) == col) && (col < D);
        const double v = dg ? dT.get(e) : 0.0;
        if (row & 1)
          tri += v;
        else
          trr += v;
      }
      for (int o = 32; o >= 1; o >>= 1) {
        trr += __shfl_xor(trr, o);
        tri += __shfl_xor(tri, o);
      }
      for (int k = 0; k < K; ++k) {
        const double* tk = tabg + (long)(k + 1) * (IMG + 4);
        double re = 0.0, im = 0.0;
#pragma unroll
        for (int e = 0; e < NE; ++e) {
          const double z = dT.get(e);
          const int o = eoff(e);
          const bool odd = erow(e) & 1;
          re = fma(z, tk[o], re);
          const double go = tk[odd ? o - W : o + W];  // the other half of the same complex element
          im = odd ? fma(-z, go, im) : fma(z, go, im);
        }
        for (int o = 32; o >= 1; o >>= 1) {
          re += __shfl_xor(re, o);
          im += __shfl_xor(im, o);
        }
        // trace part conj(tr Z) mu_k: every wave adds the share of its own diagonal elements
        re += fma(tk[IMG + 0], trr, tk[IMG + 1] * tri);
        im += fma(tk[IMG + 1], trr, -tk[IMG + 0] * tri);
        if (cm.lane == 0) red[WV * 16 + k] = fma(pr, re, pi * im);
      }
    }
    Regs V;
    zero(V);
    product(i0, i1, V);
    Aa = V;
    ams_r += mu_r;
    ams_i = c3p_phase_add(ams_i, mu_i);
    __syncthreads();  // partial sums visible
    if (WV == 0 && cm.lane < K)
      A.grad[((long)cm.sample * K + cm.lane) * A.N + cm.n0 + t] =
          red[cm.lane] + red[16 + cm.lane] + red[32 + cm.lane] + red[48 + cm.lane];
  }
}

template <int NIG, int NJ, int W, bool XG>
__global__ void __launch_bounds__(256, 1) midd_grad_general_kernel(MidGradArgs A) {
  using C = MD<NIG, NJ>;
  constexpr int IMG = C::ROWS * W;
  const int tid = threadIdx.x;
  MidCommon cm;
  cm.lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  cm.r = cm.lane >> 4;
  cm.b = (cm.lane >> 2) & 3;
  cm.c = cm.lane & 3;
  cm.D = A.Dm;
  cm.nbk = (2 * cm.D + 3) / 4;
  cm.K = XG ? 0 : A.K;
  const int K = cm.K;
  double* i0 = c3p_md_lds;
  double* i1 = i0 + IMG;
  double* i2 = i1 + IMG;
  double* i3 = i2 + IMG;
  cm.sg = i3 + IMG;
  __shared__ double red[64];
  __shared__ double redn[NW];
  const long chain = blockIdx.x;
  cm.sample = (int)(chain / A.S);
  const int seg = (int)(chain - (long)cm.sample * A.S);
  cm.n0 = (int)(((long)seg * A.N) / A.S);
  const int n1 = (int)(((long)(seg + 1) * A.N) / A.S);
  cm.len = n1 - cm.n0;
  cm.aoff = (4 * cm.b + (cm.c & ~1) + ((cm.c ^ cm.r) & 1)) * W + (cm.r >> 1);
  cm.boff = cm.r * W + cm.c;
  cm.lsmall = (4 * cm.b + cm.r) * W + cm.c;
  cm.lbig = cm.r * W + 4 * cm.b + cm.c;
  cm.negmask = (((cm.c & 1) == 0) && ((cm.r & 1) == 1)) ? 0x80000000u : 0u;
  for (int e = tid; e < 4 * IMG; e += 256) c3p_md_lds[e] = 0.0;
  __syncthreads();
  double nrm = 0.0;
  if constexpr (XG) {
    // supplied generators: the 1-norm of X_n^H is the row-sum norm of X_n (slot 3 of the hmeta pass)
    cm.tabs = nullptr;
    const double* mt = A.meta + ((long)cm.sample * A.N + cm.n0) * 4;
    for (int t = tid; t < cm.len; t += 256) nrm = fmax(nrm, mt[(long)t * 4 + 3]);
    for (int o = 32; o >= 1; o >>= 1) nrm = fmax(nrm, __shfl_xor(nrm, o));
    if (cm.lane == 0) redn[wave] = nrm;
    __syncthreads();
    nrm = fmax(fmax(redn[0], redn[1]), fmax(redn[2], redn[3]));
    __syncthreads();
  } else {
    cm.tabs = A.tables_h + (long)(A.tab_per_sample ? cm.sample : 0) * (1 + K) * (IMG + 4);  // the matrix exponentiated is X^H
    nrm = cm.tabs[IMG + 2];
  }
  for (int k = 0; k < K; ++k) {
    const double* s = A.signals + ((long)cm.sample * K + k) * A.N + cm.n0;
    double cmax = 0.0;
    for (int t = tid; t < cm.len; t += 256) {
      const double v = s[t];
      cm.sg[k * A.Lmax + t] = v;
      cmax = fmax(cmax, fabs(v));
    }
    for (int o = 32; o >= 1; o >>= 1) cmax = fmax(cmax, __shfl_xor(cmax, o));
    if (cm.lane == 0) redn[wave] = cmax;
    __syncthreads();
    cmax = fmax(fmax(redn[0], redn[1]), fmax(redn[2], redn[3]));
    __syncthreads();
    nrm = fma(cmax, cm.tabs[(long)(k + 1) * (IMG + 4) + IMG + 2], nrm);
  }
  nrm = md_rfl(nrm);
  int ps = 0;
  {
    double p = C3P_T18_THETA;
    while (p < nrm && ps < 40) {
      p *= 2.0;
      ++ps;
    }
  }
  cm.ps = __builtin_amdgcn_readfirstlane(ps);
  cm.pr = 0;
  cm.t18 = 1;
  cm.scale = ldexp(1.0, -cm.ps);
  cm.buf0 = i0;
  cm.buf1 = i1;
  cm.buf2 = i2;
  __syncthreads();
  switch (wave) {
    case 0: midd_grad_general_body<NIG, NJ, W, 0, XG>(A, cm, chain, i0, i1, i2, i3, red); break;
    case 1: midd_grad_general_body<NIG, NJ, W, 1, XG>(A, cm, chain, i0, i1, i2, i3, red); break;
    case 2: midd_grad_general_body<NIG, NJ, W, 2, XG>(A, cm, chain, i0, i1, i2, i3, red); break;
    default: midd_grad_general_body<NIG, NJ, W, 3, XG>(A, cm, chain, i0, i1, i2, i3, red); break;
  }
}

// ---------------------------------------------------------------------------------------------
// Backward sweep for REAL Hamiltonians in the mid-D layout: reverse mode through the real cos / sin evaluation of
// midd_real_body (method and derivation: smalld_grad_real_kernel in c3p_smalld.hip).  The adjoint state is carried
// transposed, N = M^T; every cotangent of the symmetric stage is  A_bar B + B A_bar = P + P^T  with ONE real product
// P = A_bar B: the product goes through an LDS image and each lane adds the mirror element (linear combinations are
// formed BEFORE the transposition where two products feed the same cotangent).  Per slice 7 + 4 + 11 + 4 real products
// (+ 2 forward and 3 backward per squaring) against 18 + 3 s COMPLEX ones of the pair evaluation above.
// Eight real images: slots 0..3 hold the operands of the running product, 4 / 5 the products to be transposed,
// 6 / 7 the product R = dU N for the update N <- R conj(dU) (stored transposed in the swizzled layout, whose left
// operands are read in the B pattern).  Always the degree-16 / 17 polynomials, up to MGR<NIG>::MAXS squarings; other chains
// are left to midd_grad_kernel (same tables, same norm bound: same decision).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int mgr_tswz(int x) { return x & 15; }

template <int NIG, int NJ, int W, int WV, bool DEG20>
__device__ __forceinline__ void midd_grad_real_body(const MidGradArgs& A, const MidCommon& cm, long chain, double* red) {
  constexpr int NIGR = MDR<NIG, W>::NIGR;
  constexpr int WI = MDR<NIG, W>::WI;
  constexpr bool SWZ = MDR<NIG, W>::SWZ;
  using T = WaveTiles<NIGR, NJ, WI, WV>;
  constexpr int IMG = MD<NIG, NJ>::ROWS * W;  // complex table image
  constexpr int IMGR = 16 * NIGR * WI, NE = T::NE;
  constexpr int KP = MDR<NIG, W>::KP;
  typedef TileRegs<T::NBW, T::NSW> Regs;
  const int D = cm.D, K = cm.K;
  const int rbig = cm.r, rsmall = 4 * cm.b + cm.r;
  const int cbig = 4 * cm.b + cm.c, csmall = cm.c;
  const double* tabs = cm.tabs;
  double* const qbig0 = c3p_md_lds + cm.rbig[0];
  double* const qbig1 = c3p_md_lds + cm.rbig[1];
  double* const qsm0 = c3p_md_lds + cm.rsm[0];
  double* const qsm1 = c3p_md_lds + cm.rsm[1];
  auto erow = [&](int e) -> int { return T::lrow(e, cm.r, cm.b, cm.c); };
  auto ecol = [&](int e) -> int { return T::lcol(e, cm.r, cm.b, cm.c); };
  constexpr bool PW = Sched<NIGR, NJ>::PW;  // pinwheel deal: one lane offset per element
  int pwoff[NE > 0 ? NE : 1];
#pragma unroll
  for (int e = 0; e < NE; ++e) pwoff[e] = PW ? md_pw_off(erow(e), ecol(e)) : 0;
  auto st = [&](auto img, const Regs& v) {
    constexpr int I = decltype(img)::value;
#pragma unroll
    for (int e = 0; e < NE; ++e) {
      if constexpr (PW)
        c3p_md_lds[I * IMGR + pwoff[e]] = v.get(e);
      else
        (T::is_big(e) ? (T::col0(e) < 16 ? qbig0 : qbig1) : (T::col0(e) < 16 ? qsm0 : qsm1))[I * IMGR + T::off0(e)] = v.get(e);
    }
  };
  // the mirror position of every element of the lane: offset of (col, row) in an image, 0 / 1 mask (inside the matrix)
  int toff[NE > 0 ? NE : 1], soff[NE > 0 ? NE : 1];
  unsigned inbits = 0, dbits = 0;
#pragma unroll
  for (int e = 0; e < NE; ++e) {
    const int row = erow(e), col = ecol(e);
    const bool in = row < D && col < D;
    const int rr = in ? row : 0, cc = in ? col : 0;
    // images 4 / 5 only pass products to their mirror reads: in the 32-wide classes they have their own swizzle,
    // element (row, col) at row * 32 + (col ^ (row & 15)).  A 64-bit LDS access is served 16 lanes per cycle (32 banks x 4 B);
    // the 16 lanes of a group hold ONE row and 16 columns in a store (any XOR keeps them distinct mod 16) and ONE column and
    // 16 rows in a mirror read: (row' ^ (col' & 15)) mod 16 is distinct over the 16 col' -- conflict-free both ways (the
    // operand swizzle put the 16 lanes of a mirror read on one bank pair)
    toff[e] = SWZ ? cc * 32 + (rr ^ mgr_tswz(cc)) : cc * WI + rr;
    soff[e] = SWZ ? row * 32 + (col ^ mgr_tswz(row)) : 0;
    inbits |= in ? (1u << e) : 0u;
    dbits |= (row == col && col < D) ? (1u << e) : 0u;
  }
  auto dmask = [&](int e) -> double { return (dbits >> e) & 1u ? 1.0 : 0.0; };
  // out = f (P + P^T) for the product P the workgroup has just stored in image I
  auto mirror = [&](auto img, const Regs& P, double f, Regs& out) {
    constexpr int I = decltype(img)::value;
#pragma unroll
    for (int e = 0; e < NE; ++e) {
      const double m = c3p_md_lds[I * IMGR + toff[e]];
      out.set(e, f * (P.get(e) + ((inbits >> e) & 1u ? m : 0.0)));
    }
  };
  // store into a mirror image (4 / 5)
  auto st_T = [&](auto img, const Regs& v) {
    constexpr int I = decltype(img)::value;
    if constexpr (SWZ) {
#pragma unroll
      for (int e = 0; e < NE; ++e) c3p_md_lds[I * IMGR + soff[e]] = v.get(e);
    } else {
      st(img, v);
    }
  };
  auto zero = [&](Regs& v) {
#pragma unroll
    for (int e = 0; e < NE; ++e) v.set(e, 0.0);
  };
  Regs dummy;
  // dY/dc_k at the lane's positions (-scale Im(table k); k = 0: the drift): in registers for the whole segment
  // (TABREG = false reads them from the tables (L2) at every use: measured 18% slower at D = 36, although it spills less)
  constexpr bool TABREG = true;
  Regs Tab[TABREG ? KP + 1 : 1];
  int tix[NE > 0 ? NE : 1];
  double tmu_r[KP + 1], tmu_i[KP + 1];
#pragma unroll
  for (int e = 0; e < NE; ++e) {
    const int row = erow(e), col = ecol(e);
    tix[e] = (row < D && col < D) ? (2 * row + 1) * W + col : W;
  }
#pragma unroll
  for (int k = 0; k <= KP; ++k) {
    const double* tk = tabs + (long)(k <= K ? k : 0) * (IMG + 4);
    const double on = k <= K ? -cm.scale : 0.0;
    if constexpr (TABREG) {
#pragma unroll
      for (int e = 0; e < NE; ++e) Tab[k].set(e, (inbits >> e) & 1u ? on * tk[tix[e]] : 0.0);
    }
    tmu_r[k] = md_rfl(k <= K ? tk[IMG + 0] : 0.0);
    tmu_i[k] = md_rfl(k <= K ? tk[IMG + 1] : 0.0);
  }
  auto tab_get = [&](int k, Regs& out) {  // k <= KP, compile-time after unrolling
    if constexpr (TABREG) {
      out = Tab[k];
    } else {
      const double* tk = tabs + (long)(k <= K ? k : 0) * (IMG + 4);
      const double on = k <= K ? -cm.scale : 0.0;
#pragma unroll
      for (int e = 0; e < NE; ++e) out.set(e, (inbits >> e) & 1u ? on * tk[tix[e]] : 0.0);
    }
  };
  // N = M^T at the end of the segment
  Regs Nr, Ni;
  {
    const double* src = reinterpret_cast<const double*>(A.Mb) + chain * D * D * 2;
#pragma unroll
    for (int e = 0; e < NE; ++e) {
      const int row = erow(e), col = ecol(e);
      const bool in = row < D && col < D;
      Nr.set(e, in ? src[(col * D + row) * 2 + 0] : 0.0);
      Ni.set(e, in ? src[(col * D + row) * 2 + 1] : 0.0);
    }
  }
  const int ps = cm.ps;
  for (int t = cm.len - 1; t >= 0; --t) {
    // ---- forward: Y, W = Y^2, W^2, {W^3, W^4}, cos Y, sin(Y)/Y, sin Y (midd_real_body, degree 16 / 17) ----
    Regs Y;
    tab_get(0, Y);
#pragma unroll
    for (int k = 0; k < KP; ++k) {
      const double c0 = k < K ? cm.sg[k * A.Lmax + t] : 0.0;
      Regs Tk;
      tab_get(k + 1, Tk);
#pragma unroll
      for (int e = 0; e < NE; ++e) Y.set(e, fma(c0, Tk.get(e), Y.get(e)));
    }
    md_bar();  // the previous slice's last products have left the images
    st(IC<0>{}, Y);
    md_bar();
    Regs W1, W2, W3, W4, Cm, Sp, acc, acs, Sn;
    zero(W1);
    mm_real<NIGR, NJ, W, WV, 0, 0, 0, 0, 0>(cm, W1, dummy);
    st(IC<1>{}, W1);
    md_bar();
    zero(W2);
    mm_real<NIGR, NJ, W, WV, 0, 1, 1, 1, 1>(cm, W2, dummy);
    st(IC<2>{}, W2);
    md_bar();
    zero(W3);
    zero(W4);
    mm_real<NIGR, NJ, W, WV, 2, 1, 2, 2, 2>(cm, W3, W4);
    auto rc = [&](Regs& out, double c0, double c1, double c2, double c3) {
#pragma unroll
      for (int e = 0; e < NE; ++e) {
        double v = c1 * W1.get(e);
        v = fma(c2, W2.get(e), v);
        v = fma(c3, W3.get(e), v);
        out.set(e, fma(c0, dmask(e), v));
      }
    };
    // coefficients of W^j in cos Y / (sin Y / Y): Taylor for the degree-20 variant, the economised degree-8 pair (theta = 1.85,
    // c3p_common.h) otherwise -- forward evaluation and its adjoint below use the same table
    auto ca = [&](int j) { return DEG20 ? ((j & 1) ? -c3p_inv_fact[2 * j] : c3p_inv_fact[2 * j]) : c3p_mm8_cos[j]; };
    auto sa = [&](int j) { return DEG20 ? ((j & 1) ? -c3p_inv_fact[2 * j + 1] : c3p_inv_fact[2 * j + 1]) : c3p_mm8_sinc[j]; };
    // the power the Horner step runs in: W^4 (degree 16), W^5 = W^2 W^3 (degree 20, theta_20 = 1.49: one squaring less)
    Regs H;
    if constexpr (DEG20) {
      st(IC<3>{}, W3);
      md_bar();
      zero(H);
      mm_real<NIGR, NJ, W, WV, 0, 2, 2, 3, 3>(cm, H, dummy);
#pragma unroll
      for (int e = 0; e < NE; ++e) {
        double a = -c3p_inv_fact[10] * dmask(e), s = -c3p_inv_fact[11] * dmask(e);
        a = fma(c3p_inv_fact[12], W1.get(e), a), s = fma(c3p_inv_fact[13], W1.get(e), s);
        a = fma(-c3p_inv_fact[14], W2.get(e), a), s = fma(-c3p_inv_fact[15], W2.get(e), s);
        a = fma(c3p_inv_fact[16], W3.get(e), a), s = fma(c3p_inv_fact[17], W3.get(e), s);
        a = fma(-c3p_inv_fact[18], W4.get(e), a), s = fma(-c3p_inv_fact[19], W4.get(e), s);
        acc.set(e, fma(c3p_inv_fact[20], H.get(e), a));
        acs.set(e, fma(c3p_inv_fact[21], H.get(e), s));
      }
      st(IC<7>{}, H);
    } else {
      H = W4;
      rc(acc, ca(4), ca(5), ca(6), ca(7));
      rc(acs, sa(4), sa(5), sa(6), sa(7));
#pragma unroll
      for (int e = 0; e < NE; ++e) {
        acc.set(e, fma(ca(8), W4.get(e), acc.get(e)));
        acs.set(e, fma(sa(8), W4.get(e), acs.get(e)));
      }
      st(IC<7>{}, H);
    }
    st(IC<4>{}, acc);
    st(IC<5>{}, acs);
    md_bar();
    rc(Cm, ca(0), ca(1), ca(2), ca(3));
    rc(Sp, sa(0), sa(1), sa(2), sa(3));
    if constexpr (DEG20) {
#pragma unroll
      for (int e = 0; e < NE; ++e) {
        Cm.set(e, fma(c3p_inv_fact[8], W4.get(e), Cm.get(e)));
        Sp.set(e, fma(c3p_inv_fact[9], W4.get(e), Sp.get(e)));
      }
    }
    mm_real<NIGR, NJ, W, WV, 1, 7, 7, 4, 5>(cm, Cm, Sp);
    st(IC<6>{}, Sp);
    md_bar();
    zero(Sn);
    mm_real<NIGR, NJ, W, WV, 0, 0, 0, 6, 6>(cm, Sn, dummy);
    // squarings, every level kept: C' = 2 C^2 - I, S' = 2 S C
    constexpr int MAXS = MGR<NIG>::MAXS;
    Regs Cl[MAXS], Sl[MAXS];
    auto square = [&](auto ia, auto ib, int lvl) {
      constexpr int IA = decltype(ia)::value, IB = decltype(ib)::value;
      Cl[lvl] = Cm;
      Sl[lvl] = Sn;
      st(ia, Cm);
      st(ib, Sn);
      md_bar();
      Regs C2, SC;
      zero(C2);
      zero(SC);
      mm_real<NIGR, NJ, W, WV, 2, IA, IB, IA, IA>(cm, C2, SC);
#pragma unroll
      for (int e = 0; e < NE; ++e) {
        Cm.set(e, fma(2.0, C2.get(e), -dmask(e)));
        Sn.set(e, 2.0 * SC.get(e));
      }
    };
    if (ps > 0) square(IC<1>{}, IC<2>{}, 0);
    if constexpr (MAXS > 1)
      if (ps > 1) square(IC<3>{}, IC<4>{}, 1);
    // ---- R = dU N = (C - iS)(Nr + i Ni):  Rr = C Nr + S Ni,  Ri = C Ni - S Nr ----
    md_bar();
    st(IC<0>{}, Cm);
    st(IC<1>{}, Sn);
    st(IC<2>{}, Nr);
    st(IC<3>{}, Ni);
    md_bar();
    Regs Rr, Ri;
    zero(Rr);
    zero(Ri);
    if constexpr (PW) {
      mm_real<NIGR, NJ, W, WV, 3, 0, 1, 2, 3>(cm, Rr, Ri);  // all four products in one pass over the operands
    } else {
      mm_real<NIGR, NJ, W, WV, 1, 1, 1, 3, 2>(cm, Rr, Ri);  // S Ni, S Nr
#pragma unroll
      for (int e = 0; e < NE; ++e) Ri.set(e, -Ri.get(e));
      mm_real<NIGR, NJ, W, WV, 1, 0, 0, 2, 3>(cm, Rr, Ri);  // + C Nr, + C Ni
    }
    // tr N (= tr M), this wave's share
    double trr = 0.0, tri = 0.0;
#pragma unroll
    for (int e = 0; e < NE; ++e) {
      trr = fma(dmask(e), Nr.get(e), trr);
      tri = fma(dmask(e), Ni.get(e), tri);
    }
    st_T(IC<4>{}, Rr);
    st_T(IC<5>{}, Ri);
    if constexpr (!SWZ) {
      st(IC<6>{}, Rr);
      st(IC<7>{}, Ri);
    }
    md_bar();
    // ---- cotangents of cos / sin: C_bar = sym Re R, S_bar = -sym Im R ----
    // (swizzled classes: the mirror elements are also R^T at this lane's positions -- the image of the LEFT operand of
    // the update of N, which those classes read in the B pattern -- stored with the ordinary conflict-free tile store;
    // a transposed store into the operand layout puts the 16 lanes of a group on one bank pair)
    Regs Cb, Sb;
    {
      Regs tRr, tRi;
#pragma unroll
      for (int e = 0; e < NE; ++e) {
        const double mr = c3p_md_lds[4 * IMGR + toff[e]], mi = c3p_md_lds[5 * IMGR + toff[e]];
        tRr.set(e, (inbits >> e) & 1u ? mr : 0.0);
        tRi.set(e, (inbits >> e) & 1u ? mi : 0.0);
        Cb.set(e, 0.5 * (Rr.get(e) + tRr.get(e)));
        Sb.set(e, -0.5 * (Ri.get(e) + tRi.get(e)));
      }
      if constexpr (SWZ) {
        st(IC<6>{}, tRr);
        st(IC<7>{}, tRi);
      }
    }
    // ---- back through the squarings: C_bar = 2 {C_bar', C} + {S_bar', S},  S_bar = {S_bar', C}  ({A, B} = AB + BA) ----
    auto unsquare = [&](int lvl) {
      st(IC<0>{}, Cb);
      st(IC<1>{}, Sb);
      st(IC<2>{}, Cl[lvl]);
      st(IC<3>{}, Sl[lvl]);
      md_bar();
      Regs P1, P2;
      zero(P1);
      zero(P2);
      mm_real<NIGR, NJ, W, WV, 2, 0, 1, 2, 2>(cm, P1, P2);  // C_bar' C, S_bar' C
#pragma unroll
      for (int e = 0; e < NE; ++e) P1.set(e, 2.0 * P1.get(e));
      mm_real<NIGR, NJ, W, WV, 0, 1, 1, 3, 3>(cm, P1, dummy);  // + S_bar' S
      st_T(IC<4>{}, P1);
      st_T(IC<5>{}, P2);
      md_bar();
      mirror(IC<4>{}, P1, 1.0, Cb);
      mirror(IC<5>{}, P2, 1.0, Sb);
    };
    if constexpr (MAXS > 1)
      if (ps > 1) unsquare(1);
    if (ps > 0) unsquare(0);
    // ---- S = Y Sp:  Y_bar = sym(S_bar Sp) (kept doubled),  Sp_bar = sym(S_bar Y) ----
    Regs Yb2, Spb;
    {
      st(IC<0>{}, Sb);
      st(IC<1>{}, Sp);
      st(IC<2>{}, Y);
      md_bar();
      Regs Pa, Pb;
      zero(Pa);
      zero(Pb);
      mm_real<NIGR, NJ, W, WV, 1, 0, 0, 1, 2>(cm, Pa, Pb);
      st_T(IC<4>{}, Pa);
      st_T(IC<5>{}, Pb);
      md_bar();
      mirror(IC<4>{}, Pa, 1.0, Yb2);
      mirror(IC<5>{}, Pb, 0.5, Spb);
    }
    // ---- Cm = Cm0 + H acc, Sp = Sp0 + H acs (W4b2: twice the cotangent of H from these two) ----
    Regs W4b2, accb, acsb;
    {
      st(IC<0>{}, Cb);
      st(IC<1>{}, acc);
      st(IC<2>{}, H);
      md_bar();
      Regs Pc, Pd;
      zero(Pc);
      zero(Pd);
      mm_real<NIGR, NJ, W, WV, 1, 0, 0, 1, 2>(cm, Pc, Pd);  // C_bar acc, C_bar H
      st_T(IC<4>{}, Pc);
      st_T(IC<5>{}, Pd);
      md_bar();
      mirror(IC<4>{}, Pc, 1.0, W4b2);
      mirror(IC<5>{}, Pd, 0.5, accb);
      st(IC<0>{}, Spb);
      st(IC<1>{}, acs);  // (image 2 still holds H)
      md_bar();
      Regs Pe, Pf;
      zero(Pe);
      zero(Pf);
      mm_real<NIGR, NJ, W, WV, 1, 0, 0, 1, 2>(cm, Pe, Pf);  // Sp_bar acs, Sp_bar H
      st_T(IC<4>{}, Pe);
      st_T(IC<5>{}, Pf);
      md_bar();
      Regs tmp;
      mirror(IC<4>{}, Pe, 1.0, tmp);
      mirror(IC<5>{}, Pf, 0.5, acsb);
#pragma unroll
      for (int e = 0; e < NE; ++e) W4b2.set(e, W4b2.get(e) + tmp.get(e));
    }
    Regs W1b, W2b, W3b, W4b;
    if constexpr (DEG20) {
      Regs W5b;
#pragma unroll
      for (int e = 0; e < NE; ++e) {
        const double ab = accb.get(e), sb = acsb.get(e), cb = Cb.get(e), pb = Spb.get(e);
        W1b.set(e, -(c3p_inv_fact[2] * cb + c3p_inv_fact[3] * pb) + (c3p_inv_fact[12] * ab + c3p_inv_fact[13] * sb));
        W2b.set(e, (c3p_inv_fact[4] * cb + c3p_inv_fact[5] * pb) - (c3p_inv_fact[14] * ab + c3p_inv_fact[15] * sb));
        W3b.set(e, -(c3p_inv_fact[6] * cb + c3p_inv_fact[7] * pb) + (c3p_inv_fact[16] * ab + c3p_inv_fact[17] * sb));
        W4b.set(e, (c3p_inv_fact[8] * cb + c3p_inv_fact[9] * pb) - (c3p_inv_fact[18] * ab + c3p_inv_fact[19] * sb));
        W5b.set(e, 0.5 * W4b2.get(e) + c3p_inv_fact[20] * ab + c3p_inv_fact[21] * sb);
      }
      // ---- W5 = W2 W3:  W2_bar += sym(W5_bar W3),  W3_bar += sym(W5_bar W2) ----
      st(IC<0>{}, W5b);
      st(IC<1>{}, W2);
      st(IC<2>{}, W3);
      md_bar();
      Regs Pm, Pn;
      zero(Pm);
      zero(Pn);
      mm_real<NIGR, NJ, W, WV, 1, 0, 0, 1, 2>(cm, Pm, Pn);  // W5_bar W2, W5_bar W3
      st_T(IC<4>{}, Pm);
      st_T(IC<5>{}, Pn);
      md_bar();
      Regs q, h;
      mirror(IC<4>{}, Pm, 0.5, q);
      mirror(IC<5>{}, Pn, 0.5, h);
#pragma unroll
      for (int e = 0; e < NE; ++e) {
        W3b.set(e, W3b.get(e) + q.get(e));
        W2b.set(e, W2b.get(e) + h.get(e));
      }
    } else {
#pragma unroll
      for (int e = 0; e < NE; ++e) {
        const double ab = accb.get(e), sb = acsb.get(e), cb = Cb.get(e), pb = Spb.get(e);
        W1b.set(e, ca(1) * cb + sa(1) * pb + ca(5) * ab + sa(5) * sb);
        W2b.set(e, ca(2) * cb + sa(2) * pb + ca(6) * ab + sa(6) * sb);
        W3b.set(e, ca(3) * cb + sa(3) * pb + ca(7) * ab + sa(7) * sb);
        W4b.set(e, 0.5 * W4b2.get(e) + ca(8) * ab + sa(8) * sb);
      }
    }
    // ---- W4 = W2^2, W3 = W W2:  W2_bar += {W4_bar, W2} + sym(W3_bar W),  W_bar += sym(W3_bar W2) ----
    {
      st(IC<0>{}, W3b);
      st(IC<1>{}, W1);
      st(IC<2>{}, W2);
      st(IC<3>{}, W4b);
      md_bar();
      Regs Pg, Ph, Pi;
      zero(Pg);
      zero(Ph);
      zero(Pi);
      mm_real<NIGR, NJ, W, WV, 1, 0, 0, 1, 2>(cm, Pg, Ph);   // W3_bar W, W3_bar W2
      mm_real<NIGR, NJ, W, WV, 0, 3, 3, 2, 2>(cm, Pi, dummy);  // W4_bar W2
#pragma unroll
      for (int e = 0; e < NE; ++e) Pi.set(e, fma(0.5, Pg.get(e), Pi.get(e)));
      st_T(IC<4>{}, Pi);
      st_T(IC<5>{}, Ph);
      md_bar();
      Regs q, h;
      mirror(IC<4>{}, Pi, 1.0, q);
      mirror(IC<5>{}, Ph, 0.5, h);
#pragma unroll
      for (int e = 0; e < NE; ++e) {
        W2b.set(e, W2b.get(e) + q.get(e));
        W1b.set(e, W1b.get(e) + h.get(e));
      }
    }
    // ---- W2 = W^2:  W_bar += {W2_bar, W} ----
    {
      st(IC<0>{}, W2b);  // (image 1 still holds W)
      md_bar();
      Regs Pj;
      zero(Pj);
      mm_real<NIGR, NJ, W, WV, 0, 0, 0, 1, 1>(cm, Pj, dummy);
      st_T(IC<4>{}, Pj);
      md_bar();
      Regs q;
      mirror(IC<4>{}, Pj, 1.0, q);
#pragma unroll
      for (int e = 0; e < NE; ++e) W1b.set(e, W1b.get(e) + q.get(e));
    }
    // ---- W = Y^2:  Y_bar = sym(S_bar Sp) + {W_bar, Y} ----
    Regs Yb;
    {
      st(IC<0>{}, W1b);
      st(IC<2>{}, Y);
      md_bar();
      Regs Pk;
      zero(Pk);
      mm_real<NIGR, NJ, W, WV, 0, 0, 0, 2, 2>(cm, Pk, dummy);
      st_T(IC<4>{}, Pk);
      md_bar();
      mirror(IC<4>{}, Pk, 1.0, Yb);
#pragma unroll
      for (int e = 0; e < NE; ++e) Yb.set(e, fma(0.5, Yb2.get(e), Yb.get(e)));
    }
    // ---- grad[k] = <Y_bar, dY/dc_k> + Re(mu_k conj(tr N)) ----
#pragma unroll
    for (int k = 0; k < KP; ++k) {
      double part = fma(tmu_r[k + 1], trr, tmu_i[k + 1] * tri);
      Regs Tk;
      tab_get(k + 1, Tk);
#pragma unroll
      for (int e = 0; e < NE; ++e) part = fma(Yb.get(e), Tk.get(e), part);
      for (int o = 32; o >= 1; o >>= 1) part += __shfl_xor(part, o);
      if (cm.lane == 0) red[WV * 16 + k] = part;
    }
    // ---- N <- R conj(dU):  Nr = Rr C - Ri S,  Ni = Rr S + Ri C  (R: images 6 / 7) ----
    st(IC<0>{}, Cm);
    st(IC<1>{}, Sn);
    md_bar();  // (also: partial sums visible)
    if (WV == 0 && cm.lane < K)
      A.grad[((long)cm.sample * K + cm.lane) * A.N + cm.n0 + t] =
          red[cm.lane] + red[16 + cm.lane] + red[32 + cm.lane] + red[48 + cm.lane];
    if (t > 0) {
      zero(Nr);
      zero(Ni);
      if constexpr (PW) {
        mm_real<NIGR, NJ, W, WV, 4, 6, 7, 0, 1>(cm, Nr, Ni);  // Rr C - Ri S, Rr S + Ri C in one pass
      } else {
        mm_real<NIGR, NJ, W, WV, 2, 7, 6, 1, 1>(cm, Nr, Ni);  // Ri S, Rr S
#pragma unroll
        for (int e = 0; e < NE; ++e) Nr.set(e, -Nr.get(e));
        mm_real<NIGR, NJ, W, WV, 2, 6, 7, 0, 0>(cm, Nr, Ni);  // + Rr C, + Ri C
      }
    }
  }
}

template <int NIG, int NJ, int W>
__global__ void __launch_bounds__(256, (MDR<NIG, W>::SWZ ? 2 : 1)) midd_grad_real_kernel(MidGradArgs A) {
  using C = MD<NIG, NJ>;
  constexpr int IMG = C::ROWS * W;
  constexpr int WI = MDR<NIG, W>::WI, IMGR = 16 * MDR<NIG, W>::NIGR * WI;
  const int tid = threadIdx.x;
  MidCommon cm;
  cm.lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  cm.r = cm.lane >> 4;
  cm.b = (cm.lane >> 2) & 3;
  cm.c = cm.lane & 3;
  cm.D = A.Dm;
  cm.nbk = (2 * cm.D + 3) / 4;
  cm.nbkR = (cm.D + 3) / 4;
  cm.K = A.K;
  const int K = A.K;
  cm.sg = c3p_md_lds + MGR_SLOTS * IMGR;
  __shared__ double red[64];
  __shared__ double redn[NW];
  const long chain = blockIdx.x;
  cm.sample = (int)(chain / A.S);
  cm.tabs = A.tables + (long)(A.tab_per_sample ? cm.sample : 0) * (1 + K) * (IMG + 4);
  bool realH = K <= MDR<NIG, W>::KP;
  for (int k = 0; k <= K; ++k) realH = realH && (cm.tabs[(long)k * (IMG + 4) + IMG + 3] == 0.0);
  if (__builtin_amdgcn_readfirstlane((int)realH) == 0) return;
  const int seg = (int)(chain - (long)cm.sample * A.S);
  cm.n0 = (int)(((long)seg * A.N) / A.S);
  const int n1 = (int)(((long)(seg + 1) * A.N) / A.S);
  cm.len = n1 - cm.n0;
  {
    const int sw = MDR<NIG, W>::SWZ ? 16 * (cm.r & 1) : 0;
    cm.aoffR = (4 * cm.b + cm.c) * WI + cm.r;
    cm.rbig[0] = cm.r * WI + 4 * cm.b + cm.c + sw;
    cm.rbig[1] = cm.r * WI + 4 * cm.b + cm.c - sw;
    cm.rsm[0] = (4 * cm.b + cm.r) * WI + cm.c + sw;
    cm.rsm[1] = (4 * cm.b + cm.r) * WI + cm.c - sw;
    cm.rblk[0] = cm.r * WI + cm.c + sw;
    cm.rblk[1] = cm.r * WI + cm.c - sw;
    cm.rb12 = cm.r * WI + ((12 + 4 * cm.b + cm.c) ^ sw);
    cm.rctr = (4 * cm.b + cm.r) * WI + ((12 + cm.c) ^ sw ^ (4 * cm.b));
  }
  double nrm = cm.tabs[IMG + 2];
  for (int k = 0; k < K; ++k) {
    const double* s = A.signals + ((long)cm.sample * K + k) * A.N + cm.n0;
    double cmax = 0.0;
    for (int t = tid; t < cm.len; t += 256) cmax = fmax(cmax, fabs(s[t]));
    for (int o = 32; o >= 1; o >>= 1) cmax = fmax(cmax, __shfl_xor(cmax, o));
    if (cm.lane == 0) redn[wave] = cmax;
    __syncthreads();
    cmax = fmax(fmax(redn[0], redn[1]), fmax(redn[2], redn[3]));
    __syncthreads();
    nrm = fma(cmax, cm.tabs[(long)(k + 1) * (IMG + 4) + IMG + 2], nrm);
  }
  nrm = md_rfl(nrm);
  cm.ps = __builtin_amdgcn_readfirstlane(mgr_squarings(nrm));
  if (cm.ps > MGR<NIG>::MAXS) return;
  // degree 20 (theta_20 = 1.49) where it saves a squaring: 8 + 13 products against 7 + 11 + 5 per squaring
  const int deg20 = __builtin_amdgcn_readfirstlane((int)(cm.ps > 0 && ldexp(nrm, 1 - cm.ps) <= 1.49));
  cm.ps -= deg20;
  for (int e = tid; e < MGR_SLOTS * IMGR; e += 256) c3p_md_lds[e] = 0.0;
  for (int k = 0; k < K; ++k) {
    const double* s = A.signals + ((long)cm.sample * K + k) * A.N + cm.n0;
    for (int t = tid; t < cm.len; t += 256) cm.sg[k * A.Lmax + t] = s[t];
  }
  cm.pr = 0;
  cm.t18 = 1;
  cm.scale = ldexp(1.0, -cm.ps);
  __syncthreads();
  if (deg20) {
    switch (wave) {
      case 0: midd_grad_real_body<NIG, NJ, W, 0, true>(A, cm, chain, red); break;
      case 1: midd_grad_real_body<NIG, NJ, W, 1, true>(A, cm, chain, red); break;
      case 2: midd_grad_real_body<NIG, NJ, W, 2, true>(A, cm, chain, red); break;
      default: midd_grad_real_body<NIG, NJ, W, 3, true>(A, cm, chain, red); break;
    }
  } else {
    switch (wave) {
      case 0: midd_grad_real_body<NIG, NJ, W, 0, false>(A, cm, chain, red); break;
      case 1: midd_grad_real_body<NIG, NJ, W, 1, false>(A, cm, chain, red); break;
      case 2: midd_grad_real_body<NIG, NJ, W, 2, false>(A, cm, chain, red); break;
      default: midd_grad_real_body<NIG, NJ, W, 3, false>(A, cm, chain, red); break;
    }
  }
}

#if C3P_MIDD_HAS(3)
template <int NIG, int NJ, int W>
hipError_t launch_grad_t(const MidGradArgs& A, hipStream_t st) {
  constexpr int IMG = MD<NIG, NJ>::ROWS * W;
  const size_t lds = (size_t)(4 * IMG + A.K * A.Lmax) * sizeof(double);
  if (lds > 158 * 1024) return hipErrorInvalidValue;
  auto kern = midd_grad_kernel<NIG, NJ, W>;
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  C3P_LAUNCH(kern, dim3((unsigned)((long)A.B * A.S)), dim3(256), lds, st, A);
  return hipGetLastError();
}

template <int NIG, int NJ, int W, bool XG>
hipError_t launch_grad_general_x(const MidGradArgs& A, hipStream_t st) {
  constexpr int IMG = MD<NIG, NJ>::ROWS * W;
  const size_t lds = (size_t)(4 * IMG + (XG ? 0 : A.K) * A.Lmax) * sizeof(double);
  if (lds > 158 * 1024) return hipErrorInvalidValue;
  auto kern = midd_grad_general_kernel<NIG, NJ, W, XG>;
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  C3P_LAUNCH(kern, dim3((unsigned)((long)A.B * A.S)), dim3(256), lds, st, A);
  return hipGetLastError();
}
template <int NIG, int NJ, int W>
hipError_t launch_grad_general_t(const MidGradArgs& A, hipStream_t st) {
  return A.hs != nullptr ? launch_grad_general_x<NIG, NJ, W, true>(A, st) : launch_grad_general_x<NIG, NJ, W, false>(A, st);
}

// *launched = false (and hipSuccess) when the real sweep does not apply: more control lines than the kernel keeps in
// registers, or the eight real images and the segment's control amplitudes do not fit the LDS
template <int NIG, int NJ, int W>
hipError_t launch_grad_real_t(const MidGradArgs& A, hipStream_t st, bool* launched) {
  constexpr int IMGR = 16 * MDR<NIG, W>::NIGR * MDR<NIG, W>::WI;
  const size_t lds = (size_t)(MGR_SLOTS * IMGR + A.K * A.Lmax) * sizeof(double);
  *launched = false;
  if (A.K > MDR<NIG, W>::KP || lds > 158 * 1024) return hipSuccess;
  auto kern = midd_grad_real_kernel<NIG, NJ, W>;
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  C3P_LAUNCH(kern, dim3((unsigned)((long)A.B * A.S)), dim3(256), lds, st, A);
  *launched = true;
  return hipGetLastError();
}

#endif

}  // namespace

#if C3P_MIDD_HAS(1)
// geometry classes: Dm -> (NIG, NJ, W)
bool c3p_midd_geometry(int Dm, int* nig, int* nj, int* w) {
  if (Dm < 13 || Dm > 40) return false;
  const int NBI = (Dm + 1) / 2;
  *nig = (NBI + 3) / 4;
  *nj = (Dm + 3) / 4;
  switch (*nj) {
    case 4: *w = 17; break;   // D 13..16
    case 5: *w = 21; break;   // D 17..20
    case 6: *w = 25; break;   // D 21..24
    case 7: *w = 29; break;   // D 25..28
    case 8: *w = 33; break;   // D 29..32
    case 9: *w = 37; break;   // D 33..36
    default: *w = 41; break;  // D 37..40
  }
  return true;
}

size_t c3p_midd_table_doubles(int Dm, int K) {
  int nig, nj, w;
  if (!c3p_midd_geometry(Dm, &nig, &nj, &w)) return 0;
  return (size_t)(1 + K) * ((size_t)16 * nig * w + 4);
}

size_t c3p_midd_lds_bytes(int Dm, int K, int Lmax) {
  int nig, nj, w;
  if (!c3p_midd_geometry(Dm, &nig, &nj, &w)) return 0;
  (void)Lmax;  // the chain kernel stages the control amplitudes in chunks of SGC slices
  return ((size_t)3 * 16 * nig * w + (size_t)K * SGC) * sizeof(double);
}

size_t c3p_midd_grad_image_bytes(int Dm) {
  int nig, nj, w;
  if (!c3p_midd_geometry(Dm, &nig, &nj, &w)) return 0;
  const int nigr = (nig + 1) / 2;
  const size_t general = (size_t)4 * 16 * nig * w, real = (size_t)8 * 16 * nigr * (nigr <= 2 ? 32 : w);
  return (general > real ? general : real) * sizeof(double);
}

hipError_t c3p_launch_midd_chain(const MidArgs& A, hipStream_t st) {
  int nig, nj, w;
  if (!c3p_midd_geometry(A.Dm, &nig, &nj, &w)) return hipErrorInvalidValue;
  if (nig == 2 && nj == 4) return launch_t<2, 4, 17>(A, st);
  if (nig == 3 && nj == 5) return launch_t<3, 5, 21>(A, st);
  if (nig == 3 && nj == 6) return launch_t<3, 6, 25>(A, st);
  if (nig == 4 && nj == 7) return launch_t<4, 7, 29>(A, st);
  if (nig == 4 && nj == 8) return launch_t<4, 8, 33>(A, st);
  if (nig == 5 && nj == 9) return launch_t<5, 9, 37>(A, st);
  if (nig == 5 && nj == 10) return launch_t<5, 10, 41>(A, st);
  return hipErrorInvalidValue;
}

hipError_t c3p_launch_midd_prep(const MidPrepArgs& P, int nsamp, hipStream_t st) {
  C3P_LAUNCH(midd_prep_kernel, dim3((unsigned)(nsamp * (1 + P.K))), dim3(256), 0, st, P);
  return hipGetLastError();
}
#endif

#if C3P_MIDD_HAS(2)
hipError_t c3p_launch_midd_real(const MidArgs& A, hipStream_t st) {
  int nig, nj, w;
  if (!c3p_midd_geometry(A.Dm, &nig, &nj, &w)) return hipErrorInvalidValue;
  if (nig == 2 && nj == 4) return launch_real_t<2, 4, 17>(A, st);
  if (nig == 3 && nj == 5) return launch_real_t<3, 5, 21>(A, st);
  if (nig == 3 && nj == 6) return launch_real_t<3, 6, 25>(A, st);
  if (nig == 4 && nj == 7) return launch_real_t<4, 7, 29>(A, st);
  if (nig == 4 && nj == 8) return launch_real_t<4, 8, 33>(A, st);
  if (nig == 5 && nj == 9) return launch_real_t<5, 9, 37>(A, st);
  if (nig == 5 && nj == 10) return launch_real_t<5, 10, 41>(A, st);
  return hipErrorInvalidValue;
}
#endif

#if C3P_MIDD_HAS(3)
namespace {
hipError_t launch_grad_real(const MidGradArgs& A, int nig, int nj, hipStream_t st, bool* launched) {
  if (nig == 2 && nj == 4) return launch_grad_real_t<2, 4, 17>(A, st, launched);
  if (nig == 3 && nj == 5) return launch_grad_real_t<3, 5, 21>(A, st, launched);
  if (nig == 3 && nj == 6) return launch_grad_real_t<3, 6, 25>(A, st, launched);
  if (nig == 4 && nj == 7) return launch_grad_real_t<4, 7, 29>(A, st, launched);
  if (nig == 4 && nj == 8) return launch_grad_real_t<4, 8, 33>(A, st, launched);
  if (nig == 5 && nj == 9) return launch_grad_real_t<5, 9, 37>(A, st, launched);
  if (nig == 5 && nj == 10) return launch_grad_real_t<5, 10, 41>(A, st, launched);
  return hipErrorInvalidValue;
}
}  // namespace

hipError_t c3p_launch_midd_grad_general(const MidGradArgs& A, hipStream_t st) {
  int nig, nj, w;
  if (!c3p_midd_geometry(A.Dm, &nig, &nj, &w)) return hipErrorInvalidValue;
  if (nig == 2 && nj == 4) return launch_grad_general_t<2, 4, 17>(A, st);
  if (nig == 3 && nj == 5) return launch_grad_general_t<3, 5, 21>(A, st);
  if (nig == 3 && nj == 6) return launch_grad_general_t<3, 6, 25>(A, st);
  if (nig == 4 && nj == 7) return launch_grad_general_t<4, 7, 29>(A, st);
  if (nig == 4 && nj == 8) return launch_grad_general_t<4, 8, 33>(A, st);
  if (nig == 5 && nj == 9) return launch_grad_general_t<5, 9, 37>(A, st);
  if (nig == 5 && nj == 10) return launch_grad_general_t<5, 10, 41>(A, st);
  return hipErrorInvalidValue;
}

hipError_t c3p_launch_midd_grad(const MidGradArgs& A_, hipStream_t st) {
  int nig, nj, w;
  if (!c3p_midd_geometry(A_.Dm, &nig, &nj, &w)) return hipErrorInvalidValue;
  MidGradArgs A = A_;
  A.skip_real = 0;
  // real Hamiltonians first (reverse mode through the cos / sin evaluation), then the general sweep for the rest; the
  // per-slice generator cotangents (zout) only exist in the general sweep
  if (A.zout == nullptr && !c3p_opt_on(C3P_OPT_no_real_grad)) {
    bool launched = false;
    hipError_t e = launch_grad_real(A, nig, nj, st, &launched);
    if (e != hipSuccess) return e;
    if (launched) A.skip_real = 1;
  }
  if (nig == 2 && nj == 4) return launch_grad_t<2, 4, 17>(A, st);
  if (nig == 3 && nj == 5) return launch_grad_t<3, 5, 21>(A, st);
  if (nig == 3 && nj == 6) return launch_grad_t<3, 6, 25>(A, st);
  if (nig == 4 && nj == 7) return launch_grad_t<4, 7, 29>(A, st);
  if (nig == 4 && nj == 8) return launch_grad_t<4, 8, 33>(A, st);
  if (nig == 5 && nj == 9) return launch_grad_t<5, 9, 37>(A, st);
  if (nig == 5 && nj == 10) return launch_grad_t<5, 10, 41>(A, st);
  return hipErrorInvalidValue;
}
#endif
