// Lindblad superoperator chains in REAL arithmetic on the f64 matrix cores: Dm = D^2 = 49, 64 (padded to 65), 81 -- the
// 81 x 81 superoperators of two qutrits (BASELINE cfg4, c3/libraries/propagation.py:551-585).
//
// The Lindblad generator L(rho) = -i [H, rho] + sum_c C rho C^+ - 1/2 {C^+ C, rho} maps Hermitian matrices to Hermitian
// matrices whenever H is Hermitian (whatever the collapse operators are).  In a basis of Hermitian matrices
//     E_ii,   (E_ij + E_ji) / sqrt 2,   i (E_ji - E_ij) / sqrt 2        (i < j)
// the superoperator is therefore a REAL Dm x Dm matrix: X' = T X T^+ with the (sparse, unitary) change of basis T from the
// row-major vectorisation the reference uses (tf_utils.py:271-280).  The whole chain runs in that basis,
//     U = prod_n exp(X_n) = T^+ ( prod_n exp(X'_n) ) T,
// with real products -- a quarter of the multiplications of a complex product (a third of the three-product form of
// c3p_regd.hip) and half the bytes per matrix: every matrix a slice needs (X, X^2, the T18 combinations, the running
// product) now fits the register file, so NOTHING leaves the CU between the tables and the segment product (the complex
// kernel parks five tile sets per slice in a global arena: 1 TB per cfg4 launch).  The basis change costs two tiny kernels:
// the generator tables are transformed once per call (regr_prep_kernel, four generator elements per table element) and
// the segment products / slice propagators are transformed back in place (hb_to_complex_kernel).  Same result up to
// rounding (T is unitary with entries 0, 1, +-1/sqrt 2, +-i/sqrt 2).
//
// Non-Hermitian Hamiltonians (the C ABI takes any matrix) leave an imaginary part in X': regr_prep_kernel flags every
// table whose imaginary part exceeds 1e-14 of its largest element, the real kernel takes the samples whose tables are all
// flagged real and the complex kernel of c3p_regd.hip the others (both are launched; a workgroup skips what is not its own).
//
// Layout and product loop follow c3p_regd.hip: one workgroup of four waves (one per SIMD, 512 registers), a (Dm-1)^2
// CORE in registers (wave w owns 4 n columns as n x n tiles in the C/D layout of v_mfma_f64_4x4x4_4b_f64, which is also a
// valid B operand once rotated by 0..3 lane groups) plus a one-element BORDER in LDS slots; the left operand streams from
// one LDS image (real, row stride LD); exp = T18 (Bader-Blanes-Casas, 5 products) + s squarings, then the chain product.
#include <cstdio>
#include <utility>

#include "c3p_common.h"
#include "c3p_kernels.h"
#include "c3p_midd.h"
#include "c3p_regd.h"
#include "c3p_regr_common.h"

extern __shared__ __attribute__((aligned(16))) double c3p_rr_lds[];

// Round 5: the running product U of the segment is parked in LDS between chain products (one tile set, 51 KB at Dm = 81;
// every lane reads back its own elements: no barrier) instead of staying in registers for the whole slice.  Five tile sets
// (R, accumulators, X / B2, X^2 / B3, U = 250 registers) + operands did not fit the 256 VALU-addressable registers: the
// compiler kept part of them in the accumulation registers and 23 doubles per lane in SCRATCH, and the s_memtime probes
// (-DC3P_REGR_TIMING) showed the phases between products -- a few hundred instructions each -- taking 9 - 17 k cycles: scratch
// round trips.  -DC3P_REGR_ULDS=0 builds the round-4 form.
#ifndef C3P_REGR_ULDS
#define C3P_REGR_ULDS 1
#endif

namespace {

// The chain loop of one wave.  NW = 4: one wave per SIMD, wave w owns the 4 NRG columns of column group w (NJ = NRG column
// blocks).  NW = 8: two waves per SIMD (waves w and w + 4 share one), the column blocks of a group are split between them
// (NJ = ceil(NRG / 2) for the first, floor for the second: jj0 = first block of this wave inside its group) -- the
// non-MFMA instructions of one wave (A-fragment reads, tile rotations) issue under the MFMAs of the other.
// LEAN: the two-workgroups-per-CU form (256 registers per wave, half the LDS): only the operands of the running product
// (Rm, acc) and ONE parked set (X^2, then B3) stay in registers; X is re-assembled from the (L2-resident) tables where the
// T18 combinations need it, and B2 and the running product U are parked in a per-workgroup global arena (two sets,
// 0.2 MB per slice and chain against the 1.0 MB of the complex kernel).  What it buys: the second workgroup's MFMAs run
// under this one's latencies -- the phases between products (element-wise work, LDS round trips, barriers, table loads)
// are ~40 % of a slice with one workgroup per CU.
template <int NRG, int NJ, int NW, bool DUS, bool UNROLL, bool LEAN, bool QT = false>
__device__ __forceinline__ void regr_chain_body(const MidArgs& A, long long* dbg, double* arena_base, const int cg, const int jj0, const int wave) {
  using G = RR<NRG>;
  constexpr int DM = G::DM, LD = G::LD, BS = G::BS, DMP = G::DMP;
  constexpr int TSET = G::TSET;
  constexpr int RR_THREADS = 64 * NW;
  constexpr bool ULDS = (C3P_REGR_ULDS != 0) && !LEAN && NW == 4;  // U parked in LDS (see the top of the file)
  constexpr bool UOUT = LEAN || ULDS;                              // U does not live in registers
  const bool second = jj0 != 0;                         // the second wave of a pair
  const bool col_owner = (NW == 4) || second;           // column DM-1 of a product: the lighter wave of the pair
  const bool corner_owner = (cg == 0) && !second;
  const int tid0 = threadIdx.x;
  int tid = tid0, lane = tid & 63;
  int q = lane >> 4, b = (lane >> 2) & 3, p = lane & 3;
  double* img = c3p_rr_lds;
  double* brd = img + G::IMG_D;
  double* cpart = brd + S_NSLOT * BS;
  double* rpart = cpart + 4 * DMP;
  double* sg = rpart + 4 * DMP;
  double* red = sg + RR_KMAX * RR_CH;
  double* ulds = red + 2 * RR_MAXWAVES;  // ULDS: element (tile, thread) of U at tile * 256 + tid (conflict free)
  const int col0 = 4 * NRG * cg + 4 * jj0;  // first column of this wave
  int rowC = 4 * b + q;                     // row of a C/D-layout element inside its row group
  int rowA = 4 * b + p;                     // row of an A-fragment element inside its row group
  const int K = A.K;
  // lane indices are re-derived from an opaque copy of the thread id at the start of every phase: otherwise the compiler
  // hoists every address / mask that depends on them out of the slice loop and spills them
  auto refresh = [&]() {
    int t_ = tid0;
    asm volatile("" : "+v"(t_));
    tid = t_;
    lane = tid & 63;
    q = lane >> 4, b = (lane >> 2) & 3, p = lane & 3;
    rowC = 4 * b + q;
    rowA = 4 * b + p;
  };

#ifdef C3P_REGR_TIMING
  long long tacc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  long long tq = 0;
#define RR_TICK(i)                  \
  {                                 \
    const long long tn = clock64(); \
    tacc[i] += tn - tq;             \
    tq = tn;                        \
  }
#else
#define RR_TICK(i)
#endif
  double Rm[NRG][NJ];   // right operand of the next product
  double acc[NRG][NJ];  // accumulators = the product
  double Xs[LEAN ? 1 : NRG][LEAN ? 1 : NJ];  // X, later B2 (LEAN: re-assembled / parked in the arena)
  double A2s[NRG][NJ];  // X^2, later B3
  double Us[UOUT ? 1 : NRG][UOUT ? 1 : NJ];  // running product of the segment (LEAN: in the arena, ULDS: in LDS)
  double* arena = arena_base + (long)blockIdx.x * 2 * TSET;  // LEAN: sets {B2, U} of this workgroup
  auto park = [&](int set, const double (&v)[NRG][NJ]) {
    double* dst = rr_ubase(arena + set * TSET);
#pragma unroll
    for (int Ig = 0; Ig < NRG; ++Ig)
#pragma unroll
      for (int jj = 0; jj < NJ; ++jj) dst[(Ig * NJ + jj) * RR_THREADS + tid] = v[Ig][jj];
  };
  auto unpark = [&](int set, double (&v)[NRG][NJ]) {
    const double* src = rr_ubase(arena + set * TSET);
#pragma unroll
    for (int Ig = 0; Ig < NRG; ++Ig)
#pragma unroll
      for (int jj = 0; jj < NJ; ++jj) v[Ig][jj] = src[(Ig * NJ + jj) * RR_THREADS + tid];
  };

  auto park_u = [&](const double (&v)[NRG][NJ]) {
#pragma unroll
    for (int Ig = 0; Ig < NRG; ++Ig)
#pragma unroll
      for (int jj = 0; jj < NJ; ++jj) ulds[(Ig * NJ + jj) * RR_THREADS + tid] = v[Ig][jj];
  };
  auto unpark_u = [&](double (&v)[NRG][NJ]) {
#pragma unroll
    for (int Ig = 0; Ig < NRG; ++Ig)
#pragma unroll
      for (int jj = 0; jj < NJ; ++jj) v[Ig][jj] = ulds[(Ig * NJ + jj) * RR_THREADS + tid];
  };

  auto mfma = [](double a, double bb, double c) -> double { return __builtin_amdgcn_mfma_f64_4x4x4f64(a, bb, c, 0, 0, 0); };
  auto zero_acc = [&]() {
#pragma unroll
    for (int Ig = 0; Ig < NRG; ++Ig)
#pragma unroll
      for (int jj = 0; jj < NJ; ++jj) acc[Ig][jj] = 0.0;
  };
  // a border element (slot index tid) also sits in the image: row DM-1 or column DM-1
  auto border_to_image = [&](double v) {
    if (tid < DM) img[(DM - 1) * LD + tid] = v;
    else if (tid < 2 * DM - 1) img[(tid - DM) * LD + DM - 1] = v;
  };

  // C = (init) + L R: L = the LDS image, R = Rm with borders in slot sr; the accumulators carry the initial value (its
  // border in slot si, or si < 0); border of C -> slot sd (by finalize, after the barrier that publishes the partials).
  // Column DM-1 of C is one more B column (the border column of R, from its slot): four partial sums, one per column
  // group; row DM-1 of C uses the tiles of R block by block (A operand = row DM-1 of L in the i = 0
  // row of every block): four partial sums, one per MFMA block.  The k = DM-1 terms: finalize / the rank-1 update.
  double cornerA = 0.0, cornerR = 0.0;
  auto product = [&](int sr) {
    const double* rb = brd + sr * BS;
#pragma unroll
    for (int Ig = 0; Ig < NRG; ++Ig)
#pragma unroll
      for (int jj = 0; jj < NJ; ++jj) asm volatile("" : "+v"(Rm[Ig][jj]));
    double cP[NRG];
    double rP[NJ + 1];
#pragma unroll
    for (int Ig = 0; Ig < NRG; ++Ig) cP[Ig] = 0.0;
#pragma unroll
    for (int jj = 0; jj <= NJ; ++jj) rP[jj] = 0.0;
    {
      // one pass = the four K-steps of one row group of R (row 0 of Rm; the rows move up at the end of the pass).  K-step
      // s: MFMA block b multiplies by block (b - s) mod 4 of the tile, its A fragment is taken at the k of that block.
      const double* pa = img + rowA * LD;
      const double* pr = img + (DM - 1) * LD + rowC;  // row DM-1 of L at this lane's k
      int ko[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) ko[s] = 4 * ((b - s) & 3) + q;
      double aC[NRG], aN[NRG];
      double br[NJ];
#pragma unroll
      for (int Ig = 0; Ig < NRG; ++Ig) aC[Ig] = pa[16 * Ig * LD + ko[0]];
#pragma unroll
      for (int jj = 0; jj < NJ; ++jj) br[jj] = Rm[0][jj];
      RR_TICK(10)
      // One pass.  IT >= 0: pass IT of the UNROLLED loop (row IT of Rm and the image offsets are static: no register
      // rotation -- the rolled form moves the tiles of R up one row per pass, ~90 register-file copies per 106 MFMAs once R
      // lives in the accumulation registers); IT = -1: the rolled form (row 0, pointers advance).
      int itr = 0;
      auto pass = [&](auto it_) {
        constexpr int IT = decltype(it_)::value;
        constexpr int R0 = IT < 0 ? 0 : IT;
        constexpr int R1 = IT < 0 ? 1 : (IT + 1 < NRG ? IT + 1 : IT);
        constexpr int OFF = IT < 0 ? 0 : 16 * IT;
        rr_static_for<4>([&](auto s_) {
          constexpr int sx = decltype(s_)::value;
          constexpr int sn = (sx + 1) & 3, rn = (sx == 3) ? R1 : R0;  // next step: rotation, row of Rm
          // source order = issue order (sched_barrier): one MFMA, then at most one piece of the next step's operand
          // preparation (an A-fragment load or a tile rotation), so that it issues in the shadow of the matrix pipe
          double brN[NJ];
          rr_static_for<NRG>([&](auto Ig_) {
            constexpr int Ig = decltype(Ig_)::value;
            rr_static_for<NJ>([&](auto jj_) {
              constexpr int jj = decltype(jj_)::value;
              constexpr int u = Ig * NJ + jj;  // preparation slot: NRG + NJ pieces over NRG * NJ slots
              constexpr int OPS = (NRG + NJ + NRG * NJ - 1) / (NRG * NJ);
              acc[Ig][jj] = mfma(aC[Ig], br[jj], acc[Ig][jj]);
              rr_static_for<OPS>([&](auto o_) {
                constexpr int op = u * OPS + decltype(o_)::value;
                if constexpr (op < NRG) {
                  aN[op] = pa[16 * op * LD + OFF + (sx == 3 ? 16 : 0) + ko[sn]];  // (the very last prefetch is unused)
                } else if constexpr (op < NRG + NJ) {
                  brN[op - NRG] = rr_rot<sn>(Rm[rn][op - NRG]);
                }
              });
              __builtin_amdgcn_sched_barrier(0);
            });
          });
#pragma unroll
          for (int Ig = 0; Ig < NRG; ++Ig) aC[Ig] = aN[Ig];
#pragma unroll
          for (int jj = 0; jj < NJ; ++jj) br[jj] = brN[jj];
          // (-DC3P_REGR_BB: a basic-block boundary per K-step -- an opaque scalar branch that jumps over an s_sleep -- as in rounds 2 - 4,
          // when it kept the scheduler from merging K-steps; with the per-instruction sched_barrier above it only costs its taken
          // branch: 108.9 -> 106.1 ms per 512-sample cfg4 batch without it)
#ifdef C3P_REGR_BB
          if (rr_opq(0) != 0) asm volatile("s_sleep 1");
#endif
        });
        // row DM-1 (and, on one wave, the corner): partial sums over the k of each MFMA block
        {
          const double va = pr[OFF];
          const double ar = (p == 0) ? va : 0.0;
#pragma unroll
          for (int jj = 0; jj < NJ; ++jj) rP[jj] = mfma(ar, Rm[R0][jj], rP[jj]);
          if (corner_owner) {
            const double vb = rb[DM + (IT < 0 ? 16 * itr : OFF) + rowC];
            const double cb = (p == 0) ? vb : 0.0;
            rP[NJ] = mfma(ar, cb, rP[NJ]);
          }
        }
        if constexpr (IT < 0) {
#pragma unroll
          for (int Ig = 0; Ig + 1 < NRG; ++Ig)
#pragma unroll
            for (int jj = 0; jj < NJ; ++jj) Rm[Ig][jj] = Rm[Ig + 1][jj];
          pa += 16;
          pr += 16;
          ++itr;
        }
      };
      if constexpr (UNROLL) {
        rr_static_for<NRG>([&](auto it_) { pass(it_); });
      } else {
#pragma unroll 1
        for (int it = 0; it < NRG; ++it) pass(std::integral_constant<int, -1>{});
      }
    }
    // Column DM-1 of C: one more B column (the border column of R, from its slot).  MFMA block b of column group g takes
    // the k of block (b - g) mod 4 of every row group: over the four groups every block meets every k -- four partial
    // sums, one per group.  Its own short loop (A fragments one row group ahead): inside the main loop, as "the K-step
    // s == g", the branch was if-converted into four times the MFMAs on the two-waves-per-SIMD instance.
    // operands of the rank-1 update (k = DM-1) further down: fetched here, so that their LDS round trip runs under the column loop
    double a80[NRG], b80[NJ];
#pragma unroll
    for (int Ig = 0; Ig < NRG; ++Ig) a80[Ig] = img[(16 * Ig + rowC) * LD + DM - 1];
#pragma unroll
    for (int jj = 0; jj < NJ; ++jj) b80[jj] = rb[col0 + 4 * jj + p];
    if (col_owner) {
      const int kb = 4 * ((b - cg) & 3) + q;
      const double* pa = img + rowA * LD + kb;
      const double* pc = rb + DM + kb;
      double aB[NRG], aBn[NRG];
#pragma unroll
      for (int Ig = 0; Ig < NRG; ++Ig) aB[Ig] = pa[16 * Ig * LD];
#pragma unroll 1
      for (int it = 0; it < NRG; ++it) {
        const double vb = pc[16 * it];
#pragma unroll
        for (int Ig = 0; Ig < NRG; ++Ig) aBn[Ig] = pa[16 * Ig * LD + 16 * it + (it + 1 < NRG ? 16 : 0)];
        const double cb = (p == 0) ? vb : 0.0;
#pragma unroll
        for (int Ig = 0; Ig < NRG; ++Ig) cP[Ig] = mfma(aB[Ig], cb, cP[Ig]);
#pragma unroll
        for (int Ig = 0; Ig < NRG; ++Ig) aB[Ig] = aBn[Ig];
      }
    }
    RR_TICK(0)
    if (q == 0) {
#pragma unroll
      for (int jj = 0; jj < NJ; ++jj) rpart[b * DMP + col0 + 4 * jj + p] = rP[jj];
      if (corner_owner && p == 0) rpart[b * DMP + DM - 1] = rP[NJ];
    }
    if (p == 0 && col_owner) {
#pragma unroll
      for (int Ig = 0; Ig < NRG; ++Ig) cpart[cg * DMP + 16 * Ig + rowC] = cP[Ig];
    }
    cornerA = img[(DM - 1) * LD + DM - 1];
    cornerR = rb[BS - 1];
    // k = DM-1: rank-1 update of the core
    {
#pragma unroll
      for (int Ig = 0; Ig < NRG; ++Ig)
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj) acc[Ig][jj] = fma(a80[Ig], b80[jj], acc[Ig][jj]);
    }
  };
  // after the barrier: the border of the product, one element per lane
  auto finalize = [&](int sr, int si, int sd) {
    if (tid < BS) {
      double v, f;
      if (tid < DM || tid == BS - 1) {
        const int j = tid < DM ? tid : DM - 1;
        v = (rpart[j] + rpart[DMP + j]) + (rpart[2 * DMP + j] + rpart[3 * DMP + j]);
        f = brd[sr * BS + tid];
        v = fma(cornerA, f, v);
      } else {
        const int i = tid - DM;
        v = (cpart[i] + cpart[DMP + i]) + (cpart[2 * DMP + i] + cpart[3 * DMP + i]);
        f = img[i * LD + DM - 1];
        v = fma(f, cornerR, v);
      }
      if (si >= 0) v += brd[si * BS + tid];
      brd[sd * BS + tid] = v;
    }
  };
  auto image_from_C = [&]() {
#pragma unroll
    for (int Ig = 0; Ig < NRG; ++Ig)
#pragma unroll
      for (int jj = 0; jj < NJ; ++jj) img[(16 * Ig + rowC) * LD + col0 + 4 * jj + p] = acc[Ig][jj];
  };
  auto R_from_C = [&]() {
#pragma unroll
    for (int Ig = 0; Ig < NRG; ++Ig)
#pragma unroll
      for (int jj = 0; jj < NJ; ++jj) Rm[Ig][jj] = acc[Ig][jj];
  };
  // dst[row][col] = f C (f = e^{trace shift}), row-major REAL DR x DR; borders from slot.  DR = A.Dm <= DM is the true
  // dimension (smaller matrices run zero padded: the padding stays decoupled through every product).
  const int DR = A.Dm;
  auto store_out = [&](double* dst_, int slot, double f) {
    double* dst = rr_ubase(dst_);
#pragma unroll
    for (int Ig = 0; Ig < NRG; ++Ig)
#pragma unroll
      for (int jj = 0; jj < NJ; ++jj) {
        const int row = 16 * Ig + rowC, col = col0 + 4 * jj + p;
        if (row < DR && col < DR) dst[(long)row * DR + col] = f * acc[Ig][jj];
      }
    if (tid < BS - 1) {
      const double v = brd[slot * BS + tid];
      const int row = tid < DM ? DM - 1 : tid - DM, col = tid < DM ? tid : DM - 1;
      if (row < DR && col < DR) dst[(long)row * DR + col] = f * v;
    }
  };

  // QT: the transposed running product of the segment after every slice (backward sweep in the Hermitian basis, c3p_regrg.hip)
  auto store_qT = [&](double* dst_, int slot, double f) {
    double* dst = rr_ubase(dst_);
#pragma unroll
    for (int Ig = 0; Ig < NRG; ++Ig)
#pragma unroll
      for (int jj = 0; jj < NJ; ++jj) {
        const int row = 16 * Ig + rowC, col = col0 + 4 * jj + p;
        if (row < DR && col < DR) dst[(long)col * DR + row] = f * acc[Ig][jj];
      }
    if (tid < BS - 1) {
      const double v = brd[slot * BS + tid];
      const int row = tid < DM ? DM - 1 : tid - DM, col = tid < DM ? tid : DM - 1;
      if (row < DR && col < DR) dst[(long)col * DR + row] = f * v;
    }
  };
  const long nchains = (long)A.B * A.S;
  const long mat_c = (long)DR * DR;  // complex elements of an output slot; the real result goes to its second half
  for (long chain = blockIdx.x; chain < nchains; chain += gridDim.x) {
    const int sample = (int)(chain / A.S);
    if (!c3p_hb_sample_is_real(A.hb_tabflag, A.tab_per_sample ? sample : 0, K)) continue;  // the complex kernel's
    const int seg = (int)(chain - (long)sample * A.S);
    const int n0 = (int)(((long)seg * A.N) / A.S);
    const int n1 = (int)(((long)(seg + 1) * A.N) / A.S);
    const int len = n1 - n0;
    const double* tabs = A.hb_tables + (long)(A.tab_per_sample ? sample : 0) * (1 + K) * G::TAB_D;
    auto meta = [&](int k1) -> const double* { return tabs + (long)k1 * G::TAB_D + (TSET + BS); };
    __syncthreads();  // the previous chain is done with the LDS
    // plan: squarings from ||G0||_1 + sum_k max_t |c_k(t)| ||G_k||_1 over the segment
    double nrm = meta(0)[1];
    double kmaxv[RR_KMAX];
    for (int k = 0; k < K; ++k) {
      const double* s = A.signals + ((long)sample * K + k) * A.N + n0;
      double cmax = 0.0;
      for (int t = tid; t < len; t += RR_THREADS) cmax = fmax(cmax, fabs(s[t]));
      for (int o = 32; o >= 1; o >>= 1) cmax = fmax(cmax, __shfl_xor(cmax, o));
      if (lane == 0) red[wave] = cmax;
      __syncthreads();
      cmax = 0.0;
      for (int w = 0; w < NW; ++w) cmax = fmax(cmax, red[w]);
      __syncthreads();
      nrm = fma(cmax, meta(k + 1)[1], nrm);
      kmaxv[k] = cmax;
    }
    nrm = rr_rfl(nrm);
    // Round 6: the generator is real in the Hermitian basis, skew-symmetric up to the dissipator (H is Hermitian on this path): when the
    // symmetric part is small the economised T18 parameters apply (radius 2.0 instead of 1.13: cfg4 needs no squaring)
    double nsym = meta(0)[3];
    for (int k = 0; k < K; ++k) nsym = fma(kmaxv[k], meta(k + 1)[3], nsym);
    const int econ = __builtin_amdgcn_readfirstlane((int)(rr_rfl(nsym) <= C3P_T18N_MAX_NONNORMAL && !(A.no_t18n & 1)));
    const double* tc = c3p_t18_tab[econ];
    int s18 = 0;
    {
      double pth = econ ? C3P_T18N_THETA : C3P_T18_THETA;
      while (pth < nrm && s18 < 40) {
        pth *= 2.0;
        ++s18;
      }
    }
    const int ps = __builtin_amdgcn_readfirstlane(s18);
    const double scale = ldexp(1.0, -ps);
#ifdef C3P_REGR_PLAN_PRINT
    if (blockIdx.x < 4 && tid == 0) printf("regr chain %d: norm bound %.4f, symmetric part %.4f, economised %d, squarings %d\n", (int)blockIdx.x, nrm, nsym, econ, ps);
#endif

    double mu = 0.0, mus = 0.0;
    auto stage_signals = [&](int t) {  // slices [t, t + RR_CH) of the segment
      for (int e = tid; e < K * RR_CH; e += RR_THREADS) {
        const int k = e / RR_CH, tt = e - k * RR_CH;
        sg[e] = (t + tt < len) ? A.signals[((long)sample * K + k) * A.N + n0 + t + tt] : 0.0;
      }
    };
    // tiles of X = 2^-s (G0 + sum_k c_k G_k) for slice tt of the staged chunk (tables: L2 resident, shared by all workgroups)
    auto assemble_tiles = [&](int tt, auto& dst) {
      for (int k1 = 0; k1 <= K; ++k1) {
        const double* src = rr_ubase(tabs + (long)k1 * G::TAB_D);
        double v[NRG][NJ];
#pragma unroll
        for (int Ig = 0; Ig < NRG; ++Ig)
#pragma unroll
          for (int jj = 0; jj < NJ; ++jj) v[Ig][jj] = src[(Ig * NRG + jj0 + jj) * 256 + cg * 64 + lane];
        __builtin_amdgcn_sched_barrier(0);
        const double w = k1 ? scale * sg[(k1 - 1) * RR_CH + tt] : scale;
#pragma unroll
        for (int Ig = 0; Ig < NRG; ++Ig)
#pragma unroll
          for (int jj = 0; jj < NJ; ++jj) dst[Ig][jj] = k1 ? fma(w, v[Ig][jj], dst[Ig][jj]) : w * v[Ig][jj];
      }
    };
    // X = 2^-s (G0 + sum_k c_k G_k) for slice tt of the staged chunk (tables: L2 resident, shared by all workgroups):
    // tiles -> Xs, Rm and the image, borders -> slot M0 and the image; trace shift of the slice -> mu
    auto assemble = [&](int tt) {
      double bv = 0.0;
      if (K <= 2) {
        // (cfg4: K = 2) the trace shifts and border elements of all three tables in ONE round trip to L2 -- the rolled loop below
        // waited for a load per control line, three dependent round trips in the phase after the chain product
        const int btid = tid < BS ? tid : 0;
        double mk[3], bk[3];
#pragma unroll
        for (int k1 = 0; k1 < 3; ++k1) {
          const int ks = k1 <= K ? k1 : 0;
          mk[k1] = meta(ks)[0];
          bk[k1] = tabs[(long)ks * G::TAB_D + TSET + btid];
        }
        const double c1 = K >= 1 ? sg[0 * RR_CH + tt] : 0.0, c2 = K >= 2 ? sg[1 * RR_CH + tt] : 0.0;
        mu = fma(c2, mk[2], fma(c1, mk[1], mk[0]));
        bv = scale * fma(c2, bk[2], fma(c1, bk[1], bk[0]));
      } else {
        mu = meta(0)[0];
        if (tid < BS) bv = scale * tabs[TSET + tid];
        for (int k = 0; k < K; ++k) {
          const double c = sg[k * RR_CH + tt];
          mu = fma(c, meta(k + 1)[0], mu);
          if (tid < BS) bv = fma(scale * c, tabs[(long)(k + 1) * G::TAB_D + TSET + tid], bv);
        }
      }
      if (tid < BS) {
        brd[S_M0 * BS + tid] = bv;
        border_to_image(bv);
      }
      if constexpr (LEAN) {
        assemble_tiles(tt, Rm);
#pragma unroll
        for (int Ig = 0; Ig < NRG; ++Ig)
#pragma unroll
          for (int jj = 0; jj < NJ; ++jj) img[(16 * Ig + rowC) * LD + col0 + 4 * jj + p] = Rm[Ig][jj];
      } else {
        assemble_tiles(tt, Xs);
#pragma unroll
        for (int Ig = 0; Ig < NRG; ++Ig)
#pragma unroll
          for (int jj = 0; jj < NJ; ++jj) {
            Rm[Ig][jj] = Xs[Ig][jj];
            img[(16 * Ig + rowC) * LD + col0 + 4 * jj + p] = Xs[Ig][jj];
          }
      }
    };

    int t = 0;
    bool first = true;
    int ucur = S_U0;
    stage_signals(0);
    __syncthreads();
    assemble(0);
    zero_acc();
    int op = OP_P1, sr = S_M0, si = -1, sd = S_M1, sq_left = 0;
    __syncthreads();
#ifdef C3P_REGR_TIMING
    tq = clock64();
#endif
    for (;;) {
      refresh();
      product(sr);
      RR_TICK(1)
      rr_bar();  // A: everyone is done with the image; border partials are visible
      RR_TICK(2)
      refresh();
      finalize(sr, si, sd);
      const int op_in = op;
      bool next_slice = false;
      if (op == OP_P1) {  // C = A2: kept, and the right operand of the next product (the image still holds X)
#pragma unroll
        for (int Ig = 0; Ig < NRG; ++Ig)
#pragma unroll
          for (int jj = 0; jj < NJ; ++jj) A2s[Ig][jj] = acc[Ig][jj];
        R_from_C();
        zero_acc();
        op = OP_P2, sr = S_M1, si = -1, sd = S_M2;
      } else if (op == OP_P2) {  // C = A3 = X A2: A3 becomes both operands
        image_from_C();
        if (tid < BS) border_to_image(brd[S_M2 * BS + tid]);
        R_from_C();
        zero_acc();
        op = OP_P3, sr = S_M2, si = -1, sd = S_M3;
      } else if (op == OP_P3) {  // C = A6: the T18 combinations of X, A2, A3 (image), A6 (C)
        if constexpr (LEAN) assemble_tiles(t % RR_CH, Rm);  // X again (the right operand is spent: its registers are free)
#pragma unroll
        for (int Ig = 0; Ig < NRG; ++Ig)
#pragma unroll
          for (int jj = 0; jj < NJ; ++jj) {
            const double dg = (16 * Ig + rowC == col0 + 4 * jj + p) ? 1.0 : 0.0;
            const double x = LEAN ? Rm[Ig][jj] : Xs[Ig][jj], a2 = A2s[Ig][jj], a6 = acc[Ig][jj];
            // A3: the right operand of A6 = A3 A3 still sits in Rm (the lean form re-assembled X there: it reads the image)
            const double a3 = LEAN ? img[(16 * Ig + rowC) * LD + col0 + 4 * jj + p] : Rm[Ig][jj];
            // B1 -> image (left operand of A9 = B1 B5 + B4)
            img[(16 * Ig + rowC) * LD + col0 + 4 * jj + p] = fma(tc[C3P_I_A31], a3, fma(tc[C3P_I_A21], a2, tc[C3P_I_A11] * x));
            // B2, B3 stay in registers (in the places of X and A2)
            const double b2 = fma(tc[C3P_I_B61], a6, fma(tc[C3P_I_B31], a3, fma(tc[C3P_I_B21], a2, tc[C3P_I_B11] * x)));
            if constexpr (LEAN) rr_ubase(arena)[(Ig * NJ + jj) * RR_THREADS + tid] = b2;
            else Xs[Ig][jj] = b2;
            A2s[Ig][jj] = fma(tc[C3P_I_B62], a6, fma(tc[C3P_I_B32], a3, fma(tc[C3P_I_B22], a2, fma(tc[C3P_I_B12], x, tc[C3P_I_B02] * dg))));
            // B5 -> right operand, B4 -> initial value of the accumulators
            Rm[Ig][jj] = fma(tc[C3P_I_B64], a6, fma(tc[C3P_I_B34], a3, tc[C3P_I_B24] * a2));
            acc[Ig][jj] = fma(tc[C3P_I_B63], a6, fma(tc[C3P_I_B33], a3, fma(tc[C3P_I_B23], a2, fma(tc[C3P_I_B13], x, tc[C3P_I_B03] * dg))));
          }
        if (tid < BS) {
          const double dg = (tid == DM - 1 || tid == BS - 1) ? 1.0 : 0.0;
          const double x = brd[S_M0 * BS + tid], a2 = brd[S_M1 * BS + tid], a3 = brd[S_M2 * BS + tid], a6 = brd[S_M3 * BS + tid];
          border_to_image(fma(tc[C3P_I_A31], a3, fma(tc[C3P_I_A21], a2, tc[C3P_I_A11] * x)));
          brd[S_M3 * BS + tid] = fma(tc[C3P_I_B61], a6, fma(tc[C3P_I_B31], a3, fma(tc[C3P_I_B21], a2, tc[C3P_I_B11] * x)));                            // B2
          brd[S_M0 * BS + tid] = fma(tc[C3P_I_B62], a6, fma(tc[C3P_I_B32], a3, fma(tc[C3P_I_B22], a2, fma(tc[C3P_I_B12], x, tc[C3P_I_B02] * dg))));       // B3
          brd[S_M1 * BS + tid] = fma(tc[C3P_I_B64], a6, fma(tc[C3P_I_B34], a3, tc[C3P_I_B24] * a2));                                                  // B5
          brd[S_M2 * BS + tid] = fma(tc[C3P_I_B63], a6, fma(tc[C3P_I_B33], a3, fma(tc[C3P_I_B23], a2, fma(tc[C3P_I_B13], x, tc[C3P_I_B03] * dg))));       // B4
        }
        op = OP_P4, sr = S_M1, si = S_M2, sd = S_R0;
      } else if (op == OP_P4) {  // C = A9: left operand B3 + A9, right operand A9, initial value B2
#pragma unroll
        for (int Ig = 0; Ig < NRG; ++Ig)
#pragma unroll
          for (int jj = 0; jj < NJ; ++jj) img[(16 * Ig + rowC) * LD + col0 + 4 * jj + p] = A2s[Ig][jj] + acc[Ig][jj];
        if (tid < BS) border_to_image(brd[S_M0 * BS + tid] + brd[S_R0 * BS + tid]);
        R_from_C();
#pragma unroll
        for (int Ig = 0; Ig < NRG; ++Ig)
#pragma unroll
          for (int jj = 0; jj < NJ; ++jj) {
            if constexpr (!LEAN) acc[Ig][jj] = Xs[Ig][jj];
          }
        if constexpr (LEAN) unpark(0, acc);
        op = OP_EX, sr = S_R0, si = S_M3, sd = S_R1, sq_left = ps;
      } else if (op == OP_EX) {  // C = T18 or one of its squarings
        if (sq_left > 0) {
          image_from_C();
          if (tid < BS) border_to_image(brd[sd * BS + tid]);
          R_from_C();
          zero_acc();
          sr = sd, sd = sd ^ 1, si = -1;
          --sq_left;
        } else {  // C = exp(X - mu)
          if constexpr (DUS) {
            store_out(reinterpret_cast<double*>(A.dUs_out + ((long)sample * A.N + n0 + t) * mat_c) + mat_c, sd, exp(mu));
          }
          if (first) {
            if constexpr (LEAN) {
              park(1, acc);
            } else if constexpr (ULDS) {
              if (len > 1) park_u(acc);
            } else {
#pragma unroll
              for (int Ig = 0; Ig < NRG; ++Ig)
#pragma unroll
                for (int jj = 0; jj < NJ; ++jj) Us[Ig][jj] = acc[Ig][jj];
            }
            if (tid < BS) brd[S_U0 * BS + tid] = brd[sd * BS + tid];
            ucur = S_U0;
            first = false;
            mus = mu;
            if constexpr (QT) store_qT(A.hb_qT + ((long)sample * A.N + n0 + t) * mat_c, sd, exp(mus));
            next_slice = true;
          } else {  // U <- C U
            image_from_C();
            if (tid < BS) border_to_image(brd[sd * BS + tid]);
            if constexpr (LEAN) {
              unpark(1, Rm);
            } else if constexpr (ULDS) {
              unpark_u(Rm);
            } else {
#pragma unroll
              for (int Ig = 0; Ig < NRG; ++Ig)
#pragma unroll
                for (int jj = 0; jj < NJ; ++jj) Rm[Ig][jj] = Us[Ig][jj];
            }
            zero_acc();
            mus += mu;
            op = OP_CH, sr = ucur, si = -1, sd = ucur ^ 1;
          }
        }
      } else {  // OP_CH: C = the running product
        if constexpr (LEAN) {
          if (t + 1 < len) park(1, acc);  // (the last one is written out from the accumulators)
        } else if constexpr (ULDS) {
          if (t + 1 < len) park_u(acc);
        } else {
#pragma unroll
          for (int Ig = 0; Ig < NRG; ++Ig)
#pragma unroll
            for (int jj = 0; jj < NJ; ++jj) Us[Ig][jj] = acc[Ig][jj];
        }
        ucur ^= 1;
        if constexpr (QT) store_qT(A.hb_qT + ((long)sample * A.N + n0 + t) * mat_c, ucur, exp(mus));
        next_slice = true;
      }
      if (next_slice) {
        ++t;
        if (t == len) break;
        if ((t % RR_CH) == 0) {
          stage_signals(t);
          __syncthreads();
        }
        assemble(t % RR_CH);
        zero_acc();
        op = OP_P1, sr = S_M0, si = -1, sd = S_M1;
      }
      RR_TICK(4 + op_in)
      rr_bar();  // B: image and border slots of this phase are visible
      RR_TICK(3)
    }
#ifdef C3P_REGR_TIMING
    if (blockIdx.x == 0 && tid == 0 && dbg)
      for (int i = 0; i < 12; ++i) dbg[i] = tacc[i];
#endif
    // segment product (basis change and frame-rotation row phases are separate epilogues); the last product sits in
    // Us = acc and its border in slot ucur
    if constexpr (!UOUT) {
#pragma unroll
      for (int Ig = 0; Ig < NRG; ++Ig)
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj) acc[Ig][jj] = Us[Ig][jj];
    }
    store_out(reinterpret_cast<double*>(A.seg_out + chain * mat_c) + mat_c, ucur, exp(mus));
  }
}

template <int NRG, int NW, bool DUS, bool UNROLL, bool LEAN, bool QT = false>
__global__ void __launch_bounds__(64 * NW, LEAN ? 2 : NW / 4) regr_chain_kernel(MidArgs A, double* arena, long long* dbg) {
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  if constexpr (NW == 4) {
    regr_chain_body<NRG, NRG, 4, DUS, UNROLL, LEAN, QT>(A, dbg, arena, wave, 0, wave);
  } else {
    constexpr int NJ0 = (NRG + 1) / 2, NJ1 = NRG / 2;
    if (wave < 4) regr_chain_body<NRG, NJ0, 8, DUS, UNROLL, false>(A, dbg, arena, wave, 0, wave);
    else regr_chain_body<NRG, NJ1, 8, DUS, UNROLL, false>(A, dbg, arena, wave - 4, NJ0, wave);
  }
}

// ---- the change of basis (c3p_hb_row / c3p_hb_col: c3p_regd.h) ---------------------------------------------------------
// Real generator tables in the kernel's layout: G' = Re(T G T^+) with G the Lindblad generator pieces
// L0 = dt (clp - i (H0 (x) I - I (x) H0^T)), Lk = -i dt (Hk (x) I - I (x) Hk^T) (propagation.py:565-582), trace shifted;
// flag = 1 when |Im(T G T^+)| <= 1e-14 max |T G T^+| everywhere (a Hermitian Hamiltonian).
__global__ void __launch_bounds__(256) regr_prep_kernel(RegdPrepArgs P, double* tables, int* tabflag, int transpose) {
  __shared__ double red0[256], red1[256];
  __shared__ double mu_s;
  const int tid = threadIdx.x;
  const int ti = blockIdx.x % (1 + P.K);
  const int sample = blockIdx.x / (1 + P.K);
  const int D = P.Dm, Dh = P.Dh;
  const int DP = c3p_regd_class(D);
  const int NRG = (DP - 1) / 16, NJ = NRG;
  const cplx* h = (ti == 0) ? P.h0 + (long)sample * P.h0_bstride : P.hks + (long)sample * P.hks_bstride + (long)(ti - 1) * Dh * Dh;
  auto gelem = [&](int row, int col) -> cplx {
    const int i = row / Dh, j = row - i * Dh, k = col / Dh, l = col - k * Dh;
    cplx v = (ti == 0) ? P.clp[(long)row * D + col] : cmake(0, 0);
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
    return cscale(v, P.dt);
  };
  // element (a, b) of T G T^+ (transpose: of its transpose -- the backward sweep exponentiates X^T)
  auto helem = [&](int a_, int b_) -> cplx {
    const int a = transpose ? b_ : a_, b = transpose ? a_ : b_;
    int ia[2], ib[2];
    cplx ta[2], tb[2];
    const int na = c3p_hb_row(a, Dh, ia, ta), nb = c3p_hb_row(b, Dh, ib, tb);
    cplx s = cmake(0.0, 0.0);
    for (int x = 0; x < na; ++x)
      for (int y = 0; y < nb; ++y) cfma(s, cmul(ta[x], cconj(tb[y])), gelem(ia[x], ib[y]));
    return s;
  };
  // (round 6) NO trace shift in the real kernels: the trace of the real generator is the dissipator's, and shifting by it makes the
  // shifted chain product grow like e^{|mu| n} while e^{sum mu} underflows (c3p_smalld.hip: build_tables).  The table slot stays.
  if (tid == 0) mu_s = 0.0;
  __syncthreads();
  const double mu = mu_s;
  double cs = 0, mre = 0, mim = 0, csym = 0;
  for (int j = tid; j < D; j += 256) {
    double s = 0, ssym = 0;
    for (int i = 0; i < D; ++i) {
      const cplx v = helem(i, j);
      mre = fmax(mre, fabs(v.x));
      mim = fmax(mim, fabs(v.y));
      s += fabs(i == j ? v.x - mu : v.x);
      // symmetric (non-skew) part of the real generator: what keeps it from being normal with an imaginary spectrum (round 6:
      // the economised T18 parameters are used while its 1-norm over the segment stays below C3P_T18N_MAX_NONNORMAL)
      ssym += fabs(0.5 * (v.x + helem(j, i).x) - (i == j ? mu : 0.0));
    }
    cs = fmax(cs, s);
    csym = fmax(csym, ssym);
  }
  __syncthreads();
  red0[tid] = cs;
  red1[tid] = mre;
  __syncthreads();
  double nrm = 0, gmax = 0;
  if (tid == 0)
    for (int i = 0; i < 256; ++i) {
      nrm = fmax(nrm, red0[i]);
      gmax = fmax(gmax, red1[i]);
    }
  __syncthreads();
  red0[tid] = mim;
  red1[tid] = csym;
  __syncthreads();
  const int TSET = NRG * NJ * 256, BS = 2 * DP;
  const long TAB_D = (long)TSET + BS + 4;
  double* out = tables + ((long)sample * (1 + P.K) + ti) * TAB_D;
  for (int e = tid; e < TSET + BS; e += 256) {
    int row, col;
    if (e < TSET) {
      const int tile = e >> 8, t = e & 255;
      const int w = t >> 6, l = t & 63;
      const int Ig = tile / NJ, jj = tile - Ig * NJ;
      row = 16 * Ig + 4 * ((l >> 2) & 3) + (l >> 4);
      col = 4 * NJ * w + 4 * jj + (l & 3);
    } else {
      const int eb = e - TSET;
      row = eb < DP ? DP - 1 : eb - DP;
      col = eb < DP ? eb : DP - 1;
    }
    double g = 0.0;
    if (row < D && col < D) {
      g = helem(row, col).x;
      if (row == col) g -= mu;
    }
    out[e] = g;
  }
  if (tid == 0) {
    double gim = 0, nsym = 0;
    for (int i = 0; i < 256; ++i) gim = fmax(gim, red0[i]), nsym = fmax(nsym, red1[i]);
    double* m = out + (TSET + BS);
    m[0] = mu;
    m[1] = nrm;
    m[2] = gim;
    m[3] = nsym;  // 1-norm of the symmetric part (trace-shifted)
    tabflag[sample * (1 + P.K) + ti] = (gim <= 1e-14 * gmax) ? 1 : 0;
  }
}

// In place: the REAL Dm x Dm matrix M' in the second half of every complex slot -> the complex matrix T^+ M' T in the slot.
// One workgroup per matrix; matrices of samples the complex kernel took are left alone.
__global__ void __launch_bounds__(256) hb_to_complex_kernel(cplx* mats, int mats_per_sample, const int* tabflag, int tab_per_sample,
                                                            int K, int Dh) {
  const int Dm = Dh * Dh;
  const long m = blockIdx.x;
  const int sample = (int)(m / mats_per_sample);
  if (!c3p_hb_sample_is_real(tabflag, tab_per_sample ? sample : 0, K)) return;
  cplx* slot = mats + m * (long)Dm * Dm;
  const double* src = reinterpret_cast<const double*>(slot) + (long)Dm * Dm;
  double* sm = c3p_rr_lds;
  for (int e = threadIdx.x; e < Dm * Dm; e += 256) sm[e] = src[e];
  __syncthreads();
  for (int e = threadIdx.x; e < Dm * Dm; e += 256) {
    const int al = e / Dm, be = e - al * Dm;
    int ia[2], ib[2];
    cplx ta[2], tb[2];
    const int na = c3p_hb_col(al, Dh, ia, ta), nb = c3p_hb_col(be, Dh, ib, tb);
    cplx s = cmake(0.0, 0.0);
    for (int x = 0; x < na; ++x)
      for (int y = 0; y < nb; ++y) {
        const cplx c = cmul(cconj(ta[x]), tb[y]);
        const double v = sm[ia[x] * Dm + ib[y]];
        s.x = fma(c.x, v, s.x);
        s.y = fma(c.y, v, s.y);
      }
    slot[e] = s;
  }
}

template <int NRG, int NW, bool LEAN>
hipError_t launch_rr(const MidArgs& A, void* arena, hipStream_t st) {
  const size_t lds = ((size_t)RR<NRG>::LDS_D + ((C3P_REGR_ULDS != 0) && !LEAN && NW == 4 ? (size_t)RR<NRG>::TSET : 0)) * sizeof(double);
  const long nchains = (long)A.B * A.S;
  const long maxg = LEAN ? 2 * C3P_REGD_MAX_WGS : C3P_REGD_MAX_WGS;
  const unsigned grid = (unsigned)(nchains < maxg ? nchains : maxg);
  auto go = [&](auto kern) -> hipError_t {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    long long* dbg = nullptr;
#ifdef C3P_REGR_TIMING
    static long long* dbg_dev = nullptr;
    if (!dbg_dev) (void)hipMalloc(&dbg_dev, 12 * sizeof(long long));
    dbg = dbg_dev;
#endif
    C3P_LAUNCH(kern, dim3(grid), dim3(64 * NW), lds, st, A, reinterpret_cast<double*>(arena), dbg);
#ifdef C3P_REGR_TIMING
    {
      long long h[12];
      (void)hipStreamSynchronize(st);
      (void)hipMemcpy(h, dbg, sizeof h, hipMemcpyDeviceToHost);
      fprintf(stderr, "[regr timing, cycles of wave 0 / block 0, all its chains] mfma %lld  borders %lld  barA %lld  barB %lld | post P1 %lld P2 %lld P3 %lld P4 %lld EX %lld CH %lld | product entry %lld\n",
              h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7], h[8], h[9], h[10]);
    }
#endif
    return hipGetLastError();
  };
#ifdef C3P_REGR_VARIANTS
  const bool unroll = c3p_opt(C3P_OPT_regr_rolled) <= 0;
  if (A.dUs_out) return unroll ? go(regr_chain_kernel<NRG, NW, true, true, LEAN>) : go(regr_chain_kernel<NRG, NW, true, false, LEAN>);
  return unroll ? go(regr_chain_kernel<NRG, NW, false, true, LEAN>) : go(regr_chain_kernel<NRG, NW, false, false, LEAN>);
#else
  if constexpr (NW == 4 && !LEAN) {
    if (A.hb_qT) return A.dUs_out ? hipErrorInvalidValue : go(regr_chain_kernel<NRG, 4, false, true, false, true>);
  }
  if (A.dUs_out) return go(regr_chain_kernel<NRG, NW, true, true, LEAN>);
  return go(regr_chain_kernel<NRG, NW, false, true, LEAN>);
#endif
}

}  // namespace

// Lindblad superoperators whose dimension D^2 falls into a class of the register-resident kernels: D = 7 (49), 8 (64 in the
// 65 class), 9 (81)
bool c3p_regr_supported(int Dh, int Dm) { return Dm == Dh * Dh && (Dh == 7 || Dh == 8 || Dh == 9) && c3p_regd_supported(Dm); }

size_t c3p_regr_table_doubles(int Dm, int K) {
  const int DP = c3p_regd_class(Dm);
  const int n = (DP - 1) / 16;
  return (size_t)(1 + K) * ((size_t)n * n * 256 + 2 * DP + 4);
}

hipError_t c3p_launch_regr_prep_t(const RegdPrepArgs& P, int nsamp, double* tables, int* tabflag, int transpose, hipStream_t st) {
  C3P_LAUNCH(regr_prep_kernel, dim3((unsigned)(nsamp * (1 + P.K))), dim3(256), 0, st, P, tables, tabflag, transpose);
  return hipGetLastError();
}
hipError_t c3p_launch_regr_prep(const RegdPrepArgs& P, int nsamp, double* tables, int* tabflag, hipStream_t st) {
  return c3p_launch_regr_prep_t(P, nsamp, tables, tabflag, 0, st);
}

// arena: c3p_regr_arena_bytes() of scratch (the two-workgroups-per-CU form parks two tile sets per workgroup there)
size_t c3p_regr_arena_bytes(int Dm) {
  const int n = (c3p_regd_class(Dm) - 1) / 16;
  return (size_t)2 * C3P_REGD_MAX_WGS * 2 * n * n * 256 * sizeof(double);
}

hipError_t c3p_launch_regr_chain(const MidArgs& A, void* arena, hipStream_t st) {
  if (A.K > RR_KMAX || !A.hb_tabflag || !A.hb_tables) return hipErrorInvalidValue;
  // The regular build holds ONE form: four waves (one per SIMD, 512 registers), the product loop unrolled over the row groups
  // of the right operand.  -DC3P_REGR_VARIANTS (tools/ab_build.sh) adds the forms it was measured against
  // (profiles/r04/regr_variants.txt): regr_waves = 8 (two waves per SIMD, column blocks split), regr_waves = 2 (two lean
  // workgroups per CU, B2 and U parked in a global arena), regr_rolled = 1 (rolled product loop).
  const int cls = c3p_regd_class(A.Dm);
#ifdef C3P_REGR_VARIANTS
  const long mode = c3p_opt(C3P_OPT_regr_waves);
#define C3P_RR_CASE(N)                                        \
  if (mode == 8) return launch_rr<N, 8, false>(A, arena, st); \
  if (mode == 2) return launch_rr<N, 4, true>(A, arena, st);  \
  return launch_rr<N, 4, false>(A, arena, st);
#else
#define C3P_RR_CASE(N) return launch_rr<N, 4, false>(A, arena, st);
#endif
  if (cls == 49) { C3P_RR_CASE(3) }
  if (cls == 65) { C3P_RR_CASE(4) }
  if (cls == 81) { C3P_RR_CASE(5) }
#undef C3P_RR_CASE
  return hipErrorInvalidValue;
}

hipError_t c3p_launch_hb_to_complex(cplx* mats, long nmat, int mats_per_sample, const int* tabflag, int tab_per_sample, int K,
                                    int Dh, hipStream_t st) {
  if (nmat <= 0) return hipSuccess;
  const size_t lds = (size_t)Dh * Dh * Dh * Dh * sizeof(double);
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(hb_to_complex_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  C3P_LAUNCH(hb_to_complex_kernel, dim3((unsigned)nmat), dim3(256), lds, st, mats, mats_per_sample, tabflag, tab_per_sample, K, Dh);
  return hipGetLastError();
}
