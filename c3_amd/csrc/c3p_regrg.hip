// Backward sweep (control gradient) of the Lindblad superoperator chains in the Hermitian basis, REAL arithmetic on the f64
// matrix cores: Dm = D^2 = 49, 64 (in the 65 class), 81 -- the gradient of BASELINE cfg4's path.  The reference tapes
// tf_propagation_lind like every other path (c3/libraries/propagation.py:551-585 under c3/optimizers/optimizer.py:206-216).
//
// In the basis of c3p_regr.hip the chain is a product of real matrices E_n = exp(X_n), X_n = G'_0 + sum_k c_k(n) G'_k.  With the
// time axis cut into S segments, Q_n = E_n ... E_n0 the LOCAL prefix inside a segment (stored transposed by the forward kernel,
// MidArgs.hb_qT) and the left adjoint of a segment's end with the prefix at its start folded in (c3p_launch_regr_scan),
//     Lam_n = (E_{n1-1} ... E_{n+1})^T [R^T U_bar' Pstart^T],        E_bar_n = Lam_n Q_{n-1}^T,        Lam_{n-1} = E_n^T Lam_n,
//     grad[k, n] = < L(X_n)^*[E_bar_n], G'_k > + mu_k <U_bar', U'>,   L(X)^* = L(X^T)   (real matrices),
// nothing is inverted (the slices are not orthogonal: a dissipative chain has no cheap inverse).  Value and Frechet derivative
// of the slice exponential come from ONE forward-mode pair evaluation at A = 2^-s X_n^T in direction dA = 2^-s E_bar_n.
//
// What shapes the evaluation is what a CU can hold: the forward kernel's five tile sets already fill the 512 registers of its
// one wave per SIMD, and the pair evaluation of its T18 polynomial keeps ~13 matrices of 51 KB live.  So the polynomial is a
// Taylor polynomial of degree 2J + 2 in Horner form in A^2, whose left operands are FIXED for the whole slice:
//     H_J = c_2J I + c_2J+1 A + c_2J+2 A2,      H_j = (c_2j I + c_2j+1 A) + A2 H_j+1,
//     dH_j = c_2j+1 dA + dA2 H_j+1 + A2 dH_j+1,  A2 = A A,  dA2 = A dA + dA A,          T = H_0, dT = dH_0,
// then s squarings (T, dT) <- (T T, T dT + dT T).  A2 and dA2 sit in TWO LDS images (106 KB of the 160 KB) for all 3 J Horner
// products, the running pair (H, dH), dA, the right operand and the accumulators are the five register tile sets, A itself is
// re-assembled from the (L2-resident) generator tables wherever a combination needs it, and Lam waits in a per-workgroup
// global tile set (51 KB, written and read once per slice by the same thread).  3 + 3 J + 3 s products for the pair, 2 for
// E_bar and Lam: 29 per slice at cfg4's norm for every degree 8 .. 16 (the degree with the fewest products is chosen per
// segment), against 7 of the forward pass -- and nothing but Q^T (52 KB per slice, read once) comes from HBM.
//
// Product loop, register layout, the (Dm-1)^2 core + one-element border split and the border slots are those of
// c3p_regr.hip (c3p_regr_common.h); the slice is a state machine around ONE copy of the product code.
#include <cstdio>

#include "c3p_common.h"
#include "c3p_kernels.h"
#include "c3p_regd.h"
#include "c3p_regr_common.h"

extern __shared__ __attribute__((aligned(16))) double c3p_rg_lds[];

namespace {

enum { B_Q = 0, B_LAM, B_DA, B_A, B_A2, B_DA2, B_H, B_DH, B_T, B_I, B_NSLOT };
enum { G_EBAR = 0, G_D2A, G_D2B, G_A2, G_HD1, G_HD2, G_HV, G_SQ1, G_SQ2, G_SQ3, G_LAM };

__device__ const double rg_invfact[24] = {1.0,
                                          1.0,
                                          1.0 / 2,
                                          1.0 / 6,
                                          1.0 / 24,
                                          1.0 / 120,
                                          1.0 / 720,
                                          1.0 / 5040,
                                          1.0 / 40320,
                                          1.0 / 362880,
                                          1.0 / 3628800,
                                          1.0 / 39916800,
                                          1.0 / 479001600,
                                          1.0 / 6227020800.0,
                                          1.0 / 87178291200.0,
                                          1.0 / 1307674368000.0,
                                          1.0 / 20922789888000.0,
                                          1.0 / 355687428096000.0,
                                          1.0 / 6402373705728000.0,
                                          1.0 / 121645100408832000.0,
                                          1.0 / 2432902008176640000.0,
                                          1.0 / 51090942171709440000.0,
                                          1.0 / 1124000727777607680000.0,
                                          1.0 / 25852016738884976640000.0};

template <int NRG>
struct RG {
  using G = RR<NRG>;
  static constexpr int LDS_D = 2 * G::IMG_D + B_NSLOT * G::BS + 8 * G::DMP + RR_KMAX * RR_CH + 16 + 4 * RR_KMAX;
};

template <int NRG>
__global__ void __launch_bounds__(256, 1) regr_grad_kernel(RegrGradArgs A) {
  using G = RR<NRG>;
  constexpr int DM = G::DM, LD = G::LD, BS = G::BS, DMP = G::DMP, TSET = G::TSET, NJ = NRG;
  constexpr int THREADS = 256;
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int cg = wave;
  const bool corner_owner = (cg == 0);
  const int tid0 = threadIdx.x;
  int tid = tid0, lane = tid & 63;
  int q = lane >> 4, b = (lane >> 2) & 3, p = lane & 3;
  double* img1 = c3p_rg_lds;
  double* img2 = img1 + G::IMG_D;
  double* brd = img2 + G::IMG_D;
  double* cpart = brd + B_NSLOT * BS;
  double* rpart = cpart + 4 * DMP;
  double* sg = rpart + 4 * DMP;
  double* red = sg + RR_KMAX * RR_CH;
  double* gred = red + 16;
  const int col0 = 4 * NRG * cg;
  int rowC = 4 * b + q;
  int rowA = 4 * b + p;
  const int K = A.K;
  const int DR = A.Dm;
  // lane indices re-derived from an opaque copy of the thread id at the start of every phase (c3p_regr.hip: otherwise every
  // address that depends on them is hoisted out of the slice loop and spilled)
  auto refresh = [&]() __attribute__((always_inline)) {
    int t_ = tid0;
    asm volatile("" : "+v"(t_));
    tid = t_;
    lane = tid & 63;
    q = lane >> 4, b = (lane >> 2) & 3, p = lane & 3;
    rowC = 4 * b + q;
    rowA = 4 * b + p;
  };

  double Rm[NRG][NJ];   // right operand of the next product
  double acc[NRG][NJ];  // accumulators = the product
  double Hs[NRG][NJ];   // H_j, then T
  double dHs[NRG][NJ];  // dA2, then dH_j, then dT
  double Ds[NRG][NJ];   // dA
  double* arena = A.arena + (long)blockIdx.x * TSET;

  auto mfma = [](double a, double bb, double c) __attribute__((always_inline)) -> double { return __builtin_amdgcn_mfma_f64_4x4x4f64(a, bb, c, 0, 0, 0); };
  auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int Ig = 0; Ig < NRG; ++Ig)
#pragma unroll
      for (int jj = 0; jj < NJ; ++jj) acc[Ig][jj] = 0.0;
  };
  auto copy_set = [&](double (&dst)[NRG][NJ], const double (&src)[NRG][NJ]) __attribute__((always_inline)) {
#pragma unroll
    for (int Ig = 0; Ig < NRG; ++Ig)
#pragma unroll
      for (int jj = 0; jj < NJ; ++jj) dst[Ig][jj] = src[Ig][jj];
  };
  // border element tid of a matrix also sits in its image: row DM-1 or column DM-1 (always written by its owner thread)
  auto border_to_image = [&](double* img, double v) __attribute__((always_inline)) {
    if (tid < DM) img[(DM - 1) * LD + tid] = v;
    else if (tid < 2 * DM - 1) img[(tid - DM) * LD + DM - 1] = v;
  };
  auto image_from = [&](double* img, const double (&v)[NRG][NJ], int slot) __attribute__((always_inline)) {
#pragma unroll
    for (int Ig = 0; Ig < NRG; ++Ig)
#pragma unroll
      for (int jj = 0; jj < NJ; ++jj) img[(16 * Ig + rowC) * LD + col0 + 4 * jj + p] = v[Ig][jj];
    if (tid < BS) border_to_image(img, brd[slot * BS + tid]);
  };

  // acc += L R: L = an LDS image, R = Rm with its border in slot sr (c3p_regr.hip: product).  Column DM-1 of the product is
  // one more B column, row DM-1 uses the tiles of R block by block; partial sums through cpart / rpart, summed by finalize.
  double cornerA = 0.0, cornerR = 0.0;
  auto product = [&](const double* img, int sr) __attribute__((always_inline)) {
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
      rr_static_for<NRG>([&](auto it_) {
        constexpr int IT = decltype(it_)::value;
        constexpr int R0 = IT;
        constexpr int R1 = (IT + 1 < NRG ? IT + 1 : IT);
        constexpr int OFF = 16 * IT;
        rr_static_for<4>([&](auto s_) {
          constexpr int sx = decltype(s_)::value;
          constexpr int sn = (sx + 1) & 3, rn = (sx == 3) ? R1 : R0;  // next step: rotation, row of Rm
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
#ifdef C3P_REGR_BB
          if (rr_opq(0) != 0) asm volatile("s_sleep 1");  // a basic-block boundary per K-step (rounds 2 - 4; see c3p_regr.hip)
#endif
        });
        {
          const double va = pr[OFF];
          const double ar = (p == 0) ? va : 0.0;
#pragma unroll
          for (int jj = 0; jj < NJ; ++jj) rP[jj] = mfma(ar, Rm[R0][jj], rP[jj]);
          if (corner_owner) {
            const double vb = rb[DM + OFF + rowC];
            const double cb = (p == 0) ? vb : 0.0;
            rP[NJ] = mfma(ar, cb, rP[NJ]);
          }
        }
      });
    }
    {
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
    if (q == 0) {
#pragma unroll
      for (int jj = 0; jj < NJ; ++jj) rpart[b * DMP + col0 + 4 * jj + p] = rP[jj];
      if (corner_owner && p == 0) rpart[b * DMP + DM - 1] = rP[NJ];
    }
    if (p == 0) {
#pragma unroll
      for (int Ig = 0; Ig < NRG; ++Ig) cpart[cg * DMP + 16 * Ig + rowC] = cP[Ig];
    }
    cornerA = img[(DM - 1) * LD + DM - 1];
    cornerR = rb[BS - 1];
    {  // k = DM-1: rank-1 update of the core
      double a80[NRG], b80[NJ];
#pragma unroll
      for (int Ig = 0; Ig < NRG; ++Ig) a80[Ig] = img[(16 * Ig + rowC) * LD + DM - 1];
#pragma unroll
      for (int jj = 0; jj < NJ; ++jj) b80[jj] = rb[col0 + 4 * jj + p];
#pragma unroll
      for (int Ig = 0; Ig < NRG; ++Ig)
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj) acc[Ig][jj] = fma(a80[Ig], b80[jj], acc[Ig][jj]);
    }
  };
  // after the barrier: the border of the product (+ the border si of the accumulators' initial value), one element per lane
  auto finalize = [&](const double* img, int sr, int si, int sd) __attribute__((always_inline)) {
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

  const long nchains = (long)A.B * A.S;
  const long msz = (long)DR * DR;
  for (long chain = blockIdx.x; chain < nchains; chain += gridDim.x) {
    const int sample = (int)(chain / A.S);
    const int seg = (int)(chain - (long)sample * A.S);
    const int n0 = (int)(((long)seg * A.N) / A.S);
    const int n1 = (int)(((long)(seg + 1) * A.N) / A.S);
    const int len = n1 - n0;
    const long tab_off = (long)(A.tab_per_sample ? sample : 0) * (1 + K) * G::TAB_D;
    const double* tabs_t = A.tables_t + tab_off;  // G'^T: X_n^T
    const double* tabs_f = A.tables + tab_off;    // G': the inner products
    auto meta_t = [&](int k1) __attribute__((always_inline)) -> const double* { return tabs_t + (long)k1 * G::TAB_D + (TSET + BS); };
    auto meta_f = [&](int k1) __attribute__((always_inline)) -> const double* { return tabs_f + (long)k1 * G::TAB_D + (TSET + BS); };
    __syncthreads();  // the previous chain is done with the LDS
    // plan: degree and squarings from ||G0^T||_1 + sum_k max_t |c_k(t)| ||G_k^T||_1 over the segment
    double nrm = meta_t(0)[1];
    for (int k = 0; k < K; ++k) {
      const double* s = A.signals + ((long)sample * K + k) * A.N + n0;
      double cmax = 0.0;
      for (int t = tid; t < len; t += THREADS) cmax = fmax(cmax, fabs(s[t]));
      for (int o = 32; o >= 1; o >>= 1) cmax = fmax(cmax, __shfl_xor(cmax, o));
      if (lane == 0) red[wave] = cmax;
      __syncthreads();
      cmax = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
      __syncthreads();
      nrm = fma(cmax, meta_t(k + 1)[1], nrm);
    }
    nrm = rr_rfl(nrm);
    int pJ = 3, ps = 0;
    {
      // Taylor degree 2J + 2 = 8, 12, 16, 20: backward-error bounds for unit roundoff 2^-53 (Al-Mohy & Higham; the thresholds of
      // plan_q4 in c3p_smalld.hip); 3 J + 3 s products
      const double th[4] = {5.45e-2, 3.18e-1, 8.16e-1, 1.49};
      int best = 1 << 30;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int Ji = 3 + 2 * i;
        if (A.degree != 0 && A.degree != 2 * Ji + 2) continue;
        int si = 0;
        double pth = th[i];
        while (pth < nrm && si < 40) {
          pth *= 2.0;
          ++si;
        }
        const int cost = 3 * Ji + 3 * si;
        if (cost < best) best = cost, pJ = Ji, ps = si;
      }
    }
    const int J = __builtin_amdgcn_readfirstlane(pJ);
    const int nsq = __builtin_amdgcn_readfirstlane(ps);
    const double scale = ldexp(1.0, -nsq);
    const double tau = A.tau[sample];

    double mu = 0.0;
    auto stage_signals = [&](int chunk) __attribute__((always_inline)) {  // slices [chunk RR_CH, (chunk + 1) RR_CH) of the segment
      for (int e = tid; e < K * RR_CH; e += THREADS) {
        const int k = e / RR_CH, tt = e - k * RR_CH;
        const int tg = chunk * RR_CH + tt;
        sg[e] = (tg < len) ? A.signals[((long)sample * K + k) * A.N + n0 + tg] : 0.0;
      }
    };
    // dst = f * A tiles, A = 2^-s (G0 + sum_k c_k G_k)^T for slice tt of the staged chunk (tables: L2 resident)
    auto assemble_tiles = [&](int tt, double f, double (&dst)[NRG][NJ]) __attribute__((always_inline)) {
      for (int k1 = 0; k1 <= K; ++k1) {
        const double* src = rr_ubase(tabs_t + (long)k1 * G::TAB_D);
        double v[NRG][NJ];
#pragma unroll
        for (int Ig = 0; Ig < NRG; ++Ig)
#pragma unroll
          for (int jj = 0; jj < NJ; ++jj) v[Ig][jj] = src[(Ig * NRG + jj) * 256 + cg * 64 + lane];
        __builtin_amdgcn_sched_barrier(0);
        const double w = f * (k1 ? scale * sg[(k1 - 1) * RR_CH + tt] : scale);
#pragma unroll
        for (int Ig = 0; Ig < NRG; ++Ig)
#pragma unroll
          for (int jj = 0; jj < NJ; ++jj) dst[Ig][jj] = k1 ? fma(w, v[Ig][jj], dst[Ig][jj]) : w * v[Ig][jj];
      }
    };
    // border of A -> slot B_A; trace shift of the slice -> mu
    auto assemble_border = [&](int tt) __attribute__((always_inline)) {
      double bv = 0.0;
      if (K <= 2) {
        // (K <= 2, cfg4) trace shifts and border elements of the three tables in ONE round trip to L2 instead of one per line,
        // as in the forward kernel (c3p_regr.hip)
        const int btid = tid < BS ? tid : 0;
        double mk[3], bk[3];
#pragma unroll
        for (int k1 = 0; k1 < 3; ++k1) {
          const int ks = k1 <= K ? k1 : 0;
          mk[k1] = meta_t(ks)[0];
          bk[k1] = tabs_t[(long)ks * G::TAB_D + TSET + btid];
        }
        const double c1 = K >= 1 ? sg[0 * RR_CH + tt] : 0.0, c2 = K >= 2 ? sg[1 * RR_CH + tt] : 0.0;
        mu = fma(c2, mk[2], fma(c1, mk[1], mk[0]));
        bv = scale * fma(c2, bk[2], fma(c1, bk[1], bk[0]));
      } else {
        mu = meta_t(0)[0];
        if (tid < BS) bv = scale * tabs_t[TSET + tid];
        for (int k = 0; k < K; ++k) {
          const double c = sg[k * RR_CH + tt];
          mu = fma(c, meta_t(k + 1)[0], mu);
          if (tid < BS) bv = fma(scale * c, tabs_t[(long)(k + 1) * G::TAB_D + TSET + tid], bv);
        }
      }
      if (tid < BS) brd[B_A * BS + tid] = bv;
    };
    const bool bdiag = (tid == DM - 1 || tid == BS - 1);  // the corner entries of a border slot

    // ---- chain start: Lam of the segment's end -> accumulators + slot B_LAM
    {
      const double* lm = rr_ubase(A.lam + chain * msz);
#pragma unroll
      for (int Ig = 0; Ig < NRG; ++Ig)
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj) {
          const int row = 16 * Ig + rowC, col = col0 + 4 * jj + p;
          acc[Ig][jj] = (row < DR && col < DR) ? lm[(long)row * DR + col] : 0.0;
        }
      if (tid < BS) {
        const int row = tid < DM ? DM - 1 : tid - DM, col = tid < DM ? (tid == BS - 1 ? DM - 1 : tid) : DM - 1;
        const int r2 = tid == BS - 1 ? DM - 1 : row, c2 = tid == BS - 1 ? DM - 1 : col;
        brd[B_LAM * BS + tid] = (r2 < DR && c2 < DR) ? lm[(long)r2 * DR + c2] : 0.0;
      }
    }
    int t = len - 1;
    stage_signals(t / RR_CH);
    __syncthreads();

    int op = G_EBAR, sr = B_Q, si = -1, sd = B_DA, j = 0, sq_left = 0;
    const double* pimg = img1;
    double emu = 1.0;
    // E_bar (in acc, border in B_DA) is complete: dA, then A -> image 1 and the right operand dA for A dA
    auto post_ebar = [&]() __attribute__((always_inline)) {
#pragma unroll
      for (int Ig = 0; Ig < NRG; ++Ig)
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj) Ds[Ig][jj] = scale * acc[Ig][jj];
      if (tid < BS) brd[B_DA * BS + tid] *= scale;
      const int tt = t % RR_CH;
      assemble_border(tt);
      assemble_tiles(tt, 1.0, Rm);
      image_from(img1, Rm, B_A);
      copy_set(Rm, Ds);
      zero_acc();
      op = G_D2A, pimg = img1, sr = B_DA, si = -1, sd = B_T;
    };
    // Lam (in acc, border in B_LAM) of slice t: image 1, the arena, Q_{t-1}^T as the right operand
    auto begin_slice = [&]() __attribute__((always_inline)) {
      image_from(img1, acc, B_LAM);
      {
        double* dst = rr_ubase(arena);
#pragma unroll
        for (int Ig = 0; Ig < NRG; ++Ig)
#pragma unroll
          for (int jj = 0; jj < NJ; ++jj) dst[(Ig * NJ + jj) * THREADS + tid] = acc[Ig][jj];
      }
      if (t > 0) {
        const double* qt = rr_ubase(A.qT + ((long)sample * A.N + n0 + t - 1) * msz);
#pragma unroll
        for (int Ig = 0; Ig < NRG; ++Ig)
#pragma unroll
          for (int jj = 0; jj < NJ; ++jj) {
            const int row = 16 * Ig + rowC, col = col0 + 4 * jj + p;
            Rm[Ig][jj] = (row < DR && col < DR) ? qt[(long)row * DR + col] : 0.0;
          }
        if (tid < BS) {
          const int r2 = (tid < DM || tid == BS - 1) ? DM - 1 : tid - DM;
          const int c2 = tid == BS - 1 ? DM - 1 : (tid < DM ? tid : DM - 1);
          brd[B_Q * BS + tid] = (r2 < DR && c2 < DR) ? qt[(long)r2 * DR + c2] : 0.0;
        }
        zero_acc();
        op = G_EBAR, pimg = img1, sr = B_Q, si = -1, sd = B_DA;
      } else {  // the first slice of the segment: Q = 1, E_bar = Lam
        if (tid < BS) brd[B_DA * BS + tid] = brd[B_LAM * BS + tid];
        post_ebar();
      }
    };
    // right operand dH_j+1 and initial value c_2j+1 dA of dH_j
    auto setup_hd1 = [&]() __attribute__((always_inline)) {
      const double a1 = rg_invfact[2 * j + 1];
      copy_set(Rm, dHs);
#pragma unroll
      for (int Ig = 0; Ig < NRG; ++Ig)
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj) acc[Ig][jj] = a1 * Ds[Ig][jj];
      if (tid < BS) brd[B_I * BS + tid] = a1 * brd[B_DA * BS + tid];
      op = G_HD1, pimg = img1, sr = B_DH, si = B_I, sd = B_T;
    };
    auto setup_sq = [&]() __attribute__((always_inline)) {
      image_from(img1, Hs, B_H);
      image_from(img2, dHs, B_DH);
      copy_set(Rm, dHs);
      zero_acc();
      op = G_SQ1, pimg = img1, sr = B_DH, si = -1, sd = B_T;
    };
    // (T, dT) complete: the lane's share of <dT, G_k>, then Lam <- T Lam
    auto setup_fin = [&]() __attribute__((always_inline)) {
      for (int k = 0; k < K; ++k) {
        const double* src = rr_ubase(tabs_f + (long)(k + 1) * G::TAB_D);
        double pk = 0.0;
#pragma unroll
        for (int Ig = 0; Ig < NRG; ++Ig)
#pragma unroll
          for (int jj = 0; jj < NJ; ++jj) pk = fma(dHs[Ig][jj], src[(Ig * NRG + jj) * 256 + cg * 64 + lane], pk);
        if (tid < BS - 1) pk = fma(brd[B_DH * BS + tid], src[TSET + tid], pk);
        for (int o = 32; o >= 1; o >>= 1) pk += __shfl_xor(pk, o);
        if (lane == 0) gred[wave * RR_KMAX + k] = pk;
      }
      image_from(img1, Hs, B_H);
      {
        const double* src = rr_ubase(arena);
#pragma unroll
        for (int Ig = 0; Ig < NRG; ++Ig)
#pragma unroll
          for (int jj = 0; jj < NJ; ++jj) Rm[Ig][jj] = src[(Ig * NJ + jj) * THREADS + tid];
      }
      zero_acc();
      emu = exp(mu);
      op = G_LAM, pimg = img1, sr = B_LAM, si = -1, sd = B_LAM;
    };

    begin_slice();
    rr_bar();
    for (;;) {
      refresh();
      product(pimg, sr);
      rr_bar();  // A: everyone is done with the image; border partials are visible
      refresh();
      finalize(pimg, sr, si, sd);
      bool done = false;
      if (op == G_EBAR) {
        post_ebar();
      } else if (op == G_D2A) {  // acc = A dA: dA -> image 2, right operand A, keep accumulating
        image_from(img2, Ds, B_DA);
        assemble_tiles(t % RR_CH, 1.0, Rm);
        op = G_D2B, pimg = img2, sr = B_A, si = B_T, sd = B_DA2;
      } else if (op == G_D2B) {  // acc = dA2; A (still the right operand) times A next
        copy_set(dHs, acc);
        zero_acc();
        op = G_A2, pimg = img1, sr = B_A, si = -1, sd = B_A2;
      } else if (op == G_A2) {  // acc = A2: the fixed images, the top of the Horner scheme
        const double c0 = rg_invfact[2 * J], c1 = rg_invfact[2 * J + 1], c2 = rg_invfact[2 * J + 2];
        image_from(img2, dHs, B_DA2);
        image_from(img1, acc, B_A2);
#pragma unroll
        for (int Ig = 0; Ig < NRG; ++Ig)
#pragma unroll
          for (int jj = 0; jj < NJ; ++jj) {
            const double dg = (16 * Ig + rowC == col0 + 4 * jj + p) ? c0 : 0.0;
            Hs[Ig][jj] = fma(c2, acc[Ig][jj], fma(c1, Rm[Ig][jj], dg));
            dHs[Ig][jj] = fma(c2, dHs[Ig][jj], c1 * Ds[Ig][jj]);
          }
        if (tid < BS) {
          brd[B_H * BS + tid] = fma(c2, brd[B_A2 * BS + tid], fma(c1, brd[B_A * BS + tid], bdiag ? c0 : 0.0));
          brd[B_DH * BS + tid] = fma(c2, brd[B_DA2 * BS + tid], c1 * brd[B_DA * BS + tid]);
        }
        j = J - 1;
        sq_left = nsq;
        setup_hd1();
      } else if (op == G_HD1) {  // acc = c dA + A2 dH_j+1; + dA2 H_j+1
        copy_set(Rm, Hs);
        op = G_HD2, pimg = img2, sr = B_H, si = B_T, sd = B_DH;
      } else if (op == G_HD2) {  // acc = dH_j; H_j = (c_2j + c_2j+1 A) + A2 H_j+1 (H_j+1 is still the right operand)
        copy_set(dHs, acc);
        const double a0 = rg_invfact[2 * j], a1 = rg_invfact[2 * j + 1];
        assemble_tiles(t % RR_CH, a1, acc);
#pragma unroll
        for (int Ig = 0; Ig < NRG; ++Ig)
#pragma unroll
          for (int jj = 0; jj < NJ; ++jj) acc[Ig][jj] += (16 * Ig + rowC == col0 + 4 * jj + p) ? a0 : 0.0;
        if (tid < BS) brd[B_I * BS + tid] = fma(a1, brd[B_A * BS + tid], bdiag ? a0 : 0.0);
        op = G_HV, pimg = img1, sr = B_H, si = B_I, sd = B_H;
      } else if (op == G_HV) {  // acc = H_j
        copy_set(Hs, acc);
        if (j > 0) {
          --j;
          setup_hd1();
        } else if (sq_left > 0) {
          setup_sq();
        } else {
          setup_fin();
        }
      } else if (op == G_SQ1) {  // acc = T dT; + dT T
        copy_set(Rm, Hs);
        op = G_SQ2, pimg = img2, sr = B_H, si = B_T, sd = B_DH;
      } else if (op == G_SQ2) {  // acc = the new dT; T T next (T is still the right operand)
        copy_set(dHs, acc);
        zero_acc();
        op = G_SQ3, pimg = img1, sr = B_H, si = -1, sd = B_H;
      } else if (op == G_SQ3) {
        copy_set(Hs, acc);
        --sq_left;
        if (sq_left > 0) setup_sq();
        else setup_fin();
      } else {  // G_LAM: acc = T Lam = e^-mu E_n^T Lam_n
        if (tid < K) {
          const double s4 = (gred[tid] + gred[RR_KMAX + tid]) + (gred[2 * RR_KMAX + tid] + gred[3 * RR_KMAX + tid]);
          A.grad[((long)sample * K + tid) * A.N + n0 + t] = fma(emu, s4, meta_f(tid + 1)[0] * tau);
        }
#pragma unroll
        for (int Ig = 0; Ig < NRG; ++Ig)
#pragma unroll
          for (int jj = 0; jj < NJ; ++jj) acc[Ig][jj] *= emu;
        if (tid < BS) brd[B_LAM * BS + tid] *= emu;
        --t;
        if (t < 0) {
          done = true;
        } else {
          if ((t % RR_CH) == RR_CH - 1) stage_signals(t / RR_CH);
          begin_slice();
        }
      }
      if (done) break;
      rr_bar();  // B: images and border slots of this phase are visible
    }
  }
}

// ---- cotangent in the Hermitian basis, segment scan ------------------------------------------------------------------------
// out[b] = Re(T diag(e^{-i phi_b}) U_bar[b] T^+): real Dm x Dm row-major.  One workgroup per sample.
__global__ void __launch_bounds__(256) hb_ubar_kernel(const cplx* Ubar, const double* fr_phase, int Dh, double* out) {
  const int Dm = Dh * Dh;
  const long bidx = blockIdx.x;
  const cplx* ub = Ubar + bidx * (long)Dm * Dm;
  double* o = out + bidx * (long)Dm * Dm;
  for (int e = threadIdx.x; e < Dm * Dm; e += 256) {
    const int a = e / Dm, c = e - a * Dm;
    int ia[2], ic[2];
    cplx ta[2], tc[2];
    const int na = c3p_hb_row(a, Dh, ia, ta), nc = c3p_hb_row(c, Dh, ic, tc);
    double s = 0.0;
    for (int x = 0; x < na; ++x) {
      cplx ph = cmake(1.0, 0.0);
      if (fr_phase) {
        double sn, cs;
        sincos(fr_phase[bidx * Dm + ia[x]], &sn, &cs);
        ph = cmake(cs, -sn);
      }
      for (int y = 0; y < nc; ++y) {
        const cplx w = cmul(cmul(ta[x], cconj(tc[y])), cmul(ph, ub[(long)ia[x] * Dm + ic[y]]));
        s += w.x;
      }
    }
    o[e] = s;
  }
}

// C = op(A) op(B), real n x n row-major in global memory, staged through LDS (sa, sb: n x n each); all 256 threads
template <bool AT, bool BT>
__device__ void rg_mm(double* C, const double* Am, const double* Bm, int n, double* sa, double* sb) {
  const int tid = threadIdx.x;
  __syncthreads();
  for (int e = tid; e < n * n; e += 256) {
    const int i = e / n, k = e - i * n;
    sa[e] = AT ? Am[(long)k * n + i] : Am[e];
    sb[e] = BT ? Bm[(long)k * n + i] : Bm[e];
  }
  __syncthreads();
  for (int e = tid; e < n * n; e += 256) {
    const int i = e / n, jx = e - i * n;
    double s = 0.0;
    for (int k = 0; k < n; ++k) s = fma(sa[i * n + k], sb[k * n + jx], s);
    C[e] = s;
  }
  __syncthreads();
}

// per sample: pre[j] = S_{j-1} ... S_0 (j >= 1), suf[j] = S_{j+1}^T ... S_{S-1}^T U_bar', tau = <suf[0], S_0>
__global__ void __launch_bounds__(256) regr_scan_seq_kernel(const cplx* seg_slots, const double* ubar, int S, int Dm, double* pre, double* suf,
                                                            double* tau) {
  __shared__ double redt[256];
  const long msz = (long)Dm * Dm;
  const long bidx = blockIdx.x;
  double* sa = c3p_rg_lds;
  double* sb = sa + msz;
  auto segm = [&](int jx) -> const double* { return reinterpret_cast<const double*>(seg_slots + (bidx * S + jx) * msz) + msz; };
  double* pb = pre + bidx * S * msz;
  double* sfx = suf + bidx * S * msz;
  const int tid = threadIdx.x;
  if (S > 1) {
    for (long e = tid; e < msz; e += 256) pb[msz + e] = segm(0)[e];
    for (int jx = 2; jx < S; ++jx) rg_mm<false, false>(pb + jx * msz, segm(jx - 1), pb + (jx - 1) * msz, Dm, sa, sb);
  }
  for (long e = tid; e < msz; e += 256) sfx[(S - 1) * msz + e] = ubar[bidx * msz + e];
  for (int jx = S - 1; jx >= 1; --jx) rg_mm<true, false>(sfx + (jx - 1) * msz, segm(jx), sfx + jx * msz, Dm, sa, sb);
  __syncthreads();
  double part = 0.0;
  for (long e = tid; e < msz; e += 256) part = fma(sfx[e], segm(0)[e], part);
  redt[tid] = part;
  __syncthreads();
  if (tid == 0) {
    double s = 0.0;
    for (int i = 0; i < 256; ++i) s += redt[i];
    tau[bidx] = s;
  }
}

// lam[b, j] = suf[b, j] pre[b, j]^T (j = 0: suf[b, 0])
__global__ void __launch_bounds__(256) regr_scan_fold_kernel(const double* pre, const double* suf, int S, int Dm, double* lam) {
  const long msz = (long)Dm * Dm;
  const long m = blockIdx.x;
  const int jx = (int)(m % S);
  if (jx == 0) {
    for (long e = threadIdx.x; e < msz; e += 256) lam[m * msz + e] = suf[m * msz + e];
    return;
  }
  double* sa = c3p_rg_lds;
  rg_mm<false, true>(lam + m * msz, suf + m * msz, pre + m * msz, Dm, sa, sa + msz);
}

template <int NRG>
hipError_t launch_rg(const RegrGradArgs& A, hipStream_t st) {
  const size_t lds = (size_t)RG<NRG>::LDS_D * sizeof(double);
  const long nchains = (long)A.B * A.S;
  const unsigned grid = (unsigned)(nchains < C3P_REGD_MAX_WGS ? nchains : C3P_REGD_MAX_WGS);
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(regr_grad_kernel<NRG>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  C3P_LAUNCH(regr_grad_kernel<NRG>, dim3(grid), dim3(256), lds, st, A);
  return hipGetLastError();
}

}  // namespace

size_t c3p_regr_grad_arena_bytes(int Dm) {
  const int n = (c3p_regd_class(Dm) - 1) / 16;
  return (size_t)C3P_REGD_MAX_WGS * n * n * 256 * sizeof(double);
}

hipError_t c3p_launch_regr_grad(const RegrGradArgs& A, hipStream_t st) {
  if (A.K > RR_KMAX || A.K < 1 || !A.tables || !A.tables_t || !A.qT || !A.lam || !A.tau || !A.arena) return hipErrorInvalidValue;
  if (A.degree != 0 && A.degree != 8 && A.degree != 12 && A.degree != 16 && A.degree != 20) return hipErrorInvalidValue;
  const int cls = c3p_regd_class(A.Dm);
  if (cls == 49) return launch_rg<3>(A, st);
  if (cls == 65) return launch_rg<4>(A, st);
  if (cls == 81) return launch_rg<5>(A, st);
  return hipErrorInvalidValue;
}

hipError_t c3p_launch_hb_ubar(const cplx* Ubar, const double* fr_phase, int B, int Dh, double* out, hipStream_t st) {
  C3P_LAUNCH(hb_ubar_kernel, dim3((unsigned)B), dim3(256), 0, st, Ubar, fr_phase, Dh, out);
  return hipGetLastError();
}

hipError_t c3p_launch_regr_scan(const cplx* seg_slots, const double* ubar, int B, int S, int Dm, double* pre, double* suf, double* lam,
                                double* tau, hipStream_t st) {
  const size_t lds = (size_t)2 * Dm * Dm * sizeof(double);
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(regr_scan_seq_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  e = hipFuncSetAttribute(reinterpret_cast<const void*>(regr_scan_fold_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  C3P_LAUNCH(regr_scan_seq_kernel, dim3((unsigned)B), dim3(256), lds, st, seg_slots, ubar, S, Dm, pre, suf, tau);
  C3P_LAUNCH(regr_scan_fold_kernel, dim3((unsigned)(B * S)), dim3(256), lds, st, pre, suf, S, Dm, lam);
  return hipGetLastError();
}
