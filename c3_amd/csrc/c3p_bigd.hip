// Big-D propagator chains on the f64 matrix cores: 41 <= Dm <= 92, in particular the
// 81 x 81 Lindblad superoperator of two qutrits (BASELINE cfg4).
//
// Same arithmetic as c3p_midd.hip (half images of the real 2x2 representation, 16x4 output
// tiles on v_mfma_f64_4x4x4_4b_f64, T18 + scaling/squaring), but a half image of an 81 x 81
// matrix is 176 x 86 doubles = 121 KB: only ONE fits in LDS.  So
//   * every matrix of a chain lives in a per-workgroup GLOBAL scratch arena (MALL/L2 resident),
//   * for each product the LEFT operand's image is copied global -> LDS once (it is reused by
//     all 41 K-steps of all 231 tiles), and the RIGHT operand's 4x4 blocks are read straight
//     from global memory, prefetched one K-step ahead (16 distinct 8-byte words per wave load),
//   * one workgroup = 8 wavefronts (2 per SIMD) owns a chain; the 231 tiles are dealt
//     29 per wave, compile-time, so a K-step is 11 LDS reads + 3-4 global loads for 29 MFMAs.
// Algorithmic intensity per product: 4.25 Mflop for 363 KB of image traffic = 11.7 flop/B,
// i.e. the kernel sits near the MI355X fp64-compute / HBM ridge; the arena (1 MB per
// workgroup) is sized to stay in the 256 MB Infinity Cache.
#include "c3p_common.h"
#include "c3p_kernels.h"
#include "c3p_midd.h"
#include "c3p_bigd.h"

extern __shared__ __attribute__((aligned(16))) double c3p_bd_lds[];

namespace {

constexpr int BW = 8;  // wavefronts per workgroup

// An opaque copy of a lane offset: stops LLVM from hoisting the (loop-invariant) per-tile 64-bit
// addresses of every image out of the slice loop, which would need hundreds of live registers.
__device__ __forceinline__ int bd_opaque(int v) {
  asm volatile("" : "+v"(v));
  return v;
}

__device__ __forceinline__ double bd_mfma4(double a, double b, double c) {
  return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ double bd_flip(double v, unsigned mask_hi) {
  unsigned long long u = __double_as_longlong(v);
  u ^= ((unsigned long long)mask_hi) << 32;
  return __longlong_as_double(u);
}
__device__ __forceinline__ double bd_rfl(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_readfirstlane(lo);
  hi = __builtin_amdgcn_readfirstlane(hi);
  return __hiloint2double(hi, lo);
}

template <int NIG, int NJ>
struct BD {
  static constexpr int ROWS = 16 * NIG;
  static constexpr int JB = NJ / BW;        // column blocks per wave: the first NJ % BW waves take JB + 1
  static constexpr int JR = NJ % BW;
  static constexpr int JS = JB + (JR ? 1 : 0);  // register tiles per wave: JS x NIG
};

struct BigCommon {
  int tid, lane, r, b, c;
  int D, nbk, K;
  int sample, n0, len;
  int aoff, boff, dbase;
  int J0, nJ;  // this wave's column blocks [J0, J0 + nJ)
  unsigned negmask;
  int ps;
  double scale;
  const double* tabs;
  double* arena;  // this workgroup's global images
  double* ldsA;
  double* sg;
};

// arena image slots
enum { G_X = 0, G_A2, G_A3, G_A6, G_T1, G_T2, G_U, G_NIMG };

// acc[jj][Ig] += (A image in LDS) * (B image in global) for the wave's column blocks.
// All waves run the same code; only J0 / nJ differ (runtime, wave-uniform).
template <int NIG, int NJ, int W>
__device__ __forceinline__ void bd_mm(const BigCommon& cm, const double* gB, double (&acc)[BD<NIG, NJ>::JS][NIG]) {
  constexpr int JS = BD<NIG, NJ>::JS;
  const double* ldsA = cm.ldsA + bd_opaque(cm.aoff);
  // global images are TILE-MAJOR (tile (J,Ig) = 64 contiguous doubles in lane order): element
  // Bh[4K + r][4J + c] sits at ((J NIG + K/4) 64) + 16 r + 4 (K%4) + c
  const double* pB = gB + bd_opaque(16 * cm.r + cm.c + cm.J0 * NIG * 64);
  const bool last = cm.nJ == JS;  // does the wave use its last register column?
  // A fragments come from LDS (prefetched one K-step ahead); B blocks come from global memory
  // (Infinity Cache / HBM latency ~1-2 us), prefetched three K-steps ahead through a 4-deep ring.
  double a0[NIG], a1[NIG], bq[4][JS];
  const int kl = cm.nbk - 1;
  auto load_b = [&](double (&bb)[JS], int K) {
    const int Kc = K < kl ? K : kl;
    const int kb = 64 * (Kc >> 2) + 4 * (Kc & 3);
#pragma unroll
    for (int jj = 0; jj < JS; ++jj) bb[jj] = (jj < JS - 1 || last) ? pB[kb + jj * NIG * 64] : 0.0;
  };
  auto load_a = [&](double (&a)[NIG], int K) {
    const int Kc = K < kl ? K : kl;
#pragma unroll
    for (int Ig = 0; Ig < NIG; ++Ig) a[Ig] = bd_flip(ldsA[Ig * 16 * W + 2 * Kc], cm.negmask);
  };
  auto fmas = [&](const double (&a)[NIG], const double (&bb)[JS]) {
#pragma unroll
    for (int jj = 0; jj < JS - 1; ++jj)
#pragma unroll
      for (int Ig = 0; Ig < NIG; ++Ig) acc[jj][Ig] = bd_mfma4(a[Ig], bb[jj], acc[jj][Ig]);
    if (last) {
#pragma unroll
      for (int Ig = 0; Ig < NIG; ++Ig) acc[JS - 1][Ig] = bd_mfma4(a[Ig], bb[JS - 1], acc[JS - 1][Ig]);
    }
  };
  load_b(bq[0], 0);
  load_b(bq[1], 1);
  load_b(bq[2], 2);
  load_a(a0, 0);
  for (int K = 0; K < cm.nbk; K += 4) {
    load_b(bq[3], K + 3);
    load_a(a1, K + 1);
    __builtin_amdgcn_sched_barrier(0);
    fmas(a0, bq[0]);
    __builtin_amdgcn_sched_barrier(0);
    load_b(bq[0], K + 4);
    load_a(a0, K + 2);
    __builtin_amdgcn_sched_barrier(0);
    if (K + 1 < cm.nbk) fmas(a1, bq[1]);
    __builtin_amdgcn_sched_barrier(0);
    load_b(bq[1], K + 5);
    load_a(a1, K + 3);
    __builtin_amdgcn_sched_barrier(0);
    if (K + 2 < cm.nbk) fmas(a0, bq[2]);
    __builtin_amdgcn_sched_barrier(0);
    load_b(bq[2], K + 6);
    load_a(a0, K + 4);
    __builtin_amdgcn_sched_barrier(0);
    if (K + 3 < cm.nbk) fmas(a1, bq[3]);
    __builtin_amdgcn_sched_barrier(0);
  }
}

template <int NIG, int NJ, int W, bool DUS>
__device__ __forceinline__ void bigd_body(const MidArgs& A, const BigCommon& cm, long chain) {
  using C = BD<NIG, NJ>;
  constexpr int JS = C::JS, IMG = NIG * NJ * 64;  // global images are tile-major
  const int D = cm.D, K = cm.K, r = cm.r;
  const int rrow = 4 * cm.b + cm.r;
  const int tbase = cm.lane + cm.J0 * NIG * 64;  // tile (jj, Ig) element in a tile-major global image: tbase + (jj NIG + Ig) 64
  double mus_r = 0.0, mus_i = 0.0;
  const double* tabs = cm.tabs;
  auto gimg = [&](int slot) -> double* { return cm.arena + (long)slot * IMG; };
  typedef double Tile[JS][NIG];

  auto jj_on = [&](int jj) -> bool { return jj < cm.nJ; };
  auto store_tiles = [&](double* img, const Tile& v) {
    const int tb = bd_opaque(tbase);
#pragma unroll
    for (int jj = 0; jj < JS; ++jj)
      if (jj_on(jj)) {
#pragma unroll
        for (int Ig = 0; Ig < NIG; ++Ig) img[tb + (jj * NIG + Ig) * 64] = v[jj][Ig];
      }
  };
  auto zero = [&](Tile& v) {
#pragma unroll
    for (int jj = 0; jj < JS; ++jj)
#pragma unroll
      for (int Ig = 0; Ig < NIG; ++Ig) v[jj][Ig] = 0.0;
  };
  // workgroup-wide: copy the left operand's image to LDS, then multiply
  auto product = [&](const double* gA, const double* gB, Tile& acc) {
    __syncthreads();  // previous product done with ldsA; global images written by all waves
    // tile-major global image -> row-major (stride W) LDS image; one coalesced 512 B read per tile
    // (15 loads in flight per wave: the global latency is paid twice, not 29 times)
    {
      constexpr int NTT = NIG * NJ, CH = 15;
      const int w0 = cm.tid >> 6;
      const double* src = gA + bd_opaque(cm.lane);
      double* dstl = cm.ldsA + bd_opaque(cm.dbase);
      for (int t0 = w0; t0 < NTT; t0 += BW * CH) {
        double v[CH];
#pragma unroll
        for (int u = 0; u < CH; ++u) {
          const int tile = t0 + u * BW;
          v[u] = tile < NTT ? src[tile * 64] : 0.0;
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < CH; ++u) {
          const int tile = t0 + u * BW;
          if (tile < NTT) {
            const int J = tile / NIG, Ig = tile - J * NIG;
            dstl[16 * Ig * W + 4 * J] = v[u];
          }
        }
      }
    }
    __syncthreads();
    bd_mm<NIG, NJ, W>(cm, gB, acc);
  };
  // The T18 block polynomials are lane-local linear combinations of X, A2, A3, A6 at the lane's
  // tile positions.  Those values live in the global arena (too many for registers), so the
  // combinations are fused into two passes, each loading the four images once per column block
  // (44 independent loads in flight): pass 1 -> B1, B5 (stored), B4, B2 (registers);
  // pass 2 -> L = B3 + A9 (stored).
  auto is_diag = [&](int Ig, int col) -> bool {
    const int row = 16 * Ig + rrow;
    return ((row & 1) == 0) && ((row >> 1) == col) && (col < D);
  };
  constexpr int NH = (NIG + 1) / 2;  // row groups are processed in two halves to bound registers
  auto pass1 = [&](Tile& b4) {       // T1 <- B1, T2 <- B5, b4 <- B4
    const double *gx = gimg(G_X), *g2 = gimg(G_A2), *g3 = gimg(G_A3), *g6 = gimg(G_A6);
    double *t1 = gimg(G_T1), *t2 = gimg(G_T2);
    const int tb = bd_opaque(tbase);
#pragma unroll
    for (int jj = 0; jj < JS; ++jj) {
      if (!jj_on(jj)) continue;
      const int col = 4 * (cm.J0 + jj) + cm.c;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        double vx[NH], v2[NH], v3[NH], v6[NH];
#pragma unroll
        for (int u = 0; u < NH; ++u) {
          const int Ig = h * NH + u;
          if (Ig < NIG) {
            const int o = tb + (jj * NIG + Ig) * 64;
            vx[u] = gx[o];
            v2[u] = g2[o];
            v3[u] = g3[o];
            v6[u] = g6[o];
          }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < NH; ++u) {
          const int Ig = h * NH + u;
          if (Ig < NIG) {
            const int o = tb + (jj * NIG + Ig) * 64;
            const double dg = is_diag(Ig, col) ? 1.0 : 0.0;
            t1[o] = fma(C3P_T18_A31, v3[u], fma(C3P_T18_A21, v2[u], C3P_T18_A11 * vx[u]));  // B1
            t2[o] = fma(C3P_T18_B64, v6[u], fma(C3P_T18_B34, v3[u], C3P_T18_B24 * v2[u]));  // B5
            b4[jj][Ig] = fma(C3P_T18_B63, v6[u], fma(C3P_T18_B33, v3[u], fma(C3P_T18_B23, v2[u],
                             fma(C3P_T18_B13, vx[u], C3P_T18_B03 * dg))));                   // B4
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };
  auto pass2 = [&](const Tile& a9, Tile& b2) {  // T1 <- B3 + A9, T2 <- A9, b2 <- B2
    const double *gx = gimg(G_X), *g2 = gimg(G_A2), *g3 = gimg(G_A3), *g6 = gimg(G_A6);
    double *t1 = gimg(G_T1), *t2 = gimg(G_T2);
    const int tb = bd_opaque(tbase);
#pragma unroll
    for (int jj = 0; jj < JS; ++jj) {
      if (!jj_on(jj)) continue;
      const int col = 4 * (cm.J0 + jj) + cm.c;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        double vx[NH], v2[NH], v3[NH], v6[NH];
#pragma unroll
        for (int u = 0; u < NH; ++u) {
          const int Ig = h * NH + u;
          if (Ig < NIG) {
            const int o = tb + (jj * NIG + Ig) * 64;
            vx[u] = gx[o];
            v2[u] = g2[o];
            v3[u] = g3[o];
            v6[u] = g6[o];
          }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < NH; ++u) {
          const int Ig = h * NH + u;
          if (Ig < NIG) {
            const int o = tb + (jj * NIG + Ig) * 64;
            const double dg = is_diag(Ig, col) ? 1.0 : 0.0;
            const double b3 = fma(C3P_T18_B62, v6[u], fma(C3P_T18_B32, v3[u], fma(C3P_T18_B22, v2[u],
                                  fma(C3P_T18_B12, vx[u], C3P_T18_B02 * dg))));
            t1[o] = b3 + a9[jj][Ig];
            t2[o] = a9[jj][Ig];
            b2[jj][Ig] = fma(C3P_T18_B61, v6[u], fma(C3P_T18_B31, v3[u], fma(C3P_T18_B21, v2[u],
                             C3P_T18_B11 * vx[u])));                                         // B2
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };
  auto store_plain = [&](double* dst, const Tile& v, double sr, double si) {
#pragma unroll
    for (int jj = 0; jj < JS; ++jj) {
      if (!jj_on(jj)) continue;
      const int col = 4 * (cm.J0 + jj) + cm.c;
#pragma unroll
      for (int Ig = 0; Ig < NIG; ++Ig) {
        const int ci = (16 * Ig + rrow) >> 1;
        const double mine = v[jj][Ig];
        const double other = __shfl_xor(mine, 16);
        const double outv = (r & 1) ? fma(sr, mine, si * other) : fma(sr, mine, -si * other);
        if (ci < D && col < D) dst[(ci * D + col) * 2 + (r & 1)] = outv;
      }
    }
  };

  Tile P, acc;
  for (int t = 0; t < cm.len; ++t) {
    // ---- X = scale (G0 + sum_k c_k G_k) ----
    double mu_r = tabs[IMG + 0], mu_i = tabs[IMG + 1];
    {
      zero(acc);
      const int tb = bd_opaque(tbase);
      // control amplitudes of this slice (K <= 16), trace shift
      double ck[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) ck[k] = 0.0;
      for (int k = 0; k < K; ++k) {
        const double c0 = cm.sg[k * A.Lmax + t];
        const double* tk = tabs + (long)(k + 1) * (IMG + 4);
        mu_r = fma(c0, tk[IMG + 0], mu_r);
        mu_i = fma(c0, tk[IMG + 1], mu_i);
      }
#pragma unroll
      for (int jj = 0; jj < JS; ++jj) {
        if (!jj_on(jj)) continue;
        // all (1+K) table loads of a column block are issued back to back
        for (int k = -1; k < K; ++k) {
          const double* tk = tabs + (long)(k + 1) * (IMG + 4);
          const double w = (k < 0) ? cm.scale : cm.scale * cm.sg[k * A.Lmax + t];
#pragma unroll
          for (int Ig = 0; Ig < NIG; ++Ig) acc[jj][Ig] = fma(w, tk[tb + (jj * NIG + Ig) * 64], acc[jj][Ig]);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      (void)ck;
      store_tiles(gimg(G_X), acc);
    }
    // ---- T18: A2 = X X, A3 = X A2, A6 = A3 A3 ----
    zero(acc);
    product(gimg(G_X), gimg(G_X), acc);
    store_tiles(gimg(G_A2), acc);
    zero(acc);
    product(gimg(G_X), gimg(G_A2), acc);
    store_tiles(gimg(G_A3), acc);
    zero(acc);
    product(gimg(G_A3), gimg(G_A3), acc);
    store_tiles(gimg(G_A6), acc);
    // ---- A9 = B1 B5 + B4 ;  P = B2 + (B3 + A9) A9 ----
    pass1(acc);                            // T1 = B1, T2 = B5, acc = B4
    product(gimg(G_T1), gimg(G_T2), acc);  // acc = A9
    __syncthreads();                       // every wave is done reading T1 / T2
    zero(P);
    pass2(acc, P);                         // T1 = B3 + A9, T2 = A9, P = B2
    product(gimg(G_T1), gimg(G_T2), P);
    // ---- squarings ----
    for (int it = 0; it < cm.ps; ++it) {
      __syncthreads();
      store_tiles(gimg(G_T1), P);
      zero(acc);
      product(gimg(G_T1), gimg(G_T1), acc);
#pragma unroll
      for (int jj = 0; jj < JS; ++jj)
#pragma unroll
        for (int Ig = 0; Ig < NIG; ++Ig) P[jj][Ig] = acc[jj][Ig];
    }
    if constexpr (DUS) {
      double sn, cs;
      sincos(mu_i, &sn, &cs);
      const double er = exp(mu_r);
      double* dst = reinterpret_cast<double*>(A.dUs_out) + ((long)cm.sample * A.N + cm.n0 + t) * D * D * 2;
      store_plain(dst, P, er * cs, er * sn);
    }
    // ---- chain: U <- E U ; U lives in the arena ----
    if (t == 0) {
      __syncthreads();
      store_tiles(gimg(G_U), P);
      mus_r = mu_r;
      mus_i = c3p_phase_add(0.0, mu_i);
    } else {
      __syncthreads();
      store_tiles(gimg(G_T1), P);
      zero(acc);
      product(gimg(G_T1), gimg(G_U), acc);
      __syncthreads();  // every wave has read the old U
      store_tiles(gimg(G_U), acc);
      mus_r += mu_r;
      mus_i = c3p_phase_add(mus_i, mu_i);
    }
  }
  // ---- result ----
  __syncthreads();
  zero(P);
#pragma unroll
  for (int jj = 0; jj < JS; ++jj)
    if (jj_on(jj)) {
#pragma unroll
      for (int Ig = 0; Ig < NIG; ++Ig) P[jj][Ig] = gimg(G_U)[bd_opaque(tbase) + (jj * NIG + Ig) * 64];
    }
  double sn, cs;
  sincos(mus_i, &sn, &cs);
  const double er = exp(mus_r);
  double* dst = reinterpret_cast<double*>(A.seg_out) + chain * D * D * 2;
  store_plain(dst, P, er * cs, er * sn);  // row phases: separate epilogue kernel (c3p_launch_rowphase)
}

template <int NIG, int NJ, int W, bool DUS>
__global__ void __launch_bounds__(512, 2) bigd_chain_kernel(MidArgs A, double* arena_base) {
  using C = BD<NIG, NJ>;
  constexpr int IMG = NIG * NJ * 64;      // tile-major global image (tables, arena)
  constexpr int LIMG = C::ROWS * W;       // row-major LDS image of the left operand
  BigCommon cm;
  cm.tid = threadIdx.x;
  cm.lane = cm.tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(cm.tid >> 6);
  cm.r = cm.lane >> 4;
  cm.b = (cm.lane >> 2) & 3;
  cm.c = cm.lane & 3;
  cm.D = A.Dm;
  cm.nbk = (2 * cm.D + 3) / 4;
  cm.K = A.K;
  const int K = A.K;
  cm.ldsA = c3p_bd_lds;
  cm.sg = cm.ldsA + LIMG;
  __shared__ double red[BW];
  // column blocks of this wave
  cm.nJ = C::JB + (wave < C::JR ? 1 : 0);
  cm.J0 = wave * C::JB + (wave < C::JR ? wave : C::JR);

  cm.aoff = (4 * cm.b + (cm.c & ~1) + ((cm.c ^ cm.r) & 1)) * W + (cm.r >> 1);
  cm.boff = cm.r * W + cm.c;
  cm.dbase = (4 * cm.b + cm.r) * W + cm.c;
  cm.negmask = (((cm.c & 1) == 0) && ((cm.r & 1) == 1)) ? 0x80000000u : 0u;

  // one arena per WORKGROUP (re-used for every chain it processes): the resident set is
  // gridDim.x x 7 images, sized by the launcher to stay inside the Infinity Cache
  cm.arena = arena_base + (long)blockIdx.x * G_NIMG * IMG;
  // zero it once: padding rows / columns of every image must stay zero (also the LDS image)
  for (long e = cm.tid; e < (long)G_NIMG * IMG; e += BW * 64) cm.arena[e] = 0.0;
  for (int e = cm.tid; e < LIMG; e += BW * 64) cm.ldsA[e] = 0.0;

  const long nchains = (long)A.B * A.S;
  for (long chain = blockIdx.x; chain < nchains; chain += gridDim.x) {
    cm.sample = (int)(chain / A.S);
    const int seg = (int)(chain - (long)cm.sample * A.S);
    cm.n0 = (int)(((long)seg * A.N) / A.S);
    const int n1 = (int)(((long)(seg + 1) * A.N) / A.S);
    cm.len = n1 - cm.n0;
    cm.tabs = A.tables + (long)(A.tab_per_sample ? cm.sample : 0) * (1 + K) * (IMG + 4);
    __syncthreads();  // previous chain's use of sg / red is over
    double nrm = cm.tabs[IMG + 2];
    for (int k = 0; k < K; ++k) {
      const double* s = A.signals + ((long)cm.sample * K + k) * A.N + cm.n0;
      double cmax = 0.0;
      for (int t = cm.tid; t < cm.len; t += BW * 64) {
        const double v = s[t];
        cm.sg[k * A.Lmax + t] = v;
        cmax = fmax(cmax, fabs(v));
      }
      for (int o = 32; o >= 1; o >>= 1) cmax = fmax(cmax, __shfl_xor(cmax, o));
      if (cm.lane == 0) red[wave] = cmax;
      __syncthreads();
      cmax = 0.0;
      for (int w = 0; w < BW; ++w) cmax = fmax(cmax, red[w]);
      __syncthreads();
      nrm = fma(cmax, cm.tabs[(long)(k + 1) * (IMG + 4) + IMG + 2], nrm);
    }
    nrm = bd_rfl(nrm);
    int s18 = 0;
    {
      double p = C3P_T18_THETA;
      while (p < nrm && s18 < 40) {
        p *= 2.0;
        ++s18;
      }
    }
    cm.ps = __builtin_amdgcn_readfirstlane(s18);
    cm.scale = ldexp(1.0, -cm.ps);
    __syncthreads();
    bigd_body<NIG, NJ, W, DUS>(A, cm, chain);
  }
}

// U[b][i][:] *= exp(i phase[b][i])  (frame-rotation row phases, experiment.py:482-509)
__global__ void rowphase_kernel(cplx* U, const double* phase, int Dm, long total) {
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  const long row = e / Dm;  // = b * Dm + i
  double sn, cs;
  sincos(phase[row], &sn, &cs);
  U[e] = cmul(cmake(cs, sn), U[e]);
}

template <int NIG, int NJ, int W>
hipError_t launch_b(const MidArgs& A, double* arena, hipStream_t st) {
  constexpr int LIMG = BD<NIG, NJ>::ROWS * W;
  const size_t lds = (size_t)(LIMG + A.K * A.Lmax) * sizeof(double);
  const long nchains = (long)A.B * A.S;
  const unsigned grid = (unsigned)(nchains < C3P_BIGD_MAX_WGS ? nchains : C3P_BIGD_MAX_WGS);
  auto go = [&](auto kern) -> hipError_t {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    C3P_LAUNCH(kern, dim3(grid), dim3(BW * 64), lds, st, A, arena);
    return hipGetLastError();
  };
  if (A.dUs_out) return go(bigd_chain_kernel<NIG, NJ, W, true>);
  return go(bigd_chain_kernel<NIG, NJ, W, false>);
}

}  // namespace

bool c3p_bigd_geometry(int Dm, int* nig, int* nj, int* w) {
  if (Dm < 41 || Dm > 92) return false;
  const int NBI = (Dm + 1) / 2;
  *nig = (NBI + 3) / 4;
  *nj = (Dm + 3) / 4;
  *w = 4 * (*nj) + 2;  // = 2 mod 4: the 16 rows of an A-slab read fall on distinct LDS bank pairs
  return true;
}

size_t c3p_bigd_table_doubles(int Dm, int K) {
  int nig, nj, w;
  if (!c3p_bigd_geometry(Dm, &nig, &nj, &w)) return 0;
  return (size_t)(1 + K) * ((size_t)nig * nj * 64 + 4);
}

size_t c3p_bigd_arena_doubles(int Dm) {  // for the whole launch (one arena per workgroup)
  int nig, nj, w;
  if (!c3p_bigd_geometry(Dm, &nig, &nj, &w)) return 0;
  return (size_t)C3P_BIGD_MAX_WGS * G_NIMG * nig * nj * 64;
}

size_t c3p_bigd_lds_bytes(int Dm, int K, int Lmax) {
  int nig, nj, w;
  if (!c3p_bigd_geometry(Dm, &nig, &nj, &w)) return 0;
  return ((size_t)16 * nig * w + (size_t)K * Lmax) * sizeof(double);
}

hipError_t c3p_launch_rowphase(cplx* U, const double* phase, int B, int Dm, hipStream_t st) {
  const long total = (long)B * Dm * Dm;
  C3P_LAUNCH(rowphase_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, U, phase, Dm, total);
  return hipGetLastError();
}

hipError_t c3p_launch_bigd_chain(const MidArgs& A, double* arena, hipStream_t st) {
  int nig, nj, w;
  if (!c3p_bigd_geometry(A.Dm, &nig, &nj, &w)) return hipErrorInvalidValue;
  // one instantiation per geometry class (4 consecutive dimensions each)
  switch (nj) {
    case 11: return launch_b<6, 11, 46>(A, arena, st);
    case 12: return launch_b<6, 12, 50>(A, arena, st);
    case 13: return launch_b<7, 13, 54>(A, arena, st);
    case 14: return launch_b<7, 14, 58>(A, arena, st);
    case 15: return launch_b<8, 15, 62>(A, arena, st);
    case 16: return launch_b<8, 16, 66>(A, arena, st);
    case 17: return launch_b<9, 17, 70>(A, arena, st);
    case 18: return launch_b<9, 18, 74>(A, arena, st);
    case 19: return launch_b<10, 19, 78>(A, arena, st);
    case 20: return launch_b<10, 20, 82>(A, arena, st);
    case 21: return launch_b<11, 21, 86>(A, arena, st);
    case 22: return launch_b<11, 22, 90>(A, arena, st);
    case 23: return launch_b<12, 23, 94>(A, arena, st);
    default: return hipErrorInvalidValue;
  }
}
