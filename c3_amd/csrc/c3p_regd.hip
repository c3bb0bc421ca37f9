// Register-resident propagator chains on the f64 matrix cores for matrix dimensions Dm = 16 n + 1
// (49, 65, 81): the 81 x 81 Lindblad superoperator of two qutrits (BASELINE cfg4,
// c3/libraries/propagation.py:551-585) and its D = 7 sibling.
//
// One workgroup of FOUR wavefronts (one per SIMD, 512 registers each) owns a chain.  A complex Dm x Dm matrix
// is split into a CORE of (Dm-1) x (Dm-1) = 16n x 16n elements and a one-element BORDER (last row, last column).
//   * Core matrices live in REGISTERS: wave w owns the 4n columns [4n w, 4n w + 4n) -- n column blocks of 4 --
//     and all 16n rows, as n x n tiles in the C/D layout of v_mfma_f64_4x4x4_4b_f64 (lane (q,b,p) of tile
//     (Ig,jj) holds row 16 Ig + 4 b + q, column 4 (n w + jj) + p), real and imaginary parts in separate registers.
//   * A product C = L R streams the LEFT operand from one LDS image (complex, row stride Dm + 1: the 16 x 4
//     A fragments are conflict-free ds_read_b128) and takes the RIGHT operand from the registers of the wave that
//     owns those columns: the B operand of K-step 4 Ig' + b' is block b' of tile (Ig',jj), broadcast to the four
//     blocks with ds_swizzle -- no global or LDS traffic for it, and the result lands in the layout the next
//     product needs.
//   * Complex products use three real products (P = Lr Rr, Q = Li Ri, R = (Lr + Li)(Rr + Ri); Re C = P - Q,
//     Im C = R - P - Q): 3 n^2 MFMAs per K-step and wave instead of 4 n^2, zero padding (the core is a multiple of
//     the tile in every direction).
//   * The border (last row / column / k = Dm - 1) is exact vector work: a rank-1 update of the core, a
//     matrix-vector product from the LDS image, a vector-matrix product against the register tiles with a lane
//     reduction; border elements of every live matrix stay in small LDS slots, one element per lane.
//   * exp: T18 (Bader-Blanes-Casas, 5 products) + s squarings + 1 chain product per slice.  Only three tile sets
//     per slice visit the per-workgroup global arena (A2, B2 and the running product U): 0.6 MB per slice instead of
//     the 2.5 MB of the arena kernel in c3p_bigd.hip; the polynomial combinations are formed in registers.
#include <utility>

#include "c3p_common.h"
#include "c3p_kernels.h"
#include "c3p_midd.h"
#include "c3p_regd.h"

extern __shared__ __attribute__((aligned(16))) double c3p_rg_lds[];

namespace {

constexpr int RG_WAVES = 4;
constexpr int RG_THREADS = 64 * RG_WAVES;
constexpr int RG_CH = 32;    // control amplitudes staged per chunk of slices
constexpr int RG_KMAX = 16;  // control lines
// border slots in LDS
enum { S_M0 = 0, S_M1, S_M2, S_M3, S_R0, S_R1, S_U0, S_U1, S_NSLOT };
enum { OP_P1 = 0, OP_P2, OP_P3, OP_P4, OP_EX, OP_CH };
// arena tile sets
enum { AR_X = 0, AR_A2, AR_B2, AR_B3, AR_U, AR_NSET };

template <int NRG>
struct RG {
  static constexpr int DM = 16 * NRG + 1;
  static constexpr int NJ = NRG;         // column blocks per wave
  static constexpr int NKS = 4 * NRG;    // K-steps of the core
  static constexpr int NT = NRG * NJ;    // tiles per wave
  static constexpr int LD = DM + 1;      // image row stride (complex elements), = 2 mod 16
  static constexpr int BS = 2 * DM;      // border slot: row DM-1 (DM elements, corner last), column DM-1 (DM elements, corner last)
  static constexpr int DMP = DM + 1;
  static constexpr int KP = (DM + 2) / 3;  // k range of one matrix-vector part
  static constexpr int IMG_C = DM * LD;
  static constexpr int TSET_C = NT * RG_THREADS;  // complex elements of a tile set: element (tile, thread) at tile * 256 + thread
  static constexpr int TAB_D = 2 * (TSET_C + BS) + 4;  // doubles per generator table: tile set, border slot, {mu_r, mu_i, norm1, 0}
  static constexpr int LDS_C = IMG_C + S_NSLOT * BS + 3 * DMP;
  static constexpr int LDS_D = 2 * LDS_C + RG_KMAX * RG_CH + 2 * RG_WAVES;
};

template <typename F, int... Is>
__device__ __forceinline__ void rg_static_for_impl(F&& f, std::integer_sequence<int, Is...>) {
  (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void rg_static_for(F&& f) {
  rg_static_for_impl(f, std::make_integer_sequence<int, N>{});
}

// value of lane (q, BP, p) for every lane (q, b, p): block BP of a C/D-layout register as the B operand of all four blocks
template <int BP>
__device__ __forceinline__ double rg_bcast(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_ds_swizzle(lo, 0x13 | (BP << 7));  // bit mode: lane' = (lane & 0b10011) | (BP << 2)
  hi = __builtin_amdgcn_ds_swizzle(hi, 0x13 | (BP << 7));
  return __hiloint2double(hi, lo);
}

// Global accesses are (uniform base in SGPRs, opaque so the address arithmetic is not hoisted out of the slice loop and
// spilled) + (32-bit per-lane byte offset): global_load/store in the saddr form, one offset register for all tiles.
__device__ __forceinline__ int rg_opq(int v) {
  asm volatile("" : "+s"(v));
  return v;
}
template <typename T>
__device__ __forceinline__ T* rg_ubase(T* p) {
  return p + rg_opq(0);
}
__device__ __forceinline__ cplx rg_ld(const cplx* ubase, unsigned voff) {
  return *reinterpret_cast<const cplx*>(reinterpret_cast<const char*>(ubase) + voff);
}
__device__ __forceinline__ void rg_st(cplx* ubase, unsigned voff, cplx v) {
  *reinterpret_cast<cplx*>(reinterpret_cast<char*>(ubase) + voff) = v;
}

__device__ __forceinline__ double rg_rfl(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_readfirstlane(lo);
  hi = __builtin_amdgcn_readfirstlane(hi);
  return __hiloint2double(hi, lo);
}

template <int NRG, bool DUS>
__global__ void __launch_bounds__(RG_THREADS, 1) regd_chain_kernel(MidArgs A, cplx* arena_base) {
  using G = RG<NRG>;
  constexpr int DM = G::DM, NJ = G::NJ, NKS = G::NKS, LD = G::LD, BS = G::BS, DMP = G::DMP, KP = G::KP;
  constexpr int TSET = G::TSET_C;
  // lane indices are re-derived from an opaque copy of the thread id at the start of every phase (refresh): otherwise
  // the compiler hoists every address / mask that depends on them out of the slice loop and spills them
  const int tid0 = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid0 >> 6);
  int tid = tid0, lane = tid & 63;
  int q = lane >> 4, b = (lane >> 2) & 3, p = lane & 3;
  cplx* img = reinterpret_cast<cplx*>(c3p_rg_lds);
  cplx* brd = img + G::IMG_C;
  cplx* cpart = brd + S_NSLOT * BS;
  double* sg = reinterpret_cast<double*>(cpart + 3 * DMP);
  double* red = sg + RG_KMAX * RG_CH;
  const int col0 = 4 * NJ * wave;  // first column of this wave
  int rowC = 4 * b + q;            // row of a C/D-layout element inside its row group
  int rowA = 4 * b + p;            // row of an A-fragment element inside its row group
  cplx* arena = arena_base + (long)blockIdx.x * AR_NSET * TSET;
  const int K = A.K;
  unsigned voff = (unsigned)tid * (unsigned)sizeof(cplx);  // this thread's byte offset inside a tile (256 complex)
  auto refresh = [&]() {
    int t_ = tid0;
    asm volatile("" : "+v"(t_));
    tid = t_;
    lane = tid & 63;
    q = lane >> 4, b = (lane >> 2) & 3, p = lane & 3;
    rowC = 4 * b + q;
    rowA = 4 * b + p;
    voff = (unsigned)tid * (unsigned)sizeof(cplx);
  };

  double Rr[NRG][NJ], Ri[NRG][NJ];                // right operand of the next product
  double aP[NRG][NJ], aQ[NRG][NJ], aR[NRG][NJ];   // accumulators; after a product aP = Re C, aR = Im C

  auto mfma = [](double a, double bb, double c) -> double { return __builtin_amdgcn_mfma_f64_4x4x4f64(a, bb, c, 0, 0, 0); };
  auto zero_acc = [&]() {
#pragma unroll
    for (int Ig = 0; Ig < NRG; ++Ig)
#pragma unroll
      for (int jj = 0; jj < NJ; ++jj) aP[Ig][jj] = aQ[Ig][jj] = aR[Ig][jj] = 0.0;
  };
  // a border element (slot index tid) also sits in the image: row DM-1 or column DM-1
  auto border_to_image = [&](cplx v) {
    if (tid < DM) img[(DM - 1) * LD + tid] = v;
    else if (tid < 2 * DM - 1) img[(tid - DM) * LD + DM - 1] = v;
  };

  // C = (init) + L R: L = the LDS image, R = (Rr, Ri) with borders in slot sr; the accumulators carry the initial
  // value (aP = Re, aQ = 0, aR = Re + Im); border of the initial value in slot si (or si < 0); border of C -> slot sd.
  // Ends before the workgroup barrier that publishes the border partials.
  auto product = [&](int sr, int si, int sd) {
    const cplx* rb = brd + sr * BS;
    {
      const cplx* pa = img + rowA * LD + q;
      cplx aC[NRG], aN[NRG];
#pragma unroll
      for (int Ig = 0; Ig < NRG; ++Ig) aC[Ig] = pa[16 * Ig * LD];
      rg_static_for<NKS>([&](auto kk_) {
        constexpr int kk = decltype(kk_)::value;
        constexpr int Igp = kk >> 2, bp = kk & 3;
        if constexpr (kk + 1 < NKS) {
#pragma unroll
          for (int Ig = 0; Ig < NRG; ++Ig) aN[Ig] = pa[16 * Ig * LD + 4 * (kk + 1)];
        }
        double br[NJ], bi[NJ], bs[NJ];
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj) {
          br[jj] = rg_bcast<bp>(Rr[Igp][jj]);
          bi[jj] = rg_bcast<bp>(Ri[Igp][jj]);
          bs[jj] = br[jj] + bi[jj];
        }
#pragma unroll
        for (int Ig = 0; Ig < NRG; ++Ig) {
          const double as = aC[Ig].x + aC[Ig].y;
#pragma unroll
          for (int jj = 0; jj < NJ; ++jj) {
            aP[Ig][jj] = mfma(aC[Ig].x, br[jj], aP[Ig][jj]);
            aQ[Ig][jj] = mfma(aC[Ig].y, bi[jj], aQ[Ig][jj]);
            aR[Ig][jj] = mfma(as, bs[jj], aR[Ig][jj]);
          }
        }
        if constexpr (kk + 1 < NKS) {
#pragma unroll
          for (int Ig = 0; Ig < NRG; ++Ig) aC[Ig] = aN[Ig];
        }
      });
    }
#pragma unroll
    for (int Ig = 0; Ig < NRG; ++Ig)
#pragma unroll
      for (int jj = 0; jj < NJ; ++jj) {
        const double t = aP[Ig][jj] + aQ[Ig][jj];
        aP[Ig][jj] = aP[Ig][jj] - aQ[Ig][jj];
        aR[Ig][jj] = aR[Ig][jj] - t;
      }
    // k = DM-1: rank-1 update of the core
    {
      cplx a80[NRG], b80[NJ];
#pragma unroll
      for (int Ig = 0; Ig < NRG; ++Ig) a80[Ig] = img[(16 * Ig + rowC) * LD + DM - 1];
#pragma unroll
      for (int jj = 0; jj < NJ; ++jj) b80[jj] = rb[col0 + 4 * jj + p];
#pragma unroll
      for (int Ig = 0; Ig < NRG; ++Ig)
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj) {
          aP[Ig][jj] = fma(a80[Ig].x, b80[jj].x, fma(-a80[Ig].y, b80[jj].y, aP[Ig][jj]));
          aR[Ig][jj] = fma(a80[Ig].x, b80[jj].y, fma(a80[Ig].y, b80[jj].x, aR[Ig][jj]));
        }
    }
    // column DM-1 of C (and the corner): L . (column DM-1 of R), three k ranges per row
    if (tid < 3 * DM) {
      const int part = tid / DM, i = tid - part * DM;
      const int k0 = part * KP, k1 = (k0 + KP < DM) ? k0 + KP : DM;
      const cplx* ar = img + i * LD;
      const cplx* cb = rb + DM;
      double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
      for (int k = k0; k < k1; ++k) {
        const cplx a = ar[k], v = cb[k];
        s0 = fma(a.x, v.x, s0);
        s1 = fma(a.y, v.y, s1);
        s2 = fma(a.x, v.y, s2);
        s3 = fma(a.y, v.x, s3);
      }
      cpart[part * DMP + i] = cmake(s0 - s1, s2 + s3);
    }
    // row DM-1 of C, columns of this wave: (row DM-1 of L) . R, partial over the lane's rows, reduced over (q, b)
    {
      cplx a8[NRG];
#pragma unroll
      for (int Ig = 0; Ig < NRG; ++Ig) a8[Ig] = img[(DM - 1) * LD + 16 * Ig + rowC];
      const cplx corner = img[(DM - 1) * LD + DM - 1];
#pragma unroll
      for (int jj = 0; jj < NJ; ++jj) {
        double vr = 0.0, vi = 0.0;
#pragma unroll
        for (int Ig = 0; Ig < NRG; ++Ig) {
          vr = fma(a8[Ig].x, Rr[Ig][jj], fma(-a8[Ig].y, Ri[Ig][jj], vr));
          vi = fma(a8[Ig].x, Ri[Ig][jj], fma(a8[Ig].y, Rr[Ig][jj], vi));
        }
#pragma unroll
        for (int m = 4; m <= 32; m <<= 1) {
          vr += __shfl_xor(vr, m);
          vi += __shfl_xor(vi, m);
        }
        if (lane < 4) {
          const int j = col0 + 4 * jj + p;
          const cplx rbj = rb[j];
          vr = fma(corner.x, rbj.x, fma(-corner.y, rbj.y, vr));
          vi = fma(corner.x, rbj.y, fma(corner.y, rbj.x, vi));
          if (si >= 0) {
            const cplx ini = brd[si * BS + j];
            vr += ini.x;
            vi += ini.y;
          }
          brd[sd * BS + j] = cmake(vr, vi);
        }
      }
    }
  };
  // after the barrier: column DM-1 (and both copies of the corner) of the product
  auto finalize = [&](int si, int sd) {
    if (tid >= DM - 1 && tid < BS) {
      const int i = (tid == DM - 1) ? DM - 1 : tid - DM;
      const cplx v0 = cpart[i], v1 = cpart[DMP + i], v2 = cpart[2 * DMP + i];
      double vr = (v0.x + v1.x) + v2.x, vi = (v0.y + v1.y) + v2.y;
      if (si >= 0) {
        const cplx ini = brd[si * BS + tid];
        vr += ini.x;
        vi += ini.y;
      }
      brd[sd * BS + tid] = cmake(vr, vi);
    }
  };
  auto image_from_C = [&]() {
#pragma unroll
    for (int Ig = 0; Ig < NRG; ++Ig)
#pragma unroll
      for (int jj = 0; jj < NJ; ++jj) img[(16 * Ig + rowC) * LD + col0 + 4 * jj + p] = cmake(aP[Ig][jj], aR[Ig][jj]);
  };
  auto R_from_C = [&]() {
#pragma unroll
    for (int Ig = 0; Ig < NRG; ++Ig)
#pragma unroll
      for (int jj = 0; jj < NJ; ++jj) {
        Rr[Ig][jj] = aP[Ig][jj];
        Ri[Ig][jj] = aR[Ig][jj];
      }
  };
  auto park_C = [&](int set) {
    cplx* dst = rg_ubase(arena + set * TSET);
#pragma unroll
    for (int Ig = 0; Ig < NRG; ++Ig)
#pragma unroll
      for (int jj = 0; jj < NJ; ++jj) rg_st(dst + (Ig * NJ + jj) * RG_THREADS, voff, cmake(aP[Ig][jj], aR[Ig][jj]));
  };
  auto unpark_R = [&](int set) {
    const cplx* src = rg_ubase(arena + set * TSET);
#pragma unroll
    for (int Ig = 0; Ig < NRG; ++Ig)
#pragma unroll
      for (int jj = 0; jj < NJ; ++jj) {
        const cplx v = rg_ld(src + (Ig * NJ + jj) * RG_THREADS, voff);
        Rr[Ig][jj] = v.x;
        Ri[Ig][jj] = v.y;
      }
  };
  auto park_R = [&](int set) {
    cplx* dst = rg_ubase(arena + set * TSET);
#pragma unroll
    for (int Ig = 0; Ig < NRG; ++Ig)
#pragma unroll
      for (int jj = 0; jj < NJ; ++jj) rg_st(dst + (Ig * NJ + jj) * RG_THREADS, voff, cmake(Rr[Ig][jj], Ri[Ig][jj]));
  };
  // dst[row][col] = f * C, row-major complex Dm x Dm (f = e^{trace shift}); borders from slot
  auto store_out = [&](cplx* dst_, int slot, double fr, double fi) {
    cplx* dst = rg_ubase(dst_);
    const unsigned lo = (unsigned)(rowC * DM + col0 + p) * (unsigned)sizeof(cplx);
#pragma unroll
    for (int Ig = 0; Ig < NRG; ++Ig)
#pragma unroll
      for (int jj = 0; jj < NJ; ++jj)
        rg_st(dst + 16 * Ig * DM + 4 * jj, lo,
              cmake(fma(fr, aP[Ig][jj], -fi * aR[Ig][jj]), fma(fr, aR[Ig][jj], fi * aP[Ig][jj])));
    if (tid < BS - 1) {
      const cplx v = brd[slot * BS + tid];
      const int row = tid < DM ? DM - 1 : tid - DM, col = tid < DM ? tid : DM - 1;
      dst[row * DM + col] = cmake(fma(fr, v.x, -fi * v.y), fma(fr, v.y, fi * v.x));
    }
  };

  const long nchains = (long)A.B * A.S;
  for (long chain = blockIdx.x; chain < nchains; chain += gridDim.x) {
    const int sample = (int)(chain / A.S);
    const int seg = (int)(chain - (long)sample * A.S);
    const int n0 = (int)(((long)seg * A.N) / A.S);
    const int n1 = (int)(((long)(seg + 1) * A.N) / A.S);
    const int len = n1 - n0;
    const double* tabs = A.tables + (long)(A.tab_per_sample ? sample : 0) * (1 + K) * G::TAB_D;
    auto meta = [&](int k1) -> const double* { return tabs + (long)k1 * G::TAB_D + 2 * (TSET + BS); };
    __syncthreads();  // the previous chain is done with the LDS
    // plan: squarings from ||G0||_1 + sum_k max_t |c_k(t)| ||G_k||_1 over the segment
    double nrm = meta(0)[2];
    for (int k = 0; k < K; ++k) {
      const double* s = A.signals + ((long)sample * K + k) * A.N + n0;
      double cmax = 0.0;
      for (int t = tid; t < len; t += RG_THREADS) cmax = fmax(cmax, fabs(s[t]));
      for (int o = 32; o >= 1; o >>= 1) cmax = fmax(cmax, __shfl_xor(cmax, o));
      if (lane == 0) red[wave] = cmax;
      __syncthreads();
      cmax = 0.0;
      for (int w = 0; w < RG_WAVES; ++w) cmax = fmax(cmax, red[w]);
      __syncthreads();
      nrm = fma(cmax, meta(k + 1)[2], nrm);
    }
    nrm = rg_rfl(nrm);
    int s18 = 0;
    {
      double pth = C3P_T18_THETA;
      while (pth < nrm && s18 < 40) {
        pth *= 2.0;
        ++s18;
      }
    }
    const int ps = __builtin_amdgcn_readfirstlane(s18);
    const double scale = ldexp(1.0, -ps);

    double mu_r = 0.0, mu_i = 0.0, mus_r = 0.0, mus_i = 0.0;
    auto stage_signals = [&](int t) {  // slices [t, t + RG_CH) of the segment
      for (int e = tid; e < K * RG_CH; e += RG_THREADS) {
        const int k = e / RG_CH, tt = e - k * RG_CH;
        sg[e] = (t + tt < len) ? A.signals[((long)sample * K + k) * A.N + n0 + t + tt] : 0.0;
      }
    };
    // X = 2^-s (G0 + sum_k c_k G_k): tiles -> (Rr, Ri), borders -> slot M0, everything -> image
    auto assemble = [&](int tt) {
      const cplx* T0 = rg_ubase(reinterpret_cast<const cplx*>(tabs));
      mu_r = meta(0)[0];
      mu_i = meta(0)[1];
#pragma unroll
      for (int Ig = 0; Ig < NRG; ++Ig)
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj) {
          const cplx v = rg_ld(T0 + (Ig * NJ + jj) * RG_THREADS, voff);
          Rr[Ig][jj] = scale * v.x;
          Ri[Ig][jj] = scale * v.y;
        }
      cplx bv = cmake(0.0, 0.0);
      if (tid < BS) {
        bv = T0[TSET + tid];
        bv.x *= scale;
        bv.y *= scale;
      }
      for (int k = 0; k < K; ++k) {
        const cplx* Tk = rg_ubase(reinterpret_cast<const cplx*>(tabs + (long)(k + 1) * G::TAB_D));
        const double c = sg[k * RG_CH + tt];
        const double w = scale * c;
        mu_r = fma(c, meta(k + 1)[0], mu_r);
        mu_i = fma(c, meta(k + 1)[1], mu_i);
#pragma unroll
        for (int Ig = 0; Ig < NRG; ++Ig)
#pragma unroll
          for (int jj = 0; jj < NJ; ++jj) {
            const cplx v = rg_ld(Tk + (Ig * NJ + jj) * RG_THREADS, voff);
            Rr[Ig][jj] = fma(w, v.x, Rr[Ig][jj]);
            Ri[Ig][jj] = fma(w, v.y, Ri[Ig][jj]);
          }
        if (tid < BS) {
          const cplx v = Tk[TSET + tid];
          bv.x = fma(w, v.x, bv.x);
          bv.y = fma(w, v.y, bv.y);
        }
      }
      if (tid < BS) {
        brd[S_M0 * BS + tid] = bv;
        border_to_image(bv);
      }
#pragma unroll
      for (int Ig = 0; Ig < NRG; ++Ig)
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj) img[(16 * Ig + rowC) * LD + col0 + 4 * jj + p] = cmake(Rr[Ig][jj], Ri[Ig][jj]);
      park_R(AR_X);
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
    for (;;) {
      refresh();
      product(sr, si, sd);
      __syncthreads();  // A: everyone is done with the image; border partials are visible
      refresh();
      finalize(si, sd);
      bool next_slice = false;
      if (op == OP_P1) {  // C = A2: it becomes the right operand (the image still holds X)
        R_from_C();
        zero_acc();
        op = OP_P2, sr = S_M1, si = -1, sd = S_M2;
      } else if (op == OP_P2) {  // C = A3 = X A2: park A2, A3 becomes both operands
        park_R(AR_A2);
        image_from_C();
        if (tid < BS) border_to_image(brd[S_M2 * BS + tid]);
        R_from_C();
        zero_acc();
        op = OP_P3, sr = S_M2, si = -1, sd = S_M3;
      } else if (op == OP_P3) {  // C = A6: the T18 combinations of X, A2 (arena), A3 (R), A6 (C)
        {
          const cplx* srcx = rg_ubase(arena + AR_X * TSET);
          const cplx* src = rg_ubase(arena + AR_A2 * TSET);
          cplx* dst = rg_ubase(arena + AR_B2 * TSET);
          cplx* dst3 = rg_ubase(arena + AR_B3 * TSET);
          cplx a2[NRG][NJ], xx[NRG][NJ];
#pragma unroll
          for (int Ig = 0; Ig < NRG; ++Ig)
#pragma unroll
            for (int jj = 0; jj < NJ; ++jj) {
              xx[Ig][jj] = rg_ld(srcx + (Ig * NJ + jj) * RG_THREADS, voff);
              a2[Ig][jj] = rg_ld(src + (Ig * NJ + jj) * RG_THREADS, voff);
            }
#pragma unroll
          for (int Ig = 0; Ig < NRG; ++Ig)
#pragma unroll
            for (int jj = 0; jj < NJ; ++jj) {
              const double dg = (16 * Ig + rowC == col0 + 4 * jj + p) ? 1.0 : 0.0;
              const double xr = xx[Ig][jj].x, xi = xx[Ig][jj].y, a2r = a2[Ig][jj].x, a2i = a2[Ig][jj].y;
              const double a3r = Rr[Ig][jj], a3i = Ri[Ig][jj], a6r = aP[Ig][jj], a6i = aR[Ig][jj];
              // B1 -> image (left operand of A9 = B1 B5 + B4)
              img[(16 * Ig + rowC) * LD + col0 + 4 * jj + p] =
                  cmake(fma(C3P_T18_A31, a3r, fma(C3P_T18_A21, a2r, C3P_T18_A11 * xr)),
                        fma(C3P_T18_A31, a3i, fma(C3P_T18_A21, a2i, C3P_T18_A11 * xi)));
              // B2 -> arena
              rg_st(dst + (Ig * NJ + jj) * RG_THREADS, voff,
                    cmake(fma(C3P_T18_B61, a6r, fma(C3P_T18_B31, a3r, fma(C3P_T18_B21, a2r, C3P_T18_B11 * xr))),
                          fma(C3P_T18_B61, a6i, fma(C3P_T18_B31, a3i, fma(C3P_T18_B21, a2i, C3P_T18_B11 * xi)))));
              // B3 -> arena
              rg_st(dst3 + (Ig * NJ + jj) * RG_THREADS, voff,
                    cmake(fma(C3P_T18_B62, a6r, fma(C3P_T18_B32, a3r, fma(C3P_T18_B22, a2r, fma(C3P_T18_B12, xr, C3P_T18_B02 * dg)))),
                          fma(C3P_T18_B62, a6i, fma(C3P_T18_B32, a3i, fma(C3P_T18_B22, a2i, C3P_T18_B12 * xi)))));
              // B5 -> right operand
              Rr[Ig][jj] = fma(C3P_T18_B64, a6r, fma(C3P_T18_B34, a3r, C3P_T18_B24 * a2r));
              Ri[Ig][jj] = fma(C3P_T18_B64, a6i, fma(C3P_T18_B34, a3i, C3P_T18_B24 * a2i));
              // B4 -> initial value of the accumulators
              const double b4r = fma(C3P_T18_B63, a6r, fma(C3P_T18_B33, a3r, fma(C3P_T18_B23, a2r, fma(C3P_T18_B13, xr, C3P_T18_B03 * dg))));
              const double b4i = fma(C3P_T18_B63, a6i, fma(C3P_T18_B33, a3i, fma(C3P_T18_B23, a2i, C3P_T18_B13 * xi)));
              aP[Ig][jj] = b4r;
              aQ[Ig][jj] = 0.0;
              aR[Ig][jj] = b4r + b4i;
            }
        }
        if (tid < BS) {
          const double dg = (tid == DM - 1 || tid == BS - 1) ? 1.0 : 0.0;
          const cplx x = brd[S_M0 * BS + tid], a2 = brd[S_M1 * BS + tid], a3 = brd[S_M2 * BS + tid], a6 = brd[S_M3 * BS + tid];
          border_to_image(cmake(fma(C3P_T18_A31, a3.x, fma(C3P_T18_A21, a2.x, C3P_T18_A11 * x.x)),
                                fma(C3P_T18_A31, a3.y, fma(C3P_T18_A21, a2.y, C3P_T18_A11 * x.y))));
          brd[S_M3 * BS + tid] = cmake(fma(C3P_T18_B61, a6.x, fma(C3P_T18_B31, a3.x, fma(C3P_T18_B21, a2.x, C3P_T18_B11 * x.x))),
                                       fma(C3P_T18_B61, a6.y, fma(C3P_T18_B31, a3.y, fma(C3P_T18_B21, a2.y, C3P_T18_B11 * x.y))));
          brd[S_M0 * BS + tid] = cmake(fma(C3P_T18_B62, a6.x, fma(C3P_T18_B32, a3.x, fma(C3P_T18_B22, a2.x, fma(C3P_T18_B12, x.x, C3P_T18_B02 * dg)))),
                                       fma(C3P_T18_B62, a6.y, fma(C3P_T18_B32, a3.y, fma(C3P_T18_B22, a2.y, C3P_T18_B12 * x.y))));
          brd[S_M1 * BS + tid] = cmake(fma(C3P_T18_B64, a6.x, fma(C3P_T18_B34, a3.x, C3P_T18_B24 * a2.x)),
                                       fma(C3P_T18_B64, a6.y, fma(C3P_T18_B34, a3.y, C3P_T18_B24 * a2.y)));
          brd[S_M2 * BS + tid] = cmake(fma(C3P_T18_B63, a6.x, fma(C3P_T18_B33, a3.x, fma(C3P_T18_B23, a2.x, fma(C3P_T18_B13, x.x, C3P_T18_B03 * dg)))),
                                       fma(C3P_T18_B63, a6.y, fma(C3P_T18_B33, a3.y, fma(C3P_T18_B23, a2.y, C3P_T18_B13 * x.y))));
        }
        op = OP_P4, sr = S_M1, si = S_M2, sd = S_R0;
      } else if (op == OP_P4) {  // C = A9: left operand B3 + A9, right operand A9, initial value B2
        {
          const cplx* src3 = rg_ubase(arena + AR_B3 * TSET);
#pragma unroll
          for (int Ig = 0; Ig < NRG; ++Ig)
#pragma unroll
            for (int jj = 0; jj < NJ; ++jj) {
              const cplx v = rg_ld(src3 + (Ig * NJ + jj) * RG_THREADS, voff);
              img[(16 * Ig + rowC) * LD + col0 + 4 * jj + p] = cmake(v.x + aP[Ig][jj], v.y + aR[Ig][jj]);
            }
        }
        if (tid < BS) {
          const cplx b3 = brd[S_M0 * BS + tid], a9 = brd[S_R0 * BS + tid];
          border_to_image(cmake(b3.x + a9.x, b3.y + a9.y));
        }
        R_from_C();
        {
          const cplx* src = rg_ubase(arena + AR_B2 * TSET);
#pragma unroll
          for (int Ig = 0; Ig < NRG; ++Ig)
#pragma unroll
            for (int jj = 0; jj < NJ; ++jj) {
              const cplx v = rg_ld(src + (Ig * NJ + jj) * RG_THREADS, voff);
              aP[Ig][jj] = v.x;
              aQ[Ig][jj] = 0.0;
              aR[Ig][jj] = v.x + v.y;
            }
        }
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
            double sn, cs;
            sincos(mu_i, &sn, &cs);
            const double er = exp(mu_r);
            store_out(A.dUs_out + ((long)sample * A.N + n0 + t) * DM * DM, sd, er * cs, er * sn);
          }
          if (first) {
            park_C(AR_U);
            if (tid < BS) brd[S_U0 * BS + tid] = brd[sd * BS + tid];
            ucur = S_U0;
            first = false;
            mus_r = mu_r;
            mus_i = c3p_phase_add(0.0, mu_i);
            next_slice = true;
          } else {  // U <- C U
            image_from_C();
            if (tid < BS) border_to_image(brd[sd * BS + tid]);
            unpark_R(AR_U);
            zero_acc();
            mus_r += mu_r;
            mus_i = c3p_phase_add(mus_i, mu_i);
            op = OP_CH, sr = ucur, si = -1, sd = ucur ^ 1;
          }
        }
      } else {  // OP_CH: C = the running product
        park_C(AR_U);
        ucur ^= 1;
        next_slice = true;
      }
      if (next_slice) {
        ++t;
        if (t == len) break;
        if ((t % RG_CH) == 0) {
          stage_signals(t);
          __syncthreads();
        }
        assemble(t % RG_CH);
        zero_acc();
        op = OP_P1, sr = S_M0, si = -1, sd = S_M1;
      }
      __syncthreads();  // B: image, border slots and arena writes of this phase are visible
    }
    // segment product (the frame-rotation row phases are a separate epilogue)
    double sn, cs;
    sincos(mus_i, &sn, &cs);
    const double er = exp(mus_r);
    store_out(A.seg_out + chain * DM * DM, ucur, er * cs, er * sn);
  }
}

// Generator tables in the kernel's layout: G = -i dt (h - mu) (unitary) or the Lindblad generator pieces
// L0 = dt (clp - i (H0 (x) I - I (x) H0^T)), Lk = -i dt (Hk (x) I - I (x) Hk^T) (propagation.py:565-582), trace shifted.
__global__ void __launch_bounds__(256) regd_prep_kernel(RegdPrepArgs P) {
  __shared__ double redr[256], redi[256];
  __shared__ double mu[2];
  const int tid = threadIdx.x;
  const int ti = blockIdx.x % (1 + P.K);
  const int sample = blockIdx.x / (1 + P.K);
  const int D = P.Dm, Dh = P.Dh;
  const int NRG = (D - 1) / 16, NJ = NRG;
  const cplx* h = (ti == 0) ? P.h0 + (long)sample * P.h0_bstride : P.hks + (long)sample * P.hks_bstride + (long)(ti - 1) * Dh * Dh;
  auto gelem = [&](int row, int col) -> cplx {
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
    mu[0] = a / D;
    mu[1] = bq / D;
  }
  __syncthreads();
  double cs = 0;
  for (int j = tid; j < D; j += 256) {
    double s = 0;
    for (int i = 0; i < D; ++i) {
      cplx v = gelem(i, j);
      if (i == j) {
        v.x -= mu[0];
        v.y -= mu[1];
      }
      s += hypot(v.x, v.y);
    }
    cs = fmax(cs, s);
  }
  __syncthreads();
  redr[tid] = cs;
  __syncthreads();
  const int TSET = NRG * NJ * 256, BS = 2 * D;
  const long TAB_D = 2L * (TSET + BS) + 4;
  double* out = P.tables + ((long)sample * (1 + P.K) + ti) * TAB_D;
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
      row = eb < D ? D - 1 : eb - D;
      col = eb < D ? eb : D - 1;
    }
    cplx g = gelem(row, col);
    if (row == col) {
      g.x -= mu[0];
      g.y -= mu[1];
    }
    out[2 * e] = g.x;
    out[2 * e + 1] = g.y;
  }
  if (tid == 0) {
    double nrm = 0;
    for (int i = 0; i < 256; ++i) nrm = fmax(nrm, redr[i]);
    double* m = out + 2L * (TSET + BS);
    m[0] = mu[0];
    m[1] = mu[1];
    m[2] = nrm;
    m[3] = 0.0;
  }
}

template <int NRG>
hipError_t launch_r(const MidArgs& A, void* arena, hipStream_t st) {
  const size_t lds = (size_t)RG<NRG>::LDS_D * sizeof(double);
  const long nchains = (long)A.B * A.S;
  const unsigned grid = (unsigned)(nchains < C3P_REGD_MAX_WGS ? nchains : C3P_REGD_MAX_WGS);
  auto go = [&](auto kern) -> hipError_t {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(RG_THREADS), lds, st, A, reinterpret_cast<cplx*>(arena));
    return hipGetLastError();
  };
  if (A.dUs_out) return go(regd_chain_kernel<NRG, true>);
  return go(regd_chain_kernel<NRG, false>);
}

}  // namespace

bool c3p_regd_supported(int Dm) { return Dm == 49 || Dm == 65 || Dm == 81; }

size_t c3p_regd_table_doubles(int Dm, int K) {
  if (!c3p_regd_supported(Dm)) return 0;
  const int n = (Dm - 1) / 16;
  return (size_t)(1 + K) * (2 * ((size_t)n * n * 256 + 2 * Dm) + 4);
}

size_t c3p_regd_arena_bytes(int Dm) {
  if (!c3p_regd_supported(Dm)) return 0;
  const int n = (Dm - 1) / 16;
  return (size_t)C3P_REGD_MAX_WGS * AR_NSET * n * n * 256 * sizeof(cplx);
}

hipError_t c3p_launch_regd_prep(const RegdPrepArgs& P, int nsamp, hipStream_t st) {
  if (!c3p_regd_supported(P.Dm)) return hipErrorInvalidValue;
  hipLaunchKernelGGL(regd_prep_kernel, dim3((unsigned)(nsamp * (1 + P.K))), dim3(256), 0, st, P);
  return hipGetLastError();
}

hipError_t c3p_launch_regd_chain(const MidArgs& A, void* arena, hipStream_t st) {
  if (A.K > RG_KMAX) return hipErrorInvalidValue;
  switch (A.Dm) {
    case 49: return launch_r<3>(A, arena, st);
    case 65: return launch_r<4>(A, arena, st);
    case 81: return launch_r<5>(A, arena, st);
    default: return hipErrorInvalidValue;
  }
}
