// Shared by the real-arithmetic register-resident kernels of the Hermitian basis: the forward chain kernel (c3p_regr.hip) and
// the backward sweep (c3p_regrg.hip).  Geometry of a class Dm = 16 NRG + 1 and the small device helpers of the product loop.
#pragma once
#include <utility>

#include "c3p_common.h"

namespace {

constexpr int RR_MAXWAVES = 8;
constexpr int RR_CH = 32;    // control amplitudes staged per chunk of slices
constexpr int RR_KMAX = 16;  // control lines
enum { S_M0 = 0, S_M1, S_M2, S_M3, S_R0, S_R1, S_U0, S_U1, S_NSLOT };
enum { OP_P1 = 0, OP_P2, OP_P3, OP_P4, OP_EX, OP_CH };

template <int NRG>
struct RR {
  static constexpr int DM = 16 * NRG + 1;
  static constexpr int NT = NRG * NRG;  // tiles per column group (4 NRG columns: one wave, or a pair of waves on one SIMD)
  // image row stride (doubles): A-fragment reads (16 rows x 4 columns per 32 lanes) at most two-way on the 32 bank
  // pairs for every rotation, the 16-lane tile stores conflict free (brute-forced: 17 mod 32, or 2 mod 4)
  static constexpr int LD = (NRG == 4) ? DM + 1 : DM;
  static constexpr int BS = 2 * DM;  // border slot: row DM-1 (DM elements, corner last), column DM-1 (DM elements, corner last)
  static constexpr int DMP = DM + 1;
  static constexpr int IMG_D = DM * LD;
  static constexpr int TSET = NT * 256;  // elements of a tile set: element (tile, column group, lane) at tile * 256 + group * 64 + lane
  static constexpr int TAB_D = TSET + BS + 4;   // doubles per generator table: tile set, border slot, {mu, norm1, max |Im|, 1-norm of the symmetric part}
  static constexpr int LDS_D = IMG_D + S_NSLOT * BS + 8 * DMP + RR_KMAX * RR_CH + 2 * RR_MAXWAVES;
};

template <typename F, int... Is>
__device__ __forceinline__ void rr_static_for_impl(F&& f, std::integer_sequence<int, Is...>) {
  (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void rr_static_for(F&& f) {
  rr_static_for_impl(f, std::make_integer_sequence<int, N>{});
}

// the four 4-lane groups of every 16-lane row rotated by S groups (DPP row_ror): MFMA block b then holds what block
// (b - S) mod 4 held
template <int S>
__device__ __forceinline__ double rr_rot(double v) {
  if constexpr (S == 0) {
    return v;
  } else {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_mov_dpp(lo, 0x120 + 4 * S, 0xf, 0xf, false);
    hi = __builtin_amdgcn_mov_dpp(hi, 0x120 + 4 * S, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
  }
}

__device__ __forceinline__ int rr_opq(int v) {
  asm volatile("" : "+s"(v));
  return v;
}
template <typename T>
__device__ __forceinline__ T* rr_ubase(T* p) {
  return p + rr_opq(0);
}
// workgroup barrier that waits for the LDS traffic only
__device__ __forceinline__ void rr_bar() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ double rr_rfl(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_readfirstlane(lo);
  hi = __builtin_amdgcn_readfirstlane(hi);
  return __hiloint2double(hi, lo);
}

}  // namespace
