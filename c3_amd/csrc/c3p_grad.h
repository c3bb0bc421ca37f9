// Launcher interface of the control-gradient kernels (c3p_grad.hip).
#pragma once
#include "c3p_common.h"

struct GradArgs {
  const cplx* h0;
  long h0_bstride;  // elements between samples (0 = shared)
  const cplx* hks;
  long hks_bstride;
  const double* signals;   // [B,K,N]
  const double* fr_phase;  // [B,D] or null
  const cplx* Ubar;        // [B,D,D] cotangent of U
  double dt;
  int B, K, N, D, ld;
  int S;  // time segments per sample, balanced: segment s covers [s N / S, (s+1) N / S)
  cplx* seg;     // [B,S,D,D] segment products (no frame rotation)
  cplx* Mb;      // [B,S,D,D] adjoint state at the END of each segment
  double* grad;  // [B,K,N]
  cplx* zout;    // [B,N,D,D] or null: Z_n = dU_n^H L(X_n, M_{n+1}), the cotangent of the generator G_n = -i dt H_n
  cplx* scratch;  // GLOBAL variant: scratch_stride elements per workgroup
  long scratch_stride;
};

#define C3P_GRAD_NMAT 19  // matrices a backward workgroup keeps (LDS or scratch)

int c3p_grad_threads(int D);
size_t c3p_grad_lds_bytes(int D);  // LDS variant footprint; > 150 KB => use the GLOBAL variant
hipError_t c3p_launch_grad_seg(const GradArgs& A, bool global_scratch, hipStream_t st);   // segment products
hipError_t c3p_launch_grad_scan(const GradArgs& A, bool global_scratch, hipStream_t st);  // adjoint state at segment ends
hipError_t c3p_launch_grad_bwd(const GradArgs& A, bool global_scratch, hipStream_t st);   // backward sweep, writes grad
