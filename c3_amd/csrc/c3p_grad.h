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
  // general (non-unitary) generators, e.g. Lindblad superoperators: h0 / hks hold G_0 / G_k with X_n = dt (G_0 + sum c_k G_k)
  // (no factor -i), the slices have no cheap inverse, so the sweep keeps the prefix product of every slice in memory
  int general;
  cplx* pre;     // [B,S,D,D] prefix product at the START of each segment        (general only; Mb then holds the LEFT adjoint
  cplx* pstore;  // [B,N,D,D] prefix product in front of every slice              A = S^H FR^H Ubar at the end of each segment)
  // fused goal (c3p_pwc_unitary_goal_vjp): when goal_rows is set the scan kernel, which forms the total product anyway,
  // evaluates the gate infidelity of U = FR P and takes ITS cotangent as Ubar (Ubar above is then not read):
  //   s = tr(G^+ U[rows, rows]),  infid = 1 - |s / L|^2 (kind 0) or 1 - (|s|^2 / L + 1) / (L + 1) (kind 1),  Ubar = c s G on (rows x rows)
  const int* goal_rows;    // [L] computational rows (fidelities.py:154-184 via tf_project_to_comp), L <= C3P_GOAL_LMAX
  const cplx* goal_ideal;  // [L,L]
  int goal_L, goal_kind;
  double* goal_infid;   // [B]
  double* goal_gphase;  // [B,D] or null: d infid / d fr_phase
  cplx* goal_U;         // [B,D,D] or null: the propagators themselves
};
#define C3P_GOAL_LMAX 64

#define C3P_GRAD_NMAT 19  // matrices a backward workgroup keeps (LDS or scratch)
#define C3P_GRAD_NMAT_GENERAL 20

int c3p_grad_threads(int D);
size_t c3p_grad_lds_bytes(int D);  // LDS variant footprint; > 150 KB => use the GLOBAL variant
hipError_t c3p_launch_grad_seg(const GradArgs& A, bool global_scratch, hipStream_t st);   // segment products
hipError_t c3p_launch_grad_scan(const GradArgs& A, bool global_scratch, hipStream_t st);  // adjoint state at segment ends
hipError_t c3p_launch_grad_bwd(const GradArgs& A, bool global_scratch, hipStream_t st);   // backward sweep, writes grad
// general generators (A.general = 1): same three passes without the unitarity shortcut
size_t c3p_grad_lds_bytes_general(int D);
hipError_t c3p_launch_grad_scan_general(const GradArgs& A, bool global_scratch, hipStream_t st);
hipError_t c3p_launch_grad_bwd_general(const GradArgs& A, bool global_scratch, hipStream_t st);
// dense Lindblad generators [nb][(K+1)][D^2 x D^2]: G_0 = -i (spre(h0) - spost(h0)) + clp, G_k = -i (spre(hk) - spost(hk))
// (propagation.py:551-582); nb = B when an operator stride is non-zero, else 1
// per-slice variant: out[b,n] = -i (spre(hs[b,n]) - spost(hs[b,n])) + clp for every slice Hamiltonian (branch B + lindbladian)
hipError_t c3p_launch_lind_slice_generators(const cplx* hs, long hs_bstride, const cplx* clp, int B, int N, int D, cplx* out, hipStream_t st);
hipError_t c3p_launch_lind_generators(const cplx* h0, long h0_bstride, const cplx* hks, long hks_bstride, const cplx* clp, int nb,
                                      int K, int D, cplx* out, hipStream_t st);
