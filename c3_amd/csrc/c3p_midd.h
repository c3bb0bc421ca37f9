// Launcher interface of the mid-D MFMA chain kernel (c3p_midd.hip).
#pragma once
#include "c3p_common.h"

struct MidArgs {
  const double* tables;   // [nsamp][(1+K)][IMG+4]
  int tab_per_sample;
  const double* signals;  // [B,K,N]
  const cplx* mats;       // [B,N,Dm,Dm] (GIVEN mode)
  // C3P_MODE_EXPM: generators supplied per slice, X_n = coef hs[b,n] (see SmallArgs)
  const cplx* hs;
  long hs_bstride;
  const double* meta;
  double coef_r, coef_i;
  const double* fr_phase;
  int B, K, N, Dm;
  int S, Lmax;
  int mode, right_order;
  int no_t18;  // debug/tuning: force the Paterson-Stockmeyer plan
  int no_t18n;  // debug/tuning: Taylor T18 parameters also for (nearly) normal generators (c3p_common.h: c3p_t18_tab)
  int no_real;  // debug/tuning: keep real Hamiltonians on the complex path
  cplx* seg_out;
  cplx* dUs_out;
  // Lindblad chains in the Hermitian basis (c3p_regr.hip): real generator tables and one flag per table (1 = real); the real
  // kernel takes the samples whose tables are all real, the complex one (given the flags) skips exactly those
  const double* hb_tables;
  const int* hb_tabflag;
  // backward sweep in the Hermitian basis (c3p_regrg.hip): when set, the real kernel also stores the TRANSPOSED local prefix
  // Q_n^T = (E_n ... E_n0)^T of every slice of its segment, real row-major [B,N,Dm,Dm]
  double* hb_qT;
};

// Backward sweep of the control gradient (c3p_grad.hip for the method)
struct MidGradArgs {
  const double* tables;  // as MidArgs
  int tab_per_sample;
  const double* signals;  // [B,K,N]
  const cplx* Mb;         // [B,S,Dm,Dm] adjoint state at the end of each segment
  double* grad;           // [B,K,N]
  cplx* zout;             // [B,N,Dm,Dm] or null: Z_n, the cotangent of G_n = -i dt H_n
  int B, K, N, Dm, S, Lmax;
  // the real-Hamiltonian sweep (midd_grad_real_kernel) has taken the chains it can: this launch skips them
  int skip_real;
  // general (non-unitary) generators, e.g. Lindblad superoperators (midd_grad_general_kernel; method in c3p_grad.hip):
  const double* tables_h;  // tables of the conjugate-transposed generators G_k^H (MidPrepArgs.conjT)
  const cplx* pre;         // [B,S,Dm,Dm] prefix product at the START of each segment; Mb = LEFT adjoint at its end
  const cplx* dUs;         // [B,N,Dm,Dm] slice propagators of the forward pass
  cplx* pstore;            // [B,N,Dm,Dm] scratch: prefix product in front of every slice
  // supplied generators (X_n = coef hs[b,n], as MidArgs): no tables, no signals; the result is zout[b,n] = the cotangent of X_n
  const cplx* hs;
  long hs_bstride;
  const double* meta;
  double coef_r, coef_i;
};

struct MidPrepArgs {
  const cplx* h0;
  long h0_bstride;
  const cplx* hks;
  long hks_bstride;
  const cplx* clp;
  double dt;
  int K, Dh, Dm, lindblad;
  int conjT;  // tables of G^H instead of G (backward sweep of general generators)
  int rows, W;
  int tile_nig, tile_nj;  // != 0: emit tile-major images (big-D kernel) instead of rows x W
  double* tables;
};

bool c3p_midd_geometry(int Dm, int* nig, int* nj, int* w);
size_t c3p_midd_table_doubles(int Dm, int K);
size_t c3p_midd_lds_bytes(int Dm, int K, int Lmax);
// LDS bytes of the images of the backward sweeps (the larger of the general and the real-Hamiltonian kernel)
size_t c3p_midd_grad_image_bytes(int Dm);
hipError_t c3p_launch_midd_chain(const MidArgs& A, hipStream_t st);
hipError_t c3p_launch_midd_real(const MidArgs& A, hipStream_t st);  // real-Hamiltonian instance only (c3p_launch_midd_chain calls it)
hipError_t c3p_launch_midd_grad(const MidGradArgs& A, hipStream_t st);
// general generators (Lindblad superoperators 16 x 16, 25 x 25, 36 x 36); hipErrorInvalidValue for other dimensions
hipError_t c3p_launch_midd_grad_general(const MidGradArgs& A, hipStream_t st);
hipError_t c3p_launch_midd_prep(const MidPrepArgs& P, int nsamp, hipStream_t st);
