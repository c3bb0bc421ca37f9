// Lindblad chains of one qubit / qutrit in real arithmetic in the Hermitian basis (c3p_smallr.hip)
#pragma once
#include <hip/hip_runtime.h>

#include "c3p_common.h"
#include "c3p_midd.h"
#include "c3p_regd.h"

struct SmallRArgs {
  const double* tables;  // [nsamp][1 + K][NB NB 16 + 4] real generator tables (c3p_launch_smallr_prep)
  int tab_per_sample;    // tables differ per sample (then S % 4 == 0)
  const double* signals; // [B,K,N]
  int B, K, N, Dm;
  int S, Lmax;           // segments per sample, ceil(N / S)
  cplx* seg_out;         // [B,S,Dm,Dm] segment products, already in the reference's (complex) vectorisation
  cplx* dUs_out;         // [B,N,Dm,Dm] slice propagators (complex vectorisation) or null
  double* seg_real;      // [B,S,Dm,Dm] the same segment products, REAL (Hermitian basis), or null: input of the real backward sweep
  double* dus_real;      // [B,N,Dm,Dm] the LOCAL prefix of every slice (product of the slices of its segment in front of it), REAL, or null
  int no_t18n;           // debug/tuning: Taylor T18 parameters also for nearly skew-symmetric generators (c3p_common.h: c3p_t18_tab)
};

// Backward sweep in the Hermitian basis (smallr_grad_kernel): the general-generator sweep of smalld_grad_general_kernel in real
// arithmetic -- one pair evaluation (value + derivative of T18) per slice at X_n^T, its direction from the adjoint (with the
// prefix at the segment start folded in) and the local prefix of the slice the forward kernel stored.
struct SmallRGradArgs {
  const double* tables;    // G' tables (inner products)
  const double* tables_t;  // G'^T tables (the matrix that is exponentiated)
  int tab_per_sample;
  const double* signals;   // [B,K,N]
  const double* pre;       // [B,S,Dm,Dm] real prefix in front of every segment
  const double* suf;       // [B,S,Dm,Dm] real left adjoint behind every segment
  const double* dus;       // [B,N,Dm,Dm] real local prefixes (SmallRArgs.dus_real)
  double* grad;            // [B,K,N]
  int B, K, N, Dm, S, Lmax;
  int no_t18n;
};

bool c3p_smallr_supported(int Dh, int Dm, int K);
size_t c3p_smallr_table_doubles(int Dm, int K);
size_t c3p_smallr_lds_bytes(int Dm, int K, int Lmax);
// arguments as c3p_launch_regr_prep (lindblad generators of h0 / hks / clp); tabflag [nsamp][1 + K]: 1 = real in the basis
hipError_t c3p_launch_smallr_prep(const RegdPrepArgs& P, int nsamp, double* tables, int* tabflag, hipStream_t st, int transpose = 0);
// both table sets (G' and G'^T) from one launch
hipError_t c3p_launch_smallr_prep_pair(const RegdPrepArgs& P, int nsamp, double* tables, double* tables_t, int* tabflag, hipStream_t st);
size_t c3p_smallr_grad_lds_bytes(int Dm, int K, int Lmax);
hipError_t c3p_launch_smallr_grad(const SmallRGradArgs& A, hipStream_t st);
// pre[j] = S_{j-1} ... S_0 (pre[0] = 1), suf[j] = S_{j+1}^T ... S_{S-1}^T U_bar' from the REAL segment products seg [B,S,Dm,Dm]
hipError_t c3p_launch_smallr_scan(const double* seg, const double* ubar, int B, int S, int Dm, double* pre, double* suf, hipStream_t st);
hipError_t c3p_launch_smallr_chain(const SmallRArgs& A, hipStream_t st);
