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
};

bool c3p_smallr_supported(int Dh, int Dm, int K);
size_t c3p_smallr_table_doubles(int Dm, int K);
size_t c3p_smallr_lds_bytes(int Dm, int K, int Lmax);
// arguments as c3p_launch_regr_prep (lindblad generators of h0 / hks / clp); tabflag [nsamp][1 + K]: 1 = real in the basis
hipError_t c3p_launch_smallr_prep(const RegdPrepArgs& P, int nsamp, double* tables, int* tabflag, hipStream_t st);
hipError_t c3p_launch_smallr_chain(const SmallRArgs& A, hipStream_t st);
