// Tiled large-matrix propagator path (c3p_tiled.hip): any matrix dimension, matrices in HBM, one batched complex GEMM
// launch per product.  Used above the register/LDS-resident kernels: Dm >= 93 (e.g. the 729 x 729 Lindblad
// superoperator of three qutrits, test/test_tunable_coupler.py:406-418 in the reference) and the supplied-generator /
// Lindblad-with-per-slice-Hamiltonian cases at Dm >= 41.
#pragma once
#include <string>

#include "c3p_common.h"

struct TiledArgs {
  int lindblad;
  int per_slice;           // h0 = per-slice Hamiltonians [N,D,D] (h0_bstride 0) or [B,N,D,D]; no hks / signals
  const cplx* h0;
  long h0_bstride;         // complex elements between samples (0 = shared)
  const cplx* hks;         // [K,D,D] or [B,K,D,D]
  long hks_bstride;
  const double* signals;   // [B,K,N]
  const cplx* clp;         // [Dm*Dm] Lindblad dissipator or null
  double dt;
  int B, K, N, D, Dm;
  const double* fr_phase;  // [B,Dm] or null
  cplx* U_out;             // [B,Dm,Dm]
  cplx* dUs_out;           // [B,N,Dm,Dm] or null
};

// samples per pass such that the workspace stays below `budget_bytes`
int c3p_tiled_chunk(int Dm, int K, int B, bool per_sample_tables, size_t budget_bytes);
size_t c3p_tiled_ws_bytes(int Dm, int K, int Bc, bool per_sample_tables);
// Runs the whole propagation (synchronises the stream once, to read the norm bound that fixes the number of
// squarings).  Returns 0 or -1 with `err` set.
int c3p_tiled_run(const TiledArgs& A, void* ws, int Bc, hipStream_t st, std::string& err);

// Vector-Jacobian product of the same path with respect to the control samples (table mode): forward sweep storing the
// adjoints of the partial products in HBM, backward sweep with one pair evaluation of T18 per slice (value + Frechet
// derivative, 15 + 3 s batched GEMMs).  No unitarity assumed: unitary AND Lindblad generators, any matrix dimension.
//   U_bar [B,Dm,Dm]: d loss = Re sum conj(U_bar) dU;   grad f64 [B,K,N]
int c3p_tiled_vjp_chunk(int Dm, int K, int N, int B, bool per_sample_tables, size_t budget_bytes);
size_t c3p_tiled_vjp_ws_bytes(int Dm, int K, int N, int Bc, bool per_sample_tables);
//   per_slice (branch B): h0 = per-slice Hamiltonians, no signals; the result is zout c128 [B,N,Dm,Dm], the cotangent of every
//   slice generator (grad unused)
int c3p_tiled_vjp_run(const TiledArgs& A, const cplx* U_bar, double* grad, cplx* zout, void* ws, int Bc, hipStream_t st, std::string& err);
