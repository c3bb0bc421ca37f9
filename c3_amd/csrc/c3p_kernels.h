// Internal launcher interface between the C-ABI layer (c3p_api.hip) and the kernel files.
#pragma once
#include "c3p_common.h"

#define C3P_MODE_UNITARY 0   // assemble -i dt H, exponentiate, chain
#define C3P_MODE_LINDBLAD 1  // assemble dt L(H), exponentiate, chain
#define C3P_MODE_EXPM 2      // exponentiate supplied matrices (no chain when N == 1)
#define C3P_MODE_GIVEN 3     // ordered product of supplied matrices

struct ChainArgs {
  const cplx* h0;
  long h0_bstride;  // elements between samples (0 = shared)
  long h0_nstride;  // elements between slices (0 = slice independent)
  const cplx* hks;
  long hks_bstride;
  const double* signals;   // [B,K,N]
  const cplx* clp;         // [Dm*Dm] Lindblad dissipator
  const cplx* mats;        // [B,N,Dm,Dm] for MODE_EXPM / MODE_GIVEN
  const double* fr_phase;  // [B,Dm] or null: row phases applied when writing seg_out
  double dt;
  int B, K, N, D, Dm, ld;
  int S, seg_len;
  int mode;
  int right_order;
  cplx* seg_out;  // [B,S,Dm,Dm]
  cplx* dUs_out;  // [B,N,Dm,Dm] or null
  cplx* scratch;  // global scratch (GLOBAL variant), scratch_stride elements per workgroup
  long scratch_stride;
};

size_t c3p_generic_lds_bytes(int Dm);
int c3p_generic_threads(int Dm);
hipError_t c3p_launch_chain_generic(const ChainArgs& A, bool global_scratch, hipStream_t st);
hipError_t c3p_launch_clp(const cplx* col, int C, int D, cplx* clp, hipStream_t st);
hipError_t c3p_launch_kron(const cplx* A, const cplx* Bm, int n, int Da, int Db, int which, cplx* out,
                           hipStream_t st);
int c3p_infid_blocks(int B);
hipError_t c3p_launch_infid(const cplx* U, int B, int D, const int* rows, int L, const cplx* ideal, int kind, double* infid,
                            double* partial, double* sum_out, hipStream_t st);
hipError_t c3p_launch_overlap(const cplx* U, int B, int D, const int* rows, int L, const cplx* ideal,
                              cplx* out, hipStream_t st);
// pre-pass of the supplied-generator modes: meta[m] = {Re mu, Im mu, ||coef H_m - mu||_1, 0}, m = b * N + n
hipError_t c3p_launch_hmeta(const cplx* hs, long bstride, long nmat, int N, int D, double cr, double ci, double* meta,
                            hipStream_t st);
