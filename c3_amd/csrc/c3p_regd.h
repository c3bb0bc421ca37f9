// Launcher interface of the register-resident MFMA chain kernel (c3p_regd.hip): matrix dimensions
// Dm = 16 n + 1, n = 3, 4, 5 (49, 65, 81) -- in particular the 81 x 81 Lindblad superoperator of two
// qutrits (BASELINE cfg4) and the 49 x 49 one of D = 7.
#pragma once
#include "c3p_common.h"
#include "c3p_midd.h"

#define C3P_REGD_MAX_WGS 256  // one workgroup per CU

struct RegdPrepArgs {
  const cplx* h0;
  long h0_bstride;
  const cplx* hks;
  long hks_bstride;
  const cplx* clp;
  double dt;
  int K, Dh, Dm, lindblad;
  double* tables;
};

bool c3p_regd_supported(int Dm);
// the kernel class (49, 65 or 81) a dimension runs in, zero padded when smaller
__host__ __device__ inline int c3p_regd_class(int Dm) { return Dm <= 49 ? 49 : (Dm <= 65 ? 65 : 81); }
size_t c3p_regd_table_doubles(int Dm, int K);   // per sample: (1 + K) generator tables
size_t c3p_regd_arena_bytes(int Dm);            // for a whole launch (one arena per workgroup)
hipError_t c3p_launch_regd_prep(const RegdPrepArgs& P, int nsamp, hipStream_t st);
// MidArgs as for the mid-D / big-D kernels (tables in the layout c3p_launch_regd_prep writes); seg_out gets the
// B x S segment products without frame-rotation phases
hipError_t c3p_launch_regd_chain(const MidArgs& A, void* arena, hipStream_t st);
