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

// ---- Lindblad chains in the Hermitian basis, real arithmetic (c3p_regr.hip) ----
// all (1 + K) generator tables of a sample real in the Hermitian basis (flags written by c3p_launch_regr_prep)?
__device__ __forceinline__ bool c3p_hb_sample_is_real(const int* tabflag, int sample_or_0, int K) {
  bool r = true;
  for (int k = 0; k <= K; ++k) r = r && (tabflag[sample_or_0 * (1 + K) + k] != 0);
  return r;
}
bool c3p_regr_supported(int Dh, int Dm);       // Dm = Dh^2 in a class of the register-resident kernels (Dh = 7, 8, 9)
size_t c3p_regr_table_doubles(int Dm, int K);  // per sample: (1 + K) real generator tables
// same arguments as c3p_launch_regd_prep (lindblad = 1); tabflag: [nsamp][1 + K]
hipError_t c3p_launch_regr_prep(const RegdPrepArgs& P, int nsamp, double* tables, int* tabflag, hipStream_t st);
// MidArgs with hb_tables / hb_tabflag set; the REAL segment products (and slice propagators) are written to the second
// half of their complex slots in seg_out (dUs_out): c3p_launch_hb_to_complex turns them into the complex matrices in place
size_t c3p_regr_arena_bytes(int Dm);  // scratch of the two-workgroups-per-CU form
hipError_t c3p_launch_regr_chain(const MidArgs& A, void* arena, hipStream_t st);
hipError_t c3p_launch_hb_to_complex(cplx* mats, long nmat, int mats_per_sample, const int* tabflag, int tab_per_sample, int K,
                                    int Dh, hipStream_t st);
