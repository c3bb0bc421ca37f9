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
// the change of basis: index a = i D + j of the row-major vectorisation, partner a' = j D + i.  Rows of T:
//   i == j: e_a;   i < j: (e_a + e_a') / sqrt 2;   i > j: i (e_a - e_a') / sqrt 2      [a' is the (j, i) element, j < i]
// c3p_hb_row: the (at most two) non-zeros of ROW a of T;  c3p_hb_col: the non-zeros of COLUMN a of T.
__device__ __forceinline__ int c3p_hb_row(int a, int D, int (&idx)[2], cplx (&t)[2]) {
  const double r = 0.70710678118654752440;
  const int i = a / D, j = a - i * D, ap = j * D + i;
  if (i == j) {
    idx[0] = a, t[0] = cmake(1.0, 0.0);
    return 1;
  }
  if (i < j) {
    idx[0] = a, t[0] = cmake(r, 0.0);
    idx[1] = ap, t[1] = cmake(r, 0.0);
  } else {
    idx[0] = ap, t[0] = cmake(0.0, -r);
    idx[1] = a, t[1] = cmake(0.0, r);
  }
  return 2;
}
__device__ __forceinline__ int c3p_hb_col(int a, int D, int (&idx)[2], cplx (&t)[2]) {
  const double r = 0.70710678118654752440;
  const int i = a / D, j = a - i * D, ap = j * D + i;
  if (i == j) {
    idx[0] = a, t[0] = cmake(1.0, 0.0);
    return 1;
  }
  if (i < j) {  // T[a, a] = r (row a is the symmetric combination), T[a', a] = -i r
    idx[0] = a, t[0] = cmake(r, 0.0);
    idx[1] = ap, t[1] = cmake(0.0, -r);
  } else {  // T[a', a] = r (row a' is the symmetric combination), T[a, a] = +i r
    idx[0] = ap, t[0] = cmake(r, 0.0);
    idx[1] = a, t[1] = cmake(0.0, r);
  }
  return 2;
}
hipError_t c3p_launch_hb_to_complex(cplx* mats, long nmat, int mats_per_sample, const int* tabflag, int tab_per_sample, int K,
                                    int Dh, hipStream_t st);

// ---- backward sweep of the Lindblad chains in the Hermitian basis (c3p_regrg.hip) ----
// c3p_launch_regr_prep with transpose = 1 tabulates G'^T (the pair evaluation of the backward sweep runs at X_n^T)
hipError_t c3p_launch_regr_prep_t(const RegdPrepArgs& P, int nsamp, double* tables, int* tabflag, int transpose, hipStream_t st);
struct RegrGradArgs {
  const double* tables;    // real tables of G' (c3p_launch_regr_prep): the inner products <X_bar, G_k>
  const double* tables_t;  // real tables of G'^T: X_n^T is assembled from these
  int tab_per_sample;
  const double* signals;   // [B,K,N]
  const double* qT;        // [B,N,Dm,Dm] transposed local prefixes (MidArgs.hb_qT)
  const double* lam;       // [B,S,Dm,Dm] left adjoint at the END of every segment with the prefix at its start folded in
  const double* tau;       // [B] <U_bar', U'>: the share of the trace shifts
  double* grad;            // [B,K,N]
  double* arena;           // C3P_REGD_MAX_WGS tile sets
  int B, K, N, Dm, S;
  int degree;              // 0: chosen per segment; 8, 12, 16, 20: forced (A/B)
};
size_t c3p_regr_grad_arena_bytes(int Dm);
hipError_t c3p_launch_regr_grad(const RegrGradArgs& A, hipStream_t st);
// U_bar'[b] = Re(T diag(e^{-i phi_b}) U_bar[b] T^+): the cotangent of the real chain product in the Hermitian basis
hipError_t c3p_launch_hb_ubar(const cplx* Ubar, const double* fr_phase, int B, int Dh, double* out, hipStream_t st);
// segment scan, real Dm x Dm matrices: seg_slots = the complex slots of the real chain kernel (real matrix in the second
// half of each); pre / suf / lam [B,S,Dm,Dm], tau [B]
hipError_t c3p_launch_regr_scan(const cplx* seg_slots, const double* ubar, int B, int S, int Dm, double* pre, double* suf, double* lam,
                                double* tau, hipStream_t st);
