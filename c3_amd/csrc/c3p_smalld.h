// Launcher interface of the small-D MFMA chain kernel (c3p_smalld.hip).
#pragma once
#include "c3p_common.h"

#define C3P_SMALLD_MAX 12

struct SmallArgs {
  const double* tables;    // [nsamp][(1+K)][MAT+4] images from smalld_prep (TABLE mode)
  // inline_tables (unitary mode): every wave builds the tables of its sample in LDS from h0 / hks itself
  int inline_tables;
  const cplx* h0;
  long h0_bstride;
  const cplx* hks;
  long hks_bstride;
  double dt;
  int tab_per_sample;      // tables differ per sample (then S % 4 == 0)
  const double* signals;   // [B,K,N]
  const cplx* mats;        // [B,N,Dm,Dm] (GIVEN mode)
  // C3P_MODE_EXPM (generators supplied per slice: X_n = coef hs[b,n], shifted by meta's mu_n)
  const cplx* hs;          // [B,N,Dm,Dm] or [N,Dm,Dm] (hs_bstride 0)
  long hs_bstride;         // complex elements between samples
  const double* meta;      // [B*N][4] = {Re mu, Im mu, ||X - mu||_1, ||X - mu||_inf} from c3p_launch_hmeta
  double coef_r, coef_i;   // -i dt -> (0, -dt); 1 for c3p_expm
  const double* fr_phase;  // [B,Dm] or null
  int B, K, N, Dm;
  int S;     // segments per sample (balanced: segment s covers [s*N/S, (s+1)*N/S))
  int Lmax;  // ceil(N/S)
  int mode;  // C3P_MODE_UNITARY / C3P_MODE_LINDBLAD (tables) or C3P_MODE_GIVEN
  int right_order;
  cplx* seg_out;  // [B,S,Dm,Dm]   (fused combine: [B,S/4,Dm,Dm] wave partials)
  cplx* dUs_out;  // [B,N,Dm,Dm] or null
  // Fused ordered combine (table modes, S % 4 == 0): every wave folds its four consecutive
  // segments in registers, publishes one partial product, and the last wave to arrive for a
  // sample (per-sample counter, agent-scope release/acquire) folds the S/4 partials into U.
  int no_t18n;  // debug/tuning: Taylor T18 parameters also for Hermitian Hamiltonians (c3p_common.h: c3p_t18_tab)
  int fuse;
  // MW mode with two waves per SIMD: slices of the segments of the first half of a sample's chains (the OLDER waves of
  // their SIMDs, which the arbiter serves first); 0 = equal segments.  Set by the launcher.
  int seg_long;
  int seg_long_c;  // the same for samples that take the COMPLEX loop (its optimum differs: 640 against 700 per mille at D = 9); 0 = seg_long
  int* counters;   // [B], zeroed by the prep kernel of the same call
  cplx* final_out;  // [B,Dm,Dm]
};

// Backward sweep of the control gradient (c3p_grad.hip for the method)
struct SmallGradArgs {
  const double* tables;  // as SmallArgs
  int tab_per_sample;
  const double* signals;  // [B,K,N]
  const cplx* Mb;         // [B,S,Dm,Dm] adjoint state at the end of each segment
  double* grad;           // [B,K,N]
  cplx* zout;             // [B,N,Dm,Dm] or null: Z_n, the cotangent of G_n = -i dt H_n
  int B, K, N, Dm, S, Lmax;
  // the real-Hamiltonian sweep (smalld_grad_real_kernel) has taken the samples it can: this launch skips them
  int skip_real;
  // general (non-unitary) generators, e.g. Lindblad superoperators (smalld_grad_general_kernel; method in c3p_grad.hip):
  const double* tables_h;  // tables of the conjugate-transposed generators G_k^H (PrepArgs.conjT)
  const cplx* pre;         // [B,S,Dm,Dm] prefix product at the START of each segment; Mb = LEFT adjoint at its end
  const cplx* dUs;         // [B,N,Dm,Dm] slice propagators of the forward pass
  cplx* pstore;            // [B,N,Dm,Dm] scratch: prefix product in front of every slice
  // supplied generators (X_n = coef hs[b,n], as SmallArgs): no tables, no signals; the result is zout[b,n] = the cotangent of X_n
  const cplx* hs;
  long hs_bstride;
  const double* meta;
  double coef_r, coef_i;
};

struct PrepArgs {
  const cplx* h0;
  long h0_bstride;
  const cplx* hks;
  long hks_bstride;
  const cplx* clp;  // Lindblad dissipator [Dm*Dm] or null
  double dt;
  int K, Dh, lindblad;
  int conjT;  // tables of G^H instead of G (backward sweep of general generators)
  double* tables;
  int* counters;  // zeroed here (one launch fewer than a memset node); may be null
  int ncounters;
};

int c3p_smalld_mat_doubles(int Dm);
int c3p_smalld_img_doubles(int Dm);  // per-chain LDS image buffer of the forward kernel
size_t c3p_smalld_table_doubles(int Dm, int K);
bool c3p_smalld_supported(int Dm);
hipError_t c3p_launch_smalld_chain(const SmallArgs& A, hipStream_t st);
hipError_t c3p_launch_smalld_grad(const SmallGradArgs& A, hipStream_t st);
hipError_t c3p_launch_smalld_grad_general(const SmallGradArgs& A, hipStream_t st);  // general generators (Lindblad)
hipError_t c3p_launch_smalld_grad_real(const SmallGradArgs& A, hipStream_t st);  // real-Hamiltonian sweep only (c3p_launch_smalld_grad calls it)
hipError_t c3p_launch_smalld_prep(const PrepArgs& P, int Dm, int nsamp, hipStream_t st);
