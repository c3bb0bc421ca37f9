// Internal launcher interface between the C-ABI layer (c3p_api.hip) and the kernel files.
#pragma once
#include "c3p_common.h"

// ---- tuning / diagnostic options -------------------------------------------------------------------------------------
// ONE table for the whole library (it replaces the getenv() calls that used to sit in the dispatch code): set through the
// C ABI (c3p_set_option(name, value), include/c3prop.h) or, ONCE at the first use, from environment variables of the same
// names in upper case with the prefix C3P_ (C3P_NO_REGD=1 python ...).  Values are integers; -1 = not set.  Reads are
// relaxed atomic loads: safe from any thread, against c3p_set_option from another one.
#define C3P_OPTION_LIST(X)                                                                                             \
  X(no_regd)            /* 41 <= Dm <= 81: arena kernel (c3p_bigd.hip) instead of the register-resident ones */       \
  X(regd_pad)           /* 0: no zero-padded classes; 2: pad everything in 41..81; unset: the measured default */      \
  X(no_hermitian_basis) /* Lindblad chains at D = 7..9 on the complex kernel (c3p_regd.hip) instead of c3p_regr.hip */ \
  X(regr_waves)         /* c3p_regr.hip, -DC3P_REGR_VARIANTS builds only: 8 = two waves per SIMD, 2 = two lean workgroups per CU */ \
  X(regr_rolled)        /* c3p_regr.hip, -DC3P_REGR_VARIANTS builds only: rolled product loop */                                \
  X(no_tiled)           /* Dm >= 93: generic kernel instead of the tiled path */                                       \
  X(no_split81)         /* small D, real path at D = 5, 9: padded tiles instead of the core + border form */             \
  X(no_smallr)          /* Lindblad D = 2, 3 with C3P_HERMITIAN_H: the complex small-D kernels, not the real ones */       \
  X(no_mw)              /* small D: one-wave workgroups + ticket instead of workgroup-per-sample */                    \
  X(mw_skew)            /* per mille of a pair's slices given to the older wave */                                     \
  X(no_fuse)            /* small D: separate combine launch */                                                         \
  X(prep_kernel)        /* small D: separate table kernel */                                                           \
  X(smalld_segments)    /* segments per sample, small D */                                                             \
  X(segments)           /* segments per sample, mid D */                                                               \
  X(no_many_rounds)     /* mid D: round-1 segment rule */                                                              \
  X(no_t18)             /* Paterson-Stockmeyer plan only */                                                            \
  X(no_t18n)            /* 1: keep the Taylor T18 parameters (radius 1.13) where the schemes for normal generators apply; 2: keep T18N, drop only the four-product scheme */ \
  X(no_real)            /* real Hamiltonians on the complex instances */                                               \
  X(no_real_grad)       /* real Hamiltonians on the general backward sweeps */                                         \
  X(grad_target)        /* chains per launch of the VALU backward sweep */                                             \
  X(grad_slots)         /* resident wave slots assumed by the backward sweeps' segment rule */                         \
  X(grad_fwd_seg2)      /* small-D gradients: 0 = forward segment products on the backward sweep's S segments, not 2 S */      \
  X(grad_chunk)         /* samples per chunk of the general-generator sweeps */                                        \
  X(regr_grad_degree)   /* Hermitian-basis Lindblad sweep: force the Taylor degree (8, 12, 16, 20) */                  \
  X(regr_grad_d6)       /* Lindblad gradient at D = 6 zero padded on the Hermitian-basis sweep (A/B: slower than mid-D) */     \
  X(tiled_grad)         /* tiled backward sweep wherever it applies */                                                 \
  X(valu_grad)          /* VALU backward sweeps instead of the matrix-core ones */                                     \
  X(tiled_graph)        /* tiled sweep: replay slices as hipGraphs */                                                  \
  X(tiled_no_batch)     /* tiled sweep: one launch per product */                                                      \
  X(tiled_tile32)       /* tiled sweep: 0 / 1 forces the 64 x 64 / 32 x 32 output tiles */                             \
  X(ode_wg)             /* ODE: workgroup-per-sample kernel of round 1 */                                              \
  X(ode_prop_rows)      /* rk4_unitary at 17 <= D <= 48 on the lane-row column kernel */                               \
  X(ode_rho_general)    /* matrix-core ODE kernel: two products per commutator for every input */                      \
  X(ode_no_seg)         /* ODE: no time segments for small final-state batches */                                     \
  X(ode_no_split)       /* ODE: rho-valued states with K > 4: lane-row kernel whatever the operators are (no device-side split) */ \
  X(ode_lind_wg)        /* ODE: Lindblad steps at 33 <= D <= 48 on the workgroup kernel of round 1 (A/B: not the matrix-core one) */

enum C3pOption {
#define C3P_OPT_ENUM(n) C3P_OPT_##n,
  C3P_OPTION_LIST(C3P_OPT_ENUM)
#undef C3P_OPT_ENUM
      C3P_OPT_COUNT
};
long c3p_opt(C3pOption o);                                     // -1 = not set
inline bool c3p_opt_on(C3pOption o) { return c3p_opt(o) > 0; }  // a switch: set to a positive value

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
