// Launcher interface of the ODE state-solver kernel (c3p_ode.hip).
#pragma once
#include "c3p_common.h"
#include "c3p_kernels.h"

#define C3P_STEP_SCHRODINGER_ID 0
#define C3P_STEP_VON_NEUMANN_ID 1
#define C3P_STEP_LINDBLAD_ID 2
#define C3P_STEP_PROPAGATOR_ID 3  // Y <- -i dt H Y on a [D,D] matrix of column states (rk4_unitary)

struct OdeArgs {
  const cplx* h0;         // [D,D]
  const cplx* hks;        // [K,D,D]
  const double* signals;  // [B,K,N]
  const cplx* col_ops;    // [C,D,D]
  const cplx* init;       // [B?,D,M]
  long init_bstride;
  double dt;
  int B, K, N, D, M, C;
  int solver, step, want_all;
  int n_steps;          // RK steps to take
  int u_stride;         // stage position in signal-sample units: u = (n + node) * u_stride
  const cplx* hs;       // optional per-sample-index Hamiltonians [B?,N,D,D] (then h0/hks unused)
  long hs_bstride;
  int hs_lerp;          // hs: 0 = the sample nearest to the stage position (branch B), 1 = linear interpolation of two samples
  int reset_each_step;  // state <- init after every step (per-step propagators)
  int transpose_out;    // store states transposed (gen_du_rk4 stacks propagated vectors as rows)
  cplx* states;
  cplx* scratch;
  long scratch_stride;
  // time segments (lane-row vector kernel, final state / propagator only): with seg_count > 0 every (sample, column) is
  // integrated over seg_count step ranges of seg_len steps from the IDENTITY, and states receives the segment maps
  // [B, seg_count, D, M] (their ordered product is the step map of the whole interval: the equations are linear)
  int seg_count, seg_len;
  // trajectory from segment start states (lane-row vector kernel, M = 1, want_all): row (b, seg) starts from
  // init[(b * seg_count + seg) * init_bstride], integrates its own step range and writes states[b, n] of that range
  int seg_traj;
  int rho_general;  // matrix-core rho kernel: do not use the Hermitian shortcut (set by its launcher from C3P_ODE_RHO_GENERAL)
  // Dispatch by what the DEVICE sees (the host cannot look into device operators): rho-valued states with more than four control
  // lines and COMPLEX operators run faster on the workgroup kernel while the batch leaves SIMDs idle on the lane rows.  Both
  // kernels are launched; complex_to_wg = 1 makes the lane-row kernel's complex instance leave at once when the operators are
  // complex (the real instance keeps real ones), wg_if_complex = 1 makes the workgroup kernel leave at once when they are real.
  int complex_to_wg, wg_if_complex;
};

// padded copies of the collapse operators for the lane-row rho kernel (c3p_ode_row.hip), written by its prep kernel
struct OdeRowAux {
  const cplx* colpad;  // [C][DP][DP]
  const cplx* coladj;  // [C][DP][DP]: [m][j][c] = conj(C_m[c][j])
  const cplx* gpad;    // [DP][DP]: sum_m C_m^+ C_m
};

size_t c3p_ode_elems(int D, int M, int C);
hipError_t c3p_launch_ode(const OdeArgs& A, bool global_scratch, hipStream_t st);

// Lane-row kernels (c3p_ode_row.hip): D <= 16, K <= 4, operators + signals (no supplied per-sample Hamiltonians)
bool c3p_ode_row_supported(const OdeArgs& A);
size_t c3p_ode_row_aux_bytes(int D, int C);
hipError_t c3p_launch_ode_row(const OdeArgs& A, void* aux, hipStream_t st);
hipError_t c3p_launch_ode_assemble_hs(const OdeArgs& A, cplx* out, hipStream_t st);  // out [B,N,D,D] = h0 + sum_k c_k hk

// Lane-row vector-state kernel for 17 <= D <= 48 (c3p_ode_rowq.hip): operators in LDS, H(t) advanced along the linear
// pieces of the control amplitudes
bool c3p_ode_rowq_supported(const OdeArgs& A);
hipError_t c3p_launch_ode_rowq(const OdeArgs& A, hipStream_t st);
// Matrix-core rho-valued solver for 17 <= D <= 48 (c3p_ode_rhoq.hip): von Neumann (D <= 48) and Lindblad (D <= 32) steps,
// one workgroup per sample, every matrix as 16 x 16 register tiles, products on v_mfma_f64_16x16x4_f64
bool c3p_ode_rhoq_supported(const OdeArgs& A);
hipError_t c3p_launch_ode_rhoq(const OdeArgs& A, hipStream_t st);
// segmented integration of small batches (c3p_ode_row.hip): segment count (0 = off), and psi_out = U psi0
int c3p_ode_row_segments(const OdeArgs& A);
hipError_t c3p_launch_ode_apply(const cplx* U, const cplx* init, long init_bstride, cplx* out, int B, int D, hipStream_t st);
hipError_t c3p_launch_ode_identity(cplx* out, int D, hipStream_t st);  // [D,D] identity on the device (no host round trip)
// trajectories of small batches (want_all) in time segments: segment count (0 = off); start states of every segment from the
// segment maps: starts[b, s] = maps[b, s-1] ... maps[b, 0] init[b]
int c3p_ode_row_traj_segments(const OdeArgs& A);
hipError_t c3p_launch_ode_starts(const cplx* maps, const cplx* init, long init_bstride, cplx* starts, int B, int S, int D, hipStream_t st);
