// Launcher interface of the ODE state-solver kernel (c3p_ode.hip).
#pragma once
#include "c3p_common.h"

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
  int reset_each_step;  // state <- init after every step (per-step propagators)
  int transpose_out;    // store states transposed (gen_du_rk4 stacks propagated vectors as rows)
  cplx* states;
  cplx* scratch;
  long scratch_stride;
};

size_t c3p_ode_elems(int D, int M, int C);
hipError_t c3p_launch_ode(const OdeArgs& A, bool global_scratch, hipStream_t st);
