// Launcher interface of the ODE state-solver kernel (c3p_ode.hip).
#pragma once
#include "c3p_common.h"

#define C3P_STEP_SCHRODINGER_ID 0
#define C3P_STEP_VON_NEUMANN_ID 1
#define C3P_STEP_LINDBLAD_ID 2

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
  cplx* states;
  cplx* scratch;
  long scratch_stride;
};

size_t c3p_ode_elems(int D, int M, int C);
hipError_t c3p_launch_ode(const OdeArgs& A, bool global_scratch, hipStream_t st);
