// Launcher interface of the signal-synthesis kernels (c3p_signal.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/c3prop.h"

struct SynthArgs {
  const double* env;      // [B,K,E,C3P_ENV_NPAR]
  const int* shape;       // [K,E], < 0 = slot unused
  const double* carrier;  // [B,K,2]: LO angular frequency, V->Hz factor
  double t_start, t_end, awg_res, sim_res;
  int B, K, E, Na, N;
  double* iq;       // [B,K,2,Na] AWG-resolution inphase / quadrature
  double* signals;  // [B,K,N]
};

hipError_t c3p_launch_synth(const SynthArgs& A, hipStream_t st);
// d loss/d signals [B,K,N] -> d loss/d env_params [B,K,E,NPAR] (amp, xy_angle, freq_offset, delta slots) and
// d loss/d carrier [B,K,2]; giq [B,K,2,Na] and gcar_part [B,K,Na,2] are scratch, A.iq is recomputed.
hipError_t c3p_launch_synth_vjp(const SynthArgs& A, const double* gsig, double* giq, double* gcar_part, double* genv,
                                double* gcar, hipStream_t st);
