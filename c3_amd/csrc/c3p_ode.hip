// ODE state solver: explicit Runge-Kutta integration of psi (Schroedinger) or rho
// (von Neumann / Lindblad) with the Hamiltonian rebuilt at every stage from linearly
// interpolated control signals -- `Hs` is never materialised.
//
// Stands in for ode_solver / ode_solver_final_state (propagation.py:687-752), the
// tableaux rk4/rk38/rk5/tsit5 (:755-883), the step functions (:886-904),
// Model.Hs_of_t (model.py:641-697) and interpolate_signal (tf_utils.py:521-559).
//
// One workgroup integrates one sample; the N steps are inherently sequential.
#include "c3p_common.h"
#include "c3p_kernels.h"
#include "c3p_ode.h"
#include "c3p_ode_tab.h"

extern __shared__ __attribute__((aligned(16))) unsigned char c3p_ode_smem[];

namespace {

typedef OdeTableau Tableau;

// Coefficients exactly as the reference writes them (c3p_ode_tab.h).
__constant__ Tableau c3p_tab[4] = C3P_ODE_TABLEAUX;

template <bool GLOBAL>
struct OMem {
  cplx* g;
  __device__ __forceinline__ cplx ld(int off) const {
    if constexpr (GLOBAL)
      return g[off];
    else
      return reinterpret_cast<cplx*>(c3p_ode_smem)[off];
  }
  __device__ __forceinline__ void st(int off, cplx v) const {
    if constexpr (GLOBAL)
      g[off] = v;
    else
      reinterpret_cast<cplx*>(c3p_ode_smem)[off] = v;
  }
};

// out = alpha * (A[D,D] @ X[D,M])  (+ out if accumulate)
template <bool G>
__device__ void mm_left(const OMem<G>& M, int out, int a, int x, int D, int Mc, cplx alpha, bool acc,
                        int tid, int nt) {
  for (int e = tid; e < D * Mc; e += nt) {
    const int i = e / Mc, j = e - i * Mc;
    cplx s = cmake(0, 0);
    for (int k = 0; k < D; ++k) cfma(s, M.ld(a + i * D + k), M.ld(x + k * Mc + j));
    s = cmul(alpha, s);
    if (acc) s = cadd(s, M.ld(out + e));
    M.st(out + e, s);
  }
  __syncthreads();
}
// out (+)= alpha * X[D,D] @ op(A)[D,D]; conjT: use A^dagger
template <bool G>
__device__ void mm_right(const OMem<G>& M, int out, int x, int a, int D, cplx alpha, bool acc, bool conjT,
                         int tid, int nt) {
  for (int e = tid; e < D * D; e += nt) {
    const int i = e / D, j = e - i * D;
    cplx s = cmake(0, 0);
    for (int k = 0; k < D; ++k) {
      const cplx av = conjT ? cconj(M.ld(a + j * D + k)) : M.ld(a + k * D + j);
      cfma(s, M.ld(x + i * D + k), av);
    }
    s = cmul(alpha, s);
    if (acc) s = cadd(s, M.ld(out + e));
    M.st(out + e, s);
  }
  __syncthreads();
}

template <bool GLOBAL>
__global__ void __launch_bounds__(256) ode_kernel(OdeArgs A) {
  const int tid = threadIdx.x, nt = blockDim.x;
  const int b = blockIdx.x;
  const int D = A.D, Mc = A.M, K = A.K, N = A.N;
  const int ssz = D * Mc, hsz = D * D;
  OMem<GLOBAL> M;
  M.g = GLOBAL ? A.scratch + (long)b * A.scratch_stride : nullptr;
  // buffer offsets (elements)
  const int oH = 0;
  const int oS = oH + hsz;          // state
  const int oY = oS + ssz;          // stage argument
  const int oK = oY + ssz;          // k1..k7
  const int oT = oK + 7 * ssz;      // temp (lindblad)
  const int oC = oT + ssz;          // col ops [C,D,D]
  const int oG = oC + A.C * hsz;    // C^dagger C per col op
  __shared__ double sigv[32];
  const Tableau& tb = c3p_tab[A.solver];
  const double* sig = A.signals + (long)b * K * N;
  const double dt = A.dt;

  if (A.wg_if_complex) {
    // launched beside the lane-row kernel (c3p_api.hip): only complex operators are this kernel's (same test as ode_mat_kernel)
    int cx = 0;
    for (int e = tid; e < (1 + K) * hsz; e += nt) cx |= ((e < hsz ? A.h0[e] : A.hks[e - hsz]).y != 0.0) ? 1 : 0;
    if (!__syncthreads_or(cx)) return;
  }
  const cplx* init = A.init + (long)b * A.init_bstride;
  for (int e = tid; e < ssz; e += nt) M.st(oS + e, init[e]);
  if (A.step == C3P_STEP_LINDBLAD_ID) {
    for (int e = tid; e < A.C * hsz; e += nt) M.st(oC + e, A.col_ops[e]);
    __syncthreads();
    for (int e = tid; e < A.C * hsz; e += nt) {
      const int c = e / hsz, r = e - c * hsz, i = r / D, j = r - i * D;
      cplx s = cmake(0, 0);
      for (int k = 0; k < D; ++k) cfma(s, cconj(M.ld(oC + c * hsz + k * D + i)), M.ld(oC + c * hsz + k * D + j));
      M.st(oG + e, s);
    }
  }
  __syncthreads();

  const cplx* hsb = A.hs ? A.hs + (long)b * A.hs_bstride : nullptr;
  for (int n = 0; n < A.n_steps; ++n) {
    for (int s = 0; s < tb.stages; ++s) {
      // interpolated control amplitudes at u = n + node  (tf_utils.py:557-559: linear,
      // linear extrapolation past the last sample)
      const double ustage = ((double)n + tb.node[s]) * (double)A.u_stride;
      if (tid < K && !hsb) {
        const double u = ustage;
        int lo = (int)floor(u);
        if (lo > N - 2) lo = N - 2;
        if (lo < 0) lo = 0;
        const double f = u - (double)lo;
        const double* y = sig + (long)tid * N;
        sigv[tid] = (N > 1) ? fma(f, y[lo + 1] - y[lo], y[lo]) : y[0];
      }
      __syncthreads();
      if (hsb) {
        // per-sample-index Hamiltonians (branch B of get_hs_of_t_ts, propagation.py:164-204):
        // stage positions are integer sample indices there
        int iu = (int)(ustage + 0.5);
        if (iu > N - 1) iu = N - 1;
        for (int e = tid; e < hsz; e += nt) M.st(oH + e, hsb[(long)iu * hsz + e]);
      } else {
        for (int e = tid; e < hsz; e += nt) {
          cplx h = A.h0[e];
          for (int k = 0; k < K; ++k) {
            const cplx x = A.hks[(long)k * hsz + e];
            h.x = fma(sigv[k], x.x, h.x);
            h.y = fma(sigv[k], x.y, h.y);
          }
          M.st(oH + e, h);
        }
      }
      // y = state + sum_j a[s][j] k_j
      for (int e = tid; e < ssz; e += nt) {
        cplx y = M.ld(oS + e);
        for (int j = 0; j < s; ++j) {
          const double a = tb.a[s][j];
          if (a != 0.0) {
            const cplx kj = M.ld(oK + j * ssz + e);
            y.x = fma(a, kj.x, y.x);
            y.y = fma(a, kj.y, y.y);
          }
        }
        M.st(oY + e, y);
      }
      __syncthreads();
      const int ok = oK + s * ssz;
      const cplx mi_dt = cmake(0.0, -dt);  // -i dt
      mm_left(M, ok, oH, oY, D, Mc, mi_dt, false, tid, nt);  // -i dt H y
      if (A.step == C3P_STEP_VON_NEUMANN_ID || A.step == C3P_STEP_LINDBLAD_ID) {
        mm_right(M, ok, oY, oH, D, cmake(0.0, dt), true, false, tid, nt);  // + i dt y H
        if (A.step == C3P_STEP_LINDBLAD_ID) {
          for (int c = 0; c < A.C; ++c) {
            mm_left(M, oT, oC + c * hsz, oY, D, D, cmake(1.0, 0.0), false, tid, nt);  // C y
            mm_right(M, ok, oT, oC + c * hsz, D, cmake(dt, 0.0), true, true, tid, nt);  // + dt C y C^+
            mm_left(M, ok, oG + c * hsz, oY, D, D, cmake(-0.5 * dt, 0.0), true, tid, nt);  // -dt/2 C^+C y
            mm_right(M, ok, oY, oG + c * hsz, D, cmake(-0.5 * dt, 0.0), true, false, tid, nt);  // -dt/2 y C^+C
          }
        }
      }
    }
    // state += sum_j b_j k_j
    for (int e = tid; e < ssz; e += nt) {
      cplx y = M.ld(oS + e);
      for (int j = 0; j < tb.stages; ++j) {
        const double bj = tb.b[j];
        if (bj != 0.0) {
          const cplx kj = M.ld(oK + j * ssz + e);
          y.x = fma(bj, kj.x, y.x);
          y.y = fma(bj, kj.y, y.y);
        }
      }
      M.st(oS + e, A.reset_each_step ? init[e] : y);
      if (A.want_all) {
        const int eo = A.transpose_out ? (e % Mc) * D + (e / Mc) : e;
        A.states[((long)b * A.n_steps + n) * ssz + eo] = y;
      }
    }
    __syncthreads();
  }
  if (!A.want_all)
    for (int e = tid; e < ssz; e += nt) {
      const int eo = A.transpose_out ? (e % Mc) * D + (e / Mc) : e;
      A.states[(long)b * ssz + eo] = M.ld(oS + e);
    }
}

}  // namespace

size_t c3p_ode_elems(int D, int M, int C) {
  return (size_t)D * D + (size_t)10 * D * M + (size_t)2 * C * D * D;
}

hipError_t c3p_launch_ode(const OdeArgs& A, bool global_scratch, hipStream_t st) {
  int threads = ((A.D * A.M + 63) / 64) * 64;
  if (threads > 256) threads = 256;
  if (threads < 64) threads = 64;
  if (global_scratch) {
    C3P_LAUNCH(ode_kernel<true>, dim3(A.B), dim3(threads), 0, st, A);
  } else {
    const size_t lds = c3p_ode_elems(A.D, A.M, A.C) * sizeof(cplx);
    // per launch: the attribute is per DEVICE, and one process may drive several GPUs
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&ode_kernel<false>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024);
    if (e != hipSuccess) return e;
    C3P_LAUNCH(ode_kernel<false>, dim3(A.B), dim3(threads), lds, st, A);
  }
  return hipGetLastError();
}
