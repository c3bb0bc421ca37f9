// Gradient of the PWC propagator with respect to the control samples.
//
// The reference differentiates the path with a GradientTape (c3/optimizers/optimizer.py:206-216 around
// the goal function; propagation.py:426-440 and tf_utils.py:144-193 are on the tape).  Given the
// cotangent Ubar of U = FR . dU_{N-1} ... dU_0, dU_n = exp(X_n), X_n = -i dt (h0 + sum_k c_k(n) hk),
// this file evaluates  grad[k,n] = Re <W_n, L(X_n, -i dt hk)>  exactly (L = Frechet derivative of exp)
// with an adjoint sweep that needs no stored partial propagators:
//
//   M_N   = (FR^H Ubar) P_N^H,                    P_N = dU_{N-1} ... dU_0
//   Z_n   = dU_n^H L(X_n, M_{n+1})                ( = int_0^1 e^{-sX} M_{n+1} e^{sX} ds )
//   grad[k,n] = Re <Z_n, -i dt hk>
//   M_n   = dU_n^H M_{n+1} dU_n                    (dU_n unitary: Hermitian Hamiltonians only)
//
// (dU_n, L(X_n, .)) come from ONE pair evaluation of the degree-18 Taylor polynomial of
// Bader-Blanes-Casas (c3p_common.h) by the product rule -- (A,dA)(B,dB) = (AB, A dB + dA B), 15 products
// + 3 per squaring -- so every slice costs 18 + 3s products against 6 + s for the forward pass.
// The time axis is cut into S segments: kernel 1 forms the segment products, kernel 2 (one workgroup per
// sample) scans them into the adjoint state at every segment end, kernel 3 sweeps each segment backwards.
//
// Correctness-first layout: one workgroup per chain, matrices in LDS (small D) or an L2-resident global
// scratch, complex 2x2 register-tiled VALU products.
#include "c3p_grad.h"

extern __shared__ __attribute__((aligned(16))) unsigned char c3p_gsmem[];

namespace {

template <bool GLOBAL>
struct GMem {
  cplx* g;
  __device__ __forceinline__ cplx ld(int off) const {
    if constexpr (GLOBAL)
      return g[off];
    else
      return reinterpret_cast<cplx*>(c3p_gsmem)[off];
  }
  __device__ __forceinline__ void st(int off, cplx v) const {
    if constexpr (GLOBAL)
      g[off] = v;
    else
      reinterpret_cast<cplx*>(c3p_gsmem)[off] = v;
  }
};

// C (= | +=) op(A) op(B); op = identity or conjugate transpose.  C must not alias A or B.
// NF != 0: the dimension is the compile-time NF (the k loop fully unrolled, its loads issued together, no integer
// division by a run-time tile count) -- the per-sample scan is a chain of dependent small products whose every cycle of
// latency is exposed.
template <bool GLOBAL, bool ACC, bool AH, bool BH, int NF = 0>
__device__ void gmm(const GMem<GLOBAL>& M, int c, int a, int b, int n_, int ld, int tid, int nt) {
  const int n = NF ? NF : n_;
  const int tn = (n + 1) >> 1;
  if constexpr (NF != 0 && !ACC) {
    // 1 x 2 tiles: NF (NF + 1) / 2 of them (45 at D = 9) keep most of the wave's lanes busy with half the multiply-adds
    // per lane of the 2 x 2 tiling (25 lanes busy)
    for (int t = tid; t < NF * tn; t += nt) {
      const int i0 = t / tn, tj = t - i0 * tn;
      const int j0 = 2 * tj, j1 = min(j0 + 1, NF - 1);
      cplx c00 = cmake(0, 0), c01 = c00;
#pragma unroll
      for (int k = 0; k < NF; ++k) {
        cplx a0, b0, b1;
        if constexpr (AH)
          a0 = cconj(M.ld(a + k * ld + i0));
        else
          a0 = M.ld(a + i0 * ld + k);
        if constexpr (BH) {
          b0 = cconj(M.ld(b + j0 * ld + k));
          b1 = cconj(M.ld(b + j1 * ld + k));
        } else {
          b0 = M.ld(b + k * ld + j0);
          b1 = M.ld(b + k * ld + j1);
        }
        cfma(c00, a0, b0);
        cfma(c01, a0, b1);
      }
      M.st(c + i0 * ld + j0, c00);
      if (j0 + 1 < NF) M.st(c + i0 * ld + j1, c01);
    }
    __syncthreads();
    return;
  }
  const int ntiles = tn * tn;
  for (int t = tid; t < ntiles; t += nt) {
    const int ti = t / tn, tj = t - ti * tn;
    const int i0 = 2 * ti, j0 = 2 * tj;
    const int i1 = min(i0 + 1, n - 1), j1 = min(j0 + 1, n - 1);
    cplx c00 = cmake(0, 0), c01 = c00, c10 = c00, c11 = c00;
#pragma unroll(NF ? NF : 2)
    for (int k = 0; k < n; ++k) {
      cplx a0, a1, b0, b1;
      if constexpr (AH) {
        a0 = cconj(M.ld(a + k * ld + i0));
        a1 = cconj(M.ld(a + k * ld + i1));
      } else {
        a0 = M.ld(a + i0 * ld + k);
        a1 = M.ld(a + i1 * ld + k);
      }
      if constexpr (BH) {
        b0 = cconj(M.ld(b + j0 * ld + k));
        b1 = cconj(M.ld(b + j1 * ld + k));
      } else {
        b0 = M.ld(b + k * ld + j0);
        b1 = M.ld(b + k * ld + j1);
      }
      cfma(c00, a0, b0);
      cfma(c01, a0, b1);
      cfma(c10, a1, b0);
      cfma(c11, a1, b1);
    }
    if constexpr (ACC) {
      c00 = cadd(c00, M.ld(c + i0 * ld + j0));
      if (j0 + 1 < n) c01 = cadd(c01, M.ld(c + i0 * ld + j1));
      if (i0 + 1 < n) {
        c10 = cadd(c10, M.ld(c + i1 * ld + j0));
        if (j0 + 1 < n) c11 = cadd(c11, M.ld(c + i1 * ld + j1));
      }
    }
    M.st(c + i0 * ld + j0, c00);
    if (j0 + 1 < n) M.st(c + i0 * ld + j1, c01);
    if (i0 + 1 < n) {
      M.st(c + i1 * ld + j0, c10);
      if (j0 + 1 < n) M.st(c + i1 * ld + j1, c11);
    }
  }
  __syncthreads();
}

template <bool GLOBAL>
__device__ __forceinline__ void mm(const GMem<GLOBAL>& M, int c, int a, int b, int n, int ld, int tid, int nt) {
  gmm<GLOBAL, false, false, false>(M, c, a, b, n, ld, tid, nt);
}
template <bool GLOBAL>
__device__ __forceinline__ void mma(const GMem<GLOBAL>& M, int c, int a, int b, int n, int ld, int tid, int nt) {
  gmm<GLOBAL, true, false, false>(M, c, a, b, n, ld, tid, nt);
}

struct Bufs {
  int A, A2, A3, A6, T1, T2, A9, T;          // values
  int dA, dA2, dA3, dA6, dT1, dT2, dA9, dT;  // derivatives (DER only)
};

__device__ __forceinline__ cplx lin5(double c0, bool diag, double c1, cplx x1, double c2, cplx x2, double c3, cplx x3,
                                     double c6, cplx x6) {
  cplx v = cmake(c1 * x1.x, c1 * x1.y);
  v.x = fma(c2, x2.x, v.x);
  v.y = fma(c2, x2.y, v.y);
  v.x = fma(c3, x3.x, v.x);
  v.y = fma(c3, x3.y, v.y);
  v.x = fma(c6, x6.x, v.x);
  v.y = fma(c6, x6.y, v.y);
  if (diag) v.x += c0;
  return v;
}

// (T, dT) = (T18(A), D T18(A)[dA]) followed by s squarings.  On return q.T / q.dT hold the results
// (buffer roles may have been swapped with T2 / dT2).
template <bool GLOBAL, bool DER>
__device__ void t18_pair(const GMem<GLOBAL>& M, Bufs& q, int s, int n, int ld, int tid, int nt) {
  mm(M, q.A2, q.A, q.A, n, ld, tid, nt);
  if constexpr (DER) {
    mm(M, q.dA2, q.A, q.dA, n, ld, tid, nt);
    mma(M, q.dA2, q.dA, q.A, n, ld, tid, nt);
  }
  mm(M, q.A3, q.A, q.A2, n, ld, tid, nt);
  if constexpr (DER) {
    mm(M, q.dA3, q.A, q.dA2, n, ld, tid, nt);
    mma(M, q.dA3, q.dA, q.A2, n, ld, tid, nt);
  }
  mm(M, q.A6, q.A3, q.A3, n, ld, tid, nt);
  if constexpr (DER) {
    mm(M, q.dA6, q.A3, q.dA3, n, ld, tid, nt);
    mma(M, q.dA6, q.dA3, q.A3, n, ld, tid, nt);
  }
  const cplx zero = cmake(0, 0);
  for (int e = tid; e < n * n; e += nt) {
    const int i = e / n, j = e - i * n, o = i * ld + j;
    const bool dg = i == j;
    const cplx x1 = M.ld(q.A + o), x2 = M.ld(q.A2 + o), x3 = M.ld(q.A3 + o), x6 = M.ld(q.A6 + o);
    M.st(q.T1 + o, lin5(0.0, false, C3P_T18_A11, x1, C3P_T18_A21, x2, C3P_T18_A31, x3, 0.0, zero));                // B1
    M.st(q.T2 + o, lin5(0.0, false, 0.0, zero, C3P_T18_B24, x2, C3P_T18_B34, x3, C3P_T18_B64, x6));                // B5
    M.st(q.A9 + o, lin5(C3P_T18_B03, dg, C3P_T18_B13, x1, C3P_T18_B23, x2, C3P_T18_B33, x3, C3P_T18_B63, x6));     // B4
    if constexpr (DER) {
      const cplx d1 = M.ld(q.dA + o), d2 = M.ld(q.dA2 + o), d3 = M.ld(q.dA3 + o), d6 = M.ld(q.dA6 + o);
      M.st(q.dT1 + o, lin5(0.0, false, C3P_T18_A11, d1, C3P_T18_A21, d2, C3P_T18_A31, d3, 0.0, zero));
      M.st(q.dT2 + o, lin5(0.0, false, 0.0, zero, C3P_T18_B24, d2, C3P_T18_B34, d3, C3P_T18_B64, d6));
      M.st(q.dA9 + o, lin5(0.0, false, C3P_T18_B13, d1, C3P_T18_B23, d2, C3P_T18_B33, d3, C3P_T18_B63, d6));
    }
  }
  __syncthreads();
  mma(M, q.A9, q.T1, q.T2, n, ld, tid, nt);  // A9 = B1 B5 + B4
  if constexpr (DER) {
    mma(M, q.dA9, q.T1, q.dT2, n, ld, tid, nt);
    mma(M, q.dA9, q.dT1, q.T2, n, ld, tid, nt);
  }
  for (int e = tid; e < n * n; e += nt) {
    const int i = e / n, j = e - i * n, o = i * ld + j;
    const bool dg = i == j;
    const cplx x1 = M.ld(q.A + o), x2 = M.ld(q.A2 + o), x3 = M.ld(q.A3 + o), x6 = M.ld(q.A6 + o);
    const cplx b3 = lin5(C3P_T18_B02, dg, C3P_T18_B12, x1, C3P_T18_B22, x2, C3P_T18_B32, x3, C3P_T18_B62, x6);
    M.st(q.T1 + o, cadd(b3, M.ld(q.A9 + o)));                                                                    // B3 + A9
    M.st(q.T + o, lin5(0.0, false, C3P_T18_B11, x1, C3P_T18_B21, x2, C3P_T18_B31, x3, C3P_T18_B61, x6));         // B2
    if constexpr (DER) {
      const cplx d1 = M.ld(q.dA + o), d2 = M.ld(q.dA2 + o), d3 = M.ld(q.dA3 + o), d6 = M.ld(q.dA6 + o);
      const cplx db3 = lin5(0.0, false, C3P_T18_B12, d1, C3P_T18_B22, d2, C3P_T18_B32, d3, C3P_T18_B62, d6);
      M.st(q.dT1 + o, cadd(db3, M.ld(q.dA9 + o)));
      M.st(q.dT + o, lin5(0.0, false, C3P_T18_B11, d1, C3P_T18_B21, d2, C3P_T18_B31, d3, C3P_T18_B61, d6));
    }
  }
  __syncthreads();
  mma(M, q.T, q.T1, q.A9, n, ld, tid, nt);  // T = B2 + (B3 + A9) A9
  if constexpr (DER) {
    mma(M, q.dT, q.T1, q.dA9, n, ld, tid, nt);
    mma(M, q.dT, q.dT1, q.A9, n, ld, tid, nt);
  }
  for (int it = 0; it < s; ++it) {
    if constexpr (DER) {
      mm(M, q.dT2, q.T, q.dT, n, ld, tid, nt);
      mma(M, q.dT2, q.dT, q.T, n, ld, tid, nt);
      const int t = q.dT;
      q.dT = q.dT2;
      q.dT2 = t;
    }
    mm(M, q.T2, q.T, q.T, n, ld, tid, nt);
    const int t = q.T;
    q.T = q.T2;
    q.T2 = t;
  }
}

struct Shared {
  double red[256];
  double mu[2];
  int s;
};

// X~ = -i dt (H_n - tr(H_n)/D) / 2^s into buffer `dst`; returns s and the removed shift mu = -i dt tr(H_n)/D.
// general generators (A.general): X = dt (G_0 + sum c_k G_k); conjT: X^H is assembled instead (all that follows -- shift, norm,
// scaling -- then refers to the matrix that is exponentiated)
template <bool GLOBAL>
__device__ void assemble(const GMem<GLOBAL>& M, const GradArgs& A, Shared& sh, int b, int n, int dst, int tid, int nt,
                         int& s_out, double& mu_r, double& mu_i, bool conjT = false) {
  const int D = A.D, ld = A.ld;
  const cplx* h0b = A.h0 + (long)b * A.h0_bstride;
  const cplx* hkb = A.hks + (long)b * A.hks_bstride;
  const double* sig = A.signals + (long)b * A.K * A.N;
  for (int e = tid; e < D * D; e += nt) {
    const int i = e / D, j = e - i * D;
    cplx h = h0b[e];
    for (int k = 0; k < A.K; ++k) {
      const double c = sig[(long)k * A.N + n];
      const cplx x = hkb[(long)k * D * D + e];
      h.x = fma(c, x.x, h.x);
      h.y = fma(c, x.y, h.y);
    }
    const cplx x = A.general ? cmake(h.x * A.dt, h.y * A.dt) : cmake(h.y * A.dt, -h.x * A.dt);  // dt G or -i dt h
    if (conjT)
      M.st(dst + j * ld + i, cconj(x));
    else
      M.st(dst + i * ld + j, x);
  }
  __syncthreads();
  if (tid == 0) {
    double tr = 0.0, ti = 0.0;
    for (int i = 0; i < D; ++i) {
      const cplx v = M.ld(dst + i * ld + i);
      tr += v.x;
      ti += v.y;
    }
    sh.mu[0] = 0.0;  // imaginary shift only (c3p_smalld.hip: build_tables)
    sh.mu[1] = ti / D;
  }
  __syncthreads();
  mu_r = sh.mu[0];
  mu_i = sh.mu[1];
  double cmax = 0.0;
  for (int j = tid; j < D; j += nt) {
    cplx d = M.ld(dst + j * ld + j);
    d.x -= mu_r;
    d.y -= mu_i;
    M.st(dst + j * ld + j, d);
    double cs = 0.0;
    for (int i = 0; i < D; ++i) cs += cabs1(i == j ? d : M.ld(dst + i * ld + j));
    cmax = fmax(cmax, cs);
  }
  sh.red[tid] = cmax;
  __syncthreads();
  if (tid == 0) {
    double nrm = 0.0;
    for (int t = 0; t < nt; ++t) nrm = fmax(nrm, sh.red[t]);
    int s = 0;
    double p = C3P_T18_THETA;
    while (p < nrm && s < 40) {
      p *= 2.0;
      ++s;
    }
    sh.s = s;
  }
  __syncthreads();
  const int s = sh.s;
  s_out = s;
  if (s > 0) {
    const double sc = ldexp(1.0, -s);
    for (int e = tid; e < D * D; e += nt) {
      const int o = (e / D) * ld + (e % D);
      M.st(dst + o, cscale(M.ld(dst + o), sc));
    }
  }
  __syncthreads();
}

template <bool GLOBAL>
__device__ void copy_in(const GMem<GLOBAL>& M, int dst, const cplx* src, int D, int ld, int tid, int nt) {
  for (int e = tid; e < D * D; e += nt) M.st(dst + (e / D) * ld + (e % D), src[e]);
  __syncthreads();
}
template <bool GLOBAL>
__device__ void copy_out(const GMem<GLOBAL>& M, cplx* dst, int src, int D, int ld, int tid, int nt) {
  for (int e = tid; e < D * D; e += nt) dst[e] = M.ld(src + (e / D) * ld + (e % D));
}

// ---- kernel 1: ordered product of every time segment (no frame rotation) ----
template <bool GLOBAL>
__global__ void __launch_bounds__(256) grad_seg_kernel(GradArgs A) {
  __shared__ Shared sh;
  const int tid = threadIdx.x, nt = blockDim.x;
  const int wg = blockIdx.x;
  const int b = wg / A.S, seg = wg - b * A.S;
  const int n0 = (int)(((long)seg * A.N) / A.S), n1 = (int)(((long)(seg + 1) * A.N) / A.S);
  const int D = A.D, ld = A.ld, msz = ld * D;
  GMem<GLOBAL> M;
  M.g = GLOBAL ? A.scratch + (long)wg * A.scratch_stride : nullptr;
  Bufs q = {};
  q.A = 0, q.A2 = msz, q.A3 = 2 * msz, q.A6 = 3 * msz, q.T1 = 4 * msz, q.T2 = 5 * msz, q.A9 = 6 * msz, q.T = 7 * msz;
  int bU = 8 * msz, bV = 9 * msz;
  double mus_r = 0.0, mus_i = 0.0;
  for (int n = n0; n < n1; ++n) {
    int s;
    double mr, mi;
    assemble(M, A, sh, b, n, q.A, tid, nt, s, mr, mi);
    t18_pair<GLOBAL, false>(M, q, s, D, ld, tid, nt);
    mus_r += mr;
    mus_i = c3p_phase_add(mus_i, mi);
    if (n == n0) {
      const int t = bU;  // U <- T by renaming
      bU = q.T;
      q.T = t;
    } else {
      mm(M, bV, q.T, bU, D, ld, tid, nt);
      const int t = bU;
      bU = bV;
      bV = t;
    }
  }
  double sn, cs;
  sincos(mus_i, &sn, &cs);
  const double er = exp(mus_r);
  const cplx ph = cmake(er * cs, er * sn);
  cplx* dst = A.seg + (long)wg * D * D;
  for (int e = tid; e < D * D; e += nt) dst[e] = cmul(ph, M.ld(bU + (e / D) * ld + (e % D)));
}

// ---- kernel 2: per sample, adjoint state M at the end of every segment ----
template <bool GLOBAL, int NF = 0>
__global__ void __launch_bounds__(256) grad_scan_kernel(GradArgs A) {
  const int tid = threadIdx.x, nt = blockDim.x;
  const int b = blockIdx.x;
  const int D = A.D, ld = A.ld, msz = ld * D;
  GMem<GLOBAL> M;
  M.g = GLOBAL ? A.scratch + (long)b * A.S * A.scratch_stride : nullptr;  // the scratch of this sample's first chain
  int bU = 0, bV = msz, bS = 2 * msz, bM = 3 * msz;
  const cplx* segb = A.seg + (long)b * A.S * D * D;
  // The scan is a chain of dependent small products: every global round trip in it is exposed.  With the matrices in LDS
  // the segment products 1 .. npre are fetched ONCE, in one pass of independent loads, into the slots the backward
  // kernel uses for its T18 intermediates (the scan needs four of the C3P_GRAD_NMAT); the rest are read when needed.
  const int npre = GLOBAL ? 0 : min(A.S - 1, C3P_GRAD_NMAT - 4);
  auto slot = [&](int j) -> int { return (3 + j) * msz; };  // 1 <= j <= npre
  for (int e = tid; e < npre * D * D; e += nt) {
    const int j = e / (D * D), r = e - j * D * D;
    M.st(slot(j + 1) + (r / D) * ld + (r % D), segb[(long)(j + 1) * D * D + r]);
  }
  // Ubar' = FR^H Ubar (row i times e^{-i phi_i}) goes to registers now as well
  const cplx* ub = A.Ubar + (long)b * D * D;
  copy_in(M, bU, segb, D, ld, tid, nt);
  for (int j = 1; j < A.S; ++j) {
    int src = bS;
    if (j <= npre)
      src = slot(j);
    else
      copy_in(M, bS, segb + (long)j * D * D, D, ld, tid, nt);
    gmm<GLOBAL, false, false, false, NF>(M, bV, src, bU, D, ld, tid, nt);
    const int t = bU;
    bU = bV;
    bV = t;
  }
  // fused goal: the total product P is in bU, so the gate infidelity of U = FR P and its cotangent are formed here
  // (fidelities.py:154-184, 290-313) instead of a forward call, an overlap kernel and host-framework ops in front of this one
  __shared__ double gsh[2 * C3P_GOAL_LMAX + 4];
  const bool goal = A.goal_rows != nullptr;
  if (goal) {
    const int L = A.goal_L;
    if (tid < L) {
      const int ra = A.goal_rows[tid];
      double sn = 0.0, cs = 1.0;
      if (A.fr_phase) sincos(A.fr_phase[(long)b * D + ra], &sn, &cs);
      cplx acc = cmake(0, 0);
      for (int c = 0; c < L; ++c) {
        const cplx g = A.goal_ideal[tid * L + c];
        acc = cadd(acc, cmul(cmake(g.x, -g.y), M.ld(bU + ra * ld + A.goal_rows[c])));
      }
      acc = cmul(cmake(cs, sn), acc);
      gsh[2 * tid] = acc.x, gsh[2 * tid + 1] = acc.y;
    }
    __syncthreads();
    if (tid == 0) {
      double sr = 0.0, si = 0.0;
      for (int a = 0; a < L; ++a) sr += gsh[2 * a], si += gsh[2 * a + 1];
      const double s2 = sr * sr + si * si;
      const double coef = A.goal_kind == 0 ? -2.0 / ((double)L * L) : -2.0 / ((double)L * (L + 1));
      A.goal_infid[b] = A.goal_kind == 0 ? 1.0 - s2 / ((double)L * L) : 1.0 - (s2 / L + 1.0) / (L + 1.0);
      gsh[2 * C3P_GOAL_LMAX] = coef * sr, gsh[2 * C3P_GOAL_LMAX + 1] = coef * si;
    }
    __syncthreads();
    // d infid / d phi_i = -Im sum_j conj(Ubar_ij) U_ij = -Im(conj(c s) * s_i), s_i the row's share of the overlap
    if (A.goal_gphase != nullptr) {
      for (int i = tid; i < D; i += nt) A.goal_gphase[(long)b * D + i] = 0.0;
      __syncthreads();
      if (tid < L) {
        const double cr = gsh[2 * C3P_GOAL_LMAX], ci = gsh[2 * C3P_GOAL_LMAX + 1];
        A.goal_gphase[(long)b * D + A.goal_rows[tid]] = -(cr * gsh[2 * tid + 1] - ci * gsh[2 * tid]);
      }
    }
  }
  for (int e = tid; e < D * D; e += nt) {
    const int i = e / D, j = e - i * D;
    double sn = 0.0, cs = 1.0;
    if (A.fr_phase) sincos(A.fr_phase[(long)b * D + i], &sn, &cs);
    cplx v;
    if (goal) {
      if (A.goal_U != nullptr) A.goal_U[(long)b * D * D + e] = cmul(cmake(cs, sn), M.ld(bU + i * ld + j));
      int ai = -1, aj = -1;
      for (int a = 0; a < A.goal_L; ++a) {
        const int r = A.goal_rows[a];
        ai = (r == i) ? a : ai;
        aj = (r == j) ? a : aj;
      }
      v = cmake(0, 0);
      if (ai >= 0 && aj >= 0) v = cmul(cmake(gsh[2 * C3P_GOAL_LMAX], gsh[2 * C3P_GOAL_LMAX + 1]), A.goal_ideal[ai * A.goal_L + aj]);
    } else {
      v = ub[e];
    }
    if (A.fr_phase) v = cmul(cmake(cs, -sn), v);
    M.st(bS + i * ld + j, v);
  }
  __syncthreads();
  gmm<GLOBAL, false, false, true, NF>(M, bM, bS, bU, D, ld, tid, nt);
  cplx* mb = A.Mb + (long)b * A.S * D * D;
  for (int j = A.S - 1; j >= 0; --j) {
    copy_out(M, mb + (long)j * D * D, bM, D, ld, tid, nt);
    if (j == 0) break;
    int src = bS;
    if (j <= npre)
      src = slot(j), __syncthreads();  // (orders the copy_out reads before M is rewritten)
    else
      copy_in(M, bS, segb + (long)j * D * D, D, ld, tid, nt);  // (its barrier does the same)
    gmm<GLOBAL, false, false, false, NF>(M, bV, bM, src, D, ld, tid, nt);                           // V = M S_j
    gmm<GLOBAL, false, true, false, NF>(M, bM, src, bV, D, ld, tid, nt);  // M = S_j^H V
  }
}

// ---- kernel 3: backward sweep of one segment ----
template <bool GLOBAL>
__global__ void __launch_bounds__(256) grad_bwd_kernel(GradArgs A) {
  __shared__ Shared sh;
  const int tid = threadIdx.x, nt = blockDim.x;
  const int wg = blockIdx.x;
  const int b = wg / A.S, seg = wg - b * A.S;
  const int n0 = (int)(((long)seg * A.N) / A.S), n1 = (int)(((long)(seg + 1) * A.N) / A.S);
  const int D = A.D, ld = A.ld, msz = ld * D;
  GMem<GLOBAL> M;
  M.g = GLOBAL ? A.scratch + (long)wg * A.scratch_stride : nullptr;
  Bufs q;
  q.A = 0, q.A2 = msz, q.A3 = 2 * msz, q.A6 = 3 * msz, q.T1 = 4 * msz, q.T2 = 5 * msz, q.A9 = 6 * msz, q.T = 7 * msz;
  q.dA = 8 * msz, q.dA2 = 9 * msz, q.dA3 = 10 * msz, q.dA6 = 11 * msz, q.dT1 = 12 * msz, q.dT2 = 13 * msz,
  q.dA9 = 14 * msz, q.dT = 15 * msz;
  int bM = 16 * msz, bV = 17 * msz;
  const int bZ = 18 * msz;
  const cplx* hkb = A.hks + (long)b * A.hks_bstride;
  copy_in(M, bM, A.Mb + (long)wg * D * D, D, ld, tid, nt);
  for (int n = n1 - 1; n >= n0; --n) {
    int s;
    double mr, mi;
    assemble(M, A, sh, b, n, q.A, tid, nt, s, mr, mi);
    const double sc = ldexp(1.0, -s);
    for (int e = tid; e < D * D; e += nt) {
      const int o = (e / D) * ld + (e % D);
      M.st(q.dA + o, cscale(M.ld(bM + o), sc));
    }
    __syncthreads();
    t18_pair<GLOBAL, true>(M, q, s, D, ld, tid, nt);  // T = e^{-mu} dU_n, dT = e^{-mu} L(X_n, M_{n+1})
    gmm<GLOBAL, false, true, false>(M, bZ, q.T, q.dT, D, ld, tid, nt);  // Z = dU^H L (the shift cancels)
    if (A.zout != nullptr) copy_out(M, A.zout + ((long)b * A.N + n) * D * D, bZ, D, ld, tid, nt);
    // grad[k,n] = Re <Z, -i dt hk> = dt sum (Zx hk_y - Zy hk_x)
    for (int k = 0; k < A.K; ++k) {
      double part = 0.0;
      for (int e = tid; e < D * D; e += nt) {
        const cplx z = M.ld(bZ + (e / D) * ld + (e % D));
        const cplx h = hkb[(long)k * D * D + e];
        part = fma(z.x, h.y, part);
        part = fma(-z.y, h.x, part);
      }
      sh.red[tid] = part;
      __syncthreads();
      if (tid == 0) {
        double tot = 0.0;
        for (int t = 0; t < nt; ++t) tot += sh.red[t];
        A.grad[((long)b * A.K + k) * A.N + n] = tot * A.dt;
      }
      __syncthreads();
    }
    if (n > n0) {
      mm(M, bV, bM, q.T, D, ld, tid, nt);                              // V = M dU
      gmm<GLOBAL, false, true, false>(M, bM, q.T, bV, D, ld, tid, nt);  // M = dU^H V
    }
  }
}


// ---------------------------------------------------------------------------------------------------------------
// General (non-unitary) generators: the Lindblad path (c3/optimizers/optimizer.py:206-216 over propagation.py:551-585).
//   U = FR dU_{N-1} ... dU_0,  dU_n = exp(X_n),  X_n = dt (G_0 + sum_k c_k(n) G_k)
//   grad[k,n] = Re <Ubar, FR S_n L(X_n; dt G_k) P_n> = Re <L(X_n^H; M_n), dt G_k>,   M_n = A_n P_n^H,
//   P_n = dU_{n-1} ... dU_0 (prefix),  A_n = S_n^H FR^H Ubar,  S_n = dU_{N-1} ... dU_{n+1} (suffix).
// A_n runs backwards (A_{n-1} = dU_n^H A_n), P_n forwards, and a dissipative slice has no cheap inverse to turn one of them
// around: the scan leaves the prefix at the START and the left adjoint at the END of every segment, the sweep of a segment first
// walks forward storing P_n of each of its slices (memory: [B,N,D,D]), then backward with ONE pair evaluation per slice
// at Y = X_n^H: its value exp(X_n^H) = dU_n^H advances A, its derivative L(X_n^H; M_n) is the cotangent of the generator.
// ---------------------------------------------------------------------------------------------------------------
template <bool GLOBAL>
__global__ void __launch_bounds__(256) grad_scan_general_kernel(GradArgs A) {
  const int tid = threadIdx.x, nt = blockDim.x;
  const int b = blockIdx.x;
  const int D = A.D, ld = A.ld, msz = ld * D;
  GMem<GLOBAL> M;
  M.g = GLOBAL ? A.scratch + (long)b * A.S * A.scratch_stride : nullptr;
  int bU = 0, bV = msz;
  const int bS = 2 * msz;
  const cplx* segb = A.seg + (long)b * A.S * D * D;
  // the two sweeps are independent chains of S - 1 products: with on-chip matrices they run as two workgroups per sample
  // (gridDim.y = 2: y = 0 the prefixes, y = 1 the left adjoints; 212 -> 110 us for 64 samples of 64 segments at Dm = 9)
  const bool do_pre = GLOBAL || gridDim.y == 1 || blockIdx.y == 0, do_adj = GLOBAL || gridDim.y == 1 || blockIdx.y == 1;
  if (do_pre) {
  for (int e = tid; e < D * D; e += nt) M.st(bU + (e / D) * ld + (e % D), cmake((e / D) == (e % D) ? 1.0 : 0.0, 0.0));
  __syncthreads();
  cplx* pb = A.pre + (long)b * A.S * D * D;
  for (int j = 0; j < A.S; ++j) {  // prefix in front of segment j
    copy_out(M, pb + (long)j * D * D, bU, D, ld, tid, nt);
    if (j == A.S - 1) break;
    copy_in(M, bS, segb + (long)j * D * D, D, ld, tid, nt);
    mm(M, bV, bS, bU, D, ld, tid, nt);
    const int t = bU;
    bU = bV;
    bV = t;
  }
  __syncthreads();
  }
  if (!do_adj) return;
  const cplx* ub = A.Ubar + (long)b * D * D;
  for (int e = tid; e < D * D; e += nt) {  // FR^H Ubar: row i times e^{-i phi_i}
    const int i = e / D, j = e - i * D;
    cplx v = ub[e];
    if (A.fr_phase) {
      double sn, cs;
      sincos(A.fr_phase[(long)b * D + i], &sn, &cs);
      v = cmul(cmake(cs, -sn), v);
    }
    M.st(bU + i * ld + j, v);
  }
  __syncthreads();
  cplx* mb = A.Mb + (long)b * A.S * D * D;
  for (int j = A.S - 1; j >= 0; --j) {  // left adjoint behind segment j
    copy_out(M, mb + (long)j * D * D, bU, D, ld, tid, nt);
    if (j == 0) break;
    copy_in(M, bS, segb + (long)j * D * D, D, ld, tid, nt);
    gmm<GLOBAL, false, true, false>(M, bV, bS, bU, D, ld, tid, nt);  // S_j^H A
    const int t = bU;
    bU = bV;
    bV = t;
  }
}

template <bool GLOBAL>
__global__ void __launch_bounds__(256) grad_bwd_general_kernel(GradArgs A) {
  __shared__ Shared sh;
  const int tid = threadIdx.x, nt = blockDim.x;
  const int wg = blockIdx.x;
  const int b = wg / A.S, seg = wg - b * A.S;
  const int n0 = (int)(((long)seg * A.N) / A.S), n1 = (int)(((long)(seg + 1) * A.N) / A.S);
  const int D = A.D, ld = A.ld, msz = ld * D;
  GMem<GLOBAL> M;
  M.g = GLOBAL ? A.scratch + (long)wg * A.scratch_stride : nullptr;
  Bufs q;
  q.A = 0, q.A2 = msz, q.A3 = 2 * msz, q.A6 = 3 * msz, q.T1 = 4 * msz, q.T2 = 5 * msz, q.A9 = 6 * msz, q.T = 7 * msz;
  q.dA = 8 * msz, q.dA2 = 9 * msz, q.dA3 = 10 * msz, q.dA6 = 11 * msz, q.dT1 = 12 * msz, q.dT2 = 13 * msz,
  q.dA9 = 14 * msz, q.dT = 15 * msz;
  const int bP = 16 * msz, bA = 17 * msz, bM = 18 * msz, bV = 19 * msz;
  const cplx* gkb = A.hks + (long)b * A.hks_bstride;
  cplx* ps = A.pstore + (long)b * A.N * D * D;
  auto phase_of = [](double mr, double mi) {
    double sn, cs;
    sincos(mi, &sn, &cs);
    const double er = exp(mr);
    return cmake(er * cs, er * sn);
  };
  // forward: the prefix in front of every slice of this segment
  copy_in(M, bP, A.pre + (long)wg * D * D, D, ld, tid, nt);
  for (int n = n0; n < n1; ++n) {
    copy_out(M, ps + (long)n * D * D, bP, D, ld, tid, nt);
    if (n == n1 - 1) break;
    int s;
    double mr, mi;
    assemble(M, A, sh, b, n, q.A, tid, nt, s, mr, mi);
    t18_pair<GLOBAL, false>(M, q, s, D, ld, tid, nt);
    mm(M, bV, q.T, bP, D, ld, tid, nt);
    const cplx ph = phase_of(mr, mi);
    for (int e = tid; e < D * D; e += nt) {
      const int o = (e / D) * ld + (e % D);
      M.st(bP + o, cmul(ph, M.ld(bV + o)));
    }
    __syncthreads();
  }
  __syncthreads();
  // backward
  copy_in(M, bA, A.Mb + (long)wg * D * D, D, ld, tid, nt);
  for (int n = n1 - 1; n >= n0; --n) {
    copy_in(M, bP, ps + (long)n * D * D, D, ld, tid, nt);
    gmm<GLOBAL, false, false, true>(M, bM, bA, bP, D, ld, tid, nt);  // M = A P_n^H
    int s;
    double mr, mi;
    assemble(M, A, sh, b, n, q.A, tid, nt, s, mr, mi, true);  // Y = X_n^H
    const double sc = ldexp(1.0, -s);
    for (int e = tid; e < D * D; e += nt) {
      const int o = (e / D) * ld + (e % D);
      M.st(q.dA + o, cscale(M.ld(bM + o), sc));
    }
    __syncthreads();
    t18_pair<GLOBAL, true>(M, q, s, D, ld, tid, nt);  // T = e^{-mu} exp(Y), dT = e^{-mu} L(Y; M)
    const cplx ph = phase_of(mr, mi);
    // grad[k,n] = Re <Z, dt G_k>, Z = e^{mu} dT
    for (int k = 0; k < A.K; ++k) {
      double part = 0.0;
      for (int e = tid; e < D * D; e += nt) {
        const cplx z = cmul(ph, M.ld(q.dT + (e / D) * ld + (e % D)));
        const cplx g = gkb[(long)k * D * D + e];
        part = fma(z.x, g.x, part);
        part = fma(z.y, g.y, part);
      }
      sh.red[tid] = part;
      __syncthreads();
      if (tid == 0) {
        double tot = 0.0;
        for (int t = 0; t < nt; ++t) tot += sh.red[t];
        A.grad[((long)b * A.K + k) * A.N + n] = tot * A.dt;
      }
      __syncthreads();
    }
    if (n > n0) {
      mm(M, bV, q.T, bA, D, ld, tid, nt);  // A <- dU_n^H A
      for (int e = tid; e < D * D; e += nt) {
        const int o = (e / D) * ld + (e % D);
        M.st(bA + o, cmul(ph, M.ld(bV + o)));
      }
      __syncthreads();
    }
  }
}

// row r = (i,j), column c = (k,l) of -i (H (x) 1 - 1 (x) H^T) [+ clp]: -i (H[i,k] d_jl - d_ik H[l,j])
__global__ void lind_gen_kernel(const cplx* h0, long h0_bs, const cplx* hks, long hk_bs, const cplx* clp, int K, int D, cplx* out) {
  const int Dm = D * D;
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (long)Dm * Dm) return;
  const int b = blockIdx.y, which = blockIdx.z;
  const cplx* H = which == 0 ? h0 + (long)b * h0_bs : hks + (long)b * hk_bs + (long)(which - 1) * D * D;
  const int r = (int)(e / Dm), c = (int)(e - (long)r * Dm);
  const int i = r / D, j = r - i * D, k = c / D, l = c - k * D;
  cplx v = which == 0 ? clp[e] : cmake(0, 0);
  if (j == l) {
    const cplx h = H[i * D + k];
    v.x += h.y;
    v.y -= h.x;
  }
  if (i == k) {
    const cplx h = H[l * D + j];
    v.x -= h.y;
    v.y += h.x;
  }
  out[((long)b * (K + 1) + which) * Dm * Dm + e] = v;
}

// dense superoperator generator of every slice Hamiltonian (no dt): row (i,j), column (k,l)
__global__ void lind_slice_gen_kernel(const cplx* hs, long hs_bs, const cplx* clp, int N, int D, cplx* out) {
  const int Dm = D * D;
  const long e = (long)blockIdx.y * blockDim.x + threadIdx.x;
  if (e >= (long)Dm * Dm) return;
  const long m = blockIdx.x;  // b N + n
  const long b = m / N, n = m - b * N;
  const cplx* H = hs + b * hs_bs + n * (long)D * D;
  const int r = (int)(e / Dm), c = (int)(e - (long)r * Dm);
  const int i = r / D, j = r - i * D, k = c / D, l = c - k * D;
  cplx v = clp[e];
  if (j == l) {
    const cplx h = H[i * D + k];
    v.x += h.y;
    v.y -= h.x;
  }
  if (i == k) {
    const cplx h = H[l * D + j];
    v.x -= h.y;
    v.y += h.x;
  }
  out[m * Dm * Dm + e] = v;
}

}  // namespace

int c3p_grad_threads(int D) { return D <= 10 ? 64 : (D <= 20 ? 128 : 256); }
size_t c3p_grad_lds_bytes(int D) { return (size_t)C3P_GRAD_NMAT * (D | 1) * D * sizeof(cplx); }

namespace {
template <typename KT>
hipError_t launch_one(KT kern, unsigned grid, const GradArgs& A, bool global_scratch, hipStream_t st) {
  const int nt = c3p_grad_threads(A.D);
  const size_t lds = global_scratch ? 0 : c3p_grad_lds_bytes(A.D);
  if (lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  C3P_LAUNCH(kern, dim3(grid), dim3(nt), lds, st, A);
  return hipGetLastError();
}
}  // namespace

hipError_t c3p_launch_grad_seg(const GradArgs& A, bool global_scratch, hipStream_t st) {
  return global_scratch ? launch_one(grad_seg_kernel<true>, (unsigned)(A.B * A.S), A, true, st)
                        : launch_one(grad_seg_kernel<false>, (unsigned)(A.B * A.S), A, false, st);
}
hipError_t c3p_launch_grad_scan(const GradArgs& A, bool global_scratch, hipStream_t st) {
  if (global_scratch) return launch_one(grad_scan_kernel<true>, (unsigned)A.B, A, true, st);
  switch (A.D) {  // small matrices: the scan with compile-time dimension
    case 2: return launch_one(grad_scan_kernel<false, 2>, (unsigned)A.B, A, false, st);
    case 3: return launch_one(grad_scan_kernel<false, 3>, (unsigned)A.B, A, false, st);
    case 4: return launch_one(grad_scan_kernel<false, 4>, (unsigned)A.B, A, false, st);
    case 5: return launch_one(grad_scan_kernel<false, 5>, (unsigned)A.B, A, false, st);
    case 6: return launch_one(grad_scan_kernel<false, 6>, (unsigned)A.B, A, false, st);
    case 7: return launch_one(grad_scan_kernel<false, 7>, (unsigned)A.B, A, false, st);
    case 8: return launch_one(grad_scan_kernel<false, 8>, (unsigned)A.B, A, false, st);
    case 9: return launch_one(grad_scan_kernel<false, 9>, (unsigned)A.B, A, false, st);
    case 10: return launch_one(grad_scan_kernel<false, 10>, (unsigned)A.B, A, false, st);
    case 11: return launch_one(grad_scan_kernel<false, 11>, (unsigned)A.B, A, false, st);
    case 12: return launch_one(grad_scan_kernel<false, 12>, (unsigned)A.B, A, false, st);
    default: return launch_one(grad_scan_kernel<false>, (unsigned)A.B, A, false, st);
  }
}
hipError_t c3p_launch_grad_bwd(const GradArgs& A, bool global_scratch, hipStream_t st) {
  return global_scratch ? launch_one(grad_bwd_kernel<true>, (unsigned)(A.B * A.S), A, true, st)
                        : launch_one(grad_bwd_kernel<false>, (unsigned)(A.B * A.S), A, false, st);
}

size_t c3p_grad_lds_bytes_general(int D) { return (size_t)C3P_GRAD_NMAT_GENERAL * (D | 1) * D * sizeof(cplx); }

namespace {
template <typename KT>
hipError_t launch_general(KT kern, unsigned grid, const GradArgs& A, bool global_scratch, hipStream_t st) {
  const int nt = c3p_grad_threads(A.D);
  const size_t lds = global_scratch ? 0 : c3p_grad_lds_bytes_general(A.D);
  if (lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  C3P_LAUNCH(kern, dim3(grid), dim3(nt), lds, st, A);
  return hipGetLastError();
}
}  // namespace

hipError_t c3p_launch_grad_scan_general(const GradArgs& A, bool global_scratch, hipStream_t st) {
  // the scan keeps three matrices: they fit the LDS at every dimension the on-chip sweeps reach
  const size_t lds3 = (size_t)3 * A.ld * A.D * sizeof(cplx);
  if (lds3 <= (size_t)150 * 1024) {
    auto kern = grad_scan_general_kernel<false>;
    if (lds3 > 48 * 1024) {
      hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3);
      if (e != hipSuccess) return e;
    }
    C3P_LAUNCH(kern, dim3((unsigned)A.B, 2), dim3(c3p_grad_threads(A.D)), lds3, st, A);
    return hipGetLastError();
  }
  (void)global_scratch;
  return launch_general(grad_scan_general_kernel<true>, (unsigned)A.B, A, true, st);
}
hipError_t c3p_launch_grad_bwd_general(const GradArgs& A, bool global_scratch, hipStream_t st) {
  return global_scratch ? launch_general(grad_bwd_general_kernel<true>, (unsigned)(A.B * A.S), A, true, st)
                        : launch_general(grad_bwd_general_kernel<false>, (unsigned)(A.B * A.S), A, false, st);
}
hipError_t c3p_launch_lind_generators(const cplx* h0, long h0_bstride, const cplx* hks, long hks_bstride, const cplx* clp, int nb,
                                      int K, int D, cplx* out, hipStream_t st) {
  const long nel = (long)D * D * D * D;
  C3P_LAUNCH(lind_gen_kernel, dim3((unsigned)((nel + 255) / 256), (unsigned)nb, (unsigned)(K + 1)), dim3(256), 0, st, h0, h0_bstride,
                     hks, hks_bstride, clp, K, D, out);
  return hipGetLastError();
}

hipError_t c3p_launch_lind_slice_generators(const cplx* hs, long hs_bstride, const cplx* clp, int B, int N, int D, cplx* out, hipStream_t st) {
  const long nel = (long)D * D * D * D;
  C3P_LAUNCH(lind_slice_gen_kernel, dim3((unsigned)((long)B * N), (unsigned)((nel + 255) / 256)), dim3(256), 0, st, hs, hs_bstride, clp, N,
                     D, out);
  return hipGetLastError();
}
