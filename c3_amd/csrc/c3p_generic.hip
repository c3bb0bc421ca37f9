// Generic propagator-chain kernel: any matrix dimension, matrices staged in LDS
// (Dm <= ~37) or in an L2-resident global scratch region (larger Dm, e.g. the
// 81x81 Lindblad superoperator).  One workgroup owns one (sample, time-segment)
// chain and walks its slices sequentially:
//
//   assemble  X = -i dt (h0 + sum_k c_k(n) hk)            (propagation.py:426-439)
//         or  X = dt L(H(n)) for the Lindblad superoperator (propagation.py:551-582)
//   E = exp(X)    scaled Taylor / Paterson-Stockmeyer + squarings (stands in for
//                 tf.linalg.expm, propagation.py:440,584)
//   U <- E U      ordered product, later slice on the left  (tf_utils.py:144-193)
//
// This is the correctness-first fallback; the specialised kernels
// (c3p_smalld.hip, c3p_mfma.hip) take over where they apply.
#include "c3p_common.h"
#include "c3p_kernels.h"

extern __shared__ __attribute__((aligned(16))) unsigned char c3p_smem[];

namespace {

template <bool GLOBAL>
struct Mem {
  cplx* g;
  __device__ __forceinline__ cplx ld(int off) const {
    if constexpr (GLOBAL)
      return g[off];
    else
      return reinterpret_cast<cplx*>(c3p_smem)[off];
  }
  __device__ __forceinline__ void st(int off, cplx v) const {
    if constexpr (GLOBAL)
      g[off] = v;
    else
      reinterpret_cast<cplx*>(c3p_smem)[off] = v;
  }
};

// C = alpha * A * B, all n x n with leading dimension ld, 2x2 register tiles.
template <bool GLOBAL>
__device__ void mm(const Mem<GLOBAL>& M, int c, int a, int b, int n, int ld, cplx alpha, int tid,
                   int nt) {
  const int tn = (n + 1) >> 1;
  const int ntiles = tn * tn;
  for (int t = tid; t < ntiles; t += nt) {
    const int ti = t / tn, tj = t - ti * tn;
    const int i0 = 2 * ti, j0 = 2 * tj;
    const int i1 = min(i0 + 1, n - 1), j1 = min(j0 + 1, n - 1);
    cplx c00 = cmake(0, 0), c01 = c00, c10 = c00, c11 = c00;
    const int ar0 = a + i0 * ld, ar1 = a + i1 * ld;
#pragma unroll 4
    for (int k = 0; k < n; ++k) {
      const cplx a0 = M.ld(ar0 + k), a1 = M.ld(ar1 + k);
      const cplx b0 = M.ld(b + k * ld + j0), b1 = M.ld(b + k * ld + j1);
      cfma(c00, a0, b0);
      cfma(c01, a0, b1);
      cfma(c10, a1, b0);
      cfma(c11, a1, b1);
    }
    M.st(c + i0 * ld + j0, cmul(alpha, c00));
    if (j0 + 1 < n) M.st(c + i0 * ld + j1, cmul(alpha, c01));
    if (i0 + 1 < n) {
      M.st(c + i1 * ld + j0, cmul(alpha, c10));
      if (j0 + 1 < n) M.st(c + i1 * ld + j1, cmul(alpha, c11));
    }
  }
  __syncthreads();
}

// Element (r, c) of the assembled generator for slice n of sample b.
__device__ __forceinline__ cplx h_elem(const ChainArgs& A, const cplx* h0b, const cplx* hkb,
                                       const double* sig, int n, int i, int j) {
  const int D = A.D;
  cplx h = h0b[(long)n * A.h0_nstride + i * D + j];
  for (int k = 0; k < A.K; ++k) {
    const double c = sig[(long)k * A.N + n];
    const cplx x = hkb[(long)k * D * D + i * D + j];
    h.x = fma(c, x.x, h.x);
    h.y = fma(c, x.y, h.y);
  }
  return h;
}

template <bool GLOBAL>
__global__ void __launch_bounds__(256) chain_kernel(ChainArgs A) {
  const int tid = threadIdx.x, nt = blockDim.x;
  const int wg = blockIdx.x;
  const int b = wg / A.S, seg = wg - b * A.S;
  const int n0 = seg * A.seg_len;
  const int n1 = min(A.N, n0 + A.seg_len);
  const int Dm = A.Dm, ld = A.ld, D = A.D;
  const int msz = ld * Dm;
  Mem<GLOBAL> M;
  M.g = GLOBAL ? A.scratch + (long)wg * A.scratch_stride : nullptr;
  __shared__ double red[264];
  __shared__ double2 red_mu;
  __shared__ int red_plan[4];

  int bX = 0, bA2 = msz, bA3 = 2 * msz, bA4 = 3 * msz, bP = 4 * msz, bT = 5 * msz, bU = 6 * msz;
  const cplx* h0b = A.h0 ? A.h0 + (long)b * A.h0_bstride : nullptr;
  const cplx* hkb = A.hks ? A.hks + (long)b * A.hks_bstride : nullptr;
  const double* sig = A.signals ? A.signals + (long)b * A.K * A.N : nullptr;
  const int nel = Dm * Dm;

  for (int n = n0; n < n1; ++n) {
    cplx alpha = cmake(1.0, 0.0);
    if (A.mode == C3P_MODE_GIVEN) {
      // E is given (ordered product of supplied matrices): load into P
      const cplx* src = A.mats + ((long)b * A.N + n) * nel;
      for (int e = tid; e < nel; e += nt) M.st(bP + (e / Dm) * ld + (e % Dm), src[e]);
      __syncthreads();
    } else {
      // ---- assemble X ----
      if (A.mode == C3P_MODE_UNITARY) {
        for (int e = tid; e < nel; e += nt) {
          const int i = e / Dm, j = e - i * Dm;
          const cplx h = h_elem(A, h0b, hkb, sig, n, i, j);
          M.st(bX + i * ld + j, cmake(h.y * A.dt, -h.x * A.dt));  // -i*dt*h
        }
      } else if (A.mode == C3P_MODE_LINDBLAD) {
        // row r = (i,j), col c = (k,l):  -i (H[i,k] d_jl - d_ik H[l,j]) + clp[r,c]
        for (int e = tid; e < nel; e += nt) {
          const int r = e / Dm, c = e - r * Dm;
          const int i = r / D, j = r - i * D, k = c / D, l = c - k * D;
          cplx v = A.clp[e];
          if (j == l) {
            const cplx h = h_elem(A, h0b, hkb, sig, n, i, k);
            v.x += h.y;
            v.y -= h.x;
          }
          if (i == k) {
            const cplx h = h_elem(A, h0b, hkb, sig, n, l, j);
            v.x -= h.y;
            v.y += h.x;
          }
          M.st(bX + r * ld + c, cscale(v, A.dt));
        }
      } else {  // C3P_MODE_EXPM: exponentiate supplied matrices
        const cplx* src = A.mats + ((long)b * A.N + n) * nel;
        for (int e = tid; e < nel; e += nt) M.st(bX + (e / Dm) * ld + (e % Dm), src[e]);
      }
      __syncthreads();
      // ---- trace shift: exp(X) = e^mu exp(X - mu I) ----
      if (tid == 0) {
        double tr = 0, ti = 0;
        for (int i = 0; i < Dm; ++i) {
          const cplx d = M.ld(bX + i * ld + i);
          tr += d.x;
          ti += d.y;
        }
        red_mu = make_double2(0.0, ti / Dm);  // imaginary shift only (c3p_smalld.hip: build_tables)
      }
      __syncthreads();
      const cplx mu = red_mu;
      for (int j = tid; j < Dm; j += nt) {
        const int o = bX + j * ld + j;
        M.st(o, csub(M.ld(o), mu));
      }
      __syncthreads();
      // ---- 1-norm, plan ----
      for (int j = tid; j < Dm; j += nt) {
        double s = 0;
        for (int i = 0; i < Dm; ++i) s += cabs1(M.ld(bX + i * ld + j));
        red[j] = s;
      }
      __syncthreads();
      if (tid == 0) {
        double nrm = 0;
        for (int j = 0; j < Dm; ++j) nrm = fmax(nrm, red[j]);
        const TaylorPlan p = c3p_pick_plan(nrm);
        red_plan[0] = p.m;
        red_plan[1] = p.q;
        red_plan[2] = p.r;
        red_plan[3] = p.s;
      }
      __syncthreads();
      const int q = red_plan[1], r = red_plan[2], s = red_plan[3];
      if (s > 0) {
        const double sc = ldexp(1.0, -s);
        for (int e = tid; e < nel; e += nt) {
          const int o = bX + (e / Dm) * ld + (e % Dm);
          M.st(o, cscale(M.ld(o), sc));
        }
        __syncthreads();
      }
      // ---- powers ----
      const cplx one = cmake(1.0, 0.0);
      mm(M, bA2, bX, bX, Dm, ld, one, tid, nt);
      if (q >= 3) mm(M, bA3, bA2, bX, Dm, ld, one, tid, nt);
      if (q >= 4) mm(M, bA4, bA2, bA2, Dm, ld, one, tid, nt);
      const int bAq = (q == 2) ? bA2 : (q == 3 ? bA3 : bA4);
      // ---- Horner over blocks: P = c_m X^q + B_{r-1}; P = P X^q + B_j ----
      for (int jb = r - 1; jb >= 0; --jb) {
        if (jb < r - 1) mm(M, bT, bP, bAq, Dm, ld, one, tid, nt);
        const double c0 = c3p_inv_fact[q * jb], c1 = c3p_inv_fact[q * jb + 1];
        const double c2 = q >= 3 ? c3p_inv_fact[q * jb + 2] : 0.0;
        const double c3 = q >= 4 ? c3p_inv_fact[q * jb + 3] : 0.0;
        const double c2q2 = q == 2 ? 0.0 : c2;
        const double cm = c3p_inv_fact[q * r];
        for (int e = tid; e < nel; e += nt) {
          const int i = e / Dm, j = e - i * Dm, o = i * ld + j;
          cplx v;
          if (jb == r - 1)
            v = cscale(M.ld(bAq + o), cm);
          else
            v = M.ld(bT + o);
          if (i == j) v.x += c0;
          const cplx x1 = M.ld(bX + o);
          v.x = fma(c1, x1.x, v.x);
          v.y = fma(c1, x1.y, v.y);
          if (q >= 3) {
            const cplx x2 = M.ld(bA2 + o);
            v.x = fma(c2q2, x2.x, v.x);
            v.y = fma(c2q2, x2.y, v.y);
          }
          if (q >= 4) {
            const cplx x3 = M.ld(bA3 + o);
            v.x = fma(c3, x3.x, v.x);
            v.y = fma(c3, x3.y, v.y);
          }
          M.st(bP + o, v);
        }
        __syncthreads();
      }
      // ---- squarings ----
      for (int it = 0; it < s; ++it) {
        mm(M, bT, bP, bP, Dm, ld, one, tid, nt);
        const int tmp = bP;
        bP = bT;
        bT = tmp;
      }
      // e^{mu 2^s}: mu was taken before scaling by 2^-s, so the factor is e^{mu}
      double sn, cs;
      sincos(mu.y, &sn, &cs);
      const double er = exp(mu.x);
      alpha = cmake(er * cs, er * sn);
    }
    // ---- optional dU write-out ----
    if (A.dUs_out) {
      cplx* dst = A.dUs_out + ((long)b * A.N + n) * nel;
      for (int e = tid; e < nel; e += nt) dst[e] = cmul(alpha, M.ld(bP + (e / Dm) * ld + (e % Dm)));
    }
    // ---- chain ----
    if (n == n0) {
      for (int e = tid; e < nel; e += nt) {
        const int o = (e / Dm) * ld + (e % Dm);
        M.st(bU + o, cmul(alpha, M.ld(bP + o)));
      }
      __syncthreads();
    } else {
      if (A.right_order)
        mm(M, bT, bU, bP, Dm, ld, alpha, tid, nt);
      else
        mm(M, bT, bP, bU, Dm, ld, alpha, tid, nt);
      const int tmp = bU;
      bU = bT;
      bT = tmp;
    }
  }
  // ---- write the segment product (row phases on the final pass) ----
  cplx* dst = A.seg_out + ((long)b * A.S + seg) * nel;
  const double* ph = A.fr_phase ? A.fr_phase + (long)b * Dm : nullptr;
  for (int e = tid; e < nel; e += nt) {
    const int i = e / Dm, j = e - i * Dm;
    cplx v = M.ld(bU + i * ld + j);
    if (ph) {
      double sn, cs;
      sincos(ph[i], &sn, &cs);
      v = cmul(cmake(cs, sn), v);
    }
    dst[e] = v;
  }
}

// clp[(i,j),(k,l)] = sum_c C[i,k] conj(C[j,l]) - 1/2 (C^+C)[i,k] d_jl - 1/2 d_ik sum_m C[m,j] conj(C[m,l])
// (the slice-independent dissipator, propagation.py:570-581)
__global__ void clp_kernel(const cplx* col, int C, int D, cplx* clp) {
  const int Dm = D * D;
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (long)Dm * Dm) return;
  const int r = e / Dm, c = e - (long)r * Dm;
  const int i = r / D, j = r - i * D, k = c / D, l = c - k * D;
  cplx v = cmake(0, 0);
  for (int ci = 0; ci < C; ++ci) {
    const cplx* Cm = col + (long)ci * D * D;
    cfma(v, Cm[i * D + k], cconj(Cm[j * D + l]));
    if (j == l) {
      cplx s = cmake(0, 0);
      for (int m = 0; m < D; ++m) cfma(s, cconj(Cm[m * D + i]), Cm[m * D + k]);
      v.x -= 0.5 * s.x;
      v.y -= 0.5 * s.y;
    }
    if (i == k) {
      cplx s = cmake(0, 0);
      for (int m = 0; m < D; ++m) cfma(s, Cm[m * D + j], cconj(Cm[m * D + l]));
      v.x -= 0.5 * s.x;
      v.y -= 0.5 * s.y;
    }
  }
  clp[e] = v;
}

// out[n,(i*Db+p),(j*Db+q)] = A[n,i,j] * B[n,p,q]   (tf_kron, tf_utils.py:257-267)
// which: 0 kron(A,B); 1 spre(A)=A(x)I; 2 spost(A)=I(x)A^T; 3 super(A)=A(x)conj(A)
__global__ void kron_kernel(const cplx* A, const cplx* Bm, int Da, int Db, int which, long total,
                            cplx* out) {
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  const int Dm = Da * Db;
  const long n = e / ((long)Dm * Dm);
  const long rem = e - n * (long)Dm * Dm;
  const int r = rem / Dm, c = rem - (long)r * Dm;
  const int i = r / Db, p = r - i * Db, j = c / Db, q = c - j * Db;
  const cplx* An = A + n * Da * Da;
  cplx v;
  if (which == 0) {
    v = cmul(An[i * Da + j], Bm[n * Db * Db + p * Db + q]);
  } else if (which == 1) {
    v = (p == q) ? An[i * Da + j] : cmake(0, 0);
  } else if (which == 2) {
    v = (i == j) ? An[q * Da + p] : cmake(0, 0);
  } else {
    v = cmul(An[i * Da + j], cconj(An[p * Da + q]));
  }
  out[e] = v;
}

// s[b] = tr(P^T U[b] P G^+) = sum_{a,c} U[b][rows[a]][rows[c]] conj(G[a][c]): the one number both
// unitary_infid and average_infid need (fidelities.py:154-184,290-313; tf_utils.py:330-436).
// One wavefront per sample.
__global__ void __launch_bounds__(64) overlap_kernel(const cplx* U, int D, const int* rows, int L,
                                                     const cplx* ideal, cplx* out) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const cplx* Ub = U + (long)b * D * D;
  double sr = 0.0, si = 0.0;
  for (int e = lane; e < L * L; e += 64) {
    const int a = e / L, c = e - a * L;
    const cplx u = Ub[rows[a] * D + rows[c]];
    const cplx g = ideal[e];
    sr += u.x * g.x + u.y * g.y;
    si += u.y * g.x - u.x * g.y;
  }
  for (int o = 32; o >= 1; o >>= 1) {
    sr += __shfl_xor(sr, o);
    si += __shfl_xor(si, o);
  }
  if (lane == 0) out[b] = cmake(sr, si);
}

// Fused goal epilogue: infid[b] from the overlap (kind 0: unitary_infid = 1 - |s / L|^2, fidelities.py:154-184; kind 1:
// average_infid = 1 - (|s|^2 / L + 1) / (L + 1), fidelities.py:290-313) and the partial sum of the block's samples --
// what an optimiser (or the all-reduce of a sharded batch, optimalcontrol_robust.py:49-70) needs instead of B matrices.
// One wavefront per sample slot, 4 per block; partial[blockIdx] = sum over the block's samples in a fixed order.
__global__ void __launch_bounds__(256) infid_kernel(const cplx* U, int B, int D, const int* rows, int L, const cplx* ideal,
                                                    int kind, double* infid, double* partial) {
  __shared__ double wsum[4];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  double acc = 0.0;
  for (long b = (long)blockIdx.x * 4 + wave; b < B; b += (long)gridDim.x * 4) {
    const cplx* Ub = U + b * D * D;
    double sr = 0.0, si = 0.0;
    for (int e = lane; e < L * L; e += 64) {
      const int a = e / L, c = e - a * L;
      const cplx u = Ub[rows[a] * D + rows[c]];
      const cplx g = ideal[e];
      sr += u.x * g.x + u.y * g.y;
      si += u.y * g.x - u.x * g.y;
    }
    for (int o = 32; o >= 1; o >>= 1) {
      sr += __shfl_xor(sr, o);
      si += __shfl_xor(si, o);
    }
    const double s2 = sr * sr + si * si;
    const double f = kind == 0 ? 1.0 - s2 / ((double)L * L) : 1.0 - (s2 / L + 1.0) / (L + 1.0);
    if (lane == 0 && infid) infid[b] = f;
    acc += f;
  }
  if (lane == 0) wsum[wave] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
}
__global__ void __launch_bounds__(64) infid_sum_kernel(const double* partial, int n, int B, double* out) {
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += 64) s += partial[i];
  for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o);
  if (threadIdx.x == 0) {
    out[0] = s;
    out[1] = (double)B;
  }
}

// Pre-pass of the supplied-generator modes (branch B of pwc, c3p_expm): one workgroup per matrix,
// X = coef * H;  meta = {Re mu, Im mu, ||X - mu I||_1, 0},  mu = tr X / D.
__global__ void __launch_bounds__(64) hmeta_kernel(const cplx* hs, long bstride, int N, int D, double cr, double ci,
                                                   double* meta) {
  __shared__ double rr[64], ri[64];
  __shared__ double mu[2];
  const int tid = threadIdx.x;
  const long m = blockIdx.x;
  const long b = m / N, n = m - b * N;
  const cplx* h = hs + b * bstride + n * (long)D * D;
  double tr = 0, ti = 0;
  for (int i = tid; i < D; i += 64) {
    const cplx x = h[i * D + i];
    tr += cr * x.x - ci * x.y;
    ti += cr * x.y + ci * x.x;
  }
  rr[tid] = tr;
  ri[tid] = ti;
  __syncthreads();
  if (tid == 0) {
    double a = 0, c = 0;
    for (int i = 0; i < 64; ++i) {
      a += rr[i];
      c += ri[i];
    }
    mu[0] = 0.0;  // imaginary shift only (c3p_smalld.hip: build_tables)
    mu[1] = c / D;
  }
  __syncthreads();
  double cs = 0;
  for (int j = tid; j < D; j += 64) {
    double sum = 0;
    for (int i = 0; i < D; ++i) {
      const cplx x = h[i * D + j];
      double vr = cr * x.x - ci * x.y, vi = cr * x.y + ci * x.x;
      if (i == j) {
        vr -= mu[0];
        vi -= mu[1];
      }
      sum += hypot(vr, vi);
    }
    cs = fmax(cs, sum);
  }
  double rs = 0;  // largest row sum: ||X - mu||_inf = the 1-norm of X^H (backward sweeps of general generators)
  for (int i = tid; i < D; i += 64) {
    double sum = 0;
    for (int j = 0; j < D; ++j) {
      const cplx x = h[i * D + j];
      double vr = cr * x.x - ci * x.y, vi = cr * x.y + ci * x.x;
      if (i == j) {
        vr -= mu[0];
        vi -= mu[1];
      }
      sum += hypot(vr, vi);
    }
    rs = fmax(rs, sum);
  }
  rr[tid] = cs;
  ri[tid] = rs;
  __syncthreads();
  if (tid == 0) {
    double nrm = 0, nri = 0;
    for (int i = 0; i < 64; ++i) {
      nrm = fmax(nrm, rr[i]);
      nri = fmax(nri, ri[i]);
    }
    double* o = meta + m * 4;
    o[0] = mu[0];
    o[1] = mu[1];
    o[2] = nrm;
    o[3] = nri;
  }
}

// Small matrices (D <= 16): one wavefront per matrix, four per workgroup, coalesced 16-byte loads staged in LDS
// (the one-workgroup-per-matrix kernel above spends its time on launch slots and strided column reads: 0.36 ms for
// the 256k 9x9 generators of a cfg2 batch, as much as the exponentials themselves).
__global__ void __launch_bounds__(256) hmeta_small_kernel(const cplx* hs, long bstride, int N, int D, double cr, double ci,
                                                          double* meta, long nmat) {
  __shared__ cplx stage[4][256];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const long m = (long)blockIdx.x * 4 + w;
  if (m >= nmat) return;  // whole wavefront
  const long b = m / N, n = m - b * N;
  const cplx* h = hs + b * bstride + n * (long)D * D;
  const int DD = D * D;
  double tr = 0.0, ti = 0.0;
  for (int e = lane; e < DD; e += 64) {
    const cplx x = h[e];
    const cplx v = cmake(cr * x.x - ci * x.y, cr * x.y + ci * x.x);
    stage[w][e] = v;
    if (e / D == e % D) {
      tr += v.x;
      ti += v.y;
    }
  }
  for (int o = 32; o > 0; o >>= 1) {
    tr += __shfl_xor(tr, o);
    ti += __shfl_xor(ti, o);
  }
  const double mur = 0.0, mui = ti / D;  // imaginary shift only (c3p_smalld.hip: build_tables)
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  double cs = 0.0;
  if (lane < D)
    for (int i = 0; i < D; ++i) {
      cplx v = stage[w][i * D + lane];
      if (i == lane) {
        v.x -= mur;
        v.y -= mui;
      }
      cs += hypot(v.x, v.y);
    }
  double rs = 0.0;  // row sums: ||X - mu||_inf = the 1-norm of X^H
  if (lane < D)
    for (int j = 0; j < D; ++j) {
      cplx v = stage[w][lane * D + j];
      if (j == lane) {
        v.x -= mur;
        v.y -= mui;
      }
      rs += hypot(v.x, v.y);
    }
  for (int o = 32; o > 0; o >>= 1) {
    cs = fmax(cs, __shfl_xor(cs, o));
    rs = fmax(rs, __shfl_xor(rs, o));
  }
  if (lane == 0) {
    double* o4 = meta + m * 4;
    o4[0] = mur;
    o4[1] = mui;
    o4[2] = cs;
    o4[3] = rs;
  }
}

}  // namespace

// ---------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------
size_t c3p_generic_lds_bytes(int Dm) {
  const int ld = Dm | 1;
  return (size_t)7 * ld * Dm * sizeof(cplx);
}

int c3p_generic_threads(int Dm) {
  const int tn = (Dm + 1) / 2;
  int t = ((tn * tn + 63) / 64) * 64;
  if (t > 256) t = 256;
  if (t < 64) t = 64;
  return t;
}

hipError_t c3p_launch_chain_generic(const ChainArgs& A, bool global_scratch, hipStream_t st) {
  const int threads = c3p_generic_threads(A.Dm);
  const dim3 grid((unsigned)((long)A.B * A.S));
  if (global_scratch) {
    C3P_LAUNCH(chain_kernel<true>, grid, dim3(threads), 0, st, A);
  } else {
    const size_t lds = c3p_generic_lds_bytes(A.Dm);
    // per launch: the attribute is per DEVICE, and one process may drive several GPUs
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&chain_kernel<false>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 4096);
    if (e != hipSuccess) return e;
    C3P_LAUNCH(chain_kernel<false>, grid, dim3(threads), lds, st, A);
  }
  return hipGetLastError();
}

hipError_t c3p_launch_clp(const cplx* col, int C, int D, cplx* clp, hipStream_t st) {
  const long total = (long)D * D * D * D;
  C3P_LAUNCH(clp_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, col, C, D,
                     clp);
  return hipGetLastError();
}

hipError_t c3p_launch_kron(const cplx* A, const cplx* Bm, int n, int Da, int Db, int which,
                           cplx* out, hipStream_t st) {
  const long Dm = (long)Da * Db;
  const long total = (long)n * Dm * Dm;
  if (total == 0) return hipSuccess;
  C3P_LAUNCH(kron_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, A, Bm, Da,
                     Db, which, total, out);
  return hipGetLastError();
}

int c3p_infid_blocks(int B) {
  const int nb = (B + 3) / 4;
  return nb < 256 ? nb : 256;
}
hipError_t c3p_launch_infid(const cplx* U, int B, int D, const int* rows, int L, const cplx* ideal, int kind, double* infid,
                            double* partial, double* sum_out, hipStream_t st) {
  const int nb = c3p_infid_blocks(B);
  C3P_LAUNCH(infid_kernel, dim3((unsigned)nb), dim3(256), 0, st, U, B, D, rows, L, ideal, kind, infid, partial);
  if (sum_out) C3P_LAUNCH(infid_sum_kernel, dim3(1), dim3(64), 0, st, partial, nb, B, sum_out);
  return hipGetLastError();
}

hipError_t c3p_launch_overlap(const cplx* U, int B, int D, const int* rows, int L, const cplx* ideal,
                              cplx* out, hipStream_t st) {
  if (B == 0) return hipSuccess;
  C3P_LAUNCH(overlap_kernel, dim3((unsigned)B), dim3(64), 0, st, U, D, rows, L, ideal, out);
  return hipGetLastError();
}

hipError_t c3p_launch_hmeta(const cplx* hs, long bstride, long nmat, int N, int D, double cr, double ci, double* meta,
                            hipStream_t st) {
  if (nmat == 0) return hipSuccess;
  if (D <= 16) {
    C3P_LAUNCH(hmeta_small_kernel, dim3((unsigned)((nmat + 3) / 4)), dim3(256), 0, st, hs, bstride, N, D, cr, ci, meta,
                       nmat);
    return hipGetLastError();
  }
  C3P_LAUNCH(hmeta_kernel, dim3((unsigned)nmat), dim3(64), 0, st, hs, bstride, N, D, cr, ci, meta);
  return hipGetLastError();
}
