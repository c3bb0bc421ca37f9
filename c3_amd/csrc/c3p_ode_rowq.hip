// Lane-row ODE vector-state solver for 17 <= D <= 48 (round 3): Schroedinger psi and the columns of rk4_unitary at the
// dimensions of BASELINE cfg3 (D = 27) and cfg5 (D = 36).  Same arithmetic and same reference functions as
// c3p_ode_row.hip (c3/libraries/propagation.py:687-883,897-899; c3/model.py:641-697; c3/utils/tf_utils.py:521-559),
// the D <= 16 mapping stretched over several 16-lane DPP rows:
//
//  * a sample owns NQ' = 2 (D <= 32) or 4 DPP rows of a wavefront (2 or 1 samples per wave); lane (q, i) owns matrix
//    row 16 q + i: its row of H(t) in registers, element 16 q + i of the state;
//  * the matrix-vector product still runs on `v_fmac_f64_dpp row_newbcast:j`: every lane keeps the NQ state elements
//    {i, 16 + i, ...} of its column position, so lane j of the OWN row can broadcast y_(16 g + j) for every column group g;
//    after a stage the new elements are exchanged between the rows of a sample through LDS (one 16-byte write, NQ reads);
//  * the operator rows do not fit the registers: h0 and the hk of the workgroup sit in LDS (odd row stride) and
//    H(t) = h0 + sum_k c_k(t) hk is NOT re-assembled per stage node.  The control amplitudes are piecewise linear in t
//    (linear interpolation between samples, tf_utils.py:557-559), so between two samples H(t) moves along
//    dH = sum_k (c_k[m+1] - c_k[m]) hk: one assembly of dH per sample interval (K LDS reads and FMAs per element) and ONE
//    FMA per element and stage node, H += dtheta dH; H is re-anchored exactly (h0 + sum_k c_k hk) every 14 steps so the
//    increments do not accumulate rounding.  (Extrapolation past the last sample continues the last interval, as the
//    reference's fill_value="extrapolate" does.)
#include <type_traits>
#include <utility>

#include "c3p_common.h"
#include "c3p_ode.h"
#include "c3p_ode_tab.h"
#include "c3p_ode_dpp.inc"

extern __shared__ __attribute__((aligned(16))) unsigned char c3p_ode_rowq_smem[];

namespace {

__constant__ OdeTableau c3p_rowq_tab[4] = C3P_ODE_TABLEAUX;

constexpr int QKMAX = 8;  // control lines: instances for <= 4 and <= 8 (operators in LDS: c3p_ode_rowq_supported bounds K D^2)
__host__ __device__ constexpr int q_lines(int K) { return K <= 4 ? 4 : QKMAX; }

struct QLds {
  int ld;  // operator row stride in elements (odd: b128 / b64 rows of consecutive lanes fall on different banks)
  int ops_off, sig_off, xch_off, k_off;
  size_t bytes;
};
__host__ __device__ inline QLds q_lds(int D, int NC, int K, bool realh, int nw, int solver) {
  QLds q;
  q.ld = NC | 1;
  q.ops_off = 0;
  const int ops = (1 + K) * (D + 1) * q.ld * (realh ? 8 : 16);  // one zero row per operator for the padding lanes
  q.sig_off = (ops + 15) & ~15;
  q.xch_off = q.sig_off + nw * 2 * q_lines(K) * 16 * 8;
  q.k_off = q.xch_off + nw * 64 * 16;
  const int stages = solver == 0 ? 0 : (solver == 1 ? 4 : 7);  // rk4: the previous stage stays in registers
  q.bytes = (size_t)q.k_off + (size_t)stages * nw * 64 * 16;
  return q;
}

template <int I, int N_, class F>
__device__ __forceinline__ void qstatic_for(F&& f) {
  if constexpr (I < N_) {
    f(std::integral_constant<int, I>{});
    qstatic_for<I + 1, N_>(f);
  }
}
__device__ __forceinline__ bool quniform(bool c) { return __builtin_amdgcn_readfirstlane((int)c) != 0; }

template <int C_>
__device__ __forceinline__ const double (&head(const double (&a)[16]))[C_] {
  return *reinterpret_cast<const double(*)[C_]>(&a);
}

__device__ __forceinline__ int q_chunk_base(int n0, int us, int N) {
  int b = us * n0;
  if (b > N - 2) b = N - 2;
  return b < 0 ? 0 : b;
}

template <int NQ, int CL, bool REALH, int NW, int QK = 4>
__global__ void __launch_bounds__(64 * NW) ode_vecq_kernel(OdeArgs A) {
  constexpr int NQP = (NQ == 2) ? 2 : 4;  // DPP rows per sample
  constexpr int SPW = 4 / NQP;            // samples per wavefront
  constexpr int NC = 16 * (NQ - 1) + CL;  // columns held per row (>= D, zero padded)
  const OdeTableau& T = c3p_rowq_tab[A.solver];
  const int S = T.stages;
  const bool subdiag = A.solver == 0;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int slot = lane / (16 * NQP), q = (lane >> 4) % NQP, i = lane & 15;
  const int D = A.D, K = A.K, N = A.N, M = A.M, us = A.u_stride;
  const QLds L = q_lds(D, NC, K, REALH, NW, A.solver);
  const int LD = L.ld;
  double* opsR = reinterpret_cast<double*>(c3p_ode_rowq_smem + L.ops_off);
  cplx* opsC = reinterpret_cast<cplx*>(c3p_ode_rowq_smem + L.ops_off);
  double* sig = reinterpret_cast<double*>(c3p_ode_rowq_smem + L.sig_off) + (wave * 2 + slot) * QK * 16;
  cplx* xch = reinterpret_cast<cplx*>(c3p_ode_rowq_smem + L.xch_off) + wave * 64;
  cplx* kst = reinterpret_cast<cplx*>(c3p_ode_rowq_smem + L.k_off);  // [stage][wave][lane]

  // operators -> LDS (zero padded columns, one zero row); real operators take the REALH instance
  bool im0 = true;
  for (int e = tid; e < (1 + K) * (D + 1) * LD; e += 64 * NW) {
    const int k = e / ((D + 1) * LD), rr = (e / LD) % (D + 1), cc = e % LD;
    cplx z = cmake(0, 0);
    if (rr < D && cc < D) z = (k == 0) ? A.h0[rr * D + cc] : A.hks[((long)(k - 1) * D + rr) * D + cc];
    im0 = im0 && (z.y == 0.0);
    if constexpr (REALH)
      opsR[e] = z.x;
    else
      opsC[e] = z;
  }
  const bool allreal = __syncthreads_and((int)im0) != 0;
  if (allreal != REALH) return;

  const long nv = (long)A.B * M;
  long v = ((long)blockIdx.x * NW + wave) * SPW + slot;
  bool live = v < nv;
  if (!live) v = nv - 1;
  int b = (int)(v / M), col = (int)(v - (long)b * M);
  // trajectory pieces (OdeArgs.seg_traj, M = 1): a WAVEFRONT integrates one time segment (its samples share the absolute step
  // index, which the chunked control amplitudes and the piecewise-linear advance of H assume), segment-major order
  int seg = 0, n_begin = 0, n_end = A.n_steps;
  if (A.seg_traj) {
    const long vw = (long)blockIdx.x * NW + wave;
    const long wps = (A.B + SPW - 1) / SPW;  // wavefronts per segment
    long sg_ = vw / wps;
    live = sg_ < A.seg_count;
    if (!live) sg_ = A.seg_count - 1;
    seg = __builtin_amdgcn_readfirstlane((int)sg_);
    const long bb = (vw - (vw / wps) * wps) * SPW + slot;
    live = live && bb < A.B;
    b = (int)(bb < A.B ? bb : A.B - 1);
    col = 0;
    n_begin = seg * A.seg_len;
    n_end = n_begin + A.seg_len < A.n_steps ? n_begin + A.seg_len : A.n_steps;
  }
  const int rowi = 16 * q + i;
  const bool rowok = (q < NQ) && rowi < D;
  const int lrow = rowok ? rowi : D;  // padding lanes read the zero row
  const cplx* init = A.init + (A.seg_traj ? (long)b * A.seg_count + seg : (long)b) * A.init_bstride;
  double pr = 0.0, pi = 0.0;
  if (rowok) {
    const cplx z = init[(long)rowi * M + col];
    pr = z.x;
    pi = z.y;
  }
  const double ir = pr, ii = pi;
  const double dt = A.dt;
  const double* sg = A.signals + (long)b * K * N;
  const int SPC = 14 / us;
  const bool loader = (q == 0);  // lanes that fetch the sample's control amplitudes (16 per chunk and control line)
  double pre[QK];
  {
    const int base = q_chunk_base(n_begin, us, N);
    int idx = base + i;
    if (idx > N - 1) idx = N - 1;
#pragma unroll
    for (int k = 0; k < QK; ++k) pre[k] = (k < K && loader) ? sg[(long)k * N + idx] : 0.0;
  }
  double Hr[NQ][16], Hi[REALH ? 1 : NQ][16];  // H(t), row of this lane (column group g, column 16 g + j)
  double Gr[NQ][16], Gi[REALH ? 1 : NQ][16];  // dH of the current sample interval
  const long ssz = (long)D * M;
  const long eo = A.transpose_out ? (long)col * D + rowi : (long)rowi * M + col;
  cplx* outp = A.states + (long)b * (A.want_all ? (long)A.n_steps : 1) * ssz + eo;

  // row of (coef0 h0 + sum_k cf[k] hk) from LDS
  auto assemble = [&](double (&Xr)[NQ][16], double (&Xi)[REALH ? 1 : NQ][16], double coef0, const double (&cf)[QK]) {
    for (int k1 = 0; k1 <= K; ++k1) {  // operator by operator: the coefficient is a scalar, the row elements stay in registers
      double ck = coef0;
      if (k1 == 1) ck = cf[0];
      if (k1 == 2) ck = cf[1];
      if (k1 == 3) ck = cf[2];
      if (k1 == 4) ck = cf[3];
      if constexpr (QK > 4) {
        if (k1 == 5) ck = cf[4];
        if (k1 == 6) ck = cf[5];
        if (k1 == 7) ck = cf[6];
        if (k1 == 8) ck = cf[7];
      }
      const int ob = k1 * (D + 1) * LD + lrow * LD;
#pragma unroll
      for (int g = 0; g < NQ; ++g) {
#pragma unroll
        for (int j = 0; j < (g == NQ - 1 ? CL : 16); ++j) {
          if constexpr (REALH) {
            const double z = opsR[ob + 16 * g + j];
            Xr[g][j] = (k1 == 0) ? ck * z : fma(ck, z, Xr[g][j]);
          } else {
            const cplx z = opsC[ob + 16 * g + j];
            Xr[g][j] = (k1 == 0) ? ck * z.x : fma(ck, z.x, Xr[g][j]);
            Xi[g][j] = (k1 == 0) ? ck * z.y : fma(ck, z.y, Xi[g][j]);
          }
        }
      }
    }
  };

  double kpr = 0.0, kpi = 0.0;  // previous stage (all tableaux: a[s][s-1] is applied from registers)
  for (int n0 = n_begin; n0 < n_end; n0 += SPC) {
    const int base = q_chunk_base(n0, us, N);
    if (loader) {
#pragma unroll
      for (int k = 0; k < QK; ++k) sig[k * 16 + i] = pre[k];
    }
    {
      const int nb = q_chunk_base(n0 + SPC, us, N);
      int idx = nb + i;
      if (idx > N - 1) idx = N - 1;
#pragma unroll
      for (int k = 0; k < QK; ++k) pre[k] = (k < K && loader) ? sg[(long)k * N + idx] : 0.0;
    }
    // exact anchor at the first node of the chunk: H = h0 + sum_k c_k(u) hk, c_k by the reference's interpolation rule
    int cur_m = -1;
    {
      const double u = (double)n0 * (double)us;
      int lo = (int)floor(u);
      if (lo > N - 2) lo = N - 2;
      if (lo < 0) lo = 0;
      const double f = u - (double)lo;
      double cf[QK];
#pragma unroll
      for (int k = 0; k < QK; ++k) {
        const double y0 = sig[k * 16 + (lo - base)], y1 = sig[k * 16 + (lo - base) + 1];
        cf[k] = fma(f, y1 - y0, y0);
      }
      assemble(Hr, Hi, 1.0, cf);
    }
    // H += (u1 - u0) dH over the sample intervals [m, m + 1] that [u0, u1] crosses (the last interval extends to infinity)
    auto advance = [&](double u0, double u1) {
      while (quniform(u0 < u1)) {
        const int fl = __builtin_amdgcn_readfirstlane((int)floor(u0));
        const int m = fl > N - 2 ? N - 2 : fl;
        const double pend = (fl >= N - 1) ? u1 : fmin(u1, (double)(fl + 1));
        if (quniform(m != cur_m)) {
          double dc[QK];
#pragma unroll
          for (int k = 0; k < QK; ++k) dc[k] = sig[k * 16 + (m - base) + 1] - sig[k * 16 + (m - base)];
          assemble(Gr, Gi, 0.0, dc);
          cur_m = m;
        }
        const double w = pend - u0;
#pragma unroll
        for (int g = 0; g < NQ; ++g) {
#pragma unroll
          for (int j = 0; j < (g == NQ - 1 ? CL : 16); ++j) {
            Hr[g][j] = fma(w, Gr[g][j], Hr[g][j]);
            if constexpr (!REALH) Hi[g][j] = fma(w, Gi[g][j], Hi[g][j]);
          }
        }
        u0 = pend;
      }
    };
    const int n1 = (n0 + SPC < n_end) ? n0 + SPC : n_end;
    for (int n = n0; n < n1; ++n) {
      double th_prev = 0.0;
      double Br = 0.0, Bi = 0.0;  // running sum_j b_j k_j
      for (int s = 0; s < S; ++s) {
        const double th = T.node[s];
        if (quniform(th != th_prev)) advance(((double)n + th_prev) * (double)us, ((double)n + th) * (double)us);
        th_prev = th;
        // stage argument, own element
        double yr = pr, yi = pi;
        if (s > 0) {
          const double a = T.a[s][s - 1];
          yr = fma(a, kpr, yr);
          yi = fma(a, kpi, yi);
          if (!subdiag) {
            for (int j = 0; j + 1 < s; ++j) {
              const double aj = T.a[s][j];
              if (aj != 0.0) {
                const cplx kk = kst[(j * NW + wave) * 64 + lane];
                yr = fma(aj, kk.x, yr);
                yi = fma(aj, kk.y, yi);
              }
            }
          }
        }
        // every lane needs the elements of its column position from all rows of the sample
        xch[lane] = cmake(yr, yi);
        double w[4] = {0.0, 0.0, 0.0, 0.0};
        qstatic_for<0, NQ>([&](auto gc) {
          constexpr int g = decltype(gc)::value;
          const cplx yg = xch[slot * 16 * NQP + 16 * g + i];
          if constexpr (REALH) {
            if constexpr (g == NQ - 1 && CL < 16)
              OdeDpp<CL>::matvec_r(w, yg.x, yg.y, head<CL>(Hr[g]));
            else
              OdeDpp<16>::matvec_r(w, yg.x, yg.y, Hr[g]);
          } else {
            if constexpr (g == NQ - 1 && CL < 16)
              OdeDpp<CL>::matvec_c(w, yg.x, yg.y, head<CL>(Hr[g]), head<CL>(Hi[g]));
            else
              OdeDpp<16>::matvec_c(w, yg.x, yg.y, Hr[g], Hi[g]);
          }
        });
        const double wr = w[0] + w[2], wi = w[1] + w[3];
        kpr = dt * wi;  // -i dt (wr + i wi)
        kpi = -dt * wr;
        const double bs = T.b[s];
        Br = fma(bs, kpr, Br);
        Bi = fma(bs, kpi, Bi);
        if (!subdiag && s + 2 < S) kst[(s * NW + wave) * 64 + lane] = cmake(kpr, kpi);
      }
      // the last node of every tableau is t + dt: H is already the first node's of the next step
      pr += Br;
      pi += Bi;
      if (A.want_all && live && rowok) outp[(long)n * ssz] = cmake(pr, pi);
      if (A.reset_each_step) {
        pr = ir;
        pi = ii;
      }
    }
  }
  if (!A.want_all && live && rowok) outp[0] = cmake(pr, pi);
}

template <int NQ, int CL, int QK>
hipError_t launch_q2(const OdeArgs& A, hipStream_t st) {
  constexpr int NQP = (NQ == 2) ? 2 : 4;
  constexpr int SPW = 4 / NQP;
  constexpr int NC = 16 * (NQ - 1) + CL;
  // (trajectory pieces: seg_count x ceil(B / SPW) wavefronts, one segment per wavefront)
  const long nv = A.seg_traj ? (long)A.seg_count * ((A.B + SPW - 1) / SPW) * SPW : (long)A.B * A.M;
  hipError_t e;
  {
    // real operators: two wavefronts per SIMD (half the registers, half the LDS)
    constexpr int NW = 8;
    const QLds L = q_lds(A.D, NC, A.K, true, NW, A.solver);
    const dim3 grid((unsigned)((nv + NW * SPW - 1) / (NW * SPW)));
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&ode_vecq_kernel<NQ, CL, true, NW, QK>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024);
    if (e != hipSuccess) return e;
    C3P_LAUNCH((ode_vecq_kernel<NQ, CL, true, NW, QK>), grid, dim3(64 * NW), L.bytes, st, A);
  }
  {
    constexpr int NW = 4;
    const QLds L = q_lds(A.D, NC, A.K, false, NW, A.solver);
    const dim3 grid((unsigned)((nv + NW * SPW - 1) / (NW * SPW)));
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&ode_vecq_kernel<NQ, CL, false, NW, QK>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024);
    if (e != hipSuccess) return e;
    C3P_LAUNCH((ode_vecq_kernel<NQ, CL, false, NW, QK>), grid, dim3(64 * NW), L.bytes, st, A);
  }
  return hipGetLastError();
}

template <int NQ, int QK>
hipError_t launch_q1k(const OdeArgs& A, hipStream_t st) {
  const int cl = A.D - 16 * (NQ - 1);
  if (cl <= 4) return launch_q2<NQ, 4, QK>(A, st);
  if (cl <= 8) return launch_q2<NQ, 8, QK>(A, st);
  if (cl <= 12) return launch_q2<NQ, 12, QK>(A, st);
  return launch_q2<NQ, 16, QK>(A, st);
}
// up to four control lines: the instance of rounds 3 (its registers); five to eight: pre / cf / dc of eight lines
template <int NQ>
hipError_t launch_q1(const OdeArgs& A, hipStream_t st) {
  return A.K <= 4 ? launch_q1k<NQ, 4>(A, st) : launch_q1k<NQ, QKMAX>(A, st);
}

int q_groups(int D) { return (D + 15) / 16; }
int q_cols(int D) {
  const int nq = q_groups(D);
  const int cl = D - 16 * (nq - 1);
  return 16 * (nq - 1) + 4 * ((cl + 3) / 4);
}

}  // namespace

bool c3p_ode_rowq_supported(const OdeArgs& A) {
  if (c3p_opt_on(C3P_OPT_ode_wg)) return false;
  if (A.D < 17 || A.D > 48 || A.K > QKMAX || A.hs || A.N < 2) return false;
  if (A.u_stride != 1 && A.u_stride != 2) return false;
  if (A.step != C3P_STEP_SCHRODINGER_ID && A.step != C3P_STEP_PROPAGATOR_ID) return false;
  // both instances are launched (the operators are inspected on the device): both must fit the LDS
  const int nc = q_cols(A.D);
  return q_lds(A.D, nc, A.K, false, 4, A.solver).bytes <= (size_t)(150 * 1024) &&
         q_lds(A.D, nc, A.K, true, 8, A.solver).bytes <= (size_t)(150 * 1024);
}

hipError_t c3p_launch_ode_rowq(const OdeArgs& A, hipStream_t st) {
  switch (q_groups(A.D)) {
    case 2: return launch_q1<2>(A, st);
    case 3: return launch_q1<3>(A, st);
    default: return hipErrorInvalidValue;
  }
}
