// Lane-row ODE state solvers for D <= 16 (round 3): explicit Runge-Kutta integration of psi (Schroedinger), of the
// columns of a propagator (rk4_unitary) and of rho (von Neumann / Lindblad), register resident, no barriers.
//
// Stands in for ode_solver / ode_solver_final_state (c3/libraries/propagation.py:687-752), the tableaux rk4 / rk38 /
// rk5 / tsit5 (:755-883), the step functions schrodinger / von_neumann / lindblad (:886-904), Model.Hs_of_t
// (c3/model.py:641-697) with interpolate_signal (c3/utils/tf_utils.py:521-559) and the rk4_unitary family
// (propagation.py:71-101,221-255) -- same arithmetic as the workgroup-per-sample kernel of c3p_ode.hip (which stays the
// path for D > 48 and Lindblad steps above D = 32; 17 <= D <= 48 runs on
// c3p_ode_rowq.hip / c3p_ode_rhoq.hip), mapped to the machine differently:
//
//  * ONE SAMPLE PER 16-LANE DPP ROW, four samples per wavefront: lane i of the row owns ROW i of the Hamiltonian
//    (the rows of h0 and of every hk stay in registers for the whole integration; H(t) = h0 + sum_k c_k(t) hk is
//    assembled per stage node, 2 K D FMAs per lane) and element i of the state vector / row i of rho.
//  * The matrix-vector product needs every lane to see all of y: `v_fmac_f64_dpp ... row_newbcast:j` multiplies the own
//    H_ij by lane j's y_j in ONE instruction at the plain fp64 FMA rate -- no LDS round trip, no lane moves, no barrier
//    (c3p_ode_dpp.inc; measured in tools/ubench_dpp.hip).  4 D FMAs per lane and stage (2 D for real operators).
//  * rho-valued steps: (H rho)_i,: = sum_j H_ij rho_j,: (own scalar x broadcast row) and (rho H)_i,: = sum_j rho_ij H_j,:
//    (own scalar x broadcast row of H): the same primitive; the anticommutator of the Lindblad step is folded into
//    L = H - (i/2) G and R = H + (i/2) G (G = sum C^+ C), -i (L rho - rho R) = -i [H, rho] - {G, rho} / 2; the jump terms
//    C rho C^+ use the broadcast primitive for C rho and wave-uniform scalar operands (SGPR) for (.) C^+.
//  * Control amplitudes: 16 samples per row are fetched one chunk ahead (one coalesced 128-byte read per control line and
//    chunk) and parked in LDS; a stage reads its two neighbours from there.  `Hs` is never materialised.
//  * RK stages of the vector kernels live in registers (tableau = template parameter, zero coefficients vanish at compile
//    time); the rho kernels keep them in lane-private LDS slots (stage loop rolled: the code of one stage is large).
//  * The Hamiltonian of the last stage node (t + dt) IS the one of the next step's first node: carried, not rebuilt.
#include <type_traits>
#include <utility>

#include "c3p_common.h"
#include "c3p_ode.h"
#include "c3p_ode_tab.h"
#include "c3p_ode_dpp.inc"

extern __shared__ __attribute__((aligned(16))) unsigned char c3p_ode_row_smem[];

namespace {

__constant__ OdeTableau c3p_row_tab[4] = C3P_ODE_TABLEAUX;

__host__ __device__ constexpr OdeTableau tab_of(int solver) {
  constexpr OdeTableau t[4] = C3P_ODE_TABLEAUX;
  return t[solver];
}

// first stage after s whose node differs from its predecessor's (H is assembled again there), -1 if none
__host__ __device__ constexpr int next_asm_stage(int solver, int s) {
  const OdeTableau t = tab_of(solver);
  for (int k = s + 1; k < t.stages; ++k)
    if (t.node[k] != t.node[k - 1]) return k;
  return -1;
}

typedef const double __attribute__((address_space(4)))* cdouble_ptr;  // constant address space: uniform loads become s_load

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

// values of one chunk: indices [base, base + 15] of a control line, 14 / u_stride steps per chunk (a stage reads
// index lo and lo + 1 with lo <= base + u_stride * steps)
__device__ __forceinline__ int chunk_steps(int us) { return 14 / us; }
__device__ __forceinline__ int chunk_base(int n0, int us, int N) {
  int b = us * n0;
  if (b > N - 2) b = N - 2;
  return b < 0 ? 0 : b;
}

// ---------------------------------------------------------------------------------------------------------------
// vector state: Schroedinger psi (M = 1) or one column of a propagator (M = D virtual samples per sample)
// ---------------------------------------------------------------------------------------------------------------
// HS: H(t) is read from per-sample-index Hamiltonians hs [B?, N, D, D] instead of being assembled from h0 / hks -- the nearest
// sample (branch B of get_hs_of_t_ts, propagation.py:164-204, as the workgroup kernel of c3p_ode.hip takes it) or, hs_lerp, the
// linear interpolation of two consecutive samples (= branch A on Hamiltonians the library assembled itself: more than four
// control lines).  Lane i reads row i, D contiguous complex numbers per stage node.
template <int DP, int KT, int SOLVER, bool REALH, bool HS = false>
__global__ void __launch_bounds__(64) ode_vec_kernel(OdeArgs A) {
  static_assert(!HS || !REALH, "supplied Hamiltonians run on the complex instance");
  using P = OdeDpp<DP>;
  constexpr int S = tab_of(SOLVER).stages;
  __shared__ double sig[4 * KT * 16];
  const int lane = threadIdx.x, r = lane >> 4, i = lane & 15;
  const int D = A.D, K = A.K, N = A.N, M = A.M, us = A.u_stride;
  const int SG = A.seg_count > 0 ? A.seg_count : 1;  // time segments per (sample, column)
  const long nv = (long)A.B * SG * M;
  long v = (long)blockIdx.x * 4 + r;
  const bool live = v < nv;
  if (!live) v = nv - 1;
  const int b = (int)(v / ((long)SG * M)), seg = (int)((v / M) % SG), col = (int)(v % M);
  const int n_begin = A.seg_count > 0 ? seg * A.seg_len : 0;
  const int n_end = A.seg_count > 0 ? (n_begin + A.seg_len < A.n_steps ? n_begin + A.seg_len : A.n_steps) : A.n_steps;
  const bool row = i < D;

  // rows of the operators (zero padded to DP columns; lanes i >= D hold zero rows)
  double h0r[DP], h0i[REALH ? 1 : DP], hkr[KT][DP], hki[KT][REALH ? 1 : DP];
  bool im0 = true;
#pragma unroll
  for (int j = 0; j < DP; ++j) {
    cplx z = cmake(0, 0);
    if (!HS && row && j < D) z = A.h0[i * D + j];
    h0r[j] = z.x;
    if constexpr (!REALH) h0i[j] = z.y;
    im0 = im0 && (z.y == 0.0);
#pragma unroll
    for (int k = 0; k < KT; ++k) {
      cplx zk = cmake(0, 0);
      if (!HS && row && j < D && k < K) zk = A.hks[((long)k * D + i) * D + j];
      hkr[k][j] = zk.x;
      if constexpr (!REALH) hki[k][j] = zk.y;
      im0 = im0 && (zk.y == 0.0);
    }
  }
  // real operators take the REALH instance, everything else the complex one (both are launched; wave-uniform exit)
  if constexpr (!HS)
    if ((__all(im0) != 0) != REALH) return;

  const cplx* init = A.init + (A.seg_traj ? (long)b * SG + seg : (long)b) * A.init_bstride;
  double pr = 0.0, pi = 0.0;
  if (row) {
    const cplx z = init[(long)i * M + col];
    pr = z.x;
    pi = z.y;
  }
  const double ir = pr, ii = pi;
  const double dt = A.dt;
  const double* sg = A.signals + (long)b * K * N;
  const cplx* hrow = HS ? A.hs + (long)b * A.hs_bstride + (long)(row ? i : 0) * D : nullptr;  // row i of sample index 0
  const int SPC = chunk_steps(us);
  double pre[KT];
  if constexpr (!HS) {
    const int base = chunk_base(n_begin, us, N);
    int idx = base + i;
    if (idx > N - 1) idx = N - 1;
#pragma unroll
    for (int k = 0; k < KT; ++k) pre[k] = (k < K) ? sg[(long)k * N + idx] : 0.0;
  }
  double Hr[DP], Hi[REALH ? 1 : DP];
  const long ssz = (long)D * M;
  const long eo = A.transpose_out ? (long)col * D + i : (long)i * M + col;
  cplx* outp = A.states + (A.seg_traj ? (long)b : (long)b * SG + seg) * (A.want_all ? (long)A.n_steps : 1) * ssz + eo;

  // The step loops run over the RELATIVE step index (wave-uniform: the four rows of a wavefront may integrate different
  // segments, i.e. different absolute steps n = n_begin + j; a row past its segment end keeps its state)
  const int nrel = A.seg_count > 0 ? A.seg_len : A.n_steps;
  for (int j0 = 0; j0 < nrel; j0 += SPC) {
    const int n0 = n_begin + j0;
    const int base = chunk_base(n0, us, N);
    if constexpr (!HS) {
#pragma unroll
      for (int k = 0; k < KT; ++k) sig[(r * KT + k) * 16 + i] = pre[k];
      const int nb = chunk_base(n0 + SPC, us, N);
      int idx = nb + i;
      if (idx > N - 1) idx = N - 1;
#pragma unroll
      for (int k = 0; k < KT; ++k) pre[k] = (k < K) ? sg[(long)k * N + idx] : 0.0;
    }
    auto assemble = [&](double theta, int n, double theta_next, int n_next) {
      // linear interpolation of the control amplitudes, linear extrapolation past the last sample (tf_utils.py:557-559)
      const double u = ((double)n + theta) * (double)us;
      if constexpr (HS) {
        // (requesting the rows one stage ahead into registers that live across the stage loop was measured: 1.7x slower)
        // (the two modes are separate straight-line loops: a per-element choice between them serialises the loads, 1.4 - 2.5x)
        const long hsz = (long)D * D;
        if (A.hs_lerp) {
          int lo = (int)floor(u);
          if (lo > N - 2) lo = N - 2;
          if (lo < 0) lo = 0;
          const double f = u - (double)lo;
          const cplx* p0 = hrow + (long)lo * hsz;
          const cplx* p1 = p0 + (N > 1 ? hsz : 0);
#pragma unroll
          for (int j = 0; j < DP; ++j) {
            cplx z0 = cmake(0, 0), z1 = cmake(0, 0);
            if (row && j < D) z0 = p0[j], z1 = p1[j];
            Hr[j] = fma(f, z1.x - z0.x, z0.x);
            if constexpr (!REALH) Hi[j] = fma(f, z1.y - z0.y, z0.y);
          }
        } else {
          int iu = (int)(u + 0.5);  // stage positions are integer sample indices there
          if (iu > N - 1) iu = N - 1;
          if (iu < 0) iu = 0;
          const cplx* p0 = hrow + (long)iu * hsz;
#pragma unroll
          for (int j = 0; j < DP; ++j) {
            cplx z0 = cmake(0, 0);
            if (row && j < D) z0 = p0[j];
            Hr[j] = z0.x;
            if constexpr (!REALH) Hi[j] = z0.y;
          }
        }
        (void)theta_next;
        (void)n_next;
        return;
      }
      int lo = (int)floor(u);
      if (lo > N - 2) lo = N - 2;
      if (lo < 0) lo = 0;
      const double f = u - (double)lo;
      int li = lo - base;
      li = li < 0 ? 0 : (li > 14 ? 14 : li);  // (only a row past its segment end can leave the chunk; its step is discarded)
      const double* sp = &sig[r * KT * 16 + li];
      double c[KT];
#pragma unroll
      for (int k = 0; k < KT; ++k) {
        const double y0 = sp[k * 16], y1 = sp[k * 16 + 1];
        c[k] = fma(f, y1 - y0, y0);
      }
#pragma unroll
      for (int j = 0; j < DP; ++j) {
        double hr = h0r[j];
#pragma unroll
        for (int k = 0; k < KT; ++k) hr = fma(c[k], hkr[k][j], hr);
        Hr[j] = hr;
        if constexpr (!REALH) {
          double hi = h0i[j];
#pragma unroll
          for (int k = 0; k < KT; ++k) hi = fma(c[k], hki[k][j], hi);
          Hi[j] = hi;
        }
      }
    };
    const int j1 = (j0 + SPC < nrel) ? j0 + SPC : nrel;
    for (int jr = j0; jr < j1; ++jr) {
      const int n = n_begin + jr;
      const bool act = n < n_end;
      double kr[S], ki[S];
      static_for<0, S>([&](auto sc) {
        constexpr int s = decltype(sc)::value;
        // (the stage position after this one that assembles H again: a later stage of this step, or -- stage 0 is carried --
        // the first such stage of the next step; supplied Hamiltonians are requested one assemble ahead)
        constexpr int s2 = next_asm_stage(SOLVER, s), s3 = s2 >= 0 ? s2 : next_asm_stage(SOLVER, 0);
        constexpr double th2 = tab_of(SOLVER).node[s3 >= 0 ? s3 : 0];
        if constexpr (s == 0) {
          if (jr == 0) assemble(tab_of(SOLVER).node[0], n, th2, s2 >= 0 ? n : n + 1);  // later steps: carried over from the previous step's last node
        } else if constexpr (tab_of(SOLVER).node[s] != tab_of(SOLVER).node[s - 1]) {
          assemble(tab_of(SOLVER).node[s], n, th2, s2 >= 0 ? n : n + 1);
        }
        double yr = pr, yi = pi;
        static_for<0, s>([&](auto jc) {
          constexpr int j = decltype(jc)::value;
          constexpr double a = tab_of(SOLVER).a[s][j];
          if constexpr (a != 0.0) {
            yr = fma(a, kr[j], yr);
            yi = fma(a, ki[j], yi);
          }
        });
        double w[4] = {0.0, 0.0, 0.0, 0.0};
        if constexpr (REALH) {
          P::matvec_r(w, yr, yi, Hr);
          kr[s] = dt * (w[1] + w[3]);
          ki[s] = -dt * (w[0] + w[2]);
        } else {
          P::matvec_c(w, yr, yi, Hr, Hi);
          kr[s] = dt * (w[1] + w[3]);  // -i dt (wr + i wi) = dt wi - i dt wr
          ki[s] = -dt * (w[0] + w[2]);
        }
      });
      double qr = pr, qi = pi;
      static_for<0, S>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        constexpr double bj = tab_of(SOLVER).b[j];
        if constexpr (bj != 0.0) {
          qr = fma(bj, kr[j], qr);
          qi = fma(bj, ki[j], qi);
        }
      });
      pr = act ? qr : pr;
      pi = act ? qi : pi;
      if (A.want_all && live && row && (act || !A.seg_traj)) outp[(long)n * ssz] = cmake(pr, pi);
      if (A.reset_each_step) {
        pr = ir;
        pi = ii;
      }
    }
  }
  if (!A.want_all && live && row) outp[0] = cmake(pr, pi);
}

// ---------------------------------------------------------------------------------------------------------------
// matrix state rho: von Neumann (C = 0) and Lindblad steps; lane i owns row i of rho
// ---------------------------------------------------------------------------------------------------------------
struct MatLds {
  int sig_off, k_off, col_off, x_off, xs_off;
  size_t bytes;
};
constexpr int MAT_KX_MAX = 12;  // control lines beyond the four whose operator rows sit in registers (their rows: LDS)
// stage slots hold the LIVE lanes only (4 rows x D lanes): 7 stages at D = 9 are 36 KB instead of 64 KB per wavefront
// (rk4: a[s][j] = 0 but for j = s - 1, the previous stage is still in registers: no slots at all)
__host__ __device__ inline MatLds mat_lds(int D, int DP, int KT, int solver, int C, int KX = 0) {
  const int stages = solver == 0 ? 0 : (solver == 1 ? 4 : 7);
  MatLds m;
  m.sig_off = 0;
  m.k_off = 4 * KT * 16 * 8;
  m.col_off = m.k_off + stages * DP * 4 * D * 16;
  m.x_off = m.col_off + C * DP * DP * 16;   // [KX][DP][DP] complex: rows of the extra control operators
  m.xs_off = m.x_off + KX * DP * DP * 16;   // [4 samples][KX][16] control amplitudes of the extra lines (current chunk)
  m.bytes = (size_t)m.xs_off + (size_t)4 * KX * 16 * 8;
  return m;
}

template <int DP, int KT, bool REALH>
__global__ void __launch_bounds__(64, 1) ode_mat_kernel(OdeArgs A, OdeRowAux X) {
  using P = OdeDpp<DP>;
  const OdeTableau& T = c3p_row_tab[A.solver];
  const int S = T.stages;
  const int lane = threadIdx.x, r = lane >> 4, i = lane & 15;
  const int D = A.D, K = A.K, N = A.N, C = A.C, us = A.u_stride;
  const int KX = K > KT ? K - KT : 0;  // (KT = 4 only: control lines beyond the four register-resident ones)
  const MatLds L = mat_lds(D, DP, KT, A.solver, C, KX);
  cplx* hx = reinterpret_cast<cplx*>(c3p_ode_row_smem + L.x_off);
  double* sigx = reinterpret_cast<double*>(c3p_ode_row_smem + L.xs_off);
  const bool subdiag = A.solver == 0;  // rk4
  double* sig = reinterpret_cast<double*>(c3p_ode_row_smem + L.sig_off);
  cplx* kst = reinterpret_cast<cplx*>(c3p_ode_row_smem + L.k_off);       // [stage][c][live lane]
  cplx* colrow = reinterpret_cast<cplx*>(c3p_ode_row_smem + L.col_off);  // [m][i][j], zero padded to DP
  long v = (long)blockIdx.x * 4 + r;
  const bool live = v < A.B;
  if (!live) v = A.B - 1;
  const int b = (int)v;
  const bool row = i < D;
  const int NL = 4 * D;                                // live lanes of the wavefront
  const int lidx = r * D + (row ? i : D - 1);          // stage-slot index (padding lanes alias a live one, reads masked)

  double h0r[DP], h0i[REALH ? 1 : DP], hkr[KT][DP], hki[KT][REALH ? 1 : DP];
  bool im0 = true;
#pragma unroll
  for (int j = 0; j < DP; ++j) {
    cplx z = cmake(0, 0);
    if (row && j < D) z = A.h0[i * D + j];
    h0r[j] = z.x;
    if constexpr (!REALH) h0i[j] = z.y;
    im0 = im0 && (z.y == 0.0);
#pragma unroll
    for (int k = 0; k < KT; ++k) {
      cplx zk = cmake(0, 0);
      if (row && j < D && k < K) zk = A.hks[((long)k * D + i) * D + j];
      hkr[k][j] = zk.x;
      if constexpr (!REALH) hki[k][j] = zk.y;
      im0 = im0 && (zk.y == 0.0);
    }
  }
  // rows of the control operators beyond the register-resident ones -> LDS (shared by the four samples of the wavefront)
  for (int e = lane; e < KX * DP * DP; e += 64) {
    const int k = e / (DP * DP), ii = (e / DP) % DP, jj = e % DP;
    cplx z = cmake(0, 0);
    if (ii < D && jj < D) z = A.hks[((long)(KT + k) * D + ii) * D + jj];
    hx[e] = z;
    im0 = im0 && (z.y == 0.0);
  }
  // the real instance takes real Hamiltonians without collapse operators, the complex one everything else
  const bool real_case = (__all(im0) != 0) && C == 0;
  if (real_case != REALH) return;
  if constexpr (!REALH) {
    if (A.complex_to_wg && C == 0) return;  // complex operators of this call: the workgroup kernel launched beside takes them
  }

  // G = sum_m C_m^+ C_m (row i, from the padded copy the prep kernel wrote), own rows of the collapse operators -> LDS
  double Gr[REALH ? 1 : DP], Gi[REALH ? 1 : DP];
  if constexpr (!REALH) {
#pragma unroll
    for (int j = 0; j < DP; ++j) {
      cplx g = cmake(0, 0);
      if (C > 0) g = X.gpad[i * DP + j];
      Gr[j] = g.x;
      Gi[j] = g.y;
    }
    for (int e = lane; e < C * DP * DP; e += 64) colrow[e] = X.colpad[e];
  }

  double Sr[DP], Si[DP];
  const cplx* init = A.init + (long)b * A.init_bstride;
#pragma unroll
  for (int c = 0; c < DP; ++c) {
    cplx z = cmake(0, 0);
    if (row && c < D) z = init[(long)i * D + c];
    Sr[c] = z.x;
    Si[c] = z.y;
  }
  const double dt = A.dt;
  const double* sg = A.signals + (long)b * K * N;
  const int SPC = chunk_steps(us);
  double pre[KT];
  {
    const int base = chunk_base(0, us, N);
    int idx = base + i;
    if (idx > N - 1) idx = N - 1;
#pragma unroll
    for (int k = 0; k < KT; ++k) pre[k] = (k < K) ? sg[(long)k * N + idx] : 0.0;
  }
  const long ssz = (long)D * D;
  cplx* outp = A.states + (long)b * (A.want_all ? (long)A.n_steps : 1) * ssz + (long)i * D;

  for (int n0 = 0; n0 < A.n_steps; n0 += SPC) {
    const int base = chunk_base(n0, us, N);
#pragma unroll
    for (int k = 0; k < KT; ++k) sig[(r * KT + k) * 16 + i] = pre[k];
    if (KX > 0) {  // (the extra lines are fetched for the chunk itself, not one ahead)
      int idx = base + i;
      if (idx > N - 1) idx = N - 1;
      for (int k = 0; k < KX; ++k) sigx[(r * KX + k) * 16 + i] = sg[(long)(KT + k) * N + idx];
    }
    {
      const int nb = chunk_base(n0 + SPC, us, N);
      int idx = nb + i;
      if (idx > N - 1) idx = N - 1;
#pragma unroll
      for (int k = 0; k < KT; ++k) pre[k] = (k < K) ? sg[(long)k * N + idx] : 0.0;
    }
    const int n1 = (n0 + SPC < A.n_steps) ? n0 + SPC : A.n_steps;
    for (int n = n0; n < n1; ++n) {
      double kr[DP], ki[DP], Br[DP], Bi[DP];  // current stage, running sum_j b_j k_j
#pragma unroll
      for (int c = 0; c < DP; ++c) kr[c] = ki[c] = Br[c] = Bi[c] = 0.0;
      for (int s = 0; s < S; ++s) {
        // H(t_stage): rows L = H - (i/2) G (left factor) and R = H + (i/2) G (right factor, read by the other lanes)
        double Lr[DP], Li[REALH ? 1 : DP], Rr[REALH ? 1 : DP], Ri[REALH ? 1 : DP];
        {
          const double u = ((double)n + T.node[s]) * (double)us;
          int lo = (int)floor(u);
          if (lo > N - 2) lo = N - 2;
          if (lo < 0) lo = 0;
          const double f = u - (double)lo;
          const double* sp = &sig[r * KT * 16 + (lo - base)];
          double c[KT];
#pragma unroll
          for (int k = 0; k < KT; ++k) {
            const double y0 = sp[k * 16], y1 = sp[k * 16 + 1];
            c[k] = fma(f, y1 - y0, y0);
          }
          double hxr[DP], hxi[REALH ? 1 : DP];  // contribution of the control lines beyond KT (operator rows from LDS)
#pragma unroll
          for (int j = 0; j < DP; ++j) {
            hxr[j] = 0.0;
            if constexpr (!REALH) hxi[j] = 0.0;
          }
          for (int k = 0; k < KX; ++k) {
            const double* spx = &sigx[(r * KX + k) * 16 + (lo - base)];
            const double y0 = spx[0], y1 = spx[1];
            const double cx = fma(f, y1 - y0, y0);
            const cplx* hrow = hx + ((long)k * DP + (row ? i : DP - 1)) * DP;
#pragma unroll
            for (int j = 0; j < DP; ++j) {
              const cplx z = hrow[j];
              hxr[j] = fma(cx, z.x, hxr[j]);
              if constexpr (!REALH) hxi[j] = fma(cx, z.y, hxi[j]);
            }
          }
#pragma unroll
          for (int j = 0; j < DP; ++j) {
            double hr = h0r[j] + (row ? hxr[j] : 0.0);
#pragma unroll
            for (int k = 0; k < KT; ++k) hr = fma(c[k], hkr[k][j], hr);
            if constexpr (REALH) {
              Lr[j] = hr;
            } else {
              double hi = h0i[j] + (row ? hxi[j] : 0.0);
#pragma unroll
              for (int k = 0; k < KT; ++k) hi = fma(c[k], hki[k][j], hi);
              // -(i/2) (gr + i gi) = gi/2 - i gr/2
              Lr[j] = fma(0.5, Gi[j], hr);
              Li[j] = fma(-0.5, Gr[j], hi);
              Rr[j] = fma(-0.5, Gi[j], hr);
              Ri[j] = fma(0.5, Gr[j], hi);
            }
          }
        }
        // stage argument Y = rho + sum_j a[s][j] k_j
        double Yr[DP], Yi[DP];
#pragma unroll
        for (int c = 0; c < DP; ++c) {
          Yr[c] = Sr[c];
          Yi[c] = Si[c];
        }
        if (s > 0) {
          const double a = T.a[s][s - 1];  // the previous stage is still in registers
#pragma unroll
          for (int c = 0; c < DP; ++c) {
            Yr[c] = fma(a, kr[c], Yr[c]);
            Yi[c] = fma(a, ki[c], Yi[c]);
          }
        }
        if (!subdiag) {
          for (int j = 0; j + 1 < s; ++j) {
            const double a0 = T.a[s][j];
            if (a0 != 0.0) {
              const double a = row ? a0 : 0.0;  // padding lanes keep their zero rows
#pragma unroll
              for (int c = 0; c < DP; ++c) {
                const cplx kk = kst[(j * DP + c) * NL + lidx];
                Yr[c] = fma(a, kk.x, Yr[c]);
                Yi[c] = fma(a, kk.y, Yi[c]);
              }
            }
          }
        }
        // W = L Y - Y R
        double Wr[DP], Wi[DP];
#pragma unroll
        for (int c = 0; c < DP; ++c) Wr[c] = Wi[c] = 0.0;
        static_for<0, DP>([&](auto jc) {
          constexpr int J = decltype(jc)::value;
          if constexpr (REALH) {
            P::template bmac_rs<J, false>(Wr, Wi, Lr[J], Yr, Yi);
          } else {
            P::template bmac_cc<J, false>(Wr, Wi, Lr[J], Li[J], Yr, Yi);
          }
        });
        static_for<0, DP>([&](auto jc) {
          constexpr int J = decltype(jc)::value;
          if constexpr (REALH) {
            P::template bmac_rv<J, true>(Wr, Wi, Yr[J], Yi[J], Lr);
          } else {
            P::template bmac_cc<J, true>(Wr, Wi, Yr[J], Yi[J], Rr, Ri);
          }
        });
        // k = -i dt W
#pragma unroll
        for (int c = 0; c < DP; ++c) {
          kr[c] = dt * Wi[c];
          ki[c] = -dt * Wr[c];
        }
        if constexpr (!REALH) {
          // jump terms: k += dt sum_m (C_m Y) C_m^+
          for (int m = 0; m < C; ++m) {
            double Tr[DP], Ti[DP];
#pragma unroll
            for (int c = 0; c < DP; ++c) Tr[c] = Ti[c] = 0.0;
            const cplx* cr = colrow + ((long)m * DP + i) * DP;
            static_for<0, DP>([&](auto jc) {
              constexpr int J = decltype(jc)::value;
              const cplx cij = cr[J];
              P::template bmac_cc<J, false>(Tr, Ti, cij.x, cij.y, Yr, Yi);
            });
#pragma unroll
            for (int c = 0; c < DP; ++c) {
              Tr[c] *= dt;
              Ti[c] *= dt;
            }
            // (T C^+)_ic = sum_j T_ij conj(C_cj): the right factor is the same for every lane -> scalar operands
            // (read through the constant address space: s_load + SGPR operands; the buffer is written by the prep kernel only)
            const cdouble_ptr ca = (cdouble_ptr)(const double*)(X.coladj + (long)m * DP * DP);  // [j][c] = conj(C_m[c][j])
#pragma unroll
            for (int j = 0; j < DP; ++j) {
#pragma unroll
              for (int c = 0; c < DP; ++c) {
                const double zx = ca[2 * (j * DP + c)], zy = ca[2 * (j * DP + c) + 1];
                kr[c] = fma(Tr[j], zx, kr[c]);
                kr[c] = fma(-Ti[j], zy, kr[c]);
                ki[c] = fma(Tr[j], zy, ki[c]);
                ki[c] = fma(Ti[j], zx, ki[c]);
              }
            }
          }
        }
        {
          const double bs = T.b[s];
#pragma unroll
          for (int c = 0; c < DP; ++c) {
            Br[c] = fma(bs, kr[c], Br[c]);
            Bi[c] = fma(bs, ki[c], Bi[c]);
          }
        }
        if (!subdiag && row && s + 2 < S) {  // later stages than the next one read it back
#pragma unroll
          for (int c = 0; c < DP; ++c) kst[(s * DP + c) * NL + lidx] = cmake(kr[c], ki[c]);
        }
      }
      // rho += sum_j b_j k_j
#pragma unroll
      for (int c = 0; c < DP; ++c) {
        Sr[c] += Br[c];
        Si[c] += Bi[c];
      }
      if (A.want_all && live && row) {
        cplx* o = outp + (long)n * ssz;
#pragma unroll
        for (int c = 0; c < DP; ++c)
          if (c < D) o[c] = cmake(Sr[c], Si[c]);
      }
    }
  }
  if (!A.want_all && live && row) {
#pragma unroll
    for (int c = 0; c < DP; ++c)
      if (c < D) outp[c] = cmake(Sr[c], Si[c]);
  }
}

// padded copies of the collapse operators for ode_mat_kernel: colpad [C][DP][DP], coladj[m][j][c] = conj(C_m[c][j]),
// gpad = sum_m C_m^+ C_m [DP][DP]
__global__ void ode_colprep_kernel(const cplx* col, int C, int D, int DP, cplx* colpad, cplx* coladj, cplx* gpad) {
  const int tid = threadIdx.x, nt = blockDim.x;
  for (int e = tid; e < C * DP * DP; e += nt) {
    const int m = e / (DP * DP), rr = (e / DP) % DP, cc = e % DP;
    cplx z = cmake(0, 0), za = cmake(0, 0);
    if (rr < D && cc < D) {
      z = col[((long)m * D + rr) * D + cc];
      za = cconj(col[((long)m * D + cc) * D + rr]);
    }
    colpad[e] = z;
    coladj[e] = za;
  }
  for (int e = tid; e < DP * DP; e += nt) {
    const int ii = e / DP, jj = e % DP;
    cplx s = cmake(0, 0);
    if (ii < D && jj < D)
      for (int m = 0; m < C; ++m)
        for (int k = 0; k < D; ++k) cfma(s, cconj(col[((long)m * D + k) * D + ii]), col[((long)m * D + k) * D + jj]);
    gpad[e] = s;
  }
}

// psi_out[b] = U[b] psi0[b]: the step map of the whole interval applied to the initial state (segmented integration)
__global__ void __launch_bounds__(64) ode_apply_kernel(const cplx* U, const cplx* init, long init_bstride, cplx* out, int D) {
  const int b = blockIdx.x, i = threadIdx.x;
  if (i >= D) return;
  const cplx* Ub = U + (long)b * D * D + (long)i * D;
  const cplx* p = init + (long)b * init_bstride;
  cplx s = cmake(0.0, 0.0);
  for (int j = 0; j < D; ++j) cfma(s, Ub[j], p[j]);
  out[(long)b * D + i] = s;
}

__global__ void ode_identity_kernel(cplx* out, int D) {
  for (int e = threadIdx.x; e < D * D; e += blockDim.x) out[e] = cmake((e / D == e % D) ? 1.0 : 0.0, 0.0);
}

// Hs[b][n] = h0 + sum_k c_k[b][n] hk for every sample index (more than four control lines on the lane-row kernels: the rows
// of four operators fit the registers, the assembled Hamiltonians are read per stage node instead, hs_lerp)
__global__ void __launch_bounds__(256) ode_assemble_hs_kernel(const cplx* h0, const cplx* hks, const double* signals, int K, int N, int D,
                                                              cplx* out) {
  const int n = (int)(blockIdx.x % (unsigned)N), b = (int)(blockIdx.x / (unsigned)N);
  const double* sg = signals + (long)b * K * N + n;
  cplx* o = out + ((long)b * N + n) * D * D;
  for (int e = threadIdx.x; e < D * D; e += blockDim.x) {
    cplx h = h0[e];
    for (int k = 0; k < K; ++k) {
      const double c = sg[(long)k * N];
      const cplx x = hks[(long)k * D * D + e];
      h.x = fma(c, x.x, h.x);
      h.y = fma(c, x.y, h.y);
    }
    o[e] = h;
  }
}

int pad_dim(int D) {
  const int dps[] = {2, 3, 4, 6, 9, 12, 16};
  for (int d : dps)
    if (D <= d) return d;
  return 0;
}

template <int DP, int KT, int SOLVER>
hipError_t launch_vec3(const OdeArgs& A, dim3 grid, hipStream_t st) {
  C3P_LAUNCH((ode_vec_kernel<DP, KT, SOLVER, true>), grid, dim3(64), 0, st, A);
  C3P_LAUNCH((ode_vec_kernel<DP, KT, SOLVER, false>), grid, dim3(64), 0, st, A);
  return hipGetLastError();
}
template <int DP, int KT>
hipError_t launch_vec2(const OdeArgs& A, dim3 grid, hipStream_t st) {
  switch (A.solver) {
    case 0: return launch_vec3<DP, KT, 0>(A, grid, st);
    case 1: return launch_vec3<DP, KT, 1>(A, grid, st);
    case 2: return launch_vec3<DP, KT, 2>(A, grid, st);
    default: return launch_vec3<DP, KT, 3>(A, grid, st);
  }
}
template <int DP>
hipError_t launch_vec1(const OdeArgs& A, dim3 grid, hipStream_t st) {
  if (A.hs) {  // supplied Hamiltonians: one (complex) instance per solver
    switch (A.solver) {
      case 0: C3P_LAUNCH((ode_vec_kernel<DP, 2, 0, false, true>), grid, dim3(64), 0, st, A); break;
      case 1: C3P_LAUNCH((ode_vec_kernel<DP, 2, 1, false, true>), grid, dim3(64), 0, st, A); break;
      case 2: C3P_LAUNCH((ode_vec_kernel<DP, 2, 2, false, true>), grid, dim3(64), 0, st, A); break;
      default: C3P_LAUNCH((ode_vec_kernel<DP, 2, 3, false, true>), grid, dim3(64), 0, st, A); break;
    }
    return hipGetLastError();
  }
  return A.K <= 2 ? launch_vec2<DP, 2>(A, grid, st) : launch_vec2<DP, 4>(A, grid, st);
}

template <int DP, int KT, bool REALH>
hipError_t launch_mat3(const OdeArgs& A, const OdeRowAux& X, dim3 grid, size_t lds, hipStream_t st) {
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&ode_mat_kernel<DP, KT, REALH>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024);
  if (e != hipSuccess) return e;
  C3P_LAUNCH((ode_mat_kernel<DP, KT, REALH>), grid, dim3(64), lds, st, A, X);
  return hipGetLastError();
}
template <int DP>
hipError_t launch_mat1(const OdeArgs& A, const OdeRowAux& X, dim3 grid, hipStream_t st) {
  hipError_t e = hipSuccess;
  if (A.K <= 2) {
    const size_t lds = mat_lds(A.D, DP, 2, A.solver, A.C).bytes;
    if (A.C == 0) e = launch_mat3<DP, 2, true>(A, X, grid, lds, st);
    if (e == hipSuccess) e = launch_mat3<DP, 2, false>(A, X, grid, lds, st);
  } else {
    const size_t lds = mat_lds(A.D, DP, 4, A.solver, A.C, A.K > 4 ? A.K - 4 : 0).bytes;
    if (A.C == 0) e = launch_mat3<DP, 4, true>(A, X, grid, lds, st);
    if (e == hipSuccess) e = launch_mat3<DP, 4, false>(A, X, grid, lds, st);
  }
  return e;
}

}  // namespace

bool c3p_ode_row_supported(const OdeArgs& A) {
  if (c3p_opt_on(C3P_OPT_ode_wg)) return false;  // A/B switch: the workgroup-per-sample kernel of c3p_ode.hip
  if (A.D > 16 || A.N < 2) return false;
  if (A.u_stride != 1 && A.u_stride != 2) return false;
  const bool vec = A.step == C3P_STEP_SCHRODINGER_ID || A.step == C3P_STEP_PROPAGATOR_ID;
  if (vec) return A.K <= 4 || A.hs;  // (supplied Hamiltonians: no control lines in the kernel)
  // rho-valued steps: control lines beyond four keep their operator rows in LDS.  (Supplied Hamiltonians were measured there:
  // the row reads from memory per stage make the lane-row kernel as slow as the workgroup kernel, 8.6 against 8.6 ms at D = 9,
  // B = 256 -- not instantiated.)
  if (A.hs || A.reset_each_step || A.transpose_out || A.K > 4 + MAT_KX_MAX) return false;
  const int DP = pad_dim(A.D);
  return mat_lds(A.D, DP, A.K <= 2 ? 2 : 4, A.solver, A.C, A.K > 4 ? A.K - 4 : 0).bytes <= (size_t)(150 * 1024);
}

hipError_t c3p_launch_ode_assemble_hs(const OdeArgs& A, cplx* out, hipStream_t st) {
  C3P_LAUNCH(ode_assemble_hs_kernel, dim3((unsigned)((long)A.N * A.B)), dim3(A.D * A.D >= 256 ? 256 : 64), 0, st, A.h0, A.hks,
                     A.signals, A.K, A.N, A.D, out);
  return hipGetLastError();
}

size_t c3p_ode_row_aux_bytes(int D, int C) {
  const int DP = pad_dim(D);
  return (size_t)(2 * C + 1) * DP * DP * sizeof(cplx);
}

hipError_t c3p_launch_ode_row(const OdeArgs& A, void* aux, hipStream_t st) {
  const int DP = pad_dim(A.D);
  const bool vec = (A.step == C3P_STEP_SCHRODINGER_ID || A.step == C3P_STEP_PROPAGATOR_ID);
  const long nv = vec ? (long)A.B * A.M * (A.seg_count > 0 ? A.seg_count : 1) : (long)A.B;
  const dim3 grid((unsigned)((nv + 3) / 4));
  if (vec) {
    switch (DP) {
      case 2: return launch_vec1<2>(A, grid, st);
      case 3: return launch_vec1<3>(A, grid, st);
      case 4: return launch_vec1<4>(A, grid, st);
      case 6: return launch_vec1<6>(A, grid, st);
      case 9: return launch_vec1<9>(A, grid, st);
      case 12: return launch_vec1<12>(A, grid, st);
      default: return launch_vec1<16>(A, grid, st);
    }
  }
  OdeRowAux X = {};
  if (A.C > 0) {
    cplx* base = static_cast<cplx*>(aux);
    const size_t one = (size_t)DP * DP;
    cplx* colpad = base;
    cplx* coladj = base + (size_t)A.C * one;
    cplx* gpad = base + (size_t)2 * A.C * one;
    C3P_LAUNCH(ode_colprep_kernel, dim3(1), dim3(256), 0, st, A.col_ops, A.C, A.D, DP, colpad, coladj, gpad);
    X.colpad = colpad;
    X.coladj = coladj;
    X.gpad = gpad;
  }
  switch (DP) {
    case 2: return launch_mat1<2>(A, X, grid, st);
    case 3: return launch_mat1<3>(A, X, grid, st);
    case 4: return launch_mat1<4>(A, X, grid, st);
    case 6: return launch_mat1<6>(A, X, grid, st);
    case 9: return launch_mat1<9>(A, X, grid, st);
    case 12: return launch_mat1<12>(A, X, grid, st);
    default: return launch_mat1<16>(A, X, grid, st);
  }
}

// Time segments for small batches (final state / propagator only): how many segments to cut the time axis into.
// 0 = integrate in one piece.  The segment maps cost D columns per sample instead of one state, and a wavefront that shares
// its SIMD runs slower (one wave alone: 0.50 us per rk4 step at D = 9, i.e. ~70 % of the SIMD's fp64 issue rate already;
// w waves per SIMD: ~0.14 + 0.34 w us), so the choice minimises  n_steps / S * c(waves per SIMD)  + the fixed cost of the
// extra launches (identity, combine, apply: ~50 us) over S = 1..8 (the small-D chain kernel folds up to 8 maps in one launch).
// Measured, D = 9, 1000 rk4 steps (direct 0.50 ms): B = 16 0.117 ms (S = 8), B = 64 0.128 ms (S = 7), B = 256 0.35 ms (S = 3).
int c3p_ode_row_segments(const OdeArgs& A) {
  if (c3p_opt_on(C3P_OPT_ode_no_seg)) return 0;
  if (A.D < 2 || A.D > 12 || A.want_all || A.reset_each_step || A.n_steps < 64) return 0;  // (the combine runs on the small-D chain kernel)
  auto step_cost = [](long rows) {  // relative time of one RK step when `rows` DPP rows (4 per wave) are spread over 1024 SIMDs
    const long w = (rows + 4095) / 4096;
    return w <= 1 ? 0.50 : 0.14 + 0.34 * (double)w;
  };
  const double fixed = 50.0 * (A.solver == 0 || A.solver == 1 ? 1.0 : 0.55);  // in units of rk4 steps' microseconds (7-stage solvers: longer steps)
  if ((long)A.B * A.M > 4096) return 0;  // more than one wave per SIMD already: throughput, not latency (not measured beyond)
  const double direct = (double)A.n_steps * step_cost((long)A.B * A.M);
  double best = direct * 0.8;  // a segmented launch has to win by a margin
  int pick = 0;
  for (int S = 2; S <= 8 && S <= A.n_steps / 32; ++S) {
    const double c = (double)((A.n_steps + S - 1) / S) * step_cost((long)A.B * S * A.D) + fixed;
    if (c < best) best = c, pick = S;
  }
  return pick;
}

// Trajectories (want_all) of small batches: the segment maps of ode_segmented give the state at the START of every segment
// (one small kernel), and a second pass integrates the B x S pieces side by side, each writing its own range of the
// trajectory.  Cost model as c3p_ode_row_segments, plus the second pass.
int c3p_ode_row_traj_segments(const OdeArgs& A) {
  if (c3p_opt_on(C3P_OPT_ode_no_seg)) return 0;
  if (A.D < 2 || A.D > 12 || !A.want_all || A.M != 1 || A.reset_each_step || A.transpose_out || A.n_steps < 64) return 0;
  auto step_cost = [](long rows) {
    const long w = (rows + 4095) / 4096;
    return w <= 1 ? 0.50 : 0.14 + 0.34 * (double)w;
  };
  const double fixed = 60.0 * (A.solver == 0 || A.solver == 1 ? 1.0 : 0.55);
  if ((long)A.B > 4096) return 0;
  const double direct = (double)A.n_steps * step_cost((long)A.B);
  double best = direct * 0.8;
  int pick = 0;
  for (int S = 2; S <= 8 && S <= A.n_steps / 32; ++S) {
    const double c = (double)((A.n_steps + S - 1) / S) * (step_cost((long)A.B * S * A.D) + step_cost((long)A.B * S)) + fixed;
    if (c < best) best = c, pick = S;
  }
  return pick;
}

namespace {
// starts[b, s] = maps[b, s-1] ... maps[b, 0] init[b]  (maps [B, S, D, D] row-major); one wavefront per sample, lane i = row i
__global__ void __launch_bounds__(64) ode_starts_kernel(const cplx* maps, const cplx* init, long init_bstride, cplx* starts, int S, int D) {
  __shared__ cplx psi[64];
  const int i = threadIdx.x;
  const long b = blockIdx.x;
  cplx v = i < D ? init[b * init_bstride + i] : cmake(0, 0);
  for (int s = 0; s < S; ++s) {
    if (i < D) starts[(b * S + s) * D + i] = v;
    psi[i] = v;
    __syncthreads();
    if (i < D) {
      const cplx* m = maps + ((b * S + s) * D + i) * D;
      cplx a = cmake(0, 0);
      for (int j = 0; j < D; ++j) cfma(a, m[j], psi[j]);
      v = a;
    }
    __syncthreads();
  }
}
}  // namespace

hipError_t c3p_launch_ode_starts(const cplx* maps, const cplx* init, long init_bstride, cplx* starts, int B, int S, int D, hipStream_t st) {
  C3P_LAUNCH(ode_starts_kernel, dim3((unsigned)B), dim3(64), 0, st, maps, init, init_bstride, starts, S, D);
  return hipGetLastError();
}

hipError_t c3p_launch_ode_identity(cplx* out, int D, hipStream_t st) {
  C3P_LAUNCH(ode_identity_kernel, dim3(1), dim3(256), 0, st, out, D);
  return hipGetLastError();
}

hipError_t c3p_launch_ode_apply(const cplx* U, const cplx* init, long init_bstride, cplx* out, int B, int D, hipStream_t st) {
  C3P_LAUNCH(ode_apply_kernel, dim3((unsigned)B), dim3(64), 0, st, U, init, init_bstride, out, D);
  return hipGetLastError();
}
