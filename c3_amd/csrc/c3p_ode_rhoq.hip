// Matrix-core ODE solver for rho-valued states at 17 <= D <= 48 (round 3): von Neumann and Lindblad steps of
// ode_solver / ode_solver_final_state (c3/libraries/propagation.py:687-752, tableaux :755-883, step functions :886-904)
// with the Hamiltonian of Model.Hs_of_t (c3/model.py:641-697) on linearly interpolated controls
// (c3/utils/tf_utils.py:521-559) -- the dimensions of BASELINE cfg3 (D = 27) and cfg5 (D = 36), where the state is a
// D x D matrix and a stage is two (von Neumann) or 2 + 2 C (Lindblad) D x D x D complex products: GEMM-shaped, so it runs
// on `v_mfma_f64_16x16x4_f64`.  (D <= 16 stays on the lane-row kernel of c3p_ode_row.hip, anything else on c3p_ode.hip.)
//
//  * One workgroup per sample, NT x NT wavefronts (NT = ceil(D / 16): 4 waves for D <= 32, 9 for D <= 48); wave (I, J)
//    owns the 16 x 16 tile (I, J) of EVERY matrix in the C/D layout of the instruction (register v of lane l = element
//    (4 v + l / 16, l % 16)): rho, the RK stages k_1 .. k_S, the operator tiles of h0 and the hk, the assembled H(t).
//    Everything elementwise (stage arguments, H(t) = h0 + sum_k c_k(t) hk, the RK update) is register arithmetic on four
//    elements per lane; the tableau is a template parameter, zero coefficients vanish at compile time.
//  * Only product OPERANDS pass through LDS: per stage every wave writes its tile of H(t) and of the stage argument Y to
//    zero-padded planes (separate real / imaginary planes, row stride 16 NT + 2 doubles: the A-fragment reads -- 16 rows x
//    2 columns per half wave -- are conflict free, the B-fragment reads two-way), one barrier, then
//        acc = H Y - Y H      in ONE accumulator pair: per K-step of 4 the A fragments H(I, k), Y(I, k) and the B
//    fragments Y(k, J), H(k, J), eight real MFMAs (four when the operators are real: that instance is picked on the device,
//    both are launched and the one that does not apply exits), alternating between the real and the imaginary accumulator.
//    The K loop stops at ceil(D / 4) steps (the padding columns are zero).  k_s = -i dt acc.
//  * Lindblad: with G = sum_m C_m^+ C_m the anticommutator is folded into L = H - (i/2) G, R = H + (i/2) G:
//    -i (L Y - Y R) = -i [H, Y] - {G, Y} / 2 (the same commutator loop on complex planes), and every jump term is two more
//    products, T = C_m Y (written back to an LDS plane by the owning waves, one barrier) and T C_m^+ (the B fragment of C^+
//    is the conjugated A-pattern read of C).  The C_m sit in LDS planes for the whole integration at D <= 32; at 33 <= D <= 48
//    they stay in memory and their fragments are fetched one product ahead (CGLOB, round 5).
//  * Hermitian shortcut (the usual case: Hermitian operators, rho(0) a density matrix): every stage argument is Hermitian,
//    so Y H = (H Y)^+ -- ONE product per commutator (half the MFMAs), written to a plane, and the other half is the
//    conjugate of the mirrored element read back behind one more barrier.  Symmetry is checked per sample in the prologue;
//    the general two-product instance stays for everything else (`C3P_ODE_RHO_GENERAL=1` forces it).
//  * Control amplitudes: a step only looks at u_stride + 2 samples per line; that window sits in one register (lane 4 k + i =
//    sample base + i of line k), is fetched ONE STEP ahead by a single vector load and read with v_readlane; the barriers
//    of the step loop wait for LDS only, so the load stays in flight.  `Hs` is never materialised.
//  * K-loop fragments are fetched three K-steps (two in the general form) ahead of their MFMAs.  Built with
//    `-mllvm -amdgpu-mfma-vgpr-form` (see __graft_entry__.py: without it the accumulators are copied between the two
//    register files around every loop iteration).  `-DC3P_RHOQ_TIMING` prints a per-phase cycle budget of a stage.
#include <type_traits>
#include <utility>

#include "c3p_common.h"
#include "c3p_ode.h"
#include "c3p_ode_tab.h"

#ifndef C3P_RHOQ_CG_LATE
#define C3P_RHOQ_CG_LATE 1  // 1: one collapse-operator fragment set live at a time (fetched behind the product that used the other: 2 - 7 % faster than both in flight, which spills)
#endif
extern __shared__ __attribute__((aligned(16))) unsigned char c3p_ode_rhoq_smem[];

namespace {

constexpr int RK = 4;  // control lines held in registers

__host__ __device__ constexpr OdeTableau rtab_of(int solver) {
  constexpr OdeTableau t[4] = C3P_ODE_TABLEAUX;
  return t[solver];
}

typedef double d4 __attribute__((ext_vector_type(4)));
typedef const double __attribute__((address_space(4)))* cdouble_ptr;  // constant address space: uniform loads become s_load

template <int I, int N, class F>
__device__ __forceinline__ void rstatic_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    rstatic_for<I + 1, N>(f);
  }
}

// MODE 0: von Neumann, real operators; 1: von Neumann, complex operators; 2: Lindblad (complex planes)
// (Lindblad at D > 32, NT = 3: the collapse operators do not get LDS planes -- 8 + 2 C planes of 19 KB do not fit -- their
// fragments are read from memory (L2) for every jump term, see CGLOB in the kernel)
__host__ __device__ constexpr int rho_planes(int NT, int mode, int C) { return mode == 0 ? 5 : (mode == 1 ? 6 : (NT >= 3 ? 8 : 8 + 2 * C)); }
__host__ __device__ constexpr size_t rho_lds_bytes(int NT, int mode, int C) {
  return (size_t)rho_planes(NT, mode, C) * (16 * NT) * (16 * NT + 2) * sizeof(double);
}

// workgroup barrier that waits for this wave's LDS traffic only (`__syncthreads` also drains the vector-memory counter, i.e.
// the signal window that is meant to stay in flight, and the trajectory stores)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ d4 mfma(double a, double b, d4 c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }

// Workgroups per CU the register budget is set for: three for the four-stage solvers on real operators (168 registers: ~20
// spilled, and still 16 - 26 % faster at B >= 768 than two workgroups with none), two for the other von Neumann instances at
// D <= 32 (256), one for Lindblad (its planes fill the LDS anyway) and for the 9-wave workgroups of D > 32 (three waves on
// one SIMD: 168 registers whatever is asked for -- the seven-stage solvers spill there).
__host__ __device__ constexpr int rho_min_wgs(int NT, int solver, int mode) {
  return NT != 2 || mode == 2 ? 1 : (mode == 0 && solver < 2 ? 3 : 2);
}

template <int NT, int SOLVER, int MODE, bool HERM>
__global__ void __launch_bounds__(64 * NT * NT, rho_min_wgs(NT, SOLVER, MODE)) ode_rhoq_kernel(OdeArgs A) {
  constexpr bool CPX = MODE != 0;
  constexpr bool LIND = MODE == 2;
  // Lindblad at 33 <= D <= 48 (round 5): the collapse operators stay in memory.  A jump term needs two fragment sets of C_m per
  // wave -- rows of tile row I (T = C_m Y) and rows of tile row J (T C_m^+) --, ceil(D / 4) complex values per lane each:
  // fetched as 16-byte loads one product AHEAD of their use (the set of the second product while the first one runs, the
  // first set of the next term while the second runs), L2 resident (C D^2 16 bytes, shared by every sample).
  constexpr bool CGLOB = LIND && NT >= 3;
  constexpr int KSMAX = 4 * NT;  // K-steps of a product at most
  constexpr int S = rtab_of(SOLVER).stages;
  constexpr int DP = 16 * NT, LD = DP + 2, PL = DP * LD;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int I = wave / NT, J = wave - I * NT;
  const int lr = lane >> 4, lc = lane & 15;
  const int D = A.D, K = A.K, N = A.N, us = A.u_stride;
  // time segments (propagator step, final result only): workgroup (b, seg) integrates steps [seg seg_len, (seg + 1) seg_len) from
  // A.init (the identity) and writes the step map of its segment to states[b, seg]
  const int SG = A.seg_count > 0 ? A.seg_count : 1;
  const int b = blockIdx.x / SG, seg = blockIdx.x - b * SG;
  const int n_begin = A.seg_count > 0 ? seg * A.seg_len : 0;
  const int n_end = A.seg_count > 0 ? (n_begin + A.seg_len < A.n_steps ? n_begin + A.seg_len : A.n_steps) : A.n_steps;
  double* lds = reinterpret_cast<double*>(c3p_ode_rhoq_smem);
  // planes: left operator (H or L), stage argument, [Lindblad: right operator R, product T, collapse operators]
  double* const Lr = lds;
  double* const Li = CPX ? lds + PL : lds;
  double* const Yr = lds + (CPX ? 2 : 1) * PL;
  double* const Yi = Yr + PL;
  double* const Rr = LIND ? lds + 4 * PL : Lr;
  double* const Ri = LIND ? lds + 5 * PL : Li;
  double* const Pr = LIND ? Rr : Yi + PL;  // H Y (Hermitian shortcut); never live together with R
  double* const Pi = Pr + PL;
  double* const Tr = lds + 6 * PL;
  double* const Ti = lds + 7 * PL;
  double* const Cp = lds + 8 * PL;  // [m][re, im] planes

  // own elements: register v <-> (row[v], col)
  const int col = 16 * J + lc;
  int eoff[4];   // offset of the element in an LDS plane
  bool valid[4];
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    const int row = 16 * I + 4 * v + lr;
    eoff[v] = row * LD + col;
    valid[v] = row < D && col < D;
  }

  // operator tiles -> registers
  double h0r[4], h0i[CPX ? 4 : 1], hkr[RK][4], hki[RK][CPX ? 4 : 1];
  bool im0 = true;
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    const int row = 16 * I + 4 * v + lr;
    cplx z = cmake(0, 0);
    if (valid[v]) z = A.h0[row * D + col];
    h0r[v] = z.x;
    if constexpr (CPX) h0i[v] = z.y;
    im0 = im0 && z.y == 0.0;
#pragma unroll
    for (int k = 0; k < RK; ++k) {
      cplx zk = cmake(0, 0);
      if (valid[v] && k < K) zk = A.hks[((long)k * D + row) * D + col];
      hkr[k][v] = zk.x;
      if constexpr (CPX) hki[k][v] = zk.y;
      im0 = im0 && zk.y == 0.0;
    }
  }
  if constexpr (!LIND) {
    const bool allreal = __syncthreads_and((int)im0) != 0;
    if (allreal != (MODE == 0)) return;  // the other instance integrates this launch
  }

  // Hermitian shortcut: with H = H^+ and rho = rho^+ every stage argument Y is Hermitian and Y H = (H Y)^+ -- ONE product
  // per commutator, the other half is the conjugate transpose of the result, fetched through LDS from the wave that owns
  // the mirrored tile.  (Lindblad: Y R = Y L^+ = (L Y)^+ likewise.)  Checked here, per sample and operator: mirrored
  // elements may differ by 1e-15 of the operator's largest element (a dressed `V^T H V` is symmetric to a few units in the
  // last place of its largest entries, not bit for bit); the result then differs from the two-product form by at most
  // 2 T |H - H^+| / 2 <= 1e-15 |H| T, the size of the rounding errors of the integration itself.
  const bool prop = A.step == C3P_STEP_PROPAGATOR_ID;  // Y' = -i H Y on a matrix of column states: one product, no mirror
  bool hm = true;
  auto herm_of = [&](const double (&xr)[4], const double (&xi)[4]) {
    double mx = 0.0;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      Yr[eoff[v]] = xr[v];
      Yi[eoff[v]] = xi[v];
      mx = fmax(mx, fmax(fabs(xr[v]), fabs(xi[v])));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmax(mx, __shfl_xor(mx, o));
    if (lane == 0) Yr[wave * LD + DP] = mx;  // the two padding columns of a plane row are never read by the products
    __syncthreads();
    double scale = 0.0;
    for (int w = 0; w < NT * NT; ++w) scale = fmax(scale, Yr[w * LD + DP]);
    const double tol = 1e-15 * scale;
    bool ok = true;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int t = col * LD + 16 * I + 4 * v + lr;
      ok = ok && fabs(Yr[t] - xr[v]) <= tol && fabs(Yi[t] + xi[v]) <= tol;
    }
    __syncthreads();
    return ok;
  };
  if (!prop) {
    double zr[4], zi[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      zr[v] = h0r[v];
      zi[v] = 0.0;
      if constexpr (CPX) zi[v] = h0i[v];
    }
    hm = herm_of(zr, zi) && hm;
#pragma unroll
    for (int k = 0; k < RK; ++k) {
      if (k < K) {
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          zr[v] = hkr[k][v];
          zi[v] = 0.0;
          if constexpr (CPX) zi[v] = hki[k][v];
        }
        hm = herm_of(zr, zi) && hm;
      }
    }
  }

  // Lindblad: collapse operators -> LDS planes, G = sum_m C_m^+ C_m -> (1/2) G in registers
  double g2r[LIND ? 4 : 1], g2i[LIND ? 4 : 1];
  if constexpr (LIND) {
    if constexpr (!CGLOB) {
      for (int m = 0; m < A.C; ++m) {
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const int row = 16 * I + 4 * v + lr;
          cplx z = cmake(0, 0);
          if (valid[v]) z = A.col_ops[((long)m * D + row) * D + col];
          Cp[(2 * m) * PL + eoff[v]] = z.x;
          Cp[(2 * m + 1) * PL + eoff[v]] = z.y;
        }
      }
      __syncthreads();
    }
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int row = 16 * I + 4 * v + lr;
      double sr = 0.0, si = 0.0;
      for (int m = 0; m < A.C; ++m) {
        if constexpr (CGLOB) {
          if (valid[v]) {
            const cplx* cm_ = A.col_ops + (long)m * D * D;
            for (int k = 0; k < D; ++k) {  // conj(C[k][row]) C[k][col]
              const cplx a = cm_[(long)k * D + row], bb = cm_[(long)k * D + col];
              sr += a.x * bb.x + a.y * bb.y;
              si += a.x * bb.y - a.y * bb.x;
            }
          }
        } else {
          const double* cr = Cp + (2 * m) * PL;
          const double* ci = cr + PL;
          for (int k = 0; k < D; ++k) {  // conj(C[k][row]) C[k][col]
            const double ar = cr[k * LD + row], ai = ci[k * LD + row], br = cr[k * LD + col], bi = ci[k * LD + col];
            sr += ar * br + ai * bi;
            si += ar * bi - ai * br;
          }
        }
      }
      g2r[v] = 0.5 * sr;
      g2i[v] = 0.5 * si;
    }
  }

  // state tile
  double pr[4], pi[4];
  const cplx* init = A.init + (long)b * A.init_bstride;
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    const int row = 16 * I + 4 * v + lr;
    cplx z = cmake(0, 0);
    if (valid[v]) z = init[(long)row * D + col];
    pr[v] = z.x;
    pi[v] = z.y;
  }
  if (prop) {
    if (!HERM) return;  // (only the one-product instance is launched for the propagator step)
  } else {
    hm = herm_of(pr, pi) && hm;
    // (the instance built for the other case integrates this sample: both are launched)
    if (((__syncthreads_and((int)hm) != 0) && !A.rho_general) != HERM) return;
  }
  constexpr bool herm = HERM;
  const bool tail_barrier = prop || !(herm && (!LIND || A.C == 0));  // see the end of a stage
  const double dt = A.dt;
  // Control amplitudes at u = (n + node) * u_stride samples: linear interpolation, linear extrapolation past the last
  // sample (tf_utils.py:557-559).  A step only looks at the samples base .. base + u_stride + 1 of every line
  // (base = u_stride n): that window lives in ONE register, lane 4 k + i holding sample base + i of line k, and is fetched
  // ONE STEP AHEAD by one vector load (the barriers of the step loop wait for LDS only, so the load stays in flight for a
  // whole step); a stage reads its two neighbours out of it with v_readlane.  Lines past K read line 0 and meet zero operators.
  const double* sg = A.signals + (long)b * K * N;
  const int wk = (lane >> 2) & 3, wi = lane & 3;  // u_stride <= 2: four samples per line
  auto window_base = [&](int n) {
    int base = us * n;
    if (base > N - 2) base = N - 2;
    return base < 0 ? 0 : base;
  };
  auto load_window = [&](int base) {
    int idx = base + wi;
    if (idx > N - 1) idx = N - 1;
    return K > 0 ? sg[(long)(wk < K ? wk : 0) * N + idx] : 0.0;
  };
  int wbase = 0;
  double winv = 0.0, nwinv = load_window(window_base(n_begin));
  auto lane_value = [&](double x, int l) {  // x of lane l (uniform l)
    const long long bits = __double_as_longlong(x);
    const int lo32 = __builtin_amdgcn_readlane((int)bits, l), hi32 = __builtin_amdgcn_readlane((int)(bits >> 32), l);
    return __longlong_as_double(((long long)hi32 << 32) | (unsigned int)lo32);
  };
  auto amplitudes = [&](double u, double (&c)[RK]) {
    int lo = (int)floor(u);
    if (lo > N - 2) lo = N - 2;
    if (lo < 0) lo = 0;
    const double f = u - (double)lo;
    int li = __builtin_amdgcn_readfirstlane(lo - wbase);  // 0 .. u_stride
    li = li < 0 ? 0 : (li > 2 ? 2 : li);
#pragma unroll
    for (int k = 0; k < RK; ++k) {
      const double y0 = lane_value(winv, 4 * k + li), y1 = lane_value(winv, 4 * k + li + 1);
      c[k] = fma(f, y1 - y0, y0);
    }
  };
  const int ksteps = (D + 3) >> 2;
  const int aoff = (16 * I + lc) * LD + lr;  // A fragment of row tile I: element (row lc, k = lr) of a K-step
  const int boff = lr * LD + 16 * J + lc;    // B fragment of column tile J: element (k = lr, column lc)
  const int coff = (16 * J + lc) * LD + lr;  // A-pattern read at row tile J: B fragment of a conjugate transpose
  const long ssz = (long)D * D;
  cplx* outp = A.states + (A.seg_count > 0 ? (long)blockIdx.x : (long)b * (A.want_all ? (long)A.n_steps : 1)) * ssz;
  long oidx[4];  // (gen_du_rk4 stacks the propagated vectors as rows: transposed store)
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    const int row = 16 * I + 4 * v + lr;
    oidx[v] = A.transpose_out ? (long)col * D + row : (long)row * D + col;
  }

#ifdef C3P_RHOQ_TIMING
  long long tp[6] = {0, 0, 0, 0, 0, 0};
#define RHOQ_T(i, a, b) tp[i] += (b) - (a)
#define RHOQ_NOW() clock64()
#else
#define RHOQ_T(i, a, b)
#define RHOQ_NOW() 0
#endif
  for (int n = n_begin; n < n_end; ++n) {
    winv = nwinv;
    wbase = window_base(n);
    nwinv = load_window(window_base(n + 1));
    double kr[S][4], ki[S][4], qr[4], qi[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      qr[v] = pr[v];
      qi[v] = pi[v];
    }
    rstatic_for<0, S>([&](auto sc) {
      constexpr int s = decltype(sc)::value;
      // H(t) at this stage's node and the stage argument, own tile
      [[maybe_unused]] const long long t0 = RHOQ_NOW();
      double cur[RK];
      amplitudes(((double)n + rtab_of(SOLVER).node[s]) * (double)us, cur);
      double yr[4], yi[4];
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        double hr = h0r[v];
#pragma unroll
        for (int k = 0; k < RK; ++k) hr = fma(cur[k], hkr[k][v], hr);
        double hi = 0.0;
        if constexpr (CPX) {
          hi = h0i[v];
#pragma unroll
          for (int k = 0; k < RK; ++k) hi = fma(cur[k], hki[k][v], hi);
        }
        yr[v] = pr[v];
        yi[v] = pi[v];
        rstatic_for<0, s>([&](auto jc) {
          constexpr int j = decltype(jc)::value;
          constexpr double a = rtab_of(SOLVER).a[s][j];
          if constexpr (a != 0.0) {
            yr[v] = fma(a, kr[j][v], yr[v]);
            yi[v] = fma(a, ki[j][v], yi[v]);
          }
        });
        if constexpr (LIND) {
          Lr[eoff[v]] = hr + g2i[v];  // L = H - (i/2) G
          Li[eoff[v]] = hi - g2r[v];
          if constexpr (!herm) {
            Rr[eoff[v]] = hr - g2i[v];  // R = H + (i/2) G
            Ri[eoff[v]] = hi + g2r[v];
          }
        } else {
          Lr[eoff[v]] = hr;
          if constexpr (CPX) Li[eoff[v]] = hi;
        }
        Yr[eoff[v]] = yr[v];
        Yi[eoff[v]] = yi[v];
      }
      [[maybe_unused]] const long long t1 = RHOQ_NOW();
      lds_barrier();
      [[maybe_unused]] const long long t2 = RHOQ_NOW();
      d4 accR = {0.0, 0.0, 0.0, 0.0}, accI = {0.0, 0.0, 0.0, 0.0};
      // fragments are fetched one K-step ahead of the MFMAs that use them (two register sets, loop unrolled by two)
      auto kloop = [&](auto nsc, auto&& load, auto&& mma) {
        constexpr int NS = decltype(nsc)::value;  // register sets: fragments are fetched NS - 1 K-steps ahead of their MFMAs
        double f[NS][8];
        rstatic_for<0, NS - 1>([&](auto ic) {
          constexpr int i = decltype(ic)::value;
          if (i < ksteps) load(i, f[i]);
        });
        for (int kk = 0; kk < ksteps; kk += NS) {
          rstatic_for<0, NS>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            if (kk + j < ksteps) {
              if (kk + j + NS - 1 < ksteps) load(kk + j + NS - 1, f[(j + NS - 1) % NS]);
              mma(f[j]);
            }
          });
        }
      };
      if constexpr (herm) {
        // acc = H Y (L Y); [0..1] H as left operand, [2..3] Y as right
        kloop(
            std::integral_constant<int, 4>{},
            [&](int kk, double (&f)[8]) {
              const int ao = aoff + 4 * kk, bo = boff + 4 * kk * LD;
              f[0] = Lr[ao];
              f[2] = Yr[bo];
              f[3] = Yi[bo];
              if constexpr (CPX) f[1] = Li[ao];
            },
            [&](const double (&f)[8]) {
              accR = mfma(f[0], f[2], accR);
              accI = mfma(f[0], f[3], accI);
              if constexpr (CPX) {
                accR = mfma(-f[1], f[3], accR);
                accI = mfma(f[1], f[2], accI);
              }
            });
        if (!prop) {
#pragma unroll
          for (int v = 0; v < 4; ++v) {
            Pr[eoff[v]] = accR[v];
            Pi[eoff[v]] = accI[v];
          }
        }
      } else {
        // acc = H Y - Y H (L Y - Y R); [0..1] H as left operand, [2..3] Y as right, [4..5] Y as left, [6..7] H (R) as right
        kloop(
            std::integral_constant<int, 3>{},
            [&](int kk, double (&f)[8]) {
              const int ao = aoff + 4 * kk, bo = boff + 4 * kk * LD;
              f[0] = Lr[ao];
              f[2] = Yr[bo];
              f[3] = Yi[bo];
              f[4] = Yr[ao];
              f[5] = Yi[ao];
              f[6] = Rr[bo];
              if constexpr (CPX) {
                f[1] = Li[ao];
                f[7] = Ri[bo];
              }
            },
            [&](const double (&f)[8]) {
              if constexpr (CPX) {
                accR = mfma(f[0], f[2], accR);
                accI = mfma(f[0], f[3], accI);
                accR = mfma(-f[1], f[3], accR);
                accI = mfma(f[1], f[2], accI);
                accR = mfma(-f[4], f[6], accR);
                accI = mfma(-f[4], f[7], accI);
                accR = mfma(f[5], f[7], accR);
                accI = mfma(-f[5], f[6], accI);
              } else {
                accR = mfma(f[0], f[2], accR);
                accI = mfma(f[0], f[3], accI);
                accR = mfma(-f[4], f[6], accR);
                accI = mfma(-f[5], f[6], accI);
              }
            });
      }
      bool p_pending = herm && !prop;  // the tiles of H Y are written, not yet fenced
      [[maybe_unused]] const long long t3 = RHOQ_NOW();
      [[maybe_unused]] const long long t4 = t3;
      // k_s = -i dt acc (+ dt sum_m C_m Y C_m^+)
      d4 jR = {0.0, 0.0, 0.0, 0.0}, jI = {0.0, 0.0, 0.0, 0.0};
      if constexpr (CGLOB) {
        // fragment (row 16 T + lc, k = 4 kk + lr) of C_m, zero outside the matrix
        cplx fa[KSMAX], fb[KSMAX];
        auto fetch = [&](cplx (&f)[KSMAX], int m, int T) {
          const int row = 16 * T + lc;
          const cplx* src = A.col_ops + ((long)m * D + (row < D ? row : 0)) * D;
#pragma unroll
          for (int kk = 0; kk < KSMAX; ++kk) {
            const int k = 4 * kk + lr;
            f[kk] = (kk < ksteps && row < D && k < D) ? src[k] : cmake(0, 0);
          }
        };
        if (A.C > 0) fetch(fa, 0, I);
        for (int m = 0; m < A.C; ++m) {
          d4 tR = {0.0, 0.0, 0.0, 0.0}, tI = {0.0, 0.0, 0.0, 0.0};
#if !C3P_RHOQ_CG_LATE
          fetch(fb, m, J);  // in flight behind the first product
#endif
#pragma unroll
          for (int kk = 0; kk < KSMAX; ++kk) {
            if (kk < ksteps) {
              const int bo = boff + 4 * kk * LD;
              const double ar = fa[kk].x, ai = fa[kk].y, br = Yr[bo], bi = Yi[bo];
              tR = mfma(ar, br, tR);
              tI = mfma(ar, bi, tI);
              tR = mfma(-ai, bi, tR);
              tI = mfma(ai, br, tI);
            }
          }
#if C3P_RHOQ_CG_LATE
          fetch(fb, m, J);  // behind the first product's instructions: the first set's registers are free again
#endif
          if (m > 0) lds_barrier();  // the previous jump term's readers of T are done
#pragma unroll
          for (int v = 0; v < 4; ++v) {
            Tr[eoff[v]] = tR[v];
            Ti[eoff[v]] = tI[v];
          }
          lds_barrier();
          p_pending = false;
#if !C3P_RHOQ_CG_LATE
          if (m + 1 < A.C) fetch(fa, m + 1, I);  // in flight behind the second product
#endif
#pragma unroll
          for (int kk = 0; kk < KSMAX; ++kk) {
            if (kk < ksteps) {
              const int ao = aoff + 4 * kk;
              const double ar = Tr[ao], ai = Ti[ao], br = fb[kk].x, bi = fb[kk].y;  // B = C^+: (br, -bi)
              jR = mfma(ar, br, jR);
              jI = mfma(ai, br, jI);
              jR = mfma(ai, bi, jR);
              jI = mfma(-ar, bi, jI);
            }
          }
#if C3P_RHOQ_CG_LATE
          if (m + 1 < A.C) fetch(fa, m + 1, I);
#endif
        }
      } else if constexpr (LIND) {
        for (int m = 0; m < A.C; ++m) {
          const double* cr = Cp + (2 * m) * PL;
          const double* ci = cr + PL;
          d4 tR = {0.0, 0.0, 0.0, 0.0}, tI = {0.0, 0.0, 0.0, 0.0};
          for (int kk = 0; kk < ksteps; ++kk) {
            const int ao = aoff + 4 * kk, bo = boff + 4 * kk * LD;
            const double ar = cr[ao], ai = ci[ao], br = Yr[bo], bi = Yi[bo];
            tR = mfma(ar, br, tR);
            tI = mfma(ar, bi, tI);
            tR = mfma(-ai, bi, tR);
            tI = mfma(ai, br, tI);
          }
          if (m > 0) lds_barrier();  // the previous jump term's readers of T are done
#pragma unroll
          for (int v = 0; v < 4; ++v) {
            Tr[eoff[v]] = tR[v];
            Ti[eoff[v]] = tI[v];
          }
          lds_barrier();
          p_pending = false;
          for (int kk = 0; kk < ksteps; ++kk) {
            const int ao = aoff + 4 * kk, co = coff + 4 * kk;
            const double ar = Tr[ao], ai = Ti[ao], br = cr[co], bi = ci[co];  // B = C^+: (br, -bi)
            jR = mfma(ar, br, jR);
            jI = mfma(ai, br, jI);
            jR = mfma(ai, bi, jR);
            jI = mfma(-ar, bi, jI);
          }
        }
      }
      if (p_pending) lds_barrier();
      [[maybe_unused]] const long long t5 = RHOQ_NOW();
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        double cr_ = accR[v], ci_ = accI[v];
        if constexpr (herm) {  // H Y - (H Y)^+: the mirrored element of the mirrored tile
          if (!prop) {
            const int t = col * LD + 16 * I + 4 * v + lr;
            cr_ -= Pr[t];
            ci_ += Pi[t];
          }
        }
        kr[s][v] = dt * (ci_ + jR[v]);
        ki[s][v] = dt * (jI[v] - cr_);
        constexpr double bs = rtab_of(SOLVER).b[s];
        if constexpr (bs != 0.0) {  // the RK update accumulates as the stages arrive (rk4: one stage live at a time)
          qr[v] = fma(bs, kr[s][v], qr[v]);
          qi[v] = fma(bs, ki[s][v], qi[v]);
        }
      }
      // every wave is done reading the planes of this stage -- not needed on the Hermitian von Neumann path: the next
      // write of L / Y follows this wave's read of P, which follows the barrier behind every wave's products
      if (tail_barrier) lds_barrier();
      [[maybe_unused]] const long long t6 = RHOQ_NOW();
      RHOQ_T(0, t0, t1);
      RHOQ_T(1, t1, t2);
      RHOQ_T(2, t2, t3);
      RHOQ_T(3, t3, t4);
      RHOQ_T(4, t4, t5);
      RHOQ_T(5, t5, t6);
    });
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      pr[v] = qr[v];
      pi[v] = qi[v];
    }
    if (A.want_all) {
#pragma unroll
      for (int v = 0; v < 4; ++v)
        if (valid[v]) outp[(long)n * ssz + oidx[v]] = cmake(pr[v], pi[v]);
    }
    if (A.reset_each_step) {  // per-step propagators: every step starts from the initial state again
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        cplx z = cmake(0, 0);
        if (valid[v]) z = init[(long)(16 * I + 4 * v + lr) * D + col];
        pr[v] = z.x;
        pi[v] = z.y;
      }
    }
  }
#ifdef C3P_RHOQ_TIMING
  if (b == 0 && (tid == 0 || tid == 64 * NT * NT - 1))
    printf("rhoq NT=%d solver=%d mode=%d herm=%d tid=%d: assemble+write %lld, barrier1 %lld, products %lld, fetch wait %lld, (jumps+) barrier2 %lld, k + tail %lld cycles per stage\n",
           NT, SOLVER, MODE, (int)herm, tid, tp[0] / (A.n_steps * S), tp[1] / (A.n_steps * S), tp[2] / (A.n_steps * S), tp[3] / (A.n_steps * S), tp[4] / (A.n_steps * S), tp[5] / (A.n_steps * S));
#endif
  if (!A.want_all) {
#pragma unroll
    for (int v = 0; v < 4; ++v)
      if (valid[v]) outp[oidx[v]] = cmake(pr[v], pi[v]);
  }
}

template <int NT, int SOLVER, int MODE, bool HERM>
hipError_t launch_rho4(const OdeArgs& A, hipStream_t st) {
  const size_t lds = rho_lds_bytes(NT, MODE, A.C);
  if (lds > 48 * 1024) {
    // per launch: the attribute is per DEVICE, and one process may drive several GPUs
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&ode_rhoq_kernel<NT, SOLVER, MODE, HERM>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024);
    if (e != hipSuccess) return e;
  }
  C3P_LAUNCH((ode_rhoq_kernel<NT, SOLVER, MODE, HERM>), dim3((unsigned)(A.B * (A.seg_count > 0 ? A.seg_count : 1))), dim3(64 * NT * NT), lds,
                     st, A);
  return hipGetLastError();
}

// Every instance that could apply is launched: the operators (real / complex) and the symmetry of operators and initial
// state (Hermitian shortcut or the general two-product form) are inspected on the device, per sample, and the instances
// that do not apply exit after the prologue.
template <int NT, int SOLVER, int MODE>
hipError_t launch_rho3(const OdeArgs& A, hipStream_t st) {
  if (A.step == C3P_STEP_PROPAGATOR_ID) return launch_rho4<NT, SOLVER, MODE, true>(A, st);
  if (!A.rho_general) {
    hipError_t e = launch_rho4<NT, SOLVER, MODE, true>(A, st);
    if (e != hipSuccess) return e;
  }
  return launch_rho4<NT, SOLVER, MODE, false>(A, st);
}

template <int NT, int SOLVER>
hipError_t launch_rho2(const OdeArgs& A, hipStream_t st) {
  if (A.step == C3P_STEP_LINDBLAD_ID) {
    return launch_rho3<NT, SOLVER, 2>(A, st);
  }
  hipError_t e = launch_rho3<NT, SOLVER, 0>(A, st);
  if (e != hipSuccess) return e;
  return launch_rho3<NT, SOLVER, 1>(A, st);
}

template <int NT>
hipError_t launch_rho1(const OdeArgs& A, hipStream_t st) {
  switch (A.solver) {
    case 0: return launch_rho2<NT, 0>(A, st);
    case 1: return launch_rho2<NT, 1>(A, st);
    case 2: return launch_rho2<NT, 2>(A, st);
    case 3: return launch_rho2<NT, 3>(A, st);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace

bool c3p_ode_rhoq_supported(const OdeArgs& A) {
  if (c3p_opt_on(C3P_OPT_ode_wg)) return false;
  if (A.D < 17 || A.D > 48 || A.M != A.D || A.K > RK || A.hs || A.N < 2) return false;
  if (A.u_stride < 1 || A.u_stride > 2) return false;
  if (A.step == C3P_STEP_PROPAGATOR_ID)
    return !c3p_opt_on(C3P_OPT_ode_prop_rows) && (A.seg_count == 0 || (!A.want_all && !A.reset_each_step && !A.transpose_out));
  if (A.seg_count > 0) return false;  // (A/B switch: the lane-row column kernel)
  if (A.reset_each_step || A.transpose_out) return false;
  if (A.step == C3P_STEP_VON_NEUMANN_ID) return true;
  if (A.step != C3P_STEP_LINDBLAD_ID) return false;
  if (A.D > 32) return !c3p_opt_on(C3P_OPT_ode_lind_wg);  // collapse operators from memory: 8 planes whatever C is
  return rho_lds_bytes(2, 2, A.C) <= (size_t)(150 * 1024);
}

hipError_t c3p_launch_ode_rhoq(const OdeArgs& A0, hipStream_t st) {
  OdeArgs A = A0;
  A.rho_general = c3p_opt_on(C3P_OPT_ode_rho_general);  // A/B switch: two products per commutator for every input
  return A.D <= 32 ? launch_rho1<2>(A, st) : launch_rho1<3>(A, st);
}
