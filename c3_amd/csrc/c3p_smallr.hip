// Lindblad chains of ONE qubit / qutrit and of two qubits (D = 2, 3, 4: 4 x 4 / 9 x 9 / 16 x 16 superoperators) in REAL arithmetic in the Hermitian basis,
// on the small-D tile layout (round 4).  tf_propagation_lind (c3/libraries/propagation.py:551-585) builds the generator
// L = -i (H (x) 1 - 1 (x) H^T) + dissipator per slice and exponentiates it; for a HERMITIAN Hamiltonian that generator maps
// Hermitian matrices to Hermitian matrices, so in the basis  E_ii, (E_ij + E_ji) / sqrt 2, i (E_ji - E_ij) / sqrt 2  it is a
// REAL Dm x Dm matrix (c3p_regr.hip uses the same fact at D = 7, 8, 9; the change of basis is c3p_hb_row / c3p_hb_col of
// c3p_regd.h).  A real general 9 x 9 product is 18 matrix instructions + a rank-1 tail here against 75 of the complex half-image product of
// c3p_smalld.hip, and everything element-wise is a quarter.
//
// The caller asserts Hermitian Hamiltonians (flag C3P_HERMITIAN_H of c3p_pwc_lindblad: the Python layer sets it after
// checking); without the flag the complex kernels run as before.  Layout: one wavefront = four chains (block b of
// v_mfma_f64_4x4x4_4b_f64 = chain b), every matrix as NB x NB register tiles (lane (r, c) of tile (I, J) holds
// M[4 I + r][4 J + c]); the left operand of a product goes through a per-chain row-major LDS image (row stride 4 NB + 1)
// and is read back in the A layout.  exp = T18 (Bader-Blanes-Casas, five products) + s squarings, as the complex loop.
// Segment products are turned back into the reference's vectorisation (U = T^+ U' T) by the chain that formed them; the
// supplied-matrix mode of the complex small-D chain kernel multiplies them in order and applies the frame phases.
#include "c3p_common.h"
#include "c3p_kernels.h"
#include "c3p_midd.h"
#include "c3p_regd.h"
#include "c3p_smallr.h"

extern __shared__ __attribute__((aligned(16))) double c3p_sr_lds[];

namespace {

template <int DM>
struct RG {
  static constexpr int NB = (DM + 3) / 4;
  // image row stride and chain stride (doubles).  64-bit LDS reads are served 32 lanes at a time on 64 dword banks: the A-layout
  // reads of the four chains of a wavefront (address b RIMG + c WR + r) were 4-way conflicted at Dm = 16 with the natural
  // 17 / 272 (68 % of the LDS cycles of both kernels, profiles/r04/smallr_pmc.json); 18 / 296 makes them conflict-free (writes
  // two-way) -- brute force over strides with the bank rule.  Dm = 9: 13 / 156 (reads two-way, writes free); Dm = 4: 5 / 20.
  static constexpr int WR = NB == 4 ? 18 : 4 * NB + 1;
  static constexpr int RIMG = NB == 4 ? 296 : 4 * NB * WR;
  static constexpr int TILES = NB * NB * 16;     // table body: tile-major, 16 elements per tile in lane order r * 4 + c
  static constexpr int TABD = TILES + 4;         // + {mu, ||G' - mu||_1, max |Im|, max |Re|}
};

__device__ __forceinline__ double sr_mfma4(double a, double b, double c) { return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0); }
typedef __attribute__((address_space(3))) const volatile double sr_lds_cvd;
__device__ __forceinline__ double sr_ld(const double* p) { return *(sr_lds_cvd*)p; }
__device__ __forceinline__ void sr_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// ---- tables: G' = Re(T G T^+), trace shifted, tile-major ------------------------------------------------------------------
// (transpose = 2: blockIdx.y = 0 tabulates G' into tables, blockIdx.y = 1 its transpose into tables_t -- one launch for both)
__global__ void __launch_bounds__(64) smallr_prep_kernel(RegdPrepArgs P, double* tables_in, double* tables_t, int* tabflag, int transpose_in) {
  const int transpose = transpose_in == 2 ? (int)blockIdx.y : transpose_in;
  double* tables = (transpose_in == 2 && blockIdx.y == 1) ? tables_t : tables_in;
  __shared__ double red0[64], red1[64];
  __shared__ double mu_s;
  const int tid = threadIdx.x;
  const int ti = blockIdx.x % (1 + P.K);
  const int sample = blockIdx.x / (1 + P.K);
  const int D = P.Dm, Dh = P.Dh;
  const int NB = (D + 3) / 4;
  const cplx* h = (ti == 0) ? P.h0 + (long)sample * P.h0_bstride : P.hks + (long)sample * P.hks_bstride + (long)(ti - 1) * Dh * Dh;
  // element (row, col) of L0 = dt (clp - i (H0 (x) I - I (x) H0^T)) / Lk = -i dt (Hk (x) I - I (x) Hk^T) (propagation.py:565-582)
  auto gelem = [&](int row, int col) -> cplx {
    const int i = row / Dh, j = row - i * Dh, k = col / Dh, l = col - k * Dh;
    cplx v = (ti == 0) ? P.clp[(long)row * D + col] : cmake(0, 0);
    if (j == l) {
      const cplx x = h[i * Dh + k];
      v.x += x.y;
      v.y -= x.x;
    }
    if (i == k) {
      const cplx x = h[l * Dh + j];
      v.x -= x.y;
      v.y += x.x;
    }
    return cscale(v, P.dt);
  };
  auto helem = [&](int a, int b) -> cplx {  // element (a, b) of T G T^+
    int ia[2], ib[2];
    cplx ta[2], tb[2];
    const int na = c3p_hb_row(a, Dh, ia, ta), nb = c3p_hb_row(b, Dh, ib, tb);
    cplx s = cmake(0.0, 0.0);
    for (int x = 0; x < na; ++x)
      for (int y = 0; y < nb; ++y) cfma(s, cmul(ta[x], cconj(tb[y])), gelem(ia[x], ib[y]));
    return s;
  };
  // every element of T G T^+ by its own thread first (Dm^2 <= 81), norms and the table from the shared copy
  __shared__ double hre[256], him[256];
  for (int e = tid; e < D * D; e += 64) {
    const cplx v = transpose ? helem(e % D, e / D) : helem(e / D, e % D);  // (transpose: the tables of G'^T, for the backward sweep)
    hre[e] = v.x;
    him[e] = v.y;
  }
  __syncthreads();
  if (tid == 0) mu_s = 0.0;  // no trace shift in the real kernels (c3p_regr.hip: regr_prep_kernel); the table slot stays
  __syncthreads();
  const double mu = mu_s;
  double cs = 0, mre = 0, mim = 0, csym = 0;
  for (int j = tid; j < D; j += 64) {
    double sm = 0, ssym = 0;
    for (int i = 0; i < D; ++i) {
      const double vx = hre[i * D + j], vy = him[i * D + j];
      mre = fmax(mre, fabs(vx));
      mim = fmax(mim, fabs(vy));
      sm += fabs(i == j ? vx - mu : vx);
      ssym += fabs(0.5 * (vx + hre[j * D + i]) - (i == j ? mu : 0.0));  // symmetric (non-skew) part: see c3p_t18_tab
    }
    cs = fmax(cs, sm);
    csym = fmax(csym, ssym);
  }
  red0[tid] = cs;
  red1[tid] = mre;
  __syncthreads();
  double nrm = 0, gmax = 0;
  if (tid == 0)
    for (int i = 0; i < 64; ++i) {
      nrm = fmax(nrm, red0[i]);
      gmax = fmax(gmax, red1[i]);
    }
  __syncthreads();
  red0[tid] = mim;
  red1[tid] = csym;
  __syncthreads();
  const int TILES = NB * NB * 16;
  double* out = tables + ((long)sample * (1 + P.K) + ti) * (TILES + 4);
  for (int e = tid; e < TILES; e += 64) {
    const int tile = e >> 4, idx = e & 15;
    const int I = tile / NB, J = tile - I * NB;
    const int row = 4 * I + (idx >> 2), col = 4 * J + (idx & 3);
    double g = 0.0;
    if (row < D && col < D) {
      g = hre[row * D + col];
      if (row == col) g -= mu;
    }
    out[e] = g;
  }
  if (tid == 0) {
    double gim = 0, nsym = 0;
    for (int i = 0; i < 64; ++i) gim = fmax(gim, red0[i]), nsym = fmax(nsym, red1[i]);
    out[TILES + 0] = mu;
    out[TILES + 1] = nrm;
    out[TILES + 2] = gim;
    out[TILES + 3] = nsym;  // 1-norm of the symmetric part of the (trace-shifted) real generator
    tabflag[sample * (1 + P.K) + ti] = (gim <= 1e-14 * gmax) ? 1 : 0;
  }
}

// ---- the chain kernel --------------------------------------------------------------------------------------------------
template <int DM>
__global__ void __launch_bounds__(64) smallr_chain_kernel(SmallRArgs A) {
  using G = RG<DM>;
  constexpr int NB = G::NB, WR = G::WR, RIMG = G::RIMG, TABD = G::TABD, TILES = G::TILES;
  typedef double RMat[NB][NB];
  const int lane = threadIdx.x;
  const int r = lane >> 4, b = (lane >> 2) & 3, c = lane & 3, idx16 = r * 4 + c;
  const int K = A.K;
  double* tab = c3p_sr_lds;                    // (1 + K) tables of this wavefront's sample
  double* img = tab + (1 + K) * TABD;          // four chain images
  double* sg = img + 4 * RIMG;                 // 4 chains x K x Lmax control amplitudes

  const long chain = (long)blockIdx.x * 4 + b;
  const long nchains = (long)A.B * A.S;
  const bool valid = chain < nchains;
  const long cc = valid ? chain : nchains - 1;
  const int sample = (int)(cc / A.S);
  const int seg = (int)(cc - (long)sample * A.S);
  const int n0 = (int)(((long)seg * A.N) / A.S);
  const int n1 = (int)(((long)(seg + 1) * A.N) / A.S);
  const int len = n1 - n0;

  // (tables per sample: S % 4 == 0 and the four chains of a wavefront share the sample; otherwise all samples share them)
  const double* gt0 = A.tables + (long)(A.tab_per_sample ? sample : 0) * (1 + K) * TABD;
  for (int e = lane; e < (1 + K) * TABD; e += 64) tab[e] = gt0[e];
  __syncthreads();
  double nrm = tab[TILES + 1], nsym = tab[TILES + 3];
  for (int k = 0; k < K; ++k) {
    const double* s = A.signals + ((long)sample * K + k) * A.N + n0;
    double cmax = 0.0;
    for (int t = idx16; t < A.Lmax; t += 16) {
      const double v = (valid && t < len) ? s[t] : 0.0;
      sg[(b * K + k) * A.Lmax + t] = v;
      cmax = fmax(cmax, fabs(v));
    }
    cmax = fmax(cmax, __shfl_xor(cmax, 1));
    cmax = fmax(cmax, __shfl_xor(cmax, 2));
    cmax = fmax(cmax, __shfl_xor(cmax, 16));
    cmax = fmax(cmax, __shfl_xor(cmax, 32));
    nrm = fma(cmax, tab[(k + 1) * TABD + TILES + 1], nrm);
    nsym = fma(cmax, tab[(k + 1) * TABD + TILES + 3], nsym);
  }
  __syncthreads();
  nrm = fmax(nrm, __shfl_xor(nrm, 4));
  nrm = fmax(nrm, __shfl_xor(nrm, 8));
  nsym = fmax(nsym, __shfl_xor(nsym, 4));
  nsym = fmax(nsym, __shfl_xor(nsym, 8));
  // round 6: economised T18 parameters (radius 2.0) while the generator is skew-symmetric up to a small symmetric part
  const int econ = __builtin_amdgcn_readfirstlane((int)(nsym <= C3P_T18N_MAX_NONNORMAL && !(A.no_t18n & 1)));
  const double* tc = c3p_t18_tab[econ];
  int ps = 0;
  {
    double pth = econ ? C3P_T18N_THETA : C3P_T18_THETA;
    while (pth < nrm && ps < 40) {
      pth *= 2.0;
      ++ps;
    }
  }
  // ... or the four-product scheme for normal generators (c3p_common.h: c3p_e4n, radius 1.35) where it saves a product
  int e4 = 0;
  if (econ && !(A.no_t18n & 2)) {
    int s4 = 0;
    double pth = C3P_E4N_THETA;
    while (pth < nrm && s4 < 40) {
      pth *= 2.0;
      ++s4;
    }
    if (4 + s4 < 5 + ps) e4 = 1, ps = s4;
  }
  e4 = __builtin_amdgcn_readfirstlane(e4);
  ps = __builtin_amdgcn_readfirstlane(ps);
  const double scale = ldexp(1.0, -ps);

  const int woff = b * RIMG + r * WR + c;  // D-layout write: + (4 I) WR + 4 J
  const int roff = b * RIMG + c * WR + r;  // A-layout read:  + (4 I) WR + 4 K
  auto to_image = [&](const RMat& M) {
    sr_sync();
#pragma unroll
    for (int I = 0; I < NB; ++I)
#pragma unroll
      for (int J = 0; J < NB; ++J) img[woff + 4 * I * WR + 4 * J] = M[I][J];
    sr_sync();
  };
  // acc += (image) Bm.  Dm = 1 (mod 4): the K dimension is exact -- NB - 1 steps of four on the matrix cores and the last
  // index as a rank-1 update (column Dm-1 of the left operand from the image, row Dm-1 of the right one from the lanes
  // (0, c) of its last tile row): 18 + a 9-FMA tail instead of 27 matrix instructions at Dm = 9
  constexpr bool TAIL = (DM % 4 == 1) && DM > 4;
  constexpr int KM = TAIL ? NB - 1 : NB;
  const int row0_lane = 4 * b + c;
  auto mm = [&](const RMat& Bm, RMat& acc) {
#pragma unroll
    for (int Kk = 0; Kk < KM; ++Kk) {
      double a[NB];
#pragma unroll
      for (int I = 0; I < NB; ++I) a[I] = sr_ld(img + roff + 4 * I * WR + 4 * Kk);
#pragma unroll
      for (int I = 0; I < NB; ++I)
#pragma unroll
        for (int J = 0; J < NB; ++J) acc[I][J] = sr_mfma4(a[I], Bm[Kk][J], acc[I][J]);
    }
    if constexpr (TAIL) {
      double a8[NB], b8[NB];
#pragma unroll
      for (int I = 0; I < NB; ++I) a8[I] = sr_ld(img + b * RIMG + (4 * I + r) * WR + (DM - 1));
#pragma unroll
      for (int J = 0; J < NB; ++J) b8[J] = __shfl(Bm[NB - 1][J], row0_lane);
#pragma unroll
      for (int I = 0; I < NB; ++I)
#pragma unroll
        for (int J = 0; J < NB; ++J) acc[I][J] = fma(a8[I], b8[J], acc[I][J]);
    }
  };
  // out = c0 I + cx X + c2 A2 + c3 A3 + c6 A6
  auto comb = [&](RMat& out, double c0, double cx, double c2, double c3, double c6, const RMat& X, const RMat& A2, const RMat& A3, const RMat& A6) {
#pragma unroll
    for (int I = 0; I < NB; ++I)
#pragma unroll
      for (int J = 0; J < NB; ++J) {
        double v = cx * X[I][J];
        v = fma(c2, A2[I][J], v);
        v = fma(c3, A3[I][J], v);
        v = fma(c6, A6[I][J], v);
        if (I == J) v += (r == c && 4 * I + r < DM) ? c0 : 0.0;
        out[I][J] = v;
      }
  };
  auto zero = [&](RMat& M) {
#pragma unroll
    for (int I = 0; I < NB; ++I)
#pragma unroll
      for (int J = 0; J < NB; ++J) M[I][J] = 0.0;
  };

  // a REAL matrix M' (in the chain's image) -> f T^+ M' T, the complex matrix in the reference's vectorisation (c3p_hb_col: at
  // most two non-zeros per column of T); the 16 lanes of a chain share its Dm^2 elements
  constexpr int Dh = (DM == 4) ? 2 : (DM == 9 ? 3 : 4);
  auto image_to_complex = [&](cplx* dst, double f, bool store) {
    for (int e = idx16; e < DM * DM; e += 16) {
      const int al = e / DM, be = e - al * DM;
      int ia[2], ib[2];
      cplx ta[2], tb[2];
      const int na = c3p_hb_col(al, Dh, ia, ta), nb = c3p_hb_col(be, Dh, ib, tb);
      cplx z = cmake(0.0, 0.0);
      for (int x = 0; x < na; ++x)
        for (int y = 0; y < nb; ++y) {
          const cplx cf = cmul(cconj(ta[x]), tb[y]);
          const double v = img[b * RIMG + ia[x] * WR + ib[y]];
          z.x = fma(cf.x, v, z.x);
          z.y = fma(cf.y, v, z.y);
        }
      if (store) dst[e] = cmake(f * z.x, f * z.y);
    }
  };

  RMat U;
  zero(U);
  double mus = 0.0;
  for (int t = 0; t < A.Lmax; ++t) {
    const bool act = valid && t < len;
    const double sc = act ? scale : 0.0;  // chains past their segment end take X = 0, E = 1
    double mu = act ? tab[TILES + 0] : 0.0;
    RMat X;
#pragma unroll
    for (int I = 0; I < NB; ++I)
#pragma unroll
      for (int J = 0; J < NB; ++J) X[I][J] = sc * tab[(I * NB + J) * 16 + idx16];
    for (int k = 0; k < K; ++k) {
      const double c0 = sg[(b * K + k) * A.Lmax + t];  // zero padded past the segment end
      const double ck = sc * c0;
      const double* tk = tab + (k + 1) * TABD;
      mu = fma(c0, tk[TILES + 0], mu);
#pragma unroll
      for (int I = 0; I < NB; ++I)
#pragma unroll
        for (int J = 0; J < NB; ++J) X[I][J] = fma(ck, tk[(I * NB + J) * 16 + idx16], X[I][J]);
    }
    RMat A2, A3, A6, P, acc;
    zero(A2), zero(A3), zero(A6);
    to_image(X);
    mm(X, A2);
    if (e4) {
      // four products: y0 = A2 (e0 A2 + e1 X); y1 = (y0 + e2 A2 + e3 X)(y0 + e4 A2) + e5 y0 + e6 A2;
      // exp = (y1 + e7 A2 + e8 X)(y1 + e9 y0 + e10 X) + e11 y1 + e12 y0 + e13 A2 + e14 X + e15 I   (A3 holds y0, A6 y1)
      {
        RMat R;
#pragma unroll
        for (int I = 0; I < NB; ++I)
#pragma unroll
          for (int J = 0; J < NB; ++J) R[I][J] = fma(c3p_e4n[0], A2[I][J], c3p_e4n[1] * X[I][J]);
        to_image(A2);
        mm(R, A3);
      }
      {
        RMat L, R;
#pragma unroll
        for (int I = 0; I < NB; ++I)
#pragma unroll
          for (int J = 0; J < NB; ++J) L[I][J] = fma(c3p_e4n[2], A2[I][J], fma(c3p_e4n[3], X[I][J], A3[I][J]));
        to_image(L);
#pragma unroll
        for (int I = 0; I < NB; ++I)
#pragma unroll
          for (int J = 0; J < NB; ++J) {
            R[I][J] = fma(c3p_e4n[4], A2[I][J], A3[I][J]);
            A6[I][J] = fma(c3p_e4n[5], A3[I][J], c3p_e4n[6] * A2[I][J]);
          }
        mm(R, A6);
      }
      {
        RMat L, R;
#pragma unroll
        for (int I = 0; I < NB; ++I)
#pragma unroll
          for (int J = 0; J < NB; ++J) L[I][J] = fma(c3p_e4n[7], A2[I][J], fma(c3p_e4n[8], X[I][J], A6[I][J]));
        to_image(L);
#pragma unroll
        for (int I = 0; I < NB; ++I)
#pragma unroll
          for (int J = 0; J < NB; ++J) {
            R[I][J] = fma(c3p_e4n[9], A3[I][J], fma(c3p_e4n[10], X[I][J], A6[I][J]));
            double v = fma(c3p_e4n[11], A6[I][J], c3p_e4n[12] * A3[I][J]);
            v = fma(c3p_e4n[13], A2[I][J], v);
            v = fma(c3p_e4n[14], X[I][J], v);
            if (I == J) v += (r == c && 4 * I + r < DM) ? c3p_e4n[15] : 0.0;
            P[I][J] = v;
          }
        mm(R, P);
      }
    } else {
    // T18 (Bader-Blanes-Casas): A2 = X X, A3 = X A2, A6 = A3 A3, A9 = B1 B5 + B4, exp = B2 + (B3 + A9) A9
    mm(A2, A3);
    to_image(A3);
    mm(A3, A6);
    {
      RMat B1, B5;
      comb(B1, 0.0, tc[C3P_I_A11], tc[C3P_I_A21], tc[C3P_I_A31], 0.0, X, A2, A3, A6);
      to_image(B1);
      comb(B5, 0.0, 0.0, tc[C3P_I_B24], tc[C3P_I_B34], tc[C3P_I_B64], X, A2, A3, A6);
      comb(acc, tc[C3P_I_B03], tc[C3P_I_B13], tc[C3P_I_B23], tc[C3P_I_B33], tc[C3P_I_B63], X, A2, A3, A6);
      mm(B5, acc);  // A9
    }
    {
      RMat L;
      comb(L, tc[C3P_I_B02], tc[C3P_I_B12], tc[C3P_I_B22], tc[C3P_I_B32], tc[C3P_I_B62], X, A2, A3, A6);
#pragma unroll
      for (int I = 0; I < NB; ++I)
#pragma unroll
        for (int J = 0; J < NB; ++J) L[I][J] += acc[I][J];
      to_image(L);
    }
    comb(P, 0.0, tc[C3P_I_B11], tc[C3P_I_B21], tc[C3P_I_B31], tc[C3P_I_B61], X, A2, A3, A6);
    mm(acc, P);
    }
    for (int it = 0; it < ps; ++it) {
      to_image(P);
      RMat Q;
      zero(Q);
      mm(P, Q);
#pragma unroll
      for (int I = 0; I < NB; ++I)
#pragma unroll
        for (int J = 0; J < NB; ++J) P[I][J] = Q[I][J];
    }
    if (A.dus_real && act) {
      // the LOCAL prefix of slice t -- the product of the slices of this segment in front of it, with their trace shifts -- for
      // the real backward sweep (M_n = A~_n Q_n^T with the prefix at the segment start folded into the adjoint: no slice
      // propagators, no second forward walk there)
      const double em = exp(mus);
      double* dr = A.dus_real + ((long)sample * A.N + n0 + t) * DM * DM;
#pragma unroll
      for (int I = 0; I < NB; ++I)
#pragma unroll
        for (int J = 0; J < NB; ++J) {
          const int row = 4 * I + r, col = 4 * J + c;
          if (row < DM && col < DM) dr[row * DM + col] = (t == 0) ? ((row == col) ? 1.0 : 0.0) : em * U[I][J];
        }
    }
    if (A.dUs_out) {  // the slice propagator e^{mu} exp(X - mu), for the backward sweep of the gradient entries
      to_image(P);
      image_to_complex(A.dUs_out + ((long)sample * A.N + n0 + (act ? t : 0)) * DM * DM, exp(mu), act);
    }
    if (t == 0) {
#pragma unroll
      for (int I = 0; I < NB; ++I)
#pragma unroll
        for (int J = 0; J < NB; ++J) U[I][J] = P[I][J];
    } else {
      if (!A.dUs_out) to_image(P);
      RMat V;
      zero(V);
      mm(U, V);
#pragma unroll
      for (int I = 0; I < NB; ++I)
#pragma unroll
        for (int J = 0; J < NB; ++J) U[I][J] = act ? V[I][J] : U[I][J];
    }
    mus += mu;
  }
  if (A.seg_real && valid) {
    const double em = exp(mus);
    double* dr = A.seg_real + cc * DM * DM;
#pragma unroll
    for (int I = 0; I < NB; ++I)
#pragma unroll
      for (int J = 0; J < NB; ++J) {
        const int row = 4 * I + r, col = 4 * J + c;
        if (row < DM && col < DM) dr[row * DM + col] = em * U[I][J];
      }
  }
  // the REAL segment product U' -> the complex matrix e^{sum mu} T^+ U' T
  to_image(U);
  image_to_complex(A.seg_out + cc * DM * DM, exp(mus), valid);
}

// ---- segment scan, real: pre[j] = S_{j-1} ... S_0 (pre[0] = 1), suf[j] = S_{j+1}^T ... S_{S-1}^T U_bar' ------------------------
// one wavefront per (sample, direction): the chains of S - 1 dependent Dm x Dm products run from LDS without workgroup barriers
template <int DM>
__global__ void __launch_bounds__(64) smallr_scan_kernel(const double* seg, const double* ubar, int S, double* pre, double* suf) {
  constexpr int M2 = DM * DM;
  __shared__ double cur[2][DM * DM], sg[DM * DM];
  const int lane = threadIdx.x;
  const long bidx = blockIdx.x;
  const bool fwd = blockIdx.y == 0;
  const double* sb = seg + bidx * S * M2;
  double* out = (fwd ? pre : suf) + bidx * S * M2;
  int w = 0;
  for (int e = lane; e < M2; e += 64) cur[0][e] = fwd ? ((e / DM == e % DM) ? 1.0 : 0.0) : ubar[bidx * M2 + e];
  sr_sync();
  // the segment product of the NEXT step is requested one step ahead (each step is otherwise a dependent global load:
  // 0.9 us per step for a 9 x 9 product)
  constexpr int PF = (M2 + 63) / 64;
  double nxt[PF];
  auto fetch = [&](int j) {
    const double* sj = sb + (long)j * M2;  // forward: S_j takes pre[j] to pre[j + 1]; backward: S_j^T takes suf[j] to suf[j - 1]
#pragma unroll
    for (int i = 0; i < PF; ++i) {
      const int e = lane + 64 * i;
      nxt[i] = e < M2 ? sj[e] : 0.0;
    }
  };
  if (S > 1) fetch(fwd ? 0 : S - 1);
  for (int step = 0; step < S; ++step) {
    const int j = fwd ? step : S - 1 - step;
    for (int e = lane; e < M2; e += 64) out[(long)j * M2 + e] = cur[w][e];
    if (step == S - 1) break;
#pragma unroll
    for (int i = 0; i < PF; ++i) {
      const int e = lane + 64 * i;
      if (e < M2) sg[e] = nxt[i];
    }
    if (step + 1 < S - 1) fetch(fwd ? step + 1 : S - 2 - step);
    sr_sync();
    for (int e = lane; e < M2; e += 64) {
      const int i = e / DM, jx = e - i * DM;
      double acc = 0.0;
#pragma unroll
      for (int k = 0; k < DM; ++k) acc = fma(fwd ? sg[i * DM + k] : sg[k * DM + i], cur[w][k * DM + jx], acc);
      cur[w ^ 1][e] = acc;
    }
    sr_sync();
    w ^= 1;
  }
}

// The same scan in blocks (S >= 16): NBK wavefronts per (sample, direction), wavefront w scans its block of segments locally
// (local prefixes / suffixes to the output, the product of the block to LDS), wavefront 0 chains the NBK block products, then
// every wavefront multiplies its local results by the offset of its block: ~3 S / NBK dependent products instead of S - 1.
template <int DM, int NBK>
__global__ void __launch_bounds__(64 * NBK) smallr_scan_blocked_kernel(const double* seg, const double* ubar, int S, double* pre, double* suf) {
  constexpr int M2 = DM * DM;
  __shared__ double curs[NBK][2][M2], sgs[NBK][M2], blk[NBK][M2], off[NBK][M2];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const long bidx = blockIdx.x;
  const bool fwd = blockIdx.y == 0;
  const double* sb = seg + bidx * S * M2;
  double* out = (fwd ? pre : suf) + bidx * S * M2;
  const int Lb = (S + NBK - 1) / NBK;
  const int j0 = wv * Lb, j1 = (j0 + Lb < S) ? j0 + Lb : S;  // this wavefront's segments [j0, j1)
  const int nloc = j1 > j0 ? j1 - j0 : 0;
  double(*cur)[M2] = curs[wv];
  double* sg = sgs[wv];
  // C = op(sg) cur[w] -> cur[w ^ 1]
  auto step_mm = [&](int w) {
    for (int e = lane; e < M2; e += 64) {
      const int i = e / DM, jx = e - i * DM;
      double acc = 0.0;
#pragma unroll
      for (int k = 0; k < DM; ++k) acc = fma(fwd ? sg[i * DM + k] : sg[k * DM + i], cur[w][k * DM + jx], acc);
      cur[w ^ 1][e] = acc;
    }
  };
  // phase 1: local scan of the block, starting from the identity; the block product S_{j1-1} ... S_{j0} (its transpose chain
  // S_{j0}^T ... S_{j1-1}^T for the suffix direction) ends in blk[wv]
  int w = 0;
  for (int e = lane; e < M2; e += 64) cur[0][e] = (e / DM == e % DM) ? 1.0 : 0.0;
  sr_sync();
  for (int step = 0; step < nloc; ++step) {
    const int j = fwd ? j0 + step : j1 - 1 - step;
    for (int e = lane; e < M2; e += 64) out[(long)j * M2 + e] = cur[w][e];  // local prefix in front of / suffix behind segment j
    for (int e = lane; e < M2; e += 64) sg[e] = sb[(long)j * M2 + e];
    sr_sync();
    step_mm(w);
    sr_sync();
    w ^= 1;
  }
  for (int e = lane; e < M2; e += 64) blk[wv][e] = cur[w][e];
  __threadfence_block();
  __syncthreads();
  // phase 2: offsets of the blocks (wavefront 0): forward off[0] = 1, off[b + 1] = blk[b] off[b]; suffix off[NBK - 1] = U_bar',
  // off[b - 1] = blk[b] off[b]  (blk of the suffix direction is already the transposed chain)
  if (wv == 0) {
    for (int e = lane; e < M2; e += 64) cur[0][e] = fwd ? ((e / DM == e % DM) ? 1.0 : 0.0) : ubar[bidx * M2 + e];
    sr_sync();
    int ww = 0;
    for (int step = 0; step < NBK; ++step) {
      const int bq = fwd ? step : NBK - 1 - step;
      for (int e = lane; e < M2; e += 64) off[bq][e] = cur[ww][e];
      if (step == NBK - 1) break;
      sr_sync();
      for (int e = lane; e < M2; e += 64) {  // plain product: blk[bq] cur
        const int i = e / DM, jx = e - i * DM;
        double acc = 0.0;
#pragma unroll
        for (int k = 0; k < DM; ++k) acc = fma(blk[bq][i * DM + k], cur[ww][k * DM + jx], acc);
        cur[ww ^ 1][e] = acc;
      }
      sr_sync();
      ww ^= 1;
    }
  }
  __syncthreads();
  // phase 3: out[j] = local[j] off[block]
  for (int step = 0; step < nloc; ++step) {
    const int j = j0 + step;
    double* o = out + (long)j * M2;
    for (int e = lane; e < M2; e += 64) sg[e] = o[e];
    sr_sync();
    for (int e = lane; e < M2; e += 64) {
      const int i = e / DM, jx = e - i * DM;
      double acc = 0.0;
#pragma unroll
      for (int k = 0; k < DM; ++k) acc = fma(sg[i * DM + k], off[wv][k * DM + jx], acc);
      o[e] = acc;
    }
    sr_sync();
  }
}

// ---- backward sweep --------------------------------------------------------------------------------------------------------
template <int DM>
// (Dm <= 9: 256 registers, ~30 spilled, TWO wavefronts per SIMD -- the sweep is a chain of dependent products at 22 % matrix-pipe
// occupancy with one; Dm = 16 keeps 512 registers: 326 would spill)
__global__ void __launch_bounds__(64, (DM <= 9 ? 2 : 1)) smallr_grad_kernel(SmallRGradArgs A) {
  using G = RG<DM>;
  constexpr int NB = G::NB, WR = G::WR, RIMG = G::RIMG, TABD = G::TABD, TILES = G::TILES;
  constexpr int M2 = DM * DM;
  typedef double RMat[NB][NB];
  const int lane = threadIdx.x;
  const int r = lane >> 4, b = (lane >> 2) & 3, c = lane & 3, idx16 = r * 4 + c;
  const int K = A.K;
  double* tab = c3p_sr_lds;             // G' tables: inner products
  double* tabt = tab + (1 + K) * TABD;  // G'^T tables: the matrix that is exponentiated
  double* img0 = tabt + (1 + K) * TABD;
  double* img1 = img0 + 4 * RIMG;
  double* sg = img1 + 4 * RIMG;

  const long chain = (long)blockIdx.x * 4 + b;
  const long nchains = (long)A.B * A.S;
  const bool valid = chain < nchains;
  const long cc = valid ? chain : nchains - 1;
  const int sample = (int)(cc / A.S);
  const int seg = (int)(cc - (long)sample * A.S);
  const int n0 = (int)(((long)seg * A.N) / A.S);
  const int n1 = (int)(((long)(seg + 1) * A.N) / A.S);
  const int len = n1 - n0;

  const long tsz = (long)(1 + K) * TABD;
  const double* gt0 = A.tables + (long)(A.tab_per_sample ? sample : 0) * tsz;
  const double* gh0 = A.tables_t + (long)(A.tab_per_sample ? sample : 0) * tsz;
  for (int e = lane; e < (int)tsz; e += 64) {
    tab[e] = gt0[e];
    tabt[e] = gh0[e];
  }
  __syncthreads();
  double nrm = tabt[TILES + 1], nsym = tabt[TILES + 3];
  for (int k = 0; k < K; ++k) {
    const double* s = A.signals + ((long)sample * K + k) * A.N + n0;
    double cmax = 0.0;
    for (int t = idx16; t < A.Lmax; t += 16) {
      const double v = (valid && t < len) ? s[t] : 0.0;
      sg[(b * K + k) * A.Lmax + t] = v;
      cmax = fmax(cmax, fabs(v));
    }
    cmax = fmax(cmax, __shfl_xor(cmax, 1));
    cmax = fmax(cmax, __shfl_xor(cmax, 2));
    cmax = fmax(cmax, __shfl_xor(cmax, 16));
    cmax = fmax(cmax, __shfl_xor(cmax, 32));
    nrm = fma(cmax, tabt[(k + 1) * TABD + TILES + 1], nrm);
    nsym = fma(cmax, tabt[(k + 1) * TABD + TILES + 3], nsym);
  }
  __syncthreads();
  nrm = fmax(nrm, __shfl_xor(nrm, 4));
  nrm = fmax(nrm, __shfl_xor(nrm, 8));
  nsym = fmax(nsym, __shfl_xor(nsym, 4));
  nsym = fmax(nsym, __shfl_xor(nsym, 8));
  const int econ = __builtin_amdgcn_readfirstlane((int)(nsym <= C3P_T18N_MAX_NONNORMAL && !(A.no_t18n & 1)));  // as in the forward kernel
  const double* tc = c3p_t18_tab[econ];
  int ps = 0;
  {
    double pth = econ ? C3P_T18N_THETA : C3P_T18_THETA;
    while (pth < nrm && ps < 40) {
      pth *= 2.0;
      ++ps;
    }
  }
  ps = __builtin_amdgcn_readfirstlane(ps);
  const double scale = ldexp(1.0, -ps);

  const int woff = b * RIMG + r * WR + c;
  const int roff = b * RIMG + c * WR + r;
  constexpr bool TAIL = (DM % 4 == 1) && DM > 4;
  constexpr int KM = TAIL ? NB - 1 : NB;
  const int row0_lane = 4 * b + c;
  auto to_image = [&](double* img, const RMat& M) {
    sr_sync();
#pragma unroll
    for (int I = 0; I < NB; ++I)
#pragma unroll
      for (int J = 0; J < NB; ++J) img[woff + 4 * I * WR + 4 * J] = M[I][J];
    sr_sync();
  };
  auto mm = [&](const double* img, const RMat& Bm, RMat& acc) {  // acc += (image) Bm
#pragma unroll
    for (int Kk = 0; Kk < KM; ++Kk) {
      double a[NB];
#pragma unroll
      for (int I = 0; I < NB; ++I) a[I] = sr_ld(img + roff + 4 * I * WR + 4 * Kk);
#pragma unroll
      for (int I = 0; I < NB; ++I)
#pragma unroll
        for (int J = 0; J < NB; ++J) acc[I][J] = sr_mfma4(a[I], Bm[Kk][J], acc[I][J]);
    }
    if constexpr (TAIL) {
      double a8[NB], b8[NB];
#pragma unroll
      for (int I = 0; I < NB; ++I) a8[I] = sr_ld(img + b * RIMG + (4 * I + r) * WR + (DM - 1));
#pragma unroll
      for (int J = 0; J < NB; ++J) b8[J] = __shfl(Bm[NB - 1][J], row0_lane);
#pragma unroll
      for (int I = 0; I < NB; ++I)
#pragma unroll
        for (int J = 0; J < NB; ++J) acc[I][J] = fma(a8[I], b8[J], acc[I][J]);
    }
  };
  auto comb = [&](RMat& out, double c0, double cx, double c2, double c3, double c6, const RMat& X, const RMat& A2, const RMat& A3, const RMat& A6) {
#pragma unroll
    for (int I = 0; I < NB; ++I)
#pragma unroll
      for (int J = 0; J < NB; ++J) {
        double v = cx * X[I][J];
        v = fma(c2, A2[I][J], v);
        v = fma(c3, A3[I][J], v);
        v = fma(c6, A6[I][J], v);
        if (I == J) v += (r == c && 4 * I + r < DM) ? c0 : 0.0;
        out[I][J] = v;
      }
  };
  auto zero = [&](RMat& M) {
#pragma unroll
    for (int I = 0; I < NB; ++I)
#pragma unroll
      for (int J = 0; J < NB; ++J) M[I][J] = 0.0;
  };
  auto load_real = [&](RMat& M, const double* src, bool on, bool TR) {  // row-major Dm x Dm (TR: its transpose) -> tiles
#pragma unroll
    for (int I = 0; I < NB; ++I)
#pragma unroll
      for (int J = 0; J < NB; ++J) {
        const int row = 4 * I + r, col = 4 * J + c;
        M[I][J] = (on && row < DM && col < DM) ? (TR ? src[col * DM + row] : src[row * DM + col]) : 0.0;
      }
  };
  auto set_identity = [&](RMat& M) {
#pragma unroll
    for (int I = 0; I < NB; ++I)
#pragma unroll
      for (int J = 0; J < NB; ++J) M[I][J] = (I == J && r == c && 4 * I + r < DM) ? 1.0 : 0.0;
  };

  // ---- backward ----
  // left adjoint behind the segment with the prefix in front of the segment folded in, A~ = suf pre^T (then M_n = A~_n Q_n^T
  // with the LOCAL prefix Q_n the forward kernel stored, and A~_{n-1} = E_n^T A~_n like the plain adjoint), without the trace
  // shifts of the slices behind it: they accumulate in ams
  RMat Aa;
  {
    RMat SF, PT;
    load_real(SF, A.suf + cc * M2, valid, false);
    load_real(PT, A.pre + cc * M2, valid, true);
    to_image(img0, SF);
    zero(Aa);
    mm(img0, PT, Aa);
  }
  double ams = 0.0;
  for (int t = A.Lmax - 1; t >= 0; --t) {
    const bool act = valid && t < len;
    const double sc = act ? scale : 0.0;
    RMat X, dX;
    {
      RMat PH, Mn;
      load_real(PH, A.dus + ((long)sample * A.N + n0 + (act ? t : 0)) * M2, act, true);
      to_image(img0, Aa);
      zero(Mn);
      mm(img0, PH, Mn);  // M_n = A~ Q_n^T
#pragma unroll
      for (int I = 0; I < NB; ++I)
#pragma unroll
        for (int J = 0; J < NB; ++J) dX[I][J] = scale * Mn[I][J];
    }
    double mu = act ? tabt[TILES + 0] : 0.0;
#pragma unroll
    for (int I = 0; I < NB; ++I)
#pragma unroll
      for (int J = 0; J < NB; ++J) X[I][J] = sc * tabt[(I * NB + J) * 16 + idx16];
    for (int k = 0; k < K; ++k) {
      const double c0 = sg[(b * K + k) * A.Lmax + t];  // zero for inactive slices
      const double ck = sc * c0;
      const double* tk = tabt + (k + 1) * TABD;
      mu = fma(c0, tk[TILES + 0], mu);
#pragma unroll
      for (int I = 0; I < NB; ++I)
#pragma unroll
        for (int J = 0; J < NB; ++J) X[I][J] = fma(ck, tk[(I * NB + J) * 16 + idx16], X[I][J]);
    }
    // pair evaluation of T18: value and derivative in the direction dX (as smalld_grad_general_kernel, real)
    to_image(img0, X);
    to_image(img1, dX);
    RMat A2, dA2, A3, dA3, A6, dA6;
    zero(A2), zero(dA2), zero(A3), zero(dA3), zero(A6), zero(dA6);
    mm(img0, X, A2);
    mm(img0, dX, dA2);
    mm(img1, X, dA2);
    mm(img0, A2, A3);
    mm(img0, dA2, dA3);
    mm(img1, A2, dA3);
    to_image(img0, A3);
    to_image(img1, dA3);
    mm(img0, A3, A6);
    mm(img0, dA3, dA6);
    mm(img1, A3, dA6);
    RMat A9, dA9;
    {
      RMat B1, dB1;
      comb(B1, 0.0, tc[C3P_I_A11], tc[C3P_I_A21], tc[C3P_I_A31], 0.0, X, A2, A3, A6);
      comb(dB1, 0.0, tc[C3P_I_A11], tc[C3P_I_A21], tc[C3P_I_A31], 0.0, dX, dA2, dA3, dA6);
      to_image(img0, B1);
      to_image(img1, dB1);
    }
    {
      RMat B5, dB5;
      comb(B5, 0.0, 0.0, tc[C3P_I_B24], tc[C3P_I_B34], tc[C3P_I_B64], X, A2, A3, A6);
      comb(dB5, 0.0, 0.0, tc[C3P_I_B24], tc[C3P_I_B34], tc[C3P_I_B64], dX, dA2, dA3, dA6);
      comb(A9, tc[C3P_I_B03], tc[C3P_I_B13], tc[C3P_I_B23], tc[C3P_I_B33], tc[C3P_I_B63], X, A2, A3, A6);
      comb(dA9, 0.0, tc[C3P_I_B13], tc[C3P_I_B23], tc[C3P_I_B33], tc[C3P_I_B63], dX, dA2, dA3, dA6);
      mm(img0, B5, A9);
      mm(img0, dB5, dA9);
      mm(img1, B5, dA9);
    }
    RMat T, dT;
    {
      RMat L, dL;
      comb(L, tc[C3P_I_B02], tc[C3P_I_B12], tc[C3P_I_B22], tc[C3P_I_B32], tc[C3P_I_B62], X, A2, A3, A6);
      comb(dL, 0.0, tc[C3P_I_B12], tc[C3P_I_B22], tc[C3P_I_B32], tc[C3P_I_B62], dX, dA2, dA3, dA6);
#pragma unroll
      for (int I = 0; I < NB; ++I)
#pragma unroll
        for (int J = 0; J < NB; ++J) {
          L[I][J] += A9[I][J];
          dL[I][J] += dA9[I][J];
        }
      to_image(img0, L);
      to_image(img1, dL);
    }
    comb(T, 0.0, tc[C3P_I_B11], tc[C3P_I_B21], tc[C3P_I_B31], tc[C3P_I_B61], X, A2, A3, A6);
    comb(dT, 0.0, tc[C3P_I_B11], tc[C3P_I_B21], tc[C3P_I_B31], tc[C3P_I_B61], dX, dA2, dA3, dA6);
    mm(img0, A9, T);
    mm(img0, dA9, dT);
    mm(img1, A9, dT);
    for (int it = 0; it < ps; ++it) {
      to_image(img0, T);
      to_image(img1, dT);
      RMat T2, dT2;
      zero(T2), zero(dT2);
      mm(img0, T, T2);
      mm(img0, dT, dT2);
      mm(img1, T, dT2);
#pragma unroll
      for (int I = 0; I < NB; ++I)
#pragma unroll
        for (int J = 0; J < NB; ++J) {
          T[I][J] = T2[I][J];
          dT[I][J] = dT2[I][J];
        }
    }
    // grad[k] = e^{ams + mu} <dT, G'_k> (the tables hold G'_k - mu_k 1: the trace part separately)
    {
      const double ef = exp(ams + mu);
      double trz = 0.0;
#pragma unroll
      for (int I = 0; I < NB; ++I) trz += (r == c && 4 * I + r < DM) ? dT[I][I] : 0.0;
      for (int k = 0; k < K; ++k) {
        const double* tk = tab + (k + 1) * TABD;
        double part = tk[TILES + 0] * trz;
#pragma unroll
        for (int I = 0; I < NB; ++I)
#pragma unroll
          for (int J = 0; J < NB; ++J) part = fma(dT[I][J], tk[(I * NB + J) * 16 + idx16], part);
        part *= ef;
        part += __shfl_xor(part, 1);
        part += __shfl_xor(part, 2);
        part += __shfl_xor(part, 16);
        part += __shfl_xor(part, 32);
        if (act && idx16 == 0) A.grad[((long)sample * K + k) * A.N + n0 + t] = part;
      }
    }
    // A <- dU_n^T A = e^{mu} T A
    {
      RMat V;
      to_image(img0, T);
      zero(V);
      mm(img0, Aa, V);
#pragma unroll
      for (int I = 0; I < NB; ++I)
#pragma unroll
        for (int J = 0; J < NB; ++J) Aa[I][J] = V[I][J];
      ams += mu;
    }
  }
}

template <int DM>
hipError_t launch_grad_t(const SmallRGradArgs& A, hipStream_t st) {
  using G = RG<DM>;
  const long nchains = (long)A.B * A.S;
  const size_t lds = (size_t)(2 * (1 + A.K) * G::TABD + 8 * G::RIMG + 4 * A.K * A.Lmax) * sizeof(double);
  if (lds > 60 * 1024) return hipErrorInvalidValue;
  C3P_LAUNCH(smallr_grad_kernel<DM>, dim3((unsigned)((nchains + 3) / 4)), dim3(64), lds, st, A);
  return hipGetLastError();
}

template <int DM>
hipError_t launch_chain_t(const SmallRArgs& A, hipStream_t st) {
  using G = RG<DM>;
  const long nchains = (long)A.B * A.S;
  const size_t lds = (size_t)((1 + A.K) * G::TABD + 4 * G::RIMG + 4 * A.K * A.Lmax) * sizeof(double);
  if (lds > 60 * 1024) return hipErrorInvalidValue;
  C3P_LAUNCH(smallr_chain_kernel<DM>, dim3((unsigned)((nchains + 3) / 4)), dim3(64), lds, st, A);
  return hipGetLastError();
}

}  // namespace

bool c3p_smallr_supported(int Dh, int Dm, int K) { return (Dh == 2 || Dh == 3 || Dh == 4) && Dm == Dh * Dh && K >= 0 && K <= 8; }

size_t c3p_smallr_table_doubles(int Dm, int K) {
  const int NB = (Dm + 3) / 4;
  return (size_t)(1 + K) * (NB * NB * 16 + 4);
}

static int sr_rimg(int NB) { return NB == 4 ? 296 : 4 * NB * (4 * NB + 1); }  // = RG<Dm>::RIMG

size_t c3p_smallr_lds_bytes(int Dm, int K, int Lmax) {
  const int NB = (Dm + 3) / 4;
  return (size_t)((1 + K) * (NB * NB * 16 + 4) + 4 * sr_rimg(NB) + 4 * K * Lmax) * sizeof(double);
}

hipError_t c3p_launch_smallr_prep(const RegdPrepArgs& P, int nsamp, double* tables, int* tabflag, hipStream_t st, int transpose) {
  C3P_LAUNCH(smallr_prep_kernel, dim3((unsigned)(nsamp * (1 + P.K))), dim3(64), 0, st, P, tables, (double*)nullptr, tabflag, transpose);
  return hipGetLastError();
}

hipError_t c3p_launch_smallr_prep_pair(const RegdPrepArgs& P, int nsamp, double* tables, double* tables_t, int* tabflag, hipStream_t st) {
  C3P_LAUNCH(smallr_prep_kernel, dim3((unsigned)(nsamp * (1 + P.K)), 2), dim3(64), 0, st, P, tables, tables_t, tabflag, 2);
  return hipGetLastError();
}

hipError_t c3p_launch_smallr_chain(const SmallRArgs& A, hipStream_t st) {
  switch (A.Dm) {
    case 4: return launch_chain_t<4>(A, st);
    case 9: return launch_chain_t<9>(A, st);
    case 16: return launch_chain_t<16>(A, st);
    default: return hipErrorInvalidValue;
  }
}

size_t c3p_smallr_grad_lds_bytes(int Dm, int K, int Lmax) {
  const int NB = (Dm + 3) / 4;
  return (size_t)(2 * (1 + K) * (NB * NB * 16 + 4) + 8 * sr_rimg(NB) + 4 * K * Lmax) * sizeof(double);
}

hipError_t c3p_launch_smallr_grad(const SmallRGradArgs& A, hipStream_t st) {
  switch (A.Dm) {
    case 4: return launch_grad_t<4>(A, st);
    case 9: return launch_grad_t<9>(A, st);
    case 16: return launch_grad_t<16>(A, st);
    default: return hipErrorInvalidValue;
  }
}

hipError_t c3p_launch_smallr_scan(const double* seg, const double* ubar, int B, int S, int Dm, double* pre, double* suf, hipStream_t st) {
  if (S >= 16 && !c3p_opt_on(C3P_OPT_no_fuse)) {  // (no_fuse: the sequential scan, A/B)
    if (Dm == 4)
      C3P_LAUNCH((smallr_scan_blocked_kernel<4, 8>), dim3((unsigned)B, 2), dim3(512), 0, st, seg, ubar, S, pre, suf);
    else if (Dm == 9)
      C3P_LAUNCH((smallr_scan_blocked_kernel<9, 8>), dim3((unsigned)B, 2), dim3(512), 0, st, seg, ubar, S, pre, suf);
    else if (Dm == 16)
      C3P_LAUNCH((smallr_scan_blocked_kernel<16, 4>), dim3((unsigned)B, 2), dim3(256), 0, st, seg, ubar, S, pre, suf);
    else
      return hipErrorInvalidValue;
    return hipGetLastError();
  }
  if (Dm == 4)
    C3P_LAUNCH(smallr_scan_kernel<4>, dim3((unsigned)B, 2), dim3(64), 0, st, seg, ubar, S, pre, suf);
  else if (Dm == 9)
    C3P_LAUNCH(smallr_scan_kernel<9>, dim3((unsigned)B, 2), dim3(64), 0, st, seg, ubar, S, pre, suf);
  else if (Dm == 16)
    C3P_LAUNCH(smallr_scan_kernel<16>, dim3((unsigned)B, 2), dim3(64), 0, st, seg, ubar, S, pre, suf);
  else
    return hipErrorInvalidValue;
  return hipGetLastError();
}
