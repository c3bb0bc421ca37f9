// Probe: does v_mfma_f64_4x4x4_4b_f64 honour cbsz / abid (A-block broadcast) on gfx950?
// For every (cbsz, abid) prints, per output block b, which (A block, B block) pair reproduces the result.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
template <int CB, int AB>
__global__ void k(const double* a, const double* b, double* o) {
  o[threadIdx.x] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[threadIdx.x], b[threadIdx.x], 0.0, CB, AB, 0);
}
static double ha[64], hb[64], ho[64];
template <int CB, int AB>
void run(double* da, double* db, double* dout) {
  hipLaunchKernelGGL((k<CB, AB>), dim3(1), dim3(64), 0, 0, da, db, dout);
  hipMemcpy(ho, dout, sizeof(ho), hipMemcpyDeviceToHost);
  printf("cbsz=%d abid=%d:", CB, AB);
  // lane = 16 r + 4 blk + c ; A lane holds A_blk[c][k=r], B lane holds B_blk[k=r][c], out lane: C_blk[r][c]
  for (int ob = 0; ob < 4; ++ob) {
    int fa = -1, fb = -1;
    for (int ja = 0; ja < 4; ++ja)
      for (int jb = 0; jb < 4; ++jb) {
        bool ok = true;
        for (int r = 0; r < 4 && ok; ++r)
          for (int c = 0; c < 4 && ok; ++c) {
            double s = 0;
            for (int kk = 0; kk < 4; ++kk) s += ha[16 * kk + 4 * ja + r] * hb[16 * kk + 4 * jb + c];
            if (fabs(s - ho[16 * r + 4 * ob + c]) > 1e-12) ok = false;
          }
        if (ok) fa = ja, fb = jb;
      }
    printf("  out%d=A%d*B%d", ob, fa, fb);
  }
  printf("\n");
  if (CB == 2) {
    // hypothesis H1: A lane l replaced by A lane (16*AB + l%16); H2: by lane (l/16*16 + 4*AB + l%4) [block bcast]; H3: lane 16*(l/16)+... print which
    for (int hyp = 0; hyp < 3; ++hyp) {
      double am[64];
      for (int l = 0; l < 64; ++l) {
        int src = hyp == 0 ? 16 * AB + l % 16 : hyp == 1 ? (l / 16) * 16 + 4 * AB + l % 4 : (l & ~3) + AB;
        am[l] = ha[src];
      }
      bool ok = true;
      for (int ob = 0; ob < 4; ++ob) for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) {
        double s2 = 0;
        for (int kk = 0; kk < 4; ++kk) s2 += am[16 * kk + 4 * ob + r] * hb[16 * kk + 4 * ob + c];
        if (fabs(s2 - ho[16 * r + 4 * ob + c]) > 1e-12) ok = false;
      }
      printf("   hyp%d %s\n", hyp, ok ? "MATCH" : "no");
    }
    printf("   raw:"); for (int l = 0; l < 64; ++l) printf(" %.6f", ho[l]); printf("\n");
  }
}
int main() {
  for (int i = 0; i < 64; ++i) ha[i] = 0.37 + 0.011 * i + 0.003 * (i % 7), hb[i] = -0.21 + 0.017 * i - 0.002 * (i % 5);
  double *da, *db, *dout;
  hipMalloc(&da, 512); hipMalloc(&db, 512); hipMalloc(&dout, 512);
  hipMemcpy(da, ha, 512, hipMemcpyHostToDevice);
  hipMemcpy(db, hb, 512, hipMemcpyHostToDevice);
  run<0, 0>(da, db, dout);
  run<1, 0>(da, db, dout); run<1, 1>(da, db, dout); run<1, 2>(da, db, dout); run<1, 3>(da, db, dout);
  run<2, 0>(da, db, dout); run<2, 1>(da, db, dout); run<2, 2>(da, db, dout); run<2, 3>(da, db, dout);
  return 0;
}
