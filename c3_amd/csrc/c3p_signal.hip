// Control-signal synthesis for the standard drive line
//   LO + AWG -> DigitalToAnalog -> Mixer -> VoltsToHertz
// (reference: c3/signal/gates.py:341-370 get_awg_signal, c3/signal/pulse.py:88-180 envelopes/mask/DRAG,
//  c3/generator/devices.py:72-122 create_ts, :306-351 DAC, :914-939 Mixer, :1073-1130 LO, :203-221 V->Hz).
// A batch is described by B x K x E envelope-parameter rows instead of B x K x N samples; the samples
// are produced in HBM where the propagator kernels read them.  Two launches: the AWG-resolution I/Q
// (a few hundred samples per line), then upsampling + mixing, one thread per output sample (HBM-bound,
// 8 B written per sample, nothing re-read from HBM: the I/Q rows stay in L2).
#include <hip/hip_runtime.h>

#include "c3p_common.h"
#include "c3p_signal.h"

namespace {

// numpy.linspace(start, stop, num)[j]: j * step + start with separate roundings, last sample = stop
__device__ __forceinline__ double linspace_at(double start, double stop, int num, int j) {
  if (num <= 1) return start;
  if (j == num - 1) return stop;
  const double step = __ddiv_rn(__dsub_rn(stop, start), (double)(num - 1));
  return __dadd_rn(__dmul_rn((double)j, step), start);
}

__device__ __forceinline__ double sigmoid(double x) { return 1.0 / (1.0 + exp(-x)); }

struct EnvP {
  double amp, xy, fo, delta, t_final, sigma, t_up, t_down, risefall, delay;
  int use_t_before, drag;
};

__device__ __forceinline__ EnvP load_env(const double* p) {
  EnvP e;
  e.amp = p[C3P_ENV_AMP];
  e.xy = p[C3P_ENV_XY_ANGLE];
  e.fo = p[C3P_ENV_FREQ_OFFSET];
  e.delta = p[C3P_ENV_DELTA];
  e.t_final = p[C3P_ENV_T_FINAL];
  e.sigma = p[C3P_ENV_SIGMA];
  e.t_up = p[C3P_ENV_T_UP];
  e.t_down = p[C3P_ENV_T_DOWN];
  e.risefall = p[C3P_ENV_RISEFALL];
  e.delay = p[C3P_ENV_DELAY];
  const int fl = (int)p[C3P_ENV_FLAGS];
  e.use_t_before = fl & C3P_ENVF_T_BEFORE;
  e.drag = fl & C3P_ENVF_DRAG;
  return e;
}

__device__ double shape_val(int shape, double t, const EnvP& p) {
  switch (shape) {
    case C3P_ENV_RECT:
      return 1.0;
    case C3P_ENV_GAUSSIAN_NONORM: {
      const double u = t - p.t_final / 2;
      return exp(-(u * u) / (2 * p.sigma * p.sigma));
    }
    case C3P_ENV_FLATTOP:
    case C3P_ENV_FLATTOP_RISEFALL: {
      const double rf = p.risefall;
      const double tu = shape == C3P_ENV_FLATTOP ? p.t_up : rf;
      const double td = shape == C3P_ENV_FLATTOP ? p.t_down : p.t_final - rf;
      return (1 + erf((t - tu) / rf)) / 2 * (1 + erf((-t + td) / rf)) / 2;
    }
    case C3P_ENV_COSINE:
      return 0.5 * (1 - cos(2 * M_PI * t / p.t_final));
    case C3P_ENV_GAUSSIAN_SIGMA:
    case C3P_ENV_GAUSSIAN: {
      const double T = p.t_final;
      const double sg = shape == C3P_ENV_GAUSSIAN_SIGMA ? p.sigma : T / 6;
      const double u = t - T / 2;
      const double offset = exp(-(T * T) / (8 * sg * sg));
      const double norm = sqrt(2 * M_PI * sg * sg) * erf(T / (sqrt(8.0) * sg)) - T * offset;
      return (exp(-(u * u) / (2 * sg * sg)) - offset) / norm;
    }
    case C3P_ENV_TRAPEZOID: {
      const double w = p.risefall * 2.5;
      double env = 1.0;
      if (t <= w) env = t / w;
      if (t >= p.t_final - w) env = (p.t_final - t) / w;
      return env;
    }
    default:
      return 0.0;
  }
}

__device__ double shape_der(int shape, double t, const EnvP& p) {
  switch (shape) {
    case C3P_ENV_GAUSSIAN_NONORM:
      return -(t - p.t_final / 2) / (p.sigma * p.sigma) * shape_val(shape, t, p);
    case C3P_ENV_FLATTOP:
    case C3P_ENV_FLATTOP_RISEFALL: {
      const double rf = p.risefall;
      const double tu = shape == C3P_ENV_FLATTOP ? p.t_up : rf;
      const double td = shape == C3P_ENV_FLATTOP ? p.t_down : p.t_final - rf;
      const double u = (t - tu) / rf, d = (-t + td) / rf;
      const double c = 2 / sqrt(M_PI) / rf;
      return (c * exp(-u * u) * (1 + erf(d)) - (1 + erf(u)) * c * exp(-d * d)) / 4;
    }
    case C3P_ENV_COSINE: {
      const double w = 2 * M_PI / p.t_final;
      return 0.5 * w * sin(w * t);
    }
    case C3P_ENV_GAUSSIAN_SIGMA:
    case C3P_ENV_GAUSSIAN: {
      const double T = p.t_final;
      const double sg = shape == C3P_ENV_GAUSSIAN_SIGMA ? p.sigma : T / 6;
      const double u = t - T / 2;
      const double offset = exp(-(T * T) / (8 * sg * sg));
      const double norm = sqrt(2 * M_PI * sg * sg) * erf(T / (sqrt(8.0) * sg)) - T * offset;
      return -u / (sg * sg) * exp(-(u * u) / (2 * sg * sg)) / norm;
    }
    case C3P_ENV_TRAPEZOID: {
      const double w = p.risefall * 2.5;
      double d = 0.0;
      if (t <= w) d = 1.0 / w;
      if (t >= p.t_final - w) d = -1.0 / w;
      return d;
    }
    default:
      return 0.0;
  }
}

__device__ __forceinline__ double mask_val(double ts_off, double dt, double tf_) {
  return sigmoid((ts_off / dt + 0.001) * 1e6) * sigmoid((0.999 * tf_ - ts_off) / dt * 1e6);
}

struct EnvEval {
  double er;       // real envelope (mask * (shape - offset))
  double ei_unit;  // imaginary DRAG part per unit delta: -dt * d env/dt (0 unless C3P_ENVF_DRAG)
  double ts_off;   // time relative to the component's start
};

// envelope of component `p` at AWG sample j (pulse.py:88-180)
__device__ EnvEval env_eval(const SynthArgs& A, int shape, const EnvP& p, int j, double a0, double a1) {
  const double t = linspace_at(a0, a1, A.Na, j);
  const double t_first = linspace_at(a0, a1, A.Na, 0), t_second = linspace_at(a0, a1, A.Na, A.Na > 1 ? 1 : 0);
  const double t0 = A.t_start + p.delay;
  const double ts_off = t - t0, off0 = t_first - t0, off1 = t_second - t0;
  const double dt = off1 - off0;
  const double tf_ = p.t_final;  // window = t_final of the component (gates.py:293-297)
  const double m = mask_val(ts_off, dt, tf_);
  const double t_before = 2 * off0 - off1;
  const double offset = p.use_t_before ? shape_val(shape, t_before, p) : 0.0;
  EnvEval r;
  r.ts_off = ts_off;
  r.er = m * (shape_val(shape, ts_off, p) - offset);
  r.ei_unit = 0.0;
  if (p.drag) {
    double denv = m * shape_der(shape, ts_off, p);
    if (p.use_t_before && j < 2) {
      double msum = 0.0;
      for (int i = 0; i < A.Na; ++i) msum += mask_val(linspace_at(a0, a1, A.Na, i) - t0, dt, tf_);
      const double doff = shape_der(shape, t_before, p) * msum;
      denv += (j == 0) ? -2 * doff : doff;
    }
    r.ei_unit = -denv * dt;
  }
  return r;
}

// one thread per AWG sample (b, k, j)
__global__ void awg_iq_kernel(SynthArgs A) {
  const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long total = (long)A.B * A.K * A.Na;
  if (gid >= total) return;
  const int j = (int)(gid % A.Na);
  const long bk = gid / A.Na;
  const int k = (int)(bk % A.K);
  const double dta = 1.0 / A.awg_res;
  const double a0 = A.t_start + dta / 2, a1 = A.t_end - dta / 2;
  double re = 0.0, im = 0.0;
  for (int e = 0; e < A.E; ++e) {
    const int shape = A.shape[k * A.E + e];
    if (shape < 0) continue;
    const EnvP p = load_env(A.env + (bk * A.E + e) * C3P_ENV_NPAR);
    const EnvEval v = env_eval(A, shape, p, j, a0, a1);
    const double er = v.er, ei = v.ei_unit * p.delta;
    double sn, cs;
    sincos(p.xy - p.fo * v.ts_off, &sn, &cs);
    re += p.amp * (er * cs - ei * sn);
    im += p.amp * (er * sn + ei * cs);
  }
  double* iq = A.iq + bk * 2 * A.Na;
  iq[j] = re;
  iq[A.Na + j] = im;
}

__device__ __forceinline__ int dac_src(int n, int Na, int N) {
  int src = (int)floor(((double)n + 0.5) * ((double)Na / (double)N));
  return src < Na - 1 ? src : Na - 1;
}

// ---- vector-Jacobian product: d loss/d signals -> d loss/d (amp, xy_angle, freq_offset, delta, carrier) ----
// step 1, one WAVEFRONT per AWG sample: its lanes take the simulation samples it feeds (sim_res / awg_res of them: 50 at the
// reference's resolutions -- one thread per AWG sample walked them one double-precision sincos after the other, 23 us for 1 280
// threads), fold them back onto I and Q with a fixed-order butterfly sum
__device__ __forceinline__ double wave_sum_b(double v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__global__ void __launch_bounds__(64) mix_bwd_kernel(SynthArgs A, const double* gsig, double* giq, double* gcar_part) {
  const long gid = blockIdx.x;  // (b, k, j)
  const int lane = threadIdx.x;
  const int j = (int)(gid % A.Na);
  const long bk = gid / A.Na;
  const double dts = 1.0 / A.sim_res;
  const double s0 = A.t_start + dts / 2, s1 = A.t_end - dts / 2;
  int n = (int)ceil((double)j * ((double)A.N / (double)A.Na) - 0.5);
  n = n < 0 ? 0 : (n > A.N ? A.N : n);
  while (n > 0 && dac_src(n - 1, A.Na, A.N) >= j) --n;
  while (n < A.N && dac_src(n, A.Na, A.N) < j) ++n;
  const double w = A.carrier[bk * 2 + 0], v2hz = A.carrier[bk * 2 + 1];
  const double* iq = A.iq + bk * 2 * A.Na;
  const double I = iq[j], Q = iq[A.Na + j];
  const double* g = gsig + bk * A.N;
  double gI = 0.0, gQ = 0.0, gw = 0.0, gv = 0.0;
  // the samples fed by j are contiguous from n on; every wavefront pass takes 64 of them
  for (int base = n; base < A.N && dac_src(base, A.Na, A.N) == j; base += 64) {
    const int m = base + lane;
    if (m < A.N && dac_src(m, A.Na, A.N) == j) {
      const double t = linspace_at(s0, s1, A.N, m);
      double sn, cs;
      sincos(w * t, &sn, &cs);
      const double gn = g[m];
      gI = fma(gn * v2hz, cs, gI);
      gQ = fma(gn * v2hz, sn, gQ);
      gw = fma(gn * v2hz * t, cs * Q - sn * I, gw);
      gv = fma(gn, cs * I + sn * Q, gv);
    }
  }
  gI = wave_sum_b(gI);
  gQ = wave_sum_b(gQ);
  gw = wave_sum_b(gw);
  gv = wave_sum_b(gv);
  if (lane == 0) {
    giq[bk * 2 * A.Na + j] = gI;
    giq[bk * 2 * A.Na + A.Na + j] = gQ;
    gcar_part[(bk * A.Na + j) * 2 + 0] = gw;
    gcar_part[(bk * A.Na + j) * 2 + 1] = gv;
  }
}

__device__ __forceinline__ double wave_sum(double v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// step 2, one wavefront per drive line (b, k): reduce over the AWG samples
__global__ void __launch_bounds__(64) awg_bwd_kernel(SynthArgs A, const double* giq, const double* gcar_part, double* genv,
                                                     double* gcar) {
  const long bk = blockIdx.x;
  const int k = (int)(bk % A.K);
  const int lane = threadIdx.x;
  const double dta = 1.0 / A.awg_res;
  const double a0 = A.t_start + dta / 2, a1 = A.t_end - dta / 2;
  const double* gI = giq + bk * 2 * A.Na;
  const double* gQ = gI + A.Na;
  double gw = 0.0, gv = 0.0;
  for (int j = lane; j < A.Na; j += 64) {
    gw += gcar_part[(bk * A.Na + j) * 2 + 0];
    gv += gcar_part[(bk * A.Na + j) * 2 + 1];
  }
  gw = wave_sum(gw);
  gv = wave_sum(gv);
  if (lane == 0) {
    gcar[bk * 2 + 0] = gw;
    gcar[bk * 2 + 1] = gv;
  }
  for (int e = 0; e < A.E; ++e) {
    double* out = genv + (bk * A.E + e) * C3P_ENV_NPAR;
    if (lane < C3P_ENV_NPAR) out[lane] = 0.0;
    const int shape = A.shape[k * A.E + e];
    if (shape < 0) continue;
    const EnvP p = load_env(A.env + (bk * A.E + e) * C3P_ENV_NPAR);
    double g_amp = 0.0, g_xy = 0.0, g_fo = 0.0, g_delta = 0.0;
    for (int j = lane; j < A.Na; j += 64) {
      const EnvEval v = env_eval(A, shape, p, j, a0, a1);
      double sn, cs;
      sincos(p.xy - p.fo * v.ts_off, &sn, &cs);
      const double er = v.er, ei = v.ei_unit * p.delta;
      const double ur = er * cs - ei * sn, ui = er * sn + ei * cs;  // env e^{i phase}
      const double zr = p.amp * ur, zi = p.amp * ui;
      const double a = gI[j], b = gQ[j];
      g_amp += a * ur + b * ui;
      g_xy += -a * zi + b * zr;
      g_fo += v.ts_off * (a * zi - b * zr);
      // d z / d delta = amp * i ei_unit e^{i phase}
      g_delta += p.amp * v.ei_unit * (-a * sn + b * cs);
    }
    g_amp = wave_sum(g_amp);
    g_xy = wave_sum(g_xy);
    g_fo = wave_sum(g_fo);
    g_delta = wave_sum(g_delta);
    if (lane == 0) {
      out[C3P_ENV_AMP] = g_amp;
      out[C3P_ENV_XY_ANGLE] = g_xy;
      out[C3P_ENV_FREQ_OFFSET] = g_fo;
      out[C3P_ENV_DELTA] = g_delta;
    }
  }
}

// one thread per simulation sample (b, k, n): nearest-neighbour upsampling, IQ mixing, V -> Hz
__global__ void mix_kernel(SynthArgs A) {
  const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long total = (long)A.B * A.K * A.N;
  if (gid >= total) return;
  const int n = (int)(gid % A.N);
  const long bk = gid / A.N;
  const double dts = 1.0 / A.sim_res;
  const double t = linspace_at(A.t_start + dts / 2, A.t_end - dts / 2, A.N, n);
  const int src = dac_src(n, A.Na, A.N);
  const double* iq = A.iq + bk * 2 * A.Na;
  const double w = A.carrier[bk * 2 + 0], v2hz = A.carrier[bk * 2 + 1];
  double sn, cs;
  sincos(w * t, &sn, &cs);
  A.signals[gid] = (cs * iq[src] + sn * iq[A.Na + src]) * v2hz;
}

}  // namespace

hipError_t c3p_launch_synth_vjp(const SynthArgs& A, const double* gsig, double* giq, double* gcar_part, double* genv,
                                double* gcar, hipStream_t st) {
  const long ta = (long)A.B * A.K * A.Na;
  if (ta == 0) return hipSuccess;
  C3P_LAUNCH(awg_iq_kernel, dim3((unsigned)((ta + 127) / 128)), dim3(128), 0, st, A);
  C3P_LAUNCH(mix_bwd_kernel, dim3((unsigned)ta), dim3(64), 0, st, A, gsig, giq, gcar_part);
  C3P_LAUNCH(awg_bwd_kernel, dim3((unsigned)(A.B * A.K)), dim3(64), 0, st, A, (const double*)giq,
                     (const double*)gcar_part, genv, gcar);
  return hipGetLastError();
}

hipError_t c3p_launch_synth(const SynthArgs& A, hipStream_t st) {
  const long ta = (long)A.B * A.K * A.Na, ts = (long)A.B * A.K * A.N;
  if (ta == 0 || ts == 0) return hipSuccess;
  C3P_LAUNCH(awg_iq_kernel, dim3((unsigned)((ta + 127) / 128)), dim3(128), 0, st, A);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  C3P_LAUNCH(mix_kernel, dim3((unsigned)((ts + 255) / 256)), dim3(256), 0, st, A);
  return hipGetLastError();
}
