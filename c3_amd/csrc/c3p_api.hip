// C ABI of libc3prop.so (see include/c3prop.h for the contract and the reference
// functions each entry point stands in for).
#include <algorithm>
#include <memory>
#include <mutex>
#include <string>
#include <vector>
#include <cstdio>
#include <cstdarg>
#include <cstring>
#include <cstdlib>
#include <cctype>

#include "../../include/c3prop.h"
#include "c3p_kernels.h"
#include "c3p_ode.h"
#include "c3p_smalld.h"
#include "c3p_smallr.h"
#include "c3p_midd.h"
#include <cxxabi.h>
#include <cstring>

#include "c3p_bigd.h"
#include "c3p_regd.h"
#include <atomic>
#include "c3p_tiled.h"
#include "c3p_signal.h"
#include "c3p_grad.h"

namespace {

thread_local std::string g_err;
thread_local int g_last_kernel = C3P_KERNEL_NONE;
// launch log of the current API call (c3p_note_launch / c3p_last_kernel_detail)
struct LaunchNote {
  const void* fn;
  const char* file;
  int count;
};
thread_local LaunchNote g_notes[48];
thread_local int g_nnotes = 0;
thread_local hipStream_t g_call_stream = nullptr;  // stream of the call in flight (capture check on workspace growth)
thread_local bool g_dry = false;  // c3p_reserve: run the planning and size the workspace, launch nothing

// ---- the options table (c3p_kernels.h) ----
const char* const g_opt_names[C3P_OPT_COUNT] = {
#define C3P_OPT_NAME(n) #n,
    C3P_OPTION_LIST(C3P_OPT_NAME)
#undef C3P_OPT_NAME
};
std::atomic<long> g_opt_val[C3P_OPT_COUNT];
std::once_flag g_opt_once;

long parse_opt_value(const char* v) {
  if (!v || !*v) return 1;  // NAME= : a switch that is set
  char* end = nullptr;
  const long x = strtol(v, &end, 10);
  if (end != v && *end == 0) return x;
  if (!strcmp(v, "all")) return 2;
  if (!strcmp(v, "unset") || !strcmp(v, "default")) return -1;
  return 1;
}

void opt_init() {
  for (int i = 0; i < C3P_OPT_COUNT; ++i) {
    std::string env = "C3P_";
    for (const char* c = g_opt_names[i]; *c; ++c) env += (char)toupper((unsigned char)*c);
    const char* v = getenv(env.c_str());  // read ONCE, here
    g_opt_val[i].store(v ? parse_opt_value(v) : -1, std::memory_order_relaxed);
  }
}

int fail(const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return -1;
}

#define HIP_TRY(expr)                                                                  \
  do {                                                                                 \
    hipError_t e__ = (expr);                                                           \
    if (e__ != hipSuccess) return fail("%s failed: %s", #expr, hipGetErrorString(e__)); \
  } while (0)

#define LAUNCH_TRY(expr) \
  do {                   \
    if (!g_dry) HIP_TRY(expr); \
  } while (0)

// Per-device workspace slots, grown lazily, freed by c3p_shutdown().
enum Slot { SL_SEG_A = 0, SL_SEG_B, SL_SCRATCH, SL_CLP, SL_TABLES, SL_COUNTERS, SL_COUNTERS2, SL_IN0, SL_IN1, SL_IN2, SL_IN3, SL_IN4, SL_IN5, SL_OUT0, SL_OUT1, SL_OUT2, SL_OUT3, SL_SEG_F, SL_COUNT };

struct DeviceWs {
  std::mutex mu;  // one lock per device: calls on different GPUs of one process do not serialise
  void* ptr[SL_COUNT] = {};
  size_t cap[SL_COUNT] = {};
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  bool ev_valid = false;
  std::atomic<int> profiling{0};  // per device, like the event pair it arms (c3p_set_profiling acts on the current device)
  std::atomic<long> generation{0};  // bumped whenever a slot is (re)allocated: steady state = constant
  // The workspace (segment products, tables, arrival counters, arena) is ONE set per device and calls are asynchronous:
  // a call on another stream than the previous one first makes its stream wait for the event recorded at the end of that
  // call, so two streams never run kernels on the shared workspace at the same time (calls on one stream are ordered
  // by the stream itself).
  hipEvent_t done = nullptr;
  hipStream_t last_stream = nullptr;
  bool has_last = false;
  // a BLOCKING stream of the library's own: calls made on the legacy default stream (what torch hands over by default)
  // cannot capture graphs; work enqueued here is ordered against the default stream by the legacy-stream rule
  hipStream_t aux = nullptr;
};

std::mutex g_mu;  // guards the table of per-device workspaces only
std::vector<std::unique_ptr<DeviceWs>> g_ws;

DeviceWs* ws_for_current_device() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  std::lock_guard<std::mutex> lk(g_mu);
  if ((int)g_ws.size() <= dev) g_ws.resize(dev + 1);
  if (!g_ws[dev]) g_ws[dev].reset(new DeviceWs());
  return g_ws[dev].get();
}

// RAII: the current device's workspace, locked for this call and ordered after the previous call's stream
struct WsLock {
  DeviceWs* w = nullptr;
  hipStream_t st = nullptr;
  bool locked = false;
  bool ok = true;
  WsLock(hipStream_t stream, bool order = true) : st(stream) {
    w = ws_for_current_device();
    if (!w) return;
    w->mu.lock();
    locked = true;
    g_call_stream = stream;
    if (order) g_nnotes = 0;  // a compute call starts a new launch log (c3p_last_kernel_ms looks with order = false)
    // a capturing stream takes no dependency on work outside its graph: the caller orders other streams before the capture
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (stream && hipStreamIsCapturing(stream, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone) order = false;
    (void)hipGetLastError();
    // The workspace is shared by the calls of a device: a call on ANOTHER stream than the previous one first waits for
    // everything enqueued on that stream so far.  The event is recorded lazily, here, on the previous stream (enqueue order
    // is fixed by the mutex), so the common one-stream caller pays no event per call (a marker packet between every two
    // kernels: ~3 us of a 140 us cfg2 batch).  A previous stream that has been destroyed meanwhile: device-wide sync.
    if (order && w->has_last && w->last_stream != st) {
      if (!w->done) (void)hipEventCreateWithFlags(&w->done, hipEventDisableTiming);
      if (!w->done || hipEventRecord(w->done, w->last_stream) != hipSuccess || hipStreamWaitEvent(st, w->done, 0) != hipSuccess) {
        (void)hipGetLastError();
        if (hipDeviceSynchronize() != hipSuccess) ok = false;
        w->has_last = false;  // everything enqueued so far is done: forget the (possibly destroyed) stream
        w->last_stream = nullptr;
      }
    }
    ordered = order;
  }
  ~WsLock() {
    if (!locked) return;
    if (ordered) {
      w->last_stream = st;
      w->has_last = true;
    }
    w->mu.unlock();
  }
  bool ordered = true;
};

int ws_get(DeviceWs* w, Slot s, size_t bytes, void** out) {
  if (bytes == 0) bytes = 16;
  if (w->cap[s] < bytes) {
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (g_call_stream && hipStreamIsCapturing(g_call_stream, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone)
      return fail("workspace growth during stream capture: run the call once outside the capture, or c3p_reserve() its shape first");
    (void)hipGetLastError();
    ++w->generation;
    if (w->ptr[s]) {
      HIP_TRY(hipDeviceSynchronize());
      HIP_TRY(hipFree(w->ptr[s]));
      w->ptr[s] = nullptr;
      w->cap[s] = 0;
    }
    size_t want = bytes + bytes / 8;
    HIP_TRY(hipMalloc(&w->ptr[s], want));
    w->cap[s] = want;
    // the arrival counters of the fused combine are self-resetting: zero once, at allocation
    if (s == SL_COUNTERS) HIP_TRY(hipMemset(w->ptr[s], 0, want));
  }
  *out = w->ptr[s];
  return 0;
}

struct ChainPlan {
  int S, seg_len;
  bool global_scratch;
};

const int kMaxGenericDm = 256;

ChainPlan plan_generic(int B, int N, int Dm) {
  ChainPlan p;
  const size_t lds = c3p_generic_lds_bytes(Dm);
  p.global_scratch = lds > (size_t)(156 * 1024);
  int wg_per_cu = 2;
  if (!p.global_scratch) {
    wg_per_cu = (int)((160 * 1024) / (lds + 3072));
    if (wg_per_cu < 1) wg_per_cu = 1;
    if (wg_per_cu > 8) wg_per_cu = 8;
  }
  const long target = 256L * wg_per_cu * 2;
  long S = (target + B - 1) / B;
  const long smax = N / 8 > 1 ? N / 8 : 1;
  if (S > smax) S = smax;
  if (S < 1) S = 1;
  p.seg_len = (int)((N + S - 1) / S);
  p.S = (N + p.seg_len - 1) / p.seg_len;
  return p;
}

// Runs one chained propagation (all modes) with the generic kernel, including the
// ordered combine of segment products.  All pointers are device pointers.
// `profile` = false keeps the timing event pair (and c3p_last_kernel) on the caller's main kernel.
int run_chain_generic(DeviceWs* w, ChainArgs base, cplx* U_out, hipStream_t st, bool profile = true) {
  const int Dm = base.Dm;
  profile = profile && w->profiling && !g_dry;
  // GIVEN mode callers may hand over matrices that live in one of the two segment slots (the segment products of the
  // big-D kernels): this launch must not write its own segment products into the buffer it is reading (and ws_get
  // must not re-allocate it), so it starts on the OTHER slot.
  const bool mats_in_a = base.mats && w->ptr[SL_SEG_A] && (const void*)base.mats == w->ptr[SL_SEG_A];
  const Slot first_slot = mats_in_a ? SL_SEG_B : SL_SEG_A;
  if (Dm > kMaxGenericDm) return fail("matrix dimension %d exceeds the generic kernel limit %d", Dm, kMaxGenericDm);
  const size_t msz = (size_t)Dm * Dm * sizeof(cplx);
  ChainPlan p = plan_generic(base.B, base.N, Dm);
  base.ld = Dm | 1;
  base.S = p.S;
  base.seg_len = p.seg_len;
  const double* final_phase = base.fr_phase;
  cplx* seg = U_out;
  if (p.S > 1) {
    void* v;
    if (ws_get(w, first_slot, (size_t)base.B * p.S * msz, &v)) return -1;
    seg = (cplx*)v;
    base.fr_phase = nullptr;
  }
  base.seg_out = seg;
  if (p.global_scratch) {
    base.scratch_stride = (long)7 * base.ld * Dm;
    void* v;
    if (ws_get(w, SL_SCRATCH, (size_t)base.B * p.S * base.scratch_stride * sizeof(cplx), &v)) return -1;
    base.scratch = (cplx*)v;
  }
  g_last_kernel = p.global_scratch ? C3P_KERNEL_GENERIC_GLOBAL : C3P_KERNEL_GENERIC_LDS;
  if (profile) {
    if (!w->ev0) {
      HIP_TRY(hipEventCreate(&w->ev0));
      HIP_TRY(hipEventCreate(&w->ev1));
    }
    HIP_TRY(hipEventRecord(w->ev0, st));
  }
  LAUNCH_TRY(c3p_launch_chain_generic(base, p.global_scratch, st));
  if (profile) {
    HIP_TRY(hipEventRecord(w->ev1, st));
    w->ev_valid = true;
  }
  // ordered combine of the S segment products: groups of 8 until <= 16 remain
  int count = p.S;
  cplx* cur = seg;
  Slot next_slot = (first_slot == SL_SEG_A) ? SL_SEG_B : SL_SEG_A;
  while (count > 1) {
    ChainArgs c = {};
    c.mode = C3P_MODE_GIVEN;
    c.mats = cur;
    c.B = base.B;
    c.N = count;
    c.D = base.D;
    c.Dm = Dm;
    c.ld = base.ld;
    c.right_order = base.right_order;
    bool global = p.global_scratch;
    if (count <= 16) {
      c.S = 1;
      c.seg_len = count;
      c.seg_out = U_out;
      c.fr_phase = final_phase;
    } else {
      c.seg_len = 8;
      c.S = (count + 7) / 8;
      void* v;
      if (ws_get(w, next_slot, (size_t)base.B * c.S * msz, &v)) return -1;
      c.seg_out = (cplx*)v;
    }
    if (global) {
      c.scratch_stride = (long)7 * c.ld * Dm;
      void* v;
      // the first launch sized SL_SCRATCH for B*S workgroups >= B*c.S
      if (ws_get(w, SL_SCRATCH, (size_t)c.B * c.S * c.scratch_stride * sizeof(cplx), &v)) return -1;
      c.scratch = (cplx*)v;
    }
    LAUNCH_TRY(c3p_launch_chain_generic(c, global, st));
    cur = c.seg_out;
    count = c.S;
    next_slot = (next_slot == SL_SEG_B) ? SL_SEG_A : SL_SEG_B;
    if (c.seg_out == U_out) break;
  }
  return 0;
}

// ---------------------------------------------------------------------------
// Small-D MFMA path (Dm <= C3P_SMALLD_LIMIT): tables -> segment chains -> ordered combine
// ---------------------------------------------------------------------------
const int kSmallDLimit = 12;

// Unitary gradients at 41 <= D <= 64: the VALU sweep (c3p_grad.hip, ~D^3 B) against the tiled sweep (c3p_tiled.hip: matrix cores,
// padded to 64-multiples, so D = 48 and D = 64 cost the same there, and ~35 launches per slice whatever B is).  Measured on one
// MI355X, N = 1000 (tools/bench_grad_41_64.py, profiles/r06/grad_41_64.txt; ms VALU / tiled): D = 48: B = 64 112 / 220,
// B = 256 426 / 505; D = 64: B = 64 254 / 224, B = 256 983 / 519.  Rule: tiled when 1.66 B (D / 48)^3 ms (the VALU estimate)
// exceeds max(220, 1.97 B) ms (the tiled one).
static inline bool tiled_unitary_grad(int D, int B) {
  if (D > 64) return true;
  if (D <= 40) return false;
  if (c3p_opt_on(C3P_OPT_tiled_grad)) return true;
  const double r = (double)D / 48.0, valu = 1.66 * B * r * r * r, tiled = std::max(220.0, 1.97 * B);
  return valu > tiled;
}  // D = 11, 12 spill ~100 registers but still beat the generic kernel

int record_start(DeviceWs* w, hipStream_t st) {
  if (!w->profiling || g_dry) return 0;
  if (!w->ev0) {
    HIP_TRY(hipEventCreate(&w->ev0));
    HIP_TRY(hipEventCreate(&w->ev1));
  }
  HIP_TRY(hipEventRecord(w->ev0, st));
  return 0;
}
int record_stop(DeviceWs* w, hipStream_t st) {
  if (!w->profiling || g_dry) return 0;
  HIP_TRY(hipEventRecord(w->ev1, st));
  w->ev_valid = true;
  return 0;
}

// ordered combine of `count` matrices per sample (cur: [B,count,Dm,Dm]) into U_out
int combine_smalld(DeviceWs* w, const cplx* cur, int B, int count, int Dm, int right_order,
                   const double* fr_phase, cplx* U_out, hipStream_t st) {
  const size_t msz = (size_t)Dm * Dm * sizeof(cplx);
  Slot next_slot = SL_SEG_B;
  while (true) {
    SmallArgs c = {};
    c.mode = C3P_MODE_GIVEN;
    c.mats = cur;
    c.B = B;
    c.N = count;
    c.Dm = Dm;
    c.right_order = right_order;
    if (count <= 8) {
      c.S = 1;
      c.seg_out = U_out;
      c.fr_phase = fr_phase;
    } else {
      c.S = (count + 3) / 4;
      void* v;
      if (ws_get(w, next_slot, (size_t)B * c.S * msz, &v)) return -1;
      c.seg_out = (cplx*)v;
    }
    c.Lmax = (count + c.S - 1) / c.S;
    LAUNCH_TRY(c3p_launch_smalld_chain(c, st));
    if (c.seg_out == U_out) break;
    cur = c.seg_out;
    count = c.S;
    next_slot = (next_slot == SL_SEG_B) ? SL_SEG_A : SL_SEG_B;
  }
  return 0;
}

int pick_segments(int B, int N, int K, int Dm, bool need_mult4, long slots = 8192) {
  // two waves per SIMD (the D = 9 kernel fits 256 registers): 256 CUs x 4 SIMDs x 2 x 4 chains are resident at once.
  // B S chains run in ceil(B S / slots) rounds of N / S slices (+ the table build and plan of the prologue):
  // take the S that minimises rounds x segment length, so that the last round is not a mostly idle tail
  // (B = 300 with the old "fill the machine twice" rule ran a second round at 2 % occupancy).
  if (c3p_opt(C3P_OPT_smalld_segments) > 0) {  // tuning override
    const long S = c3p_opt(C3P_OPT_smalld_segments);
    if (S >= 1 && S <= N && (!need_mult4 || S % 4 == 0)) return (int)S;
  }
  // (the backward sweep runs one wave per SIMD: 4096 slots)
  // the segment's control amplitudes live in LDS: 4 chains x K x Lmax doubles
  const long lds_budget = 20 * 1024 - (long)(c3p_smalld_table_doubles(Dm, K) + 4 * c3p_smalld_img_doubles(Dm)) * 8;
  const long lmax_cap = K > 0 ? lds_budget / (32L * K) : (1L << 30);
  long best = -1;
  double best_cost = 1e300;
  const long smax = N < 4096 ? N : 4096;
  for (long S = 1; S <= smax; ++S) {
    // a multiple of four lets every wave fold its four segments in registers (fused combine)
    if (S > 1 && (S % 4) != 0 && S + 3 <= N) continue;
    if (need_mult4 && (S % 4) != 0) continue;
    if ((N + S - 1) / S > lmax_cap) continue;
    const long rounds = ((long)B * S + slots - 1) / slots;
    const double cost = (double)rounds * (double)((N + S - 1) / S + 8);
    if (cost < best_cost * (1.0 - 1e-9)) {
      best_cost = cost;
      best = S;
    }
  }
  if (best < 0 && need_mult4) {
    const long S = ((std::min<long>(N, 4) + 3) / 4) * 4;
    if (S <= N && (N + S - 1) / S <= lmax_cap) best = S;
  }
  return (int)best;
}

// Lindblad chains of one qubit / qutrit whose Hamiltonians the caller declares Hermitian (C3P_HERMITIAN_H): real arithmetic
// in the Hermitian basis on the small-D tile layout (c3p_smallr.hip).  Real generator tables -> segment products (turned back
// into the reference's vectorisation by the chain that formed them) -> ordered product with the frame phases (the
// supplied-matrix mode of the complex small-D chain kernel).  Returns 1 when not applicable.
int combine_midd(DeviceWs* w, const cplx* cur, int B, int count, int Dm, int right_order, const double* fr_phase, cplx* U_out, hipStream_t st);
// segments of the real Hermitian-basis kernels at Dm = 16 (two qubits; one wavefront = four chains, one wavefront per SIMD:
// 4096 chain slots): the S that minimises rounds x segment length within the LDS of both kernels; -1 = none
int pick_segments_r(int B, int N, int K, int Dm, bool need_mult4, bool with_grad) {
  if (c3p_opt(C3P_OPT_smalld_segments) > 0) {
    const long S = c3p_opt(C3P_OPT_smalld_segments);
    if (S >= 1 && S <= N && (!need_mult4 || S % 4 == 0)) return (int)S;
  }
  long best = -1;
  double best_cost = 1e300;
  const long smax = N < 4096 ? N : 4096;
  for (long S = 1; S <= smax; ++S) {
    if (S > 1 && (S % 4) != 0 && S + 3 <= N) continue;
    if (need_mult4 && (S % 4) != 0) continue;
    const int Lmax = (int)((N + S - 1) / S);
    if (c3p_smallr_lds_bytes(Dm, K, Lmax) > (size_t)60 * 1024) continue;
    if (with_grad && c3p_smallr_grad_lds_bytes(Dm, K, Lmax) > (size_t)60 * 1024) continue;
    const long rounds = ((long)B * S + 4095) / 4096;
    const double cost = (double)rounds * (double)(Lmax + 8);
    if (cost < best_cost * (1.0 - 1e-9)) {
      best_cost = cost;
      best = S;
    }
  }
  return (int)best;
}
int run_pwc_smallr(DeviceWs* w, const cplx* h0, long h0_bs, const cplx* hks, long hk_bs, const double* signals, const cplx* clp, double dt,
                   int B, int K, int N, int D, int Dm, const double* fr_phase, cplx* U_out, hipStream_t st) {
  const bool per_sample = (h0_bs != 0) || (hk_bs != 0);
  int S = Dm > 12 ? pick_segments_r(B, N, K, Dm, per_sample, false) : pick_segments(B, N, K, Dm, per_sample);
  if (S < 0) return 1;
  while (S < N && c3p_smallr_lds_bytes(Dm, K, (N + S - 1) / S) > (size_t)60 * 1024) S += per_sample ? 4 : 1;
  if (c3p_smallr_lds_bytes(Dm, K, (N + S - 1) / S) > (size_t)60 * 1024) return 1;
  const int nsamp = per_sample ? B : 1;
  const size_t tdoubles = (size_t)nsamp * c3p_smallr_table_doubles(Dm, K);
  const size_t fl_off = (tdoubles * sizeof(double) + 255) & ~(size_t)255;
  void *tv, *sv;
  if (ws_get(w, SL_TABLES, fl_off + (size_t)nsamp * (1 + K) * sizeof(int), &tv)) return -1;
  if (ws_get(w, SL_SEG_A, (size_t)B * S * Dm * Dm * sizeof(cplx), &sv)) return -1;
  double* tabs = (double*)tv;
  int* flags = reinterpret_cast<int*>(static_cast<char*>(tv) + fl_off);
  RegdPrepArgs p = {};
  p.h0 = h0;
  p.h0_bstride = h0_bs;
  p.hks = hks;
  p.hks_bstride = hk_bs;
  p.clp = clp;
  p.dt = dt;
  p.K = K;
  p.Dh = D;
  p.Dm = Dm;
  p.lindblad = 1;
  LAUNCH_TRY(c3p_launch_smallr_prep(p, nsamp, tabs, flags, st));
  SmallRArgs a = {};
  a.no_t18n = c3p_opt_on(C3P_OPT_no_t18n) ? (int)c3p_opt(C3P_OPT_no_t18n) : 0;
  a.tables = tabs;
  a.tab_per_sample = per_sample ? 1 : 0;
  a.signals = signals;
  a.B = B;
  a.K = K;
  a.N = N;
  a.Dm = Dm;
  a.S = S;
  a.Lmax = (N + S - 1) / S;
  a.seg_out = (cplx*)sv;
  g_last_kernel = C3P_KERNEL_SMALLD;
  if (record_start(w, st)) return -1;
  LAUNCH_TRY(c3p_launch_smallr_chain(a, st));
  if (record_stop(w, st)) return -1;
  if (Dm > 12) {
    g_last_kernel = C3P_KERNEL_MFMA;
    const int keep = g_last_kernel;
    const int rc = combine_midd(w, (const cplx*)sv, B, S, Dm, 0, fr_phase, U_out, st);
    g_last_kernel = keep;
    return rc ? -1 : 0;
  }
  return combine_smalld(w, (const cplx*)sv, B, S, Dm, 0, fr_phase, U_out, st) ? -1 : 0;
}

int run_pwc_smalld(DeviceWs* w, int lindblad, const cplx* h0, long h0_bs, const cplx* hks, long hk_bs,
                   const double* signals, const cplx* clp, double dt, int B, int K, int N, int D, int Dm,
                   const double* fr_phase, cplx* U_out, cplx* dUs_out, hipStream_t st) {
  const bool per_sample = (h0_bs != 0) || (hk_bs != 0);
  const int S = pick_segments(B, N, K, Dm, per_sample);
  if (S < 0) return 1;  // not applicable -> caller falls back to the generic kernel
  const int nsamp = per_sample ? B : 1;
  const bool fuse = (S > 1) && (S % 4 == 0) && !c3p_opt_on(C3P_OPT_no_fuse);
  const bool inline_tables = !lindblad && !c3p_opt_on(C3P_OPT_prep_kernel);
  int* counters = nullptr;
  if (fuse) {
    void* cv;
    if (ws_get(w, SL_COUNTERS, (size_t)B * sizeof(int), &cv)) return -1;
    counters = (int*)cv;
  }
  SmallArgs a = {};
  a.no_t18n = c3p_opt_on(C3P_OPT_no_t18n) ? (int)c3p_opt(C3P_OPT_no_t18n) : 0;
  if (inline_tables) {
    // unitary mode: the chain kernel builds its tables itself (no dependent launch in front of it)
    a.inline_tables = 1;
    a.h0 = h0;
    a.h0_bstride = h0_bs;
    a.hks = hks;
    a.hks_bstride = hk_bs;
    a.dt = dt;
  } else {
    void* v;
    if (ws_get(w, SL_TABLES, (size_t)nsamp * c3p_smalld_table_doubles(Dm, K) * sizeof(double), &v)) return -1;
    PrepArgs p = {};
    p.h0 = h0;
    p.h0_bstride = h0_bs;
    p.hks = hks;
    p.hks_bstride = hk_bs;
    p.clp = clp;
    p.dt = dt;
    p.K = K;
    p.Dh = D;
    p.lindblad = lindblad;
    p.tables = (double*)v;
    LAUNCH_TRY(c3p_launch_smalld_prep(p, Dm, nsamp, st));
    a.tables = (const double*)v;
  }
  a.tab_per_sample = per_sample ? 1 : 0;
  a.signals = signals;
  a.B = B;
  a.K = K;
  a.N = N;
  a.Dm = Dm;
  a.S = S;
  a.Lmax = (N + S - 1) / S;
  a.mode = lindblad ? C3P_MODE_LINDBLAD : C3P_MODE_UNITARY;
  a.dUs_out = dUs_out;
  if (S == 1) {
    a.seg_out = U_out;
    a.fr_phase = fr_phase;
  } else {
    void* sv;
    if (ws_get(w, SL_SEG_A, (size_t)B * S * Dm * Dm * sizeof(cplx), &sv)) return -1;
    a.seg_out = (cplx*)sv;
    if (fuse) {
      a.fuse = 1;
      a.counters = counters;
      a.final_out = U_out;
      a.fr_phase = fr_phase;
    }
  }
  g_last_kernel = C3P_KERNEL_SMALLD;
  if (record_start(w, st)) return -1;
  LAUNCH_TRY(c3p_launch_smalld_chain(a, st));
  if (record_stop(w, st)) return -1;
  if (S > 1 && !fuse) return combine_smalld(w, a.seg_out, B, S, Dm, 0, fr_phase, U_out, st);
  return 0;
}

int combine_midd(DeviceWs* w, const cplx* cur, int B, int count, int Dm, int right_order, const double* fr_phase,
                 cplx* U_out, hipStream_t st);

// Gradient on the mid-D MFMA kernels (13 <= D <= 40): tables, forward segment products (chain kernel, no
// combine), the per-sample scan of c3p_grad.hip, then the pair-T18 backward sweep.  1 = not applicable.
int run_vjp_midd(DeviceWs* w, GradArgs& G, hipStream_t st) {
  const int D = G.D, B = G.B, K = G.K, N = G.N;
  int nig, nj, wd;
  if (!c3p_midd_geometry(D, &nig, &nj, &wd) || K > 16) return 1;
  const size_t img_bytes = (size_t)16 * nig * wd * sizeof(double);
  // backward sweep: one workgroup per CU (two for the real-Hamiltonian kernel at D <= 32); two to four rounds
  long target = D <= 32 ? 2048 : 512;  // (D <= 32: four rounds of the two-per-CU sweep; 1024 -> 2048 measured -2 % at cfg3)
  if (c3p_opt(C3P_OPT_grad_target) > 0) target = c3p_opt(C3P_OPT_grad_target);  // tuning override
  long S = (target + B - 1) / B;
  const long smax = N / 8 > 1 ? N / 8 : 1;
  if (S > smax) S = smax;
  if (S < 1) S = 1;
  (void)img_bytes;
  auto lds_need = [&](long s) { return c3p_midd_grad_image_bytes(D) + (size_t)K * ((N + s - 1) / s) * sizeof(double); };
  while (lds_need(S) > (size_t)150 * 1024 && S < N) ++S;
  if (lds_need(S) > (size_t)150 * 1024) return 1;
  const bool per_sample = (G.h0_bstride != 0) || (G.hks_bstride != 0);
  const int nsamp = per_sample ? B : 1;
  void* v;
  if (ws_get(w, SL_TABLES, (size_t)nsamp * c3p_midd_table_doubles(D, K) * sizeof(double), &v)) return -1;
  MidPrepArgs p = {};
  p.h0 = G.h0;
  p.h0_bstride = G.h0_bstride;
  p.hks = G.hks;
  p.hks_bstride = G.hks_bstride;
  p.dt = G.dt;
  p.K = K;
  p.Dh = D;
  p.Dm = D;
  p.rows = 16 * nig;
  p.W = wd;
  p.tables = (double*)v;
  LAUNCH_TRY(c3p_launch_midd_prep(p, nsamp, st));
  const size_t segb = (size_t)B * S * D * D * sizeof(cplx);
  void *sv, *mv;
  if (ws_get(w, SL_SEG_A, segb, &sv)) return -1;
  if (ws_get(w, SL_SEG_B, segb, &mv)) return -1;
  MidArgs a = {};
  a.tables = p.tables;
  a.tab_per_sample = per_sample ? 1 : 0;
  a.signals = G.signals;
  a.B = B;
  a.K = K;
  a.N = N;
  a.Dm = D;
  a.S = (int)S;
  a.Lmax = (int)((N + S - 1) / S);
  a.mode = C3P_MODE_UNITARY;
  a.seg_out = (cplx*)sv;
  LAUNCH_TRY(c3p_launch_midd_chain(a, st));
  G.S = (int)S;
  G.seg = (cplx*)sv;
  G.Mb = (cplx*)mv;
  const bool scan_global = c3p_grad_lds_bytes(D) > 150 * 1024;
  if (scan_global) {
    // the scan keeps four matrices per sample: one small scratch region per workgroup
    G.scratch_stride = ((long)4 * G.ld * D + G.S - 1) / G.S;
    void* gv;
    if (ws_get(w, SL_SCRATCH, (size_t)B * G.S * G.scratch_stride * sizeof(cplx), &gv)) return -1;
    G.scratch = (cplx*)gv;
  }
  LAUNCH_TRY(c3p_launch_grad_scan(G, scan_global, st));
  MidGradArgs g = {};
  g.tables = p.tables;
  g.tab_per_sample = a.tab_per_sample;
  g.signals = G.signals;
  g.Mb = G.Mb;
  g.grad = G.grad;
  g.zout = G.zout;
  g.B = B;
  g.K = K;
  g.N = N;
  g.Dm = D;
  g.S = (int)S;
  g.Lmax = a.Lmax;
  LAUNCH_TRY(c3p_launch_midd_grad(g, st));
  return 0;
}

// Time segments per sample for the workgroup-per-chain kernels.  B S chains run in ceil(B S / slots) rounds of
// `slots` resident workgroups, each N / S slices long (+ a few slices' worth of prologue): take the S that
// minimises rounds x segment length, so that the last round is not a mostly idle tail.
static long pick_segments_rounds(long B, long N, long slots, long smax, long min_len = 100) {
  if (c3p_opt(C3P_OPT_segments) > 0) {  // tuning override
    const long S = c3p_opt(C3P_OPT_segments);
    if (S >= 1 && S <= (N > 1 ? N : 1)) return S;
  }
  long best = 1;
  double best_cost = 1e300;
  if (smax > 96) smax = 96;
  for (long S = 1; S <= smax; ++S) {
    const long rounds = (B * S + slots - 1) / slots;
    const double cost = (double)rounds * (double)((N + S - 1) / S + 4);
    if (cost < best_cost * (1.0 - 1e-9)) {
      best_cost = cost;
      best = S;
    }
  }
  // The workgroups that share a SIMD do not advance at the same rate (the arbiter serves the older wave first), so a
  // round does not end for all of them at once: with MANY more workgroups than slots the SIMDs stay shared until the
  // very end, with one or two rounds the last workgroup of every CU runs a good part of its segment alone.  Measured
  // (cfg5, B = 1024): S = 1 (2 rounds) 6.13e3, S = 4 6.28e3, S = 12 6.33e3 propagators/s; cfg3 (B = 512): 3 -> 12 segments
  // +1 %.  At least 12 rounds, segments of at least min_len slices (100; 400 in the 16-row class, whose slices are short
  // against the per-segment prologue: D = 13 measured 20 % slower with 100).
  if (!c3p_opt_on(C3P_OPT_no_many_rounds)) {
    long want = (12 * slots + B - 1) / B;
    const long cap = N / min_len > 1 ? N / min_len : 1;
    if (want > cap) want = cap;
    if (want > smax) want = smax;
    if (want > best) best = want;
  }
  return best;
}

// Supplied generators on the mid-D MFMA kernel (13 <= D <= 40); see run_xg_smalld.
int run_xg_midd(DeviceWs* w, const cplx* hs, long hs_bstride, double coef_r, double coef_i, int B, int N, int D,
                const double* fr_phase, cplx* U_out, cplx* dUs_out, hipStream_t st) {
  int nig, nj, wd;
  if (!c3p_midd_geometry(D, &nig, &nj, &wd)) return 1;
  const size_t lds0 = c3p_midd_lds_bytes(D, 0, 0);
  int wg_per_cu = (int)((156 * 1024) / (lds0 + 4096));
  if (wg_per_cu > 3) wg_per_cu = 3;
  if (wg_per_cu < 1) wg_per_cu = 1;
  const long smax = N / 8 > 1 ? N / 8 : 1;
  const long S = pick_segments_rounds(B, N, 256L * wg_per_cu, smax, D <= 16 ? 400 : 100);
  void* mv;
  if (ws_get(w, SL_TABLES, (size_t)B * N * 4 * sizeof(double), &mv)) return -1;
  LAUNCH_TRY(c3p_launch_hmeta(hs, hs_bstride, (long)B * N, N, D, coef_r, coef_i, (double*)mv, st));
  MidArgs a = {};
  a.hs = hs;
  a.hs_bstride = hs_bstride;
  a.meta = (const double*)mv;
  a.coef_r = coef_r;
  a.coef_i = coef_i;
  a.B = B;
  a.K = 0;
  a.N = N;
  a.Dm = D;
  a.S = (int)S;
  a.Lmax = (int)((N + S - 1) / S);
  a.mode = C3P_MODE_EXPM;
  a.dUs_out = dUs_out;
  a.no_t18 = c3p_opt_on(C3P_OPT_no_t18) ? 1 : 0;
  a.no_t18n = c3p_opt_on(C3P_OPT_no_t18n) ? (int)c3p_opt(C3P_OPT_no_t18n) : 0;
  a.no_real = c3p_opt_on(C3P_OPT_no_real) ? 1 : 0;
  if (S == 1) {
    a.seg_out = U_out;
    a.fr_phase = fr_phase;
  } else {
    void* sv;
    if (ws_get(w, SL_SEG_A, (size_t)B * S * D * D * sizeof(cplx), &sv)) return -1;
    a.seg_out = (cplx*)sv;
  }
  g_last_kernel = C3P_KERNEL_MFMA;
  if (record_start(w, st)) return -1;
  LAUNCH_TRY(c3p_launch_midd_chain(a, st));
  if (record_stop(w, st)) return -1;
  if (S > 1) return combine_midd(w, a.seg_out, B, (int)S, D, 0, fr_phase, U_out, st);
  return 0;
}

// Supplied generators (branch B of pwc: per-slice Hamiltonians, and c3p_expm) on the small-D MFMA kernel:
// X_n = coef * hs[b,n]; the hmeta pre-pass supplies the trace shift and the norm of every matrix.
int run_xg_smalld(DeviceWs* w, const cplx* hs, long hs_bstride, double coef_r, double coef_i, int B, int N, int D,
                  const double* fr_phase, cplx* U_out, cplx* dUs_out, hipStream_t st) {
  const int S = pick_segments(B, N, 0, D, false);
  if (S < 0) return 1;
  void* mv;
  if (ws_get(w, SL_TABLES, (size_t)B * N * 4 * sizeof(double), &mv)) return -1;
  LAUNCH_TRY(c3p_launch_hmeta(hs, hs_bstride, (long)B * N, N, D, coef_r, coef_i, (double*)mv, st));
  SmallArgs a = {};
  a.no_t18n = c3p_opt_on(C3P_OPT_no_t18n) ? (int)c3p_opt(C3P_OPT_no_t18n) : 0;
  a.hs = hs;
  a.hs_bstride = hs_bstride;
  a.meta = (const double*)mv;
  a.coef_r = coef_r;
  a.coef_i = coef_i;
  a.B = B;
  a.K = 0;
  a.N = N;
  a.Dm = D;
  a.S = S;
  a.Lmax = (N + S - 1) / S;
  a.mode = C3P_MODE_EXPM;
  a.dUs_out = dUs_out;
  const bool fuse = (S > 1) && (S % 4 == 0) && !c3p_opt_on(C3P_OPT_no_fuse);
  if (S == 1) {
    a.seg_out = U_out;
    a.fr_phase = fr_phase;
  } else {
    void* sv;
    if (ws_get(w, SL_SEG_A, (size_t)B * S * D * D * sizeof(cplx), &sv)) return -1;
    a.seg_out = (cplx*)sv;
    if (fuse) {
      void* cv;
      if (ws_get(w, SL_COUNTERS, (size_t)B * sizeof(int), &cv)) return -1;
      a.fuse = 1;
      a.counters = (int*)cv;
      a.final_out = U_out;
      a.fr_phase = fr_phase;
    }
  }
  g_last_kernel = C3P_KERNEL_SMALLD;
  if (record_start(w, st)) return -1;
  LAUNCH_TRY(c3p_launch_smalld_chain(a, st));
  if (record_stop(w, st)) return -1;
  if (S > 1 && !fuse) return combine_smalld(w, a.seg_out, B, S, D, 0, fr_phase, U_out, st);
  return 0;
}

// Gradient on the small-D MFMA kernels: forward segment products (unfused smalld chain kernel), the
// per-sample scan of c3p_grad.hip, then the pair-T18 backward sweep.  Returns 1 when not applicable.
int run_vjp_smalld(DeviceWs* w, GradArgs& G, hipStream_t st) {
  const int D = G.D, B = G.B, K = G.K, N = G.N;
  const bool per_sample = (G.h0_bstride != 0) || (G.hks_bstride != 0);
  const int S = pick_segments(B, N, K, D, per_sample, c3p_opt(C3P_OPT_grad_slots) > 0 ? c3p_opt(C3P_OPT_grad_slots) : 4096);
  if (S < 0) return 1;
  const int nsamp = per_sample ? B : 1;
  void* v;
  if (ws_get(w, SL_TABLES, (size_t)nsamp * c3p_smalld_table_doubles(D, K) * sizeof(double), &v)) return -1;
  PrepArgs p = {};
  p.h0 = G.h0;
  p.h0_bstride = G.h0_bstride;
  p.hks = G.hks;
  p.hks_bstride = G.hks_bstride;
  p.dt = G.dt;
  p.K = K;
  p.Dh = D;
  p.tables = (double*)v;
  LAUNCH_TRY(c3p_launch_smalld_prep(p, D, nsamp, st));
  const size_t segb = (size_t)B * S * D * D * sizeof(cplx);
  void *sv, *mv;
  if (ws_get(w, SL_SEG_A, segb, &sv)) return -1;
  if (ws_get(w, SL_SEG_B, segb, &mv)) return -1;
  // The backward sweep runs one wave per SIMD (S is chosen for its 4096 chain slots), the forward chain kernel two: the
  // segment products are formed on 2 S segments (twice the waves: 0.24 -> 0.15 ms at cfg2) and multiplied in pairs by one
  // launch of the chain kernel's supplied-matrix mode (grad_fwd_seg2 = 0: one pass on S segments)
  int Sf = S;
  void* fv = sv;
  if (c3p_opt(C3P_OPT_grad_fwd_seg2) != 0 && (long)B * S <= 4096 && 2L * S <= N / 8) {
    Sf = 2 * S;
    if (ws_get(w, SL_SEG_F, 2 * segb, &fv)) return -1;
  }
  SmallArgs a = {};
  a.no_t18n = c3p_opt_on(C3P_OPT_no_t18n) ? (int)c3p_opt(C3P_OPT_no_t18n) : 0;
  a.tables = p.tables;
  a.tab_per_sample = per_sample ? 1 : 0;
  a.signals = G.signals;
  a.B = B;
  a.K = K;
  a.N = N;
  a.Dm = D;
  a.S = Sf;
  a.Lmax = (N + Sf - 1) / Sf;
  a.mode = C3P_MODE_UNITARY;
  a.seg_out = (cplx*)fv;
  LAUNCH_TRY(c3p_launch_smalld_chain(a, st));
  if (Sf != S) {
    SmallArgs c = {};
    c.mode = C3P_MODE_GIVEN;
    c.mats = (const cplx*)fv;
    c.B = B * S;
    c.N = 2;
    c.Dm = D;
    c.S = 1;
    c.Lmax = 2;
    c.seg_out = (cplx*)sv;
    LAUNCH_TRY(c3p_launch_smalld_chain(c, st));
  }
  a.S = S;
  a.Lmax = (N + S - 1) / S;
  G.S = S;
  G.seg = (cplx*)sv;
  G.Mb = (cplx*)mv;
  LAUNCH_TRY(c3p_launch_grad_scan(G, false, st));
  SmallGradArgs g = {};
  g.tables = p.tables;
  g.tab_per_sample = a.tab_per_sample;
  g.signals = G.signals;
  g.Mb = G.Mb;
  g.grad = G.grad;
  g.zout = G.zout;
  g.B = B;
  g.K = K;
  g.N = N;
  g.Dm = D;
  g.S = S;
  g.Lmax = a.Lmax;
  LAUNCH_TRY(c3p_launch_smalld_grad(g, st));
  return 0;
}

// Lindblad gradient on the small-D MFMA kernels (superoperators up to 12 x 12, D <= 3), general-generator form: slice
// propagators + segment products from the forward chain kernel, the general scan of c3p_grad.hip (prefix at the start, left
// adjoint at the end of every segment), then smalld_grad_general_kernel.  In two halves around a block of memory that holds
// what the forward half leaves for the backward one -- the workspace (c3p_pwc_lindblad_vjp) or a caller-owned tape
// (c3p_pwc_lindblad_taped / _vjp_taped: the forward half also serves U, one chain pass per evaluation instead of two).
struct LindSmallSizes {
  size_t tabs, seg, dus;  // bytes: both table sets, segment products [B,S,Dm,Dm], slice propagators [B,N,Dm,Dm]
  size_t total() const { return tabs + seg + dus; }
};
LindSmallSizes lind_small_sizes(int B, int K, int N, int Dm, int S, int nsamp) {
  const size_t msz = (size_t)Dm * Dm * sizeof(cplx);
  auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
  LindSmallSizes z;
  z.tabs = up(2 * (size_t)nsamp * c3p_smalld_table_doubles(Dm, K) * sizeof(double));
  z.seg = up((size_t)B * S * msz);
  z.dus = up((size_t)B * N * msz);
  return z;
}
struct LindSmallBufs {
  double* tabs;
  cplx *seg, *dus;
};
LindSmallBufs lind_small_carve(void* base, const LindSmallSizes& z) {
  char* p = static_cast<char*>(base);
  LindSmallBufs b;
  b.tabs = reinterpret_cast<double*>(p);
  b.seg = reinterpret_cast<cplx*>(p + z.tabs);
  b.dus = reinterpret_cast<cplx*>(p + z.tabs + z.seg);
  return b;
}
// segments of the small-D Lindblad sweep, -1 when the shape is not served (tables + images beyond the LDS of the backward kernel)
int lind_small_segments(int B, int K, int N, int Dm, bool need_mult4) {
  if (Dm > 12) return -1;
  // (8192 chain slots: the real sweep of c3p_smallr.hip runs two wavefronts per SIMD; for the complex one -- one per SIMD -- two
  // rounds of half-length segments cost what one round did)
  int S = pick_segments(B, N, K, Dm, need_mult4, 8192);
  // (short segments cost more than the cost model of pick_segments knows here: the segment scan is sequential over S and every
  // chain has a prologue -- below 8 slices per segment the 4096-slot choice is the faster one (with the blocked scan of
  // c3p_smallr.hip; 16 before it): B = 64, N = 1000 takes 128 segments of 8, 0.286 -> 0.259 ms per gradient call)
  if (S > 0 && N / S < 8) S = pick_segments(B, N, K, Dm, need_mult4, 4096);
  if (S < 0) return -1;
  // (the backward kernel keeps BOTH table sets in LDS)
  if ((2 * c3p_smalld_table_doubles(Dm, K) + 8 * (size_t)c3p_smalld_mat_doubles(Dm) + 4 * (size_t)K * ((N + S - 1) / S)) * sizeof(double) > 60 * 1024)
    return -1;
  return S;
}
int lind_small_forward(DeviceWs* w, const LindSmallBufs& bf, const cplx* h0, long h0_bs, const cplx* hks, long hk_bs, const double* signals,
                       const cplx* clp, double dt, int B, int K, int N, int D, int Dm, int S, bool hermitian, hipStream_t st) {
  const bool per_sample = (h0_bs != 0) || (hk_bs != 0);
  const int nsamp = per_sample ? B : 1;
  const size_t tdoubles = (size_t)nsamp * c3p_smalld_table_doubles(Dm, K);
  PrepArgs p = {};
  p.h0 = h0;
  p.h0_bstride = h0_bs;
  p.hks = hks;
  p.hks_bstride = hk_bs;
  p.clp = clp;
  p.dt = dt;
  p.K = K;
  p.Dh = D;
  p.lindblad = 1;
  p.tables = bf.tabs;
  LAUNCH_TRY(c3p_launch_smalld_prep(p, Dm, nsamp, st));
  p.conjT = 1;
  p.tables = bf.tabs + tdoubles;
  LAUNCH_TRY(c3p_launch_smalld_prep(p, Dm, nsamp, st));
  if (hermitian && c3p_smallr_supported(D, Dm, K) && !c3p_opt_on(C3P_OPT_no_smallr) &&
      c3p_smallr_lds_bytes(Dm, K, (N + S - 1) / S) <= (size_t)60 * 1024) {
    // declared Hermitian Hamiltonians (C3P_HERMITIAN_H): segment products and slice propagators from the real kernels of
    // c3p_smallr.hip (the backward sweep reads them in the complex vectorisation: the kernel converts what it stores)
    const size_t rdoubles = (size_t)nsamp * c3p_smallr_table_doubles(Dm, K);
    const size_t fl_off = (rdoubles * sizeof(double) + 255) & ~(size_t)255;
    void* tv;
    if (ws_get(w, SL_TABLES, fl_off + (size_t)nsamp * (1 + K) * sizeof(int), &tv)) return -1;
    RegdPrepArgs rp = {};
    rp.h0 = h0;
    rp.h0_bstride = h0_bs;
    rp.hks = hks;
    rp.hks_bstride = hk_bs;
    rp.clp = clp;
    rp.dt = dt;
    rp.K = K;
    rp.Dh = D;
    rp.Dm = Dm;
    rp.lindblad = 1;
    LAUNCH_TRY(c3p_launch_smallr_prep(rp, nsamp, (double*)tv, reinterpret_cast<int*>(static_cast<char*>(tv) + fl_off), st));
    SmallRArgs ra = {};
    ra.no_t18n = c3p_opt_on(C3P_OPT_no_t18n) ? (int)c3p_opt(C3P_OPT_no_t18n) : 0;
  ra.no_t18n = c3p_opt_on(C3P_OPT_no_t18n) ? (int)c3p_opt(C3P_OPT_no_t18n) : 0;
    ra.tables = (const double*)tv;
    ra.tab_per_sample = per_sample ? 1 : 0;
    ra.signals = signals;
    ra.B = B;
    ra.K = K;
    ra.N = N;
    ra.Dm = Dm;
    ra.S = S;
    ra.Lmax = (N + S - 1) / S;
    ra.seg_out = bf.seg;
    ra.dUs_out = bf.dus;
    LAUNCH_TRY(c3p_launch_smallr_chain(ra, st));
    return 0;
  }
  SmallArgs a = {};
  a.no_t18n = c3p_opt_on(C3P_OPT_no_t18n) ? (int)c3p_opt(C3P_OPT_no_t18n) : 0;
  a.tables = bf.tabs;
  a.tab_per_sample = per_sample ? 1 : 0;
  a.signals = signals;
  a.B = B;
  a.K = K;
  a.N = N;
  a.Dm = Dm;
  a.S = S;
  a.Lmax = (N + S - 1) / S;
  a.mode = C3P_MODE_LINDBLAD;
  a.seg_out = bf.seg;
  a.dUs_out = bf.dus;
  LAUNCH_TRY(c3p_launch_smalld_chain(a, st));
  return 0;
}
int lind_small_backward(DeviceWs* w, const LindSmallBufs& bf, bool per_sample, const double* signals, int B, int K, int N, int Dm, int S,
                        const double* fr_phase, const cplx* Ubar, double* grad, hipStream_t st) {
  const int nsamp = per_sample ? B : 1;
  const size_t tdoubles = (size_t)nsamp * c3p_smalld_table_doubles(Dm, K);
  const size_t msz = (size_t)Dm * Dm * sizeof(cplx);
  void *mv, *bv;
  if (ws_get(w, SL_SEG_B, (size_t)B * S * msz, &mv)) return -1;
  if (ws_get(w, SL_SEG_A, ((size_t)B * S + (size_t)B * N) * msz, &bv)) return -1;  // (SL_OUT0 stages grad_signals, SL_OUT1 is the untaped block)
  cplx* pre = (cplx*)bv;
  cplx* pstore = pre + (size_t)B * S * Dm * Dm;
  GradArgs G = {};
  G.Ubar = Ubar;
  G.fr_phase = fr_phase;
  G.B = B;
  G.K = K;
  G.N = N;
  G.D = Dm;
  G.ld = Dm | 1;
  G.S = S;
  G.seg = bf.seg;
  G.Mb = (cplx*)mv;
  G.pre = pre;
  G.general = 1;
  LAUNCH_TRY(c3p_launch_grad_scan_general(G, false, st));
  SmallGradArgs g = {};
  g.tables = bf.tabs;
  g.tables_h = bf.tabs + tdoubles;
  g.tab_per_sample = per_sample ? 1 : 0;
  g.signals = signals;
  g.Mb = G.Mb;
  g.pre = pre;
  g.dUs = bf.dus;
  g.pstore = pstore;
  g.grad = grad;
  g.B = B;
  g.K = K;
  g.N = N;
  g.Dm = Dm;
  g.S = S;
  g.Lmax = (N + S - 1) / S;
  LAUNCH_TRY(c3p_launch_smalld_grad_general(g, st));
  return 0;
}
// The same two halves in REAL arithmetic in the Hermitian basis (declared Hermitian Hamiltonians, C3P_HERMITIAN_H; c3p_smallr.hip):
// real tables of G' and G'^T, real segment products and slice propagators from the forward half; the cotangent in the basis
// (hb_ubar), the real segment scan and the real pair-evaluation sweep in the backward half.  The block is laid out inside
// the memory the complex halves would use (it is smaller in every part).
struct LindSmallRSizes {
  size_t tabs, seg, dus;
  size_t total() const { return tabs + seg + dus; }
};
LindSmallRSizes lind_smallr_sizes(int B, int K, int N, int Dm, int S, int nsamp) {
  auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
  LindSmallRSizes z;
  z.tabs = up(2 * (size_t)nsamp * c3p_smallr_table_doubles(Dm, K) * sizeof(double)) + up((size_t)nsamp * (1 + K) * sizeof(int));
  z.seg = up((size_t)B * S * Dm * Dm * sizeof(double));
  z.dus = up((size_t)B * N * Dm * Dm * sizeof(double));
  return z;
}
struct LindSmallRBufs {
  double *tabs, *tabs_t;
  int* flags;
  double *seg, *dus;
};
LindSmallRBufs lind_smallr_carve(void* base, const LindSmallRSizes& z, int nsamp, int Dm, int K) {
  char* p = static_cast<char*>(base);
  const size_t td = (size_t)nsamp * c3p_smallr_table_doubles(Dm, K);
  LindSmallRBufs b;
  b.tabs = reinterpret_cast<double*>(p);
  b.tabs_t = b.tabs + td;
  b.flags = reinterpret_cast<int*>(p + ((2 * td * sizeof(double) + 255) & ~(size_t)255));
  b.seg = reinterpret_cast<double*>(p + z.tabs);
  b.dus = reinterpret_cast<double*>(p + z.tabs + z.seg);
  return b;
}
bool lind_smallr_ok(bool hermitian, int D, int Dm, int K, int N, int S) {
  return hermitian && c3p_smallr_supported(D, Dm, K) && !c3p_opt_on(C3P_OPT_no_smallr) &&
         c3p_smallr_lds_bytes(Dm, K, (N + S - 1) / S) <= (size_t)60 * 1024 && c3p_smallr_grad_lds_bytes(Dm, K, (N + S - 1) / S) <= (size_t)60 * 1024;
}
// forward half; seg_complex (workspace, [B,S,Dm,Dm]) receives the segment products in the reference's vectorisation for U
int lind_smallr_forward(const LindSmallRBufs& bf, const cplx* h0, long h0_bs, const cplx* hks, long hk_bs, const double* signals, const cplx* clp,
                        double dt, int B, int K, int N, int D, int Dm, int S, cplx* seg_complex, hipStream_t st) {
  const bool per_sample = (h0_bs != 0) || (hk_bs != 0);
  const int nsamp = per_sample ? B : 1;
  RegdPrepArgs rp = {};
  rp.h0 = h0;
  rp.h0_bstride = h0_bs;
  rp.hks = hks;
  rp.hks_bstride = hk_bs;
  rp.clp = clp;
  rp.dt = dt;
  rp.K = K;
  rp.Dh = D;
  rp.Dm = Dm;
  rp.lindblad = 1;
  LAUNCH_TRY(c3p_launch_smallr_prep_pair(rp, nsamp, bf.tabs, bf.tabs_t, bf.flags, st));
  SmallRArgs ra = {};
  ra.no_t18n = c3p_opt_on(C3P_OPT_no_t18n) ? (int)c3p_opt(C3P_OPT_no_t18n) : 0;
  ra.tables = bf.tabs;
  ra.tab_per_sample = per_sample ? 1 : 0;
  ra.signals = signals;
  ra.B = B;
  ra.K = K;
  ra.N = N;
  ra.Dm = Dm;
  ra.S = S;
  ra.Lmax = (N + S - 1) / S;
  ra.seg_out = seg_complex;
  ra.seg_real = bf.seg;
  ra.dus_real = bf.dus;
  LAUNCH_TRY(c3p_launch_smallr_chain(ra, st));
  return 0;
}
int lind_smallr_backward(DeviceWs* w, const LindSmallRBufs& bf, bool per_sample, const double* signals, int B, int K, int N, int D, int Dm, int S,
                         const double* fr_phase, const cplx* Ubar, double* grad, hipStream_t st) {
  const size_t m8 = (size_t)Dm * Dm * sizeof(double);
  void *uv, *pv, *sv;
  if (ws_get(w, SL_SCRATCH, (size_t)B * m8, &uv)) return -1;
  if (ws_get(w, SL_SEG_A, (size_t)B * S * m8, &pv)) return -1;
  if (ws_get(w, SL_SEG_B, (size_t)B * S * m8, &sv)) return -1;
  LAUNCH_TRY(c3p_launch_hb_ubar(Ubar, fr_phase, B, D, (double*)uv, st));
  LAUNCH_TRY(c3p_launch_smallr_scan(bf.seg, (const double*)uv, B, S, Dm, (double*)pv, (double*)sv, st));
  SmallRGradArgs g = {};
  g.no_t18n = c3p_opt_on(C3P_OPT_no_t18n) ? (int)c3p_opt(C3P_OPT_no_t18n) : 0;
  g.tables = bf.tabs;
  g.tables_t = bf.tabs_t;
  g.tab_per_sample = per_sample ? 1 : 0;
  g.signals = signals;
  g.pre = (const double*)pv;
  g.suf = (const double*)sv;
  g.dus = bf.dus;
  g.grad = grad;
  g.B = B;
  g.K = K;
  g.N = N;
  g.Dm = Dm;
  g.S = S;
  g.Lmax = (N + S - 1) / S;
  LAUNCH_TRY(c3p_launch_smallr_grad(g, st));
  return 0;
}

// two qubits (16 x 16 superoperators), declared Hermitian: both halves on the real kernels; 1 = not applicable
int run_vjp_lind_smallr16(DeviceWs* w, const cplx* h0, long h0_bs, const cplx* hks, long hk_bs, const double* signals, const cplx* clp,
                          double dt, int B, int K, int N, int D, int Dm, const double* fr_phase, const cplx* Ubar, double* grad, hipStream_t st) {
  const bool per_sample = (h0_bs != 0) || (hk_bs != 0);
  const int S = pick_segments_r(B, N, K, Dm, per_sample, true);
  if (S < 0 || !lind_smallr_ok(true, D, Dm, K, N, S)) return 1;
  const int nsamp = per_sample ? B : 1;
  const LindSmallRSizes zr = lind_smallr_sizes(B, K, N, Dm, S, nsamp);
  void *blk, *sc;
  if (ws_get(w, SL_OUT1, zr.total(), &blk)) return -1;
  if (ws_get(w, SL_OUT2, (size_t)B * S * Dm * Dm * sizeof(cplx), &sc)) return -1;
  const LindSmallRBufs rb = lind_smallr_carve(blk, zr, nsamp, Dm, K);
  if (lind_smallr_forward(rb, h0, h0_bs, hks, hk_bs, signals, clp, dt, B, K, N, D, Dm, S, (cplx*)sc, st)) return -1;
  return lind_smallr_backward(w, rb, per_sample, signals, B, K, N, D, Dm, S, fr_phase, Ubar, grad, st) ? -1 : 0;
}

// Returns 1 when not applicable.
int run_vjp_lind_smalld(DeviceWs* w, const cplx* h0, long h0_bs, const cplx* hks, long hk_bs, const double* signals, const cplx* clp,
                        double dt, int B, int K, int N, int D, int Dm, const double* fr_phase, const cplx* Ubar, double* grad,
                        bool hermitian, hipStream_t st) {
  const bool per_sample = (h0_bs != 0) || (hk_bs != 0);
  const int S = lind_small_segments(B, K, N, Dm, per_sample);
  if (S < 0) return 1;
  const LindSmallSizes z = lind_small_sizes(B, K, N, Dm, S, per_sample ? B : 1);
  void* blk;
  if (ws_get(w, SL_OUT1, z.total(), &blk)) return -1;
  if (lind_smallr_ok(hermitian, D, Dm, K, N, S)) {
    const int nsamp = per_sample ? B : 1;
    const LindSmallRSizes zr = lind_smallr_sizes(B, K, N, Dm, S, nsamp);
    const LindSmallRBufs rb = lind_smallr_carve(blk, zr, nsamp, Dm, K);
    void* sc;
    if (ws_get(w, SL_OUT2, (size_t)B * S * Dm * Dm * sizeof(cplx), &sc)) return -1;  // (the complex segment products: not needed here)
    if (lind_smallr_forward(rb, h0, h0_bs, hks, hk_bs, signals, clp, dt, B, K, N, D, Dm, S, (cplx*)sc, st)) return -1;
    return lind_smallr_backward(w, rb, per_sample, signals, B, K, N, D, Dm, S, fr_phase, Ubar, grad, st) ? -1 : 0;
  }
  const LindSmallBufs bf = lind_small_carve(blk, z);
  if (lind_small_forward(w, bf, h0, h0_bs, hks, hk_bs, signals, clp, dt, B, K, N, D, Dm, S, hermitian, st)) return -1;
  return lind_small_backward(w, bf, per_sample, signals, B, K, N, Dm, S, fr_phase, Ubar, grad, st) ? -1 : 0;
}

// ---------------------------------------------------------------------------
// Mid-D MFMA path (13 <= Dm <= 40): one 4-wave workgroup per chain, matrices as LDS images
// ---------------------------------------------------------------------------
int combine_midd(DeviceWs* w, const cplx* cur, int B, int count, int Dm, int right_order,
                 const double* fr_phase, cplx* U_out, hipStream_t st) {
  const size_t msz = (size_t)Dm * Dm * sizeof(cplx);
  Slot next_slot = SL_SEG_B;
  while (true) {
    MidArgs c = {};
    c.mode = C3P_MODE_GIVEN;
    c.mats = cur;
    c.B = B;
    c.N = count;
    c.Dm = Dm;
    c.right_order = right_order;
    if (count <= 8) {
      c.S = 1;
      c.seg_out = U_out;
      c.fr_phase = fr_phase;
    } else {
      c.S = (count + 3) / 4;
      void* v;
      if (ws_get(w, next_slot, (size_t)B * c.S * msz, &v)) return -1;
      c.seg_out = (cplx*)v;
    }
    c.Lmax = (count + c.S - 1) / c.S;
    LAUNCH_TRY(c3p_launch_midd_chain(c, st));
    if (c.seg_out == U_out) break;
    cur = c.seg_out;
    count = c.S;
    next_slot = (next_slot == SL_SEG_B) ? SL_SEG_A : SL_SEG_B;
  }
  return 0;
}

int run_pwc_midd(DeviceWs* w, int lindblad, const cplx* h0, long h0_bs, const cplx* hks, long hk_bs,
                 const double* signals, const cplx* clp, double dt, int B, int K, int N, int D, int Dm,
                 const double* fr_phase, cplx* U_out, cplx* dUs_out, hipStream_t st) {
  int nig, nj, wd;
  if (!c3p_midd_geometry(Dm, &nig, &nj, &wd)) return 1;
  const bool per_sample = (h0_bs != 0) || (hk_bs != 0);
  // workgroups resident at once are LDS-limited (three images per workgroup); aim at two rounds
  const size_t lds0 = c3p_midd_lds_bytes(Dm, K, 0);
  int wg_per_cu = (int)((156 * 1024) / (lds0 + 4096));
  if (wg_per_cu > 3) wg_per_cu = 3;
  if (wg_per_cu < 1) wg_per_cu = 1;
  const long smax = N / 8 > 1 ? N / 8 : 1;
  const long S = pick_segments_rounds(B, N, 256L * wg_per_cu, smax, Dm <= 16 ? 400 : 100);
  if (lds0 > 158 * 1024) return 1;
  const int nsamp = per_sample ? B : 1;
  void* v;
  if (ws_get(w, SL_TABLES, (size_t)nsamp * c3p_midd_table_doubles(Dm, K) * sizeof(double), &v)) return -1;
  MidPrepArgs p = {};
  p.h0 = h0;
  p.h0_bstride = h0_bs;
  p.hks = hks;
  p.hks_bstride = hk_bs;
  p.clp = clp;
  p.dt = dt;
  p.K = K;
  p.Dh = D;
  p.Dm = Dm;
  p.lindblad = lindblad;
  p.rows = 16 * nig;
  p.W = wd;
  p.tables = (double*)v;
  LAUNCH_TRY(c3p_launch_midd_prep(p, nsamp, st));
  MidArgs a = {};
  a.tables = (const double*)v;
  a.tab_per_sample = per_sample ? 1 : 0;
  a.signals = signals;
  a.B = B;
  a.K = K;
  a.N = N;
  a.Dm = Dm;
  a.S = (int)S;
  a.Lmax = (int)((N + S - 1) / S);
  a.mode = lindblad ? C3P_MODE_LINDBLAD : C3P_MODE_UNITARY;
  a.dUs_out = dUs_out;
  a.no_t18 = c3p_opt_on(C3P_OPT_no_t18) ? 1 : 0;
  a.no_t18n = c3p_opt_on(C3P_OPT_no_t18n) ? (int)c3p_opt(C3P_OPT_no_t18n) : 0;
  a.no_real = c3p_opt_on(C3P_OPT_no_real) ? 1 : 0;
  if (S == 1) {
    a.seg_out = U_out;
    a.fr_phase = fr_phase;
  } else {
    void* sv;
    if (ws_get(w, SL_SEG_A, (size_t)B * S * Dm * Dm * sizeof(cplx), &sv)) return -1;
    a.seg_out = (cplx*)sv;
  }
  g_last_kernel = C3P_KERNEL_MFMA;
  if (record_start(w, st)) return -1;
  LAUNCH_TRY(c3p_launch_midd_chain(a, st));
  if (record_stop(w, st)) return -1;
  if (S > 1) return combine_midd(w, a.seg_out, B, (int)S, Dm, 0, fr_phase, U_out, st);
  return 0;
}

// Lindblad gradient on the mid-D MFMA kernels (16 x 16, 25 x 25, 36 x 36 superoperators: D = 4, 5, 6), general-generator
// form; see run_vjp_lind_smalld.  Returns 1 when not applicable.
int run_vjp_lind_midd(DeviceWs* w, const cplx* h0, long h0_bs, const cplx* hks, long hk_bs, const double* signals, const cplx* clp,
                      double dt, int B, int K, int N, int D, int Dm, const double* fr_phase, const cplx* Ubar, double* grad,
                      hipStream_t st) {
  int nig, nj, wd;
  if (!c3p_midd_geometry(Dm, &nig, &nj, &wd) || K > 16) return 1;
  if (!((nig == 2 && nj == 4) || (nig == 4 && nj == 7) || (nig == 5 && nj == 9))) return 1;
  long S = ((Dm <= 32 ? 1024 : 512) + B - 1) / B;  // one workgroup per CU: two to four rounds
  const long smax = N / 8 > 1 ? N / 8 : 1;
  if (S > smax) S = smax;
  if (S < 1) S = 1;
  auto lds_need = [&](long s) { return c3p_midd_grad_image_bytes(Dm) + (size_t)K * ((N + s - 1) / s) * sizeof(double); };
  while (lds_need(S) > (size_t)150 * 1024 && S < N) ++S;
  if (lds_need(S) > (size_t)150 * 1024) return 1;
  const bool per_sample = (h0_bs != 0) || (hk_bs != 0);
  const int nsamp = per_sample ? B : 1;
  const size_t tdoubles = (size_t)nsamp * c3p_midd_table_doubles(Dm, K);
  void* v;
  if (ws_get(w, SL_TABLES, 2 * tdoubles * sizeof(double), &v)) return -1;
  double* tabs = (double*)v;
  MidPrepArgs p = {};
  p.h0 = h0;
  p.h0_bstride = h0_bs;
  p.hks = hks;
  p.hks_bstride = hk_bs;
  p.clp = clp;
  p.dt = dt;
  p.K = K;
  p.Dh = D;
  p.Dm = Dm;
  p.lindblad = 1;
  p.rows = 16 * nig;
  p.W = wd;
  p.tables = tabs;
  LAUNCH_TRY(c3p_launch_midd_prep(p, nsamp, st));
  p.conjT = 1;
  p.tables = tabs + tdoubles;
  LAUNCH_TRY(c3p_launch_midd_prep(p, nsamp, st));
  const size_t msz = (size_t)Dm * Dm * sizeof(cplx);
  void *sv, *mv, *bv;
  if (ws_get(w, SL_SEG_A, (size_t)B * S * msz, &sv)) return -1;
  if (ws_get(w, SL_SEG_B, (size_t)B * S * msz, &mv)) return -1;
  if (ws_get(w, SL_OUT1, ((size_t)B * S + 2 * (size_t)B * N) * msz, &bv)) return -1;  // (SL_OUT0 stages grad_signals)
  cplx* pre = (cplx*)bv;
  cplx* dUs = pre + (size_t)B * S * Dm * Dm;
  cplx* pstore = dUs + (size_t)B * N * Dm * Dm;
  MidArgs a = {};
  a.tables = tabs;
  a.tab_per_sample = per_sample ? 1 : 0;
  a.signals = signals;
  a.B = B;
  a.K = K;
  a.N = N;
  a.Dm = Dm;
  a.S = (int)S;
  a.Lmax = (int)((N + S - 1) / S);
  a.mode = C3P_MODE_LINDBLAD;
  a.seg_out = (cplx*)sv;
  a.dUs_out = dUs;
  a.no_t18 = c3p_opt_on(C3P_OPT_no_t18) ? 1 : 0;
  a.no_t18n = c3p_opt_on(C3P_OPT_no_t18n) ? (int)c3p_opt(C3P_OPT_no_t18n) : 0;
  a.no_real = 1;
  LAUNCH_TRY(c3p_launch_midd_chain(a, st));
  GradArgs G = {};
  G.Ubar = Ubar;
  G.fr_phase = fr_phase;
  G.B = B;
  G.K = K;
  G.N = N;
  G.D = Dm;
  G.ld = Dm | 1;
  G.S = (int)S;
  G.seg = (cplx*)sv;
  G.Mb = (cplx*)mv;
  G.pre = pre;
  G.general = 1;
  LAUNCH_TRY(c3p_launch_grad_scan_general(G, false, st));
  MidGradArgs g = {};
  g.tables = tabs;
  g.tables_h = tabs + tdoubles;
  g.tab_per_sample = a.tab_per_sample;
  g.signals = signals;
  g.Mb = G.Mb;
  g.pre = pre;
  g.dUs = dUs;
  g.pstore = pstore;
  g.grad = grad;
  g.B = B;
  g.K = K;
  g.N = N;
  g.Dm = Dm;
  g.S = (int)S;
  g.Lmax = a.Lmax;
  LAUNCH_TRY(c3p_launch_midd_grad_general(g, st));
  return 0;
}

// Gradient for supplied per-slice generators X_n = coef hs[b,n] (branch B of pwc: the result is the cotangent of every X_n) on the
// on-chip general-generator sweeps (D <= 40): hmeta pre-pass, forward chain kernel in its supplied-generator mode (segment
// products + slice propagators), general scan, then the small-D / mid-D sweep kernel in its supplied-generator mode.
// Nothing is assumed about the generators.  Returns 1 when not applicable.
int run_vjp_xg_general(DeviceWs* w, const cplx* hs, long hs_bstride, double coef_r, double coef_i, int B, int N, int D,
                       const double* fr_phase, const cplx* Ubar, cplx* zout, hipStream_t st) {
  const bool small = D <= kSmallDLimit && c3p_smalld_supported(D);
  int nig = 0, nj = 0, wd = 0;
  if (!small && !c3p_midd_geometry(D, &nig, &nj, &wd)) return 1;
  long S;
  if (small) {
    S = pick_segments(B, N, 0, D, false, 4096);
    if (S < 0) return 1;
  } else {
    S = ((D <= 32 ? 1024 : 512) + B - 1) / B;
    const long smax = N / 8 > 1 ? N / 8 : 1;
    if (S > smax) S = smax;
    if (S < 1) S = 1;
    if (c3p_midd_grad_image_bytes(D) > (size_t)150 * 1024) return 1;
  }
  void *mv, *sv, *bv, *av;
  if (ws_get(w, SL_TABLES, (size_t)B * N * 4 * sizeof(double), &mv)) return -1;
  const size_t msz = (size_t)D * D * sizeof(cplx);
  if (ws_get(w, SL_SEG_A, (size_t)B * S * msz, &sv)) return -1;
  if (ws_get(w, SL_SEG_B, (size_t)B * S * msz, &av)) return -1;
  if (ws_get(w, SL_OUT1, ((size_t)B * S + 2 * (size_t)B * N) * msz, &bv)) return -1;  // (SL_OUT0 stages gen_bar_out)
  cplx* pre = (cplx*)bv;
  cplx* dUs = pre + (size_t)B * S * D * D;
  cplx* pstore = dUs + (size_t)B * N * D * D;
  LAUNCH_TRY(c3p_launch_hmeta(hs, hs_bstride, (long)B * N, N, D, coef_r, coef_i, (double*)mv, st));
  const int Lmax = (int)((N + S - 1) / S);
  if (small) {
    SmallArgs a = {};
    a.no_t18n = c3p_opt_on(C3P_OPT_no_t18n) ? (int)c3p_opt(C3P_OPT_no_t18n) : 0;
    a.hs = hs;
    a.hs_bstride = hs_bstride;
    a.meta = (const double*)mv;
    a.coef_r = coef_r;
    a.coef_i = coef_i;
    a.B = B;
    a.N = N;
    a.Dm = D;
    a.S = (int)S;
    a.Lmax = Lmax;
    a.mode = C3P_MODE_EXPM;
    a.seg_out = (cplx*)sv;
    a.dUs_out = dUs;
    LAUNCH_TRY(c3p_launch_smalld_chain(a, st));
  } else {
    MidArgs a = {};
    a.hs = hs;
    a.hs_bstride = hs_bstride;
    a.meta = (const double*)mv;
    a.coef_r = coef_r;
    a.coef_i = coef_i;
    a.B = B;
    a.N = N;
    a.Dm = D;
    a.S = (int)S;
    a.Lmax = Lmax;
    a.mode = C3P_MODE_EXPM;
    a.seg_out = (cplx*)sv;
    a.dUs_out = dUs;
    a.no_t18 = c3p_opt_on(C3P_OPT_no_t18) ? 1 : 0;
    a.no_t18n = c3p_opt_on(C3P_OPT_no_t18n) ? (int)c3p_opt(C3P_OPT_no_t18n) : 0;
    a.no_real = 1;
    LAUNCH_TRY(c3p_launch_midd_chain(a, st));
  }
  GradArgs G = {};
  G.Ubar = Ubar;
  G.fr_phase = fr_phase;
  G.B = B;
  G.N = N;
  G.D = D;
  G.ld = D | 1;
  G.S = (int)S;
  G.seg = (cplx*)sv;
  G.Mb = (cplx*)av;
  G.pre = pre;
  G.general = 1;
  LAUNCH_TRY(c3p_launch_grad_scan_general(G, false, st));
  if (small) {
    SmallGradArgs g = {};
    g.Mb = G.Mb;
    g.pre = pre;
    g.dUs = dUs;
    g.pstore = pstore;
    g.zout = zout;
    g.hs = hs;
    g.hs_bstride = hs_bstride;
    g.meta = (const double*)mv;
    g.coef_r = coef_r;
    g.coef_i = coef_i;
    g.B = B;
    g.N = N;
    g.Dm = D;
    g.S = (int)S;
    g.Lmax = Lmax;
    LAUNCH_TRY(c3p_launch_smalld_grad_general(g, st));
  } else {
    MidGradArgs g = {};
    g.Mb = G.Mb;
    g.pre = pre;
    g.dUs = dUs;
    g.pstore = pstore;
    g.zout = zout;
    g.hs = hs;
    g.hs_bstride = hs_bstride;
    g.meta = (const double*)mv;
    g.coef_r = coef_r;
    g.coef_i = coef_i;
    g.B = B;
    g.N = N;
    g.Dm = D;
    g.S = (int)S;
    g.Lmax = Lmax;
    LAUNCH_TRY(c3p_launch_midd_grad_general(g, st));
  }
  return 0;
}

// ---------------------------------------------------------------------------
// Big-D MFMA path (81 x 81 Lindblad superoperators): 8-wave workgroup per chain, global arena
// ---------------------------------------------------------------------------
int run_pwc_bigd(DeviceWs* w, int lindblad, const cplx* h0, long h0_bs, const cplx* hks, long hk_bs,
                 const double* signals, const cplx* clp, double dt, int B, int K, int N, int D, int Dm,
                 const double* fr_phase, cplx* U_out, cplx* dUs_out, hipStream_t st) {
  int nig, nj, wd;
  if (!c3p_bigd_geometry(Dm, &nig, &nj, &wd)) return 1;
  const bool per_sample = (h0_bs != 0) || (hk_bs != 0);
  long S = (C3P_BIGD_MAX_WGS + B - 1) / B;
  const long smax = N / 8 > 1 ? N / 8 : 1;
  if (S > smax) S = smax;
  if (S < 1) S = 1;
  while (c3p_bigd_lds_bytes(Dm, K, (int)((N + S - 1) / S)) > 150 * 1024 && S < N) ++S;
  if (c3p_bigd_lds_bytes(Dm, K, (int)((N + S - 1) / S)) > 150 * 1024) return 1;
  const int nsamp = per_sample ? B : 1;
  void* v;
  if (ws_get(w, SL_TABLES, (size_t)nsamp * c3p_bigd_table_doubles(Dm, K) * sizeof(double), &v)) return -1;
  MidPrepArgs p = {};
  p.h0 = h0;
  p.h0_bstride = h0_bs;
  p.hks = hks;
  p.hks_bstride = hk_bs;
  p.clp = clp;
  p.dt = dt;
  p.K = K;
  p.Dh = D;
  p.Dm = Dm;
  p.lindblad = lindblad;
  p.rows = 16 * nig;
  p.W = wd;
  p.tile_nig = nig;
  p.tile_nj = nj;
  p.tables = (double*)v;
  LAUNCH_TRY(c3p_launch_midd_prep(p, nsamp, st));
  void* av;
  if (ws_get(w, SL_SCRATCH, c3p_bigd_arena_doubles(Dm) * sizeof(double), &av)) return -1;
  MidArgs a = {};
  a.tables = (const double*)v;
  a.tab_per_sample = per_sample ? 1 : 0;
  a.signals = signals;
  a.B = B;
  a.K = K;
  a.N = N;
  a.Dm = Dm;
  a.S = (int)S;
  a.Lmax = (int)((N + S - 1) / S);
  a.mode = lindblad ? C3P_MODE_LINDBLAD : C3P_MODE_UNITARY;
  a.dUs_out = dUs_out;
  cplx* seg = U_out;
  if (S > 1) {
    void* sv;
    if (ws_get(w, SL_SEG_A, (size_t)B * S * Dm * Dm * sizeof(cplx), &sv)) return -1;
    seg = (cplx*)sv;
  }
  a.seg_out = seg;
  g_last_kernel = C3P_KERNEL_MFMA;
  if (record_start(w, st)) return -1;
  LAUNCH_TRY(c3p_launch_bigd_chain(a, (double*)av, st));
  if (record_stop(w, st)) return -1;
  if (S == 1 && fr_phase) LAUNCH_TRY(c3p_launch_rowphase(U_out, fr_phase, B, Dm, st));
  if (S > 1) {
    // ordered combine of the few segment products with the generic kernel (GIVEN mode)
    ChainArgs c = {};
    c.mode = C3P_MODE_GIVEN;
    c.mats = seg;
    c.B = B;
    c.N = (int)S;
    c.D = D;
    c.Dm = Dm;
    c.fr_phase = fr_phase;
    const int keep = g_last_kernel;
    const int rc = run_chain_generic(w, c, U_out, st, /*profile=*/false);  // the event pair stays on the main kernel
    g_last_kernel = keep;
    if (rc) return -1;
  }
  return 0;
}

// ---------------------------------------------------------------------------
// Register-resident MFMA path (Dm = 49, 65, 81; the 81 x 81 Lindblad superoperators of cfg4): 4-wave workgroup
// per chain, right operands / accumulators in registers, three-real-product complex arithmetic (c3p_regd.hip)
// ---------------------------------------------------------------------------
int run_pwc_regd(DeviceWs* w, int lindblad, const cplx* h0, long h0_bs, const cplx* hks, long hk_bs,
                 const double* signals, const cplx* clp, double dt, int B, int K, int N, int D, int Dm,
                 const double* fr_phase, cplx* U_out, cplx* dUs_out, hipStream_t st) {
  if (!c3p_regd_supported(Dm) || K > 16) return 1;
  const bool per_sample = (h0_bs != 0) || (hk_bs != 0);
  long S = (C3P_REGD_MAX_WGS + B - 1) / B;
  const long smax = N / 8 > 1 ? N / 8 : 1;
  if (S > smax) S = smax;
  if (S < 1) S = 1;
  const int nsamp = per_sample ? B : 1;
  // Lindblad superoperators of Hermitian Hamiltonians: the whole chain in real arithmetic in the Hermitian basis
  // (c3p_regr.hip); the complex kernel keeps the samples whose tables are not real there
  const bool hb = lindblad && c3p_regr_supported(D, Dm) && !c3p_opt_on(C3P_OPT_no_hermitian_basis);
  const size_t ctab = (((size_t)nsamp * c3p_regd_table_doubles(Dm, K) * sizeof(double)) + 255) & ~(size_t)255;
  const size_t rtab = hb ? ((((size_t)nsamp * c3p_regr_table_doubles(Dm, K) * sizeof(double)) + 255) & ~(size_t)255) : 0;
  const size_t ftab = hb ? (size_t)nsamp * (1 + K) * sizeof(int) : 0;
  void* v;
  if (ws_get(w, SL_TABLES, ctab + rtab + ftab, &v)) return -1;
  RegdPrepArgs p = {};
  p.h0 = h0;
  p.h0_bstride = h0_bs;
  p.hks = hks;
  p.hks_bstride = hk_bs;
  p.clp = clp;
  p.dt = dt;
  p.K = K;
  p.Dh = D;
  p.Dm = Dm;
  p.lindblad = lindblad;
  p.tables = (double*)v;
  LAUNCH_TRY(c3p_launch_regd_prep(p, nsamp, st));
  double* rtables = reinterpret_cast<double*>(static_cast<char*>(v) + ctab);
  int* tabflag = reinterpret_cast<int*>(static_cast<char*>(v) + ctab + rtab);
  if (hb) LAUNCH_TRY(c3p_launch_regr_prep(p, nsamp, rtables, tabflag, st));
  void* av;
  if (ws_get(w, SL_SCRATCH, std::max(c3p_regd_arena_bytes(Dm), hb ? c3p_regr_arena_bytes(Dm) : (size_t)0), &av)) return -1;
  MidArgs a = {};
  a.tables = (const double*)v;
  a.tab_per_sample = per_sample ? 1 : 0;
  a.signals = signals;
  a.B = B;
  a.K = K;
  a.N = N;
  a.Dm = Dm;
  a.S = (int)S;
  a.Lmax = (int)((N + S - 1) / S);
  a.mode = lindblad ? C3P_MODE_LINDBLAD : C3P_MODE_UNITARY;
  a.dUs_out = dUs_out;
  if (hb) {
    a.hb_tables = rtables;
    a.hb_tabflag = tabflag;
  }
  a.no_t18n = c3p_opt_on(C3P_OPT_no_t18n) ? (int)c3p_opt(C3P_OPT_no_t18n) : 0;
  cplx* seg = U_out;
  if (S > 1) {
    void* sv;
    if (ws_get(w, SL_SEG_A, (size_t)B * S * Dm * Dm * sizeof(cplx), &sv)) return -1;
    seg = (cplx*)sv;
  }
  a.seg_out = seg;
  g_last_kernel = C3P_KERNEL_MFMA;
  if (record_start(w, st)) return -1;
  if (hb) LAUNCH_TRY(c3p_launch_regr_chain(a, av, st));
  LAUNCH_TRY(c3p_launch_regd_chain(a, av, st));
  if (record_stop(w, st)) return -1;
  if (hb) {
    // back to the reference's vectorisation, in place (real results sit in the second half of their complex slots)
    LAUNCH_TRY(c3p_launch_hb_to_complex(seg, (long)B * S, (int)S, tabflag, per_sample ? 1 : 0, K, D, st));
    if (dUs_out) LAUNCH_TRY(c3p_launch_hb_to_complex(dUs_out, (long)B * N, N, tabflag, per_sample ? 1 : 0, K, D, st));
  }
  if (S == 1 && fr_phase) LAUNCH_TRY(c3p_launch_rowphase(U_out, fr_phase, B, Dm, st));
  if (S > 1) {
    // ordered combine of the few segment products with the generic kernel (GIVEN mode)
    ChainArgs c = {};
    c.mode = C3P_MODE_GIVEN;
    c.mats = seg;
    c.B = B;
    c.N = (int)S;
    c.D = D;
    c.Dm = Dm;
    c.fr_phase = fr_phase;
    const int keep = g_last_kernel;
    const int rc = run_chain_generic(w, c, U_out, st, /*profile=*/false);  // the event pair stays on the main kernel
    g_last_kernel = keep;
    if (rc) return -1;
  }
  return 0;
}

// ---------------------------------------------------------------------------
// Tiled large-matrix path (c3p_tiled.hip): matrices in HBM, one batched MFMA GEMM launch per product
// ---------------------------------------------------------------------------
int run_pwc_tiled(DeviceWs* w, const ChainArgs& a, bool per_slice, cplx* U_out, hipStream_t st) {
  const bool lindblad = a.mode == C3P_MODE_LINDBLAD;
  const bool per_sample = !per_slice && (a.h0_bstride != 0 || a.hks_bstride != 0);
  const int K = per_slice ? 0 : a.K;
  const int Bc = c3p_tiled_chunk(a.Dm, K, a.B, per_sample, (size_t)24 << 30);
  void* v;
  if (ws_get(w, SL_SCRATCH, c3p_tiled_ws_bytes(a.Dm, K, Bc, per_sample), &v)) return -1;
  TiledArgs t = {};
  t.lindblad = lindblad ? 1 : 0;
  t.per_slice = per_slice ? 1 : 0;
  t.h0 = a.h0;
  t.h0_bstride = a.h0_bstride;
  t.hks = a.hks;
  t.hks_bstride = a.hks_bstride;
  t.signals = a.signals;
  t.clp = a.clp;
  t.dt = a.dt;
  t.B = a.B;
  t.K = K;
  t.N = a.N;
  t.D = a.D;
  t.Dm = a.Dm;
  t.fr_phase = a.fr_phase;
  t.U_out = U_out;
  t.dUs_out = a.dUs_out;
  g_last_kernel = C3P_KERNEL_MFMA;
  if (g_dry) return 0;  // c3p_reserve: the arena above is all this path owns
  std::string err;
  // many launches per call: the timing pair brackets the whole host-driven slice loop
  if (record_start(w, st)) return -1;
  if (c3p_tiled_run(t, v, Bc, st, err)) return fail("%s", err.c_str());
  if (record_stop(w, st)) return -1;
  return 0;
}

// Tiled backward sweep (c3p_tiled.hip): any matrix dimension, unitary and Lindblad generators
// Budget (bytes) for the per-sample forward intermediates a backward sweep keeps in HBM: 24 GB, or 60 % of what the device
// has free plus what the workspace slot that will hold them already owns (a caching allocator of the host framework may hold
// most of the 288 GB; a smaller part has less) -- the sweeps run in chunks of samples below it.
static size_t grad_store_budget(DeviceWs* w, Slot slot) {
  size_t budget = (size_t)24 << 30;
  size_t fr = 0, tot = 0;
  if (hipMemGetInfo(&fr, &tot) == hipSuccess) {
    const size_t avail = (size_t)(0.6 * (double)fr) + (w ? w->cap[slot] : 0);
    if (avail < budget) budget = avail;
  }
  if (budget < ((size_t)64 << 20)) budget = (size_t)64 << 20;
  return budget;
}

// ---------------------------------------------------------------------------
// Lindblad control gradient at D = 7, 8, 9 (49 x 49 .. 81 x 81 superoperators, cfg4) in the Hermitian basis: forward chain
// kernel with the transposed local prefixes kept in HBM, real segment scan, on-chip backward sweep (c3p_regrg.hip).
// The forward half returns 1 when a Hamiltonian is not Hermitian (complex tables in that basis): the caller falls back.
// ---------------------------------------------------------------------------
// D = 7, 8, 9 in their own classes.  D = 6 (36 x 36) zero padded in the 49 class is an A/B switch (regr_grad_d6 = 1): measured
// SLOWER than the complex mid-D general sweep -- 13.4 against 10.3 ms at B = 64, N = 500, 105 against 80 ms at B = 256, N = 1000
// (profiles/r04/grad_lindblad_d6_*.json): (49 / 36)^3 = 2.5x padded work eats the factor of real arithmetic
static bool lind_regr_grad_ok(int D, int Dm) {
  if (c3p_regr_supported(D, Dm)) return true;
  return Dm == D * D && D == 6 && c3p_opt_on(C3P_OPT_regr_grad_d6);
}
struct LindRegrBufs {  // device buffers of one forward pass (the library's workspace, or a caller-owned tape)
  double *tab_f, *tab_t;
  int *flag_f, *flag_t;
  cplx* seg;   // [B,S] complex slots, the real matrix in the second half of each
  double* qT;  // [B,N,Dm,Dm]
};
static long lind_regr_segments(int B, int N) {
  // fill the 256 workgroup slots evenly (rounds of 256 chains), few segments
  long S = 1;
  const long smax = N / 4 > 1 ? N / 4 : 1;
  double best = -1.0;
  for (long s = 1; s <= 32 && s <= smax; ++s) {
    const long chains = (long)B * s, rounds = (chains + C3P_REGD_MAX_WGS - 1) / C3P_REGD_MAX_WGS;
    const double eff = (double)chains / (double)(rounds * C3P_REGD_MAX_WGS) - 0.004 * (double)s;
    if (eff > best + 1e-12) best = eff, S = s;
  }
  if (c3p_opt(C3P_OPT_segments) > 0) S = std::min<long>(c3p_opt(C3P_OPT_segments), smax);
  return S;
}
struct LindRegrSizes {
  size_t tab1, ftab, seg, qT;
  size_t total() const { return 2 * tab1 + 2 * ftab + seg + qT; }
};
static LindRegrSizes lind_regr_sizes(int B, int K, int N, int Dm, long S, int nsamp) {
  auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
  LindRegrSizes z;
  z.tab1 = up((size_t)nsamp * c3p_regr_table_doubles(Dm, K) * sizeof(double));
  z.ftab = up((size_t)nsamp * (1 + K) * sizeof(int));
  z.seg = up((size_t)B * S * Dm * Dm * sizeof(cplx));
  z.qT = up((size_t)B * N * Dm * Dm * sizeof(double));
  return z;
}
static LindRegrBufs lind_regr_carve(void* base, const LindRegrSizes& z) {
  char* c = static_cast<char*>(base);
  LindRegrBufs b;
  b.tab_f = reinterpret_cast<double*>(c);
  b.tab_t = reinterpret_cast<double*>(c + z.tab1);
  b.flag_f = reinterpret_cast<int*>(c + 2 * z.tab1);
  b.flag_t = reinterpret_cast<int*>(c + 2 * z.tab1 + z.ftab);
  b.seg = reinterpret_cast<cplx*>(c + 2 * z.tab1 + 2 * z.ftab);
  b.qT = reinterpret_cast<double*>(c + 2 * z.tab1 + 2 * z.ftab + z.seg);
  return b;
}
// tables of G' and G'^T, Hermiticity flags (read back: one synchronisation per call), forward chain with Q^T
int lind_regr_forward(DeviceWs* w, const LindRegrBufs& bf, const cplx* h0, long h0_bs, const cplx* hks, long hk_bs, const double* signals,
                      const cplx* clp, double dt, int B, int K, int N, int D, int Dm, long S, hipStream_t st) {
  const bool per_sample = (h0_bs != 0) || (hk_bs != 0);
  const int nsamp = per_sample ? B : 1;
  RegdPrepArgs p = {};
  p.h0 = h0;
  p.h0_bstride = h0_bs;
  p.hks = hks;
  p.hks_bstride = hk_bs;
  p.clp = clp;
  p.dt = dt;
  p.K = K;
  p.Dh = D;
  p.Dm = Dm;
  p.lindblad = 1;
  LAUNCH_TRY(c3p_launch_regr_prep_t(p, nsamp, bf.tab_f, bf.flag_f, 0, st));
  LAUNCH_TRY(c3p_launch_regr_prep_t(p, nsamp, bf.tab_t, bf.flag_t, 1, st));
  {
    std::vector<int> hf((size_t)nsamp * (1 + K));
    HIP_TRY(hipMemcpyAsync(hf.data(), bf.flag_f, hf.size() * sizeof(int), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    for (int f : hf)
      if (!f) return 1;  // a non-Hermitian Hamiltonian: complex generator in the Hermitian basis
  }
  void* av;
  if (ws_get(w, SL_SCRATCH, std::max(c3p_regr_grad_arena_bytes(Dm), c3p_regr_arena_bytes(Dm)), &av)) return -1;
  MidArgs a = {};
  a.tab_per_sample = per_sample ? 1 : 0;
  a.signals = signals;
  a.B = B;
  a.K = K;
  a.N = N;
  a.Dm = Dm;
  a.S = (int)S;
  a.Lmax = (int)((N + S - 1) / S);
  a.mode = C3P_MODE_LINDBLAD;
  a.hb_tables = bf.tab_f;
  a.hb_tabflag = bf.flag_f;
  a.hb_qT = bf.qT;
  a.no_t18n = c3p_opt_on(C3P_OPT_no_t18n) ? (int)c3p_opt(C3P_OPT_no_t18n) : 0;
  a.seg_out = bf.seg;
  LAUNCH_TRY(c3p_launch_regr_chain(a, av, st));
  return 0;
}
// cotangent in the Hermitian basis, segment scan, backward sweep (scratch from the workspace)
int lind_regr_backward(DeviceWs* w, const LindRegrBufs& bf, bool per_sample, const double* signals, int B, int K, int N, int D, int Dm,
                       long S, const double* fr_phase, const cplx* Ubar, double* grad, hipStream_t st) {
  const size_t msz = (size_t)Dm * Dm;
  void *bv, *av;
  if (ws_get(w, SL_SEG_B, ((size_t)3 * B * S + B) * msz * sizeof(double) + (size_t)B * sizeof(double), &bv)) return -1;
  if (ws_get(w, SL_SCRATCH, std::max(c3p_regr_grad_arena_bytes(Dm), c3p_regr_arena_bytes(Dm)), &av)) return -1;
  double* pre = (double*)bv;
  double* suf = pre + (size_t)B * S * msz;
  double* lam = suf + (size_t)B * S * msz;
  double* ubr = lam + (size_t)B * S * msz;
  double* tau = ubr + (size_t)B * msz;
  LAUNCH_TRY(c3p_launch_hb_ubar(Ubar, fr_phase, B, D, ubr, st));
  LAUNCH_TRY(c3p_launch_regr_scan(bf.seg, ubr, B, (int)S, Dm, pre, suf, lam, tau, st));
  RegrGradArgs g = {};
  g.tables = bf.tab_f;
  g.tables_t = bf.tab_t;
  g.tab_per_sample = per_sample ? 1 : 0;
  g.signals = signals;
  g.qT = bf.qT;
  g.lam = lam;
  g.tau = tau;
  g.grad = grad;
  g.arena = (double*)av;
  g.B = B;
  g.K = K;
  g.N = N;
  g.Dm = Dm;
  g.S = (int)S;
  g.degree = c3p_opt(C3P_OPT_regr_grad_degree) > 0 ? (int)c3p_opt(C3P_OPT_regr_grad_degree) : 0;
  LAUNCH_TRY(c3p_launch_regr_grad(g, st));
  return 0;
}
int run_vjp_lind_regr(DeviceWs* w, const cplx* h0, long h0_bs, const cplx* hks, long hk_bs, const double* signals, const cplx* clp,
                      double dt, int B, int K, int N, int D, int Dm, const double* fr_phase, const cplx* Ubar, double* grad,
                      hipStream_t st) {
  if (!lind_regr_grad_ok(D, Dm) || K > 16 || K < 1) return 1;
  const bool per_sample = (h0_bs != 0) || (hk_bs != 0);
  const long S = lind_regr_segments(B, N);
  const LindRegrSizes z = lind_regr_sizes(B, K, N, Dm, S, per_sample ? B : 1);
  void *tv, *sv, *qv;
  if (ws_get(w, SL_TABLES, 2 * z.tab1 + 2 * z.ftab, &tv)) return -1;
  if (ws_get(w, SL_SEG_A, z.seg, &sv)) return -1;
  if (ws_get(w, SL_OUT1, z.qT, &qv)) return -1;
  LindRegrBufs bf = lind_regr_carve(tv, z);
  bf.seg = (cplx*)sv;
  bf.qT = (double*)qv;
  const int rc = lind_regr_forward(w, bf, h0, h0_bs, hks, hk_bs, signals, clp, dt, B, K, N, D, Dm, S, st);
  if (rc != 0) return rc;
  return lind_regr_backward(w, bf, per_sample, signals, B, K, N, D, Dm, S, fr_phase, Ubar, grad, st);
}

int run_vjp_tiled(DeviceWs* w, int lindblad, const cplx* h0, long h0_bs, const cplx* hks, long hk_bs, const double* signals,
                  const cplx* clp, double dt, int B, int K, int N, int D, int Dm, const double* fr_phase, const cplx* U_bar,
                  double* grad, hipStream_t st, bool per_slice = false, cplx* zout = nullptr) {
  const bool per_sample = !per_slice && ((h0_bs != 0) || (hk_bs != 0));
  const int Bc = c3p_tiled_vjp_chunk(Dm, K, N, B, per_sample, grad_store_budget(w, SL_SCRATCH));
  void* v;
  if (ws_get(w, SL_SCRATCH, c3p_tiled_vjp_ws_bytes(Dm, K, N, Bc, per_sample), &v)) return -1;
  if (g_dry) return 0;
  TiledArgs t = {};
  t.lindblad = lindblad;
  t.per_slice = per_slice ? 1 : 0;
  t.h0 = h0;
  t.h0_bstride = h0_bs;
  t.hks = hks;
  t.hks_bstride = hk_bs;
  t.signals = signals;
  t.clp = clp;
  t.dt = dt;
  t.B = B;
  t.K = K;
  t.N = N;
  t.D = D;
  t.Dm = Dm;
  t.fr_phase = fr_phase;
  // the per-slice launch sequences are replayed as hipGraphs, which the legacy default stream cannot capture
  hipStream_t run = st;
  if (!run) {
    if (!w->aux) HIP_TRY(hipStreamCreate(&w->aux));
    run = w->aux;
  }
  std::string err;
  if (c3p_tiled_vjp_run(t, U_bar, grad, zout, v, Bc, run, err)) return fail("%s", err.c_str());
  return 0;
}

// Host-pointer staging helpers --------------------------------------------------
struct Stage {
  DeviceWs* w;
  hipStream_t st;
  int in_slot = SL_IN0;
  int out_slot = SL_OUT0;
  struct Back {
    void* host;
    void* dev;
    size_t bytes;
  };
  std::vector<Back> backs;
  int in(const void* host, size_t bytes, const void** dev) {
    if (!host || bytes == 0) {
      *dev = nullptr;
      return 0;
    }
    void* d;
    if (in_slot > SL_IN5) return fail("internal: out of input staging slots");
    if (ws_get(w, (Slot)in_slot++, bytes, &d)) return -1;
    HIP_TRY(hipMemcpyAsync(d, host, bytes, hipMemcpyHostToDevice, st));
    *dev = d;
    return 0;
  }
  int out(void* host, size_t bytes, void** dev) {
    if (!host || bytes == 0) {
      *dev = nullptr;
      return 0;
    }
    void* d;
    if (out_slot > SL_OUT3) return fail("internal: out of output staging slots");
    if (ws_get(w, (Slot)out_slot++, bytes, &d)) return -1;
    backs.push_back({host, d, bytes});
    *dev = d;
    return 0;
  }
  int finish() {
    for (auto& b : backs) HIP_TRY(hipMemcpyAsync(b.host, b.dev, b.bytes, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return 0;
  }
};

int pwc_common(int lindblad, const void* h0, int64_t h0_bstride, const void* hks, int64_t hks_bstride,
               const double* signals, const void* col_ops, int C, double dt, int B, int K, int N, int D,
               int flags, const double* fr_phase, void* U_out, void* dUs_out, void* stream) {
  if (B < 0 || N < 0 || D <= 0 || K < 0) return fail("bad sizes B=%d K=%d N=%d D=%d", B, K, N, D);
  if (!U_out) return fail("U_out is NULL");
  if (B == 0) return 0;
  if (N == 0) return fail("empty time grid (N == 0)");
  if (!h0) return fail("h0 is NULL");
  if (K > 0 && (!hks || !signals)) return fail("K > 0 but hks/signals missing");
  if (lindblad && (!col_ops || C <= 0)) return fail("lindblad propagation needs col_ops");
  const bool per_slice = (flags & C3P_PER_SLICE_H) != 0;
  const int Dm = lindblad ? D * D : D;
  const size_t cs = sizeof(cplx);
  hipStream_t st = (hipStream_t)stream;
  WsLock lk(st);
  DeviceWs* w = lk.w;
  if (!w) return fail("no HIP device");
  if (!lk.ok) return fail("hipStreamWaitEvent on the previous call's stream failed");
  Stage sg{w, st};
  const void *d_h0 = h0, *d_hks = hks, *d_sig = signals, *d_col = col_ops, *d_ph = fr_phase;
  void *d_U = U_out, *d_dUs = dUs_out;
  if (flags & C3P_HOST_PTRS) {
    const size_t h0_one = (size_t)(per_slice ? N : 1) * D * D;
    const size_t h0_elems = h0_bstride ? (size_t)(B - 1) * h0_bstride + h0_one : h0_one;
    const size_t hk_one = (size_t)K * D * D;
    const size_t hk_elems = hks_bstride ? (size_t)(B - 1) * hks_bstride + hk_one : hk_one;
    if (sg.in(h0, h0_elems * cs, &d_h0)) return -1;
    if (sg.in(hks, K ? hk_elems * cs : 0, &d_hks)) return -1;
    if (sg.in(signals, (size_t)B * K * N * sizeof(double), &d_sig)) return -1;
    if (sg.in(col_ops, lindblad ? (size_t)C * D * D * cs : 0, &d_col)) return -1;
    if (sg.in(fr_phase, fr_phase ? (size_t)B * Dm * sizeof(double) : 0, &d_ph)) return -1;
    if (sg.out(U_out, (size_t)B * Dm * Dm * cs, &d_U)) return -1;
    if (sg.out(dUs_out, dUs_out ? (size_t)B * N * Dm * Dm * cs : 0, &d_dUs)) return -1;
  }
  ChainArgs a = {};
  a.h0 = (const cplx*)d_h0;
  a.h0_bstride = h0_bstride;
  a.h0_nstride = per_slice ? (long)D * D : 0;
  a.hks = (const cplx*)d_hks;
  a.hks_bstride = hks_bstride;
  a.signals = (const double*)d_sig;
  a.fr_phase = (const double*)d_ph;
  a.dt = dt;
  a.B = B;
  a.K = K;
  a.N = N;
  a.D = D;
  a.Dm = Dm;
  a.mode = lindblad ? C3P_MODE_LINDBLAD : C3P_MODE_UNITARY;
  a.dUs_out = (cplx*)d_dUs;
  if (lindblad) {
    void* v;
    if (ws_get(w, SL_CLP, (size_t)Dm * Dm * cs, &v)) return -1;
    LAUNCH_TRY(c3p_launch_clp((const cplx*)d_col, C, D, (cplx*)v, st));
    a.clp = (const cplx*)v;
  }
  bool done = false;
  if (!(flags & C3P_FORCE_GENERIC) && per_slice && !lindblad && K == 0 && D <= kSmallDLimit && c3p_smalld_supported(D)) {
    // branch B (propagation.py:295-308): X_n = -i dt H_n
    const int rc = run_xg_smalld(w, a.h0, a.h0_bstride, 0.0, -dt, B, N, D, a.fr_phase, (cplx*)d_U, a.dUs_out, st);
    if (rc < 0) return -1;
    done = (rc == 0);
  }
  if (!done && !(flags & C3P_FORCE_GENERIC) && per_slice && !lindblad && K == 0 && D >= 13 && D <= 40) {
    const int rc = run_xg_midd(w, a.h0, a.h0_bstride, 0.0, -dt, B, N, D, a.fr_phase, (cplx*)d_U, a.dUs_out, st);
    if (rc < 0) return -1;
    done = (rc == 0);
  }
  if (!done && !(flags & C3P_FORCE_GENERIC) && per_slice && lindblad && K == 0 && Dm <= 40 &&
      (size_t)B * N * Dm * Dm * cs <= ((size_t)16 << 30)) {
    // branch B with model.lindbladian (propagation.py:295-308 into :551-585): the dense superoperator generator of every slice,
    // then the supplied-generator mode of the matrix-core chain kernels with X_n = dt L_n
    void* gv;
    if (ws_get(w, SL_SCRATCH, (size_t)B * N * Dm * Dm * cs, &gv)) return -1;
    LAUNCH_TRY(c3p_launch_lind_slice_generators(a.h0, a.h0_bstride, a.clp, B, N, D, (cplx*)gv, st));
    const long gbs = (long)N * Dm * Dm;
    const int rc = (Dm <= kSmallDLimit && c3p_smalld_supported(Dm))
                       ? run_xg_smalld(w, (const cplx*)gv, gbs, dt, 0.0, B, N, Dm, a.fr_phase, (cplx*)d_U, a.dUs_out, st)
                       : run_xg_midd(w, (const cplx*)gv, gbs, dt, 0.0, B, N, Dm, a.fr_phase, (cplx*)d_U, a.dUs_out, st);
    if (rc < 0) return -1;
    done = (rc == 0);
  }
  if (!done && !(flags & C3P_FORCE_GENERIC) && !per_slice && lindblad && (flags & C3P_HERMITIAN_H) && !a.dUs_out &&
      c3p_smallr_supported(D, Dm, K) && !c3p_opt_on(C3P_OPT_no_smallr)) {
    const int rc = run_pwc_smallr(w, a.h0, a.h0_bstride, a.hks, a.hks_bstride, a.signals, a.clp, dt, B, K, N, D, Dm, a.fr_phase, (cplx*)d_U, st);
    if (rc < 0) return -1;
    done = (rc == 0);
  }
  if (!done && !(flags & C3P_FORCE_GENERIC) && !per_slice && Dm <= kSmallDLimit && c3p_smalld_supported(Dm) && K <= 8) {
    const int rc = run_pwc_smalld(w, lindblad, a.h0, a.h0_bstride, a.hks, a.hks_bstride, a.signals, a.clp, dt,
                                  B, K, N, D, Dm, a.fr_phase, (cplx*)d_U, a.dUs_out, st);
    if (rc < 0) return -1;
    done = (rc == 0);
  }
  if (!done && !(flags & C3P_FORCE_GENERIC) && !per_slice && Dm >= 13 && Dm <= 40 && K <= 16) {
    const int rc = run_pwc_midd(w, lindblad, a.h0, a.h0_bstride, a.hks, a.hks_bstride, a.signals, a.clp, dt, B,
                                K, N, D, Dm, a.fr_phase, (cplx*)d_U, a.dUs_out, st);
    if (rc < 0) return -1;
    done = (rc == 0);
  }
  if (!done && !(flags & C3P_FORCE_GENERIC) && !per_slice && c3p_regd_supported(Dm) && K <= 16 && !c3p_opt_on(C3P_OPT_no_regd)) {
    const int rc = run_pwc_regd(w, lindblad, a.h0, a.h0_bstride, a.hks, a.hks_bstride, a.signals, a.clp, dt, B,
                                K, N, D, Dm, a.fr_phase, (cplx*)d_U, a.dUs_out, st);
    if (rc < 0) return -1;
    done = (rc == 0);
  }
  if (!done && !(flags & C3P_FORCE_GENERIC) && !per_slice && K <= 16) {
    const int rc = run_pwc_bigd(w, lindblad, a.h0, a.h0_bstride, a.hks, a.hks_bstride, a.signals, a.clp, dt, B,
                                K, N, D, Dm, a.fr_phase, (cplx*)d_U, a.dUs_out, st);
    if (rc < 0) return -1;
    done = (rc == 0);
  }
  // beyond the on-chip kernels: Dm >= 93, and supplied / per-slice generators at Dm >= 41 (the generic kernel would run
  // those from global scratch at a few percent of the roofline, and stops at Dm = 256)
  if (!done && !(flags & C3P_FORCE_GENERIC) && !c3p_opt_on(C3P_OPT_no_tiled) && (Dm >= 93 || (per_slice && Dm >= 41)) &&
      (per_slice ? K == 0 : true)) {
    if (run_pwc_tiled(w, a, per_slice, (cplx*)d_U, st)) return -1;
    done = true;
  }
  if (!done && run_chain_generic(w, a, (cplx*)d_U, st)) return -1;
  if (flags & C3P_HOST_PTRS) return sg.finish();
  return 0;
}

}  // namespace

long c3p_opt(C3pOption o) {
  std::call_once(g_opt_once, opt_init);
  return g_opt_val[o].load(std::memory_order_relaxed);
}

void c3p_note_launch(const void* host_fn, const char* file) {
  for (int i = 0; i < g_nnotes; ++i)
    if (g_notes[i].fn == host_fn) {
      ++g_notes[i].count;
      return;
    }
  if (g_nnotes < (int)(sizeof(g_notes) / sizeof(g_notes[0]))) g_notes[g_nnotes++] = LaunchNote{host_fn, file, 1};
}

extern "C" {

int c3p_version(void) { return 1; }

int c3p_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return n;
}

const char* c3p_last_error(void) { return g_err.c_str(); }
int c3p_last_kernel(void) { return g_last_kernel; }

int c3p_last_kernel_detail(char* buf, int cap) {
  std::string out;
  for (int i = 0; i < g_nnotes; ++i) {
    const LaunchNote& n = g_notes[i];
    const char* raw = hipKernelNameRefByPtr(n.fn, nullptr);
    (void)hipGetLastError();
    std::string name = raw ? raw : "?";
    int status = 1;
    char* dm = abi::__cxa_demangle(name.c_str(), nullptr, nullptr, &status);
    if (status == 0 && dm) name = dm;
    free(dm);
    // "void (anonymous namespace)::kernel<...>(Args)" -> "kernel<...>"
    size_t p0 = name.rfind(')');
    if (p0 != std::string::npos) {  // drop the parameter list: the matching '(' of the LAST ')'
      int depth = 0;
      for (size_t q = p0 + 1; q-- > 0;) {
        if (name[q] == ')') ++depth;
        else if (name[q] == '(' && --depth == 0) {
          name.erase(q);
          break;
        }
      }
    }
    const std::string anon = "(anonymous namespace)::";
    for (size_t q; (q = name.find(anon)) != std::string::npos;) name.erase(q, anon.size());
    if (name.rfind("void ", 0) == 0) name.erase(0, 5);
    const char* base = strrchr(n.file, '/');
    if (!out.empty()) out += "; ";
    out += std::string(base ? base + 1 : n.file) + ": " + name;
    if (n.count > 1) out += " x" + std::to_string(n.count);
  }
  if (buf && cap > 0) {
    const size_t m = std::min(out.size(), (size_t)cap - 1);
    memcpy(buf, out.data(), m);
    buf[m] = 0;
  }
  return (int)out.size();
}

int c3p_set_option(const char* name, const char* value) {
  if (!name) return fail("c3p_set_option: null name");
  std::call_once(g_opt_once, opt_init);
  for (int i = 0; i < C3P_OPT_COUNT; ++i)
    if (!strcmp(name, g_opt_names[i])) {
      g_opt_val[i].store(value ? parse_opt_value(value) : -1, std::memory_order_relaxed);
      return 0;
    }
  return fail("c3p_set_option: unknown option '%s'", name);
}

long c3p_get_option(const char* name) {
  if (!name) return -2;
  std::call_once(g_opt_once, opt_init);
  for (int i = 0; i < C3P_OPT_COUNT; ++i)
    if (!strcmp(name, g_opt_names[i])) return g_opt_val[i].load(std::memory_order_relaxed);
  return -2;
}

int c3p_set_profiling(int enable) {
  DeviceWs* w = ws_for_current_device();
  if (!w) return fail("no HIP device");
  w->profiling = enable ? 1 : 0;
  return 0;
}

long c3p_workspace_generation(void) {
  DeviceWs* w = ws_for_current_device();
  return w ? w->generation.load() : -1;
}

int c3p_reserve(int lindblad, int B, int K, int N, int D, int C, int flags) {
  if (flags & C3P_HOST_PTRS) return fail("c3p_reserve sizes the device workspace: not for C3P_HOST_PTRS calls");
  // the planning code of the real call with launches switched off: every slot it would touch is allocated at the size
  // it would need, so later calls of this shape (and smaller ones) never allocate, free or synchronise
  static double dummy[4];
  void* p = dummy;  // never dereferenced: device pointers are only handed to kernels
  g_dry = true;
  const int rc = pwc_common(lindblad ? 1 : 0, p, 0, p, 0, (const double*)p, lindblad ? p : nullptr, lindblad ? C : 0, 1.0, B, K, N, D,
                            flags, nullptr, p, nullptr, nullptr);
  g_dry = false;
  return rc;
}

double c3p_last_kernel_ms(void) {
  WsLock lk(nullptr, false);
  DeviceWs* w = lk.w;
  if (!w || !w->ev_valid) return -1.0;
  float ms = 0.f;
  if (hipEventElapsedTime(&ms, w->ev0, w->ev1) != hipSuccess) {
    (void)hipGetLastError();
    return -1.0;
  }
  return (double)ms;
}

void c3p_shutdown(void) {
  std::vector<DeviceWs*> all;
  {
    std::lock_guard<std::mutex> lk(g_mu);
    for (auto& p : g_ws) all.push_back(p.get());
  }
  int cur = 0;
  (void)hipGetDevice(&cur);
  for (size_t d = 0; d < all.size(); ++d) {
    DeviceWs* w = all[d];
    if (!w) continue;
    std::lock_guard<std::mutex> lk(w->mu);
    bool any = w->ev0 != nullptr || w->done != nullptr || w->aux != nullptr;
    for (int s = 0; s < SL_COUNT; ++s) any = any || w->ptr[s];
    if (!any) continue;
    (void)hipSetDevice((int)d);
    (void)hipDeviceSynchronize();
    for (int s = 0; s < SL_COUNT; ++s) {
      if (w->ptr[s]) (void)hipFree(w->ptr[s]);
      w->ptr[s] = nullptr;
      w->cap[s] = 0;
    }
    if (w->ev0) {
      (void)hipEventDestroy(w->ev0);
      (void)hipEventDestroy(w->ev1);
      w->ev0 = w->ev1 = nullptr;
      w->ev_valid = false;
    }
    if (w->done) {
      (void)hipEventDestroy(w->done);
      w->done = nullptr;
    }
    if (w->aux) {
      (void)hipStreamDestroy(w->aux);
      w->aux = nullptr;
    }
  }
  // the remembered stream may be destroyed by the caller after a shutdown: never touch it again
  for (DeviceWs* w : all)
    if (w) {
      std::lock_guard<std::mutex> lk(w->mu);
      w->has_last = false;
      w->last_stream = nullptr;
    }
  (void)hipSetDevice(cur);
}

int c3p_pwc_unitary(const void* h0, int64_t h0_bstride, const void* hks, int64_t hks_bstride,
                    const double* signals, double dt, int B, int K, int N, int D, int flags,
                    const double* fr_phase, void* U_out, void* dUs_out, void* stream) {
  return pwc_common(0, h0, h0_bstride, hks, hks_bstride, signals, nullptr, 0, dt, B, K, N, D, flags,
                    fr_phase, U_out, dUs_out, stream);
}

int c3p_pwc_lindblad(const void* h0, int64_t h0_bstride, const void* hks, int64_t hks_bstride,
                     const double* signals, const void* col_ops, int C, double dt, int B, int K,
                     int N, int D, int flags, const double* fr_phase, void* U_out, void* dUs_out,
                     void* stream) {
  return pwc_common(1, h0, h0_bstride, hks, hks_bstride, signals, col_ops, C, dt, B, K, N, D, flags,
                    fr_phase, U_out, dUs_out, stream);
}

int c3p_expm(const void* A, int n, int D, int flags, void* out, void* stream) {
  if (n < 0 || D <= 0) return fail("bad sizes n=%d D=%d", n, D);
  if (n == 0) return 0;
  if (!A || !out) return fail("NULL matrix pointer");
  hipStream_t st = (hipStream_t)stream;
  WsLock lk(st);
  DeviceWs* w = lk.w;
  if (!w) return fail("no HIP device");
  if (!lk.ok) return fail("hipStreamWaitEvent on the previous call's stream failed");
  Stage sg{w, st};
  const void* d_A = A;
  void* d_out = out;
  const size_t bytes = (size_t)n * D * D * sizeof(cplx);
  if (flags & C3P_HOST_PTRS) {
    if (sg.in(A, bytes, &d_A)) return -1;
    if (sg.out(out, bytes, &d_out)) return -1;
  }
  bool done = false;
  if (!(flags & C3P_FORCE_GENERIC) && D <= kSmallDLimit && c3p_smalld_supported(D)) {
    const int rc = run_xg_smalld(w, (const cplx*)d_A, (long)D * D, 1.0, 0.0, n, 1, D, nullptr, (cplx*)d_out, nullptr, st);
    if (rc < 0) return -1;
    done = (rc == 0);
  }
  if (!done && !(flags & C3P_FORCE_GENERIC) && D >= 13 && D <= 40) {
    const int rc = run_xg_midd(w, (const cplx*)d_A, (long)D * D, 1.0, 0.0, n, 1, D, nullptr, (cplx*)d_out, nullptr, st);
    if (rc < 0) return -1;
    done = (rc == 0);
  }
  ChainArgs a = {};
  a.mode = C3P_MODE_EXPM;
  a.mats = (const cplx*)d_A;
  a.B = n;
  a.N = 1;
  a.D = D;
  a.Dm = D;
  if (!done && run_chain_generic(w, a, (cplx*)d_out, st)) return -1;
  if (flags & C3P_HOST_PTRS) return sg.finish();
  return 0;
}

int c3p_matmul_chain(const void* M, int B, int N, int D, int flags, void* out, void* stream) {
  if (B < 0 || N < 0 || D <= 0) return fail("bad sizes B=%d N=%d D=%d", B, N, D);
  if (B == 0) return 0;
  if (N == 0) return fail("empty matrix list (N == 0)");
  if (!M || !out) return fail("NULL matrix pointer");
  hipStream_t st = (hipStream_t)stream;
  WsLock lk(st);
  DeviceWs* w = lk.w;
  if (!w) return fail("no HIP device");
  if (!lk.ok) return fail("hipStreamWaitEvent on the previous call's stream failed");
  Stage sg{w, st};
  const void* d_M = M;
  void* d_out = out;
  if (flags & C3P_HOST_PTRS) {
    if (sg.in(M, (size_t)B * N * D * D * sizeof(cplx), &d_M)) return -1;
    if (sg.out(out, (size_t)B * D * D * sizeof(cplx), &d_out)) return -1;
  }
  ChainArgs a = {};
  a.mode = C3P_MODE_GIVEN;
  a.mats = (const cplx*)d_M;
  a.B = B;
  a.N = N;
  a.D = D;
  a.Dm = D;
  a.right_order = (flags & C3P_ORDER_RIGHT) ? 1 : 0;
  if (!(flags & C3P_FORCE_GENERIC) && D <= kSmallDLimit && c3p_smalld_supported(D)) {
    g_last_kernel = C3P_KERNEL_SMALLD;
    if (combine_smalld(w, (const cplx*)d_M, B, N, D, a.right_order, nullptr, (cplx*)d_out, st)) return -1;
  } else if (!(flags & C3P_FORCE_GENERIC) && D >= 13 && D <= 40) {
    g_last_kernel = C3P_KERNEL_MFMA;
    if (combine_midd(w, (const cplx*)d_M, B, N, D, a.right_order, nullptr, (cplx*)d_out, st)) return -1;
  } else if (run_chain_generic(w, a, (cplx*)d_out, st)) {
    return -1;
  }
  if (flags & C3P_HOST_PTRS) return sg.finish();
  return 0;
}

static int kron_common(const void* A, const void* Bm, int n, int Da, int Db, int which, int flags,
                       void* out, void* stream) {
  if (n < 0 || Da <= 0 || Db <= 0) return fail("bad sizes n=%d Da=%d Db=%d", n, Da, Db);
  if (n == 0) return 0;
  if (!A || !out || (which == 0 && !Bm)) return fail("NULL matrix pointer");
  hipStream_t st = (hipStream_t)stream;
  WsLock lk(st);
  DeviceWs* w = lk.w;
  if (!w) return fail("no HIP device");
  if (!lk.ok) return fail("hipStreamWaitEvent on the previous call's stream failed");
  Stage sg{w, st};
  const void *d_A = A, *d_B = Bm;
  void* d_out = out;
  const size_t Dm = (size_t)Da * Db;
  if (flags & C3P_HOST_PTRS) {
    if (sg.in(A, (size_t)n * Da * Da * sizeof(cplx), &d_A)) return -1;
    if (sg.in(which == 0 ? Bm : nullptr, (size_t)n * Db * Db * sizeof(cplx), &d_B)) return -1;
    if (sg.out(out, (size_t)n * Dm * Dm * sizeof(cplx), &d_out)) return -1;
  }
  LAUNCH_TRY(c3p_launch_kron((const cplx*)d_A, (const cplx*)d_B, n, Da, Db, which, (cplx*)d_out, st));
  if (flags & C3P_HOST_PTRS) return sg.finish();
  return 0;
}

int c3p_kron(const void* A, const void* Bm, int n, int Da, int Db, int flags, void* out, void* stream) {
  return kron_common(A, Bm, n, Da, Db, 0, flags, out, stream);
}

int c3p_superop(const void* A, int n, int D, int which, int flags, void* out, void* stream) {
  if (which < 0 || which > 2) return fail("bad superoperator kind %d", which);
  return kron_common(A, nullptr, n, D, D, which + 1, flags, out, stream);
}

}  // extern "C"

namespace {
// Segmented integration (see c3p_ode_solve): `a` describes the plain problem (M = 1: a state, or M = D: rk4_unitary).
// U_out (or NULL) receives the step map of the whole interval [B,D,D]; psi_out (or NULL) = map x init.
int ode_segmented(DeviceWs* w, OdeArgs a, int nseg, const cplx* init, long init_bstride, cplx* U_out, cplx* psi_out, hipStream_t st) {
  const int D = a.D, B = a.B;
  const size_t cs = sizeof(cplx);
  void *v_id, *v_maps, *v_U = U_out;
  if (ws_get(w, SL_CLP, (size_t)D * D * cs, &v_id)) return -1;
  if (ws_get(w, SL_SEG_A, (size_t)B * nseg * D * D * cs, &v_maps)) return -1;
  if (!v_U && ws_get(w, SL_SCRATCH, (size_t)B * D * D * cs, &v_U)) return -1;
  if (g_dry) return 0;
  LAUNCH_TRY(c3p_launch_ode_identity((cplx*)v_id, D, st));
  a.M = D;
  a.init = (const cplx*)v_id;
  a.init_bstride = 0;
  a.want_all = 0;
  a.seg_count = nseg;
  a.seg_len = (a.n_steps + nseg - 1) / nseg;
  a.states = (cplx*)v_maps;
  if (D > 16) {
    // 17 <= D <= 48: the segment maps on the matrix-core kernel (propagator step: one product per stage), combined on the
    // mid-D chain kernel
    a.step = C3P_STEP_PROPAGATOR_ID;
    LAUNCH_TRY(c3p_launch_ode_rhoq(a, st));
    if (combine_midd(w, (const cplx*)v_maps, B, nseg, D, 0, nullptr, (cplx*)v_U, st)) return -1;
  } else {
    LAUNCH_TRY(c3p_launch_ode_row(a, nullptr, st));
    if (combine_smalld(w, (const cplx*)v_maps, B, nseg, D, 0, nullptr, (cplx*)v_U, st)) return -1;
  }
  if (psi_out) HIP_TRY(c3p_launch_ode_apply((const cplx*)v_U, init, init_bstride, psi_out, B, D, st));
  return 0;
}

// Trajectory (want_all) of a small batch of Schroedinger states at D <= 12 in time segments: the segment maps as above, the
// state at the start of every segment from them, then the B x nseg pieces of the trajectory side by side (lane-row kernel).
int ode_trajectory_segmented(DeviceWs* w, OdeArgs a, int nseg, hipStream_t st) {
  const int D = a.D, B = a.B;
  const size_t cs = sizeof(cplx);
  void *v_id, *v_maps, *v_st;
  if (ws_get(w, SL_CLP, (size_t)D * D * cs, &v_id)) return -1;
  if (ws_get(w, SL_SEG_A, (size_t)B * nseg * D * D * cs, &v_maps)) return -1;
  if (ws_get(w, SL_SEG_B, (size_t)B * nseg * D * cs, &v_st)) return -1;
  if (g_dry) return 0;
  LAUNCH_TRY(c3p_launch_ode_identity((cplx*)v_id, D, st));
  OdeArgs m = a;
  m.M = D;
  m.init = (const cplx*)v_id;
  m.init_bstride = 0;
  m.want_all = 0;
  m.seg_count = nseg;
  m.seg_len = (a.n_steps + nseg - 1) / nseg;
  m.states = (cplx*)v_maps;
  if (D > 16) {
    m.step = C3P_STEP_PROPAGATOR_ID;  // 17 <= D <= 48: the segment maps on the matrix-core kernel (one product per stage)
    LAUNCH_TRY(c3p_launch_ode_rhoq(m, st));
  } else {
    LAUNCH_TRY(c3p_launch_ode_row(m, nullptr, st));
  }
  LAUNCH_TRY(c3p_launch_ode_starts((const cplx*)v_maps, a.init, a.init_bstride, (cplx*)v_st, B, nseg, D, st));
  OdeArgs t = a;
  t.init = (const cplx*)v_st;
  t.init_bstride = D;
  t.seg_count = nseg;
  t.seg_len = m.seg_len;
  t.seg_traj = 1;
  if (D > 16)
    LAUNCH_TRY(c3p_launch_ode_rowq(t, st));
  else
    LAUNCH_TRY(c3p_launch_ode_row(t, nullptr, st));
  return 0;
}

// trajectories of few Schroedinger states at 17 <= D <= 48: segment maps on the matrix-core kernel, pieces on the lane-row
// kernel of c3p_ode_rowq.hip (one time segment per wavefront)
int ode_mfma_traj_segments(const OdeArgs& a0) {
  if (c3p_opt_on(C3P_OPT_ode_no_seg)) return 0;
  if (a0.D < 17 || a0.D > 48 || !a0.want_all || a0.M != 1 || a0.reset_each_step || a0.transpose_out || a0.hs || a0.K > 4 || a0.n_steps < 64) return 0;
  OdeArgs a = a0;
  a.step = C3P_STEP_PROPAGATOR_ID;
  a.M = a.D;
  a.want_all = 0;
  a.seg_count = 2;
  if (!c3p_ode_rhoq_supported(a) || !c3p_ode_rowq_supported(a0)) return 0;
  long S = 512 / a0.B;
  if (S > a0.n_steps / 16) S = a0.n_steps / 16;
  if (S > 64) S = 64;
  return S >= 4 ? (int)S : 0;
}

// Time segments for few samples at 17 <= D <= 48 (final state / propagator only): one workgroup per (sample, segment) on the
// matrix-core kernel.  A sample alone keeps ONE CU busy for n_steps x ~5 us; S segments spread it over S CUs (the chip has
// 256), at the price of the ordered product of S maps.  min_gain: how many segments the change has to offer at least
// (a vector state that would otherwise run on the lane-row kernel pays 1.5x per step for becoming a matrix).
int ode_mfma_segments(const OdeArgs& a0, int min_segments) {
  if (c3p_opt_on(C3P_OPT_ode_no_seg)) return 0;
  if (a0.D < 17 || a0.D > 48 || a0.want_all || a0.reset_each_step || a0.transpose_out || a0.hs || a0.K > 4 || a0.n_steps < 64) return 0;
  int nig, nj, wd;
  if (!c3p_midd_geometry(a0.D, &nig, &nj, &wd)) return 0;  // (the ordered product of the maps runs on the mid-D chain kernel: D <= 40)
  OdeArgs a = a0;
  a.step = C3P_STEP_PROPAGATOR_ID;
  a.M = a.D;
  a.seg_count = 2;
  if (!c3p_ode_rhoq_supported(a)) return 0;
  long S = 512 / a.B;
  if (S > a.n_steps / 16) S = a.n_steps / 16;
  if (S > 64) S = 64;
  return S >= min_segments ? (int)S : 0;
}
}  // namespace

extern "C" {

int c3p_ode_solve(const void* h0, const void* hks, const double* signals, const void* col_ops,
                  int C, double dt, int B, int K, int N, int D, int solver, int step,
                  const void* init, int64_t init_bstride, int want_all, int flags, void* states,
                  void* stream) {
  if (B < 0 || N < 0 || D <= 0 || K < 0 || K > 32) return fail("bad sizes B=%d K=%d N=%d D=%d", B, K, N, D);
  if (solver < 0 || solver > 3) return fail("unknown solver id %d", solver);
  if (step < 0 || step > 2) return fail("unknown step function id %d", step);
  if (B == 0) return 0;
  if (N < 2) return fail("the ODE solver needs at least two time samples (N=%d)", N);
  if (!h0 || !init || !states) return fail("NULL pointer argument");
  if (K > 0 && (!hks || !signals)) return fail("K > 0 but hks/signals missing");
  if (step == C3P_STEP_LINDBLAD && (!col_ops || C <= 0)) return fail("lindblad step needs col_ops");
  if (step != C3P_STEP_LINDBLAD) C = 0;
  const int M = (step == C3P_STEP_SCHRODINGER) ? 1 : D;
  const size_t cs = sizeof(cplx);
  hipStream_t st = (hipStream_t)stream;
  WsLock lk(st);
  DeviceWs* w = lk.w;
  if (!w) return fail("no HIP device");
  if (!lk.ok) return fail("hipStreamWaitEvent on the previous call's stream failed");
  Stage sg{w, st};
  const void *d_h0 = h0, *d_hks = hks, *d_sig = signals, *d_col = col_ops, *d_init = init;
  void* d_states = states;
  const size_t out_elems = (size_t)B * (want_all ? N : 1) * D * M;
  if (flags & C3P_HOST_PTRS) {
    const size_t init_elems = init_bstride ? (size_t)(B - 1) * init_bstride + (size_t)D * M : (size_t)D * M;
    if (sg.in(h0, (size_t)D * D * cs, &d_h0)) return -1;
    if (sg.in(hks, (size_t)K * D * D * cs, &d_hks)) return -1;
    if (sg.in(signals, (size_t)B * K * N * sizeof(double), &d_sig)) return -1;
    if (sg.in(col_ops, (size_t)C * D * D * cs, &d_col)) return -1;
    if (sg.in(init, init_elems * cs, &d_init)) return -1;
    if (sg.out(states, out_elems * cs, &d_states)) return -1;
  }
  OdeArgs a = {};
  a.h0 = (const cplx*)d_h0;
  a.hks = (const cplx*)d_hks;
  a.signals = (const double*)d_sig;
  a.col_ops = (const cplx*)d_col;
  a.init = (const cplx*)d_init;
  a.init_bstride = init_bstride;
  a.dt = dt;
  a.B = B;
  a.K = K;
  a.N = N;
  a.D = D;
  a.M = M;
  a.C = C;
  a.solver = solver;
  a.step = step;
  a.want_all = want_all ? 1 : 0;
  a.n_steps = N;
  a.u_stride = 1;
  a.states = (cplx*)d_states;
  if (K > 4 && D <= 16 && N >= 2 && step == C3P_STEP_SCHRODINGER && !c3p_opt_on(C3P_OPT_ode_wg)) {
    // more than four control lines, vector states: the lane-row kernels hold the rows of four operators in registers; H is
    // assembled for every sample index first (B N D^2 numbers) and the kernel interpolates it between consecutive samples --
    // the same linear interpolation, taken after the sum instead of before it.  Beyond 8 GB, and for rho-valued states (no
    // faster there), the workgroup kernel takes the call.
    const size_t hb = (size_t)B * N * D * D * cs;
    if (hb <= ((size_t)8 << 30)) {
      void* hv;
      if (ws_get(w, SL_SEG_F, hb, &hv)) return -1;
      LAUNCH_TRY(c3p_launch_ode_assemble_hs(a, (cplx*)hv, st));
      a.hs = (const cplx*)hv;
      a.hs_bstride = (long)N * D * D;
      a.hs_lerp = 1;
      if (!c3p_ode_row_supported(a)) {  // (stage slots beyond the LDS: the workgroup kernel assembles H itself)
        a.hs = nullptr;
        a.hs_bstride = 0;
        a.hs_lerp = 0;
      }
    }
  }
  if (c3p_ode_row_supported(a)) {
    // lane-row kernels (c3p_ode_row.hip): one sample per 16-lane row, state and operator rows in registers
    void* aux = nullptr;
    if (C > 0 && ws_get(w, SL_TABLES, c3p_ode_row_aux_bytes(D, C), &aux)) return -1;
    g_last_kernel = C3P_KERNEL_ODE_ROW;
    if (record_start(w, st)) return -1;
    const int nseg = (step == C3P_STEP_SCHRODINGER) ? c3p_ode_row_segments(a) : 0;
    const int tseg = (step == C3P_STEP_SCHRODINGER && nseg == 0) ? c3p_ode_row_traj_segments(a) : 0;
    if (tseg > 0) {
      // small batch, whole trajectory: segment maps -> start state of every segment -> the pieces side by side
      if (ode_trajectory_segmented(w, a, tseg, st)) return -1;
    } else if (nseg > 0) {
      // small batch, final state only: the equations are linear, so the interval is cut into nseg time segments, every
      // segment integrates the D columns of the identity (its step map), the maps are multiplied in order by the small-D
      // chain kernel and the product is applied to the initial state -- D x the arithmetic on nseg x D x the lanes
      if (ode_segmented(w, a, nseg, (const cplx*)d_init, init_bstride, nullptr, (cplx*)d_states, st)) return -1;
    } else {
      // rho-valued states, more than four control lines, no collapse operators, a batch that leaves SIMDs idle on the lane rows
      // (one wave = four samples; measured crossover between 2048 and 16 384 samples, profiles/r04/ode_fallbacks.json): COMPLEX
      // operators run 1.1 - 1.3x faster on the workgroup kernel there, real ones 1.4 - 1.7x faster on the lane rows.  The device
      // decides (both kernels look at the operators and one of them leaves at once), as regr_prep_kernel does for the PWC path.
      const bool split = step == C3P_STEP_VON_NEUMANN && K > 4 && C == 0 && B <= 4096 && !c3p_opt_on(C3P_OPT_ode_no_split);
      a.complex_to_wg = split ? 1 : 0;
      // (the host cannot know which of the two did the work: a distinct id, so that profiles do not book a workgroup-kernel
      // run on the lane rows; the second launch -- B workgroups of 256 threads that leave at once for real operators -- is paid
      // on every such call)
      if (split) g_last_kernel = C3P_KERNEL_ODE_ROW_OR_WG;
      LAUNCH_TRY(c3p_launch_ode_row(a, aux, st));
      if (split) {
        OdeArgs g = a;
        g.complex_to_wg = 0;
        g.wg_if_complex = 1;
        const size_t elems = c3p_ode_elems(D, M, C);
        const bool global = elems * cs > (size_t)(150 * 1024);
        if (global) {
          void* v;
          if (ws_get(w, SL_SCRATCH, (size_t)B * elems * cs, &v)) return -1;
          g.scratch = (cplx*)v;
          g.scratch_stride = (long)elems;
        }
        LAUNCH_TRY(c3p_launch_ode(g, global, st));
      }
    }
    if (record_stop(w, st)) return -1;
    if (flags & C3P_HOST_PTRS) return sg.finish();
    return 0;
  }
  if (step == C3P_STEP_SCHRODINGER) {
    const int tseg = ode_mfma_traj_segments(a);
    if (tseg > 0) {
      g_last_kernel = C3P_KERNEL_ODE_ROW;
      if (record_start(w, st)) return -1;
      if (ode_trajectory_segmented(w, a, tseg, st)) return -1;
      if (record_stop(w, st)) return -1;
      if (flags & C3P_HOST_PTRS) return sg.finish();
      return 0;
    }
    const int nseg = ode_mfma_segments(a, 4);
    if (nseg > 0) {
      // few states at 17 <= D <= 48, final state only: segment maps on the matrix-core kernel, psi = U psi0
      g_last_kernel = C3P_KERNEL_ODE_MFMA;
      if (record_start(w, st)) return -1;
      if (ode_segmented(w, a, nseg, (const cplx*)d_init, init_bstride, nullptr, (cplx*)d_states, st)) return -1;
      if (record_stop(w, st)) return -1;
      if (flags & C3P_HOST_PTRS) return sg.finish();
      return 0;
    }
  }
  if (c3p_ode_rowq_supported(a)) {
    // 17 <= D <= 48 vector states: several DPP rows per sample, operators in LDS (c3p_ode_rowq.hip)
    g_last_kernel = C3P_KERNEL_ODE_ROW;
    if (record_start(w, st)) return -1;
    LAUNCH_TRY(c3p_launch_ode_rowq(a, st));
    if (record_stop(w, st)) return -1;
    if (flags & C3P_HOST_PTRS) return sg.finish();
    return 0;
  }
  if (c3p_ode_rhoq_supported(a)) {
    // 17 <= D <= 48 rho-valued states: register tiles + 16x16x4 fp64 MFMA products (c3p_ode_rhoq.hip)
    g_last_kernel = C3P_KERNEL_ODE_MFMA;
    if (record_start(w, st)) return -1;
    LAUNCH_TRY(c3p_launch_ode_rhoq(a, st));
    if (record_stop(w, st)) return -1;
    if (flags & C3P_HOST_PTRS) return sg.finish();
    return 0;
  }
  g_last_kernel = C3P_KERNEL_ODE_WG;
  const size_t elems = c3p_ode_elems(D, M, C);
  const bool global = elems * cs > (size_t)(150 * 1024);
  if (global) {
    void* v;
    if (ws_get(w, SL_SCRATCH, (size_t)B * elems * cs, &v)) return -1;
    a.scratch = (cplx*)v;
    a.scratch_stride = (long)elems;
  }
  LAUNCH_TRY(c3p_launch_ode(a, global, st));
  if (flags & C3P_HOST_PTRS) return sg.finish();
  return 0;
}

int c3p_rk4_unitary(const void* h0, const void* hks, const double* signals, const void* hs,
                    int64_t hs_bstride, double dt, int B, int K, int Ns, int D, int flags,
                    void* U_out, void* dUs_out, void* stream) {
  if (B < 0 || Ns < 0 || D <= 0 || K < 0 || K > 32) return fail("bad sizes B=%d K=%d Ns=%d D=%d", B, K, Ns, D);
  if (B == 0) return 0;
  if (Ns < 3) return fail("rk4_unitary needs at least three Hamiltonian samples (Ns=%d)", Ns);
  if (!U_out) return fail("U_out is NULL");
  if (!hs && (!h0 || (K > 0 && (!hks || !signals)))) return fail("NULL Hamiltonian input");
  const int n_steps = (Ns - 1) / 2;  // range(0, len(h) - 2, 2), propagation.py:79,251
  const size_t cs = sizeof(cplx);
  hipStream_t st = (hipStream_t)stream;
  WsLock lk(st);
  DeviceWs* w = lk.w;
  if (!w) return fail("no HIP device");
  if (!lk.ok) return fail("hipStreamWaitEvent on the previous call's stream failed");
  Stage sg{w, st};
  const void *d_h0 = h0, *d_hks = hks, *d_sig = signals, *d_hs = hs;
  void *d_U = U_out, *d_dUs = dUs_out;
  if (flags & C3P_HOST_PTRS) {
    if (hs) {
      const size_t one = (size_t)Ns * D * D;
      if (sg.in(hs, (hs_bstride ? (size_t)(B - 1) * hs_bstride + one : one) * cs, &d_hs)) return -1;
    } else {
      if (sg.in(h0, (size_t)D * D * cs, &d_h0)) return -1;
      if (sg.in(hks, (size_t)K * D * D * cs, &d_hks)) return -1;
      if (sg.in(signals, (size_t)B * K * Ns * sizeof(double), &d_sig)) return -1;
    }
    if (sg.out(U_out, (size_t)B * D * D * cs, &d_U)) return -1;
    if (sg.out(dUs_out, dUs_out ? (size_t)B * n_steps * D * D * cs : 0, &d_dUs)) return -1;
  }
  // identity initial state, shared by all samples
  void* v_id;
  if (ws_get(w, SL_CLP, (size_t)D * D * cs, &v_id)) return -1;
  LAUNCH_TRY(c3p_launch_ode_identity((cplx*)v_id, D, st));  // (on the device: no host round trip, capturable)
  OdeArgs a = {};
  a.h0 = (const cplx*)d_h0;
  a.hks = (const cplx*)d_hks;
  a.signals = (const double*)d_sig;
  a.hs = (const cplx*)d_hs;
  a.hs_bstride = hs_bstride;
  a.init = (const cplx*)v_id;
  a.init_bstride = 0;
  a.dt = dt;
  a.B = B;
  a.K = hs ? 0 : K;
  a.N = Ns;
  a.D = D;
  a.M = D;
  a.C = 0;
  a.solver = C3P_SOLVER_RK4;
  a.step = C3P_STEP_PROPAGATOR_ID;
  a.n_steps = n_steps;
  a.u_stride = 2;
  if (c3p_ode_row_supported(a)) {
    // every column of the propagator is an independent Schroedinger problem: B x D vector states on the lane-row kernel
    g_last_kernel = C3P_KERNEL_ODE_ROW;
    a.want_all = 0;
    a.states = (cplx*)d_U;
    const int nseg = c3p_ode_row_segments(a);
    if (nseg > 0) {  // few gates: time segments fill the chip (see c3p_ode_solve)
      if (ode_segmented(w, a, nseg, nullptr, 0, (cplx*)d_U, nullptr, st)) return -1;
    } else {
      LAUNCH_TRY(c3p_launch_ode_row(a, nullptr, st));
    }
    if (d_dUs) {
      a.want_all = 1;
      a.reset_each_step = 1;
      a.transpose_out = 1;
      a.states = (cplx*)d_dUs;
      LAUNCH_TRY(c3p_launch_ode_row(a, nullptr, st));
    }
    if (flags & C3P_HOST_PTRS) return sg.finish();
    return 0;
  }
  if (c3p_ode_rhoq_supported(a)) {
    // 17 <= D <= 48: the propagator is a D x D matrix state with ONE product per stage: matrix-core kernel (c3p_ode_rhoq.hip)
    g_last_kernel = C3P_KERNEL_ODE_MFMA;
    a.want_all = 0;
    a.states = (cplx*)d_U;
    const int nseg = ode_mfma_segments(a, 2);
    if (nseg > 0) {  // few gates: one workgroup per (gate, time segment)
      if (ode_segmented(w, a, nseg, nullptr, 0, (cplx*)d_U, nullptr, st)) return -1;
    } else {
      LAUNCH_TRY(c3p_launch_ode_rhoq(a, st));
    }
    if (d_dUs) {
      a.want_all = 1;
      a.reset_each_step = 1;
      a.transpose_out = 1;
      a.states = (cplx*)d_dUs;
      LAUNCH_TRY(c3p_launch_ode_rhoq(a, st));
    }
    if (flags & C3P_HOST_PTRS) return sg.finish();
    return 0;
  }
  if (c3p_ode_rowq_supported(a)) {
    g_last_kernel = C3P_KERNEL_ODE_ROW;
    a.want_all = 0;
    a.states = (cplx*)d_U;
    LAUNCH_TRY(c3p_launch_ode_rowq(a, st));
    if (d_dUs) {
      a.want_all = 1;
      a.reset_each_step = 1;
      a.transpose_out = 1;
      a.states = (cplx*)d_dUs;
      LAUNCH_TRY(c3p_launch_ode_rowq(a, st));
    }
    if (flags & C3P_HOST_PTRS) return sg.finish();
    return 0;
  }
  g_last_kernel = C3P_KERNEL_ODE_WG;
  const size_t elems = c3p_ode_elems(D, D, 0);
  const bool global = elems * cs > (size_t)(150 * 1024);
  if (global) {
    void* v;
    if (ws_get(w, SL_SCRATCH, (size_t)B * elems * cs, &v)) return -1;
    a.scratch = (cplx*)v;
    a.scratch_stride = (long)elems;
  }
  a.want_all = 0;
  a.states = (cplx*)d_U;
  LAUNCH_TRY(c3p_launch_ode(a, global, st));
  if (d_dUs) {
    a.want_all = 1;
    a.reset_each_step = 1;
    a.transpose_out = 1;
    a.states = (cplx*)d_dUs;
    LAUNCH_TRY(c3p_launch_ode(a, global, st));
  }
  if (flags & C3P_HOST_PTRS) return sg.finish();
  return 0;
}

int c3p_gate_overlap(const void* U, int B, int D, const int32_t* comp_rows, int L, const void* ideal,
                     int flags, void* overlap_out, void* stream) {
  if (B < 0 || D <= 0 || L <= 0 || L > D) return fail("bad sizes B=%d D=%d L=%d", B, D, L);
  if (B == 0) return 0;
  if (!U || !comp_rows || !ideal || !overlap_out) return fail("NULL pointer argument");
  const size_t cs = sizeof(cplx);
  hipStream_t st = (hipStream_t)stream;
  WsLock lk(st);
  DeviceWs* w = lk.w;
  if (!w) return fail("no HIP device");
  if (!lk.ok) return fail("hipStreamWaitEvent on the previous call's stream failed");
  Stage sg{w, st};
  const void *d_U = U, *d_rows = comp_rows, *d_G = ideal;
  void* d_out = overlap_out;
  if (flags & C3P_HOST_PTRS) {
    for (int a = 0; a < L; ++a)
      if (comp_rows[a] < 0 || comp_rows[a] >= D) return fail("comp_rows[%d]=%d outside [0,%d)", a, comp_rows[a], D);
    if (sg.in(U, (size_t)B * D * D * cs, &d_U)) return -1;
    if (sg.in(comp_rows, (size_t)L * sizeof(int32_t), &d_rows)) return -1;
    if (sg.in(ideal, (size_t)L * L * cs, &d_G)) return -1;
    if (sg.out(overlap_out, (size_t)B * cs, &d_out)) return -1;
  }
  LAUNCH_TRY(c3p_launch_overlap((const cplx*)d_U, B, D, (const int*)d_rows, L, (const cplx*)d_G, (cplx*)d_out, st));
  if (flags & C3P_HOST_PTRS) return sg.finish();
  return 0;
}

int c3p_pwc_lindblad_vjp(const void* h0, int64_t h0_bstride, const void* hks, int64_t hks_bstride, const double* signals,
                         const void* col_ops, int C, double dt, int B, int K, int N, int D, int flags, const double* fr_phase,
                         const void* U_bar, double* grad_signals, void* stream) {
  if (B < 0 || K <= 0 || N <= 0 || D <= 0 || C <= 0) return fail("bad sizes B=%d K=%d N=%d D=%d C=%d", B, K, N, D, C);
  if (flags & (C3P_PER_SLICE_H | C3P_ORDER_RIGHT)) return fail("c3p_pwc_lindblad_vjp: unsupported flag");
  if (B == 0) return 0;
  if (!h0 || !hks || !signals || !col_ops || !U_bar || !grad_signals) return fail("NULL pointer argument");
  if (h0_bstride < 0 || hks_bstride < 0) return fail("negative batch stride");
  const size_t cs = sizeof(cplx);
  const int Dm = D * D;
  hipStream_t st = (hipStream_t)stream;
  WsLock lk(st);
  DeviceWs* w = lk.w;
  if (!w) return fail("no HIP device");
  if (!lk.ok) return fail("hipStreamWaitEvent on the previous call's stream failed");
  Stage sg{w, st};
  const void *d_h0 = h0, *d_hks = hks, *d_sig = signals, *d_ph = fr_phase, *d_ub = U_bar, *d_col = col_ops;
  void* d_grad = grad_signals;
  if (flags & C3P_HOST_PTRS) {
    if (sg.in(h0, ((size_t)(B - 1) * (size_t)h0_bstride + (size_t)D * D) * cs, &d_h0)) return -1;
    if (sg.in(hks, ((size_t)(B - 1) * (size_t)hks_bstride + (size_t)K * D * D) * cs, &d_hks)) return -1;
    if (sg.in(signals, (size_t)B * K * N * sizeof(double), &d_sig)) return -1;
    if (sg.in(col_ops, (size_t)C * D * D * cs, &d_col)) return -1;
    if (sg.in(U_bar, (size_t)B * Dm * Dm * cs, &d_ub)) return -1;
    if (fr_phase && sg.in(fr_phase, (size_t)B * Dm * sizeof(double), &d_ph)) return -1;
    if (sg.out(grad_signals, (size_t)B * K * N * sizeof(double), &d_grad)) return -1;
  }
  void* clp;
  if (ws_get(w, SL_CLP, (size_t)Dm * Dm * cs, &clp)) return -1;
  LAUNCH_TRY(c3p_launch_clp((const cplx*)d_col, C, D, (cplx*)clp, st));
  // The general-generator sweeps keep the slice propagators and the prefix of every slice: 2 N D^4 complex per sample.  Large
  // batches are processed in chunks of samples that keep that below 24 GB (C3P_GRAD_CHUNK overrides the chunk size).
  auto in_chunks = [&](auto&& run) -> int {  // run(b0, nb) -> 0 done, 1 not applicable, -1 error
    long Bc = (long)(grad_store_budget(w, SL_OUT1) / (2 * (size_t)N * Dm * Dm * cs + 1));
    if (c3p_opt(C3P_OPT_grad_chunk) > 0) Bc = c3p_opt(C3P_OPT_grad_chunk);
    if (Bc < 1) Bc = 1;
    for (long b0 = 0; b0 < B; b0 += Bc) {
      const int rc = run(b0, (int)(B - b0 < Bc ? B - b0 : Bc));
      if (rc != 0) return rc;
    }
    return 0;
  };
  const cplx* p_h0 = (const cplx*)d_h0;
  const cplx* p_hk = (const cplx*)d_hks;
  const double* p_sig = (const double*)d_sig;
  const double* p_ph = (const double*)d_ph;
  const cplx* p_ub = (const cplx*)d_ub;
  double* p_grad = (double*)d_grad;
  const long gsz = (long)Dm * Dm;
  auto phase_at = [&](long b0) { return p_ph ? p_ph + b0 * Dm : nullptr; };
  if (Dm <= kSmallDLimit && c3p_smalld_supported(Dm) && K <= 8 && !(flags & C3P_FORCE_GENERIC) && !c3p_opt_on(C3P_OPT_tiled_grad) &&
      !c3p_opt_on(C3P_OPT_valu_grad)) {
    // superoperators up to 12 x 12 (D <= 3): the general-generator sweep on the small-D matrix-core kernels
    if (record_start(w, st)) return -1;
    const int rc = in_chunks([&](long b0, int nb) {
      return run_vjp_lind_smalld(w, p_h0 + b0 * h0_bstride, h0_bstride, p_hk + b0 * hks_bstride, hks_bstride, p_sig + b0 * K * N,
                                 (const cplx*)clp, dt, nb, K, N, D, Dm, phase_at(b0), p_ub + b0 * gsz, p_grad + b0 * K * N,
                                 (flags & C3P_HERMITIAN_H) != 0, st);
    });
    if (rc < 0) return -1;
    if (rc == 0) {
      g_last_kernel = C3P_KERNEL_SMALLD;
      if (record_stop(w, st)) return -1;
      if (flags & C3P_HOST_PTRS) return sg.finish();
      return 0;
    }
  }
  if (D == 4 && (flags & C3P_HERMITIAN_H) && K <= 8 && !(flags & C3P_FORCE_GENERIC) && !c3p_opt_on(C3P_OPT_no_smallr) &&
      !c3p_opt_on(C3P_OPT_tiled_grad) && !c3p_opt_on(C3P_OPT_valu_grad)) {
    // two qubits with declared Hermitian Hamiltonians: the real Hermitian-basis kernels of c3p_smallr.hip, both halves
    if (record_start(w, st)) return -1;
    const int rc = in_chunks([&](long b0, int nb) {
      return run_vjp_lind_smallr16(w, p_h0 + b0 * h0_bstride, h0_bstride, p_hk + b0 * hks_bstride, hks_bstride, p_sig + b0 * K * N,
                                   (const cplx*)clp, dt, nb, K, N, D, Dm, phase_at(b0), p_ub + b0 * gsz, p_grad + b0 * K * N, st);
    });
    if (rc < 0) return -1;
    if (rc == 0) {
      g_last_kernel = C3P_KERNEL_MFMA;
      if (record_stop(w, st)) return -1;
      if (flags & C3P_HOST_PTRS) return sg.finish();
      return 0;
    }
  }
  // 49 x 49 .. 81 x 81 superoperators (D = 7, 8, 9; cfg4) and, zero padded in the 49 class, 36 x 36 (D = 6): on-chip backward sweep in
  // the Hermitian basis, real arithmetic; the transposed local prefix of every slice (N D^4 doubles per sample) is kept in HBM:
  // chunks of samples below the budget.  1 = a Hamiltonian is not Hermitian (the caller goes on to the complex sweeps).
  auto try_regr = [&]() -> int {
    if (record_start(w, st)) return -1;
    long Bc = (long)(grad_store_budget(w, SL_OUT1) / ((size_t)N * Dm * Dm * sizeof(double) + 1));
    if (c3p_opt(C3P_OPT_grad_chunk) > 0) Bc = c3p_opt(C3P_OPT_grad_chunk);
    if (Bc < 1) Bc = 1;
    int rc = 0;
    for (long b0 = 0; b0 < B && rc == 0; b0 += Bc) {
      const int nb = (int)(B - b0 < Bc ? B - b0 : Bc);
      rc = run_vjp_lind_regr(w, p_h0 + b0 * h0_bstride, h0_bstride, p_hk + b0 * hks_bstride, hks_bstride, p_sig + b0 * K * N,
                             (const cplx*)clp, dt, nb, K, N, D, Dm, phase_at(b0), p_ub + b0 * gsz, p_grad + b0 * K * N, st);
      if (rc == 1 && b0 > 0) return fail("internal: the Hermitian-basis sweep declined a later chunk");
    }
    if (rc != 0) return rc;
    g_last_kernel = C3P_KERNEL_MFMA;
    if (record_stop(w, st)) return -1;
    if (flags & C3P_HOST_PTRS) return sg.finish() ? -1 : 0;
    return 0;
  };
  const bool regr_ok = lind_regr_grad_ok(D, Dm) && !(flags & C3P_FORCE_GENERIC) && !c3p_opt_on(C3P_OPT_tiled_grad) &&
                       !c3p_opt_on(C3P_OPT_valu_grad) && !c3p_opt_on(C3P_OPT_no_hermitian_basis);
  if (regr_ok && D == 6) {
    const int rc = try_regr();
    if (rc <= 0) return rc;
  }
  if (Dm >= 13 && Dm <= 36 && !(flags & C3P_FORCE_GENERIC) && !c3p_opt_on(C3P_OPT_tiled_grad) && !c3p_opt_on(C3P_OPT_valu_grad)) {
    // 16 x 16 .. 36 x 36 superoperators (D = 4, 5, 6): the same sweep on the mid-D matrix-core kernels
    if (record_start(w, st)) return -1;
    const int rc = in_chunks([&](long b0, int nb) {
      return run_vjp_lind_midd(w, p_h0 + b0 * h0_bstride, h0_bstride, p_hk + b0 * hks_bstride, hks_bstride, p_sig + b0 * K * N,
                               (const cplx*)clp, dt, nb, K, N, D, Dm, phase_at(b0), p_ub + b0 * gsz, p_grad + b0 * K * N, st);
    });
    if (rc < 0) return -1;
    if (rc == 0) {
      g_last_kernel = C3P_KERNEL_MFMA;
      if (record_stop(w, st)) return -1;
      if (flags & C3P_HOST_PTRS) return sg.finish();
      return 0;
    }
  }
  if (Dm <= 36 && !c3p_opt_on(C3P_OPT_tiled_grad)) {
    // the same sweep in three VALU kernels on dense generator tables (c3p_grad.hip, general form): fallback and second opinion
    bool global = false;
    auto run_valu = [&](long b0, int nb) -> int {
      const int nt = (h0_bstride || hks_bstride) ? nb : 1;
      void* tab;
      if (ws_get(w, SL_TABLES, (size_t)nt * (K + 1) * gsz * cs, &tab)) return -1;
      LAUNCH_TRY(c3p_launch_lind_generators(p_h0 + b0 * h0_bstride, h0_bstride, p_hk + b0 * hks_bstride, hks_bstride, (const cplx*)clp, nt,
                                            K, D, (cplx*)tab, st));
      GradArgs A = {};
      A.h0 = (const cplx*)tab;
      A.h0_bstride = nt > 1 ? (long)(K + 1) * gsz : 0;
      A.hks = (const cplx*)tab + gsz;
      A.hks_bstride = A.h0_bstride;
      A.signals = p_sig + b0 * K * N;
      A.fr_phase = phase_at(b0);
      A.Ubar = p_ub + b0 * gsz;
      A.dt = dt;
      A.B = nb;
      A.K = K;
      A.N = N;
      A.D = Dm;
      A.ld = Dm | 1;
      A.grad = p_grad + b0 * K * N;
      A.general = 1;
      long S = 4096 / nb;
      if (S > N / 8) S = N / 8;
      if (S < 1) S = 1;
      A.S = (int)S;
      void* v;
      if (ws_get(w, SL_SEG_A, (size_t)nb * A.S * gsz * cs, &v)) return -1;
      A.seg = (cplx*)v;
      if (ws_get(w, SL_SEG_B, (size_t)nb * A.S * gsz * cs, &v)) return -1;
      A.Mb = (cplx*)v;
      if (ws_get(w, SL_OUT1, ((size_t)nb * A.S + (size_t)nb * N) * gsz * cs, &v)) return -1;  // (SL_OUT0 stages grad_signals)
      A.pre = (cplx*)v;
      A.pstore = A.pre + (size_t)nb * A.S * gsz;
      global = c3p_grad_lds_bytes_general(Dm) > 150 * 1024;
      if (global) {
        A.scratch_stride = (long)C3P_GRAD_NMAT_GENERAL * A.ld * Dm;
        if (ws_get(w, SL_SCRATCH, (size_t)nb * A.S * A.scratch_stride * cs, &v)) return -1;
        A.scratch = (cplx*)v;
      }
      LAUNCH_TRY(c3p_launch_grad_seg(A, global, st));
      LAUNCH_TRY(c3p_launch_grad_scan_general(A, global, st));
      LAUNCH_TRY(c3p_launch_grad_bwd_general(A, global, st));
      return 0;
    };
    if (record_start(w, st)) return -1;
    if (in_chunks(run_valu) != 0) return -1;
    g_last_kernel = global ? C3P_KERNEL_GENERIC_GLOBAL : C3P_KERNEL_GENERIC_LDS;
    if (record_stop(w, st)) return -1;
    if (flags & C3P_HOST_PTRS) return sg.finish();
    return 0;
  }
  if (regr_ok && D != 6) {
    const int rc = try_regr();
    if (rc <= 0) return rc;
  }
  g_last_kernel = C3P_KERNEL_MFMA;
  if (record_start(w, st)) return -1;
  if (run_vjp_tiled(w, 1, (const cplx*)d_h0, h0_bstride, (const cplx*)d_hks, hks_bstride, (const double*)d_sig, (const cplx*)clp, dt, B,
                    K, N, D, Dm, (const double*)d_ph, (const cplx*)d_ub, (double*)d_grad, st))
    return -1;
  if (record_stop(w, st)) return -1;
  if (flags & C3P_HOST_PTRS) return sg.finish();
  return 0;
}

// ---- open-system evaluation from ONE forward pass: a caller-owned tape between c3p_pwc_lindblad_taped and its vjp ----
size_t c3p_pwc_lindblad_tape_bytes(int B, int K, int N, int D, int* segments_out) {
  if (B > 0 && K >= 1 && K <= 8 && N > 0 && D == 4) {
    // two qubits: the real Hermitian-basis kernels only (the taped calls then need C3P_HERMITIAN_H)
    const int S = c3p_opt_on(C3P_OPT_no_smallr) ? -1 : pick_segments_r(B, N, K, 16, true, true);
    if (segments_out) *segments_out = S > 0 ? S : 0;
    return S > 0 ? lind_smallr_sizes(B, K, N, 16, S, B).total() : 0;
  }
  if (B > 0 && K >= 1 && K <= 8 && N > 0 && D >= 2 && D <= 3) {
    // superoperators up to 9 x 9 on the small-D kernels: tables, segment products and the slice propagators (the generator is
    // not anti-Hermitian: the backward sweep cannot recompute prefixes from the adjoint side); per-sample operators need S % 4 = 0
    const int S = lind_small_segments(B, K, N, D * D, true);
    if (segments_out) *segments_out = S > 0 ? S : 0;
    return S > 0 ? lind_small_sizes(B, K, N, D * D, S, B).total() : 0;
  }
  if (B <= 0 || K < 1 || N <= 0 || D <= 0 || !c3p_regr_supported(D, D * D) || K > 16) {
    if (segments_out) *segments_out = 0;
    return 0;
  }
  const long S = lind_regr_segments(B, N);
  if (segments_out) *segments_out = (int)S;
  return lind_regr_sizes(B, K, N, D * D, S, B).total();
}

int c3p_pwc_lindblad_taped(const void* h0, int64_t h0_bstride, const void* hks, int64_t hks_bstride, const double* signals,
                           const void* col_ops, int C, double dt, int B, int K, int N, int D, int flags, const double* fr_phase,
                           void* U_out, void* tape, size_t tape_bytes, int segments, void* stream) {
  if (flags & ~C3P_HERMITIAN_H) return fail("c3p_pwc_lindblad_taped takes device pointers and no flags but C3P_HERMITIAN_H");
  if (B <= 0 || K < 1 || K > 16 || N <= 0 || D <= 0) return fail("bad sizes B=%d K=%d N=%d D=%d", B, K, N, D);
  const int Dm = D * D;
  if (!h0 || !hks || !signals || !col_ops || C <= 0 || !U_out || !tape) return fail("NULL pointer argument");
  if (h0_bstride < 0 || hks_bstride < 0) return fail("negative batch stride");
  if (segments < 1 || segments > N) return fail("bad segment count %d", segments);
  if (D <= 4) {
    // small-D kernels: the forward half of run_vjp_lind_smalld writes into the tape, U = the ordered product of its segments
    int seg_chk = 0;
    const size_t need = c3p_pwc_lindblad_tape_bytes(B, K, N, D, &seg_chk);
    if (need == 0 || seg_chk != segments) return fail("taped Lindblad evaluation: shape not served or segment count %d != %d (c3p_pwc_lindblad_tape_bytes)", segments, seg_chk);
    if (tape_bytes < need) return fail("tape too small: %zu bytes, need %zu (c3p_pwc_lindblad_tape_bytes)", tape_bytes, need);
    hipStream_t st = (hipStream_t)stream;
    WsLock lk(st);
    DeviceWs* w = lk.w;
    if (!w) return fail("no HIP device");
    if (!lk.ok) return fail("hipStreamWaitEvent on the previous call's stream failed");
    void* clp;
    if (ws_get(w, SL_CLP, (size_t)Dm * Dm * sizeof(cplx), &clp)) return -1;
    LAUNCH_TRY(c3p_launch_clp((const cplx*)col_ops, C, D, (cplx*)clp, st));
    if (lind_smallr_ok((flags & C3P_HERMITIAN_H) != 0, D, Dm, K, N, segments)) {
      // declared Hermitian: the tape holds the REAL tables, segment products and slice propagators (c3p_pwc_lindblad_vjp_taped
      // must be given the same flag); U from the complex copies of the segment products
      const bool ps = (h0_bstride != 0) || (hks_bstride != 0);
      const int nsamp = ps ? B : 1;
      const LindSmallRBufs rb = lind_smallr_carve(tape, lind_smallr_sizes(B, K, N, Dm, segments, nsamp), nsamp, Dm, K);
      void* sc;
      if (ws_get(w, SL_OUT2, (size_t)B * segments * Dm * Dm * sizeof(cplx), &sc)) return -1;
      if (record_start(w, st)) return -1;
      if (lind_smallr_forward(rb, (const cplx*)h0, h0_bstride, (const cplx*)hks, hks_bstride, signals, (const cplx*)clp, dt, B, K, N, D, Dm, segments,
                              (cplx*)sc, st))
        return -1;
      if (record_stop(w, st)) return -1;
      g_last_kernel = Dm > 12 ? C3P_KERNEL_MFMA : C3P_KERNEL_SMALLD;
      const int keep = g_last_kernel;
      const int rc = Dm > 12 ? combine_midd(w, (const cplx*)sc, B, segments, Dm, 0, fr_phase, (cplx*)U_out, st)
                             : combine_smalld(w, (const cplx*)sc, B, segments, Dm, 0, fr_phase, (cplx*)U_out, st);
      g_last_kernel = keep;
      return rc ? -1 : 0;
    }
    if (D == 4) return fail("the taped Lindblad evaluation at D = 4 runs in the Hermitian basis: declare the Hamiltonians Hermitian (C3P_HERMITIAN_H) or use the untaped pair");
    const LindSmallBufs bf = lind_small_carve(tape, lind_small_sizes(B, K, N, Dm, segments, B));
    if (record_start(w, st)) return -1;
    if (lind_small_forward(w, bf, (const cplx*)h0, h0_bstride, (const cplx*)hks, hks_bstride, signals, (const cplx*)clp, dt, B, K, N, D, Dm, segments,
                           (flags & C3P_HERMITIAN_H) != 0, st))
      return -1;
    if (record_stop(w, st)) return -1;
    g_last_kernel = C3P_KERNEL_SMALLD;
    if (segments == 1) {
      HIP_TRY(hipMemcpyAsync(U_out, bf.seg, (size_t)B * Dm * Dm * sizeof(cplx), hipMemcpyDeviceToDevice, st));
      if (fr_phase) LAUNCH_TRY(c3p_launch_rowphase((cplx*)U_out, fr_phase, B, Dm, st));
      return 0;
    }
    return combine_smalld(w, bf.seg, B, segments, Dm, 0, fr_phase, (cplx*)U_out, st) ? -1 : 0;
  }
  if (!c3p_regr_supported(D, Dm)) return fail("the taped Lindblad evaluation serves D = 2, 3, 4 (small-D tile kernels) and D = 7, 8, 9 (Hermitian-basis kernels), got D=%d", D);
  {
    // the segment count is the library's choice (c3p_pwc_lindblad_tape_bytes), as at D <= 4: the chain, scan and backward kernels
    // are tuned and tested for that range only, and the tape layout follows from it
    int seg_chk = 0;
    if (c3p_pwc_lindblad_tape_bytes(B, K, N, D, &seg_chk) == 0 || seg_chk != segments)
      return fail("taped Lindblad evaluation: shape not served or segment count %d != %d (c3p_pwc_lindblad_tape_bytes)", segments, seg_chk);
  }
  const LindRegrSizes z = lind_regr_sizes(B, K, N, Dm, segments, B);
  if (tape_bytes < z.total()) return fail("tape too small: %zu bytes, need %zu (c3p_pwc_lindblad_tape_bytes)", tape_bytes, z.total());
  hipStream_t st = (hipStream_t)stream;
  WsLock lk(st);
  DeviceWs* w = lk.w;
  if (!w) return fail("no HIP device");
  if (!lk.ok) return fail("hipStreamWaitEvent on the previous call's stream failed");
  const size_t cs = sizeof(cplx);
  void* clp;
  if (ws_get(w, SL_CLP, (size_t)Dm * Dm * cs, &clp)) return -1;
  LAUNCH_TRY(c3p_launch_clp((const cplx*)col_ops, C, D, (cplx*)clp, st));
  const LindRegrBufs bf = lind_regr_carve(tape, z);
  if (record_start(w, st)) return -1;
  const int rc = lind_regr_forward(w, bf, (const cplx*)h0, h0_bstride, (const cplx*)hks, hks_bstride, signals, (const cplx*)clp, dt, B, K, N, D,
                                   Dm, segments, st);
  if (rc < 0) return -1;
  if (rc == 1) return fail("a Hamiltonian is not Hermitian: its Lindblad generator is complex in the Hermitian basis (use c3p_pwc_lindblad and c3p_pwc_lindblad_vjp)");
  if (record_stop(w, st)) return -1;
  g_last_kernel = C3P_KERNEL_MFMA;
  // U: the segment products back in the reference's vectorisation (a copy: the tape keeps the real ones), ordered combine
  const bool per_sample = (h0_bstride != 0) || (hks_bstride != 0);
  const long S = segments;
  cplx* segc = (cplx*)U_out;
  if (S > 1) {
    void* sv;
    if (ws_get(w, SL_SEG_A, (size_t)B * S * Dm * Dm * cs, &sv)) return -1;
    segc = (cplx*)sv;
  }
  HIP_TRY(hipMemcpyAsync(segc, bf.seg, (size_t)B * S * Dm * Dm * cs, hipMemcpyDeviceToDevice, st));
  LAUNCH_TRY(c3p_launch_hb_to_complex(segc, (long)B * S, (int)S, bf.flag_f, per_sample ? 1 : 0, K, D, st));
  if (S == 1 && fr_phase) LAUNCH_TRY(c3p_launch_rowphase((cplx*)U_out, fr_phase, B, Dm, st));
  if (S > 1) {
    ChainArgs c = {};
    c.mode = C3P_MODE_GIVEN;
    c.mats = segc;
    c.B = B;
    c.N = (int)S;
    c.D = D;
    c.Dm = Dm;
    c.fr_phase = fr_phase;
    const int keep = g_last_kernel;
    const int rc2 = run_chain_generic(w, c, (cplx*)U_out, st, /*profile=*/false);
    g_last_kernel = keep;
    if (rc2) return -1;
  }
  return 0;
}

int c3p_pwc_lindblad_vjp_taped(const void* tape, size_t tape_bytes, int segments, int per_sample_operators, const double* signals, int B,
                               int K, int N, int D, int flags, const double* fr_phase, const void* U_bar, double* grad_signals,
                               void* stream) {
  if (flags & ~C3P_HERMITIAN_H) return fail("c3p_pwc_lindblad_vjp_taped takes device pointers and no flags but C3P_HERMITIAN_H (as given to c3p_pwc_lindblad_taped)");
  if (B <= 0 || K < 1 || K > 16 || N <= 0 || D <= 0) return fail("bad sizes B=%d K=%d N=%d D=%d", B, K, N, D);
  const int Dm = D * D;
  if (!tape || !signals || !U_bar || !grad_signals) return fail("NULL pointer argument");
  if (segments < 1 || segments > N) return fail("bad segment count %d", segments);
  if (D <= 4) {
    int seg_chk = 0;
    const size_t need = c3p_pwc_lindblad_tape_bytes(B, K, N, D, &seg_chk);
    if (need == 0 || seg_chk != segments) return fail("taped Lindblad evaluation: shape not served or segment count %d != %d", segments, seg_chk);
    if (tape_bytes < need) return fail("tape too small: %zu bytes, need %zu", tape_bytes, need);
    hipStream_t st = (hipStream_t)stream;
    WsLock lk(st);
    DeviceWs* w = lk.w;
    if (!w) return fail("no HIP device");
    if (!lk.ok) return fail("hipStreamWaitEvent on the previous call's stream failed");
    if (lind_smallr_ok((flags & C3P_HERMITIAN_H) != 0, D, Dm, K, N, segments)) {
      const int nsamp = per_sample_operators ? B : 1;
      const LindSmallRBufs rb = lind_smallr_carve(const_cast<void*>(tape), lind_smallr_sizes(B, K, N, Dm, segments, nsamp), nsamp, Dm, K);
      if (record_start(w, st)) return -1;
      if (lind_smallr_backward(w, rb, per_sample_operators != 0, signals, B, K, N, D, Dm, segments, fr_phase, (const cplx*)U_bar, grad_signals, st))
        return -1;
      if (record_stop(w, st)) return -1;
      g_last_kernel = Dm > 12 ? C3P_KERNEL_MFMA : C3P_KERNEL_SMALLD;
      return 0;
    }
    if (D == 4) return fail("the taped Lindblad evaluation at D = 4 needs C3P_HERMITIAN_H, as given to c3p_pwc_lindblad_taped");
    const LindSmallBufs bf = lind_small_carve(const_cast<void*>(tape), lind_small_sizes(B, K, N, Dm, segments, B));
    if (record_start(w, st)) return -1;
    if (lind_small_backward(w, bf, per_sample_operators != 0, signals, B, K, N, Dm, segments, fr_phase, (const cplx*)U_bar, grad_signals, st)) return -1;
    if (record_stop(w, st)) return -1;
    g_last_kernel = C3P_KERNEL_SMALLD;
    return 0;
  }
  if (!c3p_regr_supported(D, Dm)) return fail("the taped Lindblad evaluation serves D = 2, 3, 4 and D = 7, 8, 9, got D=%d", D);
  {
    // recomputed from the option table as it is NOW: an option that changes the segment count (or the kernel family) between
    // the taped forward call and this one would reinterpret the tape -- refuse what can be detected
    int seg_chk = 0;
    if (c3p_pwc_lindblad_tape_bytes(B, K, N, D, &seg_chk) == 0 || seg_chk != segments)
      return fail("taped Lindblad vjp: segment count %d != %d (c3p_pwc_lindblad_tape_bytes; were library options changed since the forward call?)", segments, seg_chk);
  }
  const LindRegrSizes z = lind_regr_sizes(B, K, N, Dm, segments, B);
  if (tape_bytes < z.total()) return fail("tape too small: %zu bytes, need %zu", tape_bytes, z.total());
  hipStream_t st = (hipStream_t)stream;
  WsLock lk(st);
  DeviceWs* w = lk.w;
  if (!w) return fail("no HIP device");
  if (!lk.ok) return fail("hipStreamWaitEvent on the previous call's stream failed");
  const LindRegrBufs bf = lind_regr_carve(const_cast<void*>(tape), z);
  if (record_start(w, st)) return -1;
  if (lind_regr_backward(w, bf, per_sample_operators != 0, signals, B, K, N, D, Dm, segments, fr_phase, (const cplx*)U_bar, grad_signals, st)) return -1;
  if (record_stop(w, st)) return -1;
  g_last_kernel = C3P_KERNEL_MFMA;
  return 0;
}

int c3p_gate_infid(const void* U, int B, int D, const int32_t* comp_rows, int L, const void* ideal, int kind, int flags,
                   double* infid_out, double* sum_out, void* stream) {
  if (B < 0 || D <= 0 || L <= 0 || L > D) return fail("bad sizes B=%d D=%d L=%d", B, D, L);
  if (kind != 0 && kind != 1) return fail("unknown infidelity kind %d (0 = unitary_infid, 1 = average_infid)", kind);
  if (!U || !comp_rows || !ideal || (!infid_out && !sum_out)) return fail("NULL pointer argument");
  hipStream_t st = (hipStream_t)stream;
  WsLock lk(st);
  DeviceWs* w = lk.w;
  if (!w) return fail("no HIP device");
  if (!lk.ok) return fail("hipStreamWaitEvent on the previous call's stream failed");
  const size_t cs = sizeof(cplx);
  Stage sg{w, st};
  const void *d_U = U, *d_rows = comp_rows, *d_G = ideal;
  void *d_inf = infid_out, *d_sum = sum_out;
  if (flags & C3P_HOST_PTRS) {
    for (int a = 0; a < L; ++a)
      if (comp_rows[a] < 0 || comp_rows[a] >= D) return fail("comp_rows[%d]=%d outside [0,%d)", a, comp_rows[a], D);
    if (sg.in(U, (size_t)B * D * D * cs, &d_U)) return -1;
    if (sg.in(comp_rows, (size_t)L * sizeof(int32_t), &d_rows)) return -1;
    if (sg.in(ideal, (size_t)L * L * cs, &d_G)) return -1;
    if (sg.out(infid_out, infid_out ? (size_t)B * sizeof(double) : 0, &d_inf)) return -1;
    if (sg.out(sum_out, sum_out ? 2 * sizeof(double) : 0, &d_sum)) return -1;
  }
  void* part;
  if (ws_get(w, SL_COUNTERS2, 256 * sizeof(double), &part)) return -1;
  if (B == 0) {
    if (d_sum) HIP_TRY(hipMemsetAsync(d_sum, 0, 2 * sizeof(double), st));
  } else {
    LAUNCH_TRY(c3p_launch_infid((const cplx*)d_U, B, D, (const int*)d_rows, L, (const cplx*)d_G, kind, (double*)d_inf,
                                (double*)part, (double*)d_sum, st));
  }
  if (flags & C3P_HOST_PTRS) return sg.finish();
  return 0;
}

int c3p_synth_signals(const double* env_params, const int32_t* env_shapes, const double* carrier,
                      double t_start, double t_end, double awg_res, double sim_res, int B, int K,
                      int E, int flags, double* awg_iq_out, double* signals_out, void* stream) {
  if (B < 0 || K <= 0 || E <= 0) return fail("bad sizes B=%d K=%d E=%d", B, K, E);
  if (!(awg_res > 0.0) || !(sim_res > 0.0)) return fail("resolutions must be positive");
  const double span = t_end > t_start ? t_end - t_start : t_start - t_end;
  const int N = (int)(span * sim_res), Na = (int)(span * awg_res);  // devices.py:72-84
  if (N <= 0 || Na <= 1) return fail("empty time grid: N=%d Na=%d", N, Na);
  if (B == 0) return 0;
  if (!env_params || !env_shapes || !carrier || !signals_out) return fail("NULL pointer argument");
  hipStream_t st = (hipStream_t)stream;
  WsLock lk(st);
  DeviceWs* w = lk.w;
  if (!w) return fail("no HIP device");
  if (!lk.ok) return fail("hipStreamWaitEvent on the previous call's stream failed");
  Stage sg{w, st};
  const void *d_env = env_params, *d_shape = env_shapes, *d_car = carrier;
  void *d_iq = awg_iq_out, *d_sig = signals_out;
  const size_t iq_bytes = (size_t)B * K * 2 * Na * sizeof(double);
  if (flags & C3P_HOST_PTRS) {
    for (int a = 0; a < K * E; ++a)
      if (env_shapes[a] >= C3P_ENV_NSHAPES) return fail("env_shapes[%d]=%d is not a C3P_ENV_* id", a, env_shapes[a]);
    if (sg.in(env_params, (size_t)B * K * E * C3P_ENV_NPAR * sizeof(double), &d_env)) return -1;
    if (sg.in(env_shapes, (size_t)K * E * sizeof(int32_t), &d_shape)) return -1;
    if (sg.in(carrier, (size_t)B * K * 2 * sizeof(double), &d_car)) return -1;
    if (sg.out(signals_out, (size_t)B * K * N * sizeof(double), &d_sig)) return -1;
    if (awg_iq_out && sg.out(awg_iq_out, iq_bytes, &d_iq)) return -1;
  }
  if (!d_iq) {
    if (ws_get(w, SL_SCRATCH, iq_bytes, &d_iq)) return -1;
  }
  SynthArgs A;
  A.env = (const double*)d_env;
  A.shape = (const int*)d_shape;
  A.carrier = (const double*)d_car;
  A.t_start = t_start;
  A.t_end = t_end;
  A.awg_res = awg_res;
  A.sim_res = sim_res;
  A.B = B;
  A.K = K;
  A.E = E;
  A.Na = Na;
  A.N = N;
  A.iq = (double*)d_iq;
  A.signals = (double*)d_sig;
  LAUNCH_TRY(c3p_launch_synth(A, st));
  if (flags & C3P_HOST_PTRS) return sg.finish();
  return 0;
}

// fused goal of c3p_pwc_unitary_goal_vjp (device pointers after staging)
struct GoalSpec {
  const int32_t* rows;
  int L;
  const void* ideal;
  int kind;
  double* infid;
  double* gphase;
  void* U_out;
};
static int unitary_vjp_branch_a(const void* h0, int64_t h0_bstride, const void* hks, int64_t hks_bstride, const double* signals,
                                double dt, int B, int K, int N, int D, int flags, const double* fr_phase, const void* U_bar,
                                double* grad_signals, void* gen_bar_out, void* stream, const GoalSpec* goal);

int c3p_pwc_unitary_vjp(const void* h0, int64_t h0_bstride, const void* hks, int64_t hks_bstride,
                        const double* signals, double dt, int B, int K, int N, int D, int flags,
                        const double* fr_phase, const void* U_bar, double* grad_signals, void* gen_bar_out,
                        void* stream) {
  if (flags & C3P_ORDER_RIGHT) return fail("c3p_pwc_unitary_vjp: unsupported flag");
  if (flags & C3P_PER_SLICE_H) {
    // branch B of pwc (propagation.py:295-308): the Hamiltonians are handed over per slice, so the result is the cotangent of
    // every slice GENERATOR G_n = -i dt H_n (gen_bar_out) -- what the tape propagates on into model.get_Hamiltonian
    if (B < 0 || N <= 0 || D <= 0) return fail("bad sizes B=%d N=%d D=%d", B, N, D);
    if (B == 0) return 0;
    if (!h0 || !U_bar || !gen_bar_out) return fail("per-slice gradient: h0 (the Hamiltonians), U_bar and gen_bar_out are required");
    const size_t cs = sizeof(cplx);
    hipStream_t st = (hipStream_t)stream;
    WsLock lk(st);
    DeviceWs* w = lk.w;
    if (!w) return fail("no HIP device");
    if (!lk.ok) return fail("hipStreamWaitEvent on the previous call's stream failed");
    Stage sg{w, st};
    const void *d_h = h0, *d_ph = fr_phase, *d_ub = U_bar;
    void* d_z = gen_bar_out;
    if (flags & C3P_HOST_PTRS) {
      if (sg.in(h0, (size_t)(h0_bstride ? B : 1) * N * D * D * cs, &d_h)) return -1;
      if (sg.in(U_bar, (size_t)B * D * D * cs, &d_ub)) return -1;
      if (fr_phase && sg.in(fr_phase, (size_t)B * D * sizeof(double), &d_ph)) return -1;
      if (sg.out(gen_bar_out, (size_t)B * N * D * D * cs, &d_z)) return -1;
    }
    if (D <= 40 && !(flags & C3P_FORCE_GENERIC) && !c3p_opt_on(C3P_OPT_tiled_grad)) {
      // on-chip general-generator sweeps (nothing assumed about the slice Hamiltonians), in chunks of samples that keep the
      // slice propagators + prefixes (2 N D^2 complex per sample) below 24 GB
      long Bc = (long)(grad_store_budget(w, SL_OUT1) / (2 * (size_t)N * D * D * cs + 1));
      if (c3p_opt(C3P_OPT_grad_chunk) > 0) Bc = c3p_opt(C3P_OPT_grad_chunk);
      if (Bc < 1) Bc = 1;
      int rc = 0;
      for (long b0 = 0; b0 < B && rc == 0; b0 += Bc) {
        const int nb = (int)(B - b0 < Bc ? B - b0 : Bc);
        rc = run_vjp_xg_general(w, (const cplx*)d_h + b0 * h0_bstride, h0_bstride, 0.0, -dt, nb, N, D,
                                d_ph ? (const double*)d_ph + b0 * D : nullptr, (const cplx*)d_ub + b0 * D * D,
                                (cplx*)d_z + b0 * (long)N * D * D, st);
      }
      if (rc < 0) return -1;
      if (rc == 0) {
        g_last_kernel = D <= kSmallDLimit ? C3P_KERNEL_SMALLD : C3P_KERNEL_MFMA;
        if (flags & C3P_HOST_PTRS) return sg.finish();
        return 0;
      }
    }
    g_last_kernel = C3P_KERNEL_MFMA;
    if (run_vjp_tiled(w, 0, (const cplx*)d_h, h0_bstride, nullptr, 0, nullptr, nullptr, dt, B, 0, N, D, D, (const double*)d_ph,
                      (const cplx*)d_ub, nullptr, st, true, (cplx*)d_z))
      return -1;
    if (flags & C3P_HOST_PTRS) return sg.finish();
    return 0;
  }
  if (!U_bar) return fail("NULL pointer argument");
  return unitary_vjp_branch_a(h0, h0_bstride, hks, hks_bstride, signals, dt, B, K, N, D, flags, fr_phase, U_bar, grad_signals,
                              gen_bar_out, stream, nullptr);
}

int c3p_pwc_unitary_goal_vjp(const void* h0, int64_t h0_bstride, const void* hks, int64_t hks_bstride, const double* signals,
                             double dt, int B, int K, int N, int D, int flags, const double* fr_phase, const int32_t* comp_rows,
                             int L, const void* ideal, int kind, double* infid_out, double* grad_signals, double* grad_fr_phase,
                             void* U_out, void* stream) {
  if (flags & (C3P_ORDER_RIGHT | C3P_PER_SLICE_H)) return fail("c3p_pwc_unitary_goal_vjp: unsupported flag");
  if (L <= 0 || L > D || L > C3P_GOAL_LMAX) return fail("bad computational subspace size L=%d (D=%d, at most %d)", L, D, C3P_GOAL_LMAX);
  if (kind != 0 && kind != 1) return fail("unknown infidelity kind %d", kind);
  if (!comp_rows || !ideal || !infid_out) return fail("NULL pointer argument");
  if (grad_fr_phase && !fr_phase) return fail("grad_fr_phase needs fr_phase");
  if (D > 64 || (D > 40 && (B >= 384 || c3p_opt_on(C3P_OPT_tiled_grad))))
    return fail("the fused goal runs on the on-chip and VALU backward sweeps (D <= 40, or D <= 64 below 384 samples): use "
                "c3p_pwc_unitary, c3p_gate_overlap and c3p_pwc_unitary_vjp for D=%d B=%d", D, B);
  if (flags & C3P_HOST_PTRS)
    for (int a = 0; a < L; ++a)
      if (comp_rows[a] < 0 || comp_rows[a] >= D) return fail("comp_rows[%d]=%d outside [0,%d)", a, comp_rows[a], D);
  GoalSpec g = {comp_rows, L, ideal, kind, infid_out, grad_fr_phase, U_out};
  return unitary_vjp_branch_a(h0, h0_bstride, hks, hks_bstride, signals, dt, B, K, N, D, flags, fr_phase, nullptr, grad_signals,
                              nullptr, stream, &g);
}

static int unitary_vjp_branch_a(const void* h0, int64_t h0_bstride, const void* hks, int64_t hks_bstride, const double* signals,
                                double dt, int B, int K, int N, int D, int flags, const double* fr_phase, const void* U_bar,
                                double* grad_signals, void* gen_bar_out, void* stream, const GoalSpec* goal) {
  if (B < 0 || K <= 0 || N <= 0 || D <= 0) return fail("bad sizes B=%d K=%d N=%d D=%d", B, K, N, D);
  if (D > 64 && gen_bar_out) return fail("gen_bar_out (per-slice generator cotangents) is available for D <= 64, got %d", D);
  if (B == 0) return 0;
  if (!h0 || !hks || !signals || (!U_bar && !goal) || !grad_signals) return fail("NULL pointer argument");
  if (h0_bstride < 0 || hks_bstride < 0) return fail("negative batch stride");
  const size_t cs = sizeof(cplx);
  hipStream_t st = (hipStream_t)stream;
  WsLock lk(st);
  DeviceWs* w = lk.w;
  if (!w) return fail("no HIP device");
  if (!lk.ok) return fail("hipStreamWaitEvent on the previous call's stream failed");
  Stage sg{w, st};
  const void *d_h0 = h0, *d_hks = hks, *d_sig = signals, *d_ph = fr_phase, *d_ub = U_bar;
  void* d_grad = grad_signals;
  void* d_zout = gen_bar_out;
  if (flags & C3P_HOST_PTRS) {
    // the adjoint recurrence M_n = dU_n^H M_{n+1} dU_n needs unitary slices: Hermitian generators
    const cplx* hh[2] = {(const cplx*)h0, (const cplx*)hks};
    const long cnt[2] = {h0_bstride ? (long)B : 1L, (hks_bstride ? (long)B : 1L) * K};
    for (int a = 0; a < 2; ++a)
      for (long m = 0; m < cnt[a]; ++m) {
        const cplx* H = hh[a] + m * D * D;
        double dev = 0.0, mag = 0.0;
        for (int i = 0; i < D; ++i)
          for (int j = 0; j < D; ++j) {
            dev = fmax(dev, hypot(H[i * D + j].x - H[j * D + i].x, H[i * D + j].y + H[j * D + i].y));
            mag = fmax(mag, hypot(H[i * D + j].x, H[i * D + j].y));
          }
        if (dev > 1e-9 * mag) return fail("c3p_pwc_unitary_vjp needs Hermitian Hamiltonians (deviation %.3g)", dev);
      }
    if (sg.in(h0, ((size_t)(B - 1) * (size_t)h0_bstride + (size_t)D * D) * cs, &d_h0)) return -1;
    if (sg.in(hks, ((size_t)(B - 1) * (size_t)hks_bstride + (size_t)K * D * D) * cs, &d_hks)) return -1;
    if (sg.in(signals, (size_t)B * K * N * sizeof(double), &d_sig)) return -1;
    if (U_bar && sg.in(U_bar, (size_t)B * D * D * cs, &d_ub)) return -1;
    if (fr_phase && sg.in(fr_phase, (size_t)B * D * sizeof(double), &d_ph)) return -1;
    if (sg.out(grad_signals, (size_t)B * K * N * sizeof(double), &d_grad)) return -1;
    if (gen_bar_out && sg.out(gen_bar_out, (size_t)B * N * D * D * cs, &d_zout)) return -1;
  }
  GoalSpec gd = {};
  if (goal) {
    gd = *goal;
    if (flags & C3P_HOST_PTRS) {
      const void* t;
      void* o;
      if (sg.in(goal->rows, (size_t)goal->L * sizeof(int32_t), &t)) return -1;
      gd.rows = (const int32_t*)t;
      if (sg.in(goal->ideal, (size_t)goal->L * goal->L * cs, &t)) return -1;
      gd.ideal = t;
      if (sg.out(goal->infid, (size_t)B * sizeof(double), &o)) return -1;
      gd.infid = (double*)o;
      if (goal->gphase) {
        if (sg.out(goal->gphase, (size_t)B * D * sizeof(double), &o)) return -1;
        gd.gphase = (double*)o;
      }
      if (goal->U_out) {
        if (sg.out(goal->U_out, (size_t)B * D * D * cs, &o)) return -1;
        gd.U_out = o;
      }
    }
  }
  GradArgs A = {};
  if (goal) {
    A.goal_rows = (const int*)gd.rows;
    A.goal_L = gd.L;
    A.goal_ideal = (const cplx*)gd.ideal;
    A.goal_kind = gd.kind;
    A.goal_infid = gd.infid;
    A.goal_gphase = gd.gphase;
    A.goal_U = (cplx*)gd.U_out;
  }
  A.h0 = (const cplx*)d_h0;
  A.h0_bstride = h0_bstride;
  A.hks = (const cplx*)d_hks;
  A.hks_bstride = hks_bstride;
  A.signals = (const double*)d_sig;
  A.fr_phase = (const double*)d_ph;
  A.Ubar = (const cplx*)d_ub;
  A.dt = dt;
  A.B = B;
  A.K = K;
  A.N = N;
  A.D = D;
  A.ld = D | 1;
  A.grad = (double*)d_grad;
  A.zout = (cplx*)d_zout;
  if (record_start(w, st)) return -1;
  bool done = false;
  if (!(flags & C3P_FORCE_GENERIC) && D <= kSmallDLimit && c3p_smalld_supported(D) && K <= 8) {
    const int rc = run_vjp_smalld(w, A, st);
    if (rc < 0) return -1;
    done = (rc == 0);
    if (done) g_last_kernel = C3P_KERNEL_SMALLD;
  }
  if (!done && !(flags & C3P_FORCE_GENERIC) && D >= 13 && D <= 40) {
    const int rc = run_vjp_midd(w, A, st);
    if (rc < 0) return -1;
    done = (rc == 0);
    if (done) g_last_kernel = C3P_KERNEL_MFMA;
  }
  // beyond the on-chip sweeps: forward partials in HBM, one pair evaluation of T18 per slice on the tiled MFMA GEMM.  ~35
  // launches per slice: below a few hundred samples the launches, not the GEMMs, set the time, and the VALU sweep (which
  // stops at D = 64) is faster for 41 <= D <= 64 (profiles/r03/grad_tiled.json: D = 48, B = 64: 151 ms against 545 ms)
  // (the fused goal entry has no tiled form: it stays on the VALU sweep in its whole domain)
  const bool tiled_grad = goal ? false : tiled_unitary_grad(D, B);
  if (!done && !(flags & C3P_FORCE_GENERIC) && tiled_grad && !gen_bar_out) {
    if (run_vjp_tiled(w, 0, A.h0, h0_bstride, A.hks, hks_bstride, A.signals, nullptr, dt, B, K, N, D, D, A.fr_phase, A.Ubar, A.grad, st))
      return -1;
    done = true;
    g_last_kernel = C3P_KERNEL_MFMA;
  }
  if (!done && D > 64) return fail("the VALU gradient kernels support D <= 64, got %d", D);
  if (!done) {
    long S = 4096 / B;
    if (S > N / 8) S = N / 8;
    if (S < 1) S = 1;
    A.S = (int)S;
    void* v;
    if (ws_get(w, SL_SEG_A, (size_t)B * A.S * D * D * cs, &v)) return -1;
    A.seg = (cplx*)v;
    if (ws_get(w, SL_SEG_B, (size_t)B * A.S * D * D * cs, &v)) return -1;
    A.Mb = (cplx*)v;
    const bool global = c3p_grad_lds_bytes(D) > 150 * 1024;
    if (global) {
      A.scratch_stride = (long)C3P_GRAD_NMAT * A.ld * D;
      if (ws_get(w, SL_SCRATCH, (size_t)B * A.S * A.scratch_stride * cs, &v)) return -1;
      A.scratch = (cplx*)v;
    }
    g_last_kernel = global ? C3P_KERNEL_GENERIC_GLOBAL : C3P_KERNEL_GENERIC_LDS;
    LAUNCH_TRY(c3p_launch_grad_seg(A, global, st));
    LAUNCH_TRY(c3p_launch_grad_scan(A, global, st));
    LAUNCH_TRY(c3p_launch_grad_bwd(A, global, st));
  }
  if (record_stop(w, st)) return -1;
  if (flags & C3P_HOST_PTRS) return sg.finish();
  return 0;
}

int c3p_synth_signals_vjp(const double* env_params, const int32_t* env_shapes, const double* carrier,
                          double t_start, double t_end, double awg_res, double sim_res, int B, int K,
                          int E, int flags, const double* grad_signals, double* grad_env, double* grad_carrier,
                          void* stream) {
  if (B < 0 || K <= 0 || E <= 0) return fail("bad sizes B=%d K=%d E=%d", B, K, E);
  if (!(awg_res > 0.0) || !(sim_res > 0.0)) return fail("resolutions must be positive");
  const double span = t_end > t_start ? t_end - t_start : t_start - t_end;
  const int N = (int)(span * sim_res), Na = (int)(span * awg_res);
  if (N <= 0 || Na <= 1) return fail("empty time grid: N=%d Na=%d", N, Na);
  if (B == 0) return 0;
  if (!env_params || !env_shapes || !carrier || !grad_signals || !grad_env || !grad_carrier)
    return fail("NULL pointer argument");
  hipStream_t st = (hipStream_t)stream;
  WsLock lk(st);
  DeviceWs* w = lk.w;
  if (!w) return fail("no HIP device");
  if (!lk.ok) return fail("hipStreamWaitEvent on the previous call's stream failed");
  Stage sg{w, st};
  const void *d_env = env_params, *d_shape = env_shapes, *d_car = carrier, *d_gs = grad_signals;
  void *d_ge = grad_env, *d_gc = grad_carrier;
  if (flags & C3P_HOST_PTRS) {
    for (int a = 0; a < K * E; ++a)
      if (env_shapes[a] >= C3P_ENV_NSHAPES) return fail("env_shapes[%d]=%d is not a C3P_ENV_* id", a, env_shapes[a]);
    if (sg.in(env_params, (size_t)B * K * E * C3P_ENV_NPAR * sizeof(double), &d_env)) return -1;
    if (sg.in(env_shapes, (size_t)K * E * sizeof(int32_t), &d_shape)) return -1;
    if (sg.in(carrier, (size_t)B * K * 2 * sizeof(double), &d_car)) return -1;
    if (sg.in(grad_signals, (size_t)B * K * N * sizeof(double), &d_gs)) return -1;
    if (sg.out(grad_env, (size_t)B * K * E * C3P_ENV_NPAR * sizeof(double), &d_ge)) return -1;
    if (sg.out(grad_carrier, (size_t)B * K * 2 * sizeof(double), &d_gc)) return -1;
  }
  // scratch: iq | giq | gcar_part, each B*K*2*Na doubles
  const size_t part = (size_t)B * K * 2 * Na;
  void* v;
  if (ws_get(w, SL_SCRATCH, 3 * part * sizeof(double), &v)) return -1;
  SynthArgs A;
  A.env = (const double*)d_env;
  A.shape = (const int*)d_shape;
  A.carrier = (const double*)d_car;
  A.t_start = t_start;
  A.t_end = t_end;
  A.awg_res = awg_res;
  A.sim_res = sim_res;
  A.B = B;
  A.K = K;
  A.E = E;
  A.Na = Na;
  A.N = N;
  A.iq = (double*)v;
  A.signals = nullptr;
  LAUNCH_TRY(c3p_launch_synth_vjp(A, (const double*)d_gs, A.iq + part, A.iq + 2 * part, (double*)d_ge, (double*)d_gc, st));
  if (flags & C3P_HOST_PTRS) return sg.finish();
  return 0;
}

}  // extern "C"
