/* c3prop.h -- C ABI of libc3prop.so, the MI355X (gfx950) implementation of
 * C3's piecewise-constant propagator hot path.
 *
 * The reference (q-optimize/c3) is pure Python/TensorFlow: it has no FFI for
 * this path.  The boundary it exposes is the Python plugin slot
 * `Experiment.set_prop_method(callable)` (c3/experiment.py:76-91) whose callable
 * is invoked as `propagation(model, gen, instr, folding_stack, batch_size)`
 * (c3/experiment.py:472-478).  The numerical inner boundary that this library
 * replaces is
 *
 *     dUs = tf_batch_propagate(h0, hks, signals, dt, batch_size, col_ops, lindbladian)
 *                                               (c3/libraries/propagation.py:460-515)
 *     U   = tf_matmul_n(dUs, folding_stack)     (c3/utils/tf_utils.py:144-163)
 *
 * plus the ODE state solver `ode_solver` (c3/libraries/propagation.py:687-752).
 * Each entry point below cites the reference function it stands in for.
 * `INTEGRATION.md` shows the ctypes binding a C3 maintainer would add.
 *
 * Conventions
 *  - complex128 = interleaved (re, im) doubles, C (row-major) order -- numpy /
 *    TensorFlow layout.  Pointers are typed `const void*` / `void*` for those.
 *  - All pointers are DEVICE pointers unless C3P_HOST_PTRS is set in `flags`,
 *    in which case the library stages inputs/outputs through its own device
 *    buffers (synchronous).
 *  - `stream` is a hipStream_t (NULL = the default stream).  Device-pointer
 *    calls are asynchronous with respect to the host.
 *  - Return value 0 = success; negative = error, message via c3p_last_error()
 *    (thread-local).  The Python wrapper re-raises `Exception("C3:Error: ...")`
 *    like the reference (c3/experiment.py:465-468).
 *  - The caller owns every buffer it passes.  The library owns only a lazily
 *    grown per-device workspace, released by c3p_shutdown().
 *  - Thread-safe.  The workspace is ONE set per device, guarded by a per-device
 *    mutex at enqueue time (different devices do not serialise each other).  Calls
 *    on one stream are ordered by the stream; a call on ANOTHER stream of the same
 *    device first makes its stream wait (hipStreamWaitEvent) for an event recorded,
 *    at that moment, on the stream of the previous call (a one-stream caller pays no
 *    event per call), so two streams never run kernels on the shared workspace at
 *    once -- results are correct from any number of streams, but calls on one
 *    device do not overlap each other.
 */
#ifndef C3PROP_H
#define C3PROP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* the library is built with -fvisibility=hidden: exactly the entry points declared here are exported */
#pragma GCC visibility push(default)

/* flags */
#define C3P_HOST_PTRS 0x1     /* all data pointers are host memory                    */
#define C3P_PER_SLICE_H 0x2   /* h0 is [N,D,D] (or [B,N,D,D] with h0_bstride): branch B
                                 of pwc, use_control_fields=False (propagation.py:295-308) */
#define C3P_ORDER_RIGHT 0x4   /* c3p_matmul_chain: elems[0]@...@elems[N-1] (tf_matmul_right) */
#define C3P_FORCE_GENERIC 0x8 /* use the generic LDS kernel even where a specialised one exists */
#define C3P_HERMITIAN_H 0x10  /* c3p_pwc_lindblad: the caller DECLARES h0 and every hk Hermitian (the library does not check
                                 device memory).  The Lindblad generator is then real in a basis of Hermitian matrices and the
                                 chain of one qubit / qutrit or two qubits (D = 2, 3, 4) runs in real arithmetic (c3p_smallr.hip); other shapes
                                 ignore the flag (D = 7, 8, 9 detect the case on the device).  Declared wrongly: wrong results.
                                 The reference makes no such distinction (propagation.py:551-585): c3_amd/propagation.py sets
                                 the flag after checking the arrays. */

/* kernels selected (returned by c3p_last_kernel, for tests/bench reporting) */
#define C3P_KERNEL_NONE 0
#define C3P_KERNEL_GENERIC_LDS 1
#define C3P_KERNEL_GENERIC_GLOBAL 2
#define C3P_KERNEL_SMALLD 3
#define C3P_KERNEL_MFMA 4
#define C3P_KERNEL_ODE_WG 5  /* workgroup-per-sample ODE kernel (c3p_ode.hip)            */
#define C3P_KERNEL_ODE_ROW 6 /* lane-row ODE kernels (c3p_ode_row.hip, c3p_ode_rowq.hip)  */
#define C3P_KERNEL_ODE_MFMA 7 /* matrix-core rho-valued ODE kernel, 17 <= D <= 48 (c3p_ode_rhoq.hip) */
#define C3P_KERNEL_ODE_ROW_OR_WG 8 /* both launched; the DEVICE picks: real operators -> lane rows, complex -> workgroup kernel */

/* ODE solver / step ids (propagation.py:27-32 solver_slicing; :886-904 steps) */
#define C3P_SOLVER_RK4 0
#define C3P_SOLVER_RK38 1
#define C3P_SOLVER_RK5 2
#define C3P_SOLVER_TSIT5 3
#define C3P_STEP_SCHRODINGER 0
#define C3P_STEP_VON_NEUMANN 1
#define C3P_STEP_LINDBLAD 2

/* Tuning / diagnostic options: one process-wide table of integers (kernel selection switches, segment counts; the list
 * with one line per option is C3P_OPTION_LIST in c3_amd/csrc/c3p_kernels.h).  `value` is a decimal integer, "all" (= 2)
 * or NULL / "unset" (back to the default, -1).  The table is initialised ONCE, at the first use, from environment
 * variables of the same names in upper case with the prefix C3P_ (C3P_NO_REGD=1 python ...): later changes of the
 * environment are not seen -- use this call.  Thread-safe (atomic loads / stores); a call in flight on another thread
 * sees either value.  No reference counterpart (the reference has no kernel selection).  Unknown name: -1 and
 * c3p_last_error(); c3p_get_option returns -2 for an unknown name, -1 for an option that is not set. */
int c3p_set_option(const char* name, const char* value);
long c3p_get_option(const char* name);

int c3p_version(void);
int c3p_device_count(void);
const char* c3p_last_error(void);
int c3p_last_kernel(void);
/* The kernels the most recent compute call of this thread launched, as "file: kernel<template arguments> xCOUNT; ..." in
 * launch order (names resolved from the host function pointers: what ran, not what was planned).  Writes at most cap - 1
 * characters + NUL into buf (may be NULL) and returns the full length.  Diagnostic: INTEGRATION.md's dispatch table is
 * generated from it (tools/dispatch_table.py).  No reference counterpart. */
int c3p_last_kernel_detail(char* buf, int cap);
/* Device time (ms) of the most recent c3p_pwc_* call's main kernel, measured
 * with hipEvents on the stream it was launched on; valid after that stream has
 * been synchronised.  Only recorded when profiling was enabled with
 * c3p_set_profiling(1).  Returns < 0 if nothing was recorded. */
int c3p_set_profiling(int enable); /* per device: acts on the current device's workspace */
double c3p_last_kernel_ms(void);
void c3p_shutdown(void);

/* Workspace management.  The library owns one workspace per device (segment products, generator tables, arenas), grown
 * lazily by the first call that needs more -- which synchronises the device, frees and allocates (not legal inside a
 * stream capture, and a latency cliff in a serving loop).  c3p_reserve sizes it for one call shape ahead of time
 * (arguments as c3p_pwc_unitary / c3p_pwc_lindblad; device-pointer calls only): afterwards calls of that shape, or of
 * any shape that needs no more, neither allocate nor synchronise, and ONE c3p_pwc_* call can be captured into a hipGraph
 * (a call whose workspace would have to grow during a capture fails with an error instead).
 * c3p_workspace_generation counts (re)allocations of the current device's workspace: constant = steady state.
 * Reference analogue: none (TensorFlow owns its allocator; nothing on this path is retained between calls). */
int c3p_reserve(int lindblad, int B, int K, int N, int D, int C, int flags);
long c3p_workspace_generation(void);

/* U[b] = diag(exp(i fr_phase[b])) * prod_n exp(-i (h0 + sum_k signals[b,k,n] hks[k]) dt)
 *
 * Replaces, for B independent parameter samples at once:
 *   tf_propagation_vectorized (propagation.py:426-440) via tf_batch_propagate (:460-515),
 *   tf_matmul_n (tf_utils.py:144-193) / tf_matmul_left (:120-129),
 *   and the frame-rotation left-multiply of compute_propagators (experiment.py:482-509).
 *
 *   h0       c128 [D,D]; with C3P_PER_SLICE_H: [N,D,D]; element stride between samples
 *            h0_bstride (0 = shared by all samples)
 *   hks      c128 [K,D,D] (ignored when K == 0); stride between samples hks_bstride (0 = shared)
 *   signals  f64  [B,K,N]  real control amplitudes (propagation.py:293 casts them to c128)
 *   fr_phase f64  [B,D] or NULL
 *   U_out    c128 [B,D,D]
 *   dUs_out  c128 [B,N,D,D] or NULL (partial propagators, experiment.py:523-533)
 */
int c3p_pwc_unitary(const void* h0, int64_t h0_bstride, const void* hks, int64_t hks_bstride,
                    const double* signals, double dt, int B, int K, int N, int D, int flags,
                    const double* fr_phase, void* U_out, void* dUs_out, void* stream);

/* Lindblad superoperator propagator (propagation.py:551-585):
 *   L[n] = -i (H[n] (x) I - I (x) H[n]^T) + sum_c [ (C(x)I)(I(x)C^T)^+ - 1/2 (C(x)I)^+(C(x)I) - 1/2 (I(x)C^T)(I(x)C^T)^+ ]
 *   U[b] = diag(exp(i fr_phase[b])) * prod_n exp(L[n] dt),   matrices are [D*D, D*D]
 *   col_ops  c128 [C,D,D]
 *   fr_phase f64 [B,D*D] or NULL (phases of tf_super(FR), experiment.py:499-503)
 *   U_out    c128 [B,D*D,D*D];  dUs_out c128 [B,N,D*D,D*D] or NULL
 */
int c3p_pwc_lindblad(const void* h0, int64_t h0_bstride, const void* hks, int64_t hks_bstride,
                     const double* signals, const void* col_ops, int C, double dt, int B, int K,
                     int N, int D, int flags, const double* fr_phase, void* U_out, void* dUs_out,
                     void* stream);

/* out[i] = expm(A[i]) for a batch of n [D,D] matrices: the tf.linalg.expm call sites
 * (propagation.py:378,422,440,456,584). */
int c3p_expm(const void* A, int n, int D, int flags, void* out, void* stream);

/* Ordered product of a list: out[b] = M[b,N-1] @ ... @ M[b,1] @ M[b,0]
 * (tf_matmul_n tf_utils.py:144-163 and tf_matmul_left :120-129; with C3P_ORDER_RIGHT
 * tf_matmul_right :132-141).  M is c128 [B,N,D,D]. */
int c3p_matmul_chain(const void* M, int B, int N, int D, int flags, void* out, void* stream);

/* Superoperator builders (tf_utils.py:257-289) on a batch of n [D,D] matrices:
 *   which = 0: spre(A) = A (x) I;  1: spost(A) = I (x) A^T;  2: super(A) = spre(A) spost(A^+) = A (x) conj(A)
 *   out c128 [n, D*D, D*D] */
int c3p_superop(const void* A, int n, int D, int which, int flags, void* out, void* stream);
/* out[i] = kron(A[i], B[i]);  A [n,Da,Da], B [n,Db,Db], out [n,Da*Db,Da*Db] (tf_kron, tf_utils.py:257-267) */
int c3p_kron(const void* A, const void* Bm, int n, int Da, int Db, int flags, void* out, void* stream);

/* ODE state solver (propagation.py:687-752 + model.py:641-697 + tf_utils.py:521-559):
 * integrates B independent samples with N RK steps each.
 *   solver   C3P_SOLVER_*;  step C3P_STEP_* (LINDBLAD needs col_ops, C > 0)
 *   ts0, dt  first sample time and spacing of the (uniform) signal grid
 *   init     c128 [B, D, M] with M = 1 (state vector, SCHRODINGER) or M = D (density matrix);
 *            init_bstride = 0 shares one initial state
 *   states   c128 [B,N,D,M] if want_all else [B,D,M]
 */
int c3p_ode_solve(const void* h0, const void* hks, const double* signals, const void* col_ops,
                  int C, double dt, int B, int K, int N, int D, int solver, int step,
                  const void* init, int64_t init_bstride, int want_all, int flags, void* states,
                  void* stream);

/* RK4 "unitary" provider (propagation.py:71-101,221-255: rk4_unitary, gen_u_rk4, gen_dus_rk4,
 * gen_du_rk4, rk4_step; Hamiltonians from get_hs_of_t_ts :104-204 at prop_res = 2):
 * every RK4 step consumes three consecutive Hamiltonian samples h[2j], h[2j+1], h[2j+2].
 *   signals  f64 [B,K,Ns] on the doubled-resolution grid (branch A), or
 *   hs       c128 [Ns,D,D] / [B,Ns,D,D] per-sample Hamiltonians (branch B; hs_bstride 0 = shared)
 *   dt       the full RK step (ts[prop_res] - ts[0])
 *   U_out    c128 [B,D,D]: columns are the propagated basis vectors (gen_u_rk4 :246-255)
 *   dUs_out  c128 [B,(Ns-1)/2,D,D] or NULL: per-step maps with propagated basis vectors as ROWS,
 *            exactly as gen_du_rk4 (:85-92) stacks them
 */
int c3p_rk4_unitary(const void* h0, const void* hks, const double* signals, const void* hs,
                    int64_t hs_bstride, double dt, int B, int K, int Ns, int D, int flags,
                    void* U_out, void* dUs_out, void* stream);

/* Gradient of the propagators with respect to the control samples.  The reference obtains
 * it by taping the goal function (c3/optimizers/optimizer.py:206-216, optimalcontrol.py:219-226); the
 * taped path is c3p_pwc_unitary's (propagation.py:426-440, tf_utils.py:144-193, experiment.py:482-509).
 * Vector-Jacobian product: with d loss = Re sum_ij conj(U_bar[b,i,j]) dU[b,i,j],
 *   grad_signals[b,k,n] = d loss / d signals[b,k,n]      (exact: Frechet derivative of every slice)
 *   U_bar c128 [B,D,D]; grad_signals f64 [B,K,N]; other arguments as c3p_pwc_unitary (branch A only,
 *   Hermitian h0 / hks: the adjoint sweep uses the unitarity of the slices; checked for host pointers).
 *   gen_bar_out c128 [B,N,D,D] or NULL (D <= 64): Z[b,n], the cotangent of the slice generator G_n = -i dt H_n
 *   (d loss = Re sum conj(Z) dG); the gradient w.r.t. MODEL parameters follows by contraction, e.g.
 *   d loss / d h0 = i dt sum_n Z[b,n] (what ModelLearning differentiates, c3/optimizers/modellearning.py:300-341).
 * With C3P_PER_SLICE_H (branch B of pwc, propagation.py:295-308): h0 = the per-slice Hamiltonians [N,D,D] (h0_bstride 0) or
 *   [B,N,D,D], hks / signals / grad_signals NULL, K = 0, and gen_bar_out [B,N,D,D] (required) receives the cotangent of every
 *   slice generator G_n = -i dt H_n, i.e. d loss / d H_n = i dt Z[b,n]: what the tape hands back to model.get_Hamiltonian.
 *   Any dimension: on-chip general-generator sweeps up to D = 40 (nothing assumed about the Hamiltonians), the tiled
 *   backward sweep above.
 */
int c3p_pwc_unitary_vjp(const void* h0, int64_t h0_bstride, const void* hks, int64_t hks_bstride,
                        const double* signals, double dt, int B, int K, int N, int D, int flags,
                        const double* fr_phase, const void* U_bar, double* grad_signals, void* gen_bar_out,
                        void* stream);

/* One optimiser evaluation of a closed-system gate goal, fused: the goal AND its gradient w.r.t. the control samples from ONE
 * pass over the chains.  The reference evaluates goal_run = fid_func(compute_propagators()) (c3/optimizers/optimalcontrol.py:
 * 200-228) under a GradientTape (c3/optimizers/optimizer.py:206-216); the three-call form here is c3p_pwc_unitary ->
 * c3p_gate_overlap (+ the cotangent, element-wise) -> c3p_pwc_unitary_vjp, whose third call recomputes the segment products of
 * the first.  This entry forms them once: the per-sample scan of the backward pass, which multiplies the segment products
 * anyway, evaluates infid[b] of U = FR P and takes its cotangent
 *   U_bar[b] = c s G on (comp_rows x comp_rows),  s = tr(G^+ U[rows,rows]),  c = -2/L^2 (kind 0, unitary_infid fidelities.py:
 *   154-184) or -2/(L(L+1)) (kind 1, average_infid :290-313)
 * as the start of the adjoint sweep.  Arguments as c3p_pwc_unitary_vjp (branch A only, Hermitian h0 / hks) and c3p_gate_infid:
 *   infid_out f64 [B]; grad_signals f64 [B,K,N] = d infid[b] / d signals[b,k,n];
 *   grad_fr_phase f64 [B,D] or NULL = d infid[b] / d fr_phase[b,i] (needs fr_phase); U_out c128 [B,D,D] or NULL.
 * On the on-chip and VALU backward sweeps (D <= 40, or D <= 64 below 384 samples); other shapes: error, use the three calls. */
int c3p_pwc_unitary_goal_vjp(const void* h0, int64_t h0_bstride, const void* hks, int64_t hks_bstride,
                             const double* signals, double dt, int B, int K, int N, int D, int flags,
                             const double* fr_phase, const int32_t* comp_rows, int L, const void* ideal, int kind,
                             double* infid_out, double* grad_signals, double* grad_fr_phase, void* U_out, void* stream);

/* The same vector-Jacobian product through the LINDBLAD path (c3p_pwc_lindblad; the reference tapes
 * tf_propagation_lind just as well, c3/libraries/propagation.py:551-585 under c3/optimizers/optimizer.py:206-216):
 *   U_bar c128 [B,D^2,D^2] (d loss = Re sum conj(U_bar) dU of the superoperators), grad_signals f64 [B,K,N],
 *   fr_phase f64 [B,D^2] or NULL (the row phases c3p_pwc_lindblad applies).
 * The slices are not unitary, so the adjoint state cannot be propagated backwards through inverses: the forward partial
 * products are kept in HBM (N matrices of D^4 complex per sample, processed in chunks of samples that fit 24 GB) and the
 * backward sweep evaluates value and Frechet derivative of every slice's exponential together on the tiled MFMA GEMM
 * (c3p_tiled.hip).  c3p_pwc_unitary_vjp uses the same sweep above D = 40 (any dimension; gen_bar_out up to D = 64: above 40 on the VALU sweep).
 * D = 7, 8, 9 (49 x 49 .. 81 x 81 superoperators, Hermitian Hamiltonians): the whole evaluation in REAL arithmetic in the Hermitian
 * basis -- forward chain kernel keeping the transposed local prefix of every slice, real segment scan, on-chip backward sweep
 * (c3p_regrg.hip); the call synchronises the stream once (it reads back whether every Hamiltonian is Hermitian: a complex
 * generator in that basis falls back to the tiled sweep).
 * D <= 6 (superoperators up to 36 x 36): three kernels per call instead of ~35 launches per slice -- segment products, a scan
 * that leaves prefix and left adjoint at the segment boundaries, and a sweep that stores the prefix of every slice of its
 * segment and evaluates the pair at X_n^H on the way back (general-generator form on the matrix cores: c3p_smalld.hip for
 * D <= 3, c3p_midd.hip for D = 4 .. 6; c3p_grad.hip is the VALU fallback; C3P_TILED_GRAD=1 selects the tiled sweep). */
int c3p_pwc_lindblad_vjp(const void* h0, int64_t h0_bstride, const void* hks, int64_t hks_bstride,
                         const double* signals, const void* col_ops, int C, double dt, int B, int K, int N, int D,
                         int flags, const double* fr_phase, const void* U_bar, double* grad_signals, void* stream);

/* Open-system optimiser evaluation from ONE forward pass (D = 7, 8, 9: 49 x 49 .. 81 x 81 superoperators, Hermitian Hamiltonians;
 * D = 2, 3: 4 x 4 / 9 x 9 superoperators on the small-D kernels, any Hamiltonian, up to 8 control lines -- there the tape holds
 * the generator tables, the segment products and the slice propagators; D = 4 (two qubits) with C3P_HERMITIAN_H in `flags` of
 * BOTH calls: the real Hermitian-basis kernels -- the flag selects the layout of the tape at D = 2, 3 as well).
 * c3p_pwc_lindblad followed by c3p_pwc_lindblad_vjp computes the chain twice (the second time with the per-slice prefixes the
 * backward sweep reads).  Here the forward call records what the backward pass needs -- the generator tables in the Hermitian
 * basis, the real segment products, the transposed local prefix of every slice -- on a TAPE the caller owns (device memory, as
 * the reference's GradientTape keeps the forward intermediates of tf_propagation_lind, c3/libraries/propagation.py:551-585 under
 * c3/optimizers/optimizer.py:206-216), and the vector-Jacobian product runs from the tape: no second forward pass, nothing of
 * the evaluation lives in the library's shared workspace between the two calls.
 *   c3p_pwc_lindblad_tape_bytes: bytes of the tape for a shape (0: shape not served) and the segment count it is laid out for
 *     (pass it to both calls unchanged: both calls recompute it and fail on a mismatch, for every D.  The layout of the tape
 *     also follows from the library's option table -- c3p_set_option must not be called between the taped forward call and
 *     its vjp; a change that alters the segment count is refused by the vjp call);
 *   c3p_pwc_lindblad_taped: arguments and U_out as c3p_pwc_lindblad (device pointers, flags = 0, no dUs_out);
 *   c3p_pwc_lindblad_vjp_taped: U_bar, fr_phase, grad_signals as c3p_pwc_lindblad_vjp; signals = the ones the tape was recorded
 *     with; per_sample_operators = whether h0 / hks had a batch stride.
 * At D = 7, 8, 9 a non-Hermitian Hamiltonian is an error of the taped forward call (use the untaped pair). */
size_t c3p_pwc_lindblad_tape_bytes(int B, int K, int N, int D, int* segments_out);
int c3p_pwc_lindblad_taped(const void* h0, int64_t h0_bstride, const void* hks, int64_t hks_bstride, const double* signals,
                           const void* col_ops, int C, double dt, int B, int K, int N, int D, int flags, const double* fr_phase,
                           void* U_out, void* tape, size_t tape_bytes, int segments, void* stream);
int c3p_pwc_lindblad_vjp_taped(const void* tape, size_t tape_bytes, int segments, int per_sample_operators, const double* signals,
                               int B, int K, int N, int D, int flags, const double* fr_phase, const void* U_bar,
                               double* grad_signals, void* stream);

/* Control-signal synthesis for the standard drive line LO + AWG -> DAC -> Mixer -> VoltsToHertz
 * (Instruction.get_awg_signal c3/signal/gates.py:341-370, Envelope/EnvelopeDrag
 * c3/signal/pulse.py:88-180, Device.create_ts c3/generator/devices.py:72-122, DigitalToAnalog :306-351,
 * Mixer :914-939, LO :1073-1130 noiseless branch, VoltsToHertz :203-221, envelope shapes
 * c3/libraries/envelopes.py:26,195,201,228,254,374,401,421,470).  A batch is described by B x K x E envelope
 * parameter rows instead of B x K x N samples:
 *   signals[b,k,n] = v2hz * (cos(w t_n) I(t_n) + sin(w t_n) Q(t_n)),
 *   I + iQ = nearest-neighbour upsampling of sum_e amp_e env_e(t - t0_e) exp(i (xy_e - fo_e (t - t0_e)))
 *   sampled on the AWG grid, env_e = mask * (shape - shape(t_before)) [+ i DRAG term].
 *   env_params  f64 [B,K,E,C3P_ENV_NPAR]   rows laid out by the C3P_ENV_* slots below
 *   env_shapes  int32 [K,E]                C3P_ENV_* shape ids, negative = unused slot
 *   carrier     f64 [B,K,2]                {LO angular frequency [rad/s], V_to_Hz factor}
 *   N = (int)(|t_end - t_start| * sim_res), Na = (int)(|t_end - t_start| * awg_res)
 *   awg_iq_out  f64 [B,K,2,Na] or NULL     AWG-resolution inphase, quadrature
 *   signals_out f64 [B,K,N]                what c3p_pwc_* take as `signals`
 */
#define C3P_ENV_NO_DRIVE 0
#define C3P_ENV_RECT 1
#define C3P_ENV_GAUSSIAN_NONORM 2
#define C3P_ENV_FLATTOP 3
#define C3P_ENV_FLATTOP_RISEFALL 4
#define C3P_ENV_COSINE 5
#define C3P_ENV_GAUSSIAN_SIGMA 6 /* envelopes.py:374-398 */
#define C3P_ENV_GAUSSIAN 7       /* envelopes.py:401-418: sigma = t_final / 6 */
#define C3P_ENV_TRAPEZOID 8      /* envelopes.py:201-225 */
#define C3P_ENV_NSHAPES 9

#define C3P_ENV_AMP 0
#define C3P_ENV_XY_ANGLE 1
#define C3P_ENV_FREQ_OFFSET 2
#define C3P_ENV_DELTA 3
#define C3P_ENV_T_FINAL 4
#define C3P_ENV_SIGMA 5
#define C3P_ENV_T_UP 6
#define C3P_ENV_T_DOWN 7
#define C3P_ENV_RISEFALL 8
#define C3P_ENV_DELAY 9
#define C3P_ENV_FLAGS 10 /* C3P_ENVF_* bits stored as a double */
#define C3P_ENV_NPAR 12
#define C3P_ENVF_T_BEFORE 1 /* Envelope(use_t_before=True) */
#define C3P_ENVF_DRAG 2     /* EnvelopeDrag: imag = -delta * dt * d env/dt */

int c3p_synth_signals(const double* env_params, const int32_t* env_shapes, const double* carrier,
                      double t_start, double t_end, double awg_res, double sim_res, int B, int K,
                      int E, int flags, double* awg_iq_out, double* signals_out, void* stream);

/* Vector-Jacobian product of c3p_synth_signals for the commonly optimised pulse parameters (the same
 * GradientTape of optimizers/optimizer.py:206-216 covers Instruction.get_awg_signal gates.py:341-370):
 *   grad_signals f64 [B,K,N]             d loss / d signals (e.g. from c3p_pwc_unitary_vjp)
 *   grad_env     f64 [B,K,E,C3P_ENV_NPAR] d loss / d {amp, xy_angle, freq_offset, delta} in their slots,
 *                                        0 in every other slot (times, sigma, flags are not differentiated)
 *   grad_carrier f64 [B,K,2]             d loss / d {LO angular frequency, V_to_Hz}
 */
int c3p_synth_signals_vjp(const double* env_params, const int32_t* env_shapes, const double* carrier,
                          double t_start, double t_end, double awg_res, double sim_res, int B, int K,
                          int E, int flags, const double* grad_signals, double* grad_env, double* grad_carrier,
                          void* stream);

/* Fidelity epilogue: overlap[b] = tr(P^T U[b] P G^+), the number behind
 * unitary_infid = 1 - |overlap/L|^2 (c3/libraries/fidelities.py:154-184, tf_unitary_overlap
 * c3/utils/tf_utils.py:330-366) and average_infid = 1 - (|overlap|^2/L + 1)/(L + 1)
 * (fidelities.py:290-313, tf_average_fidelity tf_utils.py:380-401).  The projector of
 * tf_project_to_comp (tf_utils.py:428-436, qt_utils.py:178-193) is a 0/1 selection, passed
 * as the L row indices `comp_rows` of the computational states in the full space.
 *   U c128 [B,D,D]; comp_rows int32 [L]; ideal c128 [L,L]; overlap_out c128 [B]
 */
int c3p_gate_overlap(const void* U, int B, int D, const int32_t* comp_rows, int L, const void* ideal,
                     int flags, void* overlap_out, void* stream);

/* The goal itself, fused: infid[b] = unitary_infid (kind 0: 1 - |overlap / L|^2, fidelities.py:154-184)
 * or average_infid (kind 1: 1 - (|overlap|^2 / L + 1) / (L + 1), fidelities.py:290-313) of every propagator and their sum
 * over the batch -- B scalars (or two) leave the device instead of B matrices, and a batch sharded over GPUs exchanges
 * ONE all-reduce of sum_out instead of an all-gather of U (the mean over noise / parameter instances is what
 * OptimalControlRobust.goal_run_with_grad forms, c3/optimizers/optimalcontrol_robust.py:49-70).
 *   infid_out f64 [B] or NULL; sum_out f64 [2] = {sum_b infid[b], B} or NULL (deterministic summation order)
 */
int c3p_gate_infid(const void* U, int B, int D, const int32_t* comp_rows, int L, const void* ideal, int kind,
                   int flags, double* infid_out, double* sum_out, void* stream);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif /* C3PROP_H */
