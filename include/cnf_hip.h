/*
 * cnf_hip.h -- C ABI of libcnf_hip.so (and of its siblings built for other nonlinearities, see cnf_activation):
 * the MI355X (gfx950) FFJORD CNF hot path.
 *
 * The reference has no native code and no FFI: its hot path is Python calling torchdiffeq
 *     odeint_adjoint(odefunc, (z, logpz), integration_times, atol=[a,a], rtol=[r,r], method=solver)
 * from CNF.forward (ffjord/lib/layers/cnf.py, un-vendored submodule: /root/reference/.gitmodules:4-6;
 * dependency pinned at /root/reference/requirements.txt:17; the only in-tree caller is
 * /root/reference/model_plot.py:93-101).  The entry points below are what a binding for THAT call would
 * bind: one typed call per solve instead of a Python callback per Runge-Kutta stage.
 *
 * Conventions
 *   - every pointer is a BORROWED DEVICE pointer (fp32, contiguous row-major) unless marked "host";
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it, nothing synchronises;
 *   - the library never allocates device memory: the caller passes a workspace of
 *     cnf_workspace_bytes() bytes (16-byte aligned) that the call zeroes (with a kernel, never a memset node) and uses
 *     (see CNF_FLAG_CLEAN);
 *   - return value 0 = enqueued; non-zero = rejected before launch, message in cnf_last_error();
 *   - run-time failures of the integrator (step underflow, non-finite state, step limit) are
 *     reported in CnfStats.status, which lives in device memory and is read by the caller after
 *     it has synchronised the stream.  They mirror torchdiffeq's assertions
 *     ("underflow in dt", "non-finite values in state `y`", "max_num_steps exceeded").
 *
 * Parameter vector `params` (cnf_param_count() floats): for each ConcatSquashLinear layer l in order,
 *     _layer.weight (out x in, row-major) | _layer.bias (out) | _hyper_bias.weight (out)
 *     | _hyper_gate.weight (out) | _hyper_gate.bias (out)
 * i.e. torch.cat([p.reshape(-1) for p in odenet.parameters()]) of the reference's ODEnet
 * (state_dict order recorded in SURVEY.md Appendix A).  Gradients come back in the same layout.
 */
#ifndef CNF_HIP_H
#define CNF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ODEnet description: dim -> hidden x n_hidden -> dim, ConcatSquashLinear layers, the library's nonlinearity (tanh for
 * libcnf_hip.so) between layers.
 * Replaces the Python object graph ODEfunc(ODEnet(...)).            */
typedef struct CnfNetDesc {
    int32_t dim;        /* data dimension d: 1, 2 or 16                                          */
    int32_t hidden;     /* hidden width: 64, or 256 (dim 1 | 2; narrower nets: zero-padded by the caller) */
    int32_t n_hidden;   /* number of hidden layers: 3                                            */
    int32_t divergence; /* 0 = brute force (exact trace, `divergence_bf`), 1 = Hutchinson
                           (`divergence_approx`, probe `e` fixed for the whole solve); | CNF_DIV_RESIDUAL */
} CnfNetDesc;

enum { CNF_SOLVER_DOPRI5 = 0, CNF_SOLVER_RK4 = 1, CNF_SOLVER_EULER = 2, CNF_SOLVER_MIDPOINT = 3 };   /* 1..3: fixed grid */
enum { CNF_DIV_EXACT = 0, CNF_DIV_HUTCHINSON = 1 };
/* OR-ed into CnfNetDesc.divergence: ODEfunc(residual=True) -- the dynamics are dy = f(t, y) - y and the divergence is
 * reduced by dim accordingly (ffjord/lib/layers/odefunc.py, UPSTREAM-RECALL; the reference's runs use False). */
enum { CNF_DIV_RESIDUAL = 256 };

struct CnfStats;

/* Solver options: the keyword arguments of torchdiffeq.odeint (rtol, atol, method, options). */
typedef struct CnfSolverOpts {
    int32_t solver;      /* CNF_SOLVER_*                                                          */
    float rtol, atol;    /* dopri5 tolerances (the same pair for z and logp, as CNF passes them)   */
    float step_size;     /* fixed-grid solvers: options['step_size']; <= 0 means one step over [t0, t1] */
    int32_t tile_rows;   /* dopri5 step control: 0 = one step size for the whole batch (reference
                            semantics); 16 = every 16-row tile controls its own step (throughput
                            evaluation mode)                              */
    int32_t max_steps;   /* dopri5 attempt limit per solve (0 = 1<<20)                            */
    int32_t flags;       /* CNF_FLAG_* (0 = none); unknown bits are rejected                      */
    const struct CnfStats *after; /* NULL, or the (device) CnfStats block of the solve whose OUTPUT this solve
                            consumes -- the forward solve of the previous block, the forward solve an adjoint
                            solve reverses, the adjoint solve of the next block.  If that block's status is
                            non-zero when this kernel starts (stream order), it exits at once with the same
                            status and touches no output: a failure travels down a chain of solves without
                            a host read-back, and the Adam updates at its end skip themselves.  Must not be
                            this call's own `stats`.  (No counterpart in the reference, which raises from
                            Python at the failing step.)                                              */
} CnfSolverOpts;

/* CNF_FLAG_CLEAN: the caller guarantees that the first 16 KiB + 64 bytes of `workspace` are zero and that
 * stats->status is zero -- true for a zero-initialised workspace / stats block that only flagged calls have used
 * since (every batch-global dopri5 solve leaves those bytes zero again when it ends, a failed one included unless
 * it timed out).  The call then enqueues no memset at all: tight loops and captured graphs save four fill nodes
 * per training iteration.  stats->status becomes sticky (it is only ever raised).  Ignored (memsets kept) where it
 * does not apply: rk4, tile-local control, more than 4096 rows.                                            */
enum { CNF_FLAG_CLEAN = 1, CNF_FLAG_SPREAD = 2, CNF_FLAG_ADJ_CONTROL_ALL = 4, CNF_FLAG_FWD_FOUR_WAVES = 8 };

/* CNF_FLAG_FWD_FOUR_WAVES (cnf_solve_fwd, cnf_solve_fwd_static; diagnostics): run a batch-global dopri5 forward solve of
 * hidden width 64, dim <= 2, at most 4096 rows on the four-wave kernel (every wave carries the value and the tangent
 * streams of its feature tile) instead of the eight-wave one (value stream on four waves, tangent streams and log p on the
 * second wave of each SIMD, one barrier interval behind; twelve waves -- a helper group per tangent stream -- at dim 2 with
 * the exact divergence).  Same arithmetic, identical bits; for A/B timing and the test
 * that holds the two kernels to each other.  Ignored where the eight-wave kernel does not apply.
 * Fixed-grid solvers (CNF_SOLVER_RK4 | EULER | MIDPOINT; cnf_solve_fwd, cnf_solve_fwd_static, cnf_solve_bwd), hidden width 64,
 * dim <= 2, at most 4096 rows: such solves run one 16-row tile per WORKGROUP (the feature-split evaluations of the dopri5
 * kernels on the fixed grid: 4-6x shorter than one wave per tile at the training batch size; the adjoint with
 * CNF_FLAG_SPREAD on two eight-wave workgroups per tile); the flag keeps the one-wave-per-tile kernels that larger batches
 * use (same grid and step counts; results equal up to the hidden layers' arithmetic, ~1e-7 relative per evaluation).     */

/* CNF_FLAG_SPREAD (cnf_solve_bwd, at most half as many 16-row tiles as the device has
 * compute units -- 2048 rows on an MI355X; ignored otherwise): the adjoint runs TWO workgroups per tile.  Both
 * integrate the tile identically; each keeps half of the parameter-gradient sums, which shortens every augmented
 * evaluation.  Results are those of an unflagged call (same sums, added in the same order).  The launch then occupies
 * every compute unit: do not flag solves that are meant to overlap with other batch-global solves on the same device
 * (two trainings on two streams), whose workgroups must all be resident at once.                              */

/* CNF_FLAG_ADJ_CONTROL_ALL (cnf_solve_bwd, cnf_solve_bwd_train; ignored by the forward solves): which tensors of the
 * adjoint state drive the backward solve's step control.  Unflagged = `state_only`: (z, logp) alone, which is what the
 * reference's length-2 tolerance lists do in torchdiffeq 0.0.1 as recalled (the solver zips tolerances with the state
 * tuple; SURVEY.md Appendix D.3 -- unverifiable offline, hence this switch).  Flagged: the per-row adjoint tensors a_z and
 * a_logp are controlled like z and logp -- each adds its own mean-square error ratio to the accept test and its own
 * terms to the initial-step heuristic, in the order of the state tuple.  Because a_logp is constant its d1 is 0 and the
 * heuristic's d0 / d1 is infinite, as it is in the original's arithmetic: the first step is then (0.01 / max d1)^(1/5)
 * (cnf_device.h spells the NaN / inf bookkeeping out).  The two batch-level quadratures of the adjoint state, a_t and
 * a_theta, stay outside the controller in both modes.  Neither mode is pinned by a reference
 * artefact; backward NFE is the only thing that differs.                                                       */

enum {
    CNF_OK = 0,
    CNF_ERR_DT_UNDERFLOW = 1, /* torchdiffeq: "underflow in dt"                                   */
    CNF_ERR_NONFINITE = 2,    /* torchdiffeq: "non-finite values in state `y`"                    */
    CNF_ERR_MAX_STEPS = 3,    /* torchdiffeq: max_num_steps exceeded                              */
    CNF_ERR_SYNC_TIMEOUT = 4, /* a grid-wide reduction did not complete (should never happen)     */
    CNF_ERR_DEBUG_BOUNDS = 5  /* debug build only (`make -C gan-flow_amd/csrc debug`, -DCNF_DEBUG_BOUNDS): a kernel derived a
                                 sub-range of the workspace that lies outside the bytes cnf_workspace_bytes() names for the
                                 call; the access was redirected to the workspace's first bytes and the details printed */
};

/* Written by the kernels (device memory, 32 bytes).  For tile-local step control the counters are the
 * maxima over tiles and `row_evals` is the sum over tiles of rows*evaluations (FLOP accounting). */
typedef struct CnfStats {
    int32_t status;      /* CNF_OK or CNF_ERR_*                                                   */
    int32_t nfe;         /* function evaluations as the reference's ODEfunc._num_evals counts them */
    int32_t n_accepted;
    int32_t n_rejected;
    float last_dt;
    int32_t evals_run;   /* evaluations actually executed (the adjoint re-evaluates FSAL stages)   */
    int64_t row_evals;   /* sum over rows of evaluations executed                                 */
} CnfStats;

/* The ODEnet nonlinearity this library was compiled for (ODEnet(nonlinearity=...), the reference's --nonlinearity):
 * 0 = tanh (libcnf_hip.so; the reference's runs and checkpoints), 1 = softplus (libcnf_hip_softplus.so), 2 = elu
 * (libcnf_hip_elu.so), 3 = relu (libcnf_hip_relu.so).  Every library exports this same interface; the caller binds the one that matches its net.    */
int cnf_activation(void);

/* Number of floats in `params` for this net; < 0 if the net is not supported. */
int64_t cnf_param_count(const CnfNetDesc *net);

/* Bytes of workspace the solve entry points need for a batch of `rows` (both directions). */
size_t cnf_workspace_bytes(const CnfNetDesc *net, int64_t rows);

/* Last error message of the calling thread (host string, never NULL). */
const char *cnf_last_error(void);

/* One evaluation of the dynamics, ODEfunc.forward(t, (y, logp)) -> (f, -div)   (SURVEY.md 8a a5).
 *   z (rows x dim), e (rows x dim, Hutchinson probe or NULL), f_out (rows x dim), neg_div_out (rows). */
int cnf_eval_field(const CnfNetDesc *net, const float *params, float t, const float *z, const float *e,
                   int64_t rows, float *f_out, float *neg_div_out, void *stream);

/* Forward solve: odeint(odefunc, (x, logp0), [t0, t1]) -> state at t1  (a7, a9, a11).
 *   logp0 may be NULL (zeros).  t0 > t1 integrates backwards in time (reverse=True / sample_fn).  */
int cnf_solve_fwd(const CnfNetDesc *net, const float *params, const float *x, const float *logp0,
                  const float *e, int64_t rows, float t0, float t1, const CnfSolverOpts *opts,
                  float *z_out, float *logp_out, void *workspace, size_t workspace_bytes,
                  CnfStats *stats, void *stream);

/* cnf_solve_fwd for callers that cannot allocate device memory or hand over a workspace: code under HIP graph capture,
 * tight serving loops (sample_fn / density_fn per request), harnesses that forbid hipMalloc.  NO COUNTERPART IN THE
 * REFERENCE (whose solver allocates tensors at every Runge-Kutta stage).  Same arguments, results and error behaviour as
 * cnf_solve_fwd -- bit-identical outputs and stats -- minus `workspace`: the granules of the grid-wide reductions (and, at
 * hidden width 256, the bf16 pieces of the two hidden matrices) live in a file-scope device array inside the library,
 * cnf_static_scratch_bytes() = 802 880 bytes per GPU the library has run on, held until the process ends.
 * The call allocates nothing (no hipMalloc / hipFree, no allocating library routine); it enqueues one clear of the first
 * 16 448 bytes of that array (a kernel of the library's own, so that it is a kernel node under capture) and the solve kernel(s).  The array's per-device address is looked up on the first call on
 * each device (hipGetSymbolAddress): make that first call outside stream capture.
 * Limits: those of cnf_solve_fwd (batch-global control: <= 4096 rows at hidden width 256, <= 16 384 at 64; tile-local and
 * fixed-grid: any number of rows); a solve whose scratch need exceeds the array is rejected with a message.
 * Every call on a GPU shares that GPU's array: calls that OVERLAP in time on one device (two streams, two threads) race on
 * the granules -- serialise them, or give the overlapping solves their own workspaces through cnf_solve_fwd.
 * CNF_FLAG_CLEAN only spares the memset of `stats` here (the library clears its granules itself).                    */
size_t cnf_static_scratch_bytes(void);
int cnf_solve_fwd_static(const CnfNetDesc *net, const float *params, const float *x, const float *logp0,
                         const float *e, int64_t rows, float t0, float t1, const CnfSolverOpts *opts,
                         float *z_out, float *logp_out, CnfStats *stats, void *stream);

/* Forward solve with SEVERAL output times: odeint(odefunc, (x, logp0), times) with len(times) > 2 -- what CNF.forward
 * passes on when the caller gives it `integration_times` (ffjord/lib/layers/cnf.py: `state_t = odeint(...)`, returned
 * stacked over t; the call /root/reference/requirements.txt:17 provides).  ONE adaptive integration from times[0] to
 * times[n_times - 1] whose steps ignore the output times; the state at every times[i] is read off the quartic interpolant
 * of the accepted step that first reaches it (torchdiffeq 0.0.1: `while next_t > rk_state.t1: step`, then
 * `_interp_evaluate`).  times: HOST array of n_times floats, strictly increasing or strictly decreasing,
 * 2 <= n_times <= CNF_MAX_TIMES (copied into the kernel's arguments: no device copy, capturable).
 * z_out: n_times x rows x dim, logp_out: n_times x rows; slab 0 is the input state.
 * dopri5 only; hidden width 64 (the one-tile-per-wave kernels of csrc/cnf_fwd.hip: batch-global control up to 16 384 rows,
 * tile-local any number).  The adjoint of such a solve is a chain of cnf_solve_bwd calls over the intervals
 * [times[i-1], times[i]], i = n_times - 1 .. 1, each started from slab i with the output cotangent of slab i added to the
 * carried one -- exactly OdeintAdjointMethod.backward's loop (gan_flow_amd/lib/layers/cnf.py::_SolveTimes).            */
#define CNF_MAX_TIMES 32
int cnf_solve_fwd_times(const CnfNetDesc *net, const float *params, const float *x, const float *logp0,
                        const float *e, int64_t rows, const float *times, int n_times, const CnfSolverOpts *opts,
                        float *z_out, float *logp_out, void *workspace, size_t workspace_bytes,
                        CnfStats *stats, void *stream);

/* Regularised dynamics: RegularizedODEfunc with non-empty regularization_fns (ffjord/lib/layers/wrappers/
 * cnf_regularization.py, un-vendored submodule /root/reference/.gitmodules:4-6; the wrapper class every reference
 * checkpoint carries).  Each regulariser is one more state of the ODE, d r / dt = reg_fn(dz/dt) averaged over the batch,
 * started at 0; CNF.forward integrates it with tolerances 1e20 (it never fails the accept test) -- but the solver's
 * first-step heuristic takes rtol[0] / atol[0] for EVERY state tensor, so the regularisers do steer the first step of
 * both solves, and the kernels reproduce that (parity unpinned: no reference artefact records such a run).
 *   CNF_REG_L1: l1_regularzation_fn, mean over the batch of mean_i |dz_i/dt|
 *   CNF_REG_L2: l2_regularzation_fn, mean over the batch of ||dz/dt||_2 / sqrt(dim)
 * cnf_solve_fwd_reg = cnf_solve_fwd + reg_out[n] (device): the regularisation states at t1.
 * cnf_solve_bwd_reg = cnf_solve_bwd + reg_t1[n] (device: those states, they scale the first backward step) and
 *   grad_reg[n] (device: dL/d(state)); the adjoint of a regulariser state is constant, and enters the sweep as
 *   grad_reg / rows * d reg_fn / d(dz/dt) added to the cotangent of dz/dt.
 * dopri5 with batch-global step control, hidden width 64, dim 1 | 2, up to 16 384 rows (the one-tile-per-wave kernels of
 * csrc/cnf_fwd.hip / cnf_bwd.hip); kinds: up to CNF_MAX_REG entries, in the order of the reference's regularization_fns. */
enum { CNF_REG_L1 = 1, CNF_REG_L2 = 2 };
#define CNF_MAX_REG 2
typedef struct CnfRegDesc {
    int32_t n;                     /* 1 .. CNF_MAX_REG */
    int32_t kind[CNF_MAX_REG];
} CnfRegDesc;
int cnf_solve_fwd_reg(const CnfNetDesc *net, const CnfRegDesc *reg, const float *params, const float *x, const float *logp0,
                      const float *e, int64_t rows, float t0, float t1, const CnfSolverOpts *opts,
                      float *z_out, float *logp_out, float *reg_out, void *workspace, size_t workspace_bytes,
                      CnfStats *stats, void *stream);
int cnf_solve_bwd_reg(const CnfNetDesc *net, const CnfRegDesc *reg, const float *params, const float *z1, const float *logp1,
                      const float *e, const float *reg_t1, const float *grad_z1, const float *grad_logp1,
                      const float *grad_reg, int64_t rows, float t0, float t1, const CnfSolverOpts *opts,
                      float *grad_x, float *grad_logp0, float *grad_params, void *workspace, size_t workspace_bytes,
                      CnfStats *stats, void *stream);

/* One evaluation of the adjoint's augmented dynamics (the closure `augmented_dynamics` inside
 * OdeintAdjointMethod.backward, a10): given z, a_z, a_logp returns
 *   f, -div, vjp_z = -(a_z . df/dz) + a_logp * d(div)/dz   (rows x dim)
 *   vjp_params (cnf_param_count floats) = -(a . d(f,-div)/dtheta) summed over rows.               */
int cnf_eval_aug(const CnfNetDesc *net, const float *params, float t, const float *z, const float *e,
                 const float *a_z, const float *a_logp, int64_t rows, float *f_out, float *neg_div_out,
                 float *vjp_z_out, float *vjp_params_out, void *workspace, size_t workspace_bytes,
                 void *stream);

/* Backward solve: OdeintAdjointMethod.backward for the single interval [t0, t1] (a10).
 *   inputs : z1, logp1 = state at t1 saved by the forward; grad_z1, grad_logp1 = dL/d(state at t1)
 *   outputs: grad_x, grad_logp0 = dL/d(state at t0); grad_params (overwritten).
 *   dopri5: error control covers (z, logp) only, as the reference's length-2 tolerance lists imply;
 *   CNF_FLAG_ADJ_CONTROL_ALL adds (a_z, a_logp).
 *   CNF_SOLVER_RK4 | EULER | MIDPOINT (hidden width 64: dim 1 | 2 up to 65 536 rows, dim 16 up to 4096 rows with 16-byte
 *   aligned z1, e, grad_z1, grad_x and params; hidden width 256: up to 4096 rows): the augmented state is integrated
 *   backwards with the same fixed-grid method, the grid starting at t1 with spacing opts->step_size (<= 0: one step), as
 *   odeint_adjoint(..., method=solver, options={'step_size': h}) does; stats: nfe = stages x steps, n_accepted = steps.  */
int cnf_solve_bwd(const CnfNetDesc *net, const float *params, const float *z1, const float *logp1,
                  const float *e, const float *grad_z1, const float *grad_logp1, int64_t rows,
                  float t0, float t1, const CnfSolverOpts *opts, float *grad_x, float *grad_logp0,
                  float *grad_params, void *workspace, size_t workspace_bytes, CnfStats *stats,
                  void *stream);

/* Negative mean log-likelihood of a batch under the flow and its gradient w.r.t. the solver outputs: the
 * reference's training loss (standard_normal_logprob(z).sum(1) - delta_logp, loss = -mean;
 * run as `python3 -m ffjord.gu_maf`, /root/reference/README.md:41,57).
 *   z (rows x dim), delta_logp (rows): outputs of cnf_solve_fwd;  loss_out: 1 float (device);
 *   grad_z (rows x dim) = z / rows and grad_delta_logp (rows) = 1 / rows, either may be NULL.
 * Deterministic (fixed summation order).                                                                 */
int cnf_nll_loss(const float *z, const float *delta_logp, int64_t rows, int32_t dim, float *loss_out,
                 float *grad_z, float *grad_delta_logp, void *stream);

/* Adam on a flat parameter vector, the update the reference's driver applies with torch.optim.Adam
 * (lr 1e-3, weight_decay 1e-5 added to the gradient).                 */
typedef struct CnfAdamOpts {
    double lr, beta1, beta2, eps, weight_decay; /* doubles, as torch holds them: 1 - beta is formed in double */
} CnfAdamOpts;

/*   params, exp_avg, exp_avg_sq (n floats each) are updated in place from grad (n floats);
 *   state: 2 int32 in device memory, zeroed by the caller before the first step: {step count, internal};
 *   the step count is read and incremented on the device, so the call can be captured in a hipGraph.    */
/*   guard (nullable): the CnfStats block of the solve that produced `grad`; if its status is non-zero the kernel
 *   changes nothing -- parameters, moments and step count keep their bits (a failed solve leaves garbage gradients). */
int cnf_adam_step(float *params, const float *grad, float *exp_avg, float *exp_avg_sq, int32_t *state,
                  int64_t n, const CnfAdamOpts *opts, const struct CnfStats *guard, void *stream);

/* The tail of a training iteration for one CNF block in two kernels: the adjoint solve of cnf_solve_bwd, then the sum
 * of its per-workgroup gradient partials with the Adam update of cnf_adam_step as that kernel's epilogue (what the
 * reference's driver does with `loss.backward(); optimizer.step()`).  Same arguments as those
 * two calls, and the same bits in every output.  Two things differ:
 *   - grad_z1 == NULL (then grad_logp1 must be NULL as well) selects the cotangents of the flow's NLL,
 *     grad_z1 = z1 / rows and grad_logp1 = 1 / rows, formed inside the adjoint kernel, and stores that loss
 *     (cnf_nll_loss's value up to summation order) to loss_out: for the LAST block of a flow this replaces cnf_nll_loss;
 *   - grad_params may be NULL (the summed gradient is then not stored anywhere).
 * `params` is updated in place AFTER the solve has read it -- unless the adjoint solve ended with a non-zero status
 * (its own failure, or one inherited through opts->after): then the update kernel changes nothing.
 * Hidden width 64, at most 4096 rows; otherwise the call fails and the three separate entry points are to be used.
 * Fixed-grid solvers: dim <= 2 and explicit cotangents only (grad_z1 / grad_logp1 from cnf_nll_loss; the in-kernel loss
 * rides on a grid-wide reduction of the dopri5 controller, which a fixed grid does not have).                          */
int cnf_solve_bwd_train(const CnfNetDesc *net, float *params, const float *z1, const float *logp1,
                        const float *e, const float *grad_z1, const float *grad_logp1, int64_t rows,
                        float t0, float t1, const CnfSolverOpts *opts, float *grad_x, float *grad_logp0,
                        float *grad_params, float *loss_out, float *exp_avg, float *exp_avg_sq,
                        int32_t *adam_state, const CnfAdamOpts *adam, void *workspace, size_t workspace_bytes,
                        CnfStats *stats, void *stream);

/* The study's training data drawn on the device: GausUniffMixture.get_sample (/root/reference/gu.py:35-68), which the
 * reference draws with numpy on the host and copies to the GPU every iteration.  Same distribution --
 * rows i.i.d.: component uniform over n_mixture, then with probability unif_p[c] uniform on mean[c] +- unif_intsect[c] *
 * sigma[c], else N(mean[c], sigma[c]^2); with extend_dim a second column U(0, 1) (gu.py:63-66) -- from another random
 * stream (Philox4x32-10 keyed by `seed`, counter = (row, call number)): numpy's global-state stream cannot be matched on a
 * device, the host sampler stays the bit-exact restatement.
 *   out: rows x (extend_dim ? 2 : 1) floats;  state: 2 int32 in device memory, zeroed by the caller before the first
 *   call: {call number, internal}; the kernel advances the call number itself, so a captured graph draws a fresh batch
 *   at every replay.  mean[] holds the component means themselves (the reference's cumulative sums of mean_dist).   */
enum { CNF_GU_MAX_MIXTURE = 16 };
typedef struct CnfGuDesc {
    int32_t n_mixture;   /* 1 .. CNF_GU_MAX_MIXTURE */
    int32_t extend_dim;  /* 0 | 1 */
    float mean[CNF_GU_MAX_MIXTURE], sigma[CNF_GU_MAX_MIXTURE], unif_intsect[CNF_GU_MAX_MIXTURE], unif_p[CNF_GU_MAX_MIXTURE];
} CnfGuDesc;
int cnf_gu_sample(const CnfGuDesc *gu, uint64_t seed, int32_t *state, int64_t rows, float *out, void *stream);

/* The same generator addressed by GLOBAL row index, stateless: rows row0 .. row0 + rows - 1 of draw number `call` (the
 * Philox counter is (row, call), exactly cnf_gu_sample's; cnf_gu_sample_rows(gu, seed, c, 0, n, ..) writes what the c-th
 * cnf_gu_sample call of n rows writes).  For sharded evaluation (BASELINE config 3): a rank draws ITS
 * rows of a data set of any size, and the set is the same for every number of ranks.  The reference has no counterpart:
 * its evaluation data are one host array (/root/reference/model_plot.py:93-101).  With n_mixture = 1, unif_p = 0 the rows
 * are N(mean, sigma^2): the prior draws of the sampling protocol and the Gaussian "tail" half of config 3's points.   */
int cnf_gu_sample_rows(const CnfGuDesc *gu, uint64_t seed, uint32_t call, int64_t row0, int64_t rows, float *out, void *stream);

/* Diagnostics (no counterpart in the reference): fill the LDS of every compute unit with `pattern`.  LDS keeps the
 * previous kernel's content, so the tests poison it (NaN bit patterns) before a solve to prove that no result
 * depends on it.  scratch_word: one uint32 of device memory (never written in practice).                    */
int cnf_debug_fill_lds(uint32_t pattern, uint32_t *scratch_word, void *stream);

/* Diagnostics (no counterpart in the reference): cnf_eval_field computed the way the forward dopri5 solves at hidden
 * width 64 and dim <= 2 compute it -- hidden layers on the bf16 matrix pipe, every fp32 operand split into three bf16
 * pieces, fp32 sums -- so that tests can hold that arithmetic against the oracle and against
 * float64 one evaluation at a time.  Same arguments and conventions as cnf_eval_field.                       */
int cnf_debug_eval_field_b16(const CnfNetDesc *net, const float *params, float t, const float *z, const float *e,
                             int64_t rows, float *f_out, float *neg_div_out, void *stream);

/* Diagnostics (no counterpart in the reference): while enabled, cnf_solve_fwd / cnf_solve_bwd record a HIP event on
 * the launch stream immediately before and after their solve kernel (not the memsets, not the slab reduction);
 * cnf_debug_last_kernel_ms waits for the second event and returns the elapsed time of the most recent solve kernel.
 * Process-wide, one device at a time, not for use under graph capture: this is how bench.py measures the
 * `roofline` kernel duration on the stream the kernel runs on.                                              */
int cnf_debug_kernel_timing(int enable);
int cnf_debug_last_kernel_ms(float *ms_out);

#ifdef __cplusplus
}
#endif
#endif /* CNF_HIP_H */
