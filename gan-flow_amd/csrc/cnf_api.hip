// cnf_api.hip -- the extern "C" surface declared in include/cnf_hip.h: argument checking, workspace
// carving, launch.  Nothing here allocates or synchronises.
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <string>

#include "cnf_launch.h"
#include "cnf_split.h"

namespace cnf {
hipError_t launch_eval_aug(int D, bool hutch, const SolveArgs& a, float t, const float* z, const float* a_z,
                           const float* a_logp, float* f_out, float* nd_out, float* vjp_z_out, hipStream_t s);
hipError_t launch_solve_bwd(int D, bool hutch, const SolveArgs& a, hipStream_t s);
hipError_t launch_fixed_bwd(int D, bool hutch, const SolveArgs& a, hipStream_t s);
hipError_t launch_split16_fixed_bwd(const SolveArgs& a, hipStream_t s);
hipError_t launch_reg_mean(const float* reg_rows, long rows, int nreg, float* out, hipStream_t s);
hipError_t launch_wide_fixed_bwd(int D, bool hutch, const SolveArgs& a, float* ws_floats, hipStream_t s);
// feature-split (one tile per workgroup) variants, cnf_split.hip
hipError_t launch_split_fwd(int D, bool hutch, const SolveArgs& a, hipStream_t s);
hipError_t launch_split_bwd(int D, bool hutch, const SolveArgs& a, hipStream_t s);
hipError_t launch_split_fixed_fwd(int D, bool hutch, const SolveArgs& a, hipStream_t s);
hipError_t launch_split_fixed_bwd(int D, bool hutch, const SolveArgs& a, hipStream_t s);
hipError_t launch_split_aug(int D, bool hutch, const SolveArgs& a, float t, const float* z, const float* a_z,
                            const float* a_l, float* f_out, float* nd_out, float* vz_out, hipStream_t s);
// loss and optimizer step, cnf_train.hip
hipError_t launch_nll_loss(const float* z, const float* dlogp, long rows, int dim, float* loss_out, float* grad_z,
                           float* grad_dlogp, hipStream_t s);
hipError_t launch_adam(float* p, const float* g, float* m, float* v, int* state, long n, const CnfAdamOpts& o,
                       const CnfStats* guard, hipStream_t s);
hipError_t launch_fill_lds(unsigned pattern, unsigned* sink, hipStream_t s);
// training data on the device, cnf_data.hip
hipError_t launch_gu_sample(const CnfGuDesc& d, unsigned long long seed, int* state, long rows, float* out, hipStream_t s);
hipError_t launch_gu_sample_rows(const CnfGuDesc& d, unsigned long long seed, unsigned call, long row0, long rows, float* out,
                                 hipStream_t s);
// hidden width 256, forward path (cnf_wide.hip)
hipError_t launch_wide_field(int D, bool hutch, const float* params, float t, const float* z, const float* e, long rows,
                             float* f_out, float* nd_out, hipStream_t s, float res);
hipError_t launch_wide_solve(int D, bool hutch, const SolveArgs& a, int solver, bool global, hipStream_t s);
// hidden width 256, adjoint (cnf_wide_bwd.hip); ws_floats = workspace past the granules
size_t wide_bwd_workspace_floats(int D, long rows);
hipError_t launch_wide_bwd(int D, bool hutch, const SolveArgs& a, float* ws_floats, hipStream_t s);
hipError_t launch_wide_aug(int D, bool hutch, const SolveArgs& a, float* ws_floats, float t, const float* z, const float* a_z,
                           const float* a_l, float* f_out, float* nd_out, float* vz_out, hipStream_t s);
}  // namespace cnf

namespace {
// cnf_debug_kernel_timing: events bracketing the solve kernel of the most recent solve call (process-wide, diagnostics)
bool g_timing = false;
hipEvent_t g_ev0 = nullptr, g_ev1 = nullptr;
}  // namespace

namespace cnf {
void dbg_timing_begin(hipStream_t s) {
    if (g_timing) (void)hipEventRecord(g_ev0, s);
}
void dbg_timing_end(hipStream_t s) {
    if (g_timing) (void)hipEventRecord(g_ev1, s);
}
}  // namespace cnf

namespace {

thread_local std::string g_err = "";

int fail(const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return 1;
}

// Clearing device memory before a launch: a kernel of our own, not hipMemsetAsync.  Under stream capture a memset becomes
// a graph memset node, and a captured, unflagged solve replayed AFTER eager solves on the same device came back with a
// non-zero status word (the words its kernel assigns were right; tools/dbg_graph_memset.py): the
// node no longer cleared the block it was captured for.  A kernel node carries its arguments by value.  Every size
// cleared here is a multiple of 4 bytes and every pointer 4-byte aligned (workspaces: 16).
__global__ void zero_words_kernel(uint32_t* __restrict__ p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 0u;
}
hipError_t zero_async(void* p, int /*value: always 0*/, size_t bytes, hipStream_t s) {
    const size_t n = bytes / 4;
    if (n == 0) return hipSuccess;
    const size_t blocks = (n + 255) / 256;
    hipLaunchKernelGGL(zero_words_kernel, dim3((unsigned)(blocks < 2048 ? blocks : 2048)), dim3(256), 0, s, (uint32_t*)p, n);
    return hipGetLastError();
}
constexpr int64_t kFixedBwdMaxRows = 65536;   // fixed-grid adjoint: one gradient slab (37 KB) per 64 rows in the workspace
constexpr size_t kGranBytes = 2 * (size_t)cnf::kMaxWG * 4 * sizeof(unsigned long long) + 64;  // 16 KiB of granules + the ticket word

constexpr int kWideH = 256;   // cnf_wide.hip
constexpr size_t kWidePieceBytes = 786432;   // cnf_wide.h: bf16 pieces of the two hidden matrices (forward dopri5 solves)

bool is_wide(const CnfNetDesc* n) { return n->hidden == kWideH; }
bool is_residual(const CnfNetDesc* n) { return (n->divergence & CNF_DIV_RESIDUAL) != 0; }

bool supported(const CnfNetDesc* n) {
    if (!n || (n->hidden != cnf::kH && n->hidden != kWideH) || n->n_hidden != 3) return false;
    const int kind = n->divergence & ~CNF_DIV_RESIDUAL;
    const bool hutch = kind == CNF_DIV_HUTCHINSON;
    if (kind != CNF_DIV_EXACT && !hutch) return false;
    if (is_wide(n)) return (n->dim == 1 && !hutch) || n->dim == 2;
    return (n->dim == 1 && !hutch) || n->dim == 2 || (n->dim == 16 && hutch);
}

int check_net(const CnfNetDesc* n) {
    if (!n) return fail("net is NULL");
    if (!supported(n))
        return fail("unsupported net: dim=%d hidden=%d n_hidden=%d divergence=%d (built: hidden 64 x 3 with dim 1|2 exact, "
                    "dim 2|16 Hutchinson; hidden 256 x 3 with dim 1|2 exact, dim 2 Hutchinson)",
                    n->dim, n->hidden, n->n_hidden, n->divergence);
    return 0;
}

long n_wg(long rows) { return (rows + cnf::kRowsPerWG - 1) / cnf::kRowsPerWG; }

// Compute units of the current device (cached per device).  The grid-wide reductions of batch-global step control spin
// on every workgroup of the launch, so all of them must be resident at once: at most one workgroup per CU is assumed.
int device_cus() {
    static int cached[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return cnf::kMaxWG;
    if (cached[dev] == 0) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = cnf::kMaxWG;
        cached[dev] = n;
    }
    return cached[dev];
}

// batch-global solves of up to 16*256 rows run one tile per workgroup (cnf_split.hip): every workgroup resident
bool use_split(const CnfNetDesc* n, long rows) {
    return rows <= (long)cnf::kRows * cnf::kMaxWG;
}
// the one-tile-per-wave adjoint kernels (rows > 4096) exist for dim <= 2 only
int check_adjoint_rows(const char* who, const CnfNetDesc* n, long rows) {
    if (n->dim > 2 && !use_split(n, rows))
        return fail("%s: the adjoint for dim=%d supports at most %d rows per call (got %ld)", who, n->dim,
                    cnf::kRows * cnf::kMaxWG, rows);
    return 0;
}
// CNF_FLAG_SPREAD applies to the one-tile-per-workgroup adjoints (hidden 64 and 256) when twice the tiles still fit the device
bool can_spread(const CnfNetDesc* n, long rows, int cus) {
    const long tiles = (rows + cnf::kRows - 1) / cnf::kRows;
    return use_split(n, rows) && 2 * tiles <= cnf::kMaxWG && 2 * tiles <= cus;
}
long n_slabs(const CnfNetDesc* n, long rows) {
    if (!use_split(n, rows)) return n_wg(rows);
    const long tiles = (rows + cnf::kRows - 1) / cnf::kRows;
    return can_spread(n, rows, cnf::kMaxWG) ? 2 * tiles : tiles;   // room for a spread launch whether or not it is asked for
}

size_t align16(size_t x) { return (x + 15) & ~(size_t)15; }

}  // namespace

extern "C" {

const char* cnf_last_error(void) { return g_err.c_str(); }

int64_t cnf_param_count(const CnfNetDesc* net) {
    if (!supported(net)) return -1;
    return cnf::make_layout(net->dim, net->hidden).total;
}

size_t cnf_workspace_bytes(const CnfNetDesc* net, int64_t rows) {
    if (!supported(net) || rows <= 0) return 0;
    if (is_wide(net)) {
        // forward: granules + the bf16 pieces of the two hidden matrices (768 KB, written per dopri5 launch).  Adjoint
        // (rows <= 4096): transposed matrices + 8 gradient slots per 16-row tile in the same region (always larger).
        if (rows > (int64_t)cnf::kRows * cnf::kMaxWG) return kGranBytes + kWidePieceBytes;
        const size_t bwd = cnf::wide_bwd_workspace_floats(net->dim, rows) * sizeof(float);
        return kGranBytes + (bwd > kWidePieceBytes ? bwd : kWidePieceBytes);
    }
    const size_t P = (size_t)cnf::make_layout(net->dim, net->hidden).total;
    return kGranBytes + align16((size_t)n_slabs(net, rows) * P * sizeof(float));
}

int cnf_eval_field(const CnfNetDesc* net, const float* params, float t, const float* z, const float* e,
                   int64_t rows, float* f_out, float* neg_div_out, void* stream) {
    if (check_net(net)) return 1;
    if (!params || !z || !f_out || !neg_div_out) return fail("cnf_eval_field: NULL pointer");
    if (rows <= 0) return fail("cnf_eval_field: rows must be positive (got %ld)", (long)rows);
    const bool hutch = (net->divergence & ~CNF_DIV_RESIDUAL) == CNF_DIV_HUTCHINSON;
    if (hutch && !e) return fail("cnf_eval_field: Hutchinson divergence needs the probe e");
    if (is_wide(net)) {
        if (((uintptr_t)params & 15) != 0) return fail("cnf_eval_field: params must be 16-byte aligned for hidden width 256");
        hipError_t werr = cnf::launch_wide_field(net->dim, hutch, params, t, z, e, rows, f_out, neg_div_out, (hipStream_t)stream,
                                                 is_residual(net) ? 1.f : 0.f);
        if (werr != hipSuccess) return fail("cnf_eval_field: launch failed: %s", hipGetErrorString(werr));
        return 0;
    }
    hipError_t err = cnf::launch_field(net->dim, hutch, params, t, z, e, rows, f_out, neg_div_out, (hipStream_t)stream,
                                       is_residual(net) ? 1.f : 0.f);
    if (err != hipSuccess) return fail("cnf_eval_field: launch failed: %s", hipGetErrorString(err));
    return 0;
}

int cnf_debug_eval_field_b16(const CnfNetDesc* net, const float* params, float t, const float* z, const float* e,
                             int64_t rows, float* f_out, float* neg_div_out, void* stream) {
    if (check_net(net)) return 1;
    if (!params || !z || !f_out || !neg_div_out) return fail("cnf_debug_eval_field_b16: NULL pointer");
    if (rows <= 0) return fail("cnf_debug_eval_field_b16: rows must be positive (got %ld)", (long)rows);
    if (is_wide(net) || net->dim > 2) return fail("cnf_debug_eval_field_b16: hidden width 64 and dim <= 2 only");
    const bool hutch = (net->divergence & ~CNF_DIV_RESIDUAL) == CNF_DIV_HUTCHINSON;
    if (hutch && !e) return fail("cnf_debug_eval_field_b16: Hutchinson divergence needs the probe e");
    hipError_t err = cnf::launch_field(net->dim, hutch, params, t, z, e, rows, f_out, neg_div_out, (hipStream_t)stream,
                                       is_residual(net) ? 1.f : 0.f, true);
    if (err != hipSuccess) return fail("cnf_debug_eval_field_b16: launch failed: %s", hipGetErrorString(err));
    return 0;
}

// debug build: CNF_DEBUG_SHRINK_WS=<bytes> makes the checker believe the workspace ends that many bytes early (the memory is
// still there): the self-test of dbg_ws (tools/dbg_bounds_selftest.py) provokes a report without a real overrun
static size_t dbg_shrink() {
#ifdef CNF_DEBUG_BOUNDS
    static const size_t v = [] { const char* e = getenv("CNF_DEBUG_SHRINK_WS"); return e ? (size_t)strtoull(e, nullptr, 10) : (size_t)0; }();
    return v;
#else
    return 0;
#endif
}

// what a FORWARD solve touches of its workspace: the granules, and at hidden width 256 the bf16 weight pieces behind them
static size_t fwd_workspace_need(const CnfNetDesc* net) { return kGranBytes + (is_wide(net) ? kWidePieceBytes : 0); }

// need_override > 0: the workspace is checked against that many bytes instead of cnf_workspace_bytes (cnf_solve_fwd_static)
static int fill_common(const char* who, const CnfNetDesc* net, const float* params, int64_t rows, float t0, float t1,
                       const CnfSolverOpts* opts, void* workspace, size_t workspace_bytes, CnfStats* stats,
                       cnf::SolveArgs& a, bool& global, size_t need_override = 0) {
    if (check_net(net)) return 1;
    if (!params || !opts || !stats || !workspace) return fail("%s: NULL pointer", who);
    if (rows <= 0) return fail("%s: rows must be positive (got %ld)", who, (long)rows);
    if (!(t0 != t1) || !std::isfinite(t0) || !std::isfinite(t1)) return fail("%s: t0 and t1 must be finite and distinct", who);
    if (((uintptr_t)workspace & 15) != 0) return fail("%s: workspace must be 16-byte aligned", who);
    const size_t need = need_override ? need_override : cnf_workspace_bytes(net, rows);
    if (workspace_bytes < need) return fail("%s: workspace too small (%zu < %zu bytes)", who, workspace_bytes, need);
    a = cnf::SolveArgs{};
    a.params = params;
    a.rows = rows;
    a.t0 = t0;
    a.t1 = t1;
    a.rtol = opts->rtol;
    a.atol = opts->atol;
    a.max_steps = opts->max_steps > 0 ? opts->max_steps : (1 << 20);
    a.gran = (unsigned long long*)workspace;
    a.slab = (float*)((char*)workspace + kGranBytes);
    a.ws_lo = (const char*)workspace;
    a.ws_hi = (const char*)workspace + need - dbg_shrink();
    a.stats = stats;
    a.n_grid = 1;
    a.residual = is_residual(net) ? 1 : 0;
    a.step_size = 0.f;
    a.after = opts->after;
    a.fwd4 = (opts->flags & CNF_FLAG_FWD_FOUR_WAVES) ? 1 : 0;
    if (opts->after == stats) return fail("%s: opts->after must be another block than this solve's stats", who);
    if (opts->flags & ~(CNF_FLAG_CLEAN | CNF_FLAG_SPREAD | CNF_FLAG_ADJ_CONTROL_ALL | CNF_FLAG_FWD_FOUR_WAVES)) return fail("%s: unknown bits in opts->flags (0x%x)", who, (unsigned)opts->flags);
    global = true;
    if (opts->solver == CNF_SOLVER_DOPRI5) {
        if (!(opts->rtol > 0.f) || !(opts->atol > 0.f)) return fail("%s: dopri5 needs rtol > 0 and atol > 0", who);
        if (opts->tile_rows != 0 && opts->tile_rows != cnf::kRows)
            return fail("%s: tile_rows must be 0 (batch-global step control) or %d", who, cnf::kRows);
        global = opts->tile_rows == 0;
        if (global && is_wide(net) && rows > (int64_t)cnf::kRows * cnf::kMaxWG)
            return fail("%s: batch-global step control at hidden width 256 supports at most %d rows per solve (got %ld); "
                        "split the batch or use tile_rows=%d", who, cnf::kRows * cnf::kMaxWG, (long)rows, cnf::kRows);
        if (global && n_wg(rows) > cnf::kMaxWG)
            return fail("%s: batch-global step control supports at most %d rows per solve (got %ld); split the batch into "
                        "chunks as the reference's evaluation does, or use tile_rows=%d", who,
                        cnf::kMaxWG * cnf::kRowsPerWG, (long)rows, cnf::kRows);
        if (global) {
            const long wgs = (is_wide(net) || use_split(net, rows)) ? (rows + cnf::kRows - 1) / cnf::kRows : n_wg(rows);
            if (wgs > 1 && wgs > device_cus())
                return fail("%s: batch-global step control needs %ld co-resident workgroups but the device has %d compute units; "
                            "split the batch or use tile_rows=%d", who, wgs, device_cus(), cnf::kRows);
        }
    } else if (opts->solver == CNF_SOLVER_RK4 || opts->solver == CNF_SOLVER_EULER || opts->solver == CNF_SOLVER_MIDPOINT) {
        a.fixed_kind = opts->solver;
        if (opts->step_size > 0.f) {
            // the fixed grid of torchdiffeq: ceil((t1-t0)/h + 1) points, the last clipped to t1 (fp32 arithmetic)
            const float sgn = t1 >= t0 ? 1.f : -1.f;
            const float tau0 = sgn * t0, tau1 = sgn * t1;
            const float span = tau1 - tau0;
            const float q = span / opts->step_size;
            const float npts = std::ceil(q + 1.0f);
            if (!(npts >= 2.f) || npts > 1e7f) return fail("%s: step_size gives an unusable grid (%g points)", who, npts);
            const float last = (npts - 1.0f) * opts->step_size + tau0;
            if (last < tau1) return fail("%s: fixed grid does not reach t1 (torchdiffeq asserts time_grid[-1] == t[-1])", who);
            a.n_grid = (int)npts - 1;
            a.step_size = opts->step_size;
        }
    } else {
        return fail("%s: unknown solver %d", who, opts->solver);
    }
    return 0;
}

}  // extern "C"

namespace {
// cnf_solve_fwd and cnf_solve_fwd_static.  library_scratch: `workspace` is the library's own static array, which
// other calls have used: its granules are always cleared (CNF_FLAG_CLEAN then only spares the stats block's memset)
int solve_fwd_impl(const char* who, bool library_scratch, const CnfNetDesc* net, const float* params, const float* x,
                   const float* logp0, const float* e, int64_t rows, float t0, float t1, const CnfSolverOpts* opts,
                   float* z_out, float* logp_out, void* workspace, size_t workspace_bytes, CnfStats* stats, void* stream) {
    cnf::SolveArgs a;
    bool global;
    const size_t need = library_scratch && net && supported(net) ? fwd_workspace_need(net) : 0;
    if (fill_common(who, net, params, rows, t0, t1, opts, workspace, workspace_bytes, stats, a, global, need)) return 1;
    if (!x || !z_out || !logp_out) return fail("%s: NULL pointer", who);
    const bool hutch = (net->divergence & ~CNF_DIV_RESIDUAL) == CNF_DIV_HUTCHINSON;
    if (hutch && !e) return fail("%s: Hutchinson divergence needs the probe e", who);
    a.x = x;
    a.logp0 = logp0;
    a.e = e;
    a.z_out = z_out;
    a.logp_out = logp_out;
    hipStream_t s = (hipStream_t)stream;
    // CNF_FLAG_CLEAN: batch-global dopri5 kernels that clean up after themselves (cnf_split.hip, cnf_wide.hip)
    const bool clean = (opts->flags & CNF_FLAG_CLEAN) && opts->solver == CNF_SOLVER_DOPRI5 && global &&
                       (is_wide(net) || use_split(net, rows));
    if (is_wide(net) && ((uintptr_t)params & 15) != 0) return fail("%s: params must be 16-byte aligned for hidden width 256", who);
    hipError_t err = hipSuccess;
    if (!clean || library_scratch) err = zero_async(workspace, 0, kGranBytes, s);
    if (!clean && err == hipSuccess) err = zero_async(stats, 0, sizeof(CnfStats), s);
    if (err == hipSuccess) {
        if (is_wide(net)) err = cnf::launch_wide_solve(net->dim, hutch, a, opts->solver, global, s);
        else if (opts->solver == CNF_SOLVER_DOPRI5 && global && use_split(net, rows)) err = cnf::launch_split_fwd(net->dim, hutch, a, s);
        // fixed-grid solves of up to 4096 rows: one tile per workgroup (cnf_split_fwd_tw.h, kFixedSplitNote)
        else if (opts->solver != CNF_SOLVER_DOPRI5 && net->dim <= 2 && use_split(net, rows) && !a.fwd4) err = cnf::launch_split_fixed_fwd(net->dim, hutch, a, s);
        else err = cnf::launch_solve_fwd(net->dim, hutch, a, opts->solver, global, s);
    }
    if (err != hipSuccess) return fail("%s: launch failed: %s", who, hipGetErrorString(err));
    return 0;
}
}  // namespace

extern "C" {

int cnf_solve_fwd(const CnfNetDesc* net, const float* params, const float* x, const float* logp0, const float* e,
                  int64_t rows, float t0, float t1, const CnfSolverOpts* opts, float* z_out, float* logp_out,
                  void* workspace, size_t workspace_bytes, CnfStats* stats, void* stream) {
    return solve_fwd_impl("cnf_solve_fwd", false, net, params, x, logp0, e, rows, t0, t1, opts, z_out, logp_out, workspace,
                          workspace_bytes, stats, stream);
}

int cnf_solve_fwd_times(const CnfNetDesc* net, const float* params, const float* x, const float* logp0, const float* e,
                        int64_t rows, const float* times, int n_times, const CnfSolverOpts* opts, float* z_out, float* logp_out,
                        void* workspace, size_t workspace_bytes, CnfStats* stats, void* stream) {
    const char* who = "cnf_solve_fwd_times";
    if (!times) return fail("%s: NULL pointer", who);
    if (n_times < 2 || n_times > CNF_MAX_TIMES) return fail("%s: n_times must be in [2, %d] (got %d)", who, CNF_MAX_TIMES, n_times);
    for (int i = 0; i < n_times; ++i)
        if (!std::isfinite(times[i])) return fail("%s: times[%d] is not finite", who, i);
    const bool up = times[1] > times[0];
    for (int i = 1; i < n_times; ++i)
        if (!(up ? times[i] > times[i - 1] : times[i] < times[i - 1]))
            return fail("%s: times must be strictly increasing or strictly decreasing (times[%d] = %g after %g)", who, i,
                        (double)times[i], (double)times[i - 1]);
    if (n_times == 2)
        return solve_fwd_impl(who, false, net, params, x, logp0, e, rows, times[0], times[1], opts, z_out, logp_out, workspace,
                              workspace_bytes, stats, stream);
    cnf::SolveArgs a;
    bool global;
    if (fill_common(who, net, params, rows, times[0], times[n_times - 1], opts, workspace, workspace_bytes, stats, a, global)) return 1;
    if (!x || !z_out || !logp_out) return fail("%s: NULL pointer", who);
    if (opts->solver != CNF_SOLVER_DOPRI5) return fail("%s: several output times are built for dopri5 only", who);
    if (is_wide(net)) return fail("%s: several output times are built for hidden width 64 only", who);
    const bool hutch = (net->divergence & ~CNF_DIV_RESIDUAL) == CNF_DIV_HUTCHINSON;
    if (hutch && !e) return fail("%s: Hutchinson divergence needs the probe e", who);
    if (global && n_wg(rows) > device_cus())
        return fail("%s: batch-global step control needs %ld co-resident workgroups but the device has %d compute units; "
                    "split the batch or use tile_rows=%d", who, (long)n_wg(rows), device_cus(), cnf::kRows);
    a.x = x;
    a.logp0 = logp0;
    a.e = e;
    a.z_out = z_out;
    a.logp_out = logp_out;
    a.n_times = n_times;
    for (int i = 0; i < n_times; ++i) a.times[i] = times[i];
    hipStream_t s = (hipStream_t)stream;
    hipError_t err = zero_async(workspace, 0, kGranBytes, s);
    if (err == hipSuccess) err = zero_async(stats, 0, sizeof(CnfStats), s);
    if (err == hipSuccess) err = cnf::launch_solve_fwd(net->dim, hutch, a, opts->solver, global, s);
    if (err != hipSuccess) return fail("%s: launch failed: %s", who, hipGetErrorString(err));
    return 0;
}

// ---- cnf_solve_fwd_static: the forward solve for callers that cannot allocate ------------------------------------------
// The workspace of the forward solve lives in the library: one file-scope device array per GPU (each device that loads the
// code object holds its own copy, at its own address, until the process ends).  `scratch` is the HOST-side stand-in of that
// array -- never a device address: the per-device address comes from hipGetSymbolAddress, once per device, and only that
// pointer is ever handed to a memset or a kernel.
constexpr size_t kStaticScratchBytes = kGranBytes + kWidePieceBytes;   // 802 880 bytes: granules + the H = 256 weight pieces
static __device__ __align__(256) unsigned char scratch[kStaticScratchBytes];

namespace {
std::mutex g_scratch_mu;
void* g_scratch_ptr[64] = {nullptr};   // per device, filled on the first call with that device current

int static_scratch(const char* who, void** out) {
    int dev = -1;
    hipError_t err = hipGetDevice(&dev);
    if (err != hipSuccess) return fail("%s: hipGetDevice failed: %s", who, hipGetErrorString(err));
    if (dev < 0 || dev >= 64) return fail("%s: device index %d is outside the 64 devices the library keeps scratch pointers for", who, dev);
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    if (!g_scratch_ptr[dev]) {
        void* ptr = nullptr;
        err = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));
        if (err != hipSuccess || !ptr)
            return fail("%s: hipGetSymbolAddress(scratch) failed on device %d: %s (the first call on a device must not be "
                        "under stream capture)", who, dev, hipGetErrorString(err));
        g_scratch_ptr[dev] = ptr;
    }
    *out = g_scratch_ptr[dev];
    return 0;
}
}  // namespace

size_t cnf_static_scratch_bytes(void) { return kStaticScratchBytes; }

int cnf_solve_fwd_static(const CnfNetDesc* net, const float* params, const float* x, const float* logp0, const float* e,
                         int64_t rows, float t0, float t1, const CnfSolverOpts* opts, float* z_out, float* logp_out,
                         CnfStats* stats, void* stream) {
    const char* who = "cnf_solve_fwd_static";
    if (check_net(net)) return 1;
    const size_t need = fwd_workspace_need(net);
    if (need > kStaticScratchBytes)
        return fail("%s: this solve needs %zu bytes of scratch, the library's static array holds %zu; use cnf_solve_fwd with a "
                    "workspace of cnf_workspace_bytes()", who, need, kStaticScratchBytes);
    void* ptr = nullptr;
    if (static_scratch(who, &ptr)) return 1;
    return solve_fwd_impl(who, true, net, params, x, logp0, e, rows, t0, t1, opts, z_out, logp_out, ptr, kStaticScratchBytes,
                          stats, stream);
}

int cnf_eval_aug(const CnfNetDesc* net, const float* params, float t, const float* z, const float* e, const float* a_z,
                 const float* a_logp, int64_t rows, float* f_out, float* neg_div_out, float* vjp_z_out,
                 float* vjp_params_out, void* workspace, size_t workspace_bytes, void* stream) {
    if (check_net(net)) return 1;
    if (!params || !z || !a_z || !a_logp || !f_out || !neg_div_out || !vjp_z_out || !vjp_params_out || !workspace)
        return fail("cnf_eval_aug: NULL pointer");
    if (rows <= 0) return fail("cnf_eval_aug: rows must be positive (got %ld)", (long)rows);
    if (n_wg(rows) > cnf::kMaxWG) return fail("cnf_eval_aug: at most %d rows", cnf::kMaxWG * cnf::kRowsPerWG);
    const size_t need = cnf_workspace_bytes(net, rows);
    if (workspace_bytes < need) return fail("cnf_eval_aug: workspace too small (%zu < %zu bytes)", workspace_bytes, need);
    const bool hutch = (net->divergence & ~CNF_DIV_RESIDUAL) == CNF_DIV_HUTCHINSON;
    if (hutch && !e) return fail("cnf_eval_aug: Hutchinson divergence needs the probe e");
    if (check_adjoint_rows("cnf_eval_aug", net, rows)) return 1;
    cnf::SolveArgs a{};
    a.params = params;
    a.rows = rows;
    a.residual = is_residual(net) ? 1 : 0;
    a.e = e;
    a.gran = (unsigned long long*)workspace;
    a.slab = (float*)((char*)workspace + kGranBytes);
    a.ws_lo = (const char*)workspace;
    a.ws_hi = (const char*)workspace + need - dbg_shrink();
    a.grad_params = vjp_params_out;
    hipStream_t s = (hipStream_t)stream;
    if (is_wide(net)) {
        if (rows > (int64_t)cnf::kRows * cnf::kMaxWG) return fail("cnf_eval_aug: at most %d rows at hidden width 256", cnf::kRows * cnf::kMaxWG);
        if (((uintptr_t)params & 15) != 0 || ((uintptr_t)workspace & 15) != 0)
            return fail("cnf_eval_aug: params and workspace must be 16-byte aligned for hidden width 256");
        hipError_t werr = cnf::launch_wide_aug(net->dim, hutch, a, (float*)((char*)workspace + kGranBytes), t, z, a_z, a_logp,
                                               f_out, neg_div_out, vjp_z_out, s);
        if (werr != hipSuccess) return fail("cnf_eval_aug: launch failed: %s", hipGetErrorString(werr));
        return 0;
    }
    hipError_t err = zero_async(workspace, 0, need, s);   // granules + per-workgroup gradient slabs
    if (err == hipSuccess) {
        if (use_split(net, rows)) err = cnf::launch_split_aug(net->dim, hutch, a, t, z, a_z, a_logp, f_out, neg_div_out, vjp_z_out, s);
        else err = cnf::launch_eval_aug(net->dim, hutch, a, t, z, a_z, a_logp, f_out, neg_div_out, vjp_z_out, s);
    }
    if (err != hipSuccess) return fail("cnf_eval_aug: launch failed: %s", hipGetErrorString(err));
    return 0;
}

}  // extern "C"

namespace {
// cnf_solve_bwd and the solve half of cnf_solve_bwd_train.  train: the cotangents may be left out (loss_out set:
// the kernel forms those of the NLL itself) and the gradient slabs stay in the workspace for launch_adam_slabs.
int solve_bwd_impl(const char* who, bool train, const CnfNetDesc* net, const float* params, const float* z1, const float* logp1,
                   const float* e, const float* grad_z1, const float* grad_logp1, int64_t rows, float t0, float t1,
                   const CnfSolverOpts* opts, float* grad_x, float* grad_logp0, float* grad_params, float* loss_out,
                   void* workspace, size_t workspace_bytes, CnfStats* stats, void* stream, cnf::SolveArgs& a) {
    bool global;
    if (net && supported(net) && is_wide(net) && rows > (int64_t)cnf::kRows * cnf::kMaxWG)
        return fail("%s: the adjoint at hidden width 256 supports at most %d rows per call (got %ld)", who,
                    cnf::kRows * cnf::kMaxWG, (long)rows);
    if (fill_common(who, net, params, rows, t0, t1, opts, workspace, workspace_bytes, stats, a, global)) return 1;
    const bool nll = train && !grad_z1;
    if (!z1 || !logp1 || !grad_x || !grad_logp0 || (!train && !grad_params)) return fail("%s: NULL pointer", who);
    if (!nll && (!grad_z1 || !grad_logp1)) return fail("%s: NULL pointer (grad_z1 / grad_logp1)", who);
    if (nll && (grad_logp1 || !loss_out))
        return fail("%s: grad_z1 == NULL asks for the NLL cotangents: grad_logp1 must be NULL too and loss_out set", who);
    if (train && (is_wide(net) || !use_split(net, rows)))
        return fail("%s: built for hidden width 64 and at most %d rows (got hidden %d, %ld rows); use cnf_nll_loss, "
                    "cnf_solve_bwd and cnf_adam_step", who, cnf::kRows * cnf::kMaxWG, net->hidden, (long)rows);
    const bool fixed = opts->solver != CNF_SOLVER_DOPRI5;   // euler / midpoint / rk4: the adjoint runs on the same fixed grid, backwards
    if (fixed) {
        // cnf_solve_bwd_train on a fixed grid: the one-tile-per-workgroup kernels (cnf_split.hip) with the caller's cotangents
        if (train && (nll || net->dim > 2 || (opts->flags & CNF_FLAG_FWD_FOUR_WAVES)))
            return fail("%s: with a fixed-grid solver it takes explicit cotangents (cnf_nll_loss), dim <= 2 and no "
                        "CNF_FLAG_FWD_FOUR_WAVES; otherwise cnf_solve_bwd and cnf_adam_step", who);
        if (!is_wide(net) && net->dim <= 2 && rows > kFixedBwdMaxRows)
            return fail("%s: the fixed-grid adjoint supports at most %ld rows per call (got %ld)", who, (long)kFixedBwdMaxRows, (long)rows);
        if (opts->step_size > 0.f) {
            // the grid of the backward solve starts at t1: torchdiffeq's grid constructor on the negated time axis
            const float sgn = t0 >= t1 ? 1.f : -1.f;
            const float tau0 = sgn * t1, tau1 = sgn * t0;
            const float npts = std::ceil((tau1 - tau0) / opts->step_size + 1.0f);
            if (!(npts >= 2.f) || npts > 1e7f) return fail("%s: step_size gives an unusable grid (%g points)", who, npts);
            if ((npts - 1.0f) * opts->step_size + tau0 < tau1)
                return fail("%s: fixed grid does not reach t0 (torchdiffeq asserts time_grid[-1] == t[-1])", who);
            a.n_grid = (int)npts - 1;
        }
    }
    if (!global) return fail("%s: the adjoint pass needs batch-global step control (tile_rows = 0)", who);
    const bool hutch = (net->divergence & ~CNF_DIV_RESIDUAL) == CNF_DIV_HUTCHINSON;
    if (hutch && !e) return fail("%s: Hutchinson divergence needs the probe e", who);
    if (check_adjoint_rows(who, net, rows)) return 1;
    a.loss_out = nll ? loss_out : nullptr;
    a.keep_slabs = train ? 1 : 0;
    a.x = z1;
    a.logp0 = logp1;
    a.e = e;
    a.grad_z1 = grad_z1;
    a.grad_logp1 = grad_logp1;
    a.grad_x = grad_x;
    a.grad_logp0 = grad_logp0;
    a.grad_params = grad_params;
    a.spread = (opts->flags & CNF_FLAG_SPREAD) && can_spread(net, rows, device_cus()) ? 1 : 0;
    a.adj_all = (opts->flags & CNF_FLAG_ADJ_CONTROL_ALL) ? 1 : 0;
    hipStream_t s = (hipStream_t)stream;
    if (is_wide(net)) {
        if (((uintptr_t)params & 15) != 0) return fail("%s: params must be 16-byte aligned for hidden width 256", who);
        hipError_t werr = hipSuccess;
        if (!(opts->flags & CNF_FLAG_CLEAN)) {
            werr = zero_async(workspace, 0, kGranBytes, s);   // the gradient slots are written before they are read
            if (werr == hipSuccess) werr = zero_async(stats, 0, sizeof(CnfStats), s);
        }
        if (werr == hipSuccess && fixed && (opts->flags & CNF_FLAG_CLEAN)) werr = zero_async(stats, 0, sizeof(CnfStats), s);
        if (werr == hipSuccess) werr = fixed ? cnf::launch_wide_fixed_bwd(net->dim, hutch, a, (float*)((char*)workspace + kGranBytes), s)
                                             : cnf::launch_wide_bwd(net->dim, hutch, a, (float*)((char*)workspace + kGranBytes), s);
        if (werr != hipSuccess) return fail("%s: launch failed: %s", who, hipGetErrorString(werr));
        return 0;
    }
    hipError_t err = hipSuccess;
    if (fixed && net->dim > 2) {   // d = 16: one workgroup per tile (cnf_split16.hip), the slab is stored, not added to
        const uintptr_t al = (uintptr_t)z1 | (uintptr_t)e | (uintptr_t)grad_z1 | (uintptr_t)grad_x | (uintptr_t)params;
        if ((al & 15) != 0) return fail("%s: the fixed-grid adjoint at dim 16 needs 16-byte aligned z1, e, grad_z1, grad_x and params", who);
        err = zero_async(stats, 0, sizeof(CnfStats), s);
        if (err == hipSuccess) err = cnf::launch_split16_fixed_bwd(a, s);
        if (err != hipSuccess) return fail("%s: launch failed: %s", who, hipGetErrorString(err));
        return 0;
    }
    if (fixed && use_split(net, rows) && !a.fwd4) {   // one tile per workgroup (cnf_split.hip, fixed_bwd_split_kernel): the slabs are stored
        err = zero_async(stats, 0, sizeof(CnfStats), s);
        if (err == hipSuccess) err = cnf::launch_split_fixed_bwd(net->dim, hutch, a, s);
        if (err != hipSuccess) return fail("%s: launch failed: %s", who, hipGetErrorString(err));
        return 0;
    }
    if (fixed) {
        err = zero_async(workspace, 0, cnf_workspace_bytes(net, rows), s);   // the gradient slabs are accumulated into
        if (err == hipSuccess) err = zero_async(stats, 0, sizeof(CnfStats), s);
        if (err == hipSuccess) err = cnf::launch_fixed_bwd(net->dim, hutch, a, s);
        if (err != hipSuccess) return fail("%s: launch failed: %s", who, hipGetErrorString(err));
        return 0;
    }
    if ((opts->flags & CNF_FLAG_CLEAN) && use_split(net, rows)) {
        // nothing to clear: granules and status are clean by contract, the slabs are stored (not accumulated) by the
        // first accepted step
    } else if (use_split(net, rows)) {
        err = zero_async(workspace, 0, kGranBytes, s);
        if (err == hipSuccess) err = zero_async(stats, 0, sizeof(CnfStats), s);
    } else {
        err = zero_async(workspace, 0, cnf_workspace_bytes(net, rows), s);   // granules + gradient slabs
        if (err == hipSuccess) err = zero_async(stats, 0, sizeof(CnfStats), s);
    }
    if (err == hipSuccess) err = use_split(net, rows) ? cnf::launch_split_bwd(net->dim, hutch, a, s) : cnf::launch_solve_bwd(net->dim, hutch, a, s);
    if (err != hipSuccess) return fail("%s: launch failed: %s", who, hipGetErrorString(err));
    return 0;
}
}  // namespace

extern "C" {

int cnf_solve_bwd(const CnfNetDesc* net, const float* params, const float* z1, const float* logp1, const float* e,
                  const float* grad_z1, const float* grad_logp1, int64_t rows, float t0, float t1,
                  const CnfSolverOpts* opts, float* grad_x, float* grad_logp0, float* grad_params, void* workspace,
                  size_t workspace_bytes, CnfStats* stats, void* stream) {
    cnf::SolveArgs a;
    return solve_bwd_impl("cnf_solve_bwd", false, net, params, z1, logp1, e, grad_z1, grad_logp1, rows, t0, t1, opts, grad_x,
                          grad_logp0, grad_params, nullptr, workspace, workspace_bytes, stats, stream, a);
}

// ---- regularised dynamics (CnfRegDesc, include/cnf_hip.h) ---------------------------------------------------------------
static int check_reg(const char* who, const CnfNetDesc* net, const CnfRegDesc* reg, int64_t rows, const CnfSolverOpts* opts) {
    if (check_net(net)) return 1;
    if (!reg || !opts) return fail("%s: NULL pointer", who);
    if (reg->n < 1 || reg->n > CNF_MAX_REG) return fail("%s: reg->n must be 1..%d (got %d)", who, CNF_MAX_REG, reg->n);
    for (int k = 0; k < reg->n; ++k)
        if (reg->kind[k] != CNF_REG_L1 && reg->kind[k] != CNF_REG_L2)
            return fail("%s: reg->kind[%d] = %d: built are CNF_REG_L1 and CNF_REG_L2", who, k, reg->kind[k]);
    if (is_wide(net) || net->dim > 2) return fail("%s: regularised dynamics are built for hidden width 64 and dim 1 | 2", who);
    if (opts->solver != CNF_SOLVER_DOPRI5 || opts->tile_rows != 0)
        return fail("%s: regularised dynamics are built for dopri5 with batch-global step control", who);
    if (opts->flags & CNF_FLAG_ADJ_CONTROL_ALL) return fail("%s: CNF_FLAG_ADJ_CONTROL_ALL is not built for regularised dynamics", who);
    if (rows > 0 && (n_wg(rows) > cnf::kMaxWG || n_wg(rows) > device_cus()))
        return fail("%s: at most %ld rows per solve on this device (got %ld)", who,
                    (long)cnf::kRowsPerWG * (cnf::kMaxWG < device_cus() ? cnf::kMaxWG : device_cus()), (long)rows);
    return 0;
}

int cnf_solve_fwd_reg(const CnfNetDesc* net, const CnfRegDesc* reg, const float* params, const float* x, const float* logp0,
                      const float* e, int64_t rows, float t0, float t1, const CnfSolverOpts* opts, float* z_out, float* logp_out,
                      float* reg_out, void* workspace, size_t workspace_bytes, CnfStats* stats, void* stream) {
    const char* who = "cnf_solve_fwd_reg";
    if (check_reg(who, net, reg, rows, opts)) return 1;
    cnf::SolveArgs a;
    bool global;
    if (fill_common(who, net, params, rows, t0, t1, opts, workspace, workspace_bytes, stats, a, global)) return 1;
    if (!x || !z_out || !logp_out || !reg_out) return fail("%s: NULL pointer", who);
    const bool hutch = (net->divergence & ~CNF_DIV_RESIDUAL) == CNF_DIV_HUTCHINSON;
    if (hutch && !e) return fail("%s: Hutchinson divergence needs the probe e", who);
    a.x = x;
    a.logp0 = logp0;
    a.e = e;
    a.z_out = z_out;
    a.logp_out = logp_out;
    a.nreg = reg->n;
    for (int k = 0; k < reg->n; ++k) a.reg_kind[k] = reg->kind[k];
    a.reg_rows = a.slab;   // reg->n x rows floats: the head of the gradient region, which a forward solve does not use
    hipStream_t s = (hipStream_t)stream;
    hipError_t err = zero_async(workspace, 0, kGranBytes, s);
    if (err == hipSuccess) err = zero_async(stats, 0, sizeof(CnfStats), s);
    if (err == hipSuccess) err = cnf::launch_solve_fwd(net->dim, hutch, a, CNF_SOLVER_DOPRI5, true, s);
    if (err == hipSuccess) err = cnf::launch_reg_mean(a.reg_rows, rows, reg->n, reg_out, s);
    if (err != hipSuccess) return fail("%s: launch failed: %s", who, hipGetErrorString(err));
    return 0;
}

int cnf_solve_bwd_reg(const CnfNetDesc* net, const CnfRegDesc* reg, const float* params, const float* z1, const float* logp1,
                      const float* e, const float* reg_t1, const float* grad_z1, const float* grad_logp1, const float* grad_reg,
                      int64_t rows, float t0, float t1, const CnfSolverOpts* opts, float* grad_x, float* grad_logp0,
                      float* grad_params, void* workspace, size_t workspace_bytes, CnfStats* stats, void* stream) {
    const char* who = "cnf_solve_bwd_reg";
    if (check_reg(who, net, reg, rows, opts)) return 1;
    cnf::SolveArgs a;
    bool global;
    if (fill_common(who, net, params, rows, t0, t1, opts, workspace, workspace_bytes, stats, a, global)) return 1;
    if (!z1 || !logp1 || !reg_t1 || !grad_z1 || !grad_logp1 || !grad_reg || !grad_x || !grad_logp0 || !grad_params)
        return fail("%s: NULL pointer", who);
    const bool hutch = (net->divergence & ~CNF_DIV_RESIDUAL) == CNF_DIV_HUTCHINSON;
    if (hutch && !e) return fail("%s: Hutchinson divergence needs the probe e", who);
    a.x = z1;
    a.logp0 = logp1;
    a.e = e;
    a.grad_z1 = grad_z1;
    a.grad_logp1 = grad_logp1;
    a.grad_x = grad_x;
    a.grad_logp0 = grad_logp0;
    a.grad_params = grad_params;
    a.nreg = reg->n;
    for (int k = 0; k < reg->n; ++k) a.reg_kind[k] = reg->kind[k];
    a.reg_val = reg_t1;
    a.reg_cot = grad_reg;
    hipStream_t s = (hipStream_t)stream;
    // the one-tile-per-wave adjoint: n_wg gradient slabs, accumulated into (cnf_workspace_bytes covers them for any row count)
    const size_t P = (size_t)cnf::make_layout(net->dim, net->hidden).total;
    hipError_t err = zero_async(workspace, 0, kGranBytes + align16((size_t)n_wg(rows) * P * sizeof(float)), s);
    if (err == hipSuccess) err = zero_async(stats, 0, sizeof(CnfStats), s);
    if (err == hipSuccess) err = cnf::launch_solve_bwd(net->dim, hutch, a, s);
    if (err != hipSuccess) return fail("%s: launch failed: %s", who, hipGetErrorString(err));
    return 0;
}

int cnf_solve_bwd_train(const CnfNetDesc* net, float* params, const float* z1, const float* logp1, const float* e,
                        const float* grad_z1, const float* grad_logp1, int64_t rows, float t0, float t1,
                        const CnfSolverOpts* opts, float* grad_x, float* grad_logp0, float* grad_params, float* loss_out,
                        float* exp_avg, float* exp_avg_sq, int32_t* adam_state, const CnfAdamOpts* adam, void* workspace,
                        size_t workspace_bytes, CnfStats* stats, void* stream) {
    const char* who = "cnf_solve_bwd_train";
    if (!exp_avg || !exp_avg_sq || !adam_state || !adam) return fail("%s: NULL pointer (optimizer state)", who);
    if (!(adam->lr >= 0.) || !(adam->beta1 >= 0. && adam->beta1 < 1.) || !(adam->beta2 >= 0. && adam->beta2 < 1.) ||
        !(adam->eps >= 0.) || !(adam->weight_decay >= 0.))
        return fail("%s: invalid hyper-parameters (lr %g, betas %g %g, eps %g, weight_decay %g)", who, adam->lr, adam->beta1,
                    adam->beta2, adam->eps, adam->weight_decay);
    cnf::SolveArgs a;
    if (solve_bwd_impl(who, true, net, params, z1, logp1, e, grad_z1, grad_logp1, rows, t0, t1, opts, grad_x, grad_logp0,
                       grad_params, loss_out, workspace, workspace_bytes, stats, stream, a))
        return 1;
    const cnf::ParamLayout L = cnf::make_layout(net->dim, net->hidden);
    const int tiles = (int)((rows + cnf::kRows - 1) / cnf::kRows);
    hipError_t err = cnf::launch_adam_slabs(a.slab, tiles, L.total, a.spread ? L.w(2) : 0, grad_params, params, exp_avg,
                                            exp_avg_sq, adam_state, *adam, stats, (hipStream_t)stream);
    if (err != hipSuccess) return fail("%s: launch failed: %s", who, hipGetErrorString(err));
    return 0;
}

int cnf_nll_loss(const float* z, const float* delta_logp, int64_t rows, int32_t dim, float* loss_out, float* grad_z,
                 float* grad_delta_logp, void* stream) {
    if (!z || !delta_logp || !loss_out) return fail("cnf_nll_loss: NULL pointer");
    if (rows <= 0 || dim <= 0) return fail("cnf_nll_loss: rows and dim must be positive (got %ld, %d)", (long)rows, dim);
    hipError_t err = cnf::launch_nll_loss(z, delta_logp, rows, dim, loss_out, grad_z, grad_delta_logp, (hipStream_t)stream);
    if (err != hipSuccess) return fail("cnf_nll_loss: launch failed: %s", hipGetErrorString(err));
    return 0;
}

int cnf_adam_step(float* params, const float* grad, float* exp_avg, float* exp_avg_sq, int32_t* state, int64_t n,
                  const CnfAdamOpts* opts, const CnfStats* guard, void* stream) {
    if (!params || !grad || !exp_avg || !exp_avg_sq || !state || !opts) return fail("cnf_adam_step: NULL pointer");
    if (n <= 0) return fail("cnf_adam_step: n must be positive (got %ld)", (long)n);
    if (!(opts->lr >= 0.) || !(opts->beta1 >= 0. && opts->beta1 < 1.) || !(opts->beta2 >= 0. && opts->beta2 < 1.) ||
        !(opts->eps >= 0.) || !(opts->weight_decay >= 0.))
        return fail("cnf_adam_step: invalid hyper-parameters (lr %g, betas %g %g, eps %g, weight_decay %g)", opts->lr,
                    opts->beta1, opts->beta2, opts->eps, opts->weight_decay);
    hipError_t err = cnf::launch_adam(params, grad, exp_avg, exp_avg_sq, state, n, *opts, guard, (hipStream_t)stream);
    if (err != hipSuccess) return fail("cnf_adam_step: launch failed: %s", hipGetErrorString(err));
    return 0;
}

int cnf_gu_sample(const CnfGuDesc* gu, uint64_t seed, int32_t* state, int64_t rows, float* out, void* stream) {
    if (!gu || !state || !out) return fail("cnf_gu_sample: NULL pointer");
    if (rows <= 0) return fail("cnf_gu_sample: rows must be positive (got %ld)", (long)rows);
    if (gu->n_mixture < 1 || gu->n_mixture > CNF_GU_MAX_MIXTURE)
        return fail("cnf_gu_sample: n_mixture must be 1..%d (got %d)", CNF_GU_MAX_MIXTURE, gu->n_mixture);
    for (int c = 0; c < gu->n_mixture; ++c)
        if (!(gu->sigma[c] > 0.f) || !(gu->unif_intsect[c] >= 0.f) || !(gu->unif_p[c] >= 0.f && gu->unif_p[c] <= 1.f) ||
            !std::isfinite(gu->mean[c]))
            return fail("cnf_gu_sample: component %d: need sigma > 0, unif_intsect >= 0, 0 <= unif_p <= 1, finite mean", c);
    hipError_t err = cnf::launch_gu_sample(*gu, seed, state, rows, out, (hipStream_t)stream);
    if (err != hipSuccess) return fail("cnf_gu_sample: launch failed: %s", hipGetErrorString(err));
    return 0;
}

static int check_gu(const char* who, const CnfGuDesc* gu) {
    if (gu->n_mixture < 1 || gu->n_mixture > CNF_GU_MAX_MIXTURE)
        return fail("%s: n_mixture must be 1..%d (got %d)", who, CNF_GU_MAX_MIXTURE, gu->n_mixture);
    for (int c = 0; c < gu->n_mixture; ++c)
        if (!(gu->sigma[c] > 0.f) || !(gu->unif_intsect[c] >= 0.f) || !(gu->unif_p[c] >= 0.f && gu->unif_p[c] <= 1.f) ||
            !std::isfinite(gu->mean[c]))
            return fail("%s: component %d: need sigma > 0, unif_intsect >= 0, 0 <= unif_p <= 1, finite mean", who, c);
    return 0;
}

int cnf_gu_sample_rows(const CnfGuDesc* gu, uint64_t seed, uint32_t call, int64_t row0, int64_t rows, float* out, void* stream) {
    if (!gu || !out) return fail("cnf_gu_sample_rows: NULL pointer");
    if (rows <= 0 || row0 < 0) return fail("cnf_gu_sample_rows: need rows > 0 and row0 >= 0 (got %ld, %ld)", (long)rows, (long)row0);
    if (check_gu("cnf_gu_sample_rows", gu)) return 1;
    hipError_t err = cnf::launch_gu_sample_rows(*gu, seed, call, row0, rows, out, (hipStream_t)stream);
    if (err != hipSuccess) return fail("cnf_gu_sample_rows: launch failed: %s", hipGetErrorString(err));
    return 0;
}

int cnf_debug_fill_lds(uint32_t pattern, uint32_t* scratch_word, void* stream) {
    if (!scratch_word) return fail("cnf_debug_fill_lds: NULL pointer");
    hipError_t err = cnf::launch_fill_lds(pattern, scratch_word, (hipStream_t)stream);
    if (err != hipSuccess) return fail("cnf_debug_fill_lds: launch failed: %s", hipGetErrorString(err));
    return 0;
}

#ifndef CNF_ACT
#define CNF_ACT 0
#endif
int cnf_activation(void) { return CNF_ACT; }

int cnf_debug_kernel_timing(int enable) {
    if (enable && !g_ev0) {
        hipError_t err = hipEventCreate(&g_ev0);
        if (err == hipSuccess) err = hipEventCreate(&g_ev1);
        if (err != hipSuccess) return fail("cnf_debug_kernel_timing: hipEventCreate failed: %s", hipGetErrorString(err));
    }
    g_timing = enable != 0;
    return 0;
}

int cnf_debug_last_kernel_ms(float* ms_out) {
    if (!ms_out) return fail("cnf_debug_last_kernel_ms: NULL pointer");
    if (!g_ev0) return fail("cnf_debug_last_kernel_ms: timing was never enabled");
    hipError_t err = hipEventSynchronize(g_ev1);
    if (err == hipSuccess) err = hipEventElapsedTime(ms_out, g_ev0, g_ev1);
    if (err != hipSuccess) return fail("cnf_debug_last_kernel_ms: %s", hipGetErrorString(err));
    return 0;
}

}  // extern "C"
