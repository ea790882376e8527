// cnf_launch.h -- kernel argument block and launcher prototypes shared by the .hip translation units.
#pragma once
#include <atomic>
#include <mutex>

#include "cnf_device.h"

namespace cnf {

constexpr int kMaxTimes = CNF_MAX_TIMES;   // include/cnf_hip.h

struct SolveArgs {
    const float* params;
    const float* x;       // state at t0 (rows x D)
    const float* logp0;   // nullable
    const float* e;       // Hutchinson probe (rows x D) or null
    float* z_out;
    float* logp_out;
    long rows;
    float t0, t1;
    float rtol, atol;
    float step_size;      // rk4 grid spacing (only read when n_grid > 1)
    int n_grid;           // fixed-grid solvers: number of steps
    int fixed_kind;       // CNF_SOLVER_RK4 (3/8 rule) | CNF_SOLVER_EULER | CNF_SOLVER_MIDPOINT
    int max_steps;
    unsigned long long* gran;  // grid-reduction granules, zeroed before the launch
    CnfStats* stats;           // zeroed before the launch
    // adjoint only
    const float* grad_z1;
    const float* grad_logp1;
    float* grad_x;
    float* grad_logp0;
    float* grad_params;
    float* slab;          // per-workgroup parameter-gradient partials [nwg][P]
    int spread;           // adjoint, cnf_split.hip: two workgroups per tile, each keeping half of the d/dtheta sums
    float* loss_out;      // adjoint, cnf_solve_bwd_train: non-null = the cotangents are those of the flow's NLL
                          // (grad_z1 = z1 / rows, grad_logp1 = 1 / rows; grad_z1 / grad_logp1 are not read) and the
                          // loss goes here
    int residual;         // CNF_DIV_RESIDUAL: the dynamics are f(z) - z (ODEfunc(residual=True))
    int keep_slabs;       // adjoint: leave the per-workgroup gradient slabs for launch_adam_slabs (no reduction kernel)
    int adj_all;          // adjoint, CNF_FLAG_ADJ_CONTROL_ALL: a_z and a_logp under step control as well (cnf_device.h)
    int fwd4;             // CNF_FLAG_FWD_FOUR_WAVES: the four-wave forward kernel where the eight-wave one would run
    const CnfStats* after;  // CnfSolverOpts.after (nullable): exit at once with that block's status when it is non-zero
    // cnf_solve_fwd_times (dopri5_kernel, cnf_fwd.hip): n_times > 2 = the state at every times[i] goes to slab i of z_out /
    // logp_out (slab 0 = the input; t0 = times[0], t1 = times[n_times - 1]); 0 = the ordinary solve
    int n_times;
    float times[kMaxTimes];
    // regularised dynamics (cnf_solve_fwd_reg / cnf_solve_bwd_reg; dopri5_kernel<.., GLOBAL> and dopri5_bwd_kernel): nreg = 0 off
    int nreg;
    int reg_kind[CNF_MAX_REG];   // CNF_REG_L1 | CNF_REG_L2
    float* reg_rows;             // forward: every row's integral of its regularisers' integrands, [nreg][rows]
    const float* reg_val;        // adjoint: the regularisation states at t1 (device, nreg floats)
    const float* reg_cot;        // adjoint: dL/d(state) (device, nreg floats)
    // the bytes of the caller's workspace this call may touch: [workspace, workspace + cnf_workspace_bytes(net, rows)) -- what a
    // caller that allocates exactly the advertised size owns.  Read by the debug build only (dbg_ws below).
    const char* ws_lo;
    const char* ws_hi;
};

// ---- debug build: `make debug` (-DCNF_DEBUG_BOUNDS; libraries in gan-flow_amd/debug/, loaded through CNF_HIP_LIBDIR) ----------
// Every sub-range of the workspace that a kernel derives from its block index, a slot index or the launch geometry -- gradient
// slabs, stage slots and running totals, weight-piece areas, the granule banks -- passes through dbg_ws() where the pointer is
// formed.  Release build: the identity, no code.  Debug build: a range outside [ws_lo, ws_hi) is reported (site = source line,
// block, wave, offsets), the solve's status word is raised to CNF_ERR_DEBUG_BOUNDS, and the pointer is redirected to the
// workspace's first bytes so that the run ends with a status instead of a GPU fault (a fault can take the whole node down).
// Row-indexed inputs and outputs are clamped to [0, rows) where the index is formed, and the parameter vector is indexed by
// compile-time layouts: neither needs a run-time check.
template <class T>
__device__ __forceinline__ T* dbg_ws(const SolveArgs& a, T* p, size_t bytes, int site) {
#ifdef CNF_DEBUG_BOUNDS
    const char* c = reinterpret_cast<const char*>(p);
    if (a.ws_hi != nullptr && (c < a.ws_lo || c + bytes > a.ws_hi)) {
        if ((threadIdx.x & 63) == 0)
            printf("CNF_DEBUG_BOUNDS line %d block %d wave %d: range [%ld, %ld) of a workspace of %ld bytes\n", site, (int)blockIdx.x,
                   (int)(threadIdx.x >> 6), (long)(c - a.ws_lo), (long)(c - a.ws_lo) + (long)bytes, (long)(a.ws_hi - a.ws_lo));
        if (a.stats) atomicMax(&a.stats->status, (int)CNF_ERR_DEBUG_BOUNDS);
        return reinterpret_cast<T*>(const_cast<char*>(a.ws_lo));
    }
#endif
    return p;
}
// the granule banks and the ticket word behind them (GridSync, cnf_device.h): kGranWords + 1 eight-byte words, and a grid that
// fits the banks
__device__ __forceinline__ unsigned long long* dbg_gran(const SolveArgs& a, int site) {
#ifdef CNF_DEBUG_BOUNDS
    if (gridDim.x > (unsigned)kMaxWG) {
        if (threadIdx.x == 0) printf("CNF_DEBUG_BOUNDS line %d: grid of %d workgroups, granule banks hold %d\n", site, (int)gridDim.x, kMaxWG);
        if (a.stats) atomicMax(&a.stats->status, (int)CNF_ERR_DEBUG_BOUNDS);
    }
#endif
    return dbg_ws(a, a.gran, (size_t)(kGranWords + 1) * sizeof(unsigned long long), site);
}

hipError_t launch_field(int D, bool hutch, const float* params, float t, const float* z, const float* e, long rows,
                        float* f_out, float* nd_out, hipStream_t s, float res = 0.f, bool b16 = false);
hipError_t launch_solve_fwd(int D, bool hutch, const SolveArgs& a, int solver, bool global, hipStream_t s);
// d = 16, Hutchinson, rows <= 4096, x / e / z_out / params 16-byte aligned (cnf_split16.hip)
hipError_t launch_split16_fwd(const SolveArgs& a, hipStream_t s);
hipError_t launch_split16_bwd(const SolveArgs& a, hipStream_t s);   // + grad_z1 / grad_x aligned

// Batch-global kernels spin on grid-wide reductions, so every workgroup of the launch must be resident at once.  The API
// layer bounds the grid by the device's compute-unit count; this asks the runtime what THIS kernel's registers and LDS
// actually allow (workgroups per compute unit x compute units; cached per kernel and device, so the query runs once, outside
// any stream capture when the caller warms up first) and rejects a grid that could not be co-resident on an idle device
// instead of letting it run into the spin limit.  It cannot see other tenants of the GPU: a launch that shares the device
// with foreign work still ends in CNF_ERR_SYNC_TIMEOUT (bounded spin), not in a hang.
template <class K>
inline hipError_t check_resident(K kernel, int threads, int grid) {
    // one table per kernel SIGNATURE (K is the function-pointer type): every kernel x device of that signature shares it.
    // Entries are written under the mutex and published by the release store of the count; readers scan the published
    // prefix without the lock (launch path, also under stream capture).
    struct Entry { const void* fn; int dev; int capacity; };
    constexpr int kSlots = 512;
    static Entry cache[kSlots];
    static std::atomic<int> n_cached{0};
    static std::mutex mu;
    if (grid <= 1) return hipSuccess;
    int dev = 0;
    hipError_t err = hipGetDevice(&dev);
    if (err != hipSuccess) return err;
    const void* fn = reinterpret_cast<const void*>(kernel);
    const int n = n_cached.load(std::memory_order_acquire);
    for (int i = 0; i < n; ++i)
        if (cache[i].fn == fn && cache[i].dev == dev) return grid <= cache[i].capacity ? hipSuccess : hipErrorCooperativeLaunchTooLarge;
    int per_cu = 0, cus = 0;
    err = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, threads, 0);
    if (err == hipSuccess) err = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (err != hipSuccess) return err;
    const int capacity = per_cu * cus;
    {
        std::lock_guard<std::mutex> lock(mu);
        const int m = n_cached.load(std::memory_order_relaxed);
        bool have = false;
        for (int i = 0; i < m; ++i) have |= cache[i].fn == fn && cache[i].dev == dev;
        if (!have && m < kSlots) {
            cache[m] = Entry{fn, dev, capacity};
            n_cached.store(m + 1, std::memory_order_release);
        }
    }
    return grid <= capacity ? hipSuccess : hipErrorCooperativeLaunchTooLarge;
}

// diagnostics (cnf_debug_kernel_timing): when enabled, record a HIP event on `s` right before / after a solve kernel
void dbg_timing_begin(hipStream_t s);
void dbg_timing_end(hipStream_t s);

}  // namespace cnf
