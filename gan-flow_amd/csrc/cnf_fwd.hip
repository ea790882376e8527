// cnf_fwd.hip -- forward kernels: one field evaluation, fixed-grid RK4, adaptive dopri5 (gfx950).
//
// These replace, inside CNF.forward, torchdiffeq's `odeint(func, (z, logp), [t0, t1], method=...)`
// (controller restated from Appendix B).  One persistent launch
// integrates the whole interval: weights are staged in LDS once, each wave keeps the state of its
// 16 rows in registers, the Runge-Kutta slopes in wave-private LDS.
#include <type_traits>

#include "cnf_launch.h"

namespace cnf {

// ------------------------------------------------------------------------------------------------
// single evaluation (ODEfunc.forward)
// ------------------------------------------------------------------------------------------------
template <int D, bool HUTCH, bool B16 = false>   // B16: hidden layers on the bf16 pipe (cnf_debug_eval_field_b16)
__global__ __launch_bounds__(kThreads) void field_kernel(const float* __restrict__ params, float t,
                                                         const float* __restrict__ z, const float* __restrict__ e,
                                                         long rows, float* __restrict__ f_out,
                                                         float* __restrict__ neg_div_out, float res) {
    __shared__ typename std::conditional<B16, NetLdsB16<D>, NetLds<D>>::type net;
    __shared__ WaveLds<D> wl[kNW];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, j = lane & 15;
    if constexpr (B16) {
        stage_net_b16(net, params, tid, kThreads);
    } else {
        stage_net(net, params, tid);
    }
    __syncthreads();
    const long base = ((long)blockIdx.x * kNW + w) * kRows;
    if (base >= rows) return;
    const long row = base + j;
    const long rr = row < rows ? row : rows - 1;
    float zz[D], ee[D], f[D], div;
#pragma unroll
    for (int m = 0; m < D; ++m) {
        zz[m] = z[rr * D + m];
        ee[m] = HUTCH ? e[rr * D + m] : 0.f;
    }
    if constexpr (B16) {
        field_fwd_b16<D, HUTCH>(net, wl[w], lane, t, zz, ee, f, div);
    } else {
        field_fwd<D, HUTCH>(net, wl[w], lane, t, zz, ee, f, div);
    }
    #pragma unroll
    for (int m = 0; m < D; ++m) f[m] = fmaf(-res, zz[m], f[m]);   // residual dynamics f(z) - z (res = 0 or 1)
    div -= res * (float)D;
    if (lane < 16 && row < rows) {
#pragma unroll
        for (int m = 0; m < D; ++m) f_out[row * D + m] = f[m];
        neg_div_out[row] = -div;
    }
}

// Tile-local and fixed-grid solves are throughput work: waves of different tiles share a SIMD so that one wave's tanh / gate / Runge-
// Kutta arithmetic runs in the shadow of another's MFMAs.  Occupancy is what buys that overlap: the register budget
// is capped for kEvalWaves waves per SIMD (measured on 2^22 rows of the trained stiff field: 2 waves 81, 3 waves 91
// TFLOP/s at d=1), and a workgroup is NW waves that share one copy of the weight images.
#ifndef CNF_EVAL_WAVES
#define CNF_EVAL_WAVES 3
#endif
#ifndef CNF_EVAL_NW
#define CNF_EVAL_NW 4
#endif
constexpr int kEvalWaves = CNF_EVAL_WAVES, kEvalNW = CNF_EVAL_NW;
// Tile-local dopri5 at d <= 2 runs its hidden layers on the bf16 matrix pipe with split operands (field_fwd_b16,
// cnf_device.h).  Its three-piece weight image is 48 KB: six waves share one, so that two workgroups (twelve waves =
// three per SIMD) fit the 160 KB of a compute unit.  -DCNF_EVAL_BF16=0 builds the fp32-MFMA kernel instead.
#ifndef CNF_EVAL_BF16
#define CNF_EVAL_BF16 1
#endif
#ifndef CNF_EVAL_B16_NW
#define CNF_EVAL_B16_NW 4
#endif
#ifndef CNF_EVAL_B16_WAVES
#define CNF_EVAL_B16_WAVES 2
#endif
constexpr bool kEvalB16 = CNF_EVAL_BF16 != 0;
constexpr int kEvalB16NW = CNF_EVAL_B16_NW, kEvalB16Waves = CNF_EVAL_B16_WAVES;

// ------------------------------------------------------------------------------------------------
// fixed-grid RK4, 3/8 rule (Appendix B.6).  No cross-row coupling: waves are independent.
// ------------------------------------------------------------------------------------------------
template <int D, bool HUTCH>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(D > 2 ? 1 : kEvalWaves, D > 2 ? 8 : kEvalWaves)))
void rk4_kernel(SolveArgs a) {
    if (a.after && a.after->status != 0) {   // CnfSolverOpts.after: the solve whose output this one consumes failed
        if (blockIdx.x == 0 && threadIdx.x == 0) atomicMax(&a.stats->status, a.after->status);
        return;
    }
    const float res = a.residual ? 1.f : 0.f;   // CNF_DIV_RESIDUAL: dynamics f(z) - z
    __shared__ NetLds<D> net;
    __shared__ WaveLds<D> wl[kNW];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, j = lane & 15;
    stage_net(net, a.params, tid);
    __syncthreads();
    const long base = ((long)blockIdx.x * kNW + w) * kRows;
    if (base >= a.rows) return;
    const long row = base + j;
    const long rr = row < a.rows ? row : a.rows - 1;
    constexpr int C = D + 1;
    float y[C], ee[D];
#pragma unroll
    for (int m = 0; m < D; ++m) {
        y[m] = a.x[rr * D + m];
        ee[m] = HUTCH ? a.e[rr * D + m] : 0.f;
    }
    y[D] = a.logp0 ? a.logp0[rr] : 0.f;
    const float sgn = a.t1 >= a.t0 ? 1.f : -1.f;
    const float tau0 = sgn * a.t0, tau1 = sgn * a.t1;

    for (int step = 0; step < a.n_grid; ++step) {
        // grid point i = i*step_size + tau0 (product rounded before the add, like arange()*h + t0)
        float ta = a.n_grid == 1 ? tau0 : __fadd_rn(__fmul_rn((float)step, a.step_size), tau0);
        float tb = (step + 1 == a.n_grid) ? tau1 : __fadd_rn(__fmul_rn((float)(step + 1), a.step_size), tau0);
        const float dt = tb - ta;
        float k1[C], k2[C], k3[C], yi[C];
        float ti = ta;
#pragma unroll
        for (int c = 0; c < C; ++c) yi[c] = y[c];
        const int nst = a.fixed_kind == CNF_SOLVER_EULER ? 1 : (a.fixed_kind == CNF_SOLVER_MIDPOINT ? 2 : 4);
        for (int st = 0; st < nst; ++st) {
            float zz[D], f[D], div, k[C];
#pragma unroll
            for (int m = 0; m < D; ++m) zz[m] = yi[m];
            field_fwd<D, HUTCH>(net, wl[w], lane, sgn * ti, zz, ee, f, div);
            #pragma unroll
            for (int m = 0; m < D; ++m) f[m] = fmaf(-res, zz[m], f[m]);   // residual dynamics f(z) - z (res = 0 or 1)
            div -= res * (float)D;
#pragma unroll
            for (int m = 0; m < D; ++m) k[m] = sgn * f[m];
            k[D] = -sgn * div;
            if (a.fixed_kind == CNF_SOLVER_EULER) {            // y + dt f(t, y)
#pragma unroll
                for (int c = 0; c < C; ++c) y[c] = y[c] + dt * k[c];
            } else if (a.fixed_kind == CNF_SOLVER_MIDPOINT) {  // y + dt f(t + dt/2, y + f(t, y) dt/2)
                if (st == 0) {
                    ti = ta + dt / 2;
#pragma unroll
                    for (int c = 0; c < C; ++c) yi[c] = y[c] + k[c] * dt / 2;
                } else {
#pragma unroll
                    for (int c = 0; c < C; ++c) y[c] = y[c] + dt * k[c];
                }
            } else if (st == 0) {
                ti = ta + dt / 3;
#pragma unroll
                for (int c = 0; c < C; ++c) { k1[c] = k[c]; yi[c] = y[c] + dt * k[c] / 3; }
            } else if (st == 1) {
                ti = ta + dt * 2 / 3;
#pragma unroll
                for (int c = 0; c < C; ++c) { k2[c] = k[c]; yi[c] = y[c] + dt * (k1[c] / -3 + k[c]); }
            } else if (st == 2) {
                ti = ta + dt;
#pragma unroll
                for (int c = 0; c < C; ++c) { k3[c] = k[c]; yi[c] = y[c] + dt * (k1[c] - k2[c] + k[c]); }
            } else {
#pragma unroll
                for (int c = 0; c < C; ++c) y[c] = y[c] + (k1[c] + 3 * k2[c] + 3 * k3[c] + k[c]) * (dt / 8);
            }
        }
    }
    if (lane < 16 && row < a.rows) {
#pragma unroll
        for (int m = 0; m < D; ++m) a.z_out[row * D + m] = y[m];
        a.logp_out[row] = y[D];
    }
    if (lane == 0) {
        const int nst_total = a.fixed_kind == CNF_SOLVER_EULER ? 1 : (a.fixed_kind == CNF_SOLVER_MIDPOINT ? 2 : 4);
        atomicMax(&a.stats->nfe, nst_total * a.n_grid);
        atomicMax(&a.stats->n_accepted, a.n_grid);
        atomicMax(&a.stats->evals_run, nst_total * a.n_grid);
        long nrow = a.rows - base < kRows ? a.rows - base : kRows;
        atomicAdd((unsigned long long*)&a.stats->row_evals, (unsigned long long)(nrow * nst_total * a.n_grid));
    }
}

// ------------------------------------------------------------------------------------------------
// adaptive Dormand-Prince 5(4) with FSAL and quartic dense output (Appendix B.2-B.6)
//   GLOBAL = true : one (t, dt) for the whole batch; error ratio = mean over the whole batch per
//                   state tensor (reference semantics) -> one grid-wide reduction per attempt
//   GLOBAL = false: every 16-row tile (one wave) runs its own controller; waves never synchronise
// The loop is a small state machine around ONE inlined copy of the dynamics.
// ------------------------------------------------------------------------------------------------
template <int D, bool HUTCH, bool GLOBAL, int NW, bool B16 = false>
__global__ __launch_bounds__(NW * 64)
    __attribute__((amdgpu_waves_per_eu(GLOBAL || D > 2 ? 1 : (B16 ? kEvalB16Waves : kEvalWaves),
                                       GLOBAL || D > 2 ? 8 : (B16 ? kEvalB16Waves : kEvalWaves))))
    void dopri5_kernel(SolveArgs a) {
    const float res = a.residual ? 1.f : 0.f;   // CNF_DIV_RESIDUAL: dynamics f(z) - z
    __shared__ typename std::conditional<B16, NetLdsB16<D>, NetLds<D>>::type net;
    __shared__ WaveLds<D> wl[NW];
    __shared__ RedLds red;
    // regularised dynamics (cnf_solve_fwd_reg; batch-global solves only): slopes of every row's regulariser integrals
    __shared__ float krs[GLOBAL ? NW : 1][7][CNF_MAX_REG][kRows];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, j = lane & 15;
    constexpr int C = D + 1;
    if (tid == 0) red.status = 0;
    if constexpr (B16) {
        stage_net_b16(net, a.params, tid, NW * 64);
    } else {
        stage_net(net, a.params, tid, NW * 64);
    }
    zero_slopes(wl, NW, tid, NW * 64);
    if constexpr (GLOBAL)
        for (int idx = tid; idx < NW * 7 * CNF_MAX_REG * kRows; idx += NW * 64) (&krs[0][0][0][0])[idx] = 0.f;   // as zero_slopes
    __syncthreads();
    const long base = ((long)blockIdx.x * NW + w) * kRows;
    if (!GLOBAL && base >= a.rows) return;
    const long row = base + j;
    const bool counted = row < a.rows && lane < 16;  // this lane's row enters the norms exactly once
    const long rr = row < a.rows ? row : a.rows - 1;
    WaveLds<D>& ws = wl[w];

    GridSync gs{dbg_gran(a, __LINE__), (int)gridDim.x, 0u};
    double nz, nl;  // element counts of the z and logp tensors of the control group
    if (GLOBAL) {
        nl = (double)a.rows;
    } else {
        nl = (double)(a.rows - base < kRows ? a.rows - base : kRows);
    }
    nz = nl * D;

    float y0[C], y1[C], yi[C], ee[D];
#pragma unroll
    for (int m = 0; m < D; ++m) {
        y0[m] = a.x[rr * D + m];
        ee[m] = HUTCH ? a.e[rr * D + m] : 0.f;
    }
    y0[D] = a.logp0 ? a.logp0[rr] : 0.f;
#pragma unroll
    for (int c = 0; c < C; ++c) yi[c] = y0[c];

    const float sgn = a.t1 >= a.t0 ? 1.f : -1.f;
    const double tau_end = (double)(sgn * a.t1);
    double tau = (double)(sgn * a.t0);  // start of the current step
    double dt = 0.0;
    float dtf = 0.f, t0f = (float)tau, ti = t0f;
    const float rtol = a.rtol, atol = a.atol;

    enum { PH_F0 = 0, PH_F1 = 1, PH_STAGE = 2 };
    int phase = PH_F0, stage = 0;
    int nfe = 0, nacc = 0, nrej = 0, status = 0;
    float h0 = 0.f, d1z = 0.f, d1l = 0.f;
    float sc[C];  // initial-step scale, atol + |y0| * rtol

    // Regularised dynamics: every row integrates its regularisers' integrands with the steps' own weights -- the state the
    // reference integrates is their batch mean, and the integral is linear in the integrand, so the mean is taken once,
    // after the solve (reg_mean_kernel).  With tolerances of 1e20 these states never fail the accept test; what their batch
    // means DO steer is the first step (the heuristic uses rtol[0] / atol[0] for every tensor): one more grid sum in each of
    // its two halves.
    const int nreg = GLOBAL ? a.nreg : 0;
    float r0[CNF_MAX_REG] = {0.f, 0.f}, r1[CNF_MAX_REG] = {0.f, 0.f};
    float d1r[CNF_MAX_REG] = {0.f, 0.f}, kr_f0[CNF_MAX_REG] = {0.f, 0.f};

    // Several output times (cnf_solve_fwd_times; torchdiffeq's odeint(func, y0, t) with len(t) > 2): ONE adaptive
    // integration whose steps ignore the output times; every times[i] is read off the quartic of the accepted step that
    // first reaches it (advance(next_t): `while next_t > t1: step`, then the interpolant of [t0, t1]).  Slab i of the
    // outputs holds the state at times[i]; slab 0 is the input, the last slab is written by the ordinary epilogue.
    const int n_times = a.n_times;
    const long slab_z = a.rows * D, slab_l = a.rows;
    int next_i = 1;
    if (n_times > 2 && counted) {
#pragma unroll
        for (int m = 0; m < D; ++m) a.z_out[row * D + m] = y0[m];
        a.logp_out[row] = y0[D];
    }
    // quartic dense output at te inside the accepted step [tau, tau + dt] (B.6): needs y0, y1 and the seven slopes
    auto dense_output = [&](float te, float (&out)[C]) {
        const float ta = (float)tau, tb = (float)(tau + dt);
        const float x = (te - ta) / (tb - ta);
        const float x2 = x * x, x3 = x2 * x, x4 = x3 * x;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            float mid = 0.f;
#pragma unroll
            for (int s = 0; s < 7; ++s) mid += (dtf * kCMid[s]) * ws.kst[s][c][j];
            mid = y0[c] + mid;
            const float f0 = ws.kst[0][c][j], f1 = ws.kst[6][c][j], u = y0[c], v = y1[c];
            const float ca = (-2 * dtf) * f0 + (2 * dtf) * f1 + -8 * u + -8 * v + 16 * mid;
            const float cb = (5 * dtf) * f0 + (-3 * dtf) * f1 + 18 * u + 14 * v + -32 * mid;
            const float cc = (-4 * dtf) * f0 + dtf * f1 + -11 * u + -5 * v + 16 * mid;
            const float cd = dtf * f0;
            out[c] = ca * x4 + cb * x3 + cc * x2 + cd * x + u;
        }
    };

    if (a.after) status = a.after->status;   // CnfSolverOpts.after: the solve whose output this one consumes failed
    for (; status == 0;) {
        float zz[D], f[D], div, k[C];
        bool restart = false;
#pragma unroll
        for (int m = 0; m < D; ++m) zz[m] = yi[m];
        if constexpr (B16) {
            field_fwd_b16<D, HUTCH>(net, ws, lane, sgn * ti, zz, ee, f, div);
        } else {
            field_fwd<D, HUTCH>(net, ws, lane, sgn * ti, zz, ee, f, div);
        }
        #pragma unroll
        for (int m = 0; m < D; ++m) f[m] = fmaf(-res, zz[m], f[m]);   // residual dynamics f(z) - z (res = 0 or 1)
        div -= res * (float)D;
        ++nfe;
#pragma unroll
        for (int m = 0; m < D; ++m) k[m] = sgn * f[m];
        k[D] = -sgn * div;
        float kr[CNF_MAX_REG] = {0.f, 0.f};   // slopes of this row's regulariser integrals (the solver's time runs with sgn)
        if (GLOBAL && nreg) {
#pragma unroll
            for (int q_ = 0; q_ < CNF_MAX_REG; ++q_)
                if (q_ < nreg) kr[q_] = sgn * reg_integrand<D>(a.reg_kind[q_], f);
        }

        if (phase == PH_F0) {
            // ---- f0 and the first half of the initial-step heuristic (B.2)
            double S[4] = {0, 0, 0, 0};
#pragma unroll
            for (int c = 0; c < C; ++c) {
                ws.kst[0][c][j] = k[c];
                sc[c] = atol + fabsf(y0[c]) * rtol;
                const float u = y0[c] / sc[c], v = k[c] / sc[c];
                if (counted) {
                    S[c < D ? 0 : 1] += (double)u * u;
                    S[c < D ? 2 : 3] += (double)v * v;
                }
            }
            if (GLOBAL) {
                if (!wg_allreduce4(gs, red, S, w, lane)) { status = CNF_ERR_SYNC_TIMEOUT; break; }
            } else {
                wave_allreduce4(S);
            }
            const float d0z = (float)sqrt(S[0] / nz), d0l = (float)sqrt(S[1] / nl);
            d1z = (float)sqrt(S[2] / nz);
            d1l = (float)sqrt(S[3] / nl);
            if (GLOBAL && nreg) {
                double S2[4] = {0, 0, 0, 0};
#pragma unroll
                for (int q_ = 0; q_ < CNF_MAX_REG; ++q_) {
                    krs[w][0][q_][j] = kr[q_];
                    kr_f0[q_] = kr[q_];
                    if (counted) S2[q_] = (double)kr[q_];
                }
                if (!wg_allreduce4(gs, red, S2, w, lane)) { status = CNF_ERR_SYNC_TIMEOUT; break; }
                const float d0r[CNF_MAX_REG] = {0.f, 0.f};   // the states start at 0: scale = atol
#pragma unroll
                for (int q_ = 0; q_ < CNF_MAX_REG; ++q_) d1r[q_] = fabsf((float)(S2[q_] / nl) / atol);
                h0 = initial_h0_reg(d0z, d0l, d1z, d1l, nreg, d0r, d1r);
            } else if (pymax(d0z, d0l) < 1e-5f || pymax(d1z, d1l) < 1e-5f) {
                h0 = 1e-6f;
            } else {
                h0 = 0.01f * pymax(d0z / d1z, d0l / d1l);
            }
#pragma unroll
            for (int c = 0; c < C; ++c) yi[c] = y0[c] + h0 * k[c];
            ti = t0f + h0;
            phase = PH_F1;
            continue;
        }
        if (phase == PH_F1) {
            // ---- second half: curvature estimate, first step size
            double S[4] = {0, 0, 0, 0};
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const float u = (k[c] - ws.kst[0][c][j]) / sc[c];
                if (counted) S[c < D ? 0 : 1] += (double)u * u;
            }
            if (GLOBAL && nreg && counted) {   // the change of the regularisers' batch means: the two free slots of this sum
#pragma unroll
                for (int q_ = 0; q_ < CNF_MAX_REG; ++q_) S[2 + q_] = (double)(kr[q_] - kr_f0[q_]);
            }
            if (GLOBAL) {
                if (!wg_allreduce4(gs, red, S, w, lane)) { status = CNF_ERR_SYNC_TIMEOUT; break; }
            } else {
                wave_allreduce4(S);
            }
            const float d2z = (float)sqrt(S[0] / nz) / h0, d2l = (float)sqrt(S[1] / nl) / h0;
            if (GLOBAL && nreg) {
                float d2r[CNF_MAX_REG];
#pragma unroll
                for (int q_ = 0; q_ < CNF_MAX_REG; ++q_) d2r[q_] = fabsf((float)(S[2 + q_] / nl) / atol) / h0;
                dt = (double)initial_dt_reg(h0, d1z, d1l, d2z, d2l, nreg, d1r, d2r);
            } else {
                float h1;
                if (pymax(d1z, d1l) <= 1e-15f && pymax(d2z, d2l) <= 1e-15f) {
                    h1 = fmaxf(1e-6f, h0 * 1e-3f);
                } else {
                    h1 = powf(0.01f / pymax(pymax(pymax(d1z, d1l), d2z), d2l), 0.2f);
                }
                dt = (double)fminf(100.f * h0, h1);
            }
            restart = true;
        } else {
            // ---- a Runge-Kutta stage finished: k is slope number stage+1
#pragma unroll
            for (int c = 0; c < C; ++c) ws.kst[stage][c][j] = k[c];
            if (GLOBAL && nreg) {
#pragma unroll
                for (int q_ = 0; q_ < CNF_MAX_REG; ++q_) krs[w][stage][q_][j] = kr[q_];
            }
            if (stage < 6) {
                ++stage;
            } else {
                // all seven slopes known: yi is the candidate y1, k its slope (FSAL)
                if (GLOBAL && nreg) {   // the regularisers' candidate: the sixth stage point, like every other component
                    float al6, b6[6];
                    stage_coeffs(6, al6, b6);
#pragma unroll
                    for (int q_ = 0; q_ < CNF_MAX_REG; ++q_) {
                        float acc = 0.f;
#pragma unroll
                        for (int s = 0; s < 6; ++s) acc += (dtf * b6[s]) * krs[w][s][q_][j];
                        r1[q_] = r0[q_] + acc;
                    }
                }
                double S[4] = {0, 0, 0, 0};
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    y1[c] = yi[c];
                    float err = 0.f;
#pragma unroll
                    for (int s = 0; s < 7; ++s) err += (dtf * kCErr[s]) * ws.kst[s][c][j];
                    const float tol = atol + rtol * fmaxf(fabsf(y0[c]), fabsf(y1[c]));
                    const float u = err / tol;
                    if (counted) {
                        S[c < D ? 0 : 1] += (double)u * u;
                        if (!isfinite(y0[c])) S[2] += 1.0;
                    }
                }
                if (GLOBAL) {
                    if (!wg_allreduce4(gs, red, S, w, lane)) { status = CNF_ERR_SYNC_TIMEOUT; break; }
                } else {
                    wave_allreduce4(S);
                }
                if (S[2] > 0.0) { status = CNF_ERR_NONFINITE; break; }
                const float rz = (float)(S[0] / nz), rl = (float)(S[1] / nl);
                const bool accept = (rz <= 1.f) && (rl <= 1.f);
                if (accept) {
                    ++nacc;
                    // output times this step reaches, all but the last one (uniform: times and tau are the same everywhere)
                    while (next_i < n_times - 1 && !((double)(sgn * a.times[next_i]) > tau + dt)) {
                        float yo[C];
                        dense_output(sgn * a.times[next_i], yo);
                        if (counted) {
#pragma unroll
                            for (int m = 0; m < D; ++m) a.z_out[next_i * slab_z + row * D + m] = yo[m];
                            a.logp_out[next_i * slab_l + row] = yo[D];
                        }
                        ++next_i;
                    }
                    if (!(tau_end > tau + dt)) break;  // the step reaches the end time: interpolate below
#pragma unroll
                    for (int c = 0; c < C; ++c) {
                        y0[c] = y1[c];
                        ws.kst[0][c][j] = k[c];
                    }
                    if (GLOBAL && nreg) {
#pragma unroll
                        for (int q_ = 0; q_ < CNF_MAX_REG; ++q_) {
                            r0[q_] = r1[q_];
                            krs[w][0][q_][j] = kr[q_];
                        }
                    }
                    tau = tau + dt;
                } else {
                    ++nrej;
                }
                dt = next_step_size(dt, pymax(rz, rl));
                restart = true;
            }
        }
        // (re)start a step from (tau, y0): after the initial-step heuristic and after every finished attempt
        if (restart) {
            if (!(tau + dt > tau)) { status = CNF_ERR_DT_UNDERFLOW; break; }
            if (nacc + nrej >= a.max_steps) { status = CNF_ERR_MAX_STEPS; break; }
            dtf = (float)dt;
            t0f = (float)tau;
            stage = 1;
            phase = PH_STAGE;
        }
        // input of stage `stage` (computes slope number stage+1): y0 + sum_j (dt*beta[stage-1][j]) k_j
        float alpha_, bc_[6];
        stage_coeffs(stage, alpha_, bc_);
        ti = t0f + alpha_ * dtf;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            float acc = 0.f;
#pragma unroll
            for (int s = 0; s < 6; ++s) acc += (dtf * bc_[s]) * ws.kst[s][c][j];   // zero weights past the current stage
            yi[c] = y0[c] + acc;
        }
    }

    if (status == 0) {
        // the state at the end time, off the last accepted step's quartic
        float yo[C];
        dense_output((float)tau_end, yo);
        const long last = n_times > 2 ? n_times - 1 : 0;
        if (counted) {
#pragma unroll
            for (int m = 0; m < D; ++m) a.z_out[last * slab_z + row * D + m] = yo[m];
            a.logp_out[last * slab_l + row] = yo[D];
        }
        if (GLOBAL && nreg) {   // this row's regulariser integrals at the end time, off the same quartic
            const float ta = (float)tau, tb = (float)(tau + dt);
            const float x = ((float)tau_end - ta) / (tb - ta);
#pragma unroll
            for (int q_ = 0; q_ < CNF_MAX_REG; ++q_) {
                float ks_[7];
#pragma unroll
                for (int s = 0; s < 7; ++s) ks_[s] = krs[w][s][q_][j];
                const float v = dense_quartic(dtf, x, r0[q_], r1[q_], ks_);
                if (counted && q_ < nreg) dbg_ws(a, a.reg_rows, sizeof(float) * nreg * a.rows, __LINE__)[(long)q_ * a.rows + row] = v;
            }
        }
    }
    if (lane == 0 && (!GLOBAL || (blockIdx.x == 0 && w == 0) || status != 0)) {
        if (status) atomicMax(&a.stats->status, status);
        atomicMax(&a.stats->nfe, nfe);
        atomicMax(&a.stats->n_accepted, nacc);
        atomicMax(&a.stats->n_rejected, nrej);
        atomicMax(&a.stats->evals_run, nfe);
        a.stats->last_dt = (float)dt;
        const long nrow = GLOBAL ? a.rows : (a.rows - base < kRows ? a.rows - base : kRows);
        atomicAdd((unsigned long long*)&a.stats->row_evals, (unsigned long long)(nrow * nfe));
    }
}

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------
template <int D, bool HUTCH, bool B16>
static hipError_t launch_field_t(const float* params, float t, const float* z, const float* e, long rows,
                                 float* f_out, float* nd_out, hipStream_t s, float res) {
    const int grid = (int)((rows + kRowsPerWG - 1) / kRowsPerWG);
    hipLaunchKernelGGL((field_kernel<D, HUTCH, B16>), dim3(grid), dim3(kThreads), 0, s, params, t, z, e, rows, f_out, nd_out, res);
    return hipGetLastError();
}

hipError_t launch_field(int D, bool hutch, const float* params, float t, const float* z, const float* e, long rows,
                        float* f_out, float* nd_out, hipStream_t s, float res, bool b16) {
    if (b16) {   // the arithmetic of the tile-local / split forward solves (d <= 2)
        if (D == 1 && !hutch) return launch_field_t<1, false, true>(params, t, z, e, rows, f_out, nd_out, s, res);
        if (D == 2 && !hutch) return launch_field_t<2, false, true>(params, t, z, e, rows, f_out, nd_out, s, res);
        if (D == 2 && hutch) return launch_field_t<2, true, true>(params, t, z, e, rows, f_out, nd_out, s, res);
        return hipErrorInvalidValue;
    }
    if (D == 1 && !hutch) return launch_field_t<1, false, false>(params, t, z, e, rows, f_out, nd_out, s, res);
    if (D == 2 && !hutch) return launch_field_t<2, false, false>(params, t, z, e, rows, f_out, nd_out, s, res);
    if (D == 2 && hutch) return launch_field_t<2, true, false>(params, t, z, e, rows, f_out, nd_out, s, res);
    if (D == 16 && hutch) return launch_field_t<16, true, false>(params, t, z, e, rows, f_out, nd_out, s, res);
    return hipErrorInvalidValue;
}

template <int D, bool HUTCH>
static hipError_t launch_solve_t(const SolveArgs& a, int solver, bool global, hipStream_t s) {
    const int grid = (int)((a.rows + kRowsPerWG - 1) / kRowsPerWG);
    dbg_timing_begin(s);
    if (solver != CNF_SOLVER_DOPRI5) {
        hipLaunchKernelGGL((rk4_kernel<D, HUTCH>), dim3(grid), dim3(kThreads), 0, s, a);
    } else if (global) {
        const hipError_t res = check_resident(dopri5_kernel<D, HUTCH, true, kNW>, kThreads, grid);
        if (res != hipSuccess) return res;
        hipLaunchKernelGGL((dopri5_kernel<D, HUTCH, true, kNW>), dim3(grid), dim3(kThreads), 0, s, a);
    } else {
        constexpr bool B16 = kEvalB16 && D <= 2;
        constexpr int NW = D > 2 ? kNW : (B16 ? kEvalB16NW : kEvalNW);
        const int g = (int)((a.rows + NW * kRows - 1) / (NW * kRows));
        hipLaunchKernelGGL((dopri5_kernel<D, HUTCH, false, NW, B16>), dim3(g), dim3(NW * 64), 0, s, a);
    }
    dbg_timing_end(s);
    return hipGetLastError();
}

// batch means of the per-row regulariser integrals (dopri5_kernel, a.reg_rows): one workgroup per regulariser, a fixed
// summation order (deterministic), double accumulation
__global__ __launch_bounds__(256) void reg_mean_kernel(const float* __restrict__ reg_rows, long rows, float* __restrict__ out) {
    __shared__ double part[256];
    const float* src = reg_rows + (size_t)blockIdx.x * rows;
    double s = 0.0;
    for (long i = threadIdx.x; i < rows; i += 256) s += (double)src[i];
    part[threadIdx.x] = s;
    __syncthreads();
    for (int h = 128; h > 0; h >>= 1) {
        if ((int)threadIdx.x < h) part[threadIdx.x] += part[threadIdx.x + h];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[blockIdx.x] = (float)(part[0] / (double)rows);
}
hipError_t launch_reg_mean(const float* reg_rows, long rows, int nreg, float* out, hipStream_t s) {
    hipLaunchKernelGGL(reg_mean_kernel, dim3(nreg), dim3(256), 0, s, reg_rows, rows, out);
    return hipGetLastError();
}

hipError_t launch_solve_fwd(int D, bool hutch, const SolveArgs& a, int solver, bool global, hipStream_t s) {
    if (D == 1 && !hutch) return launch_solve_t<1, false>(a, solver, global, s);
    if (D == 2 && !hutch) return launch_solve_t<2, false>(a, solver, global, s);
    if (D == 2 && hutch) return launch_solve_t<2, true>(a, solver, global, s);
    if (D == 16 && hutch) return launch_solve_t<16, true>(a, solver, global, s);
    return hipErrorInvalidValue;
}

}  // namespace cnf
