// cnf_wide.hip -- hidden width 256 (BASELINE config 4: stacked CNF blocks with 256-wide ConcatSquashLinear layers).
//
// At H = 256 a hidden layer is a real dense contraction: 256 x 256 weights = 256 KB per layer, more than the LDS
// holds, 768 MFMAs per layer per 16-row tile.  Decomposition: ONE 16-row tile per WORKGROUP, the four waves split the
// 16 output-feature tiles (wave w owns features 64w .. 64w+63 = 4 MFMA tiles, all streams).  Layer inputs are
// exchanged through LDS in the C/D register layout exactly as in cnf_split.hip (16 tiles x 3 streams = 48 KB per
// buffer, two buffers).  The WEIGHTS are not staged in LDS: with the k order of the C/D-as-B trick, the A operands
// of the four MFMAs of a 16x16 block are W[16 To + (lane & 15)][16 Ti + 4 (lane >> 4) + 0..3] -- 16 contiguous bytes
// of the row-major nn.Linear weight -- so every wave streams the 64 blocks of its output tiles per layer straight
// from the parameter vector (L2-resident, identical for all workgroups) into registers with one
// global_load_dwordx4 per block, software-prefetched one input tile ahead.  Per CU that is 256 KB per layer
// against ~25k cycles of MFMA work per wave, i.e. ~10 B/clk of L2 bandwidth: the layer is MFMA-bound.
// This file is the forward path (field evaluation, dopri5 with batch-global or tile-local control, rk4 / euler / midpoint);
// the adjoint at this width is cnf_wide_bwd.hip (its 2 x 256 x 256 gradient sums per workgroup live in workspace slots).
#include <type_traits>

#include "cnf_wide.h"

#ifndef CNF_WIDE_PF
#define CNF_WIDE_PF 1
#endif
// dopri5 forward solves at this width run their hidden layers on the bf16 matrix pipe with split operands (0: fp32 MFMAs)
#ifndef CNF_WIDE_BF16
#define CNF_WIDE_BF16 1
#endif

namespace cnf {

constexpr bool kWideB16 = CNF_WIDE_BF16 != 0;

// ---- forward dynamics -----------------------------------------------------------------------------------------
template <int D, bool HUTCH, class WS, class XC>
__device__ __forceinline__ void field_fwd_wide(const WideNetLds<D>& n, WS& ws, XC& xc,
                                               const float* __restrict__ params, int lane, int wave, float t,
                                               const float (&z)[D], const float (&e)[D], float (&f)[D], float& div,
                                               const u32x4* __restrict__ wpieces = nullptr) {
    constexpr int NT = HUTCH ? 1 : D;
    constexpr int NS = NT + 1;
    const int q = lane >> 4, j = lane & 15;
    // gates of this wave's features, one per lane
    {
        const int fi = kWF * wave + lane;
#pragma unroll
        for (int l = 0; l < 3; ++l) {
            const float g = sigmoid_f(fmaf(n.vec[l][2][fi], t, n.vec[l][3][fi]));
            ws.gate[l][0][lane] = g;
            ws.gate[l][1][lane] = fmaf(g, n.vec[l][0][fi], n.vec[l][1][fi] * t);
        }
        if (lane < 16) {
            const float g = sigmoid_f(fmaf(n.vec4[2][lane], t, n.vec4[3][lane]));
            ws.gate4[0][lane] = g;
            ws.gate4[1][lane] = fmaf(g, n.vec4[0][lane], n.vec4[1][lane] * t);
        }
        wave_lds_fence();
    }
    f32x4 act[kWTW][NS];
    // layer 1 (K = d): vector ALU
#pragma unroll
    for (int T = 0; T < kWTW; ++T) {
        const int lo = 16 * T + 4 * q, fo = kWF * wave + lo;
        const f32x4 g = *reinterpret_cast<const f32x4*>(&ws.gate[0][0][lo]);
        const f32x4 c = *reinterpret_cast<const f32x4*>(&ws.gate[0][1][lo]);
        f32x4 a = {0.f, 0.f, 0.f, 0.f}, au = {0.f, 0.f, 0.f, 0.f};
        f32x4 wcol[D];
#pragma unroll
        for (int m = 0; m < D; ++m) {
            wcol[m] = *reinterpret_cast<const f32x4*>(&n.w1[m][fo]);
            a += wcol[m] * z[m];
            if (HUTCH) au += wcol[m] * e[m];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float x = act_f(fmaf(g[r], a[r], c[r]));
            const float dg = act_d(x) * g[r];
            act[T][0][r] = x;
            if (HUTCH) {
                act[T][1][r] = dg * au[r];
            } else {
#pragma unroll
                for (int m = 0; m < NT; ++m) act[T][1 + m][r] = dg * wcol[m][r];
            }
        }
    }
    // layers 2, 3: 256 x 256 on the matrix cores, weights streamed from the (L2-resident) parameter vector
#pragma unroll
    for (int l = 0; l < 2; ++l) {
        constexpr int NB = XC::kBuffers;
        const int lb = l % NB;
        if (NB == 1 && l == 1) __syncthreads();   // every wave is past its reads of layer 2's input
        // row 64 wave + 16 T + j of W, columns 16 Ti + 4 q .. + 3
        const float* wrow = params + make_layout(D, kWH).w(1 + l) + (size_t)(kWF * wave + j) * kWH + 4 * q;
        f32x4 acc[kWTW][NS];
        if constexpr (XC::kB16) {
            // bf16 pipe, three-piece operands (cnf_device.h): activations are split by the wave that owns them and published
            // as pieces; the weight pieces of k-step ks (input tiles 2ks, 2ks+1) stream from L2 one k-step ahead.
#pragma unroll
            for (int T = 0; T < kWTW; ++T) {
                const int tg = kWTW * wave + T;
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    uint32_t pc[4][3];
#pragma unroll
                    for (int r = 0; r < 4; ++r) split3_bf16(act[T][s][r], pc[r]);
#pragma unroll
                    for (int p = 0; p < 3; ++p) {
                        uint2 v;
                        v.x = (pc[0][p] >> 16) | pc[1][p];
                        v.y = (pc[2][p] >> 16) | pc[3][p];
                        *reinterpret_cast<uint2*>(&xc.xb16[lb][s][p][tg >> 1][lane][2 * (tg & 1)]) = v;
                    }
                }
            }
            // weight pieces of this wave's 4 output tiles: [layer][Tout][k-step][piece][lane] x 16 bytes, written once per
            // launch by wide_split_weights_kernel; requested one k-step ahead (wide_matmul_b16, cnf_wide.h)
            const u32x4* wp = wpieces + ((size_t)(l * kWT + kWTW * wave) * (kWT / 2) * 3) * 64 + lane;
            WideRing wb;
            wide_wp_first(wp, wb);
            __syncthreads();
            wide_matmul_b16<NS>(xc.xb16[lb], wp, wb, lane, acc);
        } else {
#pragma unroll
        for (int T = 0; T < kWTW; ++T)
#pragma unroll
            for (int s = 0; s < NS; ++s) *reinterpret_cast<f32x4*>(&xc.xb[lb][s][kWTW * wave + T][lane][0]) = act[T][s];
        // ring of kPF + 1 register buffers: the blocks of input tiles Ti + 1 .. Ti + kPF are in flight while tile Ti
        // feeds the matrix pipe (one tile = 48 MFMAs ~ 1.5k cycles at NS = 3; an L2 round trip is of that order)
        constexpr int kPF = CNF_WIDE_PF, kRing = kPF + 1;
        f32x4 wbuf[kRing][kWTW];
#pragma unroll
        for (int p = 0; p < kPF; ++p)
#pragma unroll
            for (int T = 0; T < kWTW; ++T) wbuf[p][T] = *reinterpret_cast<const f32x4*>(wrow + 16 * T * kWH + 16 * p);   // issued before the barrier
        __syncthreads();
#pragma unroll
        for (int T = 0; T < kWTW; ++T)
#pragma unroll
            for (int s = 0; s < NS; ++s) acc[T][s] = f32x4{0.f, 0.f, 0.f, 0.f};
        static_assert(4 % kRing == 0, "ring slots must be compile-time after unrolling by 4");
#pragma unroll 4
        for (int Ti = 0; Ti < kWT; ++Ti) {
            const int Tn = Ti + kPF < kWT ? Ti + kPF : kWT - 1;   // past the end: a redundant load instead of a branch
#pragma unroll
            for (int T = 0; T < kWTW; ++T)
                wbuf[(Ti + kPF) % kRing][T] = *reinterpret_cast<const f32x4*>(wrow + 16 * T * kWH + 16 * Tn);
            f32x4 in[NS];
#pragma unroll
            for (int s = 0; s < NS; ++s) in[s] = *reinterpret_cast<const f32x4*>(&xc.xb[lb][s][Ti][lane][0]);
#pragma unroll
            for (int T = 0; T < kWTW; ++T)
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int s = 0; s < NS; ++s) acc[T][s] = mfma16(wbuf[Ti % kRing][T][r], in[s][r], acc[T][s]);
        }
        }   // fp32 MFMA path
#pragma unroll
        for (int T = 0; T < kWTW; ++T) {
            const int lo = 16 * T + 4 * q;
            const f32x4 g = *reinterpret_cast<const f32x4*>(&ws.gate[l + 1][0][lo]);
            const f32x4 c = *reinterpret_cast<const f32x4*>(&ws.gate[l + 1][1][lo]);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float x = act_f(fmaf(g[r], acc[T][0][r], c[r]));
                const float dg = act_d(x) * g[r];
                act[T][0][r] = x;
#pragma unroll
                for (int s = 1; s < NS; ++s) act[T][s][r] = dg * acc[T][s][r];
            }
        }
    }
    // layer 4 (N = d): this wave's 64 features, then across waves through LDS
#pragma unroll
    for (int m = 0; m < D; ++m) {
        float pf = 0.f, pd = 0.f;
#pragma unroll
        for (int T = 0; T < kWTW; ++T) {
            const f32x4 w4 = *reinterpret_cast<const f32x4*>(&n.w4[m][kWF * wave + 16 * T + 4 * q]);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                pf = fmaf(w4[r], act[T][0][r], pf);
                pd = fmaf(w4[r], act[T][HUTCH ? 1 : 1 + m][r], pd);
            }
        }
        pf = xq_sum(pf);
        pd = xq_sum(pd);
        if (q == 0) {
            xc.part[wave][m][j] = pf;
            xc.part[wave][D + m][j] = pd;
        }
    }
    __syncthreads();
    div = 0.f;
#pragma unroll
    for (int m = 0; m < D; ++m) {
        const float g4 = ws.gate4[0][m];
        const float u4 = wsum4(xc.part[0][m][j], xc.part[1][m][j], xc.part[2][m][j], xc.part[3][m][j]);
        const float U4 = wsum4(xc.part[0][D + m][j], xc.part[1][D + m][j], xc.part[2][D + m][j], xc.part[3][D + m][j]);
        f[m] = fmaf(g4, u4, ws.gate4[1][m]);
        const float dm = g4 * U4;
        div += HUTCH ? dm * e[m] : dm;
    }
}

// ---- one evaluation (cnf_eval_field) ------------------------------------------------------------------------------
template <int D, bool HUTCH>
__global__ __launch_bounds__(kThreads) void wide_field_kernel(const float* __restrict__ params,
                                                              float t, const float* __restrict__ z, const float* __restrict__ e,
                                                              long rows, float* __restrict__ f_out, float* __restrict__ nd_out,
                                                              float res) {
    constexpr int NS = (HUTCH ? 1 : D) + 1;
    __shared__ WideNetLds<D> net;
    __shared__ WideWaveLds<D + 1> wl[kNW];
    __shared__ WideXchg<D, NS> xc;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, j = lane & 15;
    wide_stage_net(net, params, tid);
    __syncthreads();
    const long row = (long)blockIdx.x * kRows + j;
    const long rr = row < rows ? row : rows - 1;
    float zz[D], ee[D], f[D], div;
#pragma unroll
    for (int m = 0; m < D; ++m) {
        zz[m] = z[rr * D + m];
        ee[m] = HUTCH ? e[rr * D + m] : 0.f;
    }
    field_fwd_wide<D, HUTCH>(net, wl[w], xc, params, lane, w, t, zz, ee, f, div);
    #pragma unroll
    for (int m = 0; m < D; ++m) f[m] = fmaf(-res, zz[m], f[m]);   // residual dynamics f(z) - z (res = 0 or 1)
    div -= res * (float)D;
    if (w == 0 && lane < 16 && row < rows) {
#pragma unroll
        for (int m = 0; m < D; ++m) f_out[row * D + m] = f[m];
        nd_out[row] = -div;
    }
}

// ---- fixed-grid RK4 (3/8 rule), as rk4_kernel in cnf_fwd.hip ---------------------------------------------------
template <int D, bool HUTCH>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(2, 2))) void wide_rk4_kernel(SolveArgs a) {
    if (a.after && a.after->status != 0) {   // CnfSolverOpts.after: the solve whose output this one consumes failed
        if (blockIdx.x == 0 && threadIdx.x == 0) atomicMax(&a.stats->status, a.after->status);
        return;
    }
    const float res = a.residual ? 1.f : 0.f;   // CNF_DIV_RESIDUAL: dynamics f(z) - z
    constexpr int NS = (HUTCH ? 1 : D) + 1;
    constexpr int C = D + 1;
    __shared__ WideNetLds<D> net;
    __shared__ WideWaveLds<C> wl[kNW];
    __shared__ WideXchg<D, NS, 1> xc;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, j = lane & 15;
    wide_stage_net(net, a.params, tid);
    __syncthreads();
    const long base = (long)blockIdx.x * kRows;
    const long row = base + j;
    const long rr = row < a.rows ? row : a.rows - 1;
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
            field_fwd_wide<D, HUTCH>(net, wl[w], xc, a.params, lane, w, sgn * ti, zz, ee, f, div);
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
    if (w == 0 && lane < 16 && row < a.rows) {
#pragma unroll
        for (int m = 0; m < D; ++m) a.z_out[row * D + m] = y[m];
        a.logp_out[row] = y[D];
    }
    if (tid == 0) {
        const int nst_total = a.fixed_kind == CNF_SOLVER_EULER ? 1 : (a.fixed_kind == CNF_SOLVER_MIDPOINT ? 2 : 4);
        atomicMax(&a.stats->nfe, nst_total * a.n_grid);
        atomicMax(&a.stats->n_accepted, a.n_grid);
        atomicMax(&a.stats->evals_run, nst_total * a.n_grid);
        long nrow = a.rows - base < kRows ? a.rows - base : kRows;
        atomicAdd((unsigned long long*)&a.stats->row_evals, (unsigned long long)(nrow * nst_total * a.n_grid));
    }
}

// ---- dopri5: the controller of dopri5_split_kernel (cnf_split.hip); GLOBAL = false: this tile is the control group
// Tile-local: two workgroups per CU (single exchange buffer, registers capped at 256), so that one workgroup's
// barriers, tanh and weight-stream latency are covered by the other's MFMAs.
template <int D, bool HUTCH, bool GLOBAL>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(GLOBAL || kWideB16 ? 1 : 2, GLOBAL || kWideB16 ? 8 : 2)))
void wide_dopri5_kernel(SolveArgs a) {   // (bf16 path: no register cap -- its piece buffers make d = 2 one workgroup per CU anyway, and no spills)
    const float res = a.residual ? 1.f : 0.f;   // CNF_DIV_RESIDUAL: dynamics f(z) - z
    constexpr int NS = (HUTCH ? 1 : D) + 1;
    constexpr int C = D + 1;
    __shared__ WideNetLds<D> net;
    __shared__ WideWaveLds<C> wl[kNW];
    __shared__ typename std::conditional<kWideB16, WideXchgB16<D, NS, 1>, WideXchg<D, NS, GLOBAL ? 2 : 1>>::type xc;
    __shared__ RedLds red;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, j = lane & 15;
    if (tid == 0) red.status = 0;
    wide_stage_net(net, a.params, tid);
    zero_slopes(wl, kNW, tid);
    __syncthreads();
    const long base = (long)blockIdx.x * kRows;
    const long row = base + j;
    // GLOBAL: every wave carries the same 16 rows, wave 0 reports them to the grid-wide sum.  Tile-local: every wave
    // reduces its own (identical) copy, so all four reach the same decision without a barrier.
    const bool counted = row < a.rows && lane < 16 && (GLOBAL ? w == 0 : true);
    const long rr = row < a.rows ? row : a.rows - 1;
    WideWaveLds<C>& ws = wl[w];

    GridSync gs{dbg_gran(a, __LINE__), (int)gridDim.x, 0u};
    const double nl = GLOBAL ? (double)a.rows : (double)(a.rows - base < kRows ? a.rows - base : kRows);
    const double nz = nl * D;

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
    double tau = (double)(sgn * a.t0);
    double dt = 0.0;
    float dtf = 0.f, t0f = (float)tau, ti = t0f;
    const float rtol = a.rtol, atol = a.atol;

    enum { PH_F0 = 0, PH_F1 = 1, PH_STAGE = 2 };
    int phase = PH_F0, stage = 0;
    int nfe = 0, nacc = 0, nrej = 0, status = 0;
    float h0 = 0.f, d1z = 0.f, d1l = 0.f;
    float sc[C];

    auto reduce = [&](double (&S)[4]) -> bool {
        if (GLOBAL) return wg_allreduce4(gs, red, S, w, lane);
        wave_allreduce4(S);
        return true;
    };

    if (a.after) status = a.after->status;   // CnfSolverOpts.after: the solve whose output this one consumes failed
    for (; status == 0;) {
        float zz[D], f[D], div, k[C];
        bool restart = false;
#pragma unroll
        for (int m = 0; m < D; ++m) zz[m] = yi[m];
        field_fwd_wide<D, HUTCH>(net, ws, xc, a.params, lane, w, sgn * ti, zz, ee, f, div, reinterpret_cast<const u32x4*>(dbg_ws(a, a.slab, kWidePieceBytes, __LINE__)));
        #pragma unroll
        for (int m = 0; m < D; ++m) f[m] = fmaf(-res, zz[m], f[m]);   // residual dynamics f(z) - z (res = 0 or 1)
        div -= res * (float)D;
        ++nfe;
#pragma unroll
        for (int m = 0; m < D; ++m) k[m] = sgn * f[m];
        k[D] = -sgn * div;

        if (phase == PH_F0) {
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
            if (!reduce(S)) { status = CNF_ERR_SYNC_TIMEOUT; break; }
            const float d0z = (float)sqrt(S[0] / nz), d0l = (float)sqrt(S[1] / nl);
            d1z = (float)sqrt(S[2] / nz);
            d1l = (float)sqrt(S[3] / nl);
            if (pymax(d0z, d0l) < 1e-5f || pymax(d1z, d1l) < 1e-5f) {
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
            double S[4] = {0, 0, 0, 0};
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const float u = (k[c] - ws.kst[0][c][j]) / sc[c];
                if (counted) S[c < D ? 0 : 1] += (double)u * u;
            }
            if (!reduce(S)) { status = CNF_ERR_SYNC_TIMEOUT; break; }
            const float d2z = (float)sqrt(S[0] / nz) / h0, d2l = (float)sqrt(S[1] / nl) / h0;
            float h1;
            if (pymax(d1z, d1l) <= 1e-15f && pymax(d2z, d2l) <= 1e-15f) {
                h1 = fmaxf(1e-6f, h0 * 1e-3f);
            } else {
                h1 = powf(0.01f / pymax(pymax(pymax(d1z, d1l), d2z), d2l), 0.2f);
            }
            dt = (double)fminf(100.f * h0, h1);
            restart = true;
        } else {
#pragma unroll
            for (int c = 0; c < C; ++c) ws.kst[stage][c][j] = k[c];
            if (stage < 6) {
                ++stage;
            } else {
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
                if (!reduce(S)) { status = CNF_ERR_SYNC_TIMEOUT; break; }
                if (S[2] > 0.0) { status = CNF_ERR_NONFINITE; break; }
                const float rz = (float)(S[0] / nz), rl = (float)(S[1] / nl);
                const bool accept = (rz <= 1.f) && (rl <= 1.f);
                if (accept) {
                    ++nacc;
                    if (!(tau_end > tau + dt)) break;
#pragma unroll
                    for (int c = 0; c < C; ++c) {
                        y0[c] = y1[c];
                        ws.kst[0][c][j] = k[c];
                    }
                    tau = tau + dt;
                } else {
                    ++nrej;
                }
                dt = next_step_size(dt, pymax(rz, rl));
                restart = true;
            }
        }
        if (restart) {
            if (!(tau + dt > tau)) { status = CNF_ERR_DT_UNDERFLOW; break; }
            if (nacc + nrej >= a.max_steps) { status = CNF_ERR_MAX_STEPS; break; }
            dtf = (float)dt;
            t0f = (float)tau;
            stage = 1;
            phase = PH_STAGE;
        }
        float alpha_, bc_[6];
        stage_coeffs(stage, alpha_, bc_);
        ti = t0f + alpha_ * dtf;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            float acc = 0.f;
#pragma unroll
            for (int s = 0; s < 6; ++s) acc += (dtf * bc_[s]) * ws.kst[s][c][j];
            yi[c] = y0[c] + acc;
        }
    }

    if (status == 0) {
        const float ta = (float)tau, tb = (float)(tau + dt), te = (float)tau_end;
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
            y1[c] = ca * x4 + cb * x3 + cc * x2 + cd * x + u;
        }
        if (w == 0 && lane < 16 && row < a.rows) {
#pragma unroll
            for (int m = 0; m < D; ++m) a.z_out[row * D + m] = y1[m];
            a.logp_out[row] = y1[D];
        }
    }
    if (GLOBAL) {
        grid_cleanup(gs, red, tid);
        write_stats_global(a.stats, status, nfe, nacc, nrej, nfe, (float)dt, a.rows, tid);
    } else if (tid == 0) {
        if (status) atomicMax(&a.stats->status, status);
        atomicMax(&a.stats->nfe, nfe);
        atomicMax(&a.stats->n_accepted, nacc);
        atomicMax(&a.stats->n_rejected, nrej);
        atomicMax(&a.stats->evals_run, nfe);
        if (GLOBAL) a.stats->last_dt = (float)dt;
        const long nrow = GLOBAL ? a.rows : (a.rows - base < kRows ? a.rows - base : kRows);
        atomicAdd((unsigned long long*)&a.stats->row_evals, (unsigned long long)(nrow * (long)nfe));
    }
}

// ---- launchers -----------------------------------------------------------------------------------------------------
template <int D, bool HUTCH>
static hipError_t wide_field_t(const float* params, float t, const float* z, const float* e, long rows,
                               float* f_out, float* nd_out, hipStream_t s, float res) {
    const int grid = (int)((rows + kRows - 1) / kRows);
    hipLaunchKernelGGL((wide_field_kernel<D, HUTCH>), dim3(grid), dim3(kThreads), 0, s, params, t, z, e, rows, f_out, nd_out, res);
    return hipGetLastError();
}

hipError_t launch_wide_field(int D, bool hutch, const float* params, float t, const float* z,
                             const float* e, long rows, float* f_out, float* nd_out, hipStream_t s, float res) {
    if (D == 1 && !hutch) return wide_field_t<1, false>(params, t, z, e, rows, f_out, nd_out, s, res);
    if (D == 2 && !hutch) return wide_field_t<2, false>(params, t, z, e, rows, f_out, nd_out, s, res);
    if (D == 2 && hutch) return wide_field_t<2, true>(params, t, z, e, rows, f_out, nd_out, s, res);
    return hipErrorNotSupported;
}

// The two hidden matrices as three bf16 pieces each, in the operand order of field_fwd_wide's bf16 path:
// out[layer][Tout][k-step][piece][lane] = 8 bf16 of W[16 Tout + (lane & 15)][16 (2ks + (e >> 2)) + 4 (lane >> 4) + (e & 3)].
// 786 KB at the head of the workspace's gradient region (the forward solve uses nothing else of it), once per launch.
__global__ __launch_bounds__(256) void wide_split_weights_kernel(const float* __restrict__ params, int D, u32x4* __restrict__ out) {
    const int idx = blockIdx.x * 256 + threadIdx.x;   // [layer][Tout][k-step][lane]
    if (idx >= 2 * kWT * (kWT / 2) * 64) return;
    const int lane = idx & 63, ks = (idx >> 6) & 7, To = (idx >> 9) & 15, l = idx >> 13;
    const float* w = params + make_layout(D, kWH).w(1 + l) + (size_t)(16 * To + (lane & 15)) * kWH + 4 * (lane >> 4);
    uint32_t pc[8][3];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(w + 16 * (2 * ks + h));
#pragma unroll
        for (int r = 0; r < 4; ++r) split3_bf16(v[r], pc[4 * h + r]);
    }
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        const u32x4 v = {(pc[0][p] >> 16) | pc[1][p], (pc[2][p] >> 16) | pc[3][p], (pc[4][p] >> 16) | pc[5][p], (pc[6][p] >> 16) | pc[7][p]};
        out[((size_t)((l * kWT + To) * (kWT / 2) + ks) * 3 + p) * 64 + lane] = v;
    }
}

template <int D, bool HUTCH>
static hipError_t wide_solve_t(const SolveArgs& a, int solver, bool global, hipStream_t s) {
    const int grid = (int)((a.rows + kRows - 1) / kRows);
    if (kWideB16 && solver == CNF_SOLVER_DOPRI5)
        hipLaunchKernelGGL(wide_split_weights_kernel, dim3(2 * kWT * (kWT / 2) * 64 / 256), dim3(256), 0, s, a.params, D,
                           reinterpret_cast<u32x4*>(a.slab));
    dbg_timing_begin(s);
    if (solver != CNF_SOLVER_DOPRI5) {
        hipLaunchKernelGGL((wide_rk4_kernel<D, HUTCH>), dim3(grid), dim3(kThreads), 0, s, a);
    } else if (global) {
        {
            const hipError_t res = check_resident(wide_dopri5_kernel<D, HUTCH, true>, kThreads, grid);
            if (res != hipSuccess) return res;
        }
        hipLaunchKernelGGL((wide_dopri5_kernel<D, HUTCH, true>), dim3(grid), dim3(kThreads), 0, s, a);
    } else {
        hipLaunchKernelGGL((wide_dopri5_kernel<D, HUTCH, false>), dim3(grid), dim3(kThreads), 0, s, a);
    }
    dbg_timing_end(s);
    return hipGetLastError();
}

hipError_t launch_wide_solve(int D, bool hutch, const SolveArgs& a, int solver, bool global, hipStream_t s) {
    if (D == 1 && !hutch) return wide_solve_t<1, false>(a, solver, global, s);
    if (D == 2 && !hutch) return wide_solve_t<2, false>(a, solver, global, s);
    if (D == 2 && hutch) return wide_solve_t<2, true>(a, solver, global, s);
    return hipErrorNotSupported;
}

}  // namespace cnf
