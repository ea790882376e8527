// cnf_split_fwd_tw.h -- the batch-global forward solve with EIGHT waves per workgroup (included by cnf_split.hip, part 1).
//
// dopri5_split_kernel runs one wave per SIMD and every wave carries all streams of its feature tile: the value stream and
// the d forward-mode tangent streams of the exact divergence.  Nothing hides that wave's latencies (vector ALU busy a third
// of the time, round 2's counters), and only a third of the evaluation's work is on the path the NEXT evaluation waits
// for: the field f needs the value stream alone, and nothing the integrator feeds back into the field depends on the
// divergence -- log p is integrated, error-controlled and returned, but never read by f.
//
// Roles (kFwdTwNote), as in the eight-wave adjoint (kThetaWavesNote):
//   waves 0..3 ("producers", P): the value stream of feature tile w -- gates, first layer, the two 64 x 64 products
//     (12 bf16 MFMAs each instead of 36), tanh, last layer -> f -> the Runge-Kutta state of z.  Beside x they hand over
//     act'(x) * gate per layer (dg: the factor every tangent of that feature is multiplied by).
//   waves 4..7 ("helpers", H, the second wave of each SIMD): the tangent streams of the same feature tile, ONE INTERVAL
//     BEHIND the producers -- tangent of layer l while the producers are at layer l + 1 -- then the divergence, and with it
//     everything of log p: its slopes, its state, its share of the controller's sums, its dense output.
//   Helper groups (kFwdTwGroupsNote): with the exact divergence at d = 2 there are two tangent streams, and a helper that
//     carries both has twice the producers' matrix work per interval -- its chain, not the producers', bounded the
//     eight-wave kernel.  That case runs TWELVE waves, three per SIMD (166 registers each, no spills): waves 4..7 carry
//     tangent stream 0 and own log p, waves 8..11 carry tangent stream 1.  The second group keeps an identical copy of
//     log p's state (it needs the controller's branches and barriers; it is never counted and stores nothing).
//     Measured on the stiff trained field (B = 2 048): 2.74 -> 2.68 us per evaluation; what bounds an evaluation now is the
//     producers' own chain and the three waves' shared vector issue between the last barrier of one evaluation and the first
//     of the next (two further arrangements measured slower and not kept: the next stage's gates computed by a helper
//     one evaluation ahead, 2.72; raised issue priority for the second helper group there, 2.74-2.76).
// An evaluation keeps its three workgroup barriers; what a barrier interval holds is now one producer phase and one helper
// phase side by side on each SIMD, and the dependent chain of an evaluation is the producers' alone.  The helpers' last
// phase of evaluation e (third-layer tangents, last-layer dot products) runs beside the producers' first phase of e + 1;
// before a grid-wide reduction (it needs the divergence of the evaluation just done) one extra barrier drains the helpers.
// Same arithmetic, operation for operation, as dopri5_split_kernel: bit-identical z, log p and step sequence
// (tests/test_gpu_forward.py::test_eight_wave_forward_solve_equals_four_wave).
//
// LDS hand-over buffers and why a reader is always separated from the next writer by a barrier (B1, B2, B3 = the three
// barriers of evaluation e; S0 = before B1, S1 = B1..B2, S2 = B2..B3):
//   xv[0] (x1 pieces): P writes S0(e), P reads S1(e).           xv[1] (x2): P writes S1(e), P reads S2(e).
//   dg[0]: P writes S0(e), H reads S1(e).   dg[1]: P writes S1(e), H reads S2(e).   dg[2]: P writes S2(e), H reads S0(e+1).
//   tp[0] (layer-1 tangents): H writes S1(e), H reads S2(e).    tp[1]: H writes S2(e), H reads S0(e+1).
//   partf: P writes S2(e), P reads S0(e+1).                     partd: H writes S0(e+1), H reads S1(e+1) (or behind the drain barrier).
#pragma once

namespace cnf {

template <int D, bool HUTCH>
struct alignas(16) FwdTwLds {
    static constexpr int NT = HUTCH ? 1 : D;
    uint32_t xv[2][1][3][2][64][4];     // value pieces: input of hidden layer l (publish_b16 layout, one stream)
    uint32_t tp[2][NT][3][2][64][4];    // tangent pieces: input of hidden layer l
    f32x4 dg[3][kNW][64];               // act'(x) * gate of layers 1..3: [layer][feature tile][lane]
    float partf[kNW][D][kRows];         // last layer, value stream: partial sums of every producer
    float partd[kNW][D][kRows];         // last layer, tangent streams: partial sums of every helper
    float kz[7][D][kRows];              // slopes of z (ONE copy: the four producers hold identical values)
};

// grid-wide sum of up to four doubles for an eight-wave workgroup: wave 0 (producer, z's sums) and wave 4 (helper, log p's
// sums) each deposit their 16 rows' partials; a slot owned by one role is 0 in the other's deposit, so the total of a
// workgroup is x + 0 or 0 + x -- the value wg_allreduce4 forms as ((x + 0) + 0) + 0.  Publishing and polling are those of
// wg_reduce_publish / wg_reduce_collect (cnf_device.h).  All threads call it.
__device__ inline bool tw_allreduce4(GridSync& gs, RedLds& red, double (&v)[4], int wv, int lane) {
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = row16_sum(v[i]);
    if (lane == 0 && (wv == 0 || wv == kNW)) {
#pragma unroll
        for (int i = 0; i < 4; ++i) red.part[wv / kNW][i] = v[i];
    }
    __syncthreads();
    if (gs.nwg > 1) gs.epoch += 1;
    if (wv == 0) {
        double s = 0.0;
        if (lane < 4) s = red.part[0][lane] + red.part[1][lane];
        if (gs.nwg > 1) {
            unsigned long long* bank = gs.gran + (size_t)(gs.epoch & 1u) * (size_t)(gs.nwg * 4);
            if (lane < 4) {
                const unsigned long long pk = ((unsigned long long)gs.epoch << 32) | (unsigned long long)__float_as_uint((float)s);
                __hip_atomic_store(bank + blockIdx.x * 4 + lane, pk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            const int ng = gs.nwg * 4;
            double acc = 0.0;
            bool done = false;
            constexpr int kPerLane = kMaxWG * 4 / 64;
            __builtin_amdgcn_s_sleep(CNF_PRE_POLL_SLEEP);   // the first sweep cannot succeed earlier (cnf_device.h)
            for (unsigned spin = 0; spin < kSpinLimit; ++spin) {
                unsigned long long x[kPerLane];
#pragma unroll
                for (int k = 0; k < kPerLane; ++k) {
                    const int g = lane + 64 * k;
                    x[k] = g < ng ? __hip_atomic_load(bank + g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                  : ((unsigned long long)gs.epoch << 32);
                }
                bool ok = true;
                acc = 0.0;
#pragma unroll
                for (int k = 0; k < kPerLane; ++k) {
                    ok &= (unsigned)(x[k] >> 32) == gs.epoch;
                    acc += (double)__uint_as_float((unsigned)x[k]);
                }
                if (__all(ok)) { done = true; break; }
                __builtin_amdgcn_s_sleep(1);
            }
            if (!done && lane == 0) red.status = CNF_ERR_SYNC_TIMEOUT;
            acc += dpp_mov_f64<0x124>(acc);
            acc += dpp_mov_f64<0x128>(acc);
            s = xq_sum_f64(acc);
        }
        if (lane < 4) red.tot[lane] = s;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = red.tot[i];
    return red.status == 0;
}

// what a helper carries from one evaluation into the next (its last phase runs beside the producers' next first phase)
template <int D>
struct FwdTwPending {
    bool on = false;      // an evaluation's tangents are still to be finished
    float g4[D];          // the last layer's gates at that evaluation's time
};

// the helpers' last phase of a pending evaluation: third-layer tangents, last-layer dot products -> partd
// NSH streams from stream s0 on (kFwdTwGroupsNote: a helper group carries all NT tangent streams, or one of them)
template <int D, bool HUTCH, int NSH>
__device__ __forceinline__ void fwd_tw_helper_tail(const NetLdsSmall<D>& n, const SplitWeightsB16& wb, FwdTwLds<D, HUTCH>& xc, int lane, int tile, int s0) {
    const int q = lane >> 4, j = lane & 15, fo = 16 * tile + 4 * q;
    f32x4 U[NSH];
    hidden_b16<NSH>(xc.tp[1] + s0, wb, 1, lane, U);
    const f32x4 dg = xc.dg[2][tile][lane];
    // exact divergence: tangent stream m feeds output m alone; Hutchinson: the one stream feeds every output
#pragma unroll
    for (int k = 0; k < (HUTCH ? D : NSH); ++k) {
        const int m = HUTCH ? k : s0 + k;
        const f32x4 w4 = *reinterpret_cast<const f32x4*>(&n.w4[m][fo]);
        float pd = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) pd = fmaf(w4[r], dg[r] * U[HUTCH ? 0 : k][r], pd);
        pd = xq_sum(pd);
        if (q == 0) xc.partd[tile][m][j] = pd;
    }
}
// ... and its divergence once every helper's partial sums are in LDS (behind a barrier)
template <int D, bool HUTCH>
__device__ __forceinline__ float fwd_tw_helper_div(const FwdTwLds<D, HUTCH>& xc, const FwdTwPending<D>& pend, const float (&e)[D], int j) {
    float div = 0.f;
#pragma unroll
    for (int m = 0; m < D; ++m) {
        const float U4 = sum4(xc.partd[0][m][j], xc.partd[1][m][j], xc.partd[2][m][j], xc.partd[3][m][j]);
        const float dm = pend.g4[m] * U4;
        div += HUTCH ? dm * e[m] : dm;
    }
    return div;
}

// One evaluation, three barriers.  ROLE 1 = producer: returns f (field_fwd_split's value arithmetic).  ROLE 2 = helper:
// first finishes the pending evaluation (tail before B1, divergence behind it: div_prev, valid when pend.on on entry),
// then runs this evaluation's tangents up to the second hidden layer and leaves it pending.
template <int D, bool HUTCH, int ROLE, int NSH, class WS>
__device__ __forceinline__ void fwd_eval_tw(const NetLdsSmall<D>& n, const SplitWeightsB16& wb, WS& ws, FwdTwLds<D, HUTCH>& xc, int lane,
                                            int tile, int s0, float t, const float (&z)[D], const float (&e)[D], float (&f)[D],
                                            FwdTwPending<D>& pend, float& div_prev, Prof& pr) {
    constexpr bool P = ROLE == 1;
    pr.mark(0);   // [0]: between evaluations (Runge-Kutta bookkeeping, controller, reductions)
    const int q = lane >> 4, j = lane & 15;
    const int fo = 16 * tile + 4 * q;
    f32x4 act[1];
    // ---- S0
    if constexpr (P) {
        gates_into(n, ws, lane, tile, t);
        const f32x4 g = *reinterpret_cast<const f32x4*>(&ws.gate[0][0][fo]);
        const f32x4 c = *reinterpret_cast<const f32x4*>(&ws.gate[0][1][fo]);
        f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int m = 0; m < D; ++m) a += *reinterpret_cast<const f32x4*>(&n.w1[m][fo]) * z[m];
        f32x4 dg;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float x = act_f(fmaf(g[r], a[r], c[r]));
            act[0][r] = x;
            dg[r] = act_d(x) * g[r];
        }
        xc.dg[0][tile][lane] = dg;
        publish_b16<1>(xc.xv[0], act, tile, lane);
    } else {
        if (pend.on) fwd_tw_helper_tail<D, HUTCH, NSH>(n, wb, xc, lane, tile, s0);
    }
    pr.mark(1);        // [1]: S0 work
    __syncthreads();   // B1
    pr.mark(2);        // [2]: wait at B1
    // ---- S1
    if constexpr (P) {
        f32x4 acc[1];
        hidden_b16<1>(xc.xv[0], wb, 0, lane, acc);
        const f32x4 g = *reinterpret_cast<const f32x4*>(&ws.gate[1][0][fo]);
        const f32x4 c = *reinterpret_cast<const f32x4*>(&ws.gate[1][1][fo]);
        f32x4 dg;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float x = act_f(fmaf(g[r], acc[0][r], c[r]));
            act[0][r] = x;
            dg[r] = act_d(x) * g[r];
        }
        xc.dg[1][tile][lane] = dg;
        publish_b16<1>(xc.xv[1], act, tile, lane);
    } else {
        if (pend.on) div_prev = fwd_tw_helper_div<D, HUTCH>(xc, pend, e, j);
        // the last layer's gates at THIS evaluation's time, for its divergence two intervals from now (gates_into's arithmetic)
#pragma unroll
        for (int m = 0; m < D; ++m) pend.g4[m] = sigmoid_f(fmaf(n.vec4[2][m], t, n.vec4[3][m]));
        const f32x4 dg = xc.dg[0][tile][lane];
        f32x4 tg[NSH];
        if (HUTCH) {
            f32x4 au = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int m = 0; m < D; ++m) au += *reinterpret_cast<const f32x4*>(&n.w1[m][fo]) * e[m];
#pragma unroll
            for (int r = 0; r < 4; ++r) tg[0][r] = dg[r] * au[r];
        } else {
#pragma unroll
            for (int k = 0; k < NSH; ++k) {
                const f32x4 wcol = *reinterpret_cast<const f32x4*>(&n.w1[s0 + k][fo]);
#pragma unroll
                for (int r = 0; r < 4; ++r) tg[k][r] = dg[r] * wcol[r];
            }
        }
        publish_b16<NSH>(xc.tp[0] + s0, tg, tile, lane);
    }
    pr.mark(3);        // [3]: S1 work
    __syncthreads();   // B2
    pr.mark(4);        // [4]: wait at B2
    // ---- S2
    if constexpr (P) {
        f32x4 acc[1];
        hidden_b16<1>(xc.xv[1], wb, 1, lane, acc);
        const f32x4 g = *reinterpret_cast<const f32x4*>(&ws.gate[2][0][fo]);
        const f32x4 c = *reinterpret_cast<const f32x4*>(&ws.gate[2][1][fo]);
        f32x4 dg;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float x = act_f(fmaf(g[r], acc[0][r], c[r]));
            act[0][r] = x;
            dg[r] = act_d(x) * g[r];
        }
        xc.dg[2][tile][lane] = dg;
#pragma unroll
        for (int m = 0; m < D; ++m) {
            const f32x4 w4 = *reinterpret_cast<const f32x4*>(&n.w4[m][fo]);
            float pf = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) pf = fmaf(w4[r], act[0][r], pf);
            pf = xq_sum(pf);
            if (q == 0) xc.partf[tile][m][j] = pf;
        }
    } else {
        f32x4 U[NSH];
        hidden_b16<NSH>(xc.tp[0] + s0, wb, 0, lane, U);
        const f32x4 dg = xc.dg[1][tile][lane];
        f32x4 tg[NSH];
#pragma unroll
        for (int s = 0; s < NSH; ++s)
#pragma unroll
            for (int r = 0; r < 4; ++r) tg[s][r] = dg[r] * U[s][r];
        publish_b16<NSH>(xc.tp[1] + s0, tg, tile, lane);
        pend.on = true;
    }
    pr.mark(5);        // [5]: S2 work
    __syncthreads();   // B3
    pr.mark(6);        // [6]: wait at B3
    if constexpr (P) {
#pragma unroll
        for (int m = 0; m < D; ++m) {
            const float u4 = sum4(xc.partf[0][m][j], xc.partf[1][m][j], xc.partf[2][m][j], xc.partf[3][m][j]);
            f[m] = fmaf(ws.gate4[0][m], u4, ws.gate4[1][m]);
        }
    }
}

// before the controller looks at log p: the helpers finish the pending evaluation (one more barrier for everyone)
template <int D, bool HUTCH, int ROLE, int NSH>
__device__ __forceinline__ void fwd_tw_drain(const NetLdsSmall<D>& n, const SplitWeightsB16& wb, FwdTwLds<D, HUTCH>& xc, int lane, int tile, int s0,
                                             const float (&e)[D], FwdTwPending<D>& pend, float& div_prev) {
    if constexpr (ROLE == 2) fwd_tw_helper_tail<D, HUTCH, NSH>(n, wb, xc, lane, tile, s0);
    __syncthreads();
    if constexpr (ROLE == 2) {
        div_prev = fwd_tw_helper_div<D, HUTCH>(xc, pend, e, lane & 15);
        pend.on = false;
    }
}

// ---- the solve: controller of dopri5_split_kernel, state split over the roles (z: producers, log p: helpers) ----------------
// NH helper groups (kFwdTwGroupsNote).  1: waves 4..7 carry all tangent streams.  d (exact divergence, d = 2): TWELVE waves,
// three per SIMD -- waves 4..7 carry tangent stream 0 and own log p, waves 8..11 carry tangent stream 1 (and keep a second,
// identical copy of log p's state so that they take the controller's branches; they are never counted).  A helper group's
// barrier interval then holds one stream's matrix pass, like a producer's: the helpers' chain (two streams against the
// producers' one) was the critical path of the eight-wave arrangement.  Same arithmetic per stream, same sums.
template <int D, bool HUTCH, int NH>
__global__ __launch_bounds__((1 + NH) * kThreads) void dopri5_split_tw_kernel(SolveArgs a) {
    constexpr int NTHREADS = (1 + NH) * kThreads;
    constexpr int NSH = (HUTCH ? 1 : D) / NH;   // tangent streams per helper group
    static_assert(NH == 1 || (!HUTCH && NH == D), "one helper group, or one per tangent stream");
    const float res = a.residual ? 1.f : 0.f;   // CNF_DIV_RESIDUAL: dynamics f(z) - z
    __shared__ NetLdsSmall<D> net;              // no fp32 image of the two matrices: their pieces go from L2 straight to registers
    __shared__ SplitWaveLds<1> wl[kNW];         // gates per producer (their slopes live in xc.kz; kst is not used here)
    __shared__ FwdTwLds<D, HUTCH> xc;
    __shared__ RedLds red;
    __shared__ SplitWeightsShare wshare;        // the tile's weight operands, converted once per workgroup (cnf_split.h)
    const int tid = threadIdx.x, lane = tid & 63, j = lane & 15;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);   // a scalar: the roles branch on it
    const int tile = wv & (kNW - 1);
    if (tid == 0) red.status = 0;
    const long row = (long)blockIdx.x * kRows + j;
    const bool in_rows = row < a.rows && lane < 16;
    const long rr = row < a.rows ? row : a.rows - 1;
    float zin[D], ee[D];
#pragma unroll
    for (int m = 0; m < D; ++m) {
        zin[m] = a.x[rr * D + m];
        ee[m] = HUTCH ? a.e[rr * D + m] : 0.f;
    }
    const float lp_in = a.logp0 ? a.logp0[rr] : 0.f;
    const int after_status = a.after ? a.after->status : 0;
    // this wave's weight operands (both layers, three pieces) from the parameter vector, in flight together with the rows'
    // inputs and the staging of the small vectors (the four-wave kernel stages a 32 KB fp32 image in LDS first and splits from
    // there: one more round trip and a barrier on a launch's fixed cost).  The waves of a tile divide the conversion.
    SplitWeightsB16 wb;
#ifdef CNF_FWD_TW_OWN_WEIGHTS   // A/B timing: every wave converts all of its operands itself
    load_weights_b16_global<D>(wb, a.params, tile, lane);
#else
    convert_weights_b16_shared<D, 1 + NH>(wshare, a.params, tile, wv / kNW, lane);
#endif
    const int s0 = NH > 1 && wv >= kNW ? (wv / kNW - 1) * NSH : 0;   // helpers: first tangent stream of this wave's group
    stage_net_vectors<D>(net, a.params, tid, NTHREADS);
    for (int idx = tid; idx < 7 * D * kRows; idx += NTHREADS) (&xc.kz[0][0][0])[idx] = 0.f;   // zero_slopes: 0 x stale NaN
    __syncthreads();
#ifndef CNF_FWD_TW_OWN_WEIGHTS
    collect_weights_b16_shared(wb, wshare, tile, lane);
#endif

    GridSync gs{dbg_gran(a, __LINE__), (int)gridDim.x, 0u};
    const double nl = (double)a.rows, nz = nl * D;

    auto run = [&](auto role_c) __attribute__((always_inline)) {
    constexpr int ROLE = decltype(role_c)::value;
    constexpr bool P = ROLE == 1, H = ROLE == 2;
    // this role's components of the state: producers z (C = D), helpers log p (C = 1)
    constexpr int C = P ? D : 1;
    const bool counted = in_rows && wv == (P ? 0 : kNW);
    SplitWaveLds<1>& ws = wl[tile];
    float y0[C], y1[C], yi[C], sc[C], kh[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   // kh: the helpers' slopes of log p, in registers
    if constexpr (P) {
#pragma unroll
        for (int m = 0; m < D; ++m) y0[m] = yi[m] = zin[m];
    } else {
        y0[0] = yi[0] = lp_in;
    }
#pragma unroll
    for (int c = 0; c < C; ++c) y1[c] = y0[c];
    (void)kh;

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
    FwdTwPending<D> pend;
    Prof pr;
    int stage_prev = 0;    // helpers: the stage whose divergence arrives one evaluation late

    auto kslot = [&](int s, int c) -> float { return P ? xc.kz[s][c][j] : 0.f; };
    // slope s of this role's component c
    auto kget = [&](int s, int c) -> float {
        if constexpr (P) return xc.kz[s][c][j];
        else {
            float v = kh[0];
#pragma unroll
            for (int s_ = 1; s_ < 7; ++s_) v = s == s_ ? kh[s_] : v;
            return v;
        }
    };
    auto kput = [&](int s, float v) {   // helpers
#pragma unroll
        for (int s_ = 0; s_ < 7; ++s_) kh[s_] = s == s_ ? v : kh[s_];
    };
    (void)kslot;

    status = after_status;   // CnfSolverOpts.after: the solve whose output this one consumes failed
    for (; status == 0;) {
        float zz[D], f[D], k[C], div_prev = 0.f;
        bool restart = false;
        if constexpr (P) {
#pragma unroll
            for (int m = 0; m < D; ++m) zz[m] = yi[m];
        }
        fwd_eval_tw<D, HUTCH, ROLE, NSH>(net, wb, ws, xc, lane, tile, s0, sgn * ti, zz, ee, f, pend, div_prev, pr);
        ++nfe;
        // helpers: the divergence that has just arrived belongs to the PREVIOUS evaluation (stage_prev of this attempt)
        if constexpr (H) {
            if (phase == PH_STAGE && stage > 1) kput(stage_prev, -sgn * (div_prev - res * (float)D));
        }
        const bool ends = phase != PH_STAGE || stage == 6;    // the controller needs this evaluation's divergence: drain
        if (ends) {
            fwd_tw_drain<D, HUTCH, ROLE, NSH>(net, wb, xc, lane, tile, s0, ee, pend, div_prev);
            if constexpr (H) k[0] = -sgn * (div_prev - res * (float)D);
        }
        if constexpr (P) {
#pragma unroll
            for (int m = 0; m < D; ++m) k[m] = sgn * fmaf(-res, zz[m], f[m]);
        }

        if (phase == PH_F0) {
            double S[4] = {0, 0, 0, 0};
#pragma unroll
            for (int c = 0; c < C; ++c) {
                if constexpr (P) xc.kz[0][c][j] = k[c];
                else kput(0, k[c]);
                sc[c] = atol + fabsf(y0[c]) * rtol;
                const float u = y0[c] / sc[c], v = k[c] / sc[c];
                if (counted) {
                    S[P ? 0 : 1] += (double)u * u;
                    S[P ? 2 : 3] += (double)v * v;
                }
            }
            if (!tw_allreduce4(gs, red, S, wv, lane)) { status = CNF_ERR_SYNC_TIMEOUT; break; }
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
                const float u = (k[c] - kget(0, c)) / sc[c];
                if (counted) S[P ? 0 : 1] += (double)u * u;
            }
            if (!tw_allreduce4(gs, red, S, wv, lane)) { status = CNF_ERR_SYNC_TIMEOUT; break; }
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
            if constexpr (P) {
#pragma unroll
                for (int c = 0; c < C; ++c) xc.kz[stage][c][j] = k[c];
            }
            if (stage < 6) {
                stage_prev = stage;
                ++stage;
            } else {
                if constexpr (H) kput(6, k[0]);
                double S[4] = {0, 0, 0, 0};
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    if constexpr (P) {
                        y1[c] = yi[c];
                    } else {
                        // the helpers never form log p's intermediate stage states (nothing reads them): the candidate
                        // y1 is the sixth stage's state, by stage_state's arithmetic
                        stage_state(6, dtf, y0, [&](int s, int c_) { return kget(s, c_); }, y1);
                    }
                    float err = 0.f;
#pragma unroll
                    for (int s = 0; s < 7; ++s) err += (dtf * kCErr[s]) * kget(s, c);
                    const float tol = atol + rtol * fmaxf(fabsf(y0[c]), fabsf(y1[c]));
                    const float u = err / tol;
                    if (counted) {
                        S[P ? 0 : 1] += (double)u * u;
                        if (!isfinite(y0[c])) S[2] += 1.0;
                    }
                }
                if (!tw_allreduce4(gs, red, S, wv, lane)) { status = CNF_ERR_SYNC_TIMEOUT; break; }
                if (S[2] > 0.0) { status = CNF_ERR_NONFINITE; break; }
                const float rz = (float)(S[0] / nz), rl = (float)(S[1] / nl);
                const bool accept = (rz <= 1.f) && (rl <= 1.f);
                if (accept) {
                    ++nacc;
                    if (!(tau_end > tau + dt)) break;
#pragma unroll
                    for (int c = 0; c < C; ++c) {
                        y0[c] = y1[c];
                        if constexpr (P) xc.kz[0][c][j] = k[c];
                        else kput(0, k[c]);
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
            stage_prev = 0;
            phase = PH_STAGE;
        }
        if constexpr (P) {
            ti = t0f + stage_state(stage, dtf, y0, [&](int s, int c) { return xc.kz[s][c][j]; }, yi) * dtf;
        } else {
            ti = t0f + stage_node(stage) * dtf;
        }
    }

    const unsigned taken = grid_ticket(gs, tid);   // all waves are past the last reduction's closing barrier
    if (status == 0) {
        const float ta = (float)tau, tb = (float)(tau + dt), te = (float)tau_end;
        const float x = (te - ta) / (tb - ta);
        const float x2 = x * x, x3 = x2 * x, x4 = x3 * x;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            float mid = 0.f;
#pragma unroll
            for (int s = 0; s < 7; ++s) mid += (dtf * kCMid[s]) * kget(s, c);
            mid = y0[c] + mid;
            const float f0 = kget(0, c), f1 = kget(6, c), u = y0[c], v = y1[c];
            const float ca = (-2 * dtf) * f0 + (2 * dtf) * f1 + -8 * u + -8 * v + 16 * mid;
            const float cb = (5 * dtf) * f0 + (-3 * dtf) * f1 + 18 * u + 14 * v + -32 * mid;
            const float cc = (-4 * dtf) * f0 + dtf * f1 + -11 * u + -5 * v + 16 * mid;
            const float cd = dtf * f0;
            y1[c] = ca * x4 + cb * x3 + cc * x2 + cd * x + u;
        }
        if (counted) {
            if constexpr (P) {
#pragma unroll
                for (int m = 0; m < D; ++m) a.z_out[row * D + m] = y1[m];
            } else {
                a.logp_out[row] = y1[0];
            }
        }
    }
    // grid_cleanup for NTHREADS threads
    if (gs.nwg > 1) {
        if (tid == 0) red.last = taken == (unsigned)gs.nwg - 1u;
        __syncthreads();
        if (red.last) {
            for (int i = tid; i < 2 * gs.nwg * 4; i += NTHREADS) gs.gran[i] = 0ull;
            if (tid == 0) __hip_atomic_store(reinterpret_cast<unsigned*>(gs.gran + kGranWords), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    write_stats_global(a.stats, status, nfe, nacc, nrej, nfe, (float)dt, a.rows, tid);
#ifdef CNF_PROF_CYCLES
    if (blockIdx.x == 0 && lane == 0 && (wv % kNW) == 0 && nfe > 100)
        printf("fwd tw prof wave %d nfe %d cycles/eval: between %.0f | S0 %.0f wait B1 %.0f | S1 %.0f wait B2 %.0f | S2 %.0f wait B3 %.0f\n", wv, nfe,
               (double)pr.acc[0] / nfe, (double)pr.acc[1] / nfe, (double)pr.acc[2] / nfe, (double)pr.acc[3] / nfe, (double)pr.acc[4] / nfe,
               (double)pr.acc[5] / nfe, (double)pr.acc[6] / nfe);
#endif
    };   // run
    if (wv < kNW) run(std::integral_constant<int, 1>{});
    else run(std::integral_constant<int, 2>{});
}

// ---- the fixed-grid solves (euler / midpoint / rk4 = the 3/8 rule) on the same roles (kFixedSplitNote) ---------------------
// rk4_kernel (cnf_fwd.hip) is the throughput design: one wave per tile, 32 workgroups at the training batch size.  This is
// its latency design for batches of up to 16 * 256 rows: one tile per workgroup, the evaluation of dopri5_split_tw_kernel
// (fwd_eval_tw: value stream on the producers, tangent streams and log p on the helpers one barrier interval behind) inside
// rk4_kernel's loop -- same grid (points i * step_size + t0, the last clipped to t1), same stage states and the same update
// expressions, so z and log p differ from rk4_kernel's only by the hidden layers' arithmetic (three-piece bf16 products
// summed in fp32 against the fp32 MFMA's chain: ~1e-7 relative per evaluation).  No step control, hence no grid-wide
// reduction and no granules: workgroups are independent.  The helpers receive the divergence of stage s while stage s + 1
// is evaluated; the step's last stage is drained (one more barrier per step), because log p's update needs it.
template <int D, bool HUTCH, int NH>
__global__ __launch_bounds__((1 + NH) * kThreads) void fixed_split_tw_kernel(SolveArgs a) {
    constexpr int NTHREADS = (1 + NH) * kThreads;
    constexpr int NSH = (HUTCH ? 1 : D) / NH;   // tangent streams per helper group
    static_assert(NH == 1 || (!HUTCH && NH == D), "one helper group, or one per tangent stream");
    const float res = a.residual ? 1.f : 0.f;   // CNF_DIV_RESIDUAL: dynamics f(z) - z
    __shared__ NetLdsSmall<D> net;
    __shared__ SplitWaveLds<1> wl[kNW];
    __shared__ FwdTwLds<D, HUTCH> xc;
    __shared__ SplitWeightsShare wshare;
    const int tid = threadIdx.x, lane = tid & 63, j = lane & 15;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tile = wv & (kNW - 1);
    const long row = (long)blockIdx.x * kRows + j;
    const bool in_rows = row < a.rows && lane < 16;
    const long rr = row < a.rows ? row : a.rows - 1;
    float zin[D], ee[D];
#pragma unroll
    for (int m = 0; m < D; ++m) {
        zin[m] = a.x[rr * D + m];
        ee[m] = HUTCH ? a.e[rr * D + m] : 0.f;
    }
    const float lp_in = a.logp0 ? a.logp0[rr] : 0.f;
    const int status = a.after ? a.after->status : 0;   // CnfSolverOpts.after: the solve whose output this one consumes failed
    SplitWeightsB16 wb;
    convert_weights_b16_shared<D, 1 + NH>(wshare, a.params, tile, wv / kNW, lane);
    const int s0 = NH > 1 && wv >= kNW ? (wv / kNW - 1) * NSH : 0;
    stage_net_vectors<D>(net, a.params, tid, NTHREADS);
    __syncthreads();
    collect_weights_b16_shared(wb, wshare, tile, lane);

    const int nst = a.fixed_kind == CNF_SOLVER_EULER ? 1 : (a.fixed_kind == CNF_SOLVER_MIDPOINT ? 2 : 4);
    const int n_steps = status == 0 ? a.n_grid : 0;

    auto run = [&](auto role_c) __attribute__((always_inline)) {
    constexpr int ROLE = decltype(role_c)::value;
    constexpr bool P = ROLE == 1, H = ROLE == 2;
    constexpr int C = P ? D : 1;   // this role's components of the state: producers z, helpers log p
    const bool counted = in_rows && wv == (P ? 0 : kNW);
    SplitWaveLds<1>& ws = wl[tile];
    float y[C], yi[C], k1[C], k2[C], k3[C];
    if constexpr (P) {
#pragma unroll
        for (int m = 0; m < D; ++m) y[m] = zin[m];
    } else {
        y[0] = lp_in;
    }
#pragma unroll
    for (int c = 0; c < C; ++c) k1[c] = k2[c] = k3[c] = 0.f;
    const float sgn = a.t1 >= a.t0 ? 1.f : -1.f;
    const float tau0 = sgn * a.t0, tau1 = sgn * a.t1;
    FwdTwPending<D> pend;
    Prof pr;

    for (int step = 0; step < n_steps; ++step) {
        // grid point i = i*step_size + tau0 (product rounded before the add, like arange()*h + t0): rk4_kernel's arithmetic
        const float ta = a.n_grid == 1 ? tau0 : __fadd_rn(__fmul_rn((float)step, a.step_size), tau0);
        const float tb = (step + 1 == a.n_grid) ? tau1 : __fadd_rn(__fmul_rn((float)(step + 1), a.step_size), tau0);
        const float dt = tb - ta;
        float ti = ta;
#pragma unroll
        for (int c = 0; c < C; ++c) yi[c] = y[c];
        for (int st = 0; st < nst; ++st) {
            float zz[D], f[D], k[C], div_prev = 0.f;
#pragma unroll
            for (int m = 0; m < D; ++m) zz[m] = f[m] = 0.f;   // the helpers read neither
            if constexpr (P) {
#pragma unroll
                for (int m = 0; m < D; ++m) zz[m] = yi[m];
            }
            fwd_eval_tw<D, HUTCH, ROLE, NSH>(net, wb, ws, xc, lane, tile, s0, sgn * ti, zz, ee, f, pend, div_prev, pr);
            if constexpr (H) {
                // the divergence that has just arrived belongs to the PREVIOUS evaluation: stage st - 1 of this step
                const float kp = -sgn * (div_prev - res * (float)D);
                if (st == 1) k1[0] = kp;
                else if (st == 2) k2[0] = kp;
                else if (st == 3) k3[0] = kp;
            }
            const bool last = st + 1 == nst;   // log p's update needs this evaluation's divergence: drain (uniform)
            if (last) {
                fwd_tw_drain<D, HUTCH, ROLE, NSH>(net, wb, xc, lane, tile, s0, ee, pend, div_prev);
                if constexpr (H) k[0] = -sgn * (div_prev - res * (float)D);
            }
            if constexpr (P) {
#pragma unroll
                for (int m = 0; m < D; ++m) k[m] = sgn * fmaf(-res, zz[m], f[m]);
            }
            // rk4_kernel's updates; the helpers form no stage states of log p (nothing reads them), only the step's sum
            if (a.fixed_kind == CNF_SOLVER_EULER) {            // y + dt f(t, y)
#pragma unroll
                for (int c = 0; c < C; ++c) y[c] = y[c] + dt * k[c];
            } else if (a.fixed_kind == CNF_SOLVER_MIDPOINT) {  // y + dt f(t + dt/2, y + f(t, y) dt/2)
                if (st == 0) {
                    ti = ta + dt / 2;
                    if constexpr (P) {
#pragma unroll
                        for (int c = 0; c < C; ++c) yi[c] = y[c] + k[c] * dt / 2;
                    }
                } else {
#pragma unroll
                    for (int c = 0; c < C; ++c) y[c] = y[c] + dt * k[c];
                }
            } else if (st == 0) {
                ti = ta + dt / 3;
                if constexpr (P) {
#pragma unroll
                    for (int c = 0; c < C; ++c) { k1[c] = k[c]; yi[c] = y[c] + dt * k[c] / 3; }
                }
            } else if (st == 1) {
                ti = ta + dt * 2 / 3;
                if constexpr (P) {
#pragma unroll
                    for (int c = 0; c < C; ++c) { k2[c] = k[c]; yi[c] = y[c] + dt * (k1[c] / -3 + k[c]); }
                }
            } else if (st == 2) {
                ti = ta + dt;
                if constexpr (P) {
#pragma unroll
                    for (int c = 0; c < C; ++c) { k3[c] = k[c]; yi[c] = y[c] + dt * (k1[c] - k2[c] + k[c]); }
                }
            } else {
#pragma unroll
                for (int c = 0; c < C; ++c) y[c] = y[c] + (k1[c] + 3 * k2[c] + 3 * k3[c] + k[c]) * (dt / 8);
            }
        }
    }
    if (status == 0 && counted) {
        if constexpr (P) {
#pragma unroll
            for (int m = 0; m < D; ++m) a.z_out[row * D + m] = y[m];
        } else {
            a.logp_out[row] = y[0];
        }
    }
    };   // run
    if (wv < kNW) run(std::integral_constant<int, 1>{});
    else run(std::integral_constant<int, 2>{});
    // counters as rk4_kernel reports them (the API layer zeroes the block before a fixed-grid launch)
    const int nrun = nst * n_steps;
    write_stats_global(a.stats, status, nrun, n_steps, 0, nrun, 0.f, a.rows, tid);
}

}  // namespace cnf
