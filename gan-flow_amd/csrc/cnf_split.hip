// cnf_split.hip -- latency-oriented variants of the batch-global dopri5 solves: ONE 16-row tile per WORKGROUP.
//
// The one-wave-per-tile kernels (cnf_fwd.hip / cnf_bwd.hip) are the throughput design: no barriers, every wave
// does all 64 output features of every layer.  At the training batch size (2048 rows = 128 tiles) that leaves
// 7/8 of the chip idle and one evaluation costs ~10 us (forward) / ~33 us (augmented).  Here the four waves of a
// workgroup split the OUTPUT FEATURES of every layer: wave w computes feature tile w (features 16w..16w+15) for
// all streams, i.e. a quarter of the MFMAs, of the tanh/gate arithmetic and of the saved state.  A layer's input
// needs all 64 features, so after each layer every wave publishes its tile to LDS in the C/D register layout
// (one ds_write_b128 per stream) and reads the four tiles back as B operands (ds_read_b128) behind one
// workgroup barrier: 3 barriers per forward evaluation, 6 per augmented evaluation.  2048 rows now occupy 128
// CUs and the dependent chain per evaluation is ~4x shorter.  Every wave keeps its own copy of the 16 rows'
// Runge-Kutta state (identical arithmetic, bit-identical results), so the controller logic is unchanged.
// Used when rows <= 16 * 256 (every workgroup resident for the grid-wide reductions); larger batch-global solves
// fall back to the one-wave-per-tile kernels.
#include <type_traits>

#include "cnf_split.h"
#include "cnf_adam.h"

// This text is compiled as TWO translation units: part 1 = the forward solve (cnf_split_fwd.hip defines CNF_SPLIT_PART 1 and
// includes this file; built with the max-ILP instruction scheduler, which shortens a forward evaluation by 4 % and
// lengthens an adjoint one by 1 %: Makefile), part 2 = everything else (this file compiled as it is).
// The kernels are templates: each unit instantiates only what its launchers name.
#ifndef CNF_SPLIT_PART
#define CNF_SPLIT_PART 2
#endif

namespace cnf {

// streams a tile carries: value + tangents.  d <= 2: up to 2 exact tangents; d = 16 is Hutchinson only (one tangent)
template <int D>
constexpr int kNSMaxOf = D > 2 ? 2 : 3;
// d = 16 adjoint: the four waves share ONE copy of the gates and of the Runge-Kutta slopes (they hold identical
// values and every reader is separated from the next evaluation's writers by a workgroup barrier); 4 copies of
// kst[7][33][16] would not fit next to the weight images in 160 KB.
template <int D>
constexpr bool kShareWaveLds = D > 2;

template <int C>
struct alignas(16) SplitWaveLds {
    float gate[3][2][kH];
    float gate4[2][16];
    float kst[7][C][kRows];
};

template <int D>
struct alignas(16) SplitXchg {
    float xb[2][kNSMaxOf<D>][4][64][4];   // activations / cotangents in C-D layout: [buffer][stream][feature tile][lane][r]
    float part[kNW][2 * D][kRows];   // layer-4 partial sums of every wave
};

// Forward solve: the hidden layers run on the bf16 matrix pipe with three-piece operands (cnf_device.h, field_fwd_b16).
// A wave splits its OWN 4 values per stream before publishing them: per (stream, piece) it writes 4 bf16 = 8 bytes, laid
// out so that the 16 bytes a lane reads back for k-step ks are the pieces of tiles 2ks and 2ks+1 -- the B operand of
// v_mfma_f32_16x16x32_bf16 as it is.
template <int D>
struct alignas(16) SplitXchgB16 {
    uint32_t xb16[2][kNSMaxOf<D>][3][2][64][4];   // [buffer][stream][piece][k-step][lane][tile & 1][2]
    float part[kNW][2 * D][kRows];               // layer-4 partial sums of every wave
};
// The adjoint at d <= 2 runs its four hidden-layer products per evaluation (two forward, two transposed) on the bf16 pipe
// as well (kBwdB16): forward weight pieces in 48 registers per lane (read from the parameter vector), transposed pieces
// in a 48 KB LDS image, cotangents published as pieces.  The two 64 x 64 fp32 images (64 KB) are gone from its LDS.
template <int D>
constexpr bool kBwdB16 = D <= 2;

template <int D>
struct alignas(16) NetLdsSmall {   // NetLds without the matrix images
    float w1[D][kH];
    float w4[D][kH];
    float vec[3][4][kH];
    float vec4[4][16];
};
template <int D>
using BwdNet = typename std::conditional<kBwdB16<D>, NetLdsSmall<D>, NetLds<D>>::type;
// the adjoint keeps the forward sweep's published activations (xa[l] = input of hidden layer l) until the end of the
// evaluation -- they are the B operands of that layer's outer products -- so the cotangents get buffers of their own
template <int D>
struct alignas(16) SplitXchgBwdB16 {
    uint32_t xa[2][kNSMaxOf<D>][3][2][64][4];     // [layer][stream][piece][k-step][lane][tile & 1][2]
    uint32_t xcot[2][kNSMaxOf<D>][3][2][64][4];   // pre-activation cotangents of hidden layer lc
    float part[kNW][2 * D][kRows];               // layer-4 partial sums of every wave
};
template <int D>
using BwdXchg = typename std::conditional<kBwdB16<D>, SplitXchgBwdB16<D>, SplitXchg<D>>::type;

// The gates of ONE evaluation, own share only: a wave reads the gates of its own 16 features per layer (and the last
// layer's D), so it computes just those -- lanes 0..47: layer lane >> 4, feature 16 wave + (lane & 15); lanes 48..63: the
// last layer's entry lane - 48 -- in one pass instead of four (same arithmetic per entry: identical bits).  The LDS
// layout is unchanged; entries of other waves' features are never read by this wave (with one shared copy, D = 16, the
// four waves fill it together).
template <class WS, class N>
__device__ __forceinline__ void gates_into(const N& n, WS& ws, int lane, int wave, float t) {
    const bool last = lane >= 48;
    const int l = lane >> 4, i = 16 * wave + (lane & 15), m = lane & 15;
    const float* vw = last ? &n.vec4[2][m] : &n.vec[l][2][i];   // gate weight; bias, layer bias, hyper-bias follow
    const float* vg = last ? &n.vec4[3][m] : &n.vec[l][3][i];
    const float* vb = last ? &n.vec4[0][m] : &n.vec[l][0][i];
    const float* vh = last ? &n.vec4[1][m] : &n.vec[l][1][i];
    const float g = sigmoid_f(fmaf(*vw, t, *vg));
    const float c = fmaf(g, *vb, *vh * t);
    float* og = last ? &ws.gate4[0][m] : &ws.gate[l][0][i];
    float* oc = last ? &ws.gate4[1][m] : &ws.gate[l][1][i];
    *og = g;
    *oc = c;
    wave_lds_fence();
}

__device__ __forceinline__ float sum4(float a, float b, float c, float d) { return ((a + b) + c) + d; }

// Phase clock for timing experiments (tools/build_variants.sh prof "-DCNF_PROF_CYCLES"): cycles between marks, summed
// over all evaluations of workgroup 0 / wave 0, printed when the kernel ends.  Compiles to nothing otherwise.
#ifdef CNF_PROF_CYCLES
struct Prof {
    unsigned long long last = 0, acc[32] = {};
    __device__ __forceinline__ void mark(int i) {
        const unsigned long long now = clock64();
        if (last) acc[i] += now - last;
        last = now;
    }
};
#else
struct Prof {
    __device__ __forceinline__ void mark(int) {}
};
#endif

// ---- forward dynamics, feature-split over the 4 waves ---------------------------------------------------
template <int D, bool HUTCH, class WS>
__device__ __forceinline__ void field_fwd_split(const NetLds<D>& n, const SplitWeightsB16& wb, WS& ws, SplitXchgB16<D>& xc,
                                                int lane, int wave, float t, const float (&z)[D], const float (&e)[D],
                                                float (&f)[D], float& div, Prof& pr) {
    constexpr int NT = HUTCH ? 1 : D;
    constexpr int NS = NT + 1;
    const int q = lane >> 4, j = lane & 15;
    const int fo = 16 * wave + 4 * q;
    gates_into(n, ws, lane, wave, t);
    pr.mark(1);
    f32x4 act[NS];
    {
        const f32x4 g = *reinterpret_cast<const f32x4*>(&ws.gate[0][0][fo]);
        const f32x4 c = *reinterpret_cast<const f32x4*>(&ws.gate[0][1][fo]);
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
            act[0][r] = x;
            if (HUTCH) {
                act[1][r] = dg * au[r];
            } else {
#pragma unroll
                for (int m = 0; m < NT; ++m) act[1 + m][r] = dg * wcol[m][r];
            }
        }
    }
    pr.mark(2);
#pragma unroll
    for (int l = 0; l < 2; ++l) {
        publish_b16<NS>(xc.xb16[l], act, wave, lane);
        __syncthreads();
        pr.mark(3 + 3 * l);
        f32x4 acc[NS];
        hidden_b16<NS>(xc.xb16[l], wb, l, lane, acc);
        pr.mark(4 + 3 * l);
        const f32x4 g = *reinterpret_cast<const f32x4*>(&ws.gate[l + 1][0][fo]);
        const f32x4 c = *reinterpret_cast<const f32x4*>(&ws.gate[l + 1][1][fo]);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float x = act_f(fmaf(g[r], acc[0][r], c[r]));
            const float dg = act_d(x) * g[r];
            act[0][r] = x;
#pragma unroll
            for (int s = 1; s < NS; ++s) act[s][r] = dg * acc[s][r];
        }
        pr.mark(5 + 3 * l);
    }
    // layer 4: this wave's 16 features, then across waves through LDS
#pragma unroll
    for (int m = 0; m < D; ++m) {
        const f32x4 w4 = *reinterpret_cast<const f32x4*>(&n.w4[m][fo]);
        float pf = 0.f, pd = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            pf = fmaf(w4[r], act[0][r], pf);
            pd = fmaf(w4[r], act[HUTCH ? 1 : 1 + m][r], pd);
        }
        pf = xq_sum(pf);
        pd = xq_sum(pd);
        if (q == 0) {
            xc.part[wave][m][j] = pf;
            xc.part[wave][D + m][j] = pd;
        }
    }
    pr.mark(9);
    __syncthreads();
    pr.mark(10);
    div = 0.f;
#pragma unroll
    for (int m = 0; m < D; ++m) {
        const float g4 = ws.gate4[0][m];
        const float u4 = sum4(xc.part[0][m][j], xc.part[1][m][j], xc.part[2][m][j], xc.part[3][m][j]);
        const float U4 = sum4(xc.part[0][D + m][j], xc.part[1][D + m][j], xc.part[2][D + m][j], xc.part[3][D + m][j]);
        f[m] = fmaf(g4, u4, ws.gate4[1][m]);
        const float dm = g4 * U4;
        div += HUTCH ? dm * e[m] : dm;
    }
    pr.mark(11);
}

// ---- forward solve -----------------------------------------------------------------------------------------
template <int D, bool HUTCH>
__global__ __launch_bounds__(kThreads) void dopri5_split_kernel(SolveArgs a) {
    const float res = a.residual ? 1.f : 0.f;   // CNF_DIV_RESIDUAL: dynamics f(z) - z
    constexpr int C = D + 1;
    __shared__ NetLds<D> net;
    __shared__ SplitWaveLds<C> wl[kNW];
    __shared__ SplitXchgB16<D> xc;
    __shared__ RedLds red;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, j = lane & 15;
    if (tid == 0) red.status = 0;
    const long row = (long)blockIdx.x * kRows + j;
    const bool counted = row < a.rows && lane < 16 && w == 0;   // every wave carries the same 16 rows: count them once
    const long rr = row < a.rows ? row : a.rows - 1;
    // the rows' inputs and the predecessor's status are requested BEFORE the staging: their round trips to memory overlap it
    // (a launch's fixed cost is a chain of such round trips: ~1.5 us each at NFE 14, where a solve is only ~50 us long)
    float y0[C], y1[C], yi[C], ee[D];
#pragma unroll
    for (int m = 0; m < D; ++m) {
        y0[m] = a.x[rr * D + m];
        ee[m] = HUTCH ? a.e[rr * D + m] : 0.f;
    }
    y0[D] = a.logp0 ? a.logp0[rr] : 0.f;
    const int after_status = a.after ? a.after->status : 0;
    stage_net(net, a.params, tid);
    zero_slopes(wl, kNW, tid);
    __syncthreads();
    SplitWeightsB16 wb;
    wb.load(net, w, lane);
    SplitWaveLds<C>& ws = wl[w];

    GridSync gs{dbg_gran(a, __LINE__), (int)gridDim.x, 0u};
    const double nl = (double)a.rows, nz = nl * D;

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

    Prof pr;
    pr.mark(15);
    status = after_status;   // CnfSolverOpts.after: the solve whose output this one consumes failed
    for (; status == 0;) {
        float zz[D], f[D], div, k[C];
        bool restart = false;
#pragma unroll
        for (int m = 0; m < D; ++m) zz[m] = yi[m];
        pr.mark(0);
        field_fwd_split<D, HUTCH>(net, wb, ws, xc, lane, w, sgn * ti, zz, ee, f, div, pr);
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
            if (!wg_allreduce4(gs, red, S, w, lane)) { status = CNF_ERR_SYNC_TIMEOUT; break; }
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
            if (!wg_allreduce4(gs, red, S, w, lane)) { status = CNF_ERR_SYNC_TIMEOUT; break; }
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
                pr.mark(12);
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
                pr.mark(13);
                if (!wg_allreduce4(gs, red, S, w, lane)) { status = CNF_ERR_SYNC_TIMEOUT; break; }
                pr.mark(14);
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
            pr.mark(16);
        }
        ti = t0f + stage_state(stage, dtf, y0, [&](int s, int c) { return ws.kst[s][c][j]; }, yi) * dtf;
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
            for (int s = 0; s < 7; ++s) mid += (dtf * kCMid[s]) * ws.kst[s][c][j];
            mid = y0[c] + mid;
            const float f0 = ws.kst[0][c][j], f1 = ws.kst[6][c][j], u = y0[c], v = y1[c];
            const float ca = (-2 * dtf) * f0 + (2 * dtf) * f1 + -8 * u + -8 * v + 16 * mid;
            const float cb = (5 * dtf) * f0 + (-3 * dtf) * f1 + 18 * u + 14 * v + -32 * mid;
            const float cc = (-4 * dtf) * f0 + dtf * f1 + -11 * u + -5 * v + 16 * mid;
            const float cd = dtf * f0;
            y1[c] = ca * x4 + cb * x3 + cc * x2 + cd * x + u;
        }
        if (counted) {
#pragma unroll
            for (int m = 0; m < D; ++m) a.z_out[row * D + m] = y1[m];
            a.logp_out[row] = y1[D];
        }
    }
    grid_cleanup(gs, red, tid, taken);
    write_stats_global(a.stats, status, nfe, nacc, nrej, nfe, (float)dt, a.rows, tid);
#ifdef CNF_PROF_SYNC
    if ((blockIdx.x == 0 || blockIdx.x == 77) && tid == 0 && nfe > 100)
        printf("fwd sync wg %d: reductions %u, clocks/reduction: entry -> publish %.0f, publish -> all seen %.0f (end of first sweep "
               "%.0f), all seen -> exit %.0f; sweeps/reduction %.2f, own granule seen in first sweep %.2f\n", (int)blockIdx.x,
               gs.n_red, (double)gs.t_pre / gs.n_red, (double)gs.t_total / gs.n_red, (double)gs.t_first / gs.n_red,
               (double)gs.t_post / gs.n_red, (double)gs.n_sweeps / gs.n_red, (double)gs.n_self_first / gs.n_red);
#endif
#ifdef CNF_PROF_CYCLES
    if (blockIdx.x == 0 && tid == 0 && nfe > 100) {
        printf("fwd prof nfe %d cycles/eval:", nfe);
        for (int i = 0; i < 17; ++i) printf(" [%d] %.0f", i, (double)pr.acc[i] / nfe);
        printf("\n");
    }
#endif
}

// ============================================================================================================
// adjoint, feature-split
// ============================================================================================================
template <int D>
struct SmallIdxS {   // parameters other than the two 64x64 matrices, canonical order with those matrices cut out
    static constexpr int n0 = kH * D + 4 * kH;
    static constexpr int v2 = n0;
    static constexpr int v3 = n0 + 4 * kH;
    static constexpr int l4 = n0 + 8 * kH;
    static constexpr int count = l4 + kH * D + 4 * D;
    static constexpr int padded = (count + 3) & ~3;
    __host__ __device__ static int vec(int layer) { return layer == 0 ? kH * D : (layer == 1 ? v2 : v3); }
    __host__ __device__ static int canonical(int si) { return si < n0 ? si : (si < v3 ? si + kH * kH : si + 2 * kH * kH); }
};

// The seventh stage of a Dormand-Prince step is the first stage of the next one (FSAL).  Its state slopes are reused as
// they are; its d/dtheta sum G enters the quadrature with a weight that is only known once the next step size is, so
// it is accumulated UNWEIGHTED into a buffer of its own ("last"), promoted to "first" when the step is accepted and
// added to the step's sums with weight wq[0] at commit time.  This removes the re-evaluation of the first stage
// (7 -> 6 evaluations per attempt).  Needs two more copies of the small-parameter sums in LDS: d <= 2 only.
template <int D>
constexpr bool kFsalTheta = D <= 2;

template <int D, bool B16>
struct BwdWimgT {
    float wimgT[2][16][64][4];            // as BwdNetLds::wimgT
};
template <int D>
struct BwdWimgT<D, true> {
    WimgT16 wimgT16;                      // stage_wimgT16 (cnf_split.h)
};
template <int D, bool B16>
struct BwdTransposes {
    float trB[2][kNSMaxOf<D>][kH * kRows];     // transposed layer inputs, all 64 features: [buffer][stream][feature*16 + pos(row)]
    float trA[kNW][kNSMaxOf<D>][16 * kRows];   // per wave: transposed, weighted cotangents of its own 16 features
};
template <int D>
struct BwdTransposes<D, true> {};              // kBwdB16: operands are read transposed from the piece buffers (outer_b16)
template <int D, bool TW>
struct BwdSmallLds {
    float small[kFsalTheta<D> ? 4 : 1][SmallIdxS<D>::padded];   // d/dtheta of the small parameters (waves own disjoint
                                          // features): [0] the step's weighted sum, [1], [2] the FSAL buffers, [3] the
                                          // running total over accepted steps (two-workgroups-per-tile launches)
};
// Eight-wave workgroups (kThetaWavesNote): the small-parameter sums live in the theta waves' REGISTERS (after reduce8 a lane
// owns one entry per reduction: SmallRegs); the producers hand over what they used to reduce themselves -- eight floats
// per lane and reduction -- through these buffers.  Which reduction uses which buffer, and why a reader is always
// separated from the next writer by a workgroup barrier: kSmallBufNote in aug_eval_split.
template <int D>
struct BwdSmallLds<D, true> {
    f32x4 sbuf[3][kNW][2][64];            // [buffer][producer wave][a | b][lane]
    float l4pub[2][D][kRows];             // last layer's vectors: per-row operands of the two row sums (producer wave 0)
};
template <int D, bool TW = false>
struct alignas(16) SplitBwdLds : BwdWimgT<D, kBwdB16<D>>, BwdTransposes<D, kBwdB16<D>>, BwdSmallLds<D, TW> {
    float part2[kNW][D][kRows];           // vjp_z partial sums of every wave
    float wq[8];                          // quadrature weights of the current attempt
};
// a theta wave's share of the small-parameter sums, one lane = one entry each (lane (q, j) of the wave for feature tile
// w: feature fo + qq = 16 w + 4 q + (j & 3), kind = j >> 2 of bias | hyper-bias | gate weight | gate bias):
//   e[0]  hidden layer vectors: layer 3 (part 1) or layer 2 (part 2), entry (kind, feature)
//   e[1]  layer 1 vectors (part 2), entry (kind, feature)
//   e[2]  W4[kind][feature] (part 1) or W1[feature][kind] (part 2), lanes with kind < D
//   e[3 + m]  last layer's vector qq, output m (part 1, tile 0, lanes 0..3)
template <int D>
struct ThetaAccT : BigAccT<1> {
    float e[3 + D];
    __device__ __forceinline__ void zero() {
        BigAccT<1>::zero();
#pragma unroll
        for (int i = 0; i < 3 + D; ++i) e[i] = 0.f;
    }
};

// two-workgroups-per-tile adjoint launches (d <= 2) run eight waves per workgroup, kThetaWavesNote; the switch exists
// for A/B timing (tools/build_variants.sh nothetawaves "-DCNF_NO_THETA_WAVES")
#ifdef CNF_NO_THETA_WAVES
constexpr bool kThetaWaves = false;
#else
constexpr bool kThetaWaves = true;
#endif

// kRematLaneNote (dopri5_bwd_split_kernel): which copies of the lane index pass through an opaque statement once per
// loop iteration -- 1: the producers' evaluation, 2: the theta waves' evaluation, 4: the grid reductions (both roles)
#ifndef CNF_REMAT_LANE
#define CNF_REMAT_LANE 6
#endif
constexpr int kRematLane = CNF_REMAT_LANE;

// timing experiments only (tools/build_variants.sh): drop parts of the d/dtheta work to see what they cost
#ifdef CNF_EXP_NO_SMALL
constexpr bool kExpSmall = false;
#else
constexpr bool kExpSmall = true;
#endif
#ifdef CNF_EXP_NO_DW
constexpr bool kExpDw = false;
#else
constexpr bool kExpDw = true;
#endif

// the adjoint kernels' weights: fp32 images of W and W^T in LDS, or (kBwdB16) register pieces of W + an LDS image of W^T pieces
template <int D, int NTH = kThreads, class SB>   // NTH: threads of the workgroup (the first kNW waves hold the register pieces)
__device__ __forceinline__ void stage_bwd_net(BwdNet<D>& net, SB& sb, SplitWeightsB16& wb, const float* __restrict__ p, int tid) {
    if constexpr (kBwdB16<D>) {
        stage_net_vectors<D>(net, p, tid, NTH);
        stage_wimgT16<D, NTH>(sb.wimgT16, p, tid);
        if (NTH == kThreads || tid < kThreads) load_weights_b16_global<D>(wb, p, tid >> 6, tid & 63);
    } else {
        static_assert(kBwdB16<D> || NTH == kThreads, "eight-wave workgroups: bf16 pipe only");
        stage_net(net, p, tid);
        stage_wimgT<D>(&sb.wimgT[0][0][0][0], p, tid);
    }
}

// one augmented evaluation; same mathematics as aug_eval (cnf_bwd.hip), each wave on its own feature tile
// acc / alt: the accumulators of the hidden matrices' outer products; to_alt (uniform) sends this evaluation's to `alt`
// (the FSAL buffer), so that no register block has to be copied around the call.
// mid(f, div): called by all threads once the forward half is done (f and div are final, the reverse sweep has not begun):
// the solve kernel starts its grid-wide reduction there, so that the granules travel while the sweep runs.
struct NoMid {
    template <class F>
    __device__ __forceinline__ void operator()(const F&, float) const {}
};
// ROLE (theta waves, kThetaWavesNote): 0 = this wave does everything (four-wave workgroups); 1 = state work and the
// small-parameter sums, no outer products (a producer wave of an eight-wave workgroup); 2 = ONLY the outer products of
// feature tile `wave`, read from the piece buffers the producers publish (a theta wave: it takes part in every barrier
// of the evaluation and in mid(), and touches none of z, e, az, f, div, vz).
//
// kSmallBufNote -- ROLE 1 / 2: the small-parameter sums.  Where a four-wave workgroup runs reduce8 and adds the result to
// its LDS sums, a producer stores the reduction's eight per-lane inputs (sb.sbuf) and the theta wave of the same feature
// tile reduces them and adds the result -- same factors, same order of additions -- to its registers (ThetaAccT::e).
//   part 1 (upper half):  W4 -> buffer 0 and layer 3 -> buffer 1, written before the sweep's first barrier, read behind it;
//                         the last layer's vectors -> l4pub, written behind the first barrier, read behind the second.
//   part 2 (lower half):  layer 2 -> buffer 0, written before the sweep's second barrier, read behind it; layer 1 ->
//                         buffer 1 and W1 -> buffer 2, written before the evaluation's last barrier, read behind it.
// A buffer's next writer is the same point of the NEXT evaluation, at least four barriers after its reader.  The gates
// the factors need are read before the evaluation's last barrier (the producers overwrite them right after it).
template <int D, bool HUTCH, int ROLE = 0, class WS, class SB, class Acc, class Mid>
__device__ __forceinline__ void aug_eval_split(const BwdNet<D>& n, const SplitWeightsB16& wb, SB& sb, WS& ws,
                                               BwdXchg<D>& xc, int lane,
                                               int wave, float t, const float (&z)[D], const float (&e)[D],
                                               const float (&az)[D], float al, float w, bool with_theta, float (&f)[D],
                                               float& div, float (&vz)[D], Acc& acc, Acc& alt, bool to_alt,
                                               int small_base, int part, Prof& pr, Mid&& mid) {
    constexpr int NL = Acc::kNL;
    constexpr bool B16 = kBwdB16<D>;
    constexpr bool P = ROLE != 2, T = ROLE != 1;   // this wave does the state work / the outer products
    static_assert(ROLE == 0 || D <= 2, "one reduce8 per weight block");
    static_assert(ROLE == 0 || B16, "separate theta waves read the bf16 piece buffers");
    constexpr int NT = HUTCH ? 1 : D;
    constexpr int NS = NT + 1;
    using SI = SmallIdxS<D>;
    // part (uniform over the workgroup): which d/dtheta sums this workgroup keeps.  0 = all; 1 = the upper half
    // (W3, layer-3 vectors, W4, layer-4 vectors); 2 = the lower half (W1, layer-1/2 vectors, W2).  See kSpreadNote.
    const bool th_hi = with_theta && part != 2, th_lo = with_theta && part != 1;
    float* small = nullptr;
    if constexpr (ROLE == 0) small = &sb.small[0][0] + small_base;   // an LDS address the compiler can see (no flat atomics)
    const int q = lane >> 4, j = lane & 15, qq = j & 3;
    const int fo = 16 * wave + 4 * q;
    const int pos = (j & 3) * 4 + (j >> 2);
    if constexpr (P) gates_into(n, ws, lane, wave, t);
    pr.mark(1);
    // after reduce8 lane j owns index r* = j & 3 and parameter kind j >> 2 (bias, hyper-bias, gate weight, gate bias):
    // factor = k0 * g + k1 * t + gg * (k2 * t + k3) as arithmetic on lane masks (nested selects compile to branches)
    const bool b0 = (j & 1) != 0, b1 = (j & 2) != 0;
    const int kind = j >> 2;
    const float km0 = kind == 0 ? 1.f : 0.f, km1t = kind == 1 ? t : 0.f, km23 = kind == 2 ? t : (kind == 3 ? 1.f : 0.f);
    // ROLE 1: hand a reduction's inputs to the theta wave; ROLE 2: take them, reduce (kSmallBufNote)
    auto put8 = [&](int buf, const float (&va)[4], const float (&vb)[4]) {
        if constexpr (ROLE == 1) {
            sb.sbuf[buf][wave][0][lane] = f32x4{va[0], va[1], va[2], va[3]};
            sb.sbuf[buf][wave][1][lane] = f32x4{vb[0], vb[1], vb[2], vb[3]};
        }
    };
    auto get8 = [&](int buf, float& ta, float& tb) {
        if constexpr (ROLE == 2) {
            const f32x4 A = sb.sbuf[buf][wave][0][lane], B = sb.sbuf[buf][wave][1][lane];
            const float va[4] = {A[0], A[1], A[2], A[3]}, vb[4] = {B[0], B[1], B[2], B[3]};
            reduce8(va, vb, b0, b1, ta, tb);
        }
    };
    // the hidden layers' vector sums from the reduced (st, gt) and the lane's gate value gs
    auto vec_term = [&](float st, float gt, float gs) {
        const float gg = gs * (1.0f - gs);
        const float fac = fmaf(gg, km23, fmaf(km0, gs, km1t));
        return w * fac * (kind < 2 ? st : gt);
    };
    constexpr int kBufL0 = 1, kBufW1 = 2;
    float gs_l1 = 0.f;   // ROLE 2, part 2: the lane's layer-1 gate, read before the evaluation's last barrier

    // ------------------------------------------------------------------ forward with saves (own tile)
    f32x4 xs[3], ub[3], U[2][NT], act[NS];
    f32x4 w1c[D];
    if constexpr (P) {
#pragma unroll
    for (int m = 0; m < D; ++m) w1c[m] = *reinterpret_cast<const f32x4*>(&n.w1[m][fo]);
    {
        const f32x4 g = *reinterpret_cast<const f32x4*>(&ws.gate[0][0][fo]);
        const f32x4 c = *reinterpret_cast<const f32x4*>(&ws.gate[0][1][fo]);
        const f32x4 b = *reinterpret_cast<const f32x4*>(&n.vec[0][0][fo]);
        f32x4 a = {0.f, 0.f, 0.f, 0.f}, au = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int m = 0; m < D; ++m) {
            a += w1c[m] * z[m];
            if (HUTCH) au += w1c[m] * e[m];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float x = act_f(fmaf(g[r], a[r], c[r]));
            const float dg = act_d(x) * g[r];
            xs[0][r] = x;
            ub[0][r] = a[r] + b[r];
            act[0][r] = x;
            if (HUTCH) {
                act[1][r] = dg * au[r];
            } else {
#pragma unroll
                for (int m = 0; m < NT; ++m) act[1 + m][r] = dg * w1c[m][r];
            }
        }
    }
    }
    pr.mark(2);
#pragma unroll
    for (int l = 0; l < 2; ++l) {
        f32x4 a4[NS];
        if constexpr (B16) {
            if constexpr (P) publish_b16<NS>(xc.xa[l], act, wave, lane);
            __syncthreads();
            pr.mark(3 + 3 * l);
            if constexpr (P) hidden_b16<NS>(xc.xa[l], wb, l, lane, a4);
            pr.mark(4 + 3 * l);
        } else {
#pragma unroll
            for (int s = 0; s < NS; ++s) *reinterpret_cast<f32x4*>(&xc.xb[l][s][wave][lane][0]) = act[s];
            __syncthreads();
#pragma unroll
            for (int s = 0; s < NS; ++s) a4[s] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int Ti = 0; Ti < 4; ++Ti) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(&n.wimg[l][wave * 4 + Ti][lane][0]);
                f32x4 in[NS];
#pragma unroll
                for (int s = 0; s < NS; ++s) in[s] = *reinterpret_cast<const f32x4*>(&xc.xb[l][s][Ti][lane][0]);
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int s = 0; s < NS; ++s) a4[s] = mfma16(a[r], in[s][r], a4[s]);
            }
        }
        if constexpr (P) {
        const f32x4 g = *reinterpret_cast<const f32x4*>(&ws.gate[l + 1][0][fo]);
        const f32x4 c = *reinterpret_cast<const f32x4*>(&ws.gate[l + 1][1][fo]);
        const f32x4 b = *reinterpret_cast<const f32x4*>(&n.vec[l + 1][0][fo]);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float x = act_f(fmaf(g[r], a4[0][r], c[r]));
            const float dg = act_d(x) * g[r];
            xs[l + 1][r] = x;
            ub[l + 1][r] = a4[0][r] + b[r];
            act[0][r] = x;
#pragma unroll
            for (int s = 1; s < NS; ++s) {
                U[l][s - 1][r] = a4[s][r];
                act[s][r] = dg * a4[s][r];
            }
        }
        }
        pr.mark(5 + 3 * l);
    }
    f32x4 w4[D];
    if constexpr (P) {
#pragma unroll
    for (int m = 0; m < D; ++m) {
        w4[m] = *reinterpret_cast<const f32x4*>(&n.w4[m][fo]);
        float pf = 0.f, pd = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            pf = fmaf(w4[m][r], act[0][r], pf);
            pd = fmaf(w4[m][r], act[HUTCH ? 1 : 1 + m][r], pd);
        }
        pf = xq_sum(pf);
        pd = xq_sum(pd);
        if (q == 0) {
            xc.part[wave][m][j] = pf;
            xc.part[wave][D + m][j] = pd;
        }
    }
    }
    pr.mark(9);
    // The last layer's partial sums meet behind the SAME barrier as the reverse sweep's first exchange: nothing the sweep
    // needs up to there depends on f or div (its cotangents are -a_z and a_logp), so f, div and the last layer's vector
    // gradients -- the only consumers of the cross-wave sums -- are formed after that barrier (finish_forward below).
    float g4[D];
    if constexpr (P) {
#pragma unroll
    for (int m = 0; m < D; ++m) g4[m] = ws.gate4[0][m];
    }

    // ------------------------------------------------------------------ reverse sweep (own tile)
    float lf[D];
    if constexpr (P) {
#pragma unroll
    for (int m = 0; m < D; ++m) lf[m] = -az[m];
    }
    const float ld = al;
    f32x4 cot[NS];
    if constexpr (P) {
        float ubar[D], Ubar[D];
#pragma unroll
        for (int m = 0; m < D; ++m) {
            ubar[m] = g4[m] * lf[m];
            Ubar[m] = g4[m] * ld * (HUTCH ? e[m] : 1.0f);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float xb = 0.f, Xh = 0.f;
#pragma unroll
            for (int m = 0; m < D; ++m) {
                xb = fmaf(ubar[m], w4[m][r], xb);
                if (HUTCH) Xh = fmaf(Ubar[m], w4[m][r], Xh);
                else cot[1 + (HUTCH ? 0 : m)][r] = Ubar[m] * w4[m][r];
            }
            cot[0][r] = xb;
            if (HUTCH) cot[1][r] = Xh;
        }
        if (th_hi && kExpSmall) {
            // d/dW4[m][feature] = sum_rows (ubar_m x + Ubar_m X_m): two output rows m per reduce8
#pragma unroll
            for (int m = 0; m < D; m += 2) {
                float va[4], vb[4], ta, tb;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    va[r] = ubar[m] * act[0][r] + Ubar[m] * act[HUTCH ? 1 : 1 + m][r];
                    vb[r] = m + 1 < D ? ubar[m + 1 < D ? m + 1 : m] * act[0][r] + Ubar[m + 1 < D ? m + 1 : m] * act[HUTCH ? 1 : 1 + (m + 1 < D ? m + 1 : m)][r] : 0.f;
                }
                if constexpr (ROLE == 1) {
                    put8(0, va, vb);
                } else {
                    reduce8(va, vb, b0, b1, ta, tb);
                    if (kind < 2 && m + kind < D) lds_add(&small[SI::l4 + (m + kind) * kH + fo + qq], w * (kind == 0 ? ta : tb));
                }
            }
        }
    }
    auto finish_forward = [&]() {   // behind the first barrier of the sweep
        if constexpr (P) {
        float u4[D], U4[D];
        div = 0.f;
#pragma unroll
        for (int m = 0; m < D; ++m) {
            u4[m] = sum4(xc.part[0][m][j], xc.part[1][m][j], xc.part[2][m][j], xc.part[3][m][j]);
            U4[m] = sum4(xc.part[0][D + m][j], xc.part[1][D + m][j], xc.part[2][D + m][j], xc.part[3][D + m][j]);
            f[m] = fmaf(g4[m], u4[m], ws.gate4[1][m]);
            const float dm = g4[m] * U4[m];
            div += HUTCH ? dm * e[m] : dm;
        }
        if (th_hi) {
#pragma unroll
            for (int m = 0; m < D; ++m) {
                const float pa = lf[m], pb = lf[m] * (u4[m] + n.vec4[0][m]) + ld * (HUTCH ? e[m] : 1.0f) * U4[m];
                if constexpr (ROLE == 1) {
                    if (wave == 0 && q == 0) {
                        sb.l4pub[0][m][j] = pa;
                        sb.l4pub[1][m][j] = pb;
                    }
                } else {
                    const float s4 = row16_allsum_s(pa);
                    const float gb = row16_allsum_s(pb);
                    const float gg = g4[m] * (1.0f - g4[m]);
                    const float val = qq < 2 ? s4 : gb;
                    const float fac = qq == 0 ? g4[m] : (qq == 1 ? t : (qq == 2 ? t * gg : gg));
                    if (wave == 0 && lane < 4) lds_add(&small[SI::l4 + kH * D + qq * D + m], w * fac * val);
                }
            }
        }
        }
        mid(f, div);   // every wave: the solve kernel's reduction has a workgroup barrier in it
    };

    pr.mark(12);
    // dW[16*wave + i'][:] += sum_rows (w ubar)[i'][row] * x_in[:][row] of hidden layer lc (K = the 16 rows).  Nothing on
    // the evaluation's critical path depends on these 96 MFMAs, so they are issued one layer LATE, next to the
    // reductions and gate arithmetic of the following layer: matrix pipe and vector ALU overlap.
    auto issue_dw_into = [&](BigAccT<NL>& dst, int lc) {
        if constexpr (!B16) {
            const int buf = 1 - lc;
            const int li = NL == 1 ? 0 : lc;
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const f32x4 a4 = *reinterpret_cast<const f32x4*>(&sb.trA[wave][s][j * kRows + 4 * q]);
                f32x4 b4[4];
#pragma unroll
                for (int T = 0; T < 4; ++T) b4[T] = *reinterpret_cast<const f32x4*>(&sb.trB[buf][s][(16 * T + j) * kRows + 4 * q]);
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                    for (int Ti = 0; Ti < 4; ++Ti) dst.gw[li][Ti] = mfma16(a4[ks], b4[Ti][ks], dst.gw[li][Ti]);
            }
        }
    };
    auto issue_dw = [&](int lc) {
        if constexpr (B16) {
            // unweighted outer products of this evaluation on the bf16 pipe, then one fused multiply-add per register
            f32x4 tmp[4];
            outer_b16<NS>(xc.xcot[lc], xc.xa[lc], wave, lane, tmp);
            const int li = NL == 1 ? 0 : lc;
            if (to_alt) {
#pragma unroll
                for (int Ti = 0; Ti < 4; ++Ti) alt.gw[li][Ti] += w * tmp[Ti];
            } else {
#pragma unroll
                for (int Ti = 0; Ti < 4; ++Ti) acc.gw[li][Ti] += w * tmp[Ti];
            }
        } else {
            if (to_alt) issue_dw_into(alt, lc);
            else issue_dw_into(acc, lc);
        }
    };

#pragma unroll
    for (int l = 1; l >= -1; --l) {
        const int L = l + 1;
        f32x4 g;
        float s1[4], s2[4];
        if constexpr (P) {
        g = *reinterpret_cast<const f32x4*>(&ws.gate[L][0][fo]);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float x = xs[L][r];
            const float dx = act_d(x);
            float pre = dx * cot[0][r];
            float q2 = 0.f;
#pragma unroll
            for (int k = 0; k < NT; ++k) {
                float Uk;
                if (l >= 0) {
                    Uk = U[l < 0 ? 0 : l][k][r];
                } else if (HUTCH) {
                    Uk = 0.f;
#pragma unroll
                    for (int m = 0; m < D; ++m) Uk = fmaf(w1c[m][r], e[m], Uk);
                } else {
                    Uk = w1c[k][r];
                }
                const float Xb = cot[1 + k][r];
                pre = fmaf(Xb * (g[r] * Uk), act_dd(x, dx), pre);
                const float Pb = dx * Xb;
                q2 = fmaf(Pb, Uk, q2);
                cot[1 + k][r] = g[r] * Pb;
            }
            s1[r] = pre;
            s2[r] = fmaf(pre, ub[L][r], q2);
            cot[0][r] = g[r] * pre;
        }
        }
        if constexpr (T) {
            if (l == 0 ? th_hi : th_lo) {
                if (kExpDw && l < 1) issue_dw(l + 1);   // the previous layer's outer products, next to this vector work
            }
        }
        if constexpr (ROLE == 2) if (kExpSmall) {   // kSmallBufNote
            Acc& dst = to_alt ? alt : acc;
            float ta = 0.f, tb = 0.f;
            if (l == 0 && th_hi) {          // behind the sweep's first barrier: W4 and layer 3
                get8(0, ta, tb);
                dst.e[2] += w * (kind == 0 ? ta : tb);   // every lane: entries of lanes with kind >= D are never stored
                get8(1, ta, tb);
                dst.e[0] += vec_term(ta, tb, ws.gate[2][0][fo + qq]);
            }
            if (l == -1 && th_hi) {         // behind the second: the last layer's vectors
#pragma unroll
                for (int m = 0; m < D; ++m) {
                    const float s4 = row16_allsum_s(sb.l4pub[0][m][j]);
                    const float gb = row16_allsum_s(sb.l4pub[1][m][j]);
                    const float g4m = ws.gate4[0][m];
                    const float gg = g4m * (1.0f - g4m);
                    const float val = qq < 2 ? s4 : gb;
                    const float fac = qq == 0 ? g4m : (qq == 1 ? t : (qq == 2 ? t * gg : gg));
                    if (wave == 0 && lane < 4) dst.e[3 + m] += w * fac * val;
                }
            }
            if (l == -1 && th_lo) {         // behind the second: layer 2; layer 1's gate for later
                get8(0, ta, tb);
                dst.e[0] += vec_term(ta, tb, ws.gate[1][0][fo + qq]);
                gs_l1 = ws.gate[0][0][fo + qq];
            }
        }
        if constexpr (P) if ((L == 2 ? th_hi : th_lo) && kExpSmall) {
            // every lane ends up owning one (feature fo + r*, kind) entry of this layer's four vectors: no branch, no idle lane
            if constexpr (ROLE == 1) {
                put8(L == 2 ? 1 : (L == 1 ? 0 : kBufL0), s1, s2);
            } else {
                float st, gt;
                reduce8(s1, s2, b0, b1, st, gt);
                const float gs = b1 ? (b0 ? g[3] : g[2]) : (b0 ? g[1] : g[0]);
                lds_add(&small[SI::vec(L) + kind * kH + fo + qq], vec_term(st, gt, gs));
            }
        }
        pr.mark(l == 1 ? 13 : (l == 0 ? 16 : 19));
        if (l >= 0) {
            const int lc = l < 0 ? 0 : l;
            const int buf = 1 - lc;   // layer 3 publishes through buffer 0, layer 2 through buffer 1
            if constexpr (B16) {
                if constexpr (P) publish_b16<NS>(xc.xcot[lc], cot, wave, lane);
            } else {
#pragma unroll
                for (int s = 0; s < NS; ++s) *reinterpret_cast<f32x4*>(&xc.xb[buf][s][wave][lane][0]) = cot[s];
            }
            if constexpr (!B16) if ((lc == 1 ? th_hi : th_lo) && kExpDw) {
                const f32x4 gi = *reinterpret_cast<const f32x4*>(&ws.gate[L - 1][0][fo]);
#pragma unroll
                for (int s = 0; s < NS; ++s) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float xin;
                        if (s == 0) {
                            xin = xs[L - 1][r];
                        } else {
                            const float xi = xs[L - 1][r];
                            float Uin;
                            if (lc > 0) {
                                Uin = U[lc > 0 ? lc - 1 : 0][s - 1][r];
                            } else if (HUTCH) {
                                Uin = 0.f;
#pragma unroll
                                for (int m = 0; m < D; ++m) Uin = fmaf(w1c[m][r], e[m], Uin);
                            } else {
                                Uin = w1c[s - 1][r];
                            }
                            xin = act_d(xi) * gi[r] * Uin;
                        }
                        sb.trA[wave][s][(4 * q + r) * kRows + pos] = w * cot[s][r];
                        sb.trB[buf][s][(fo + r) * kRows + pos] = xin;
                    }
                }
            }
            __syncthreads();
            if (l == 1) finish_forward();
            pr.mark(l == 1 ? 14 : 17);
            // cotangent of this layer's input, own input-feature tile: (W^T ubar)[16*wave ..]
            f32x4 o4[NS];
            if constexpr (B16) {
                if constexpr (P) hidden_b16_lds<NS>(xc.xcot[lc], sb.wimgT16[lc][wave], lane, o4);
            } else {
#pragma unroll
                for (int s = 0; s < NS; ++s) o4[s] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int To = 0; To < 4; ++To) {
                    const f32x4 a = *reinterpret_cast<const f32x4*>(&sb.wimgT[lc][wave * 4 + To][lane][0]);
                    f32x4 in[NS];
#pragma unroll
                    for (int s = 0; s < NS; ++s) in[s] = *reinterpret_cast<const f32x4*>(&xc.xb[buf][s][To][lane][0]);
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int s = 0; s < NS; ++s) o4[s] = mfma16(a[r], in[s][r], o4[s]);
                }
            }
            if constexpr (P) {
#pragma unroll
            for (int s = 0; s < NS; ++s) cot[s] = o4[s];
            }
            pr.mark(l == 1 ? 15 : 18);
        } else {
            if constexpr (P) {
#pragma unroll
            for (int m = 0; m < D; ++m) {
                float pz = 0.f;
#pragma unroll
                for (int r = 0; r < 4; ++r) pz = fmaf(w1c[m][r], cot[0][r], pz);
                pz = xq_sum(pz);
                if (q == 0) sb.part2[wave][m][j] = pz;
            }
            if (th_lo && kExpSmall) {
                // d/dW1[feature][m] = sum_rows (cot z_m + tangent cotangent_m): two input columns m per reduce8
#pragma unroll
                for (int m = 0; m < D; m += 2) {
                    const int m1 = m + 1 < D ? m + 1 : m;
                    float va[4], vb[4], ta, tb;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        va[r] = fmaf(cot[0][r], z[m], HUTCH ? cot[1][r] * e[m] : cot[1 + (HUTCH ? 0 : m)][r]);
                        vb[r] = m + 1 < D ? fmaf(cot[0][r], z[m1], HUTCH ? cot[1][r] * e[m1] : cot[1 + (HUTCH ? 0 : m1)][r]) : 0.f;
#ifdef CNF_EXP_TW_SCALAR_W1
                        // kPackedW1Note (round 2's workaround, now an A/B switch): each value on its own through an opaque
                        // statement, so that these multiply-adds are not formed as packed fp32 instructions.  In an eight-wave
                        // workgroup the LOW halves of v_pk_fma_f32 ... op_sel:[0,1,0] came out wrong in lanes 48..63 now and
                        // then.  Cause (round 3, tools/hazard_probe.hip): a packed fp32 instruction whose low lane reads the
                        // HIGH dword of src1 misreads it while the SIMD's other wave streams bf16 MFMAs.  The build now
                        // removes that form from this whole translation unit (pk_hazard_pass.py: src0 <-> src1), so these
                        // eight are packed again.
                        if constexpr (ROLE == 1) asm volatile("" : "+v"(va[r]), "+v"(vb[r]));
#endif
                    }
                    if constexpr (ROLE == 1) {
                        put8(kBufW1, va, vb);
                    } else {
                        reduce8(va, vb, b0, b1, ta, tb);
                        if (kind < 2 && m + kind < D) lds_add(&small[(fo + qq) * D + m + kind], w * (kind == 0 ? ta : tb));
                    }
                }
            }
            }
            __syncthreads();
            if constexpr (ROLE == 2) if (th_lo && kExpSmall) {   // behind the evaluation's last barrier: layer 1 and W1
                Acc& dst = to_alt ? alt : acc;
                float ta = 0.f, tb = 0.f;
                get8(kBufL0, ta, tb);
                dst.e[1] += vec_term(ta, tb, gs_l1);
                get8(kBufW1, ta, tb);
                dst.e[2] += w * (kind == 0 ? ta : tb);   // every lane: entries of lanes with kind >= D are never stored
            }
            if constexpr (P) {
#pragma unroll
            for (int m = 0; m < D; ++m)
                vz[m] = sum4(sb.part2[0][m][j], sb.part2[1][m][j], sb.part2[2][m][j], sb.part2[3][m][j]);
            }
            pr.mark(11);
        }
    }
}

// Two-workgroups-per-tile launches keep running totals over the accepted steps -- the matrix in 16 registers, the small
// parameters in sb.small[3] -- and store their slab ONCE when the solve ends: no global read-modify-write between steps.
// MATS: this wave holds matrix accumulators (every wave of a four-wave workgroup; the theta waves of an eight-wave one).
// NTH: threads of the workgroup; the small-parameter sums are spread over all of them.  w: the wave's feature tile.
template <int D, bool MATS = true, int NTH = kThreads>
__device__ inline void commit_total(SplitBwdLds<D>& sb, BigAccT<1>& tot, const BigAccT<1>& acc, const BigAccT<1>& gfirst,
                                    const BigAccT<1>& glast, int i_first, int i_last, float w_first, float w_last, int tid, int part) {
    const int i_lo = part == 1 ? SmallIdxS<D>::v3 : 0, i_hi = part == 2 ? SmallIdxS<D>::v3 : SmallIdxS<D>::count;
    __syncthreads();
    for (int i = i_lo + tid; i < i_hi; i += NTH)
        sb.small[3][i] += fmaf(w_last, sb.small[i_last][i], fmaf(w_first, sb.small[i_first][i], sb.small[0][i]));
    if constexpr (MATS) {
#pragma unroll
    for (int Ti = 0; Ti < 4; ++Ti)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            tot.gw[0][Ti][r] += fmaf(w_last, glast.gw[0][Ti][r], fmaf(w_first, gfirst.gw[0][Ti][r], acc.gw[0][Ti][r]));
    }
    __syncthreads();
}
template <int D, bool MATS = true, int NTH = kThreads>
__device__ inline void store_total(const SplitBwdLds<D>& sb, const BigAccT<1>& tot, float* __restrict__ slab_wg, int tid, int w,
                                   int part) {
    const ParamLayout L = make_layout(D, kH);
    const int lane = tid & 63, q = lane >> 4, j = lane & 15;
    const int i_lo = part == 1 ? SmallIdxS<D>::v3 : 0, i_hi = part == 2 ? SmallIdxS<D>::v3 : SmallIdxS<D>::count;
    __syncthreads();
    for (int i = i_lo + tid; i < i_hi; i += NTH) slab_wg[SmallIdxS<D>::canonical(i)] = sb.small[3][i];
    if constexpr (MATS) {
    const int l = 2 - part;   // part 1 keeps W3 (layer index 1 of the two hidden matrices), part 2 keeps W2
#pragma unroll
    for (int Ti = 0; Ti < 4; ++Ti)
#pragma unroll
        for (int r = 0; r < 4; ++r) slab_wg[L.w(1 + l) + (16 * w + 4 * q + r) * kH + 16 * Ti + j] = tot.gw[0][Ti][r];
    }
}

// eight-wave workgroups: a theta wave's own registers, matrices and small parameters alike (the arithmetic of commit_total)
template <int D>
__device__ __forceinline__ void commit_total_regs(ThetaAccT<D>& tot, const ThetaAccT<D>& acc, const ThetaAccT<D>& gfirst,
                                                  const ThetaAccT<D>& glast, float w_first, float w_last) {
#pragma unroll
    for (int Ti = 0; Ti < 4; ++Ti)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            tot.gw[0][Ti][r] += fmaf(w_last, glast.gw[0][Ti][r], fmaf(w_first, gfirst.gw[0][Ti][r], acc.gw[0][Ti][r]));
#pragma unroll
    for (int i = 0; i < 3 + D; ++i) tot.e[i] += fmaf(w_last, glast.e[i], fmaf(w_first, gfirst.e[i], acc.e[i]));
}
// the theta wave of feature tile w stores what it owns (ThetaAccT): its 16 x 64 block of the part's matrix and one entry
// per lane and small-parameter block
template <int D>
__device__ inline void store_total_regs(const ThetaAccT<D>& tot, float* __restrict__ slab_wg, int lane, int w, int part) {
    using SI = SmallIdxS<D>;
    const ParamLayout L = make_layout(D, kH);
    const int q = lane >> 4, j = lane & 15, qq = j & 3, kind = j >> 2, fo = 16 * w + 4 * q;
    const int l = 2 - part;   // part 1 keeps W3 (layer index 1 of the two hidden matrices), part 2 keeps W2
#pragma unroll
    for (int Ti = 0; Ti < 4; ++Ti)
#pragma unroll
        for (int r = 0; r < 4; ++r) slab_wg[L.w(1 + l) + (16 * w + 4 * q + r) * kH + 16 * Ti + j] = tot.gw[0][Ti][r];
    if (part == 1) {
        slab_wg[SI::canonical(SI::vec(2) + kind * kH + fo + qq)] = tot.e[0];
        if (kind < 2 && kind < D) slab_wg[SI::canonical(SI::l4 + kind * kH + fo + qq)] = tot.e[2];
        if (w == 0 && lane < 4) {
#pragma unroll
            for (int m = 0; m < D; ++m) slab_wg[SI::canonical(SI::l4 + kH * D + qq * D + m)] = tot.e[3 + m];
        }
    } else {
        slab_wg[SI::canonical(SI::vec(1) + kind * kH + fo + qq)] = tot.e[0];
        slab_wg[SI::canonical(SI::vec(0) + kind * kH + fo + qq)] = tot.e[1];
        if (kind < 2 && kind < D) slab_wg[SI::canonical((fo + qq) * D + kind)] = tot.e[2];
    }
}

template <int D>
__device__ inline void commit_step_split(SplitBwdLds<D>& sb, const BigAccS& acc, float* __restrict__ slab_wg, int tid,
                                         bool first = false, int part = 0) {
    const ParamLayout L = make_layout(D, kH);
    const int lane = tid & 63, w = tid >> 6, q = lane >> 4, j = lane & 15;
    const int i_lo = part == 1 ? SmallIdxS<D>::v3 : 0, i_hi = part == 2 ? SmallIdxS<D>::v3 : SmallIdxS<D>::count;
    __syncthreads();
    // first: the slab has not been written yet (and need not have been zeroed): store instead of accumulate
    for (int i = i_lo + tid; i < i_hi; i += kThreads) {
        float* dst = &slab_wg[SmallIdxS<D>::canonical(i)];
        *dst = (first ? 0.f : *dst) + sb.small[0][i];
    }
#pragma unroll
    for (int l = 0; l < 2; ++l)
        if (part == 0 || part == 2 - l)
#pragma unroll
        for (int Ti = 0; Ti < 4; ++Ti)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float* dst = &slab_wg[L.w(1 + l) + (16 * w + 4 * q + r) * kH + 16 * Ti + j];
                *dst = (first ? 0.f : *dst) + acc.gw[l][Ti][r];
            }
    __syncthreads();
}

// commit with the FSAL buffers: step sums + w_first * G(first stage) + w_last * G(seventh stage)
template <int D>
__device__ inline void commit_step_fsal(SplitBwdLds<D>& sb, const BigAccS& acc, const BigAccS& gfirst, const BigAccS& glast,
                                        int i_first, int i_last, float w_first, float w_last,
                                        float* __restrict__ slab_wg, int tid, bool first, int part) {
    const ParamLayout L = make_layout(D, kH);
    const int lane = tid & 63, w = tid >> 6, q = lane >> 4, j = lane & 15;
    const int i_lo = part == 1 ? SmallIdxS<D>::v3 : 0, i_hi = part == 2 ? SmallIdxS<D>::v3 : SmallIdxS<D>::count;
    __syncthreads();
    for (int i = i_lo + tid; i < i_hi; i += kThreads) {
        float* dst = &slab_wg[SmallIdxS<D>::canonical(i)];
        *dst = (first ? 0.f : *dst) + fmaf(w_last, sb.small[i_last][i], fmaf(w_first, sb.small[i_first][i], sb.small[0][i]));
    }
#pragma unroll
    for (int l = 0; l < 2; ++l)
        if (part == 0 || part == 2 - l)
#pragma unroll
        for (int Ti = 0; Ti < 4; ++Ti)
#pragma unroll
            for (int r = 0; r < 4; ++r)
            {
                float* dst = &slab_wg[L.w(1 + l) + (16 * w + 4 * q + r) * kH + 16 * Ti + j];
                *dst = (first ? 0.f : *dst) + fmaf(w_last, glast.gw[l][Ti][r], fmaf(w_first, gfirst.gw[l][Ti][r], acc.gw[l][Ti][r]));
            }
    __syncthreads();
}

// single augmented evaluation (cnf_eval_aug, rows <= 4096)
template <int D, bool HUTCH>
__global__ __launch_bounds__(kThreads) void aug_split_kernel(SolveArgs a, float t, const float* __restrict__ z,
                                                             const float* __restrict__ a_z, const float* __restrict__ a_l,
                                                             float* __restrict__ f_out, float* __restrict__ nd_out,
                                                             float* __restrict__ vz_out) {
    const float res = a.residual ? 1.f : 0.f;   // CNF_DIV_RESIDUAL: dynamics f(z) - z
    constexpr int C = 2 * D + 1;
    __shared__ BwdNet<D> net;
    __shared__ SplitBwdLds<D> sb;
    __shared__ SplitWaveLds<C> wl[kShareWaveLds<D> ? 1 : kNW];
    __shared__ BwdXchg<D> xc;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, j = lane & 15;
    SplitWeightsB16 wb;
    stage_bwd_net<D>(net, sb, wb, a.params, tid);
    for (int i = tid; i < SmallIdxS<D>::padded; i += kThreads) sb.small[0][i] = 0.f;
    __syncthreads();
    const long row = (long)blockIdx.x * kRows + j;
    const bool valid = row < a.rows;
    const long rr = valid ? row : a.rows - 1;
    float zz[D], ee[D], az[D], f[D], vz[D], div;
#pragma unroll
    for (int m = 0; m < D; ++m) {
        zz[m] = z[rr * D + m];
        ee[m] = HUTCH ? a.e[rr * D + m] : 0.f;
        az[m] = valid ? a_z[rr * D + m] : 0.f;
    }
    const float al = valid ? a_l[rr] : 0.f;
    BigAccS acc;
    acc.zero();
    Prof pr;
    aug_eval_split<D, HUTCH>(net, wb, sb, wl[kShareWaveLds<D> ? 0 : w], xc, lane, w, t, zz, ee, az, al, 1.0f, true, f, div, vz, acc, acc, false, 0, 0, pr, NoMid{});
    #pragma unroll
    for (int m = 0; m < D; ++m) {   // residual dynamics f(z) - z (res = 0 or 1): d/dz gains -I
        f[m] = fmaf(-res, zz[m], f[m]);
        vz[m] = fmaf(res, az[m], vz[m]);
    }
    div -= res * (float)D;
    if (w == 0 && lane < 16 && valid) {
#pragma unroll
        for (int m = 0; m < D; ++m) {
            f_out[row * D + m] = f[m];
            vz_out[row * D + m] = vz[m];
        }
        nd_out[row] = -div;
    }
    commit_step_split<D>(sb, acc, dbg_ws(a, a.slab + (size_t)blockIdx.x * make_layout(D, kH).total, sizeof(float) * make_layout(D, kH).total, __LINE__), tid);
}

// RUNTOT: the launch has two workgroups per tile (a.spread) and each keeps ONE matrix: accumulators of 16 registers, so
// a fourth one -- the running total over accepted steps -- fits, and the slab is stored once at the end (commit_total).
//
// kThetaWavesNote -- TW: EIGHT waves per workgroup, two per SIMD.  Waves 0..3 ("producers") integrate the tile as before
// but form no outer products; waves 4..7 ("theta waves") do nothing else: wave 4 + i reads the cotangent and layer-input
// pieces the producers have published in LDS (they are the operands of the outer products as they lie there), forms the
// 16 x 64 block of dW that producer i used to form, and keeps that block's four accumulators (attempt, two FSAL buffers,
// running total) in ITS registers.  With one wave per SIMD nothing hides the producers' latencies (the vector ALU is
// busy a third of the time); the theta wave's LDS reads, MFMAs and multiply-adds issue in those gaps.  Both kinds of
// wave run the same controller on the same totals (every decision is a function of the grid-wide sums in red.tot and of
// launch constants), so they take the same branches and meet at the same barriers; the body below is instantiated once
// per role (ROLE of aug_eval_split) from ONE text, which is what keeps the barrier sequences equal.  Each role is
// register-allocated on its own: 256 registers per wave.  The arithmetic of every sum is unchanged (bit-identical results).
template <int D, bool HUTCH, bool RUNTOT, bool TW = false>
__global__ __launch_bounds__(TW ? 2 * kThreads : kThreads) void dopri5_bwd_split_kernel(SolveArgs a) {
    static_assert(!RUNTOT || kFsalTheta<D>, "running totals are built for the FSAL kernels (d <= 2)");
    static_assert(!TW || (RUNTOT && kBwdB16<D>), "theta waves: two-workgroups-per-tile launches on the bf16 pipe");
    constexpr int NTH = TW ? 2 * kThreads : kThreads;
    const float res = a.residual ? 1.f : 0.f;   // CNF_DIV_RESIDUAL: dynamics f(z) - z
    constexpr int C = 2 * D + 1;
    constexpr int CE = D + 1;
    constexpr int NL = RUNTOT ? 1 : 2;
    // TW: ONE copy of the gates and of the Runge-Kutta slopes for the four producers, as at d = 16 (kShareWaveLds): they
    // hold identical values, a producer reads only the gates it wrote itself, and every reader of a slope is separated from
    // the next evaluation's writers by a workgroup barrier.  The 12 KB pay for the hand-over buffers (BwdSmallLds).
    constexpr bool SHARE = kShareWaveLds<D> || TW;
    __shared__ BwdNet<D> net;
    __shared__ SplitBwdLds<D, TW> sb;
    __shared__ SplitWaveLds<C> wl[SHARE ? 1 : kNW];
    __shared__ BwdXchg<D> xc;
    __shared__ RedLds red;
    const int tid = threadIdx.x, lane = tid & 63, j = lane & 15;
    const int wv = TW ? __builtin_amdgcn_readfirstlane(tid >> 6) : (tid >> 6);   // TW: a scalar, the roles branch on it
    const int w = TW ? (wv & (kNW - 1)) : wv;                                    // the wave's feature tile
    if (tid == 0) red.status = 0;
    // kSpreadNote -- a.spread: TWO workgroups per 16-row tile (grid = 2 x tiles, for batches that would leave half the
    // compute units idle).  Both integrate the tile's state identically -- same arithmetic, same totals from the grid
    // reductions, same decisions -- but each keeps only half of the d/dtheta sums (part 1 / part 2, aug_eval_split), which
    // is a third of an augmented evaluation's time.  Only the even workgroup counts its rows and writes grad_x.
    const int tile = a.spread ? (int)(blockIdx.x >> 1) : (int)blockIdx.x;
    const int part = a.spread ? 1 + (int)(blockIdx.x & 1) : 0;
    const long row = (long)tile * kRows + j;
    const bool valid = row < a.rows;
    const bool counted = valid && lane < 16 && wv == 0 && part != 2;
    const long rr = valid ? row : a.rows - 1;
    SplitWaveLds<C>& ws = wl[SHARE ? 0 : w];
    float* slab = dbg_ws(a, a.slab + (size_t)blockIdx.x * make_layout(D, kH).total, sizeof(float) * make_layout(D, kH).total, __LINE__);

    GridSync gs{dbg_gran(a, __LINE__), (int)gridDim.x, 0u};
    // kRematCtlNote: the batch sizes as doubles and the end time are formed where they are used (once per attempt) from an
    // opaque copy of the kernel argument -- hoisted, each is a 64-bit value that lives through the whole solve in registers
    // the evaluation needs (spilled to scratch at 256 registers per wave)
    auto rows_d = [&]() __attribute__((always_inline)) { long r = a.rows; asm volatile("" : "+s"(r)); return (double)r; };
    auto rows_d_long = [&]() __attribute__((always_inline)) { long r = a.rows; asm volatile("" : "+s"(r)); return r; };
    auto nl_ = [&]() __attribute__((always_inline)) { return rows_d(); };
    auto nz_ = [&]() __attribute__((always_inline)) { return rows_d() * D; };

    // a.loss_out: the cotangents are those of loss = -mean(sum_d log N(z1_d) - logp1) (nll_loss_kernel's arithmetic,
    // cnf_train.hip); the loss itself rides on the second grid reduction, whose third slot is free
    const bool nll = a.loss_out != nullptr;
    const float inv_rows = 1.0f / (float)a.rows;
    float y0[C], y1[C], yi[C], ee[D];
    float nll_row = 0.f;
#pragma unroll
    for (int m = 0; m < D; ++m) {
        y0[m] = a.x[rr * D + m];
        ee[m] = HUTCH ? a.e[rr * D + m] : 0.f;
        y0[D + 1 + m] = valid ? (nll ? y0[m] * inv_rows : a.grad_z1[rr * D + m]) : 0.f;
        nll_row += -0.91893853320467274178f - y0[m] * y0[m] * 0.5f;
    }
    y0[D] = a.logp0[rr];
    nll_row -= y0[D];
    const float al = valid ? (nll ? inv_rows : a.grad_logp1[rr]) : 0.f;
    const int after_status = a.after ? a.after->status : 0;
    // staging comes AFTER the loads above were requested: the rows' round trip to memory overlaps it (dopri5_split_kernel)
    SplitWeightsB16 wb;
    stage_bwd_net<D, NTH>(net, sb, wb, a.params, tid);
    zero_slopes(wl, SHARE ? 1 : kNW, tid, NTH);
    if constexpr (!TW)
        for (int i = tid; i < (int)(sizeof(sb.small) / sizeof(float)); i += NTH) (&sb.small[0][0])[i] = 0.f;
    __syncthreads();

    // everything from here on once per role (kThetaWavesNote): 0 = a wave of a four-wave workgroup, 1 = producer, 2 = theta wave
    auto run = [&](auto role_c) __attribute__((always_inline)) {
    constexpr int ROLE = decltype(role_c)::value;
    constexpr bool P = ROLE != 2, T = ROLE != 1;
    (void)T;
#pragma unroll
    for (int c = 0; c < C; ++c) yi[c] = y0[c];

    const float sgn = a.t0 >= a.t1 ? 1.f : -1.f;
    auto tau_end_ = [&]() __attribute__((always_inline)) { float t0 = a.t0; asm volatile("" : "+s"(t0)); return (double)(sgn * t0); };
    double tau = (double)(sgn * a.t1);
    double dt = 0.0;
    float dtf = 0.f, t0f = (float)tau, ti = t0f;
    const float rtol = a.rtol, atol = a.atol;

    enum { PH_F0 = 0, PH_F1 = 1, PH_STAGE = 2 };
    int phase = PH_F0, stage = 0;
    int nacc = 0, nrej = 0, status = 0, nrun = 0;
    float h0 = 0.f, d1z = 0.f, d1l = 0.f;
    float sc[CE];   // scale of the initial-step heuristic (the solve's initial state)
    if constexpr (P) {
#pragma unroll
    for (int c = 0; c < CE; ++c) sc[c] = atol + fabsf(y0[c]) * rtol;
    }
    constexpr bool FSAL = kFsalTheta<D>;
    // acc: the attempt's weighted sums; gfirst / glast: see kFsalTheta; total: RUNTOT.  TW: the small parameters' too.
    using AccT = typename std::conditional<TW, ThetaAccT<D>, BigAccT<NL>>::type;
    AccT acc, gfirst, glast, total;
    acc.zero();
    gfirst.zero();
    glast.zero();
    total.zero();
    int i_first = 1, i_last = 2;  // which of sb.small[1], sb.small[2] holds the first / seventh stage's sums
    float wq[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   // quadrature weights of the current attempt (every lane holds all seven)
    Prof pr;
    pr.mark(0);

    status = after_status;   // CnfSolverOpts.after: the solve whose output this one consumes failed
    for (; status == 0;) {
        float zz[D], az[D], f[D], vz[D], div, k[C];
        bool restart = false;
        // kRematLaneNote -- eight-wave workgroups run at 256 registers per wave, and the compiler hoists everything that
        // is a function of the lane index alone out of this loop: LDS addresses, lane masks, the 64-bit granule offsets of
        // the grid reductions -- one or two instructions each to recompute, but spilled to scratch once hoisted (38 dwords
        // per lane, re-read every evaluation, all of it written back to HBM when the kernel ends: 19.9 of the launch's
        // 29.8 MB of traffic).  Passing the lane index through an opaque statement once per iteration makes the values
        // derived from THAT copy loop-variant, so they are recomputed where they are used: the grid reductions' copy in
        // both roles, the evaluation's copy in the theta waves (their vector ALU is mostly idle; on the producers' chain
        // the recomputation costs what the reloads did -- CNF_REMAT_LANE bit 1, measured neutral to slightly slower).
        // 152 -> 56 bytes of scratch per lane, 6.14 -> 5.91 us per evaluation on the stiff field; same arithmetic, same bits.
        int lane_r = lane;   // the grid reductions' copy of the lane index
        if constexpr (TW && (kRematLane & 4) != 0) asm volatile("" : "+v"(lane_r));
#pragma unroll
        for (int m = 0; m < D; ++m) {
            zz[m] = yi[m];
            az[m] = yi[D + 1 + m];
        }
        // which d/dtheta sum this evaluation feeds: the step's weighted quadrature, or (FSAL) an unweighted buffer
        const bool to_first = FSAL && phase == PH_F0;
        const bool to_last = FSAL && phase == PH_STAGE && stage == 6;
        float wsel = wq[0];
#pragma unroll
        for (int s_ = 1; s_ < 7; ++s_) wsel = stage == s_ ? wq[s_] : wsel;
        const float wst = (to_first || to_last) ? 1.f : (phase == PH_STAGE ? wsel : 0.f);
        const bool with_theta = wst != 0.f;
        // FSAL evaluations accumulate into glast (zero at that point: start of the solve / start of the attempt)
        const bool to_alt = FSAL && (to_first || to_last);
        const int small = to_alt ? (to_first ? i_first : i_last) * SmallIdxS<D>::padded : 0;
        pr.mark(0);
        // The controller's sums over (z, logp) -- the initial-step norms, or the error ratios of a finished attempt -- from
        // the slopes kc of THIS evaluation.  They need f and div only, which are final when the evaluation's forward half
        // is: the grid-wide reduction is started from there (mid, below) and collected after the reverse sweep, so the
        // granules' trip through the fabric (~3 us) is hidden behind half an evaluation.  Same sums, same order, same
        // bits as reducing afterwards.  (CNF_FLAG_ADJ_CONTROL_ALL needs a_z's slope too: it reduces after the sweep.)
        auto control_sums = [&](const float (&kc)[CE], double (&S)[4]) {
#pragma unroll
            for (int c = 0; c < CE; ++c) {
                if (phase == PH_F0) {
                    const float u = y0[c] / sc[c], v = kc[c] / sc[c];
                    if (counted) {
                        S[c < D ? 0 : 1] += (double)u * u;
                        S[c < D ? 2 : 3] += (double)v * v;
                    }
                } else if (phase == PH_F1) {
                    const float u = (kc[c] - ws.kst[0][c][j]) / sc[c];
                    if (counted) S[c < D ? 0 : 1] += (double)u * u;
                } else {
                    float err = 0.f;
#pragma unroll
                    for (int s = 0; s < 6; ++s) err += (dtf * kCErr[s]) * ws.kst[s][c][j];
                    err += (dtf * kCErr[6]) * kc[c];
                    const float tol = atol + rtol * fmaxf(fabsf(y0[c]), fabsf(yi[c]));   // yi: the candidate y1
                    const float u = err / tol;
                    if (counted) {
                        S[c < D ? 0 : 1] += (double)u * u;
                        if (!isfinite(y0[c])) S[2] += 1.0;
                    }
                }
            }
            if (phase == PH_F1 && nll && counted) S[2] = (double)nll_row;
        };
        bool early = false;   // this evaluation's reduction is already under way (uniform)
        auto mid = [&](const float (&f_)[D], float div_) {
            if (a.adj_all || (phase == PH_STAGE && stage != 6)) return;
            double S[4] = {0, 0, 0, 0};
            if constexpr (P) {
            float kc[CE];
#pragma unroll
            for (int m = 0; m < D; ++m) kc[m] = sgn * fmaf(-res, zz[m], f_[m]);
            kc[D] = -sgn * (div_ - res * (float)D);
            control_sums(kc, S);
            }
            wg_reduce_publish<false, P>(gs, red, S, w, lane_r);
            early = true;
        };
        int lane_e = lane;   // the evaluation's copy (kRematLaneNote)
        if constexpr (TW && (kRematLane & ROLE) != 0) asm volatile("" : "+v"(lane_e));
        aug_eval_split<D, HUTCH, ROLE>(net, wb, sb, ws, xc, lane_e, w, sgn * ti, zz, ee, az, al, sgn * wst, with_theta, f, div, vz, acc,
                                       glast, to_alt, small, part, pr, mid);
        if constexpr (P) {
        #pragma unroll
        for (int m = 0; m < D; ++m) {   // residual dynamics f(z) - z (res = 0 or 1): d/dz gains -I
            f[m] = fmaf(-res, zz[m], f[m]);
            vz[m] = fmaf(res, az[m], vz[m]);
        }
        div -= res * (float)D;
        }
        if (to_first) gfirst = glast;   // once per solve; the restart below clears glast
        ++nrun;
        pr.mark(20);
        if constexpr (P) {
#pragma unroll
        for (int m = 0; m < D; ++m) {
            k[m] = sgn * f[m];
            k[D + 1 + m] = sgn * vz[m];
        }
        k[D] = -sgn * div;
        }

        if (phase == PH_F0) {
            double S[4] = {0, 0, 0, 0};
            if constexpr (P) {
#pragma unroll
            for (int c = 0; c < C; ++c) ws.kst[0][c][j] = k[c];
            }
            if (early) {
                if (!wg_reduce_collect<P>(gs, red, S, w, lane_r)) { status = CNF_ERR_SYNC_TIMEOUT; break; }
            } else {
                if constexpr (P) {
                float kc[CE];
#pragma unroll
                for (int c = 0; c < CE; ++c) kc[c] = k[c];
                control_sums(kc, S);
                }
                if (!wg_allreduce4<false, P>(gs, red, S, w, lane_r)) { status = CNF_ERR_SYNC_TIMEOUT; break; }
            }
            const float d0z = (float)sqrt(S[0] / nz_()), d0l = (float)sqrt(S[1] / nl_());
            d1z = (float)sqrt(S[2] / nz_());
            d1l = (float)sqrt(S[3] / nl_());
            float d0a = 0.f, d1a = 0.f, d0al = 0.f;
            if (a.adj_all) {   // CNF_FLAG_ADJ_CONTROL_ALL (cnf_device.h): a_z and a_logp take part in the heuristic
                double S2[4] = {0, 0, 0, 0};
                if constexpr (P) adj_all_f0_sums<D, C>(y0, k, al, rtol, atol, counted, S2);
                if (!wg_allreduce4<false, P>(gs, red, S2, w, lane_r)) { status = CNF_ERR_SYNC_TIMEOUT; break; }
                d0a = (float)sqrt(S2[0] / nz_());
                d1a = (float)sqrt(S2[1] / nz_());
                d0al = (float)sqrt(S2[2] / nl_());
                if (tid == 0) red.keep[0] = d1a;   // read after the next reduction's barriers
            }
            h0 = initial_h0(d0z, d0l, d1z, d1l, a.adj_all != 0, d0a, d1a, d0al);
            if (h0 < INFINITY) {
                if constexpr (P) {
#pragma unroll
                for (int c = 0; c < C; ++c) yi[c] = y0[c] + h0 * k[c];
                }
                ti = t0f + h0;
            }   // else (adj_all): the second evaluation runs at the initial state; initial_dt() ignores what it returns
            phase = PH_F1;
            continue;
        }
        if (phase == PH_F1) {
            double S[4] = {0, 0, 0, 0};
            if (early) {
                if (!wg_reduce_collect<P>(gs, red, S, w, lane_r)) { status = CNF_ERR_SYNC_TIMEOUT; break; }
            } else {
                if constexpr (P) {
                float kc[CE];
#pragma unroll
                for (int c = 0; c < CE; ++c) kc[c] = k[c];
                control_sums(kc, S);
                if (a.adj_all && counted) {
#pragma unroll
                    for (int m = 0; m < D; ++m) {
                        const int c = D + 1 + m;
                        const float u = (k[c] - ws.kst[0][c][j]) / (atol + fabsf(y0[c]) * rtol);
                        S[3] += (double)u * u;
                    }
                }
                }
                if (!wg_allreduce4<false, P>(gs, red, S, w, lane_r)) { status = CNF_ERR_SYNC_TIMEOUT; break; }
            }
            if (nll && blockIdx.x == 0 && tid == 0) *a.loss_out = -((float)S[2] * (1.0f / (float)rows_d_long()));   // inv_rows, formed here (kRematCtlNote)
            const float d2z = (float)sqrt(S[0] / nz_()) / h0, d2l = (float)sqrt(S[1] / nl_()) / h0;
            const float d2a = (float)sqrt(S[3] / nz_()) / h0;
            dt = (double)initial_dt(h0, d1z, d1l, d2z, d2l, a.adj_all != 0, a.adj_all ? red.keep[0] : 0.f, d2a);
            restart = true;
        } else {
            if constexpr (P) {
#pragma unroll
            for (int c = 0; c < C; ++c) ws.kst[stage][c][j] = k[c];
            }
            if (stage < 6) {
                ++stage;
            } else {
                double S[4] = {0, 0, 0, 0};
                if constexpr (P) {
#pragma unroll
                for (int c = 0; c < C; ++c) y1[c] = yi[c];
                }
                pr.mark(21);
                if (early) {
                    if (!wg_reduce_collect<P>(gs, red, S, w, lane_r)) { status = CNF_ERR_SYNC_TIMEOUT; break; }
                } else {
                    if constexpr (P) {
                    float kc[CE];
#pragma unroll
                    for (int c = 0; c < CE; ++c) kc[c] = k[c];
                    control_sums(kc, S);
                    if (a.adj_all) {   // a_z under error control as well
#pragma unroll
                        for (int m = 0; m < D; ++m) {
                            const int c = D + 1 + m;
                            float err = 0.f;
#pragma unroll
                            for (int s = 0; s < 7; ++s) err += (dtf * kCErr[s]) * ws.kst[s][c][j];
                            const float u = err / (atol + rtol * fmaxf(fabsf(y0[c]), fabsf(y1[c])));
                            if (counted) S[3] += (double)u * u;
                        }
                    }
                    }
                    if (!wg_allreduce4<false, P>(gs, red, S, w, lane_r)) { status = CNF_ERR_SYNC_TIMEOUT; break; }
                }
                pr.mark(22);
                if (S[2] > 0.0) { status = CNF_ERR_NONFINITE; break; }
                const float rz = (float)(S[0] / nz_()), rl = (float)(S[1] / nl_());
                const float ra = (float)(S[3] / nz_());   // 0 unless CNF_FLAG_ADJ_CONTROL_ALL (a_logp's ratio is always 0)
                const bool accept = (rz <= 1.f) && (rl <= 1.f) && (ra <= 1.f);
                if (accept) {
                    ++nacc;
                    if constexpr (TW) {
                        if constexpr (T) {
                            commit_total_regs<D>(total, acc, gfirst, glast, wq[0], wq[6]);
                        }
                    } else if constexpr (RUNTOT) commit_total<D>(sb, total, acc, gfirst, glast, i_first, i_last, wq[0], wq[6], tid, part);
                    else if constexpr (FSAL) commit_step_fsal<D>(sb, acc, gfirst, glast, i_first, i_last, wq[0], wq[6], slab, tid, nacc == 1, part);
                    else commit_step_split<D>(sb, acc, slab, tid, nacc == 1, part);
                    if (!(tau_end_() > tau + dt)) break;
                    if constexpr (P) {
#pragma unroll
                    for (int c = 0; c < C; ++c) y0[c] = y1[c];
                    }
                    if (FSAL) {
                        if constexpr (P) {
#pragma unroll
                        for (int c = 0; c < C; ++c) ws.kst[0][c][j] = k[c];   // the seventh stage is the next step's first
                        }
                        gfirst = glast;
                        const int t_ = i_first;
                        i_first = i_last;
                        i_last = t_;
                    }
                    tau = tau + dt;
                } else {
                    ++nrej;
                }
                pr.mark(23);
                dt = next_step_size(dt, pymax(pymax(rz, rl), ra));
                restart = true;
                pr.mark(24);
            }
        }
        if (restart) {
            if (!(tau + dt > tau)) { status = CNF_ERR_DT_UNDERFLOW; break; }
            if (nacc + nrej >= a.max_steps) { status = CNF_ERR_MAX_STEPS; break; }
            dtf = (float)dt;
            t0f = (float)tau;
            stage = FSAL ? 1 : 0;
            phase = PH_STAGE;
            acc.zero();
            if (FSAL) glast.zero();
            if constexpr (!TW) {
            __syncthreads();   // every wave is past its last read of small of the previous attempt
            for (int i = tid; i < SmallIdxS<D>::padded; i += kThreads) {
                sb.small[0][i] = 0.f;
                if (FSAL) sb.small[i_last][i] = 0.f;
            }
            }
            // quadrature weights of the attempt: dt b_s, or on the step that passes the end time the dense-output weights
            // (tableau entries as literals; every lane computes all seven: no table loads, no LDS hand-off)
            {
                constexpr float kB[7] = {(float)(35.0 / 384), 0.f, (float)(500.0 / 1113), (float)(125.0 / 192), (float)(-2187.0 / 6784),
                                         (float)(11.0 / 84), 0.f};
                constexpr float kM[7] = {(float)(6025192743.0 / 30085553152.0 / 2), 0.f, (float)(51252292925.0 / 65400821598.0 / 2),
                                         (float)(-2691868925.0 / 45128329728.0 / 2), (float)(187940372067.0 / 1594534317056.0 / 2),
                                         (float)(-1776094331.0 / 19743644256.0 / 2), (float)(11237099.0 / 235043384.0 / 2)};
                if (tau_end_() > tau + dt) {
#pragma unroll
                    for (int s_ = 0; s_ < 7; ++s_) wq[s_] = s_ < 6 ? dtf * kB[s_] : 0.f;
                } else {
                    const float ta = (float)tau, tb = (float)(tau + dt), te = (float)tau_end_();
                    const float x = (te - ta) / (tb - ta);
                    const float x2 = x * x, x3 = x2 * x, x4 = x3 * x;
#pragma unroll
                    for (int s_ = 0; s_ < 7; ++s_) {
                        const float b = kB[s_], cm = kM[s_];
                        float v = dtf * (x4 * (-8.f * b + 16.f * cm) + x3 * (14.f * b - 32.f * cm) + x2 * (-5.f * b + 16.f * cm));
                        if (s_ == 0) v += dtf * (-2.f * x4 + 5.f * x3 - 4.f * x2 + x);
                        if (s_ == 6) v += dtf * (2.f * x4 - 3.f * x3 + x2);
                        wq[s_] = v;
                    }
                }
                // every lane has computed the same seven weights: keep them as wave-uniform (scalar) values
                if constexpr (TW) {
#pragma unroll
                    for (int s_ = 0; s_ < 7; ++s_) wq[s_] = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(wq[s_])));
                }
            }
            if constexpr (!TW) __syncthreads();
            pr.mark(25);
        }
        if constexpr (P) {
        if (stage == 0) {
            ti = t0f;
#pragma unroll
            for (int c = 0; c < C; ++c) yi[c] = y0[c];
        } else {
            ti = t0f + stage_state(stage, dtf, y0, [&](int s, int c) { return ws.kst[s][c][j]; }, yi) * dtf;
        }
        } else {
            ti = t0f + stage_node(stage) * dtf;   // a theta wave needs the evaluation's time for the gate vectors' factors
        }
        pr.mark(26);
    }

    const unsigned taken = grid_ticket(gs, tid);   // all waves are past the last reduction's closing barrier
    int lane_x = lane;   // the row index of the outputs, formed here (kRematCtlNote)
    asm volatile("" : "+v"(lane_x));
    unsigned bx = blockIdx.x;
    asm volatile("" : "+s"(bx));
    const long row_e = (long)(a.spread ? (int)(bx >> 1) : (int)bx) * kRows + (lane_x & 15);
    if (P && status == 0) {
        const float ta = (float)tau, tb = (float)(tau + dt), te = (float)tau_end_();
        const float x = (te - ta) / (tb - ta);
        const float x2 = x * x, x3 = x2 * x, x4 = x3 * x;
#pragma unroll
        for (int m = 0; m < D; ++m) {
            const int c = D + 1 + m;
            float mid = 0.f;
#pragma unroll
            for (int s = 0; s < 7; ++s) mid += (dtf * kCMid[s]) * ws.kst[s][c][j];
            mid = y0[c] + mid;
            const float f0 = ws.kst[0][c][j], f1 = ws.kst[6][c][j], u = y0[c], v = y1[c];
            const float ca = (-2 * dtf) * f0 + (2 * dtf) * f1 + -8 * u + -8 * v + 16 * mid;
            const float cb = (5 * dtf) * f0 + (-3 * dtf) * f1 + 18 * u + 14 * v + -32 * mid;
            const float cc = (-4 * dtf) * f0 + dtf * f1 + -11 * u + -5 * v + 16 * mid;
            const float cd = dtf * f0;
            const float out = ca * x4 + cb * x3 + cc * x2 + cd * x + u;
            if (counted) a.grad_x[row_e * D + m] = out;
        }
        if (counted) a.grad_logp0[row_e] = al;
    }
    if constexpr (TW) {
        if constexpr (T) {
            store_total_regs<D>(total, slab, lane, w, part);
        }
    } else if constexpr (RUNTOT) store_total<D>(sb, total, slab, tid, w, part);
    grid_cleanup(gs, red, tid, taken);
    write_stats_global(a.stats, status, 2 + 6 * (nacc + nrej), nacc, nrej, nrun, (float)dt, a.rows, tid);
#ifdef CNF_PROF_CYCLES
    if ((blockIdx.x == 0 || blockIdx.x == 1) && (tid == 0 || tid == 128) && nrun > 100) {
        unsigned long long tot = 0;
        for (int i = 0; i < 32; ++i) tot += pr.acc[i];
        const double n = nrun;
        printf("bwd prof wg %d wave %d evals %d cycles/eval total %.0f: rk %.0f gates %.0f L1 %.0f | pub0 %.0f mm0 %.0f act0 %.0f | pub1 %.0f mm1 %.0f "
               "act1 %.0f | L4part %.0f bar %.0f | revhead %.0f | r3vec %.0f r3pub %.0f r3mm %.0f | r2vec %.0f r2pub %.0f r2mm %.0f | r1vec %.0f r1end %.0f\n",
               (int)blockIdx.x, tid >> 6, nrun, tot / n, pr.acc[0] / n, pr.acc[1] / n, pr.acc[2] / n, pr.acc[3] / n, pr.acc[4] / n, pr.acc[5] / n,
               pr.acc[6] / n, pr.acc[7] / n, pr.acc[8] / n, pr.acc[9] / n, pr.acc[10] / n, pr.acc[12] / n, pr.acc[13] / n, pr.acc[14] / n,
               pr.acc[15] / n, pr.acc[16] / n, pr.acc[17] / n, pr.acc[18] / n, pr.acc[19] / n, pr.acc[11] / n);
        printf("bwd prof wg %d wave %d between evaluations, cycles/eval: k+kst %.0f errest %.0f reduce %.0f decide+commit %.0f next_dt %.0f restart %.0f "
               "stage_state %.0f looptop %.0f\n", (int)blockIdx.x, tid >> 6, pr.acc[20] / n, pr.acc[21] / n, pr.acc[22] / n, pr.acc[23] / n,
               pr.acc[24] / n, pr.acc[25] / n, pr.acc[26] / n, pr.acc[0] / n);
    }
#endif
    };   // run
    if constexpr (!TW) {
        run(std::integral_constant<int, 0>{});
    } else {
        if (wv < kNW) run(std::integral_constant<int, 1>{});
        else run(std::integral_constant<int, 2>{});
    }
}

// ---- the adjoint solve on a fixed grid, one tile per workgroup (kFixedSplitNote, cnf_split_fwd_tw.h) -------------------------
// fixed_bwd_kernel (cnf_bwd.hip: one wave per tile, 32 workgroups at the training batch size) with the feature-split
// augmented evaluation of the dopri5 adjoint above: same grid on the negated time axis (points tau_i = -t1 + i h, the last
// clipped to -t0), same stage states and update expressions, same quadrature weights of the stages' vjp_theta (rk4 = the
// 3/8 rule's dt/8 (1, 3, 3, 1); midpoint: the second stage only; euler: dt).  No step control: nothing is rejected, so the
// weighted sums of ALL steps gather in one accumulator set and the slab is stored once, when the solve ends.
// TW = false: four waves, both matrices' sums in registers (BigAccS), the small parameters' in sb.small[0].
// TW = true (a.spread: two workgroups per tile, each keeping half of the sums): eight waves, producers and theta waves
// (kThetaWavesNote); a theta wave stores what it owns (store_total_regs).
template <int D, bool HUTCH, bool TW>
__global__ __launch_bounds__(TW ? 2 * kThreads : kThreads) void fixed_bwd_split_kernel(SolveArgs a) {
    static_assert(kFsalTheta<D> && kBwdB16<D>, "d <= 2 (d = 16: cnf_split16.hip)");
    constexpr int NTH = TW ? 2 * kThreads : kThreads;
    const float res = a.residual ? 1.f : 0.f;   // CNF_DIV_RESIDUAL: dynamics f(z) - z
    constexpr int C = 2 * D + 1;
    constexpr bool SHARE = kShareWaveLds<D> || TW;
    __shared__ BwdNet<D> net;
    __shared__ SplitBwdLds<D, TW> sb;
    __shared__ SplitWaveLds<C> wl[SHARE ? 1 : kNW];
    __shared__ BwdXchg<D> xc;
    const int tid = threadIdx.x, lane = tid & 63, j = lane & 15;
    const int wv = TW ? __builtin_amdgcn_readfirstlane(tid >> 6) : (tid >> 6);
    const int w = TW ? (wv & (kNW - 1)) : wv;
    const int tile = TW ? (int)(blockIdx.x >> 1) : (int)blockIdx.x;   // kSpreadNote
    const int part = TW ? 1 + (int)(blockIdx.x & 1) : 0;
    const long row = (long)tile * kRows + j;
    const bool valid = row < a.rows;
    const bool counted = valid && lane < 16 && wv == 0 && part != 2;
    const long rr = valid ? row : a.rows - 1;
    SplitWaveLds<C>& ws = wl[SHARE ? 0 : w];
    float* slab = dbg_ws(a, a.slab + (size_t)blockIdx.x * make_layout(D, kH).total, sizeof(float) * make_layout(D, kH).total, __LINE__);

    float y[C], ee[D];
#pragma unroll
    for (int m = 0; m < D; ++m) {
        y[m] = a.x[rr * D + m];
        ee[m] = HUTCH ? a.e[rr * D + m] : 0.f;
        y[D + 1 + m] = valid ? a.grad_z1[rr * D + m] : 0.f;   // padded rows carry zero adjoints: they add nothing to any sum
    }
    y[D] = a.logp0[rr];
    const float al = valid ? a.grad_logp1[rr] : 0.f;   // d(a_logp)/dt = 0: constant along the solve
    const int status = a.after ? a.after->status : 0;  // CnfSolverOpts.after: the solve whose output this one consumes failed
    SplitWeightsB16 wb;
    stage_bwd_net<D, NTH>(net, sb, wb, a.params, tid);
    if constexpr (!TW)
        for (int i = tid; i < (int)(sizeof(sb.small) / sizeof(float)); i += NTH) (&sb.small[0][0])[i] = 0.f;
    __syncthreads();

    const int nst = a.fixed_kind == CNF_SOLVER_EULER ? 1 : (a.fixed_kind == CNF_SOLVER_MIDPOINT ? 2 : 4);
    const int n_steps = status == 0 ? a.n_grid : 0;

    // once per role (kThetaWavesNote): 0 = a wave of a four-wave workgroup, 1 = producer, 2 = theta wave
    auto run = [&](auto role_c) __attribute__((always_inline)) {
    constexpr int ROLE = decltype(role_c)::value;
    constexpr bool P = ROLE != 2, T = ROLE != 1;
    using AccT = typename std::conditional<TW, ThetaAccT<D>, BigAccS>::type;
    AccT acc, unused;
    acc.zero();
    unused.zero();
    // the adjoint pass runs from t1 back to t0
    const float sgn = a.t0 >= a.t1 ? 1.f : -1.f;
    const float tau0 = sgn * a.t1, tau1 = sgn * a.t0;
    Prof pr;
    for (int step = 0; step < n_steps; ++step) {
        // grid point i = i*step_size + tau0 (product rounded before the add, like arange()*h + t0): fixed_bwd_kernel's arithmetic
        const float ta = a.n_grid == 1 ? tau0 : __fadd_rn(__fmul_rn((float)step, a.step_size), tau0);
        const float tb = (step + 1 == a.n_grid) ? tau1 : __fadd_rn(__fmul_rn((float)(step + 1), a.step_size), tau0);
        const float dt = tb - ta;
        float k1[C], k2[C], k3[C], yi[C];
        float ti = ta;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            yi[c] = y[c];
            k1[c] = k2[c] = k3[c] = 0.f;
        }
        for (int st = 0; st < nst; ++st) {
            float zz[D], az[D], f[D], vz[D], div = 0.f, k[C];
#pragma unroll
            for (int m = 0; m < D; ++m) {
                zz[m] = yi[m];
                az[m] = yi[D + 1 + m];
                f[m] = vz[m] = 0.f;
            }
            // weight of this stage's vjp_theta in the step's increment of a_theta
            float wq;
            if (a.fixed_kind == CNF_SOLVER_EULER) wq = dt;
            else if (a.fixed_kind == CNF_SOLVER_MIDPOINT) wq = st == 0 ? 0.f : dt;
            else wq = (st == 0 || st == 3) ? dt / 8 : 3 * (dt / 8);
            int lane_e = lane;   // kRematLaneNote
            if constexpr (TW && (kRematLane & ROLE) != 0) asm volatile("" : "+v"(lane_e));
            aug_eval_split<D, HUTCH, ROLE>(net, wb, sb, ws, xc, lane_e, w, sgn * ti, zz, ee, az, al, sgn * wq, wq != 0.f, f, div, vz, acc,
                                           unused, false, 0, part, pr, NoMid{});
            if constexpr (P) {
#pragma unroll
                for (int m = 0; m < D; ++m) {   // residual dynamics f(z) - z (res = 0 or 1): d/dz gains -I
                    f[m] = fmaf(-res, zz[m], f[m]);
                    vz[m] = fmaf(res, az[m], vz[m]);
                }
                div -= res * (float)D;
#pragma unroll
                for (int m = 0; m < D; ++m) {
                    k[m] = sgn * f[m];
                    k[D + 1 + m] = sgn * vz[m];
                }
                k[D] = -sgn * div;
            }
            // fixed_bwd_kernel's updates; a theta wave only follows the evaluation times (its gate factors need them)
            if (a.fixed_kind == CNF_SOLVER_EULER) {            // y + dt f(t, y)
                if constexpr (P) {
#pragma unroll
                    for (int c = 0; c < C; ++c) y[c] = y[c] + dt * k[c];
                }
            } else if (a.fixed_kind == CNF_SOLVER_MIDPOINT) {  // y + dt f(t + dt/2, y + f(t, y) dt/2)
                if (st == 0) {
                    ti = ta + dt / 2;
                    if constexpr (P) {
#pragma unroll
                        for (int c = 0; c < C; ++c) yi[c] = y[c] + k[c] * (dt / 2);
                    }
                } else {
                    if constexpr (P) {
#pragma unroll
                        for (int c = 0; c < C; ++c) y[c] = y[c] + dt * k[c];
                    }
                }
            } else if (st == 0) {                              // the 3/8 rule
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
                if constexpr (P) {
#pragma unroll
                    for (int c = 0; c < C; ++c) y[c] = y[c] + (k1[c] + 3 * k2[c] + 3 * k3[c] + k[c]) * (dt / 8);
                }
            }
        }
    }
    if (P && status == 0 && counted) {
#pragma unroll
        for (int m = 0; m < D; ++m) a.grad_x[row * D + m] = y[D + 1 + m];
        a.grad_logp0[row] = al;
    }
    if constexpr (TW) {
        if constexpr (T) store_total_regs<D>(acc, slab, lane, w, part);
    } else {
        commit_step_split<D>(sb, acc, slab, tid, true, part);   // stores (first = true): the slab need not be zero
    }
    };   // run
    if constexpr (!TW) {
        run(std::integral_constant<int, 0>{});
    } else {
        if (wv < kNW) run(std::integral_constant<int, 1>{});
        else run(std::integral_constant<int, 2>{});
    }
    const int nrun = nst * n_steps;
    write_stats_global(a.stats, status, nrun, n_steps, 0, nrun, 0.f, a.rows, tid);
}

// ---- launchers -----------------------------------------------------------------------------------------------
#if CNF_SPLIT_PART == 2
// One wave's share of a parameter's column of slabs: g = wv, wv + 4, ... added in that order.  The slabs were written by
// other compute units (other XCDs: every load is a trip past the local L2), so 32 loads are in flight before the first
// addition -- a 128-tile batch is ONE round trip per thread instead of four.
__device__ __forceinline__ float slab_column_sum(const float* __restrict__ src, size_t stride, int wv, int nwg) {
    float acc = 0.f;
    int g = wv;
    for (; g + 124 < nwg; g += 128) {
        float x[32];
#pragma unroll
        for (int k = 0; k < 32; ++k) x[k] = src[(size_t)(g + 4 * k) * stride];
#pragma unroll
        for (int k = 0; k < 32; ++k) acc += x[k];
    }
    for (; g + 28 < nwg; g += 32) {
        float x[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) x[k] = src[(size_t)(g + 4 * k) * stride];
#pragma unroll
        for (int k = 0; k < 8; ++k) acc += x[k];
    }
    for (; g < nwg; g += 4) acc += src[(size_t)g * stride];
    return acc;
}

// Sum of the workgroups' slabs in a FIXED order (deterministic gradients).  64 parameters per block; the four waves
// of a block take the slabs g = wave, wave + 4, ... (eight independent loads in flight per thread), LDS combines the
// four partial sums as (w0 + w1) + (w2 + w3).
// split_at > 0 (spread launches): tile g's sums for parameter p live in slab 2g (p >= split_at) or 2g + 1 (p < split_at);
// the order of the additions over tiles is that of an ordinary launch, so the two modes give identical sums.
__global__ __launch_bounds__(256) void reduce_slab_split_kernel(const float* __restrict__ slab, int nwg, int P,
                                                                float* __restrict__ out, int split_at) {
    __shared__ float part[4][64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int p = blockIdx.x * 64 + lane;
    float acc = 0.f;
    if (p < P) {
        const size_t stride = split_at > 0 ? 2 * (size_t)P : (size_t)P;
        const float* src = slab + p + (split_at > 0 && p < split_at ? (size_t)P : 0);
        acc = slab_column_sum(src, stride, wv, nwg);
    }
    part[wv][lane] = acc;
    __syncthreads();
    if (wv == 0 && p < P) out[p] = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
}

hipError_t launch_reduce_slab(const float* slab, int nwg, int P, float* out, hipStream_t s, int split_at) {
    hipLaunchKernelGGL(reduce_slab_split_kernel, dim3((P + 63) / 64), dim3(256), 0, s, slab, nwg, P, out, split_at);
    return hipGetLastError();
}

// reduce_slab_split_kernel with the Adam update of the 64 parameters as its epilogue (cnf_solve_bwd_train): the sum is
// formed exactly as above, so parameters and moments get the bits the two separate kernels would give them.
__global__ __launch_bounds__(256) void adam_slab_kernel(const float* __restrict__ slab, int nwg, int P, int split_at,
                                                        float* __restrict__ grad_out, float* __restrict__ p,
                                                        float* __restrict__ m, float* __restrict__ v, int* __restrict__ state,
                                                        CnfAdamOpts o, const CnfStats* __restrict__ guard) {
    // the adjoint solve (or a solve before it, CnfSolverOpts.after) failed: its slabs are stale or garbage.  No update, no
    // step: parameters, moments and step count keep their bits (every workgroup reads the same word: uniform exit)
    if (guard && guard->status != 0) return;
    __shared__ float part[4][64];
    __shared__ float corr[2];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + lane;
    // every independent load is issued before anything waits: the step count, the parameter and its moments (wave 0),
    // the slabs (all waves); the bias corrections (two double-precision pow) are wave 3's job while the others add
    const int step = __hip_atomic_load(&state[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    float pi = 0.f, mi = 0.f, vi = 0.f;
    if (wv == 0 && i < P) {
        pi = p[i];
        mi = m[i];
        vi = v[i];
    }
    float acc = 0.f;
    if (i < P) {
        const size_t stride = split_at > 0 ? 2 * (size_t)P : (size_t)P;
        const float* src = slab + i + (split_at > 0 && i < split_at ? (size_t)P : 0);
        acc = slab_column_sum(src, stride, wv, nwg);
    }
    part[wv][lane] = acc;
    if (threadIdx.x == 255) adam_corrections(step, o, corr);
    __syncthreads();
    if (wv == 0 && i < P) {
        const AdamCoef c = adam_coef(o, corr);
        const float g = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
        if (grad_out) grad_out[i] = g;
        adam_apply(c, g, pi, mi, vi, p, m, v, i);
    }
    adam_end(state, step);
}

hipError_t launch_adam_slabs(const float* slab, int ntiles, int P, int split_at, float* grad_out, float* p, float* m, float* v,
                             int* state, const CnfAdamOpts& o, const CnfStats* guard, hipStream_t s) {
    hipLaunchKernelGGL(adam_slab_kernel, dim3((P + 63) / 64), dim3(256), 0, s, slab, ntiles, P, split_at, grad_out, p, m, v,
                       state, o, guard);
    return hipGetLastError();
}

#endif   // CNF_SPLIT_PART == 2
#if CNF_SPLIT_PART == 1
}  // namespace cnf
#include "cnf_split_fwd_tw.h"
namespace cnf {

// eight waves per workgroup (kFwdTwNote, cnf_split_fwd_tw.h) wherever the tangent streams are a separable share of the
// work: d <= 2.  CNF_FLAG_FWD_FOUR_WAVES keeps the four-wave kernel (same bits).
#ifdef CNF_NO_FWD_HELPER_WAVES
constexpr bool kFwdHelperWaves = false;
#else
constexpr bool kFwdHelperWaves = true;
#endif

// one helper group per tangent stream (twelve waves at d = 2 with the exact divergence: kFwdTwGroupsNote); 0 keeps one group
#ifndef CNF_FWD_HELPER_GROUPS
#define CNF_FWD_HELPER_GROUPS 1
#endif

template <int D, bool HUTCH>
static hipError_t split_fwd_t(const SolveArgs& a, hipStream_t s) {
    const int grid = (int)((a.rows + kRows - 1) / kRows);
    if constexpr (D <= 2) {
        if (kFwdHelperWaves && !a.fwd4) {
            constexpr int NH = (CNF_FWD_HELPER_GROUPS != 0 && !HUTCH) ? D : 1;
            hipError_t res = check_resident(dopri5_split_tw_kernel<D, HUTCH, NH>, (1 + NH) * kThreads, grid);
            if (res != hipSuccess) return res;
            dbg_timing_begin(s);
            hipLaunchKernelGGL((dopri5_split_tw_kernel<D, HUTCH, NH>), dim3(grid), dim3((1 + NH) * kThreads), 0, s, a);
            dbg_timing_end(s);
            return hipGetLastError();
        }
    }
    hipError_t res = check_resident(dopri5_split_kernel<D, HUTCH>, kThreads, grid);
    if (res != hipSuccess) return res;
    dbg_timing_begin(s);
    hipLaunchKernelGGL((dopri5_split_kernel<D, HUTCH>), dim3(grid), dim3(kThreads), 0, s, a);
    dbg_timing_end(s);
    return hipGetLastError();
}

hipError_t launch_split_fwd(int D, bool hutch, const SolveArgs& a, hipStream_t s) {
    if (D == 1 && !hutch) return split_fwd_t<1, false>(a, s);
    if (D == 2 && !hutch) return split_fwd_t<2, false>(a, s);
    if (D == 2 && hutch) return split_fwd_t<2, true>(a, s);
    if (D == 16 && hutch) {
        // state spread over the lane quarters (cnf_split16.hip) when its 16-byte accesses are aligned
        const uintptr_t al = (uintptr_t)a.x | (uintptr_t)a.e | (uintptr_t)a.z_out | (uintptr_t)a.params;
        return (al & 15) == 0 ? launch_split16_fwd(a, s) : split_fwd_t<16, true>(a, s);
    }
    return hipErrorNotSupported;
}

// fixed-grid solves with one tile per workgroup (kFixedSplitNote, cnf_split_fwd_tw.h): d <= 2, any grid size -- no
// grid-wide reduction, so nothing has to be co-resident
template <int D, bool HUTCH>
static hipError_t split_fixed_fwd_t(const SolveArgs& a, hipStream_t s) {
    const int grid = (int)((a.rows + kRows - 1) / kRows);
    constexpr int NH = (CNF_FWD_HELPER_GROUPS != 0 && !HUTCH) ? D : 1;
    dbg_timing_begin(s);
    hipLaunchKernelGGL((fixed_split_tw_kernel<D, HUTCH, NH>), dim3(grid), dim3((1 + NH) * kThreads), 0, s, a);
    dbg_timing_end(s);
    return hipGetLastError();
}

hipError_t launch_split_fixed_fwd(int D, bool hutch, const SolveArgs& a, hipStream_t s) {
    if (D == 1 && !hutch) return split_fixed_fwd_t<1, false>(a, s);
    if (D == 2 && !hutch) return split_fixed_fwd_t<2, false>(a, s);
    if (D == 2 && hutch) return split_fixed_fwd_t<2, true>(a, s);
    return hipErrorNotSupported;
}

#endif   // CNF_SPLIT_PART == 1
#if CNF_SPLIT_PART == 2
template <int D, bool HUTCH>
static hipError_t split_bwd_t(const SolveArgs& a, hipStream_t s) {
    const int grid = (int)((a.rows + kRows - 1) / kRows);
    hipError_t res = hipSuccess;
    if constexpr (kFsalTheta<D>) {
        res = a.spread ? check_resident(dopri5_bwd_split_kernel<D, HUTCH, true, kThetaWaves>, kThetaWaves ? 2 * kThreads : kThreads, 2 * grid)
                       : check_resident(dopri5_bwd_split_kernel<D, HUTCH, false>, kThreads, grid);
    } else {
        res = check_resident(dopri5_bwd_split_kernel<D, HUTCH, false>, kThreads, a.spread ? 2 * grid : grid);
    }
    if (res != hipSuccess) return res;
    dbg_timing_begin(s);
    if constexpr (kFsalTheta<D>) {
        if (a.spread) hipLaunchKernelGGL((dopri5_bwd_split_kernel<D, HUTCH, true, kThetaWaves>), dim3(2 * grid),
                                         dim3(kThetaWaves ? 2 * kThreads : kThreads), 0, s, a);
        else hipLaunchKernelGGL((dopri5_bwd_split_kernel<D, HUTCH, false>), dim3(grid), dim3(kThreads), 0, s, a);
    } else {
        hipLaunchKernelGGL((dopri5_bwd_split_kernel<D, HUTCH, false>), dim3(a.spread ? 2 * grid : grid), dim3(kThreads), 0, s, a);
    }
    dbg_timing_end(s);
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) return err;
    if (a.keep_slabs) return hipSuccess;
    const ParamLayout L = make_layout(D, kH);
    const int P = L.total;
    hipLaunchKernelGGL(reduce_slab_split_kernel, dim3((P + 63) / 64), dim3(256), 0, s, a.slab, grid, P, a.grad_params,
                       a.spread ? L.w(2) : 0);
    return hipGetLastError();
}

hipError_t launch_split_bwd(int D, bool hutch, const SolveArgs& a, hipStream_t s) {
    if (D == 1 && !hutch) return split_bwd_t<1, false>(a, s);
    if (D == 2 && !hutch) return split_bwd_t<2, false>(a, s);
    if (D == 2 && hutch) return split_bwd_t<2, true>(a, s);
    if (D == 16 && hutch) {
        const uintptr_t al = (uintptr_t)a.x | (uintptr_t)a.e | (uintptr_t)a.grad_z1 | (uintptr_t)a.grad_x | (uintptr_t)a.params;
        return (al & 15) == 0 ? launch_split16_bwd(a, s) : split_bwd_t<16, true>(a, s);
    }
    return hipErrorNotSupported;
}

// fixed-grid adjoint with one tile per workgroup (fixed_bwd_split_kernel): d <= 2; a.spread selects the eight-wave,
// two-workgroups-per-tile launch.  The slabs are stored, not added to: the workspace need not be cleared.
template <int D, bool HUTCH>
static hipError_t split_fixed_bwd_t(const SolveArgs& a, hipStream_t s) {
    const int grid = (int)((a.rows + kRows - 1) / kRows);
    dbg_timing_begin(s);
    if (a.spread && kThetaWaves) hipLaunchKernelGGL((fixed_bwd_split_kernel<D, HUTCH, true>), dim3(2 * grid), dim3(2 * kThreads), 0, s, a);
    else hipLaunchKernelGGL((fixed_bwd_split_kernel<D, HUTCH, false>), dim3(grid), dim3(kThreads), 0, s, a);
    dbg_timing_end(s);
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) return err;
    if (a.keep_slabs) return hipSuccess;   // cnf_solve_bwd_train: launch_adam_slabs sums them
    const ParamLayout L = make_layout(D, kH);
    hipLaunchKernelGGL(reduce_slab_split_kernel, dim3((L.total + 63) / 64), dim3(256), 0, s, a.slab, grid, L.total, a.grad_params,
                       (a.spread && kThetaWaves) ? L.w(2) : 0);
    return hipGetLastError();
}

hipError_t launch_split_fixed_bwd(int D, bool hutch, const SolveArgs& a, hipStream_t s) {
    if (D == 1 && !hutch) return split_fixed_bwd_t<1, false>(a, s);
    if (D == 2 && !hutch) return split_fixed_bwd_t<2, false>(a, s);
    if (D == 2 && hutch) return split_fixed_bwd_t<2, true>(a, s);
    return hipErrorNotSupported;
}

template <int D, bool HUTCH>
static hipError_t split_aug_t(const SolveArgs& a, float t, const float* z, const float* a_z, const float* a_l, float* f_out,
                              float* nd_out, float* vz_out, hipStream_t s) {
    const int grid = (int)((a.rows + kRows - 1) / kRows);
    hipLaunchKernelGGL((aug_split_kernel<D, HUTCH>), dim3(grid), dim3(kThreads), 0, s, a, t, z, a_z, a_l, f_out, nd_out, vz_out);
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) return err;
    const int P = make_layout(D, kH).total;
    hipLaunchKernelGGL(reduce_slab_split_kernel, dim3((P + 63) / 64), dim3(256), 0, s, a.slab, grid, P, a.grad_params, 0);
    return hipGetLastError();
}

hipError_t launch_split_aug(int D, bool hutch, const SolveArgs& a, float t, const float* z, const float* a_z,
                            const float* a_l, float* f_out, float* nd_out, float* vz_out, hipStream_t s) {
    if (D == 1 && !hutch) return split_aug_t<1, false>(a, t, z, a_z, a_l, f_out, nd_out, vz_out, s);
    if (D == 2 && !hutch) return split_aug_t<2, false>(a, t, z, a_z, a_l, f_out, nd_out, vz_out, s);
    if (D == 2 && hutch) return split_aug_t<2, true>(a, t, z, a_z, a_l, f_out, nd_out, vz_out, s);
    if (D == 16 && hutch) return split_aug_t<16, true>(a, t, z, a_z, a_l, f_out, nd_out, vz_out, s);
    return hipErrorNotSupported;
}

#endif   // CNF_SPLIT_PART == 2
}  // namespace cnf
