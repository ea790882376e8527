// cnf_split16.hip -- d = 16 (Hutchinson) batch-global solves with the STATE SPREAD OVER THE LANE QUARTERS.
//
// The generic feature-split kernels (cnf_split.hip) keep, in every lane, all d components of the lane's row, replicated
// in the four lane quarters, and run the first (K = d) and last (N = d) layers on the vector ALU.  That is right for
// d <= 2 and wrong for d = 16: 17 (forward) / 33 (adjoint) Runge-Kutta components per lane, 128-FMA first layers, a last
// layer of 16 cross-quarter sums, sixteen cross-row reduction chains for the gradients of W1 and W4 -- 512 VGPRs, spills.
// Here lane (q, j) of a wave holds components 4q..4q+3 of row j (and its own copy of logp).  That IS the C/D layout
// of v_mfma_f32_16x16x4_f32 for a 16 x 16 tile, hence also a valid B operand:
//   * layer 1 (16 -> 64): 4 MFMAs per stream on the state registers themselves, A operand = 4 registers per lane
//     loaded once per kernel straight from the row-major nn.Linear weight (16 contiguous bytes);
//   * layer 4 (64 -> 16): 4 MFMAs per stream per wave on the wave's activation tile; the result lands in the state
//     layout; the four waves' partial results meet in LDS (one ds_write_b128 / four ds_read_b128 per stream);
//   * Runge-Kutta bookkeeping, error norms, I/O: 5 components per lane instead of 17, 16-byte loads and stores.
// Hidden layers, gates, controller and grid reduction are those of dopri5_split_kernel.
#include "cnf_split.h"

namespace cnf {
namespace {

constexpr int kD = 16;
constexpr int kC = 5;   // components a lane integrates: its 4 dimensions + logp

struct alignas(16) Wave16Lds {
    float gate[3][2][kH];
    float gate4[2][16];
    float kst[7][kC][64];   // Runge-Kutta slopes, one column per lane
};

// forward solve: hidden activations travel as bf16 pieces (cnf_split.h, publish_b16 / hidden_b16)
struct alignas(16) Xchg16F {
    uint32_t xb16[2][2][3][2][64][4];   // [buffer][stream][piece][k-step][lane][tile & 1][2]
    float part[kNW][2][64][4];
};

__device__ __forceinline__ f32x4 sum4v(const f32x4& a, const f32x4& b, const f32x4& c, const f32x4& d) { return ((a + b) + c) + d; }

template <class NET, class WS>
__device__ __forceinline__ void gates16(const NET& n, WS& ws, int lane, float t) {
#pragma unroll
    for (int l = 0; l < 3; ++l) {
        const float g = sigmoid_f(fmaf(n.vec[l][2][lane], t, n.vec[l][3][lane]));
        ws.gate[l][0][lane] = g;
        ws.gate[l][1][lane] = fmaf(g, n.vec[l][0][lane], n.vec[l][1][lane] * t);
    }
    if (lane < 16) {
        const float g = sigmoid_f(fmaf(n.vec4[2][lane], t, n.vec4[3][lane]));
        ws.gate4[0][lane] = g;
        ws.gate4[1][lane] = fmaf(g, n.vec4[0][lane], n.vec4[1][lane] * t);
    }
    wave_lds_fence();
}

// f(t, z) and e^T (df/dz) e for the 16 rows of the workgroup; z4 / e4 / f4: this lane's 4 components.
// w1img[r] = W1[16 wave + (lane & 15)][4 (lane >> 4) + r],  w4img[r] = W4[lane & 15][16 wave + 4 (lane >> 4) + r].
__device__ __forceinline__ void field_fwd16(const NetLds<kD>& n, const SplitWeightsB16& wb, Wave16Lds& ws, Xchg16F& xc,
                                            const f32x4& w1img, const f32x4& w4img, int lane, int wave, float t,
                                            const f32x4& z4, const f32x4& e4, f32x4& f4, float& div) {
    const int q = lane >> 4;
    const int fo = 16 * wave + 4 * q;
    gates16(n, ws, lane, t);
    f32x4 act[2];
    {
        f32x4 a = {0.f, 0.f, 0.f, 0.f}, au = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            a = mfma16(w1img[r], z4[r], a);
            au = mfma16(w1img[r], e4[r], au);
        }
        const f32x4 g = *reinterpret_cast<const f32x4*>(&ws.gate[0][0][fo]);
        const f32x4 c = *reinterpret_cast<const f32x4*>(&ws.gate[0][1][fo]);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float x = act_f(fmaf(g[r], a[r], c[r]));
            act[0][r] = x;
            act[1][r] = act_d(x) * g[r] * au[r];
        }
    }
#pragma unroll
    for (int l = 0; l < 2; ++l) {
        publish_b16<2>(xc.xb16[l], act, wave, lane);
        __syncthreads();
        f32x4 acc[2];
        hidden_b16<2>(xc.xb16[l], wb, l, lane, acc);
        const f32x4 g = *reinterpret_cast<const f32x4*>(&ws.gate[l + 1][0][fo]);
        const f32x4 c = *reinterpret_cast<const f32x4*>(&ws.gate[l + 1][1][fo]);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float x = act_f(fmaf(g[r], acc[0][r], c[r]));
            act[0][r] = x;
            act[1][r] = act_d(x) * g[r] * acc[1][r];
        }
    }
    // layer 4: this wave's 16 input features -> partial sums of all 16 outputs, in the state layout
    {
        f32x4 pf = {0.f, 0.f, 0.f, 0.f}, pd = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            pf = mfma16(w4img[r], act[0][r], pf);
            pd = mfma16(w4img[r], act[1][r], pd);
        }
        *reinterpret_cast<f32x4*>(&xc.part[wave][0][lane][0]) = pf;
        *reinterpret_cast<f32x4*>(&xc.part[wave][1][lane][0]) = pd;
    }
    __syncthreads();
    const f32x4 u = sum4v(*reinterpret_cast<const f32x4*>(&xc.part[0][0][lane][0]), *reinterpret_cast<const f32x4*>(&xc.part[1][0][lane][0]),
                          *reinterpret_cast<const f32x4*>(&xc.part[2][0][lane][0]), *reinterpret_cast<const f32x4*>(&xc.part[3][0][lane][0]));
    const f32x4 U = sum4v(*reinterpret_cast<const f32x4*>(&xc.part[0][1][lane][0]), *reinterpret_cast<const f32x4*>(&xc.part[1][1][lane][0]),
                          *reinterpret_cast<const f32x4*>(&xc.part[2][1][lane][0]), *reinterpret_cast<const f32x4*>(&xc.part[3][1][lane][0]));
    const f32x4 g4 = *reinterpret_cast<const f32x4*>(&ws.gate4[0][4 * q]);
    const f32x4 c4 = *reinterpret_cast<const f32x4*>(&ws.gate4[1][4 * q]);
    float dpart = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        f4[r] = fmaf(g4[r], u[r], c4[r]);
        dpart = fmaf(g4[r] * U[r], e4[r], dpart);
    }
    div = xq_sum(dpart);
}

// ---- forward solve: the controller of dopri5_split_kernel on 5 components per lane --------------------------------
__global__ __launch_bounds__(kThreads) void dopri5_split16_kernel(SolveArgs a) {
    const float res = a.residual ? 1.f : 0.f;   // CNF_DIV_RESIDUAL: dynamics f(z) - z
    __shared__ NetLds<kD> net;
    __shared__ Wave16Lds wl[kNW];
    __shared__ Xchg16F xc;
    __shared__ RedLds red;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, q = lane >> 4, j = lane & 15;
    if (tid == 0) red.status = 0;
    stage_net(net, a.params, tid);
    zero_slopes(wl, kNW, tid);
    const ParamLayout L = make_layout(kD, kH);
    const f32x4 w1img = *reinterpret_cast<const f32x4*>(a.params + L.w(0) + (16 * w + j) * kD + 4 * q);
    const f32x4 w4img = *reinterpret_cast<const f32x4*>(a.params + L.w(3) + j * kH + 16 * w + 4 * q);
    __syncthreads();
    SplitWeightsB16 wb;
    wb.load(net, w, lane);
    const long row = (long)blockIdx.x * kRows + j;
    // every wave carries the same 16 rows: wave 0 reports them, every quarter its own 4 dimensions, quarter 0 logp
    const bool cz = row < a.rows && w == 0;
    const bool cl = cz && q == 0;
    const long rr = row < a.rows ? row : a.rows - 1;
    Wave16Lds& ws = wl[w];

    GridSync gs{dbg_gran(a, __LINE__), (int)gridDim.x, 0u};
    const double nl = (double)a.rows, nz = nl * kD;

    float y0[kC], y1[kC], yi[kC];
    const f32x4 e4 = *reinterpret_cast<const f32x4*>(a.e + rr * kD + 4 * q);
    {
        const f32x4 x4 = *reinterpret_cast<const f32x4*>(a.x + rr * kD + 4 * q);
#pragma unroll
        for (int r = 0; r < 4; ++r) y0[r] = x4[r];
    }
    y0[4] = a.logp0 ? a.logp0[rr] : 0.f;
#pragma unroll
    for (int c = 0; c < kC; ++c) yi[c] = y0[c];

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
    float sc[kC];

    if (a.after) status = a.after->status;   // CnfSolverOpts.after: the solve whose output this one consumes failed
    for (; status == 0;) {
        f32x4 z4, f4;
        float div, k[kC];
        bool restart = false;
#pragma unroll
        for (int r = 0; r < 4; ++r) z4[r] = yi[r];
        field_fwd16(net, wb, ws, xc, w1img, w4img, lane, w, sgn * ti, z4, e4, f4, div);
#pragma unroll
        for (int r = 0; r < 4; ++r) f4[r] = fmaf(-res, z4[r], f4[r]);   // residual dynamics f(z) - z (res = 0 or 1)
        div -= res * (float)kD;
        ++nfe;
#pragma unroll
        for (int r = 0; r < 4; ++r) k[r] = sgn * f4[r];
        k[4] = -sgn * div;

        if (phase == PH_F0) {
            double S[4] = {0, 0, 0, 0};
#pragma unroll
            for (int c = 0; c < kC; ++c) {
                ws.kst[0][c][lane] = k[c];
                sc[c] = atol + fabsf(y0[c]) * rtol;
                const float u = y0[c] / sc[c], v = k[c] / sc[c];
                if (c < 4 ? cz : cl) {
                    S[c < 4 ? 0 : 1] += (double)u * u;
                    S[c < 4 ? 2 : 3] += (double)v * v;
                }
            }
            if (!wg_allreduce4<true>(gs, red, S, w, lane)) { status = CNF_ERR_SYNC_TIMEOUT; break; }
            const float d0z = (float)sqrt(S[0] / nz), d0l = (float)sqrt(S[1] / nl);
            d1z = (float)sqrt(S[2] / nz);
            d1l = (float)sqrt(S[3] / nl);
            if (pymax(d0z, d0l) < 1e-5f || pymax(d1z, d1l) < 1e-5f) {
                h0 = 1e-6f;
            } else {
                h0 = 0.01f * pymax(d0z / d1z, d0l / d1l);
            }
#pragma unroll
            for (int c = 0; c < kC; ++c) yi[c] = y0[c] + h0 * k[c];
            ti = t0f + h0;
            phase = PH_F1;
            continue;
        }
        if (phase == PH_F1) {
            double S[4] = {0, 0, 0, 0};
#pragma unroll
            for (int c = 0; c < kC; ++c) {
                const float u = (k[c] - ws.kst[0][c][lane]) / sc[c];
                if (c < 4 ? cz : cl) S[c < 4 ? 0 : 1] += (double)u * u;
            }
            if (!wg_allreduce4<true>(gs, red, S, w, lane)) { status = CNF_ERR_SYNC_TIMEOUT; break; }
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
            for (int c = 0; c < kC; ++c) ws.kst[stage][c][lane] = k[c];
            if (stage < 6) {
                ++stage;
            } else {
                double S[4] = {0, 0, 0, 0};
#pragma unroll
                for (int c = 0; c < kC; ++c) {
                    y1[c] = yi[c];
                    float err = 0.f;
#pragma unroll
                    for (int s = 0; s < 7; ++s) err += (dtf * kCErr[s]) * ws.kst[s][c][lane];
                    const float tol = atol + rtol * fmaxf(fabsf(y0[c]), fabsf(y1[c]));
                    const float u = err / tol;
                    if (c < 4 ? cz : cl) {
                        S[c < 4 ? 0 : 1] += (double)u * u;
                        if (!isfinite(y0[c])) S[2] += 1.0;
                    }
                }
                if (!wg_allreduce4<true>(gs, red, S, w, lane)) { status = CNF_ERR_SYNC_TIMEOUT; break; }
                if (S[2] > 0.0) { status = CNF_ERR_NONFINITE; break; }
                const float rz = (float)(S[0] / nz), rl = (float)(S[1] / nl);
                const bool accept = (rz <= 1.f) && (rl <= 1.f);
                if (accept) {
                    ++nacc;
                    if (!(tau_end > tau + dt)) break;
#pragma unroll
                    for (int c = 0; c < kC; ++c) {
                        y0[c] = y1[c];
                        ws.kst[0][c][lane] = k[c];
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
        for (int c = 0; c < kC; ++c) {
            float acc = 0.f;
#pragma unroll
            for (int s = 0; s < 6; ++s) acc += (dtf * bc_[s]) * ws.kst[s][c][lane];   // zero weights past the current stage
            yi[c] = y0[c] + acc;
        }
    }

    if (status == 0) {
        const float ta = (float)tau, tb = (float)(tau + dt), te = (float)tau_end;
        const float x = (te - ta) / (tb - ta);
        const float x2 = x * x, x3 = x2 * x, x4 = x3 * x;
#pragma unroll
        for (int c = 0; c < kC; ++c) {
            float mid = 0.f;
#pragma unroll
            for (int s = 0; s < 7; ++s) mid += (dtf * kCMid[s]) * ws.kst[s][c][lane];
            mid = y0[c] + mid;
            const float f0 = ws.kst[0][c][lane], f1 = ws.kst[6][c][lane], u = y0[c], v = y1[c];
            const float ca = (-2 * dtf) * f0 + (2 * dtf) * f1 + -8 * u + -8 * v + 16 * mid;
            const float cb = (5 * dtf) * f0 + (-3 * dtf) * f1 + 18 * u + 14 * v + -32 * mid;
            const float cc = (-4 * dtf) * f0 + dtf * f1 + -11 * u + -5 * v + 16 * mid;
            const float cd = dtf * f0;
            y1[c] = ca * x4 + cb * x3 + cc * x2 + cd * x + u;
        }
        if (cz) {
            f32x4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = y1[r];
            *reinterpret_cast<f32x4*>(a.z_out + row * kD + 4 * q) = o;
        }
        if (cl) a.logp_out[row] = y1[4];
    }
    grid_cleanup(gs, red, tid);
    write_stats_global(a.stats, status, nfe, nacc, nrej, nfe, (float)dt, a.rows, tid);
}


// =====================================================================================================================
// adjoint (mathematics of aug_eval_split, cnf_split.hip; state layout as above)
// =====================================================================================================================
// The adjoint's four hidden-layer products per evaluation (two forward, two transposed) and its outer products run on
// the bf16 pipe with three-piece operands, as at d <= 2 (kBwdB16, cnf_split.hip): forward weight pieces in 48 registers per
// lane, transposed pieces in a 48 KB LDS image, activations and cotangents published as pieces (cnf_split.h).
struct alignas(16) Net16Lds {   // the vector parameters; the matrices live in registers (first / last layer, W pieces) and in Bwd16Lds
    float vec[3][4][kH];
    float vec4[4][16];
};
// the forward sweep's published activations (xa[l] = input of hidden layer l) stay until the end of the evaluation: they
// are the B operands of that layer's outer products; the cotangents get buffers of their own
struct alignas(16) Xchg16B {
    uint32_t xa[2][2][3][2][64][4];     // [layer][stream][piece][k-step][lane][tile & 1][2]
    uint32_t xcot[2][2][3][2][64][4];   // pre-activation cotangents of hidden layer lc
    float part[kNW][2][64][4];          // layer-4 partial results of every wave: [wave][stream][lane][r]
};

__device__ inline void stage_net16(Net16Lds& n, const float* __restrict__ p, int tid) {
    const ParamLayout L = make_layout(kD, kH);
    for (int idx = tid; idx < 3 * kH; idx += kThreads) {
        const int l = idx / kH, i = idx % kH;
        n.vec[l][0][i] = p[L.b(l) + i];
        n.vec[l][1][i] = p[L.hb(l) + i];
        n.vec[l][2][i] = p[L.wg(l) + i];
        n.vec[l][3][i] = p[L.bg(l) + i];
    }
    if (tid < 16) {
        n.vec4[0][tid] = p[L.b(3) + tid];
        n.vec4[1][tid] = p[L.hb(3) + tid];
        n.vec4[2][tid] = p[L.wg(3) + tid];
        n.vec4[3][tid] = p[L.bg(3) + tid];
    }
}

constexpr int kCA = 9;                       // components a lane integrates: 4 of z, logp, 4 of a_z
constexpr int kCE = 5;                       // of which error-controlled: z and logp
constexpr int kSmV = 3 * 4 * kH;             // vector parameters of the three hidden-output layers
constexpr int kSm16 = kSmV + 4 * kD;         // + those of the last layer: [kind][dimension]

struct alignas(16) Bwd16Lds {
    WimgT16 wimgT16;                    // transposed pieces of the two 64 x 64 matrices (stage_wimgT16, cnf_split.h)
    float trA[kNW][2][16 * kRows];      // per wave: transposed, weighted first-layer cotangents of its own 16 features (dW1's A operand)
    float tX[kNW][2][16 * kRows];       // per wave: its tile of the last hidden layer's output, transposed (dW4's B operand)
    float tU[2][16 * kRows];            // weighted cotangents of the 16 outputs, transposed (dW4's A operand)
    float tZ[2][16 * kRows];            // [0] z, [1] e, transposed (dW1's B operands)
    float small[4][kSm16];              // d/dtheta of the vector parameters: [0] weighted sum of the attempt, [1], [2] the
                                        // unweighted sums of the first / seventh stage (FSAL, kFsalTheta in cnf_split.hip),
                                        // [3] the running total over accepted steps (two-workgroups-per-tile launches)
    float wq[8];
    // one copy for the four waves (they hold identical values; readers and the next evaluation's writers are
    // separated by workgroup barriers), as are tU and tZ
    float gate[3][2][kH];
    float gate4[2][16];
    float kst[7][kCA][64];
};

// One augmented evaluation.  Cotangents: -a_z on f, a_logp on the divergence estimate.  Outputs f4, div, vz4 =
// (df/dz)^T(-a_z) + d(a_logp div)/dz (this lane's 4 components); with_theta: adds w * d/dtheta into acc (the two
// 64 x 64 matrices), gw1 (W1 rows 16 wave + 4q + r, column j), gw4 (W4 rows 4q + r, column 16 wave + j), sb.small.
// mid(f4, div): called by all threads when the forward half is done (the solve kernel starts its grid reduction there, as
// dopri5_bwd_split_kernel does: cnf_split.hip)
template <class Mid>
__device__ __forceinline__ void aug_eval16(const Net16Lds& n, const SplitWeightsB16& wb, Bwd16Lds& sb, Xchg16B& xc, const f32x4& w1img, const f32x4& w4img,
                                           const f32x4& w1Timg, const f32x4& w4Timg, int lane, int wave, float t,
                                           const f32x4& z4, const f32x4& e4, const f32x4& az4, float al, float w,
                                           bool with_theta, f32x4& f4, float& div, f32x4& vz4, f32x4 (&dw2)[4],
                                           f32x4 (&dw3)[4], f32x4& gw1, f32x4& gw4, int part, int small_base, Mid&& mid) {
    // dw2 / dw3: sums for the two 64 x 64 matrices; gw1 / gw4: for W1 / W4.  In the spread kernel a workgroup owns one of
    // each pair and the caller passes the same registers for both (the unowned one is never touched: th_hi / th_lo)
    float* const small = &sb.small[0][0] + small_base;
    // part (CNF_FLAG_SPREAD, kSpreadNote in cnf_split.hip): 0 = all d/dtheta sums, 1 = W3, layer-3 vectors, W4, layer-4
    // vectors, 2 = W1, layer-1/2 vectors, W2
    const bool th_hi = with_theta && part != 2, th_lo = with_theta && part != 1;
    const int q = lane >> 4, j = lane & 15, qq = j & 3;
    const int fo = 16 * wave + 4 * q;
    const int pos = (j & 3) * 4 + (j >> 2);
    gates16(n, sb, lane, t);
    const bool b0 = (j & 1) != 0, b1 = (j & 2) != 0;
    const int kind = j >> 2;
    const float km0 = kind == 0 ? 1.f : 0.f, km1t = kind == 1 ? t : 0.f, km23 = kind == 2 ? t : (kind == 3 ? 1.f : 0.f);

    // ------------------------------------------------------------------ forward with saves (own tile)
    f32x4 xs[3], ub[3], U[3], act[2];
    {
        f32x4 a = {0.f, 0.f, 0.f, 0.f}, au = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            a = mfma16(w1img[r], z4[r], a);
            au = mfma16(w1img[r], e4[r], au);
        }
        const f32x4 g = *reinterpret_cast<const f32x4*>(&sb.gate[0][0][fo]);
        const f32x4 c = *reinterpret_cast<const f32x4*>(&sb.gate[0][1][fo]);
        const f32x4 b = *reinterpret_cast<const f32x4*>(&n.vec[0][0][fo]);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float x = act_f(fmaf(g[r], a[r], c[r]));
            xs[0][r] = x;
            ub[0][r] = a[r] + b[r];
            U[0][r] = au[r];
            act[0][r] = x;
            act[1][r] = act_d(x) * g[r] * au[r];
        }
    }
    if (th_lo) {
#pragma unroll
        for (int r = 0; r < 4; ++r) sb.tZ[0][(4 * q + r) * kRows + pos] = z4[r];
    }
#pragma unroll
    for (int l = 0; l < 2; ++l) {
        publish_b16<2>(xc.xa[l], act, wave, lane);
        __syncthreads();
        f32x4 a4[2];
        hidden_b16<2>(xc.xa[l], wb, l, lane, a4);
        const f32x4 g = *reinterpret_cast<const f32x4*>(&sb.gate[l + 1][0][fo]);
        const f32x4 c = *reinterpret_cast<const f32x4*>(&sb.gate[l + 1][1][fo]);
        const f32x4 b = *reinterpret_cast<const f32x4*>(&n.vec[l + 1][0][fo]);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float x = act_f(fmaf(g[r], a4[0][r], c[r]));
            xs[l + 1][r] = x;
            ub[l + 1][r] = a4[0][r] + b[r];
            U[l + 1][r] = a4[1][r];
            act[0][r] = x;
            act[1][r] = act_d(x) * g[r] * a4[1][r];
        }
    }
    {
        f32x4 pf = {0.f, 0.f, 0.f, 0.f}, pd = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            pf = mfma16(w4img[r], act[0][r], pf);
            pd = mfma16(w4img[r], act[1][r], pd);
        }
        *reinterpret_cast<f32x4*>(&xc.part[wave][0][lane][0]) = pf;
        *reinterpret_cast<f32x4*>(&xc.part[wave][1][lane][0]) = pd;
    }
    __syncthreads();
    const f32x4 u4 = sum4v(*reinterpret_cast<const f32x4*>(&xc.part[0][0][lane][0]), *reinterpret_cast<const f32x4*>(&xc.part[1][0][lane][0]),
                           *reinterpret_cast<const f32x4*>(&xc.part[2][0][lane][0]), *reinterpret_cast<const f32x4*>(&xc.part[3][0][lane][0]));
    const f32x4 U4 = sum4v(*reinterpret_cast<const f32x4*>(&xc.part[0][1][lane][0]), *reinterpret_cast<const f32x4*>(&xc.part[1][1][lane][0]),
                           *reinterpret_cast<const f32x4*>(&xc.part[2][1][lane][0]), *reinterpret_cast<const f32x4*>(&xc.part[3][1][lane][0]));
    const f32x4 g4 = *reinterpret_cast<const f32x4*>(&sb.gate4[0][4 * q]);
    const f32x4 c4 = *reinterpret_cast<const f32x4*>(&sb.gate4[1][4 * q]);
    {
        float dpart = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            f4[r] = fmaf(g4[r], u4[r], c4[r]);
            dpart = fmaf(g4[r] * U4[r], e4[r], dpart);
        }
        div = xq_sum(dpart);
    }
    mid(f4, div);

    // ------------------------------------------------------------------ reverse sweep
    f32x4 cot[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    {
        f32x4 lf, ubar, Ubar;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            lf[r] = -az4[r];
            ubar[r] = g4[r] * lf[r];
            Ubar[r] = g4[r] * al * e4[r];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            cot[0] = mfma16(w4Timg[r], ubar[r], cot[0]);
            cot[1] = mfma16(w4Timg[r], Ubar[r], cot[1]);
        }
        if (th_hi) {
            // dW4[m][feature] = sum_rows (ubar_m x + Ubar_m X): K = the 16 rows, both operands transposed through LDS
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                sb.tU[0][(4 * q + r) * kRows + pos] = w * ubar[r];
                sb.tU[1][(4 * q + r) * kRows + pos] = w * Ubar[r];
                sb.tX[wave][0][(4 * q + r) * kRows + pos] = act[0][r];
                sb.tX[wave][1][(4 * q + r) * kRows + pos] = act[1][r];
            }
            wave_lds_fence();
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const f32x4 a4 = *reinterpret_cast<const f32x4*>(&sb.tU[s][j * kRows + 4 * q]);
                const f32x4 b4 = *reinterpret_cast<const f32x4*>(&sb.tX[wave][s][j * kRows + 4 * q]);
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) gw4 = mfma16(a4[ks], b4[ks], gw4);
            }
            // bias, hyper-bias, gate weight, gate bias of the last layer: lane j < 4 of wave 0 owns kind j
            const f32x4 bp = *reinterpret_cast<const f32x4*>(&n.vec4[0][4 * q]);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float s4 = row16_allsum_s(lf[r]);
                const float gb = row16_allsum_s(lf[r] * (u4[r] + bp[r]) + al * e4[r] * U4[r]);
                const float gg = g4[r] * (1.0f - g4[r]);
                const float val = j < 2 ? s4 : gb;
                const float fac = j == 0 ? g4[r] : (j == 1 ? t : (j == 2 ? t * gg : gg));
                if (wave == 0 && j < 4) lds_add(&small[kSmV + j * kD + 4 * q + r], w * fac * val);
            }
        }
    }

    // unweighted outer products of this evaluation on the bf16 pipe (operands read transposed from the piece buffers,
    // outer_b16 in cnf_split.h), then one fused multiply-add per accumulator register with the stage's quadrature weight
    auto issue_dw = [&](int lc) {
        f32x4 tmp[4];
        outer_b16<2>(xc.xcot[lc], xc.xa[lc], wave, lane, tmp);
#pragma unroll
        for (int Ti = 0; Ti < 4; ++Ti) {
            if (lc == 0) dw2[Ti] += w * tmp[Ti];
            else dw3[Ti] += w * tmp[Ti];
        }
    };

#pragma unroll
    for (int l = 1; l >= -1; --l) {
        const int L = l + 1;
        const f32x4 g = *reinterpret_cast<const f32x4*>(&sb.gate[L][0][fo]);
        float s1[4], s2[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float x = xs[L][r];
            const float dx = act_d(x);
            const float Uk = U[L][r];
            const float Xb = cot[1][r];
            float pre = dx * cot[0][r];
            pre = fmaf(Xb * (g[r] * Uk), act_dd(x, dx), pre);
            const float Pb = dx * Xb;
            cot[1][r] = g[r] * Pb;
            s1[r] = pre;
            s2[r] = fmaf(pre, ub[L][r], Pb * Uk);
            cot[0][r] = g[r] * pre;
        }
        if (l < 1 && (l == 0 ? th_hi : th_lo)) issue_dw(l + 1);   // the previous layer's outer products, next to this vector work
        if (L == 2 ? th_hi : th_lo) {
            float st, gt;
            reduce8(s1, s2, b0, b1, st, gt);
            const float gs = b1 ? (b0 ? g[3] : g[2]) : (b0 ? g[1] : g[0]);
            const float gg = gs * (1.0f - gs);
            const float fac = fmaf(gg, km23, fmaf(km0, gs, km1t));
            lds_add(&small[L * 4 * kH + kind * kH + fo + qq], w * fac * (kind < 2 ? st : gt));
        }
        if (l >= 0) {
            const int lc = l < 0 ? 0 : l;
            publish_b16<2>(xc.xcot[lc], cot, wave, lane);   // the Wt product's B operand and the outer products' A operand
            __syncthreads();
            f32x4 o4[2];
            hidden_b16_lds<2>(xc.xcot[lc], sb.wimgT16[lc][wave], lane, o4);
            cot[0] = o4[0];
            cot[1] = o4[1];
        } else {
            // (df/dz)^T: this wave's 16 features of the first layer's cotangent -> partial sums of all 16 components
            f32x4 pz = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int r = 0; r < 4; ++r) pz = mfma16(w1Timg[r], cot[0][r], pz);
            *reinterpret_cast<f32x4*>(&xc.part[wave][0][lane][0]) = pz;
            if (th_lo) {
                // dW1[feature][m] = sum_rows (cot z_m + tangent cotangent e_m)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    sb.trA[wave][0][(4 * q + r) * kRows + pos] = w * cot[0][r];
                    sb.trA[wave][1][(4 * q + r) * kRows + pos] = w * cot[1][r];
                }
                wave_lds_fence();
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const f32x4 a4 = *reinterpret_cast<const f32x4*>(&sb.trA[wave][s][j * kRows + 4 * q]);
                    const f32x4 b4 = *reinterpret_cast<const f32x4*>(&sb.tZ[s][j * kRows + 4 * q]);
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) gw1 = mfma16(a4[ks], b4[ks], gw1);
                }
            }
            __syncthreads();
            vz4 = sum4v(*reinterpret_cast<const f32x4*>(&xc.part[0][0][lane][0]), *reinterpret_cast<const f32x4*>(&xc.part[1][0][lane][0]),
                        *reinterpret_cast<const f32x4*>(&xc.part[2][0][lane][0]), *reinterpret_cast<const f32x4*>(&xc.part[3][0][lane][0]));
        }
    }
}

// accepted step: the workgroup's sums go into its slab (stored, not added, by the first one: no zeroing pass needed)
// Sums of one evaluation (or of one attempt): the two 64 x 64 matrices, this lane's entries of dW1 and dW4.  HALF (the
// spread kernel): a workgroup keeps ONE matrix and ONE of dW1 / dW4 (part 1: W3 and W4, part 2: W2 and W1), which is what
// lets three such sets (the attempt's sums and the two FSAL buffers) fit the register file.
template <bool HALF>
struct Acc16 {
    f32x4 m[HALF ? 1 : 2][4];
    f32x4 g[HALF ? 1 : 2];
    __device__ __forceinline__ void zero() {
#pragma unroll
        for (int l = 0; l < (HALF ? 1 : 2); ++l) {
#pragma unroll
            for (int b = 0; b < 4; ++b) m[l][b] = f32x4{0.f, 0.f, 0.f, 0.f};
            g[l] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    __device__ __forceinline__ f32x4 (&dw(int l))[4] { return m[HALF ? 0 : l]; }
    __device__ __forceinline__ const f32x4 (&dw(int l) const)[4] { return m[HALF ? 0 : l]; }
    __device__ __forceinline__ f32x4& thin(int which) { return g[HALF ? 0 : which]; }   // 0: dW1, 1: dW4
    __device__ __forceinline__ const f32x4& thin(int which) const { return g[HALF ? 0 : which]; }
};

template <bool HALF>
__device__ __forceinline__ void swap16(Acc16<HALF>& a, Acc16<HALF>& b) {
    const Acc16<HALF> t = a;
    a = b;
    b = t;
}

// accepted step: the workgroup's sums go into its slab (stored, not added, by the first one: no zeroing pass needed).
// FSAL: step sums + w_first * G(first stage) + w_last * G(seventh stage), see commit_step_fsal in cnf_split.hip
template <bool HALF>
__device__ inline void commit16(Bwd16Lds& sb, const Acc16<HALF>& A, const Acc16<HALF>& F, const Acc16<HALF>& G, int i_first,
                                int i_last, float w_first, float w_last, float* __restrict__ slab_wg, int tid, bool first,
                                int part) {
    const ParamLayout L = make_layout(kD, kH);
    const int lane = tid & 63, w = tid >> 6, q = lane >> 4, j = lane & 15;
    const int i_lo = part == 1 ? 8 * kH : 0, i_hi = part == 2 ? 8 * kH : kSm16;   // layer-1/2 vectors | layer-3/4 vectors
    __syncthreads();
    for (int i = i_lo + tid; i < i_hi; i += kThreads) {
        const int canon = i < kSmV ? L.b(i / (4 * kH)) + i % (4 * kH) : L.b(3) + (i - kSmV);
        float* dst = &slab_wg[canon];
        float v = sb.small[0][i];
        if (HALF) v = fmaf(w_last, sb.small[i_last][i], fmaf(w_first, sb.small[i_first][i], v));
        *dst = (first ? 0.f : *dst) + v;
    }
#pragma unroll
    for (int l = 0; l < 2; ++l)
        if (part == 0 || part == 2 - l)
#pragma unroll
        for (int Ti = 0; Ti < 4; ++Ti)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float* dst = &slab_wg[L.w(1 + l) + (16 * w + 4 * q + r) * kH + 16 * Ti + j];
                float v = A.dw(l)[Ti][r];
                if (HALF) v = fmaf(w_last, G.dw(l)[Ti][r], fmaf(w_first, F.dw(l)[Ti][r], v));
                *dst = (first ? 0.f : *dst) + v;
            }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        if (part != 1) {
            float* d1 = &slab_wg[L.w(0) + (16 * w + 4 * q + r) * kD + j];
            float v = A.thin(0)[r];
            if (HALF) v = fmaf(w_last, G.thin(0)[r], fmaf(w_first, F.thin(0)[r], v));
            *d1 = (first ? 0.f : *d1) + v;
        }
        if (part != 2) {
            float* d4 = &slab_wg[L.w(3) + (4 * q + r) * kH + 16 * w + j];
            float v = A.thin(1)[r];
            if (HALF) v = fmaf(w_last, G.thin(1)[r], fmaf(w_first, F.thin(1)[r], v));
            *d4 = (first ? 0.f : *d4) + v;
        }
    }
    __syncthreads();
}

// Two-workgroups-per-tile launches keep RUNNING TOTALS over the accepted steps (a fourth half set of registers, a fourth
// LDS row) and store their half slab once, when the solve ends: no global read-modify-write between steps, as at d <= 2
// (commit_total / store_total, cnf_split.hip).  The totals add up in the order the read-modify-writes did: same bits.
__device__ inline void commit16_total(Bwd16Lds& sb, Acc16<true>& T, const Acc16<true>& A, const Acc16<true>& F, const Acc16<true>& G,
                                      int i_first, int i_last, float w_first, float w_last, int tid, int part) {
    const int i_lo = part == 1 ? 8 * kH : 0, i_hi = part == 2 ? 8 * kH : kSm16;   // layer-1/2 vectors | layer-3/4 vectors
    __syncthreads();
    for (int i = i_lo + tid; i < i_hi; i += kThreads)
        sb.small[3][i] += fmaf(w_last, sb.small[i_last][i], fmaf(w_first, sb.small[i_first][i], sb.small[0][i]));
#pragma unroll
    for (int Ti = 0; Ti < 4; ++Ti)
#pragma unroll
        for (int r = 0; r < 4; ++r) T.m[0][Ti][r] += fmaf(w_last, G.m[0][Ti][r], fmaf(w_first, F.m[0][Ti][r], A.m[0][Ti][r]));
#pragma unroll
    for (int r = 0; r < 4; ++r) T.g[0][r] += fmaf(w_last, G.g[0][r], fmaf(w_first, F.g[0][r], A.g[0][r]));
    __syncthreads();
}
__device__ inline void store16_total(const Bwd16Lds& sb, const Acc16<true>& T, float* __restrict__ slab_wg, int tid, int part) {
    const ParamLayout L = make_layout(kD, kH);
    const int lane = tid & 63, w = tid >> 6, q = lane >> 4, j = lane & 15;
    const int i_lo = part == 1 ? 8 * kH : 0, i_hi = part == 2 ? 8 * kH : kSm16;
    __syncthreads();
    for (int i = i_lo + tid; i < i_hi; i += kThreads) {
        const int canon = i < kSmV ? L.b(i / (4 * kH)) + i % (4 * kH) : L.b(3) + (i - kSmV);
        slab_wg[canon] = sb.small[3][i];
    }
    const int l = 2 - part;   // part 1 keeps W3 and W4, part 2 keeps W2 and W1
#pragma unroll
    for (int Ti = 0; Ti < 4; ++Ti)
#pragma unroll
        for (int r = 0; r < 4; ++r) slab_wg[L.w(1 + l) + (16 * w + 4 * q + r) * kH + 16 * Ti + j] = T.m[0][Ti][r];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        if (part == 2) slab_wg[L.w(0) + (16 * w + 4 * q + r) * kD + j] = T.g[0][r];
        else slab_wg[L.w(3) + (4 * q + r) * kH + 16 * w + j] = T.g[0][r];
    }
}

// SPREAD: launched with two workgroups per tile (a.spread); half accumulators and FSAL for the theta quadrature (six
// evaluations per attempt).  Otherwise one workgroup per tile keeps all sums and re-evaluates the first stage (seven).
template <bool SPREAD>
__global__ __launch_bounds__(kThreads) void dopri5_bwd_split16_kernel(SolveArgs a) {
    const float res = a.residual ? 1.f : 0.f;   // CNF_DIV_RESIDUAL: dynamics f(z) - z
    __shared__ Net16Lds net;
    __shared__ Bwd16Lds sb;
    __shared__ Xchg16B xc;
    __shared__ RedLds red;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, q = lane >> 4, j = lane & 15;
    const int pos = (j & 3) * 4 + (j >> 2);
    if (tid == 0) red.status = 0;
    stage_net16(net, a.params, tid);
    stage_wimgT16<kD>(sb.wimgT16, a.params, tid);
    SplitWeightsB16 wb;
    load_weights_b16_global<kD>(wb, a.params, w, lane);
    for (int i = tid; i < 7 * kCA * 64; i += kThreads) (&sb.kst[0][0][0])[i] = 0.f;
    for (int i = tid; i < 4 * kSm16; i += kThreads) (&sb.small[0][0])[i] = 0.f;
    const ParamLayout L = make_layout(kD, kH);
    const f32x4 w1img = *reinterpret_cast<const f32x4*>(a.params + L.w(0) + (16 * w + j) * kD + 4 * q);
    const f32x4 w4img = *reinterpret_cast<const f32x4*>(a.params + L.w(3) + j * kH + 16 * w + 4 * q);
    f32x4 w1Timg, w4Timg;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        w1Timg[r] = a.params[L.w(0) + (16 * w + 4 * q + r) * kD + j];
        w4Timg[r] = a.params[L.w(3) + (4 * q + r) * kH + 16 * w + j];
    }
    constexpr bool FSAL = SPREAD;
    const int tile = SPREAD ? (int)(blockIdx.x >> 1) : (int)blockIdx.x;   // kSpreadNote, cnf_split.hip
    const int part = SPREAD ? 1 + (int)(blockIdx.x & 1) : 0;
    const long row = (long)tile * kRows + j;
    const bool valid = row < a.rows;
    const bool cz = valid && w == 0 && part != 2;
    const bool cl = cz && q == 0;
    const long rr = valid ? row : a.rows - 1;
    float* slab = dbg_ws(a, a.slab + (size_t)blockIdx.x * L.total, sizeof(float) * L.total, __LINE__);

    GridSync gs{dbg_gran(a, __LINE__), (int)gridDim.x, 0u};
    const double nl = (double)a.rows, nz = nl * kD;

    // a.loss_out: NLL cotangents and the loss on the second grid reduction (see dopri5_bwd_split_kernel, cnf_split.hip)
    const bool nll = a.loss_out != nullptr;
    const float inv_rows = 1.0f / (float)a.rows;
    float nll_part = 0.f;   // this lane's four components of sum_d log N(z1_d)
    float y0[kCA], y1[kCA], yi[kCA];
    const f32x4 e4 = *reinterpret_cast<const f32x4*>(a.e + rr * kD + 4 * q);
    {
        const f32x4 x4 = *reinterpret_cast<const f32x4*>(a.x + rr * kD + 4 * q);
        const f32x4 g4 = nll ? x4 * inv_rows : *reinterpret_cast<const f32x4*>(a.grad_z1 + rr * kD + 4 * q);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            y0[r] = x4[r];
            y0[5 + r] = valid ? g4[r] : 0.f;
            sb.tZ[1][(4 * q + r) * kRows + pos] = e4[r];
            nll_part += -0.91893853320467274178f - x4[r] * x4[r] * 0.5f;
        }
    }
    y0[4] = a.logp0[rr];
    const float al = valid ? (nll ? inv_rows : a.grad_logp1[rr]) : 0.f;
#pragma unroll
    for (int c = 0; c < kCA; ++c) yi[c] = y0[c];
    __syncthreads();

    const float sgn = a.t0 >= a.t1 ? 1.f : -1.f;
    const double tau_end = (double)(sgn * a.t0);
    double tau = (double)(sgn * a.t1);
    double dt = 0.0;
    float dtf = 0.f, t0f = (float)tau, ti = t0f;
    const float rtol = a.rtol, atol = a.atol;

    enum { PH_F0 = 0, PH_F1 = 1, PH_STAGE = 2 };
    int phase = PH_F0, stage = 0;
    int nacc = 0, nrej = 0, status = 0, nrun = 0;
    float h0 = 0.f, d1z = 0.f, d1l = 0.f;
    float sc[kCE];   // scale of the initial-step heuristic (the solve's initial state)
#pragma unroll
    for (int c = 0; c < kCE; ++c) sc[c] = atol + fabsf(y0[c]) * rtol;
    Acc16<SPREAD> acc, gfirst, glast;   // acc: the attempt's weighted sums; gfirst / glast: unweighted sums of its first / seventh stage
    acc.zero();
    gfirst.zero();
    glast.zero();
    Acc16<true> total;             // SPREAD: the running total over accepted steps
    total.zero();
    int i_first = 1, i_last = 2;   // which of sb.small[1], sb.small[2] holds the first / seventh stage's sums

    if (a.after) status = a.after->status;   // CnfSolverOpts.after: the solve whose output this one consumes failed
    for (; status == 0;) {
        f32x4 z4, az4, f4, vz4;
        float div, k[kCA];
        bool restart = false;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            z4[r] = yi[r];
            az4[r] = yi[5 + r];
        }
        // which d/dtheta sum this evaluation feeds: the attempt's weighted quadrature, or an unweighted FSAL buffer
        const bool to_first = FSAL && phase == PH_F0;
        const bool to_last = FSAL && phase == PH_STAGE && stage == 6;
        const float wst = (to_first || to_last) ? 1.f : (phase == PH_STAGE ? sb.wq[stage] : 0.f);
        const bool with_theta = wst != 0.f;
        // one inlined copy of the evaluation, always accumulating into `acc`: an FSAL evaluation trades places with its
        // (zeroed) buffer before and after
        if (to_first) { gfirst.zero(); swap16(acc, gfirst); }
        if (to_last) { glast.zero(); swap16(acc, glast); }
        // the controller's sums over (z, logp) need f and div only: the grid reduction is published from the middle of the
        // evaluation and collected after the reverse sweep (see dopri5_bwd_split_kernel, cnf_split.hip); same sums, same bits
        auto control_sums = [&](const float (&kc)[kCE], double (&S)[4]) {
#pragma unroll
            for (int c = 0; c < kCE; ++c) {
                const bool cnt = c < 4 ? cz : cl;
                if (phase == PH_F0) {
                    const float u = y0[c] / sc[c], v = kc[c] / sc[c];
                    if (cnt) {
                        S[c < 4 ? 0 : 1] += (double)u * u;
                        S[c < 4 ? 2 : 3] += (double)v * v;
                    }
                } else if (phase == PH_F1) {
                    const float u = (kc[c] - sb.kst[0][c][lane]) / sc[c];
                    if (cnt) S[c < 4 ? 0 : 1] += (double)u * u;
                } else {
                    float err = 0.f;
#pragma unroll
                    for (int s = 0; s < 6; ++s) err += (dtf * kCErr[s]) * sb.kst[s][c][lane];
                    err += (dtf * kCErr[6]) * kc[c];
                    const float tol = atol + rtol * fmaxf(fabsf(y0[c]), fabsf(yi[c]));   // yi: the candidate y1
                    const float u = err / tol;
                    if (cnt) {
                        S[c < 4 ? 0 : 1] += (double)u * u;
                        if (!isfinite(y0[c])) S[2] += 1.0;
                    }
                }
            }
            if (phase == PH_F1 && nll) S[2] = (cz ? (double)nll_part : 0.0) - (cl ? (double)y0[4] : 0.0);
        };
        bool early = false;
        auto mid = [&](const f32x4& f_, float div_) {
            if (a.adj_all || (phase == PH_STAGE && stage != 6)) return;
            float kc[kCE];
#pragma unroll
            for (int r = 0; r < 4; ++r) kc[r] = sgn * fmaf(-res, z4[r], f_[r]);
            kc[4] = -sgn * (div_ - res * (float)kD);
            double S[4] = {0, 0, 0, 0};
            control_sums(kc, S);
            wg_reduce_publish<true>(gs, red, S, w, lane);
            early = true;
        };
        aug_eval16(net, wb, sb, xc, w1img, w4img, w1Timg, w4Timg, lane, w, sgn * ti, z4, e4, az4, al, sgn * wst, with_theta, f4, div,
                   vz4, acc.dw(0), acc.dw(1), acc.thin(0), acc.thin(1), part,
                   to_first ? i_first * kSm16 : (to_last ? i_last * kSm16 : 0), mid);
#pragma unroll
        for (int r = 0; r < 4; ++r) {   // residual dynamics f(z) - z (res = 0 or 1): d/dz gains -I
            f4[r] = fmaf(-res, z4[r], f4[r]);
            vz4[r] = fmaf(res, az4[r], vz4[r]);
        }
        div -= res * (float)kD;
        if (to_first) swap16(acc, gfirst);
        if (to_last) swap16(acc, glast);
        ++nrun;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            k[r] = sgn * f4[r];
            k[5 + r] = sgn * vz4[r];
        }
        k[4] = -sgn * div;

        if (phase == PH_F0) {
            double S[4] = {0, 0, 0, 0};
#pragma unroll
            for (int c = 0; c < kCA; ++c) sb.kst[0][c][lane] = k[c];
            if (early) {
                if (!wg_reduce_collect(gs, red, S, w, lane)) { status = CNF_ERR_SYNC_TIMEOUT; break; }
            } else {
                float kc[kCE];
#pragma unroll
                for (int c = 0; c < kCE; ++c) kc[c] = k[c];
                control_sums(kc, S);
                if (!wg_allreduce4<true>(gs, red, S, w, lane)) { status = CNF_ERR_SYNC_TIMEOUT; break; }
            }
            const float d0z = (float)sqrt(S[0] / nz), d0l = (float)sqrt(S[1] / nl);
            d1z = (float)sqrt(S[2] / nz);
            d1l = (float)sqrt(S[3] / nl);
            float d0a = 0.f, d1a = 0.f, d0al = 0.f;
            if (a.adj_all) {   // CNF_FLAG_ADJ_CONTROL_ALL (cnf_device.h): a_z and a_logp take part in the heuristic
                double S2[4] = {0, 0, 0, 0};
#pragma unroll
                for (int c = 5; c < kCA; ++c) {
                    const float s_ = atol + fabsf(y0[c]) * rtol;
                    const float u = y0[c] / s_, v = k[c] / s_;
                    if (cz) {
                        S2[0] += (double)u * u;
                        S2[1] += (double)v * v;
                    }
                }
                const float ua = al / (atol + fabsf(al) * rtol);
                if (cl) S2[2] += (double)ua * ua;
                if (!wg_allreduce4<true>(gs, red, S2, w, lane)) { status = CNF_ERR_SYNC_TIMEOUT; break; }
                d0a = (float)sqrt(S2[0] / nz);
                d1a = (float)sqrt(S2[1] / nz);
                d0al = (float)sqrt(S2[2] / nl);
                if (tid == 0) red.keep[0] = d1a;   // read after the next reduction's barriers
            }
            h0 = initial_h0(d0z, d0l, d1z, d1l, a.adj_all != 0, d0a, d1a, d0al);
            if (h0 < INFINITY) {
#pragma unroll
                for (int c = 0; c < kCA; ++c) yi[c] = y0[c] + h0 * k[c];
                ti = t0f + h0;
            }   // else (adj_all): the second evaluation runs at the initial state; initial_dt() ignores what it returns
            phase = PH_F1;
            continue;
        }
        if (phase == PH_F1) {
            double S[4] = {0, 0, 0, 0};
            if (early) {
                if (!wg_reduce_collect(gs, red, S, w, lane)) { status = CNF_ERR_SYNC_TIMEOUT; break; }
            } else {
                float kc[kCE];
#pragma unroll
                for (int c = 0; c < kCE; ++c) kc[c] = k[c];
                control_sums(kc, S);
                if (a.adj_all && cz) {
#pragma unroll
                    for (int c = 5; c < kCA; ++c) {
                        const float u = (k[c] - sb.kst[0][c][lane]) / (atol + fabsf(y0[c]) * rtol);
                        S[3] += (double)u * u;
                    }
                }
                if (!wg_allreduce4<true>(gs, red, S, w, lane)) { status = CNF_ERR_SYNC_TIMEOUT; break; }
            }
            if (nll && blockIdx.x == 0 && tid == 0) *a.loss_out = -((float)S[2] * inv_rows);
            const float d2z = (float)sqrt(S[0] / nz) / h0, d2l = (float)sqrt(S[1] / nl) / h0;
            const float d2a = (float)sqrt(S[3] / nz) / h0;
            dt = (double)initial_dt(h0, d1z, d1l, d2z, d2l, a.adj_all != 0, a.adj_all ? red.keep[0] : 0.f, d2a);
            restart = true;
        } else {
#pragma unroll
            for (int c = 0; c < kCA; ++c) sb.kst[stage][c][lane] = k[c];
            if (stage < 6) {
                ++stage;
            } else {
                double S[4] = {0, 0, 0, 0};
#pragma unroll
                for (int c = 0; c < kCA; ++c) y1[c] = yi[c];
                if (early) {
                    if (!wg_reduce_collect(gs, red, S, w, lane)) { status = CNF_ERR_SYNC_TIMEOUT; break; }
                } else {
                    float kc[kCE];
#pragma unroll
                    for (int c = 0; c < kCE; ++c) kc[c] = k[c];
                    control_sums(kc, S);
                    if (a.adj_all) {   // a_z under error control as well
#pragma unroll
                        for (int c = 5; c < kCA; ++c) {
                            float err = 0.f;
#pragma unroll
                            for (int s = 0; s < 7; ++s) err += (dtf * kCErr[s]) * sb.kst[s][c][lane];
                            const float u = err / (atol + rtol * fmaxf(fabsf(y0[c]), fabsf(y1[c])));
                            if (cz) S[3] += (double)u * u;
                        }
                    }
                    if (!wg_allreduce4<true>(gs, red, S, w, lane)) { status = CNF_ERR_SYNC_TIMEOUT; break; }
                }
                if (S[2] > 0.0) { status = CNF_ERR_NONFINITE; break; }
                const float rz = (float)(S[0] / nz), rl = (float)(S[1] / nl);
                const float ra = (float)(S[3] / nz);   // 0 unless CNF_FLAG_ADJ_CONTROL_ALL (a_logp's ratio is always 0)
                const bool accept = (rz <= 1.f) && (rl <= 1.f) && (ra <= 1.f);
                if (accept) {
                    ++nacc;
                    if constexpr (SPREAD) commit16_total(sb, total, acc, gfirst, glast, i_first, i_last, sb.wq[0], sb.wq[6], tid, part);
                    else commit16(sb, acc, gfirst, glast, i_first, i_last, sb.wq[0], sb.wq[6], slab, tid, nacc == 1, part);
                    if (!(tau_end > tau + dt)) break;
#pragma unroll
                    for (int c = 0; c < kCA; ++c) y0[c] = y1[c];
                    if (FSAL) {
#pragma unroll
                        for (int c = 0; c < kCA; ++c) sb.kst[0][c][lane] = k[c];   // the seventh stage is the next step's first
                        gfirst = glast;
                        const int t_ = i_first;
                        i_first = i_last;
                        i_last = t_;
                    }
                    tau = tau + dt;
                } else {
                    ++nrej;
                }
                dt = next_step_size(dt, pymax(pymax(rz, rl), ra));
                restart = true;
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
            __syncthreads();   // every wave is past its last read of small / wq of the previous attempt
            for (int i = tid; i < kSm16; i += kThreads) {
                sb.small[0][i] = 0.f;
                if (FSAL) sb.small[i_last][i] = 0.f;
            }
            if (tid < 7) {
                const int s = tid;
                float v;
                if (tau_end > tau + dt) {
                    v = s < 6 ? dtf * kBeta[5][s] : 0.f;
                } else {
                    const float ta = (float)tau, tb = (float)(tau + dt), te = (float)tau_end;
                    const float x = (te - ta) / (tb - ta);
                    const float x2 = x * x, x3 = x2 * x, x4 = x3 * x;
                    const float b = s < 6 ? kBeta[5][s] : 0.f;
                    const float cm = kCMid[s];
                    v = dtf * (x4 * (-8.f * b + 16.f * cm) + x3 * (14.f * b - 32.f * cm) + x2 * (-5.f * b + 16.f * cm));
                    if (s == 0) v += dtf * (-2.f * x4 + 5.f * x3 - 4.f * x2 + x);
                    if (s == 6) v += dtf * (2.f * x4 - 3.f * x3 + x2);
                }
                sb.wq[s] = v;
            }
            __syncthreads();
        }
        if (stage == 0) {
            ti = t0f;
#pragma unroll
            for (int c = 0; c < kCA; ++c) yi[c] = y0[c];
        } else {
            float alpha_, bc_[6];
            stage_coeffs(stage, alpha_, bc_);
            ti = t0f + alpha_ * dtf;
#pragma unroll
            for (int c = 0; c < kCA; ++c) {
                float s_ = 0.f;
#pragma unroll
                for (int s = 0; s < 6; ++s) s_ += (dtf * bc_[s]) * sb.kst[s][c][lane];   // zero weights past the current stage
                yi[c] = y0[c] + s_;
            }
        }
    }

    if (status == 0) {
        const float ta = (float)tau, tb = (float)(tau + dt), te = (float)tau_end;
        const float x = (te - ta) / (tb - ta);
        const float x2 = x * x, x3 = x2 * x, x4 = x3 * x;
        f32x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int c = 5 + r;
            float mid = 0.f;
#pragma unroll
            for (int s = 0; s < 7; ++s) mid += (dtf * kCMid[s]) * sb.kst[s][c][lane];
            mid = y0[c] + mid;
            const float f0 = sb.kst[0][c][lane], f1 = sb.kst[6][c][lane], u = y0[c], v = y1[c];
            const float ca = (-2 * dtf) * f0 + (2 * dtf) * f1 + -8 * u + -8 * v + 16 * mid;
            const float cb = (5 * dtf) * f0 + (-3 * dtf) * f1 + 18 * u + 14 * v + -32 * mid;
            const float cc = (-4 * dtf) * f0 + dtf * f1 + -11 * u + -5 * v + 16 * mid;
            const float cd = dtf * f0;
            o[r] = ca * x4 + cb * x3 + cc * x2 + cd * x + u;
        }
        if (cz) *reinterpret_cast<f32x4*>(a.grad_x + row * kD + 4 * q) = o;
        if (cl) a.grad_logp0[row] = al;
    }
    if constexpr (SPREAD) store16_total(sb, total, slab, tid, part);
    grid_cleanup(gs, red, tid);
    write_stats_global(a.stats, status, 2 + 6 * (nacc + nrej), nacc, nrej, nrun, (float)dt, a.rows, tid);
}

// ---- the adjoint on a fixed grid (euler / midpoint / rk4 3/8): what fixed_bwd_kernel (cnf_bwd.hip) is at d <= 2 -----------
// One workgroup per tile, all sums in one set of accumulators over ALL steps (there are no rejected attempts), stored once.
__global__ __launch_bounds__(kThreads) void fixed_bwd_split16_kernel(SolveArgs a) {
    const float res = a.residual ? 1.f : 0.f;   // CNF_DIV_RESIDUAL: dynamics f(z) - z
    __shared__ Net16Lds net;
    __shared__ Bwd16Lds sb;
    __shared__ Xchg16B xc;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, q = lane >> 4, j = lane & 15;
    const int pos = (j & 3) * 4 + (j >> 2);
    stage_net16(net, a.params, tid);
    stage_wimgT16<kD>(sb.wimgT16, a.params, tid);
    SplitWeightsB16 wb;
    load_weights_b16_global<kD>(wb, a.params, w, lane);
    for (int i = tid; i < 7 * kCA * 64; i += kThreads) (&sb.kst[0][0][0])[i] = 0.f;
    for (int i = tid; i < 4 * kSm16; i += kThreads) (&sb.small[0][0])[i] = 0.f;
    const ParamLayout L = make_layout(kD, kH);
    const f32x4 w1img = *reinterpret_cast<const f32x4*>(a.params + L.w(0) + (16 * w + j) * kD + 4 * q);
    const f32x4 w4img = *reinterpret_cast<const f32x4*>(a.params + L.w(3) + j * kH + 16 * w + 4 * q);
    f32x4 w1Timg, w4Timg;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        w1Timg[r] = a.params[L.w(0) + (16 * w + 4 * q + r) * kD + j];
        w4Timg[r] = a.params[L.w(3) + (4 * q + r) * kH + 16 * w + j];
    }
    const long row = (long)blockIdx.x * kRows + j;
    const bool valid = row < a.rows;
    const bool cz = valid && w == 0;
    const bool cl = cz && q == 0;
    const long rr = valid ? row : a.rows - 1;
    float* slab = dbg_ws(a, a.slab + (size_t)blockIdx.x * L.total, sizeof(float) * L.total, __LINE__);

    float y[kCA];
    const f32x4 e4 = *reinterpret_cast<const f32x4*>(a.e + rr * kD + 4 * q);
    {
        const f32x4 x4 = *reinterpret_cast<const f32x4*>(a.x + rr * kD + 4 * q);
        const f32x4 g4 = *reinterpret_cast<const f32x4*>(a.grad_z1 + rr * kD + 4 * q);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            y[r] = x4[r];
            y[5 + r] = valid ? g4[r] : 0.f;
            sb.tZ[1][(4 * q + r) * kRows + pos] = e4[r];
        }
    }
    y[4] = a.logp0[rr];
    const float al = valid ? a.grad_logp1[rr] : 0.f;
    __syncthreads();

    const int status = a.after ? a.after->status : 0;   // CnfSolverOpts.after: the solve whose output this one consumes failed
    const float sgn = a.t0 >= a.t1 ? 1.f : -1.f;
    const float tau0 = sgn * a.t1, tau1 = sgn * a.t0;   // the adjoint pass runs from t1 back to t0
    const int nst = a.fixed_kind == CNF_SOLVER_EULER ? 1 : (a.fixed_kind == CNF_SOLVER_MIDPOINT ? 2 : 4);
    Acc16<false> acc;
    acc.zero();
    int nrun = 0;
    const int n_steps = status == 0 ? a.n_grid : 0;
    for (int step = 0; step < n_steps; ++step) {
        const float ta = a.n_grid == 1 ? tau0 : __fadd_rn(__fmul_rn((float)step, a.step_size), tau0);
        const float tb = (step + 1 == a.n_grid) ? tau1 : __fadd_rn(__fmul_rn((float)(step + 1), a.step_size), tau0);
        const float dt = tb - ta;
        float k1[kCA], k2[kCA], k3[kCA], yi[kCA];
        float ti = ta;
#pragma unroll
        for (int c = 0; c < kCA; ++c) yi[c] = y[c];
        for (int st = 0; st < nst; ++st) {
            f32x4 z4, az4, f4, vz4;
            float div, k[kCA];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                z4[r] = yi[r];
                az4[r] = yi[5 + r];
            }
            float wq;   // weight of this stage's vjp_theta in the step's increment of a_theta
            if (a.fixed_kind == CNF_SOLVER_EULER) wq = dt;
            else if (a.fixed_kind == CNF_SOLVER_MIDPOINT) wq = st == 0 ? 0.f : dt;
            else wq = (st == 0 || st == 3) ? dt / 8 : 3 * (dt / 8);
            aug_eval16(net, wb, sb, xc, w1img, w4img, w1Timg, w4Timg, lane, w, sgn * ti, z4, e4, az4, al, sgn * wq, wq != 0.f, f4, div,
                       vz4, acc.dw(0), acc.dw(1), acc.thin(0), acc.thin(1), 0, 0, [](const f32x4&, float) {});
#pragma unroll
            for (int r = 0; r < 4; ++r) {   // residual dynamics f(z) - z (res = 0 or 1): d/dz gains -I
                f4[r] = fmaf(-res, z4[r], f4[r]);
                vz4[r] = fmaf(res, az4[r], vz4[r]);
            }
            div -= res * (float)kD;
            ++nrun;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                k[r] = sgn * f4[r];
                k[5 + r] = sgn * vz4[r];
            }
            k[4] = -sgn * div;
            if (a.fixed_kind == CNF_SOLVER_EULER) {
#pragma unroll
                for (int c = 0; c < kCA; ++c) y[c] = y[c] + dt * k[c];
            } else if (a.fixed_kind == CNF_SOLVER_MIDPOINT) {
                if (st == 0) {
                    ti = ta + dt / 2;
#pragma unroll
                    for (int c = 0; c < kCA; ++c) yi[c] = y[c] + k[c] * (dt / 2);
                } else {
#pragma unroll
                    for (int c = 0; c < kCA; ++c) y[c] = y[c] + dt * k[c];
                }
            } else if (st == 0) {
                ti = ta + dt / 3;
#pragma unroll
                for (int c = 0; c < kCA; ++c) { k1[c] = k[c]; yi[c] = y[c] + dt * k[c] / 3; }
            } else if (st == 1) {
                ti = ta + dt * 2 / 3;
#pragma unroll
                for (int c = 0; c < kCA; ++c) { k2[c] = k[c]; yi[c] = y[c] + dt * (k1[c] / -3 + k[c]); }
            } else if (st == 2) {
                ti = ta + dt;
#pragma unroll
                for (int c = 0; c < kCA; ++c) { k3[c] = k[c]; yi[c] = y[c] + dt * (k1[c] - k2[c] + k[c]); }
            } else {
#pragma unroll
                for (int c = 0; c < kCA; ++c) y[c] = y[c] + (k1[c] + 3 * k2[c] + 3 * k3[c] + k[c]) * (dt / 8);
            }
        }
    }
    if (status == 0) {
        const f32x4 o = {y[5], y[6], y[7], y[8]};
        if (cz) *reinterpret_cast<f32x4*>(a.grad_x + row * kD + 4 * q) = o;
        if (cl) a.grad_logp0[row] = al;
    }
    commit16<false>(sb, acc, acc, acc, 0, 0, 0.f, 0.f, slab, tid, true, 0);   // stored, not added: one commit per solve
    write_stats_global(a.stats, status, nst * a.n_grid, a.n_grid, 0, nrun, 0.f, a.rows, tid);
}

}  // namespace

hipError_t launch_split16_fixed_bwd(const SolveArgs& a, hipStream_t s) {
    const int grid = (int)((a.rows + kRows - 1) / kRows);
    dbg_timing_begin(s);
    hipLaunchKernelGGL(fixed_bwd_split16_kernel, dim3(grid), dim3(kThreads), 0, s, a);
    dbg_timing_end(s);
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) return err;
    const ParamLayout L = make_layout(kD, kH);
    return launch_reduce_slab(a.slab, grid, L.total, a.grad_params, s, 0);
}

hipError_t launch_split16_bwd(const SolveArgs& a, hipStream_t s) {
    const int grid = (int)((a.rows + kRows - 1) / kRows);
    {
        const hipError_t res = a.spread ? check_resident(dopri5_bwd_split16_kernel<true>, kThreads, 2 * grid)
                                        : check_resident(dopri5_bwd_split16_kernel<false>, kThreads, grid);
        if (res != hipSuccess) return res;
    }
    dbg_timing_begin(s);
    if (a.spread) hipLaunchKernelGGL(dopri5_bwd_split16_kernel<true>, dim3(2 * grid), dim3(kThreads), 0, s, a);
    else hipLaunchKernelGGL(dopri5_bwd_split16_kernel<false>, dim3(grid), dim3(kThreads), 0, s, a);
    dbg_timing_end(s);
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) return err;
    if (a.keep_slabs) return hipSuccess;
    const ParamLayout L = make_layout(kD, kH);
    return launch_reduce_slab(a.slab, grid, L.total, a.grad_params, s, a.spread ? L.w(2) : 0);
}

hipError_t launch_split16_fwd(const SolveArgs& a, hipStream_t s) {
    const int grid = (int)((a.rows + kRows - 1) / kRows);
    {
        const hipError_t res = check_resident(dopri5_split16_kernel, kThreads, grid);
        if (res != hipSuccess) return res;
    }
    dbg_timing_begin(s);
    hipLaunchKernelGGL(dopri5_split16_kernel, dim3(grid), dim3(kThreads), 0, s, a);
    dbg_timing_end(s);
    return hipGetLastError();
}

}  // namespace cnf
