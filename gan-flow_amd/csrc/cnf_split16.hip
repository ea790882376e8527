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
#include "cnf_launch.h"

namespace cnf {
namespace {

constexpr int kD = 16;
constexpr int kC = 5;   // components a lane integrates: its 4 dimensions + logp

struct alignas(16) Wave16Lds {
    float gate[3][2][kH];
    float gate4[2][16];
    float kst[7][kC][64];   // Runge-Kutta slopes, one column per lane
};

struct alignas(16) Xchg16 {
    float xb[2][2][4][64][4];   // hidden activations in C/D layout: [buffer][stream][feature tile][lane][r]
    float part[kNW][2][64][4];  // layer-4 partial results of every wave: [wave][stream][lane][r]
};

__device__ __forceinline__ f32x4 sum4v(const f32x4& a, const f32x4& b, const f32x4& c, const f32x4& d) { return ((a + b) + c) + d; }

__device__ __forceinline__ void gates16(const NetLds<kD>& n, Wave16Lds& ws, int lane, float t) {
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
__device__ __forceinline__ void field_fwd16(const NetLds<kD>& n, Wave16Lds& ws, Xchg16& xc, const f32x4& w1img,
                                            const f32x4& w4img, int lane, int wave, float t, const f32x4& z4,
                                            const f32x4& e4, f32x4& f4, float& div) {
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
            const float x = tanh_f(fmaf(g[r], a[r], c[r]));
            act[0][r] = x;
            act[1][r] = (1.0f - x * x) * g[r] * au[r];
        }
    }
#pragma unroll
    for (int l = 0; l < 2; ++l) {
#pragma unroll
        for (int s = 0; s < 2; ++s) *reinterpret_cast<f32x4*>(&xc.xb[l][s][wave][lane][0]) = act[s];
        __syncthreads();
        f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int Ti = 0; Ti < 4; ++Ti) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(&n.wimg[l][wave * 4 + Ti][lane][0]);
            f32x4 in[2];
#pragma unroll
            for (int s = 0; s < 2; ++s) in[s] = *reinterpret_cast<const f32x4*>(&xc.xb[l][s][Ti][lane][0]);
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int s = 0; s < 2; ++s) acc[s] = mfma16(a[r], in[s][r], acc[s]);
        }
        const f32x4 g = *reinterpret_cast<const f32x4*>(&ws.gate[l + 1][0][fo]);
        const f32x4 c = *reinterpret_cast<const f32x4*>(&ws.gate[l + 1][1][fo]);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float x = tanh_f(fmaf(g[r], acc[0][r], c[r]));
            act[0][r] = x;
            act[1][r] = (1.0f - x * x) * g[r] * acc[1][r];
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
    __shared__ NetLds<kD> net;
    __shared__ Wave16Lds wl[kNW];
    __shared__ Xchg16 xc;
    __shared__ RedLds red;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, q = lane >> 4, j = lane & 15;
    if (tid == 0) red.status = 0;
    stage_net(net, a.params, tid);
    zero_slopes(wl, kNW, tid);
    const ParamLayout L = make_layout(kD, kH);
    const f32x4 w1img = *reinterpret_cast<const f32x4*>(a.params + L.w(0) + (16 * w + j) * kD + 4 * q);
    const f32x4 w4img = *reinterpret_cast<const f32x4*>(a.params + L.w(3) + j * kH + 16 * w + 4 * q);
    __syncthreads();
    const long row = (long)blockIdx.x * kRows + j;
    // every wave carries the same 16 rows: wave 0 reports them, every quarter its own 4 dimensions, quarter 0 logp
    const bool cz = row < a.rows && w == 0;
    const bool cl = cz && q == 0;
    const long rr = row < a.rows ? row : a.rows - 1;
    Wave16Lds& ws = wl[w];

    GridSync gs{a.gran, (int)gridDim.x, 0u};
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

    for (;;) {
        f32x4 z4, f4;
        float div, k[kC];
        bool restart = false;
#pragma unroll
        for (int r = 0; r < 4; ++r) z4[r] = yi[r];
        field_fwd16(net, ws, xc, w1img, w4img, lane, w, sgn * ti, z4, e4, f4, div);
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

}  // namespace

hipError_t launch_split16_fwd(const SolveArgs& a, hipStream_t s) {
    const int grid = (int)((a.rows + kRows - 1) / kRows);
    dbg_timing_begin(s);
    hipLaunchKernelGGL(dopri5_split16_kernel, dim3(grid), dim3(kThreads), 0, s, a);
    dbg_timing_end(s);
    return hipGetLastError();
}

}  // namespace cnf
