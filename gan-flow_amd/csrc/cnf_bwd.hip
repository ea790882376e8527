// cnf_bwd.hip -- the adjoint pass: a mirrored reverse-time dopri5 over (z, logp, a_z) with fused
// vector-Jacobian products and in-register parameter-gradient accumulation (gfx950).
//
// Replaces OdeintAdjointMethod.backward of torchdiffeq 0.0.1 for the single interval CNF uses
// (SURVEY.md section 8a row a10, Appendix D).  What one "augmented dynamics" evaluation does here:
//   1. recompute the forward pass of the ODEnet with its forward-mode tangent streams, keeping per layer
//      the tanh output x, the pre-gate value acc+b and the pre-gate tangents U in registers;
//   2. sweep the layers in reverse with cotangents (-a_z) on f and (+a_logp) on div.  Because div is
//      itself built from tangents, the sweep carries a cotangent per stream and picks up the tanh''
//      cross term -- this is the hand-derived double-backward of `divergence_bf`;
//   3. d/dz gives vjp_z; d/dtheta is accumulated, pre-multiplied by the Runge-Kutta weight of the
//      stage, straight into MFMA accumulators (64x64 layers: sum over rows = the K dimension of an
//      outer-product MFMA fed from a wave-private LDS transpose) and into lane-specialised registers
//      (bias/gate/first/last layer: DPP row reductions).
// a_theta and a_t feed back into nothing and take part in no step-size decision (Appendix D.3), so the
// weighted sum over stages IS the Runge-Kutta update of a_theta; it is kept per step and committed to
// a per-wave slab in global memory when the step is accepted.  a_t is never formed (the reference
// discards it: CNF builds integration_times with torch.tensor, which detaches).
// The stage-1 slope is re-evaluated at every attempt instead of being carried over (FSAL), because its
// parameter-gradient weight (dt * b_1) is only known once the new dt is: 7 evaluations per attempt
// are executed while `nfe` reports the 6 the reference would count.
#include "cnf_launch.h"

namespace cnf {

// ---- LDS of the adjoint kernels ---------------------------------------------------------------------
template <int D>
struct SmallIdx {   // parameters other than the two 64x64 matrices, in canonical order with those matrices cut out
    static constexpr int n0 = kH * D + 4 * kH;            // layer-1 block: W1, b, hb, wg, bg
    static constexpr int v2 = n0;                         // layer-2 vectors
    static constexpr int v3 = n0 + 4 * kH;                // layer-3 vectors
    static constexpr int l4 = n0 + 8 * kH;                // layer-4 block: W4, b, hb, wg, bg
    static constexpr int count = l4 + kH * D + 4 * D;
    static constexpr int padded = (count + 3) & ~3;
    __host__ __device__ static int vec(int layer) { return layer == 0 ? kH * D : (layer == 1 ? v2 : v3); }
    __host__ __device__ static int canonical(int si) { return si < n0 ? si : (si < v3 ? si + kH * kH : si + 2 * kH * kH); }
};

template <int D>
struct alignas(16) BwdWaveLds {
    float gate[3][2][kH];
    float gate4[2][16];
    float kst[7][2 * D + 1][kRows];        // slopes of (z, logp, a_z)
    float small[SmallIdx<D>::padded];      // this wave's share of the step's d/dtheta for the small parameters
    float wq[8];                           // quadrature weights of the current attempt (indexed by stage at run time)
};

struct alignas(16) BwdNetLds {
    float wimgT[2][16][64][4];   // layers 2,3 transposed: [layer][Tin*4+Tout][lane][r] = W[16Tout+4(lane>>4)+r][16Tin+(lane&15)]
    float tr[2][kNW][kH * kRows];  // transposed images [source wave][feature][row-permuted] of one stream: cotangent, activation
};

template <int D>
__device__ inline void stage_net_T(BwdNetLds& n, const float* __restrict__ p, int tid) {
    const ParamLayout L = make_layout(D, kH);
    float* img = &n.wimgT[0][0][0][0];
    for (int idx = tid; idx < 2 * 16 * 64 * 4; idx += kThreads) {
        const int r = idx & 3, lane = (idx >> 2) & 63, tt = (idx >> 8) & 15, l = idx >> 12;
        const int row = 16 * (tt & 3) + 4 * (lane >> 4) + r;   // output feature (contracted)
        const int col = 16 * (tt >> 2) + (lane & 15);          // input feature
        img[idx] = p[L.w(1 + l) + row * kH + col];
    }
}

// sum over the 16 lanes of a row group; every lane of the group receives the total (DPP, no LDS)
__device__ __forceinline__ float row16_allsum(float v) {
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, true));  // row_half_mirror
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, true));  // row_mirror
    return v;
}

// ---- parameter-gradient accumulators of one step ------------------------------------------------------
// 64x64 layers: wave w of the workgroup owns output rows 16w..16w+15 of dW2, dW3 for ALL 64 rows of the
// workgroup (MFMA accumulators); everything else: BwdWaveLds::small of each wave.
struct BigAcc {
    f32x4 gw[2][4];   // [layer][Tin]: lane (q, j') reg r = dW[16w+4q+r][16Tin+j']
    __device__ __forceinline__ void zero() {
#pragma unroll
        for (int l = 0; l < 2; ++l)
#pragma unroll
            for (int b = 0; b < 4; ++b) gw[l][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
};

template <int D>
__device__ __forceinline__ void zero_small(BwdWaveLds<D>& ws, int lane) {
    for (int i = lane; i < SmallIdx<D>::padded; i += 64) ws.small[i] = 0.f;
}

// slab_wg[P] += this step's sums (called by the whole workgroup, between two barriers of its own)
template <int D>
__device__ inline void commit_step(BwdWaveLds<D>* wl, const BigAcc& acc, float* __restrict__ slab_wg, int tid) {
    const ParamLayout L = make_layout(D, kH);
    const int lane = tid & 63, w = tid >> 6, q = lane >> 4, j = lane & 15;
    __syncthreads();
    for (int i = tid; i < SmallIdx<D>::count; i += kThreads) {
        const float tot = ((wl[0].small[i] + wl[1].small[i]) + wl[2].small[i]) + wl[3].small[i];
        slab_wg[SmallIdx<D>::canonical(i)] += tot;
    }
#pragma unroll
    for (int l = 0; l < 2; ++l)
#pragma unroll
        for (int Ti = 0; Ti < 4; ++Ti)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                slab_wg[L.w(1 + l) + (16 * w + 4 * q + r) * kH + 16 * Ti + j] += acc.gw[l][Ti][r];
    __syncthreads();
}

// ---- one evaluation of the augmented dynamics for the 16 rows of this wave ----------------------------
// in : t, z, e (Hutchinson probe), a_z, a_l (adjoints of z and logp for this lane's row), w (weight of this
//      evaluation in the parameter-gradient quadrature; with_theta=false skips that work)
// out: f, div (as field_fwd), vz = -(a_z . df/dz) + a_l * d(div)/dz; acc += w * d/dtheta summed over rows
// regularised dynamics (cnf_solve_bwd_reg): the constant adjoints of the regulariser states enter the sweep as one more
// cotangent on dz/dt, c_k * d integrand_k / d(dz/dt) with c_k = dL/dr_k / rows (the states are batch means)
struct RegCot {
    int nreg;
    int kind[CNF_MAX_REG];
    float c[CNF_MAX_REG];
    float res;   // 1: the dynamics are f(z) - z (CNF_DIV_RESIDUAL), the regularisers see that difference
};

// rc / reg_extra (nullable): RegCot above; reg_extra[D] receives the extra cotangent (the caller's residual term needs it)
template <int D, bool HUTCH>
__device__ __forceinline__ void aug_eval(const NetLds<D>& n, const BwdNetLds& nt, BwdWaveLds<D>& ws, int lane, float t,
                                         const float (&z)[D], const float (&e)[D], const float (&az)[D], float al,
                                         float w, bool with_theta, float (&f)[D], float& div, float (&vz)[D],
                                         BigAcc& acc, BwdNetLds& sh, int wave, const RegCot* rc = nullptr,
                                         float* reg_extra = nullptr) {
    constexpr int NT = HUTCH ? 1 : D;
    constexpr int NS = NT + 1;
    using SI = SmallIdx<D>;
    const int q = lane >> 4, j = lane & 15;
    const int qq = j & 3;        // lanes j < 4 of every quarter add vector quantity j of {b, hb, wg, bg}

    // gates (same arithmetic as compute_gates, on this wave's scratch)
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

    // ------------------------------------------------------------------ forward with saves
    float xs[3][16];        // tanh outputs of layers 1..3
    float ub[3][16];        // pre-gate value: W x + b
    float U[2][NT][16];     // pre-gate tangents of layers 2,3 (layer 1: recomputed from W1)
    float act[NS][16];      // running activations (value, tangents)
#pragma unroll
    for (int T = 0; T < 4; ++T) {
        const int fo = 16 * T + 4 * q;
        const f32x4 g = *reinterpret_cast<const f32x4*>(&ws.gate[0][0][fo]);
        const f32x4 c = *reinterpret_cast<const f32x4*>(&ws.gate[0][1][fo]);
        const f32x4 b = *reinterpret_cast<const f32x4*>(&n.vec[0][0][fo]);
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
            xs[0][4 * T + r] = x;
            ub[0][4 * T + r] = a[r] + b[r];
            act[0][4 * T + r] = x;
            if (HUTCH) {
                act[1][4 * T + r] = dg * au[r];
            } else {
#pragma unroll
                for (int m = 0; m < NT; ++m) act[1 + m][4 * T + r] = dg * wcol[m][r];
            }
        }
    }
#pragma unroll
    for (int l = 0; l < 2; ++l) {
        f32x4 a4[NS][4];
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int To = 0; To < 4; ++To) a4[s][To] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int Ti = 0; Ti < 4; ++Ti) {
            f32x4 a[4];
#pragma unroll
            for (int To = 0; To < 4; ++To) a[To] = *reinterpret_cast<const f32x4*>(&n.wimg[l][To * 4 + Ti][lane][0]);
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int To = 0; To < 4; ++To)
#pragma unroll
                    for (int s = 0; s < NS; ++s) a4[s][To] = mfma16(a[To][r], act[s][4 * Ti + r], a4[s][To]);
        }
#pragma unroll
        for (int To = 0; To < 4; ++To) {
            const int fo = 16 * To + 4 * q;
            const f32x4 g = *reinterpret_cast<const f32x4*>(&ws.gate[l + 1][0][fo]);
            const f32x4 c = *reinterpret_cast<const f32x4*>(&ws.gate[l + 1][1][fo]);
            const f32x4 b = *reinterpret_cast<const f32x4*>(&n.vec[l + 1][0][fo]);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float x = act_f(fmaf(g[r], a4[0][To][r], c[r]));
                const float dg = act_d(x) * g[r];
                xs[l + 1][4 * To + r] = x;
                ub[l + 1][4 * To + r] = a4[0][To][r] + b[r];
                act[0][4 * To + r] = x;
#pragma unroll
                for (int s = 1; s < NS; ++s) {
                    U[l][s - 1][4 * To + r] = a4[s][To][r];
                    act[s][4 * To + r] = dg * a4[s][To][r];
                }
            }
        }
    }
    // layer 4 forward (act = x3, X3_m)
    float pf[D], pd[D];
#pragma unroll
    for (int m = 0; m < D; ++m) pf[m] = pd[m] = 0.f;
#pragma unroll
    for (int T = 0; T < 4; ++T) {
        const int fo = 16 * T + 4 * q;
#pragma unroll
        for (int m = 0; m < D; ++m) {
            const f32x4 w4 = *reinterpret_cast<const f32x4*>(&n.w4[m][fo]);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                pf[m] = fmaf(w4[r], act[0][4 * T + r], pf[m]);
                pd[m] = fmaf(w4[r], act[HUTCH ? 1 : 1 + m][4 * T + r], pd[m]);
            }
        }
    }
    float g4[D], u4[D], U4[D];
    div = 0.f;
#pragma unroll
    for (int m = 0; m < D; ++m) {
        g4[m] = ws.gate4[0][m];
        u4[m] = xq_sum(pf[m]);
        U4[m] = xq_sum(pd[m]);
        f[m] = fmaf(g4[m], u4[m], ws.gate4[1][m]);
        const float dm = g4[m] * U4[m];
        div += HUTCH ? dm * e[m] : dm;
    }

    // ------------------------------------------------------------------ reverse sweep
    // cotangents: lf[m] on f_m, ld on div.  cot[0] = value stream, cot[1+m] = tangent stream m.
    float lf[D];
#pragma unroll
    for (int m = 0; m < D; ++m) lf[m] = -az[m];
    if (rc) {
        float dx[D], extra[D];
#pragma unroll
        for (int m = 0; m < D; ++m) dx[m] = fmaf(-rc->res, z[m], f[m]);
        reg_cotangent<D>(rc->nreg, rc->kind, rc->c, dx, extra);
#pragma unroll
        for (int m = 0; m < D; ++m) {
            lf[m] = -(az[m] + extra[m]);
            if (reg_extra) reg_extra[m] = extra[m];
        }
    }
    const float ld = al;
    float cot[NS][16];

    // ---- layer 4 (vector ALU).  ubar4_m = g4_m lf_m ; Ubar4_m = g4_m ld (exact) | g4_m ld e_m (Hutchinson)
    {
        float ubar[D], Ubar[D];
#pragma unroll
        for (int m = 0; m < D; ++m) {
            ubar[m] = g4[m] * lf[m];
            Ubar[m] = g4[m] * ld * (HUTCH ? e[m] : 1.0f);
        }
#pragma unroll
        for (int T = 0; T < 4; ++T) {
            const int fo = 16 * T + 4 * q;
            f32x4 w4[D];
#pragma unroll
            for (int m = 0; m < D; ++m) w4[m] = *reinterpret_cast<const f32x4*>(&n.w4[m][fo]);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int s = 4 * T + r;
                float xb = 0.f, Xh = 0.f;
#pragma unroll
                for (int m = 0; m < D; ++m) {
                    xb = fmaf(ubar[m], w4[m][r], xb);
                    if (HUTCH) Xh = fmaf(Ubar[m], w4[m][r], Xh);
                    else cot[1 + (HUTCH ? 0 : m)][s] = Ubar[m] * w4[m][r];
                }
                cot[0][s] = xb;
                if (HUTCH) cot[1][s] = Xh;
                if (with_theta) {
                    // dW4[m][feature] over rows: ubar_m x3 + Ubar_m X3_(m); lane j < D adds row m = j
                    float mine = 0.f;
#pragma unroll
                    for (int m = 0; m < D; ++m) {
                        const float v = row16_allsum(ubar[m] * act[0][s] + Ubar[m] * act[HUTCH ? 1 : 1 + m][s]);
                        mine = (j == m) ? v : mine;
                    }
                    if (j < D) ws.small[SI::l4 + j * kH + fo + r] += w * mine;
                }
            }
        }
        if (with_theta) {
#pragma unroll
            for (int m = 0; m < D; ++m) {
                // only quarter 0 holds each row once for the row sum: lanes of other quarters hold the same rows
                const float s4 = row16_allsum(lf[m]);
                const float gb = row16_allsum(lf[m] * (u4[m] + n.vec4[0][m]) + ld * (HUTCH ? e[m] : 1.0f) * U4[m]);
                const float gg = g4[m] * (1.0f - g4[m]);
                const float val = qq < 2 ? s4 : gb;
                const float fac = qq == 0 ? g4[m] : (qq == 1 ? t : (qq == 2 ? t * gg : gg));
                if (lane < 4) ws.small[SI::l4 + kH * D + qq * D + m] += w * fac * val;
            }
        }
    }

    // ---- layers 3 and 2 (matrix cores), then layer 1 (vector ALU)
#pragma unroll
    for (int l = 1; l >= -1; --l) {
        const int L = l + 1;  // index into xs/ub/gate: layer L+1 in 1-based numbering
        // through the tanh of this layer; vector gradients; gated cotangents
        float s1[16], s2[16];
#pragma unroll
        for (int T = 0; T < 4; ++T) {
            const int fo = 16 * T + 4 * q;
            const f32x4 g = *reinterpret_cast<const f32x4*>(&ws.gate[L][0][fo]);
            f32x4 w1c[D];
            if (l < 0) {
#pragma unroll
                for (int m = 0; m < D; ++m) w1c[m] = *reinterpret_cast<const f32x4*>(&n.w1[m][fo]);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int s = 4 * T + r;
                const float x = xs[L][s];
                const float dx = act_d(x);
                float pre = dx * cot[0][s];
                float q2 = 0.f;
#pragma unroll
                for (int k = 0; k < NT; ++k) {
                    float Uk;
                    if (l >= 0) {
                        Uk = U[l < 0 ? 0 : l][k][s];
                    } else if (HUTCH) {
                        Uk = 0.f;
#pragma unroll
                        for (int m = 0; m < D; ++m) Uk = fmaf(w1c[m][r], e[m], Uk);
                    } else {
                        Uk = w1c[k][r];
                    }
                    const float Xb = cot[1 + k][s];
                    pre = fmaf(Xb * (g[r] * Uk), act_dd(x, dx), pre);
                    const float Pb = dx * Xb;
                    q2 = fmaf(Pb, Uk, q2);
                    cot[1 + k][s] = g[r] * Pb;       // Ubar_k
                }
                s1[s] = pre;
                s2[s] = fmaf(pre, ub[L][s], q2);
                cot[0][s] = g[r] * pre;              // ubar
            }
        }
        if (with_theta) {
#pragma unroll
            for (int T = 0; T < 4; ++T) {
                const int fo = 16 * T + 4 * q;
                const f32x4 g = *reinterpret_cast<const f32x4*>(&ws.gate[L][0][fo]);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int s = 4 * T + r;
                    const float st = row16_allsum(s1[s]);
                    const float gt = row16_allsum(s2[s]);
                    const float gg = g[r] * (1.0f - g[r]);
                    const float val = qq < 2 ? st : gt;
                    const float fac = qq == 0 ? g[r] : (qq == 1 ? t : (qq == 2 ? t * gg : gg));
                    if (j < 4) ws.small[SI::vec(L) + qq * kH + fo + r] += w * fac * val;
                }
            }
        }
        if (l >= 0) {
            const int lc = l < 0 ? 0 : l;
            if (with_theta) {
                // dW += w * sum_rows (ubar (x) x_in + sum_k Ubar_k (x) X_in,k): transpose both through LDS, rows become K
#pragma unroll
                for (int s = 0; s < NS; ++s) {
#pragma unroll
                    for (int T = 0; T < 4; ++T) {
                        const f32x4 gi = *reinterpret_cast<const f32x4*>(&ws.gate[L - 1][0][16 * T + 4 * q]);
                        f32x4 w1c[D];
                        if (lc == 0 && s > 0) {
#pragma unroll
                            for (int m = 0; m < D; ++m) w1c[m] = *reinterpret_cast<const f32x4*>(&n.w1[m][16 * T + 4 * q]);
                        }
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int sl = 4 * T + r;
                            const int addr = (16 * T + 4 * q + r) * kRows + (j & 3) * 4 + (j >> 2);
                            float xin;
                            if (s == 0) {
                                xin = xs[L - 1][sl];
                            } else {
                                const float xi = xs[L - 1][sl];
                                float Uin;
                                if (lc > 0) {
                                    Uin = U[lc > 0 ? lc - 1 : 0][s - 1][sl];
                                } else if (HUTCH) {
                                    Uin = 0.f;
#pragma unroll
                                    for (int m = 0; m < D; ++m) Uin = fmaf(w1c[m][r], e[m], Uin);
                                } else {
                                    Uin = w1c[s - 1][r];
                                }
                                xin = act_d(xi) * gi[r] * Uin;
                            }
                            sh.tr[0][wave][addr] = w * cot[s][sl];
                            sh.tr[1][wave][addr] = xin;
                        }
                    }
                    __syncthreads();
                    // this wave: output rows 16*wave .. 16*wave+15 of dW, all 64 rows of the workgroup as K
#pragma unroll
                    for (int src = 0; src < kNW; ++src) {
                        const f32x4 a4 = *reinterpret_cast<const f32x4*>(&sh.tr[0][src][(16 * wave + j) * kRows + 4 * q]);
                        f32x4 b4[4];
#pragma unroll
                        for (int T = 0; T < 4; ++T)
                            b4[T] = *reinterpret_cast<const f32x4*>(&sh.tr[1][src][(16 * T + j) * kRows + 4 * q]);
#pragma unroll
                        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                            for (int Ti = 0; Ti < 4; ++Ti) acc.gw[lc][Ti] = mfma16(a4[ks], b4[Ti][ks], acc.gw[lc][Ti]);
                    }
                    __syncthreads();
                }
            }
            // cotangents of this layer's inputs: W^T ubar, W^T Ubar_k
            f32x4 o4[NS][4];
#pragma unroll
            for (int s = 0; s < NS; ++s)
#pragma unroll
                for (int Ti = 0; Ti < 4; ++Ti) o4[s][Ti] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int To = 0; To < 4; ++To) {
                f32x4 a[4];
#pragma unroll
                for (int Ti = 0; Ti < 4; ++Ti) a[Ti] = *reinterpret_cast<const f32x4*>(&nt.wimgT[lc][Ti * 4 + To][lane][0]);
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int Ti = 0; Ti < 4; ++Ti)
#pragma unroll
                        for (int s = 0; s < NS; ++s) o4[s][Ti] = mfma16(a[Ti][r], cot[s][4 * To + r], o4[s][Ti]);
            }
#pragma unroll
            for (int s = 0; s < NS; ++s)
#pragma unroll
                for (int Ti = 0; Ti < 4; ++Ti)
#pragma unroll
                    for (int r = 0; r < 4; ++r) cot[s][4 * Ti + r] = o4[s][Ti][r];
        } else {
            // layer 1: vjp_z and dW1
            float pz[D];
#pragma unroll
            for (int m = 0; m < D; ++m) pz[m] = 0.f;
#pragma unroll
            for (int T = 0; T < 4; ++T) {
                const int fo = 16 * T + 4 * q;
                f32x4 w1c[D];
#pragma unroll
                for (int m = 0; m < D; ++m) w1c[m] = *reinterpret_cast<const f32x4*>(&n.w1[m][fo]);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int s = 4 * T + r;
                    float mine = 0.f;
#pragma unroll
                    for (int m = 0; m < D; ++m) {
                        pz[m] = fmaf(w1c[m][r], cot[0][s], pz[m]);
                        if (with_theta) {
                            // exact: tangent stream m enters through column m of W1 with a unit seed
                            const float tang = HUTCH ? cot[1][s] * e[m] : cot[1 + (HUTCH ? 0 : m)][s];
                            const float v = row16_allsum(fmaf(cot[0][s], z[m], tang));
                            mine = (j == m) ? v : mine;
                        }
                    }
                    if (with_theta && j < D) ws.small[(fo + r) * D + j] += w * mine;
                }
            }
#pragma unroll
            for (int m = 0; m < D; ++m) vz[m] = xq_sum(pz[m]);
        }
    }
}

// ---- slab -> flat gradient: sum the per-workgroup partials in a fixed order (deterministic) ---------------
__global__ void reduce_slab_kernel(const float* __restrict__ slab, int nwg, int P, float* __restrict__ out) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    float s = 0.f;
    for (int g = 0; g < nwg; ++g) s += slab[(size_t)g * P + p];
    out[p] = s;
}

// ---- single augmented evaluation (cnf_eval_aug) ----------------------------------------------------------
template <int D, bool HUTCH>
__global__ __launch_bounds__(kThreads) void aug_kernel(SolveArgs a, float t, const float* __restrict__ z,
                                                       const float* __restrict__ a_z, const float* __restrict__ a_l,
                                                       float* __restrict__ f_out, float* __restrict__ nd_out,
                                                       float* __restrict__ vz_out) {
    const float res = a.residual ? 1.f : 0.f;   // CNF_DIV_RESIDUAL: dynamics f(z) - z
    __shared__ NetLds<D> net;
    __shared__ BwdNetLds netT;
    __shared__ BwdWaveLds<D> wl[kNW];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, j = lane & 15;
    stage_net(net, a.params, tid);
    stage_net_T<D>(netT, a.params, tid);
    zero_small(wl[w], lane);
    __syncthreads();
    const long base = ((long)blockIdx.x * kNW + w) * kRows;
    const long row = base + j;
    const bool valid = row < a.rows;
    const long rr = valid ? row : a.rows - 1;
    float zz[D], ee[D], az[D], f[D], vz[D], div;
#pragma unroll
    for (int m = 0; m < D; ++m) {
        zz[m] = z[rr * D + m];
        ee[m] = HUTCH ? a.e[rr * D + m] : 0.f;
        az[m] = valid ? a_z[rr * D + m] : 0.f;    // padded rows carry zero adjoints: they add nothing to any sum
    }
    const float al = valid ? a_l[rr] : 0.f;
    BigAcc acc;
    acc.zero();
    aug_eval<D, HUTCH>(net, netT, wl[w], lane, t, zz, ee, az, al, 1.0f, true, f, div, vz, acc, netT, w);
    #pragma unroll
    for (int m = 0; m < D; ++m) {   // residual dynamics f(z) - z (res = 0 or 1): d/dz gains -I
        f[m] = fmaf(-res, zz[m], f[m]);
        vz[m] = fmaf(res, az[m], vz[m]);
    }
    div -= res * (float)D;
    if (lane < 16 && valid) {
#pragma unroll
        for (int m = 0; m < D; ++m) {
            f_out[row * D + m] = f[m];
            vz_out[row * D + m] = vz[m];
        }
        nd_out[row] = -div;
    }
    commit_step<D>(wl, acc, dbg_ws(a, a.slab + (size_t)blockIdx.x * make_layout(D, kH).total, sizeof(float) * make_layout(D, kH).total, __LINE__), tid);
}

// ---- the adjoint solve -------------------------------------------------------------------------------------
// REG: regularised dynamics (cnf_solve_bwd_reg) -- an instance of its own, so that the ordinary adjoint's registers are its own
template <int D, bool HUTCH, bool REG = false>
__global__ __launch_bounds__(kThreads) void dopri5_bwd_kernel(SolveArgs a) {
    const float res = a.residual ? 1.f : 0.f;   // CNF_DIV_RESIDUAL: dynamics f(z) - z
    __shared__ NetLds<D> net;
    __shared__ BwdNetLds netT;
    __shared__ BwdWaveLds<D> wl[kNW];
    __shared__ RedLds red;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, j = lane & 15;
    constexpr int C = 2 * D + 1;   // z, logp, a_z
    constexpr int CE = D + 1;      // components under error control (z, logp)
    if (tid == 0) red.status = 0;
    stage_net(net, a.params, tid);
    stage_net_T<D>(netT, a.params, tid);
    zero_slopes(wl, kNW, tid);
    __syncthreads();
    const long base = ((long)blockIdx.x * kNW + w) * kRows;
    const long row = base + j;
    const bool valid = row < a.rows;
    const bool counted = valid && lane < 16;
    const long rr = valid ? row : a.rows - 1;
    BwdWaveLds<D>& ws = wl[w];
    float* slab = dbg_ws(a, a.slab + (size_t)blockIdx.x * make_layout(D, kH).total, sizeof(float) * make_layout(D, kH).total, __LINE__);   // this workgroup's partial of a_theta

    GridSync gs{dbg_gran(a, __LINE__), (int)gridDim.x, 0u};
    const double nl = (double)a.rows, nz = nl * D;

    float y0[C], y1[C], yi[C], ee[D];
#pragma unroll
    for (int m = 0; m < D; ++m) {
        y0[m] = a.x[rr * D + m];
        ee[m] = HUTCH ? a.e[rr * D + m] : 0.f;
        y0[D + 1 + m] = valid ? a.grad_z1[rr * D + m] : 0.f;
    }
    y0[D] = a.logp0[rr];
    const float al = valid ? a.grad_logp1[rr] : 0.f;   // d(a_logp)/dt = 0: constant along the solve
#pragma unroll
    for (int c = 0; c < C; ++c) yi[c] = y0[c];

    // the adjoint pass runs from t1 back to t0
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
    float sc[CE];
    BigAcc acc;
    acc.zero();

    // regularised dynamics (cnf_solve_bwd_reg).  The regulariser states ride along in the reference's augmented state with
    // tolerances of 1e20: they never fail an attempt and nothing reads them at t0, so they are not integrated here.  What
    // remains of them: (i) their constant adjoints, one more cotangent on dz/dt in every evaluation (RegCot); (ii) the first
    // step, whose heuristic weighs their values at t1 and their slopes like any other controlled tensor.
    const int nreg = REG ? a.nreg : 0;
    RegCot rcot{nreg, {a.reg_kind[0], a.reg_kind[1]}, {0.f, 0.f}, res};
    float reg_scale[CNF_MAX_REG] = {atol, atol}, d0r[CNF_MAX_REG] = {0.f, 0.f}, d1r[CNF_MAX_REG] = {0.f, 0.f}, kr_f0[CNF_MAX_REG] = {0.f, 0.f};
    if (nreg) {
#pragma unroll
        for (int q_ = 0; q_ < CNF_MAX_REG; ++q_)
            if (q_ < nreg) {
                rcot.c[q_] = valid ? a.reg_cot[q_] / (float)a.rows : 0.f;   // rows past the batch are clamped copies: no cotangent
                const float r = a.reg_val[q_];
                reg_scale[q_] = atol + fabsf(r) * rtol;
                d0r[q_] = fabsf(r / reg_scale[q_]);
            }
    }

    if (a.after) status = a.after->status;   // CnfSolverOpts.after: the solve whose output this one consumes failed
    for (; status == 0;) {
        float zz[D], az[D], f[D], vz[D], div, k[C];
        bool restart = false;
#pragma unroll
        for (int m = 0; m < D; ++m) {
            zz[m] = yi[m];
            az[m] = yi[D + 1 + m];
        }
        const bool with_theta = phase == PH_STAGE && ws.wq[stage] != 0.f;
        // initial-step evaluations only need (f, div); the sweep still runs (one code path), its outputs are unused
        float rextra[D];
#pragma unroll
        for (int m = 0; m < D; ++m) rextra[m] = 0.f;
        aug_eval<D, HUTCH>(net, netT, ws, lane, sgn * ti, zz, ee, az, al, phase == PH_STAGE ? sgn * ws.wq[stage] : 0.f,
                           with_theta, f, div, vz, acc, netT, w, (REG && nreg) ? &rcot : nullptr, REG ? rextra : nullptr);
        #pragma unroll
        for (int m = 0; m < D; ++m) {   // residual dynamics f(z) - z (res = 0 or 1): d/dz gains -I
            f[m] = fmaf(-res, zz[m], f[m]);
            vz[m] = fmaf(res, az[m] + rextra[m], vz[m]);
        }
        div -= res * (float)D;
        ++nrun;
        float kr[CNF_MAX_REG] = {0.f, 0.f};   // slopes of the regulariser states, for the first step only
        if (nreg && phase != PH_STAGE) {
#pragma unroll
            for (int q_ = 0; q_ < CNF_MAX_REG; ++q_)
                if (q_ < nreg) kr[q_] = sgn * reg_integrand<D>(a.reg_kind[q_], f);
        }
#pragma unroll
        for (int m = 0; m < D; ++m) {
            k[m] = sgn * f[m];
            k[D + 1 + m] = sgn * vz[m];
        }
        k[D] = -sgn * div;

        if (phase == PH_F0) {
            double S[4] = {0, 0, 0, 0};
#pragma unroll
            for (int c = 0; c < CE; ++c) {
                ws.kst[0][c][j] = k[c];
                sc[c] = atol + fabsf(y0[c]) * rtol;
                const float u = y0[c] / sc[c], v = k[c] / sc[c];
                if (counted) {
                    S[c < D ? 0 : 1] += (double)u * u;
                    S[c < D ? 2 : 3] += (double)v * v;
                }
            }
#pragma unroll
            for (int c = CE; c < C; ++c) ws.kst[0][c][j] = k[c];
            if (!wg_allreduce4(gs, red, S, w, lane)) { status = CNF_ERR_SYNC_TIMEOUT; break; }
            const float d0z = (float)sqrt(S[0] / nz), d0l = (float)sqrt(S[1] / nl);
            d1z = (float)sqrt(S[2] / nz);
            d1l = (float)sqrt(S[3] / nl);
            float d0a = 0.f, d1a = 0.f, d0al = 0.f;
            if (a.adj_all) {   // CNF_FLAG_ADJ_CONTROL_ALL (cnf_device.h): a_z and a_logp take part in the heuristic
                double S2[4] = {0, 0, 0, 0};
                adj_all_f0_sums<D, C>(y0, k, al, rtol, atol, counted, S2);
                if (!wg_allreduce4(gs, red, S2, w, lane)) { status = CNF_ERR_SYNC_TIMEOUT; break; }
                d0a = (float)sqrt(S2[0] / nz);
                d1a = (float)sqrt(S2[1] / nz);
                d0al = (float)sqrt(S2[2] / nl);
                if (tid == 0) red.keep[0] = d1a;   // read after the next reduction's barriers
            }
            if (nreg) {   // (not combined with CNF_FLAG_ADJ_CONTROL_ALL: the API refuses that)
                double S2[4] = {0, 0, 0, 0};
#pragma unroll
                for (int q_ = 0; q_ < CNF_MAX_REG; ++q_) {
                    kr_f0[q_] = kr[q_];
                    if (counted) S2[q_] = (double)kr[q_];
                }
                if (!wg_allreduce4(gs, red, S2, w, lane)) { status = CNF_ERR_SYNC_TIMEOUT; break; }
#pragma unroll
                for (int q_ = 0; q_ < CNF_MAX_REG; ++q_) d1r[q_] = fabsf((float)(S2[q_] / nl) / reg_scale[q_]);
                h0 = initial_h0_reg(d0z, d0l, d1z, d1l, nreg, d0r, d1r);
            } else {
                h0 = initial_h0(d0z, d0l, d1z, d1l, a.adj_all != 0, d0a, d1a, d0al);
            }
            if (h0 < INFINITY) {
#pragma unroll
                for (int c = 0; c < C; ++c) yi[c] = y0[c] + h0 * k[c];
                ti = t0f + h0;
            }   // else (adj_all): the second evaluation runs at the initial state; initial_dt() ignores what it returns
            phase = PH_F1;
            continue;
        }
        if (phase == PH_F1) {
            double S[4] = {0, 0, 0, 0};
#pragma unroll
            for (int c = 0; c < CE; ++c) {
                const float u = (k[c] - ws.kst[0][c][j]) / sc[c];
                if (counted) S[c < D ? 0 : 1] += (double)u * u;
            }
            if (a.adj_all && counted) {
#pragma unroll
                for (int m = 0; m < D; ++m) {
                    const int c = D + 1 + m;
                    const float u = (k[c] - ws.kst[0][c][j]) / (atol + fabsf(y0[c]) * rtol);
                    S[3] += (double)u * u;
                }
            }
            if (nreg && counted) {   // the change of the regularisers' batch means, in the two slots this sum leaves free
#pragma unroll
                for (int q_ = 0; q_ < CNF_MAX_REG; ++q_) S[2 + q_] = (double)(kr[q_] - kr_f0[q_]);
            }
            if (!wg_allreduce4(gs, red, S, w, lane)) { status = CNF_ERR_SYNC_TIMEOUT; break; }
            const float d2z = (float)sqrt(S[0] / nz) / h0, d2l = (float)sqrt(S[1] / nl) / h0;
            if (nreg) {
                float d2r[CNF_MAX_REG];
#pragma unroll
                for (int q_ = 0; q_ < CNF_MAX_REG; ++q_) d2r[q_] = fabsf((float)(S[2 + q_] / nl) / reg_scale[q_]) / h0;
                dt = (double)initial_dt_reg(h0, d1z, d1l, d2z, d2l, nreg, d1r, d2r);
            } else {
                const float d2a = (float)sqrt(S[3] / nz) / h0;
                dt = (double)initial_dt(h0, d1z, d1l, d2z, d2l, a.adj_all != 0, a.adj_all ? red.keep[0] : 0.f, d2a);
            }
            restart = true;
        } else {
            // slope number stage+1 (stage 0 = the re-evaluated first slope)
#pragma unroll
            for (int c = 0; c < C; ++c) ws.kst[stage][c][j] = k[c];
            if (stage < 6) {
                ++stage;
            } else {
                double S[4] = {0, 0, 0, 0};
#pragma unroll
                for (int c = 0; c < C; ++c) y1[c] = yi[c];
#pragma unroll
                for (int c = 0; c < CE; ++c) {
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
                if (!wg_allreduce4(gs, red, S, w, lane)) { status = CNF_ERR_SYNC_TIMEOUT; break; }
                if (S[2] > 0.0) { status = CNF_ERR_NONFINITE; break; }
                const float rz = (float)(S[0] / nz), rl = (float)(S[1] / nl);
                const float ra = (float)(S[3] / nz);   // 0 unless CNF_FLAG_ADJ_CONTROL_ALL (a_logp's ratio is always 0)
                const bool accept = (rz <= 1.f) && (rl <= 1.f) && (ra <= 1.f);
                if (accept) {
                    ++nacc;
                    // a_theta += this step's weighted sum
                    commit_step<D>(wl, acc, slab, tid);
                    if (!(tau_end > tau + dt)) break;
#pragma unroll
                    for (int c = 0; c < C; ++c) y0[c] = y1[c];
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
            stage = 0;
            phase = PH_STAGE;
            acc.zero();
            zero_small(ws, lane);
            // quadrature weights of a_theta for this attempt: dt*b_j, or -- if this step would reach the end
            // time -- the weights of the quartic dense output at tau_end (linear in the slopes, Appendix B.6)
            if (tau_end > tau + dt) {
#pragma unroll
                for (int s = 0; s < 6; ++s) ws.wq[s] = dtf * kBeta[5][s];
                ws.wq[6] = 0.f;
            } else {
                const float ta = (float)tau, tb = (float)(tau + dt), te = (float)tau_end;
                const float x = (te - ta) / (tb - ta);
                const float x2 = x * x, x3 = x2 * x, x4 = x3 * x;
#pragma unroll
                for (int s = 0; s < 7; ++s) {
                    const float b = s < 6 ? kBeta[5][s] : 0.f;
                    const float cm = kCMid[s];
                    float v = dtf * (x4 * (-8.f * b + 16.f * cm) + x3 * (14.f * b - 32.f * cm) + x2 * (-5.f * b + 16.f * cm));
                    if (s == 0) v += dtf * (-2.f * x4 + 5.f * x3 - 4.f * x2 + x);
                    if (s == 6) v += dtf * (2.f * x4 - 3.f * x3 + x2);
                    ws.wq[s] = v;
                }
            }
            wave_lds_fence();
        }
        // input of stage `stage`: stage 0 re-evaluates the slope at (tau, y0)
        if (stage == 0) {
            ti = t0f;
#pragma unroll
            for (int c = 0; c < C; ++c) yi[c] = y0[c];
        } else {
            float alpha_, bc_[6];
            stage_coeffs(stage, alpha_, bc_);
            ti = t0f + alpha_ * dtf;
#pragma unroll
            for (int c = 0; c < C; ++c) {
                float s_ = 0.f;
#pragma unroll
                for (int s = 0; s < 6; ++s) s_ += (dtf * bc_[s]) * ws.kst[s][c][j];   // zero weights past the current stage
                yi[c] = y0[c] + s_;
            }
        }
    }

    if (status == 0) {
        // dense output of a_z at tau_end (z and logp are not needed by the caller)
        const float ta = (float)tau, tb = (float)(tau + dt), te = (float)tau_end;
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
            if (counted) a.grad_x[row * D + m] = out;
        }
        if (counted) a.grad_logp0[row] = al;
    }
    if (lane == 0 && ((blockIdx.x == 0 && w == 0) || status != 0)) {
        if (status) atomicMax(&a.stats->status, status);
        atomicMax(&a.stats->nfe, 2 + 6 * (nacc + nrej));
        atomicMax(&a.stats->n_accepted, nacc);
        atomicMax(&a.stats->n_rejected, nrej);
        atomicMax(&a.stats->evals_run, nrun);
        a.stats->last_dt = (float)dt;
        atomicAdd((unsigned long long*)&a.stats->row_evals, (unsigned long long)(a.rows * (long)nrun));
    }
}

// ---- the adjoint solve on a fixed grid ------------------------------------------------------------------------
// odeint_adjoint(..., method='euler' | 'midpoint' | 'rk4', options={'step_size': h}): the backward pass integrates the
// augmented state (z, logp, a_z, a_logp, a_t, a_theta) from t1 back to t0 with the SAME fixed-grid method (torchdiffeq's
// FixedGridODESolver on the negated time axis: grid points tau_i = -t1 + i h, the last clipped to -t0; no step control, so
// no grid-wide reductions and any number of rows).  a_theta is one more state of that integration: its increment over a
// step is the method's weighted sum of the stages' vjp_theta, i.e. the quadrature weights below (rk4 = the 3/8 rule's
// dt/8 (1, 3, 3, 1); midpoint: the first stage only moves the state; euler: dt), accumulated over ALL steps in the
// accumulators and stored once.  Same augmented evaluation (aug_eval) and the same time / sign conventions as
// dopri5_bwd_kernel.  Parity unpinned (no reference artefact holds gradients): tests compare with the oracle's adjoint.
template <int D, bool HUTCH>
__global__ __launch_bounds__(kThreads) void fixed_bwd_kernel(SolveArgs a) {
    const float res = a.residual ? 1.f : 0.f;   // CNF_DIV_RESIDUAL: dynamics f(z) - z
    __shared__ NetLds<D> net;
    __shared__ BwdNetLds netT;
    __shared__ BwdWaveLds<D> wl[kNW];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, j = lane & 15;
    constexpr int C = 2 * D + 1;   // z, logp, a_z
    stage_net(net, a.params, tid);
    stage_net_T<D>(netT, a.params, tid);
    zero_slopes(wl, kNW, tid);
    zero_small(wl[w], lane);
    __syncthreads();
    const long base = ((long)blockIdx.x * kNW + w) * kRows;
    const long row = base + j;
    const bool valid = row < a.rows;
    const bool counted = valid && lane < 16;
    const long rr = valid ? row : a.rows - 1;
    BwdWaveLds<D>& ws = wl[w];
    float* slab = dbg_ws(a, a.slab + (size_t)blockIdx.x * make_layout(D, kH).total, sizeof(float) * make_layout(D, kH).total, __LINE__);   // this workgroup's partial of a_theta

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

    // the adjoint pass runs from t1 back to t0
    const float sgn = a.t0 >= a.t1 ? 1.f : -1.f;
    const float tau0 = sgn * a.t1, tau1 = sgn * a.t0;
    const int nst = a.fixed_kind == CNF_SOLVER_EULER ? 1 : (a.fixed_kind == CNF_SOLVER_MIDPOINT ? 2 : 4);
    BigAcc acc;
    acc.zero();
    int nrun = 0;
    const int n_steps = status == 0 ? a.n_grid : 0;
    for (int step = 0; step < n_steps; ++step) {
        // grid point i = i*step_size + tau0 (product rounded before the add, like arange()*h + t0)
        const float ta = a.n_grid == 1 ? tau0 : __fadd_rn(__fmul_rn((float)step, a.step_size), tau0);
        const float tb = (step + 1 == a.n_grid) ? tau1 : __fadd_rn(__fmul_rn((float)(step + 1), a.step_size), tau0);
        const float dt = tb - ta;
        float k1[C], k2[C], k3[C], yi[C];
        float ti = ta;
#pragma unroll
        for (int c = 0; c < C; ++c) yi[c] = y[c];
        for (int st = 0; st < nst; ++st) {
            float zz[D], az[D], f[D], vz[D], div, k[C];
#pragma unroll
            for (int m = 0; m < D; ++m) {
                zz[m] = yi[m];
                az[m] = yi[D + 1 + m];
            }
            // weight of this stage's vjp_theta in the step's increment of a_theta
            float wq;
            if (a.fixed_kind == CNF_SOLVER_EULER) wq = dt;
            else if (a.fixed_kind == CNF_SOLVER_MIDPOINT) wq = st == 0 ? 0.f : dt;
            else wq = (st == 0 || st == 3) ? dt / 8 : 3 * (dt / 8);
            aug_eval<D, HUTCH>(net, netT, ws, lane, sgn * ti, zz, ee, az, al, sgn * wq, wq != 0.f, f, div, vz, acc, netT, w);
#pragma unroll
            for (int m = 0; m < D; ++m) {   // residual dynamics f(z) - z (res = 0 or 1): d/dz gains -I
                f[m] = fmaf(-res, zz[m], f[m]);
                vz[m] = fmaf(res, az[m], vz[m]);
            }
            div -= res * (float)D;
            ++nrun;
#pragma unroll
            for (int m = 0; m < D; ++m) {
                k[m] = sgn * f[m];
                k[D + 1 + m] = sgn * vz[m];
            }
            k[D] = -sgn * div;
            if (a.fixed_kind == CNF_SOLVER_EULER) {            // y + dt f(t, y)
#pragma unroll
                for (int c = 0; c < C; ++c) y[c] = y[c] + dt * k[c];
            } else if (a.fixed_kind == CNF_SOLVER_MIDPOINT) {  // y + dt f(t + dt/2, y + f(t, y) dt/2)
                if (st == 0) {
                    ti = ta + dt / 2;
#pragma unroll
                    for (int c = 0; c < C; ++c) yi[c] = y[c] + k[c] * (dt / 2);
                } else {
#pragma unroll
                    for (int c = 0; c < C; ++c) y[c] = y[c] + dt * k[c];
                }
            } else if (st == 0) {                              // the 3/8 rule, as rk4_kernel (cnf_fwd.hip)
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
    if (status == 0 && counted) {
#pragma unroll
        for (int m = 0; m < D; ++m) a.grad_x[row * D + m] = y[D + 1 + m];
        a.grad_logp0[row] = al;
    }
    commit_step<D>(wl, acc, slab, tid);   // every wave of the workgroup (barriers inside); zero sums after a failed predecessor
    if (lane == 0 && ((blockIdx.x == 0 && w == 0) || status != 0)) {
        if (status) atomicMax(&a.stats->status, status);
        atomicMax(&a.stats->nfe, nst * a.n_grid);
        atomicMax(&a.stats->n_accepted, a.n_grid);
        atomicMax(&a.stats->evals_run, nrun);
        atomicAdd((unsigned long long*)&a.stats->row_evals, (unsigned long long)(a.rows * (long)nrun));
    }
}

// ---- launchers ---------------------------------------------------------------------------------------------
template <int D, bool HUTCH>
static hipError_t launch_eval_aug_t(const SolveArgs& a, float t, const float* z, const float* a_z, const float* a_l,
                                    float* f_out, float* nd_out, float* vz_out, hipStream_t s) {
    const int grid = (int)((a.rows + kRowsPerWG - 1) / kRowsPerWG);
    hipLaunchKernelGGL((aug_kernel<D, HUTCH>), dim3(grid), dim3(kThreads), 0, s, a, t, z, a_z, a_l, f_out, nd_out, vz_out);
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) return err;
    const int P = make_layout(D, kH).total;
    hipLaunchKernelGGL(reduce_slab_kernel, dim3((P + 255) / 256), dim3(256), 0, s, a.slab, grid, P, a.grad_params);
    return hipGetLastError();
}

hipError_t launch_eval_aug(int D, bool hutch, const SolveArgs& a, float t, const float* z, const float* a_z,
                           const float* a_logp, float* f_out, float* nd_out, float* vjp_z_out, hipStream_t s) {
    if (D == 1 && !hutch) return launch_eval_aug_t<1, false>(a, t, z, a_z, a_logp, f_out, nd_out, vjp_z_out, s);
    if (D == 2 && !hutch) return launch_eval_aug_t<2, false>(a, t, z, a_z, a_logp, f_out, nd_out, vjp_z_out, s);
    if (D == 2 && hutch) return launch_eval_aug_t<2, true>(a, t, z, a_z, a_logp, f_out, nd_out, vjp_z_out, s);
    return hipErrorNotSupported;
}

template <int D, bool HUTCH>
static hipError_t launch_solve_bwd_t(const SolveArgs& a, hipStream_t s) {
    const int grid = (int)((a.rows + kRowsPerWG - 1) / kRowsPerWG);
    {
        const hipError_t res = a.nreg ? check_resident(dopri5_bwd_kernel<D, HUTCH, true>, kThreads, grid)
                                      : check_resident(dopri5_bwd_kernel<D, HUTCH, false>, kThreads, grid);
        if (res != hipSuccess) return res;
    }
    dbg_timing_begin(s);
    if (a.nreg) hipLaunchKernelGGL((dopri5_bwd_kernel<D, HUTCH, true>), dim3(grid), dim3(kThreads), 0, s, a);
    else hipLaunchKernelGGL((dopri5_bwd_kernel<D, HUTCH, false>), dim3(grid), dim3(kThreads), 0, s, a);
    dbg_timing_end(s);
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) return err;
    const int P = make_layout(D, kH).total;
    hipLaunchKernelGGL(reduce_slab_kernel, dim3((P + 255) / 256), dim3(256), 0, s, a.slab, grid, P, a.grad_params);
    return hipGetLastError();
}

hipError_t launch_solve_bwd(int D, bool hutch, const SolveArgs& a, hipStream_t s) {
    if (D == 1 && !hutch) return launch_solve_bwd_t<1, false>(a, s);
    if (D == 2 && !hutch) return launch_solve_bwd_t<2, false>(a, s);
    if (D == 2 && hutch) return launch_solve_bwd_t<2, true>(a, s);
    return hipErrorNotSupported;
}

template <int D, bool HUTCH>
static hipError_t launch_fixed_bwd_t(const SolveArgs& a, hipStream_t s) {
    const int grid = (int)((a.rows + kRowsPerWG - 1) / kRowsPerWG);
    dbg_timing_begin(s);
    hipLaunchKernelGGL((fixed_bwd_kernel<D, HUTCH>), dim3(grid), dim3(kThreads), 0, s, a);
    dbg_timing_end(s);
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) return err;
    const int P = make_layout(D, kH).total;
    hipLaunchKernelGGL(reduce_slab_kernel, dim3((P + 255) / 256), dim3(256), 0, s, a.slab, grid, P, a.grad_params);
    return hipGetLastError();
}

// fixed-grid adjoint (a.fixed_kind, a.n_grid, a.step_size as the forward fixed-grid solve reads them), hidden width 64, dim <= 2
hipError_t launch_fixed_bwd(int D, bool hutch, const SolveArgs& a, hipStream_t s) {
    if (D == 1 && !hutch) return launch_fixed_bwd_t<1, false>(a, s);
    if (D == 2 && !hutch) return launch_fixed_bwd_t<2, false>(a, s);
    if (D == 2 && hutch) return launch_fixed_bwd_t<2, true>(a, s);
    return hipErrorNotSupported;
}

}  // namespace cnf
