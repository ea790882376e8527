// cnf_wide_bwd.hip -- adjoint at hidden width 256 (BASELINE config 4 training).
//
// Same decomposition as the forward (cnf_wide.hip): one 16-row tile per workgroup, wave w owns features 64w..64w+63
// of every hidden layer for all streams.  One augmented evaluation = forward with saves, reverse sweep with a
// cotangent per stream (mathematics of aug_eval_split, cnf_split.hip), d/dz for the a_z slope, d/dtheta for the
// quadrature.  What differs from width 64 is where d/dtheta lives: 2 x 256 x 256 sums per workgroup do not fit in
// registers or LDS, so each evaluation writes its UNWEIGHTED sums (the two weight matrices as 16x16 MFMA result
// blocks, the small parameters as scalars) to a per-stage slot of the workgroup's workspace region -- plain
// coalesced stores, never read-modify-write -- and an accepted step folds the seven stage slots into the
// workgroup's running total with the Runge-Kutta weights, which are all known by then.  A rejected attempt just
// overwrites its slots; the seventh stage's slot becomes the next step's first (FSAL, see kFsalTheta).
// The W^T products of the reverse sweep stream a transposed copy of the two matrices that a prep kernel writes into
// the workspace once per call.  After the solve a reduction kernel adds the workgroups' totals in a fixed order and
// converts the block layout back to the nn.Linear layout: deterministic gradients, no float atomics.
#include "cnf_wide.h"

namespace cnf {

template <int D>
struct WideSmallIdx {   // parameters other than the two 256x256 matrices, canonical order with those matrices cut out
    static constexpr int n0 = kWH * D + 4 * kWH;
    static constexpr int v2 = n0;
    static constexpr int v3 = n0 + 4 * kWH;
    static constexpr int l4 = n0 + 8 * kWH;
    static constexpr int count = l4 + kWH * D + 4 * D;
    static constexpr int padded = (count + 3) & ~3;
    __host__ __device__ static int vec(int layer) { return layer == 0 ? kWH * D : (layer == 1 ? v2 : v3); }
    __host__ __device__ static int canonical(int si) { return si < n0 ? si : (si < v3 ? si + kWH * kWH : si + 2 * kWH * kWH); }
};

constexpr int kWideBig = 2 * kWT * kWT * 64 * 4;   // floats of the two matrices' sums, [layer][To][Ti][lane][r]
#ifndef CNF_WIDE_REMAT_LANE
#define CNF_WIDE_REMAT_LANE 5
#endif
constexpr int kWideRematLane = CNF_WIDE_REMAT_LANE;   // wide_bwd_kernel: 1 = the evaluation's, 4 = the grid reductions' copy of the lane index
constexpr int kWideSlots = 8;                      // 0..6: stage slots, 7: running total of the accepted steps
constexpr int kWideTotal = 7;

template <int D>
constexpr int kWideSlotF = kWideBig + WideSmallIdx<D>::padded;

__device__ __forceinline__ float row16_allsum_w(float v) {
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, true));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, true));
    return v;
}

// transposed copies of the two 256x256 matrices: wt[l][i][o] = W_l[o][i]
__global__ void wide_transpose_kernel(const float* __restrict__ p, int D, float* __restrict__ wt) {
    __shared__ float tile[32][33];
    const ParamLayout L = make_layout(D, kWH);
    const int l = blockIdx.z, bx = blockIdx.x * 32, by = blockIdx.y * 32;
    const float* W = p + L.w(1 + l);
    for (int r = threadIdx.y; r < 32; r += 8) tile[r][threadIdx.x] = W[(by + r) * kWH + bx + threadIdx.x];
    __syncthreads();
    for (int r = threadIdx.y; r < 32; r += 8) wt[(size_t)l * kWH * kWH + (bx + r) * kWH + by + threadIdx.x] = tile[threadIdx.x][r];
}

// ---- one augmented evaluation ---------------------------------------------------------------------------------------
// slot: this evaluation's d/dtheta slot (global), written only when with_theta; w = +-1 (time direction).
template <int D, bool HUTCH, class WS>
__device__ __forceinline__ void aug_eval_wide(const WideNetLds<D>& n, WS& ws, WideXchg<D, (HUTCH ? 1 : D) + 1>& xc,
                                              float (&part2)[kNW][D][kRows], const float* __restrict__ params,
                                              const float* __restrict__ wt, int lane, int wave, float t,
                                              const float (&z)[D], const float (&e)[D], const float (&az)[D], float al,
                                              float w, bool with_theta, float (&f)[D], float& div, float (&vz)[D],
                                              float* __restrict__ slot, int part = 0) {
    constexpr int NT = HUTCH ? 1 : D;
    constexpr int NS = NT + 1;
    using SI = WideSmallIdx<D>;
    // part: which d/dtheta sums this workgroup writes (CNF_FLAG_SPREAD, kSpreadNote in cnf_split.hip): 0 = all,
    // 1 = W3, layer-3 vectors, layer 4; 2 = W1, layer-1/2 vectors, W2
    const bool th_hi = with_theta && part != 2, th_lo = with_theta && part != 1;
    const int q = lane >> 4, j = lane & 15, qq = j & 3;
    const int g0 = (((j >> 2) * 16 + 4 * q) * 4) + (j & 3);   // gather offset of (feature tile-local j, row 4q + ks): + 4 ks
    float* slot_small = slot + kWideBig;
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

    // ------------------------------------------------------------------ forward with saves (own 4 tiles)
    f32x4 xs[3][kWTW], ub[3][kWTW], U[2][NT][kWTW], act[kWTW][NS], w1c[D][kWTW];
#pragma unroll
    for (int T = 0; T < kWTW; ++T) {
        const int lo = 16 * T + 4 * q, fo = kWF * wave + lo;
        const f32x4 g = *reinterpret_cast<const f32x4*>(&ws.gate[0][0][lo]);
        const f32x4 c = *reinterpret_cast<const f32x4*>(&ws.gate[0][1][lo]);
        const f32x4 b = *reinterpret_cast<const f32x4*>(&n.vec[0][0][fo]);
        f32x4 a = {0.f, 0.f, 0.f, 0.f}, au = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int m = 0; m < D; ++m) {
            w1c[m][T] = *reinterpret_cast<const f32x4*>(&n.w1[m][fo]);
            a += w1c[m][T] * z[m];
            if (HUTCH) au += w1c[m][T] * e[m];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float x = act_f(fmaf(g[r], a[r], c[r]));
            const float dg = act_d(x) * g[r];
            xs[0][T][r] = x;
            ub[0][T][r] = a[r] + b[r];
            act[T][0][r] = x;
            if (HUTCH) {
                act[T][1][r] = dg * au[r];
            } else {
#pragma unroll
                for (int m = 0; m < NT; ++m) act[T][1 + m][r] = dg * w1c[m][T][r];
            }
        }
    }
#pragma unroll
    for (int l = 0; l < 2; ++l) {
#pragma unroll
        for (int T = 0; T < kWTW; ++T)
#pragma unroll
            for (int s = 0; s < NS; ++s) *reinterpret_cast<f32x4*>(&xc.xb[l][s][kWTW * wave + T][lane][0]) = act[T][s];
        const float* wrow = params + make_layout(D, kWH).w(1 + l) + (size_t)(kWF * wave + j) * kWH + 4 * q;
        f32x4 wnext[kWTW];
#pragma unroll
        for (int T = 0; T < kWTW; ++T) wnext[T] = *reinterpret_cast<const f32x4*>(wrow + 16 * T * kWH);
        __syncthreads();
        f32x4 a4[kWTW][NS];
#pragma unroll
        for (int T = 0; T < kWTW; ++T)
#pragma unroll
            for (int s = 0; s < NS; ++s) a4[T][s] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
        for (int Ti = 0; Ti < kWT; ++Ti) {
            f32x4 wcur[kWTW];
#pragma unroll
            for (int T = 0; T < kWTW; ++T) wcur[T] = wnext[T];
            const int Tn = Ti + 1 < kWT ? Ti + 1 : Ti;
#pragma unroll
            for (int T = 0; T < kWTW; ++T) wnext[T] = *reinterpret_cast<const f32x4*>(wrow + 16 * T * kWH + 16 * Tn);
            f32x4 in[NS];
#pragma unroll
            for (int s = 0; s < NS; ++s) in[s] = *reinterpret_cast<const f32x4*>(&xc.xb[l][s][Ti][lane][0]);
#pragma unroll
            for (int T = 0; T < kWTW; ++T)
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int s = 0; s < NS; ++s) a4[T][s] = mfma16(wcur[T][r], in[s][r], a4[T][s]);
        }
#pragma unroll
        for (int T = 0; T < kWTW; ++T) {
            const int lo = 16 * T + 4 * q, fo = kWF * wave + lo;
            const f32x4 g = *reinterpret_cast<const f32x4*>(&ws.gate[l + 1][0][lo]);
            const f32x4 c = *reinterpret_cast<const f32x4*>(&ws.gate[l + 1][1][lo]);
            const f32x4 b = *reinterpret_cast<const f32x4*>(&n.vec[l + 1][0][fo]);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float x = act_f(fmaf(g[r], a4[T][0][r], c[r]));
                const float dg = act_d(x) * g[r];
                xs[l + 1][T][r] = x;
                ub[l + 1][T][r] = a4[T][0][r] + b[r];
                act[T][0][r] = x;
#pragma unroll
                for (int s = 1; s < NS; ++s) {
                    U[l][s - 1][T][r] = a4[T][s][r];
                    act[T][s][r] = dg * a4[T][s][r];
                }
            }
        }
    }
    f32x4 w4[D][kWTW];
#pragma unroll
    for (int m = 0; m < D; ++m) {
        float pf = 0.f, pd = 0.f;
#pragma unroll
        for (int T = 0; T < kWTW; ++T) {
            w4[m][T] = *reinterpret_cast<const f32x4*>(&n.w4[m][kWF * wave + 16 * T + 4 * q]);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                pf = fmaf(w4[m][T][r], act[T][0][r], pf);
                pd = fmaf(w4[m][T][r], act[T][HUTCH ? 1 : 1 + m][r], pd);
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
    float g4[D], u4[D], U4[D];
    div = 0.f;
#pragma unroll
    for (int m = 0; m < D; ++m) {
        g4[m] = ws.gate4[0][m];
        u4[m] = wsum4(xc.part[0][m][j], xc.part[1][m][j], xc.part[2][m][j], xc.part[3][m][j]);
        U4[m] = wsum4(xc.part[0][D + m][j], xc.part[1][D + m][j], xc.part[2][D + m][j], xc.part[3][D + m][j]);
        f[m] = fmaf(g4[m], u4[m], ws.gate4[1][m]);
        const float dm = g4[m] * U4[m];
        div += HUTCH ? dm * e[m] : dm;
    }

    // ------------------------------------------------------------------ reverse sweep
    float lf[D];
#pragma unroll
    for (int m = 0; m < D; ++m) lf[m] = -az[m];
    const float ld = al;
    f32x4 cot[NS][kWTW];
    {
        float ubar[D], Ubar[D];
#pragma unroll
        for (int m = 0; m < D; ++m) {
            ubar[m] = g4[m] * lf[m];
            Ubar[m] = g4[m] * ld * (HUTCH ? e[m] : 1.0f);
        }
#pragma unroll
        for (int T = 0; T < kWTW; ++T) {
            const int fo = kWF * wave + 16 * T + 4 * q;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float xb = 0.f, Xh = 0.f;
#pragma unroll
                for (int m = 0; m < D; ++m) {
                    xb = fmaf(ubar[m], w4[m][T][r], xb);
                    if (HUTCH) Xh = fmaf(Ubar[m], w4[m][T][r], Xh);
                    else cot[1 + (HUTCH ? 0 : m)][T][r] = Ubar[m] * w4[m][T][r];
                }
                cot[0][T][r] = xb;
                if (HUTCH) cot[1][T][r] = Xh;
                if (th_hi) {
                    float mine = 0.f;
#pragma unroll
                    for (int m = 0; m < D; ++m) {
                        const float v = row16_allsum_w(ubar[m] * act[T][0][r] + Ubar[m] * act[T][HUTCH ? 1 : 1 + m][r]);
                        mine = (j == m) ? v : mine;
                    }
                    if (j < D) slot_small[SI::l4 + j * kWH + fo + r] = w * mine;
                }
            }
        }
        if (th_hi) {
#pragma unroll
            for (int m = 0; m < D; ++m) {
                const float s4 = row16_allsum_w(lf[m]);
                const float gb = row16_allsum_w(lf[m] * (u4[m] + n.vec4[0][m]) + ld * (HUTCH ? e[m] : 1.0f) * U4[m]);
                const float gg = g4[m] * (1.0f - g4[m]);
                const float val = qq < 2 ? s4 : gb;
                const float fac = qq == 0 ? g4[m] : (qq == 1 ? t : (qq == 2 ? t * gg : gg));
                if (wave == 0 && lane < 4) slot_small[SI::l4 + kWH * D + qq * D + m] = w * fac * val;
            }
        }
    }

#pragma unroll
    for (int l = 1; l >= -1; --l) {
        const int L = l + 1;
#pragma unroll
        for (int T = 0; T < kWTW; ++T) {
            const int lo = 16 * T + 4 * q, fo = kWF * wave + lo;
            const f32x4 g = *reinterpret_cast<const f32x4*>(&ws.gate[L][0][lo]);
            float s1[4], s2[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float x = xs[L][T][r];
                const float dx = act_d(x);
                float pre = dx * cot[0][T][r];
                float q2 = 0.f;
#pragma unroll
                for (int k = 0; k < NT; ++k) {
                    float Uk;
                    if (l >= 0) {
                        Uk = U[l < 0 ? 0 : l][k][T][r];
                    } else if (HUTCH) {
                        Uk = 0.f;
#pragma unroll
                        for (int m = 0; m < D; ++m) Uk = fmaf(w1c[m][T][r], e[m], Uk);
                    } else {
                        Uk = w1c[k][T][r];
                    }
                    const float Xb = cot[1 + k][T][r];
                    pre = fmaf(Xb * (g[r] * Uk), act_dd(x, dx), pre);
                    const float Pb = dx * Xb;
                    q2 = fmaf(Pb, Uk, q2);
                    cot[1 + k][T][r] = g[r] * Pb;
                }
                s1[r] = pre;
                s2[r] = fmaf(pre, ub[L][T][r], q2);
                cot[0][T][r] = g[r] * pre;
            }
            if (L == 2 ? th_hi : th_lo) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float st = row16_allsum_w(s1[r]);
                    const float gt = row16_allsum_w(s2[r]);
                    const float gg = g[r] * (1.0f - g[r]);
                    const float val = qq < 2 ? st : gt;
                    const float fac = qq == 0 ? g[r] : (qq == 1 ? t : (qq == 2 ? t * gg : gg));
                    if (j < 4) slot_small[SI::vec(L) + qq * kWH + fo + r] = w * fac * val;
                }
            }
        }
        if (l >= 0) {
            const int lc = l < 0 ? 0 : l;
            // buffer 0 <- this layer's cotangents (all streams); buffer 1 holds the layer's inputs: layer 3's are still
            // there from the forward pass, layer 2's were overwritten by it and are rebuilt from the saves
            if (lc == 0) __syncthreads();   // every wave is done with layer 3's cotangents / inputs
#pragma unroll
            for (int T = 0; T < kWTW; ++T)
#pragma unroll
                for (int s = 0; s < NS; ++s) *reinterpret_cast<f32x4*>(&xc.xb[0][s][kWTW * wave + T][lane][0]) = cot[s][T];
            if (lc == 0 && th_lo) {
#pragma unroll
                for (int T = 0; T < kWTW; ++T) {
                    const f32x4 gi = *reinterpret_cast<const f32x4*>(&ws.gate[0][0][16 * T + 4 * q]);
#pragma unroll
                    for (int s = 0; s < NS; ++s) {
                        f32x4 xin;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float xi = xs[0][T][r];
                            if (s == 0) {
                                xin[r] = xi;
                            } else {
                                float Uin;
                                if (HUTCH) {
                                    Uin = 0.f;
#pragma unroll
                                    for (int m = 0; m < D; ++m) Uin = fmaf(w1c[m][T][r], e[m], Uin);
                                } else {
                                    Uin = w1c[s - 1][T][r];
                                }
                                xin[r] = act_d(xi) * gi[r] * Uin;
                            }
                        }
                        *reinterpret_cast<f32x4*>(&xc.xb[1][s][kWTW * wave + T][lane][0]) = xin;
                    }
                }
            }
            // transposed W, rows = this wave's INPUT features, streamed like the forward weights
            const float* trow = wt + (size_t)lc * kWH * kWH + (size_t)(kWF * wave + j) * kWH + 4 * q;
            f32x4 wnext[kWTW];
#pragma unroll
            for (int T = 0; T < kWTW; ++T) wnext[T] = *reinterpret_cast<const f32x4*>(trow + 16 * T * kWH);
            __syncthreads();
            if (lc == 1 ? th_hi : th_lo) {
                // dW[16 To + i'][16 Ti + j'] = sum_rows sum_streams cot_s[i'][row] x_s[j'][row]: K = the 16 rows,
                // operands gathered (transposed) from the C/D-layout buffers; unweighted, one 16x16 block per store
                float aop[kWTW][NS][4];
#pragma unroll
                for (int T = 0; T < kWTW; ++T)
#pragma unroll
                    for (int s = 0; s < NS; ++s)
#pragma unroll
                        for (int ks = 0; ks < 4; ++ks) aop[T][s][ks] = w * (&xc.xb[0][s][kWTW * wave + T][0][0])[g0 + 4 * ks];
                float* blk = slot + ((size_t)(lc * kWT + kWTW * wave) * kWT) * 256 + lane * 4;
#pragma unroll 2
                for (int Ti = 0; Ti < kWT; ++Ti) {
                    float bop[NS][4];
#pragma unroll
                    for (int s = 0; s < NS; ++s)
#pragma unroll
                        for (int ks = 0; ks < 4; ++ks) bop[s][ks] = (&xc.xb[1][s][Ti][0][0])[g0 + 4 * ks];
#pragma unroll
                    for (int T = 0; T < kWTW; ++T) {
                        f32x4 dacc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int s = 0; s < NS; ++s)
#pragma unroll
                            for (int ks = 0; ks < 4; ++ks) dacc = mfma16(aop[T][s][ks], bop[s][ks], dacc);
                        *reinterpret_cast<f32x4*>(blk + ((size_t)T * kWT + Ti) * 256) = dacc;
                    }
                }
            }
            f32x4 o4[kWTW][NS];
#pragma unroll
            for (int T = 0; T < kWTW; ++T)
#pragma unroll
                for (int s = 0; s < NS; ++s) o4[T][s] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
            for (int To = 0; To < kWT; ++To) {
                f32x4 wcur[kWTW];
#pragma unroll
                for (int T = 0; T < kWTW; ++T) wcur[T] = wnext[T];
                const int Tn = To + 1 < kWT ? To + 1 : To;
#pragma unroll
                for (int T = 0; T < kWTW; ++T) wnext[T] = *reinterpret_cast<const f32x4*>(trow + 16 * T * kWH + 16 * Tn);
                f32x4 in[NS];
#pragma unroll
                for (int s = 0; s < NS; ++s) in[s] = *reinterpret_cast<const f32x4*>(&xc.xb[0][s][To][lane][0]);
#pragma unroll
                for (int T = 0; T < kWTW; ++T)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int s = 0; s < NS; ++s) o4[T][s] = mfma16(wcur[T][r], in[s][r], o4[T][s]);
            }
#pragma unroll
            for (int T = 0; T < kWTW; ++T)
#pragma unroll
                for (int s = 0; s < NS; ++s) cot[s][T] = o4[T][s];
        } else {
#pragma unroll
            for (int m = 0; m < D; ++m) {
                float pz = 0.f;
#pragma unroll
                for (int T = 0; T < kWTW; ++T)
#pragma unroll
                    for (int r = 0; r < 4; ++r) pz = fmaf(w1c[m][T][r], cot[0][T][r], pz);
                pz = xq_sum(pz);
                if (q == 0) part2[wave][m][j] = pz;
            }
            if (th_lo) {
#pragma unroll
                for (int T = 0; T < kWTW; ++T) {
                    const int fo = kWF * wave + 16 * T + 4 * q;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float mine = 0.f;
#pragma unroll
                        for (int m = 0; m < D; ++m) {
                            const float tang = HUTCH ? cot[1][T][r] * e[m] : cot[1 + (HUTCH ? 0 : m)][T][r];
                            const float v = row16_allsum_w(fmaf(cot[0][T][r], z[m], tang));
                            mine = (j == m) ? v : mine;
                        }
                        if (j < D) slot_small[(fo + r) * D + j] = w * mine;
                    }
                }
            }
            __syncthreads();
#pragma unroll
            for (int m = 0; m < D; ++m) vz[m] = wsum4(part2[0][m][j], part2[1][m][j], part2[2][m][j], part2[3][m][j]);
        }
    }
}

}  // namespace cnf
#include "cnf_wide_bwd_b16.h"
namespace cnf {

// The 256 x 256 products of the adjoint on the bf16 matrix pipe (cnf_wide_bwd_b16.h); 0 rebuilds the fp32-MFMA evaluation
// above (tools/build_variants.sh, A/B timing).
#ifndef CNF_WIDE_BWD_BF16
#define CNF_WIDE_BWD_BF16 1
#endif
constexpr bool kWideBwdB16 = CNF_WIDE_BWD_BF16 != 0;
// the fixed-grid adjoint: factors + fold on the matrix cores like the dopri5 kernel (1), or result blocks per stage and the
// slot fold (0)
#ifndef CNF_WIDE_FIXED_FACT
#define CNF_WIDE_FIXED_FACT 1
#endif
constexpr bool kWideFixedFact = kWideBwdB16 && CNF_WIDE_FIXED_FACT != 0;
// head of the adjoint's workspace, in floats: W^T of the two matrices in fp32, or the pieces of W and W^T
constexpr size_t kWideBwdHeadFloats = kWideBwdB16 ? 2 * kWideSetBytes / sizeof(float) : (size_t)2 * kWH * kWH;

// Shared memory of the adjoint kernels, by evaluation.  bf16: the two piece buffers take 144 KB at three streams, so the
// slopes are ONE array for the four waves.  Every wave carries an identical copy of the Runge-Kutta state and computes
// identical slopes (f, div and vz come out of LDS sums read in one order), so all four store the SAME value to a slot, each
// reads a slot only after its own store to it or under a zero weight, and a slot is rewritten for the next step only behind
// the barriers of the step's grid reduction and commit: no wave can observe a value that differs from its own.
template <int D, int NS, int C, bool B16>
struct WideBwdShared;
template <int D, int NS, int C>
struct WideBwdShared<D, NS, C, false> {
    static constexpr bool kB16 = false;
    WideNetLds<D> net;
    WideWaveLds<C> wl[kNW];
    WideXchg<D, NS> xc;
    float part2[kNW][D][kRows];
    __device__ __forceinline__ float (&kst(int w))[7][C][kRows] { return wl[w].kst; }
    __device__ __forceinline__ void stage(const float* __restrict__ p, int tid) { wide_stage_net(net, p, tid); }
    __device__ __forceinline__ void zero(int tid) { zero_slopes(wl, kNW, tid); }
    __device__ __forceinline__ void* fold_area() { return nullptr; }
};
template <int D, int NS, int C>
struct WideBwdShared<D, NS, C, true> {
    static constexpr bool kB16 = true;
    WideBwdLds<D, NS> x;
    float kst1[7][C][kRows];
    __device__ __forceinline__ float (&kst(int))[7][C][kRows] { return kst1; }
    __device__ __forceinline__ void stage(const float* __restrict__ p, int tid) { wide_bwd_stage(x, p, tid); }
    __device__ __forceinline__ void zero(int tid) {
        for (int idx = tid; idx < 7 * C * kRows; idx += kThreads) (&kst1[0][0][0])[idx] = 0.f;   // zero_slopes: 0 x stale NaN
    }
    // an accepted step's fold stages its factors where the evaluations exchange theirs (wide_fold_factors)
    __device__ __forceinline__ void* fold_area() {
        static_assert(sizeof(WideFoldLds<NS>) <= sizeof(x.xb16), "the fold's piece area fits in the exchange buffers");
        return &x.xb16[0][0][0][0][0][0];
    }
};
// wmat: the workspace head (fp32: transposed matrices; bf16: wide_bwd_pieces_kernel's output)
template <int D, bool HUTCH, bool FACT, class SH>
__device__ __forceinline__ void aug_eval_sh(SH& sh, const float* __restrict__ params, const float* __restrict__ wmat, int lane,
                                            int wave, float t, const float (&z)[D], const float (&e)[D], const float (&az)[D],
                                            float al, float w, bool with_theta, float (&f)[D], float& div, float (&vz)[D],
                                            float* __restrict__ slot, int part = 0) {
    constexpr int NS = (HUTCH ? 1 : D) + 1;
    if constexpr (SH::kB16)
        aug_eval_wide_b16<D, HUTCH, NS, FACT>(sh.x, params, reinterpret_cast<const u32x4*>(wmat), lane, wave, t, z, e, az, al, w, with_theta,
                                        f, div, vz, slot, part);
    else
        aug_eval_wide<D, HUTCH>(sh.net, sh.wl[wave], sh.xc, sh.part2, params, wmat, lane, wave, t, z, e, az, al, w, with_theta, f,
                                div, vz, slot, part);
}

// fold the stage slots of an accepted step into the workgroup's running total: total (+)= sum_s wts[s] * slot[sl[s]]
// FACT (NS = streams): the stage slots hold the matrices' sums as factors (cnf_wide_bwd_b16.h): their share is formed here
template <int D, int NS = 0, bool FACT = false>
__device__ inline void wide_commit(float* __restrict__ region, const int (&sl)[7], const float (&wts)[7], bool first, int tid,
                                   int part = 0, void* fold_lds = nullptr) {
    constexpr int SF = kWideSlotF<D>;
    // the slots are written and folded by waves of this one workgroup (one compute unit, one vector L1): workgroup scope
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    f32x4* tot = reinterpret_cast<f32x4*>(region + (size_t)kWideTotal * SF);
    constexpr int U = 4;
    // the slot as two ranges of 16-byte words: [matrix sums | small-parameter sums]; a spread workgroup (part 1 / 2) folds
    // the half of each it wrote: part 1 the upper halves (W3; small indices >= v3), part 2 the lower ones
    constexpr int kBig4 = kWideBig / 4, kV34 = kBig4 + WideSmallIdx<D>::v3 / 4;
    static_assert(WideSmallIdx<D>::v3 % 4 == 0 && kWideBig % 8 == 0, "16-byte ranges");
    const int lo4[2] = {part == 1 ? kBig4 / 2 : 0, part == 1 ? kV34 : kBig4};
    const int hi4[2] = {part == 2 ? kBig4 / 2 : kBig4, part == 2 ? kV34 : SF / 4};
    if constexpr (FACT) {
        static_assert(2 * kWideFactLayer<(NS > 0 ? NS : 1)> <= kWideBig, "factors fit in a slot's matrix area");
#ifndef CNF_EXP_WIDE_NO_COMMIT
        for (int lc = 0; lc < 2; ++lc)
            if (part == 0 || part == 2 - lc)   // part 1 keeps W3 (lc = 1), part 2 keeps W2 (lc = 0)
                wide_fold_factors<NS>(*reinterpret_cast<WideFoldLds<(NS > 0 ? NS : 1)>*>(fold_lds), region, (size_t)SF, sl, wts, lc, first,
                                      region + (size_t)kWideTotal * SF, tid);
#endif
    }
    for (int rg = FACT ? 1 : 0; rg < 2; ++rg) {
#ifdef CNF_EXP_WIDE_NO_COMMIT   // timing experiment: no fold of the stage slots
        if (wts[0] < 1e30f) break;
#endif
        const int end = hi4[rg];
        for (int base = lo4[rg] + tid; base < end; base += kThreads * U) {
            f32x4 acc[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int i = base + u * kThreads;
                acc[u] = (!first && i < end) ? tot[i] : f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int s = 0; s < 7; ++s) {
                if (wts[s] != 0.f) {
                    const f32x4* src = reinterpret_cast<const f32x4*>(region + (size_t)sl[s] * SF);
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const int i = base + u * kThreads;
                        if (i < end) acc[u] += wts[s] * src[i];
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int i = base + u * kThreads;
                if (i < end) tot[i] = acc[u];
            }
        }
    }
    __syncthreads();
}

// ---- single augmented evaluation (cnf_eval_aug) ----------------------------------------------------------------------
template <int D, bool HUTCH>
__global__ __launch_bounds__(kThreads) void wide_aug_kernel(SolveArgs a, const float* __restrict__ wt, float t,
                                                            const float* __restrict__ z, const float* __restrict__ a_z,
                                                            const float* __restrict__ a_l, float* __restrict__ f_out,
                                                            float* __restrict__ nd_out, float* __restrict__ vz_out) {
    const float res = a.residual ? 1.f : 0.f;   // CNF_DIV_RESIDUAL: dynamics f(z) - z
    constexpr int NS = (HUTCH ? 1 : D) + 1;
    constexpr int C = 2 * D + 1;
    __shared__ WideBwdShared<D, NS, C, kWideBwdB16> sh;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, j = lane & 15;
    wt = dbg_ws(a, wt, sizeof(float) * kWideBwdHeadFloats, __LINE__);
    sh.stage(a.params, tid);
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
    float* region = dbg_ws(a, a.slab + (size_t)blockIdx.x * kWideSlots * kWideSlotF<D>, sizeof(float) * kWideSlots * kWideSlotF<D>, __LINE__);
    aug_eval_sh<D, HUTCH, false>(sh, a.params, wt, lane, w, t, zz, ee, az, al, 1.0f, true, f, div, vz, region);
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
}

// ---- adjoint solve: the controller of dopri5_bwd_split_kernel with the FSAL d/dtheta bookkeeping on slots --------------
template <int D, bool HUTCH>
__global__ __launch_bounds__(kThreads) void wide_bwd_kernel(SolveArgs a, const float* __restrict__ wt) {
    const float res = a.residual ? 1.f : 0.f;   // CNF_DIV_RESIDUAL: dynamics f(z) - z
    constexpr int NS = (HUTCH ? 1 : D) + 1;
    constexpr int C = 2 * D + 1;
    constexpr int CE = D + 1;
    __shared__ WideBwdShared<D, NS, C, kWideBwdB16> sh;
    __shared__ RedLds red;
    __shared__ float wq[8];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, j = lane & 15;
    if (tid == 0) red.status = 0;
    wt = dbg_ws(a, wt, sizeof(float) * kWideBwdHeadFloats, __LINE__);
    sh.stage(a.params, tid);
    sh.zero(tid);
    __syncthreads();
    // a.spread: two workgroups per tile, each keeping half of the d/dtheta sums (kSpreadNote, cnf_split.hip)
    const int tile = a.spread ? (int)(blockIdx.x >> 1) : (int)blockIdx.x;
    const int part = a.spread ? 1 + (int)(blockIdx.x & 1) : 0;
    const long row = (long)tile * kRows + j;
    const bool valid = row < a.rows;
    const bool counted = valid && lane < 16 && w == 0 && part != 2;
    const long rr = valid ? row : a.rows - 1;
    float (&kst)[7][C][kRows] = sh.kst(w);
    float* region = dbg_ws(a, a.slab + (size_t)blockIdx.x * kWideSlots * kWideSlotF<D>, sizeof(float) * kWideSlots * kWideSlotF<D>, __LINE__);

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
    const float al = valid ? a.grad_logp1[rr] : 0.f;
#pragma unroll
    for (int c = 0; c < C; ++c) yi[c] = y0[c];

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
    int i_first = 0, i_last = 6;   // slots of the first / seventh stage's sums; stages 1..5 use slots 1..5

    if (a.after) status = a.after->status;   // CnfSolverOpts.after: the solve whose output this one consumes failed
    for (; status == 0;) {
        float zz[D], az[D], f[D], vz[D], div, k[C];
        bool restart = false;
#pragma unroll
        for (int m = 0; m < D; ++m) {
            zz[m] = yi[m];
            az[m] = yi[D + 1 + m];
        }
        // every evaluation that enters the quadrature writes its unweighted sums to its slot; stages whose weight is
        // zero (the second stage of every step) skip the d/dtheta work, commit skips their slot
        int slot_i = stage;
        bool with_theta = false;
        if (phase == PH_F0) {
            slot_i = i_first;
            with_theta = true;
        } else if (phase == PH_STAGE) {
            slot_i = stage == 6 ? i_last : stage;
            with_theta = stage == 6 || wq[stage] != 0.f;
        }
        // kRematLaneNote (cnf_split.hip): values that are functions of the lane index alone are recomputed per iteration
        // instead of being hoisted out of the loop and spilled (this kernel runs at 512 registers per wave)
        int lane_e = lane, lane_r = lane;
        if constexpr ((kWideRematLane & 1) != 0) asm volatile("" : "+v"(lane_e));
        if constexpr ((kWideRematLane & 4) != 0) asm volatile("" : "+v"(lane_r));
        aug_eval_sh<D, HUTCH, kWideBwdB16>(sh, a.params, wt, lane_e, w, sgn * ti, zz, ee, az, al, sgn, with_theta, f, div, vz,
                              dbg_ws(a, region + (size_t)slot_i * kWideSlotF<D>, sizeof(float) * kWideSlotF<D>, __LINE__), part);
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

        if (phase == PH_F0) {
            double S[4] = {0, 0, 0, 0};
#pragma unroll
            for (int c = 0; c < CE; ++c) {
                kst[0][c][j] = k[c];
                sc[c] = atol + fabsf(y0[c]) * rtol;
                const float u = y0[c] / sc[c], v = k[c] / sc[c];
                if (counted) {
                    S[c < D ? 0 : 1] += (double)u * u;
                    S[c < D ? 2 : 3] += (double)v * v;
                }
            }
#pragma unroll
            for (int c = CE; c < C; ++c) kst[0][c][j] = k[c];
            if (!wg_allreduce4(gs, red, S, w, lane_r)) { status = CNF_ERR_SYNC_TIMEOUT; break; }
            const float d0z = (float)sqrt(S[0] / nz), d0l = (float)sqrt(S[1] / nl);
            d1z = (float)sqrt(S[2] / nz);
            d1l = (float)sqrt(S[3] / nl);
            float d0a = 0.f, d1a = 0.f, d0al = 0.f;
            if (a.adj_all) {   // CNF_FLAG_ADJ_CONTROL_ALL (cnf_device.h): a_z and a_logp take part in the heuristic
                double S2[4] = {0, 0, 0, 0};
                adj_all_f0_sums<D, C>(y0, k, al, rtol, atol, counted, S2);
                if (!wg_allreduce4(gs, red, S2, w, lane_r)) { status = CNF_ERR_SYNC_TIMEOUT; break; }
                d0a = (float)sqrt(S2[0] / nz);
                d1a = (float)sqrt(S2[1] / nz);
                d0al = (float)sqrt(S2[2] / nl);
                if (tid == 0) red.keep[0] = d1a;   // read after the next reduction's barriers
            }
            h0 = initial_h0(d0z, d0l, d1z, d1l, a.adj_all != 0, d0a, d1a, d0al);
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
                const float u = (k[c] - kst[0][c][j]) / sc[c];
                if (counted) S[c < D ? 0 : 1] += (double)u * u;
            }
            if (a.adj_all && counted) {
#pragma unroll
                for (int m = 0; m < D; ++m) {
                    const int c = D + 1 + m;
                    const float u = (k[c] - kst[0][c][j]) / (atol + fabsf(y0[c]) * rtol);
                    S[3] += (double)u * u;
                }
            }
            if (!wg_allreduce4(gs, red, S, w, lane_r)) { status = CNF_ERR_SYNC_TIMEOUT; break; }
            const float d2z = (float)sqrt(S[0] / nz) / h0, d2l = (float)sqrt(S[1] / nl) / h0;
            const float d2a = (float)sqrt(S[3] / nz) / h0;
            dt = (double)initial_dt(h0, d1z, d1l, d2z, d2l, a.adj_all != 0, a.adj_all ? red.keep[0] : 0.f, d2a);
            restart = true;
        } else {
#pragma unroll
            for (int c = 0; c < C; ++c) kst[stage][c][j] = k[c];
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
                    for (int s = 0; s < 7; ++s) err += (dtf * kCErr[s]) * kst[s][c][j];
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
                        for (int s = 0; s < 7; ++s) err += (dtf * kCErr[s]) * kst[s][c][j];
                        const float u = err / (atol + rtol * fmaxf(fabsf(y0[c]), fabsf(y1[c])));
                        if (counted) S[3] += (double)u * u;
                    }
                }
                if (!wg_allreduce4(gs, red, S, w, lane_r)) { status = CNF_ERR_SYNC_TIMEOUT; break; }
                if (S[2] > 0.0) { status = CNF_ERR_NONFINITE; break; }
                const float rz = (float)(S[0] / nz), rl = (float)(S[1] / nl);
                const float ra = (float)(S[3] / nz);   // 0 unless CNF_FLAG_ADJ_CONTROL_ALL (a_logp's ratio is always 0)
                const bool accept = (rz <= 1.f) && (rl <= 1.f) && (ra <= 1.f);
                if (accept) {
                    ++nacc;
                    const int sl[7] = {i_first, 1, 2, 3, 4, 5, i_last};
                    const float wts[7] = {wq[0], wq[1], wq[2], wq[3], wq[4], wq[5], wq[6]};
                    wide_commit<D, NS, kWideBwdB16>(region, sl, wts, nacc == 1, tid, part, sh.fold_area());
                    if (!(tau_end > tau + dt)) break;
#pragma unroll
                    for (int c = 0; c < C; ++c) {
                        y0[c] = y1[c];
                        kst[0][c][j] = k[c];   // the seventh stage is the next step's first
                    }
                    const int t_ = i_first;
                    i_first = i_last;
                    i_last = t_;
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
            stage = 1;
            phase = PH_STAGE;
            __syncthreads();   // every wave is past its last read of wq
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
                wq[s] = v;
            }
            __syncthreads();
        }
        float alpha_, bc_[6];
        stage_coeffs(stage, alpha_, bc_);
        ti = t0f + alpha_ * dtf;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            float s_ = 0.f;
#pragma unroll
            for (int s = 0; s < 6; ++s) s_ += (dtf * bc_[s]) * kst[s][c][j];
            yi[c] = y0[c] + s_;
        }
    }

    if (status == 0) {
        const float ta = (float)tau, tb = (float)(tau + dt), te = (float)tau_end;
        const float x = (te - ta) / (tb - ta);
        const float x2 = x * x, x3 = x2 * x, x4 = x3 * x;
#pragma unroll
        for (int m = 0; m < D; ++m) {
            const int c = D + 1 + m;
            float mid = 0.f;
#pragma unroll
            for (int s = 0; s < 7; ++s) mid += (dtf * kCMid[s]) * kst[s][c][j];
            mid = y0[c] + mid;
            const float f0 = kst[0][c][j], f1 = kst[6][c][j], u = y0[c], v = y1[c];
            const float ca = (-2 * dtf) * f0 + (2 * dtf) * f1 + -8 * u + -8 * v + 16 * mid;
            const float cb = (5 * dtf) * f0 + (-3 * dtf) * f1 + 18 * u + 14 * v + -32 * mid;
            const float cc = (-4 * dtf) * f0 + dtf * f1 + -11 * u + -5 * v + 16 * mid;
            const float cd = dtf * f0;
            const float out = ca * x4 + cb * x3 + cc * x2 + cd * x + u;
            if (counted) a.grad_x[row * D + m] = out;
        }
        if (counted) a.grad_logp0[row] = al;
    }
    grid_cleanup(gs, red, tid);
    write_stats_global(a.stats, status, 2 + 6 * (nacc + nrej), nacc, nrej, nrun, (float)dt, a.rows, tid);
}

// ---- the adjoint on a fixed grid (euler / midpoint / rk4 3/8): what fixed_bwd_kernel (cnf_bwd.hip) is at hidden width 64 ----
// Stage st of a step writes its unweighted d/dtheta sums to slot st; the step's commit folds them into the running total
// with the method's weights (wide_commit, as an accepted dopri5 step does).  No step control: no grid-wide reductions.
template <int D, bool HUTCH>
__global__ __launch_bounds__(kThreads) void wide_fixed_bwd_kernel(SolveArgs a, const float* __restrict__ wt) {
    const float res = a.residual ? 1.f : 0.f;   // CNF_DIV_RESIDUAL: dynamics f(z) - z
    constexpr int NS = (HUTCH ? 1 : D) + 1;
    constexpr int C = 2 * D + 1;
    __shared__ WideBwdShared<D, NS, C, kWideBwdB16> sh;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, j = lane & 15;
    sh.stage(a.params, tid);
    __syncthreads();
    const long row = (long)blockIdx.x * kRows + j;
    const bool valid = row < a.rows;
    const bool counted = valid && lane < 16 && w == 0;
    const long rr = valid ? row : a.rows - 1;
    float* region = dbg_ws(a, a.slab + (size_t)blockIdx.x * kWideSlots * kWideSlotF<D>, sizeof(float) * kWideSlots * kWideSlotF<D>, __LINE__);

    float y[C], ee[D];
#pragma unroll
    for (int m = 0; m < D; ++m) {
        y[m] = a.x[rr * D + m];
        ee[m] = HUTCH ? a.e[rr * D + m] : 0.f;
        y[D + 1 + m] = valid ? a.grad_z1[rr * D + m] : 0.f;
    }
    y[D] = a.logp0[rr];
    const float al = valid ? a.grad_logp1[rr] : 0.f;
    const int status = a.after ? a.after->status : 0;   // CnfSolverOpts.after: the solve whose output this one consumes failed
    const float sgn = a.t0 >= a.t1 ? 1.f : -1.f;
    const float tau0 = sgn * a.t1, tau1 = sgn * a.t0;   // the adjoint pass runs from t1 back to t0
    const int nst = a.fixed_kind == CNF_SOLVER_EULER ? 1 : (a.fixed_kind == CNF_SOLVER_MIDPOINT ? 2 : 4);
    int nrun = 0;
    const int n_steps = status == 0 ? a.n_grid : 0;
    for (int step = 0; step < n_steps; ++step) {
        const float ta = a.n_grid == 1 ? tau0 : __fadd_rn(__fmul_rn((float)step, a.step_size), tau0);
        const float tb = (step + 1 == a.n_grid) ? tau1 : __fadd_rn(__fmul_rn((float)(step + 1), a.step_size), tau0);
        const float dt = tb - ta;
        float k1[C], k2[C], k3[C], yi[C];
        float ti = ta;
        float wts[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < C; ++c) yi[c] = y[c];
        for (int st = 0; st < nst; ++st) {
            float zz[D], az[D], f[D], vz[D], div, k[C];
#pragma unroll
            for (int m = 0; m < D; ++m) {
                zz[m] = yi[m];
                az[m] = yi[D + 1 + m];
            }
            float wq;   // weight of this stage's vjp_theta in the step's increment of a_theta
            if (a.fixed_kind == CNF_SOLVER_EULER) wq = dt;
            else if (a.fixed_kind == CNF_SOLVER_MIDPOINT) wq = st == 0 ? 0.f : dt;
            else wq = (st == 0 || st == 3) ? dt / 8 : 3 * (dt / 8);
#pragma unroll
            for (int s_ = 0; s_ < 4; ++s_) wts[s_] = st == s_ ? wq : wts[s_];
            aug_eval_sh<D, HUTCH, kWideFixedFact>(sh, a.params, wt, lane, w, sgn * ti, zz, ee, az, al, sgn, wq != 0.f, f, div, vz,
                                  region + (size_t)st * kWideSlotF<D>, 0);
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
            if (a.fixed_kind == CNF_SOLVER_EULER) {
#pragma unroll
                for (int c = 0; c < C; ++c) y[c] = y[c] + dt * k[c];
            } else if (a.fixed_kind == CNF_SOLVER_MIDPOINT) {
                if (st == 0) {
                    ti = ta + dt / 2;
#pragma unroll
                    for (int c = 0; c < C; ++c) yi[c] = y[c] + k[c] * (dt / 2);
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
        const int sl[7] = {0, 1, 2, 3, 4, 5, 6};
        wide_commit<D, NS, kWideFixedFact>(region, sl, wts, step == 0, tid, 0, sh.fold_area());   // slots with weight zero are not read
    }
    if (status == 0 && counted) {
#pragma unroll
        for (int m = 0; m < D; ++m) a.grad_x[row * D + m] = y[D + 1 + m];
        a.grad_logp0[row] = al;
    }
    write_stats_global(a.stats, status, nst * a.n_grid, a.n_grid, 0, nrun, 0.f, a.rows, tid);
}

// sum slot `which` of every workgroup region in a fixed order and undo the block layout: out[canonical parameter index]
template <int D>
__global__ void wide_reduce_kernel(const float* __restrict__ regions, int nwg, int which, int valid, float* __restrict__ out,
                                   int spread) {
    const ParamLayout L = make_layout(D, kWH);
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= L.total) return;
    int src;
    const int m1 = L.w(1), m2 = L.w(2);
    if (p >= m1 && p < m1 + kWH * kWH) {
        const int o = (p - m1) / kWH, i = (p - m1) % kWH;
        src = ((0 * kWT + (o >> 4)) * kWT + (i >> 4)) * 256 + ((((o & 15) >> 2) * 16 + (i & 15)) * 4) + (o & 3);
    } else if (p >= m2 && p < m2 + kWH * kWH) {
        const int o = (p - m2) / kWH, i = (p - m2) % kWH;
        src = ((1 * kWT + (o >> 4)) * kWT + (i >> 4)) * 256 + ((((o & 15) >> 2) * 16 + (i & 15)) * 4) + (o & 3);
    } else {
        const int si = p < m1 ? p : (p < m2 ? p - kWH * kWH : p - 2 * kWH * kWH);
        src = kWideBig + si;
    }
    // spread launches: tile g's sums are in region 2g (parameters from W3 on) or 2g + 1 (those before W3)
    const size_t stride = (size_t)(spread ? 2 : 1) * kWideSlots * kWideSlotF<D>;
    const float* base = regions + (size_t)which * kWideSlotF<D> + src + (spread && p < m2 ? (size_t)kWideSlots * kWideSlotF<D> : 0);
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int g = 0;
    for (; g + 3 < nwg; g += 4) {
        s0 += base[(size_t)g * stride];
        s1 += base[(size_t)(g + 1) * stride];
        s2 += base[(size_t)(g + 2) * stride];
        s3 += base[(size_t)(g + 3) * stride];
    }
    for (; g < nwg; ++g) s0 += base[(size_t)g * stride];
    out[p] = valid ? (s0 + s1) + (s2 + s3) : 0.f;
}

// ---- launchers --------------------------------------------------------------------------------------------------------
size_t wide_bwd_workspace_floats(int D, long rows) {
    long nwg = (rows + kRows - 1) / kRows;
    if (2 * nwg <= kMaxWG) nwg *= 2;   // room for a CNF_FLAG_SPREAD launch
    const size_t sf = D == 1 ? (size_t)kWideSlotF<1> : (size_t)kWideSlotF<2>;
    return kWideBwdHeadFloats + (size_t)nwg * kWideSlots * sf;
}

static hipError_t wide_prep(int D, const float* params, float* wt, hipStream_t s) {
    if (kWideBwdB16)
        hipLaunchKernelGGL(wide_bwd_pieces_kernel, dim3(2 * 2 * kWT * (kWT / 2) * 64 / 256), dim3(256), 0, s, params, D,
                           reinterpret_cast<u32x4*>(wt));
    else
        hipLaunchKernelGGL(wide_transpose_kernel, dim3(kWH / 32, kWH / 32, 2), dim3(32, 8), 0, s, params, D, wt);
    return hipGetLastError();
}

template <int D, bool HUTCH>
static hipError_t wide_bwd_t(const SolveArgs& a0, float* ws_floats, hipStream_t s) {
    SolveArgs a = a0;
    float* wt = ws_floats;
    a.slab = ws_floats + kWideBwdHeadFloats;
    hipError_t err = wide_prep(D, a.params, wt, s);
    if (err != hipSuccess) return err;
    const int grid = (int)((a.rows + kRows - 1) / kRows);
    {
        const hipError_t res = check_resident(wide_bwd_kernel<D, HUTCH>, kThreads, a.spread ? 2 * grid : grid);
        if (res != hipSuccess) return res;
    }
    dbg_timing_begin(s);
    hipLaunchKernelGGL((wide_bwd_kernel<D, HUTCH>), dim3(a.spread ? 2 * grid : grid), dim3(kThreads), 0, s, a, (const float*)wt);
    dbg_timing_end(s);
    err = hipGetLastError();
    if (err != hipSuccess) return err;
    const int P = make_layout(D, kWH).total;
    hipLaunchKernelGGL((wide_reduce_kernel<D>), dim3((P + 255) / 256), dim3(256), 0, s, (const float*)a.slab, grid, kWideTotal, 1,
                       a.grad_params, a.spread);
    return hipGetLastError();
}

hipError_t launch_wide_bwd(int D, bool hutch, const SolveArgs& a, float* ws_floats, hipStream_t s) {
    if (D == 1 && !hutch) return wide_bwd_t<1, false>(a, ws_floats, s);
    if (D == 2 && !hutch) return wide_bwd_t<2, false>(a, ws_floats, s);
    if (D == 2 && hutch) return wide_bwd_t<2, true>(a, ws_floats, s);
    return hipErrorNotSupported;
}

template <int D, bool HUTCH>
static hipError_t wide_fixed_bwd_t(const SolveArgs& a0, float* ws_floats, hipStream_t s) {
    SolveArgs a = a0;
    float* wt = ws_floats;
    a.slab = ws_floats + kWideBwdHeadFloats;
    hipError_t err = wide_prep(D, a.params, wt, s);
    if (err != hipSuccess) return err;
    const int grid = (int)((a.rows + kRows - 1) / kRows);
    dbg_timing_begin(s);
    hipLaunchKernelGGL((wide_fixed_bwd_kernel<D, HUTCH>), dim3(grid), dim3(kThreads), 0, s, a, (const float*)wt);
    dbg_timing_end(s);
    err = hipGetLastError();
    if (err != hipSuccess) return err;
    const int P = make_layout(D, kWH).total;
    hipLaunchKernelGGL((wide_reduce_kernel<D>), dim3((P + 255) / 256), dim3(256), 0, s, (const float*)a.slab, grid, kWideTotal, 1,
                       a.grad_params, 0);
    return hipGetLastError();
}

// fixed-grid adjoint at hidden width 256 (a.fixed_kind, a.n_grid, a.step_size), rows <= 4096
hipError_t launch_wide_fixed_bwd(int D, bool hutch, const SolveArgs& a, float* ws_floats, hipStream_t s) {
    if (D == 1 && !hutch) return wide_fixed_bwd_t<1, false>(a, ws_floats, s);
    if (D == 2 && !hutch) return wide_fixed_bwd_t<2, false>(a, ws_floats, s);
    if (D == 2 && hutch) return wide_fixed_bwd_t<2, true>(a, ws_floats, s);
    return hipErrorNotSupported;
}

template <int D, bool HUTCH>
static hipError_t wide_aug_t(const SolveArgs& a0, float* ws_floats, float t, const float* z, const float* a_z, const float* a_l,
                             float* f_out, float* nd_out, float* vz_out, hipStream_t s) {
    SolveArgs a = a0;
    float* wt = ws_floats;
    a.slab = ws_floats + kWideBwdHeadFloats;
    hipError_t err = wide_prep(D, a.params, wt, s);
    if (err != hipSuccess) return err;
    const int grid = (int)((a.rows + kRows - 1) / kRows);
    hipLaunchKernelGGL((wide_aug_kernel<D, HUTCH>), dim3(grid), dim3(kThreads), 0, s, a, (const float*)wt, t, z, a_z, a_l, f_out,
                       nd_out, vz_out);
    err = hipGetLastError();
    if (err != hipSuccess) return err;
    const int P = make_layout(D, kWH).total;
    hipLaunchKernelGGL((wide_reduce_kernel<D>), dim3((P + 255) / 256), dim3(256), 0, s, (const float*)a.slab, grid, 0, 1,
                       a.grad_params, 0);
    return hipGetLastError();
}

hipError_t launch_wide_aug(int D, bool hutch, const SolveArgs& a, float* ws_floats, float t, const float* z, const float* a_z,
                           const float* a_l, float* f_out, float* nd_out, float* vz_out, hipStream_t s) {
    if (D == 1 && !hutch) return wide_aug_t<1, false>(a, ws_floats, t, z, a_z, a_l, f_out, nd_out, vz_out, s);
    if (D == 2 && !hutch) return wide_aug_t<2, false>(a, ws_floats, t, z, a_z, a_l, f_out, nd_out, vz_out, s);
    if (D == 2 && hutch) return wide_aug_t<2, true>(a, ws_floats, t, z, a_z, a_l, f_out, nd_out, vz_out, s);
    return hipErrorNotSupported;
}

}  // namespace cnf
