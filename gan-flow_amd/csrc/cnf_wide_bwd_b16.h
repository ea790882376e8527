// cnf_wide_bwd_b16.h -- one augmented evaluation at hidden width 256 with every 256 x 256 product on the bf16 matrix
// pipe (three-piece operands, cnf_device.h): the two forward passes, the two transposed passes and the outer products of
// d/dtheta.  Included by cnf_wide_bwd.hip; same decomposition, same saves, same small-parameter sums, same slot layout as
// aug_eval_wide there -- what changes is how the matrices and the layer inputs / cotangents reach the matrix cores:
//   * the weights: W and W^T of both hidden layers as bf16 pieces in operand order, written once per call by
//     wide_bwd_pieces_kernel into the head of the workspace (4 x 384 KB), streamed from L2 one k-step ahead;
//   * layer inputs and cotangents: published by the wave that owns them as pieces (the B operand of
//     v_mfma_f32_16x16x32_bf16 as they lie), two buffers of NS x 24 KB; the outer products read the SAME buffers
//     transposed with ds_read_b64_tr_b16 (rows along k; K = 32 holds the 16 rows twice, so one instruction adds two of the
//     six piece products: outer_b16 in cnf_split.h is the width-64 form of this).
// LDS: two piece buffers are 144 KB at three streams, so the fp32 kernel's LDS copies of the gate vectors and the four
// per-wave slope arrays do not fit beside them: the gate vectors are read from the parameter vector (L2) where they are
// used, and the slopes are ONE array every wave writes identical values to (see WideBwdShared in cnf_wide_bwd.hip).
#pragma once
#include "cnf_split.h"
#include "cnf_wide.h"

namespace cnf {

constexpr size_t kWideSetBytes = kWidePieceBytes;                 // pieces of W2, W3 (786 432 bytes); W2^T, W3^T: as many
constexpr size_t kWideSetVecs = kWideSetBytes / 16;               // 16-byte operands per set
constexpr size_t kWideLayerVecs = (size_t)kWT * (kWT / 2) * 3 * 64;

template <int D, int NS>
struct alignas(16) WideBwdLds {
    uint32_t xb16[2][NS][3][kWT / 2][64][4];   // [buffer][stream][piece][k-step][lane][tile & 1][2]
    float part[kNW][2 * D][kRows];
    float part2[kNW][D][kRows];
    float w1[D][kWH];
    float w4[D][kWH];
    float vec4[4][16];
    float gate[kNW][3][2][kWF];                // gates of every wave's 64 features
    float gate4[kNW][2][16];
};

template <int D, int NS>
__device__ inline void wide_bwd_stage(WideBwdLds<D, NS>& sh, const float* __restrict__ p, int tid) {
    const ParamLayout L = make_layout(D, kWH);
    for (int idx = tid; idx < D * kWH; idx += kThreads) {
        const int m = idx / kWH, i = idx % kWH;
        sh.w1[m][i] = p[L.w(0) + i * D + m];
        sh.w4[m][i] = p[L.w(3) + m * kWH + i];
    }
    if (tid < 16) {
        const bool in = tid < D;
        sh.vec4[0][tid] = in ? p[L.b(3) + tid] : 0.f;
        sh.vec4[1][tid] = in ? p[L.hb(3) + tid] : 0.f;
        sh.vec4[2][tid] = in ? p[L.wg(3) + tid] : 0.f;
        sh.vec4[3][tid] = in ? p[L.bg(3) + tid] : 0.f;
    }
}

// W (set 0) and W^T (set 1) of the two hidden layers as three bf16 pieces each, in the A-operand order of the products:
// out[set][layer][Tout][k-step][piece][lane] = 8 bf16 of M[16 Tout + (lane & 15)][16 (2ks + (e >> 2)) + 4 (lane >> 4) + (e & 3)],
// M = W_l (set 0) or its transpose (set 1).  Once per call, ~1.5 MB written.
__global__ __launch_bounds__(256) void wide_bwd_pieces_kernel(const float* __restrict__ params, int D, u32x4* __restrict__ out) {
    const int idx = blockIdx.x * 256 + threadIdx.x;   // [set][layer][Tout][k-step][lane]
    if (idx >= 2 * 2 * kWT * (kWT / 2) * 64) return;
    const int lane = idx & 63, ks = (idx >> 6) & 7, To = (idx >> 9) & 15, l = (idx >> 13) & 1, set = idx >> 14;
    const float* W = params + make_layout(D, kWH).w(1 + l);
    const int row = 16 * To + (lane & 15);
    uint32_t pc[8][3];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int col = 16 * (2 * ks + (e >> 2)) + 4 * (lane >> 4) + (e & 3);
        split3_bf16(set == 0 ? W[(size_t)row * kWH + col] : W[(size_t)col * kWH + row], pc[e]);
    }
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        const u32x4 v = {(pc[0][p] >> 16) | pc[1][p], (pc[2][p] >> 16) | pc[3][p], (pc[4][p] >> 16) | pc[5][p], (pc[6][p] >> 16) | pc[7][p]};
        out[(size_t)set * kWideSetVecs + ((size_t)((l * kWT + To) * (kWT / 2) + ks) * 3 + p) * 64 + lane] = v;
    }
}

// the wave's 4 values of feature tile tg, one stream, as pieces into the buffer (publish_b16's layout at 8 k-steps)
__device__ __forceinline__ void wide_publish(WideStreamBlk& blk, int tg, int lane, const f32x4& v) {
    uint32_t pc[4][3];
#pragma unroll
    for (int r = 0; r < 4; ++r) split3_bf16(v[r], pc[r]);
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        uint2 o;
        o.x = (pc[0][p] >> 16) | pc[1][p];
        o.y = (pc[2][p] >> 16) | pc[3][p];
        *reinterpret_cast<uint2*>(&blk[p][tg >> 1][lane][2 * (tg & 1)]) = o;
    }
}

// byte offset inside one stream block of the 8 bytes lane (q, j) supplies to ds_read_b64_tr_b16 for: piece, feature tile
// T (0..15), read rr (tr_piece_offset in cnf_split.h with this buffer's strides: 8 KB per piece, 1 KB per k-step)
__device__ __forceinline__ uint32_t wide_tr_offset(int piece, int T, int q, int j, int rr) {
    const int row = 8 * (q & 1) + 4 * rr + (j >> 2);
    return (uint32_t)(piece * 8192 + (T >> 1) * 1024 + (16 * (j & 3) + row) * 16 + (T & 1) * 8);
}
__device__ __forceinline__ bf16x8 wide_tr_operand(const char* base, int p0, int p1, int T, int q, int j) {
    const int piece = (q >> 1) ? p1 : p0;
    const s16x4 lo = lds_read_tr16(base, wide_tr_offset(piece, T, q, j, 0));
    const s16x4 hi = lds_read_tr16(base, wide_tr_offset(piece, T, q, j, 1));
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
}

// dW[16 To + i'][16 Ti + j'] = w * sum over rows and streams of cot_s[i'][row] x_s[j'][row] for this wave's four output
// tiles To = 4 wave + T and all sixteen Ti: unweighted 16 x 16 blocks stored to the slot (the layout wide_reduce_kernel
// undoes).  Two output tiles share every input-tile operand.  All 64 lanes must call it.
template <int NS>
__device__ __forceinline__ void wide_outer_b16(const WideStreamBlk* cot, const WideStreamBlk* xin, int wave, int lane, float w,
                                               float* __restrict__ blk) {
    const int q = lane >> 4, j = lane & 15;
#pragma unroll 1
    for (int Tp = 0; Tp < kWTW; Tp += 2) {
        bf16x8 A_hh[2][NS], A_mm[2][NS], A_hl[2][NS];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const char* ca = reinterpret_cast<const char*>(&cot[s][0][0][0][0]);
                const int To = kWTW * wave + Tp + u;
                A_hh[u][s] = wide_tr_operand(ca, 0, 0, To, q, j);
                A_mm[u][s] = wide_tr_operand(ca, 1, 1, To, q, j);
                A_hl[u][s] = wide_tr_operand(ca, 0, 2, To, q, j);
            }
#pragma unroll 2
        for (int Ti = 0; Ti < kWT; ++Ti) {
            bf16x8 B_hm[NS], B_lh[NS];
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const char* xa = reinterpret_cast<const char*>(&xin[s][0][0][0][0]);
                B_hm[s] = wide_tr_operand(xa, 0, 1, Ti, q, j);
                B_lh[s] = wide_tr_operand(xa, 2, 0, Ti, q, j);
            }
            f32x4 d[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
            // smallest products first; consecutive MFMAs go to different accumulators
#pragma unroll
            for (int s = 0; s < NS; ++s)
#pragma unroll
                for (int u = 0; u < 2; ++u) d[u] = mfma_b16(A_hl[u][s], B_lh[s], d[u]);   // h.l + l.h
#pragma unroll
            for (int s = 0; s < NS; ++s)
#pragma unroll
                for (int u = 0; u < 2; ++u) d[u] = mfma_b16(A_mm[u][s], B_hm[s], d[u]);   // m.h + m.m
#pragma unroll
            for (int s = 0; s < NS; ++s)
#pragma unroll
                for (int u = 0; u < 2; ++u) d[u] = mfma_b16(A_hh[u][s], B_hm[s], d[u]);   // h.h + h.m
#pragma unroll
            for (int u = 0; u < 2; ++u) {
#ifdef CNF_EXP_WIDE_NO_STORE   // timing experiment: the outer products without their stores (w is +-1)
                if (w > 2.f)
#endif
                *reinterpret_cast<f32x4*>(blk + ((size_t)(Tp + u) * kWT + Ti) * 256) = w * d[u];
            }
        }
    }
}

// ---- d/dtheta of the two matrices as FACTORS ----------------------------------------------------------------------------
// An evaluation's contribution to dW is sum_rows,streams cot (x) x: rank 16 NS, yet as 16 x 16 result blocks it is
// 256 x 256 floats per layer -- 262 KB stored per evaluation and read back by the step's fold, which made slot traffic half
// of a launch (measured: stores 0.29 ms + folds 0.42 ms of 1.47 ms).  The solve kernels therefore store the FACTORS (the
// cotangents and the layer inputs as they are published, fp32, [cot | x][stream][feature][row]: 98 KB per layer at three
// streams) and an accepted step forms sum_stages weight_s cot_s (x) x_s itself, on the matrix cores, K = the stages' rows,
// with the whole 256 x 256 result in registers (64 blocks per wave: the evaluation's saves are dead by then), and adds it
// to the workgroup's running total: one read-modify-write of the total per step instead of a slot per stage.
template <int NS>
constexpr int kWideFactLayer = 2 * NS * kWH * kRows;   // floats of one layer's factors in a stage slot
template <int NS>
__device__ __forceinline__ void wide_store_factor(float* __restrict__ slot, int lc, int which, int s, int tg, int lane, const f32x4& v) {
    float* dst = slot + (size_t)lc * kWideFactLayer<NS> + (size_t)(which * NS + s) * (kWH * kRows) + (16 * tg + 4 * (lane >> 4)) * kRows + (lane & 15);
#ifdef CNF_EXP_WIDE_NO_STORE   // timing experiment: no factor stores
    if (v[0] == 1.2345e-30f)
#endif
#pragma unroll
    for (int r = 0; r < 4; ++r) dst[r * kRows] = v[r];
}
// The fold of an accepted step.  Per stage with a non-zero weight: the 256 threads bring the layer's factors (96 KB at
// three streams: 8 NS sixteen-byte loads per thread, all in flight at once, requested one stage ahead) into LDS AS PIECES
// in operand order -- pl[cot | x][stream][piece][row half][feature] x 16 bytes = 8 rows of one feature: exactly what lane
// (feature, q) of a 16x16x32 operand holds, K = 32 being the 16 rows twice -- the cotangents scaled by the stage's weight
// on the way; then every wave multiplies its four output tiles with all sixteen input tiles straight out of LDS (one
// ds_read_b128 per operand, three MFMAs per block and stream: h.l + l.h, m.h + m.m, h.h + h.m).  The piece area aliases the
// evaluation's two exchange buffers (same size; no evaluation is in flight during a fold).
template <int NS>
struct WideFoldLds {
    uint32_t pl[2][NS][3][2][kWH][4];
};
template <int NS>
__device__ __forceinline__ void wide_fold_load(const float* __restrict__ fs, int tid, f32x4 (&raw)[4 * NS][2]) {
#pragma unroll
    for (int k = 0; k < 4 * NS; ++k) {   // 2 * NS * 256 * 2 eight-row groups over 256 threads
        const int g = tid + kThreads * k;
        const float* src = fs + (size_t)(g >> 9) * (kWH * kRows) + ((g >> 1) & 255) * kRows + 8 * (g & 1);
        raw[k][0] = *reinterpret_cast<const f32x4*>(src);
        raw[k][1] = *reinterpret_cast<const f32x4*>(src + 4);
    }
}
template <int NS>
__device__ __forceinline__ void wide_fold_stage(WideFoldLds<NS>& fl, int tid, const f32x4 (&raw)[4 * NS][2], float weight) {
#pragma unroll
    for (int k = 0; k < 4 * NS; ++k) {
        const int g = tid + kThreads * k;
        const int ws = g >> 9, f = (g >> 1) & 255, h = g & 1;   // ws = which * NS + stream (which: 0 cot, 1 x)
        const float scale = ws < NS ? weight : 1.0f;
        uint32_t pc[8][3];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            split3_bf16(raw[k][0][r] * scale, pc[r]);
            split3_bf16(raw[k][1][r] * scale, pc[4 + r]);
        }
#pragma unroll
        for (int p = 0; p < 3; ++p)
            *reinterpret_cast<u32x4*>(&fl.pl[0][0][0][0][0][0] + ((size_t)((ws * 3 + p) * 2 + h) * kWH + f) * 4) =
                u32x4{(pc[0][p] >> 16) | pc[1][p], (pc[2][p] >> 16) | pc[3][p], (pc[4][p] >> 16) | pc[5][p], (pc[6][p] >> 16) | pc[7][p]};
    }
}
// total[layer lc] (+)= sum_s wts[s] * cot_s (x) x_s over the stage slots sl[s] of `region`; block layout of wide_reduce_kernel.
// All threads of the workgroup call it; it starts and ends with the LDS area free (the caller's barriers).
template <int NS>
__device__ __forceinline__ void wide_fold_factors(WideFoldLds<NS>& fl, const float* __restrict__ region, size_t slot_floats,
                                                  const int (&sl)[7], const float (&wts)[7], int lc, bool first,
                                                  float* __restrict__ tot, int tid) {
    const int wave = tid >> 6, lane = tid & 63, q = lane >> 4, j = lane & 15;
    const int hi = q >> 1, h = q & 1;   // second half of K: the other piece of the pair; rows 8h .. 8h + 7
    f32x4 acc[kWTW][kWT];
#pragma unroll
    for (int T = 0; T < kWTW; ++T)
#pragma unroll
        for (int Ti = 0; Ti < kWT; ++Ti) acc[T][Ti] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto next_active = [&](int s) {
        while (s < 7 && wts[s] == 0.f) ++s;
        return s;
    };
    auto stage_ptr = [&](int s) { return region + (size_t)sl[s] * slot_floats + (size_t)lc * kWideFactLayer<NS>; };
    auto piece = [&](int which, int st, int p, int f) -> bf16x8 {
        return __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(&fl.pl[which][st][p][h][f][0]));
    };
    f32x4 raw[4 * NS][2];
    int s = next_active(0);
    if (s < 7) wide_fold_load<NS>(stage_ptr(s), tid, raw);
#pragma unroll 1
    while (s < 7) {
        wide_fold_stage<NS>(fl, tid, raw, wts[s]);
        const int sn = next_active(s + 1);
        if (sn < 7) wide_fold_load<NS>(stage_ptr(sn), tid, raw);   // in flight while the matrix cores work
        __syncthreads();
#pragma unroll 1
        for (int st = 0; st < NS; ++st) {
            bf16x8 A_hh[kWTW], A_mm[kWTW], A_hl[kWTW];
#pragma unroll
            for (int T = 0; T < kWTW; ++T) {
                const int f = 16 * (kWTW * wave + T) + j;
                A_hh[T] = piece(0, st, 0, f);
                A_mm[T] = piece(0, st, 1, f);
                A_hl[T] = piece(0, st, hi ? 2 : 0, f);
            }
#pragma unroll
            for (int Ti = 0; Ti < kWT; ++Ti) {
                const int f = 16 * Ti + j;
                const bf16x8 B_hm = piece(1, st, hi ? 1 : 0, f), B_lh = piece(1, st, hi ? 0 : 2, f);
#pragma unroll
                for (int T = 0; T < kWTW; ++T) acc[T][Ti] = mfma_b16(A_hl[T], B_lh, acc[T][Ti]);   // h.l + l.h
#pragma unroll
                for (int T = 0; T < kWTW; ++T) acc[T][Ti] = mfma_b16(A_mm[T], B_hm, acc[T][Ti]);   // m.h + m.m
#pragma unroll
                for (int T = 0; T < kWTW; ++T) acc[T][Ti] = mfma_b16(A_hh[T], B_hm, acc[T][Ti]);   // h.h + h.m
            }
        }
        __syncthreads();   // the area is free for the next stage's pieces (and, after the last one, for the caller)
        s = sn;
    }
    float* blk = tot + ((size_t)(lc * kWT + kWTW * wave) * kWT) * 256 + lane * 4;
#pragma unroll
    for (int T = 0; T < kWTW; ++T)
#pragma unroll
        for (int Ti = 0; Ti < kWT; ++Ti) {
            f32x4* dst = reinterpret_cast<f32x4*>(blk + ((size_t)T * kWT + Ti) * 256);
            *dst = first ? acc[T][Ti] : *dst + acc[T][Ti];
        }
}

// ---- one augmented evaluation (the contract of aug_eval_wide, cnf_wide_bwd.hip) ----------------------------------------
// wpc: wide_bwd_pieces_kernel's output.  slot: this evaluation's d/dtheta slot, written only when with_theta.
// FACT: the solve kernels -- store the factors of dW2 / dW3 (wide_store_factor) instead of forming result blocks.
template <int D, bool HUTCH, int NS, bool FACT>
__device__ __forceinline__ void aug_eval_wide_b16(WideBwdLds<D, NS>& sh, const float* __restrict__ params,
                                                  const u32x4* __restrict__ wpc, int lane, int wave, float t,
                                                  const float (&z)[D], const float (&e)[D], const float (&az)[D], float al,
                                                  float w, bool with_theta, float (&f)[D], float& div, float (&vz)[D],
                                                  float* __restrict__ slot, int part = 0) {
    constexpr int NT = HUTCH ? 1 : D;
    static_assert(NS == NT + 1, "streams");
    using SI = WideSmallIdx<D>;
    const ParamLayout L = make_layout(D, kWH);
    const bool th_hi = with_theta && part != 2, th_lo = with_theta && part != 1;
    const int q = lane >> 4, j = lane & 15, qq = j & 3;
    float* slot_small = slot + kWideBig;
    float (&gate)[3][2][kWF] = sh.gate[wave];
    float (&gate4)[2][16] = sh.gate4[wave];
    {
        const int fi = kWF * wave + lane;
#pragma unroll
        for (int l = 0; l < 3; ++l) {
            const float b0 = params[L.b(l) + fi], hb = params[L.hb(l) + fi], wg = params[L.wg(l) + fi], bg = params[L.bg(l) + fi];
            const float g = sigmoid_f(fmaf(wg, t, bg));
            gate[l][0][lane] = g;
            gate[l][1][lane] = fmaf(g, b0, hb * t);
        }
        if (lane < 16) {
            const float g = sigmoid_f(fmaf(sh.vec4[2][lane], t, sh.vec4[3][lane]));
            gate4[0][lane] = g;
            gate4[1][lane] = fmaf(g, sh.vec4[0][lane], sh.vec4[1][lane] * t);
        }
        wave_lds_fence();
    }

    // ------------------------------------------------------------------ forward with saves (own 4 tiles)
    f32x4 xs[3][kWTW], ub[3][kWTW], U[2][NT][kWTW], act[kWTW][NS], w1c[D][kWTW];
#pragma unroll
    for (int T = 0; T < kWTW; ++T) {
        const int lo = 16 * T + 4 * q, fo = kWF * wave + lo;
        const f32x4 g = *reinterpret_cast<const f32x4*>(&gate[0][0][lo]);
        const f32x4 c = *reinterpret_cast<const f32x4*>(&gate[0][1][lo]);
        const f32x4 b = *reinterpret_cast<const f32x4*>(params + L.b(0) + fo);
        f32x4 a = {0.f, 0.f, 0.f, 0.f}, au = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int m = 0; m < D; ++m) {
            w1c[m][T] = *reinterpret_cast<const f32x4*>(&sh.w1[m][fo]);
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
        WideStreamBlk* xb = sh.xb16[l];
#pragma unroll
        for (int T = 0; T < kWTW; ++T)
#pragma unroll
            for (int s = 0; s < NS; ++s) wide_publish(xb[s], kWTW * wave + T, lane, act[T][s]);
        if (FACT && (l == 1 ? th_hi : th_lo)) {   // the inputs of hidden layer l + 2: the x factor of its dW
#pragma unroll
            for (int T = 0; T < kWTW; ++T)
#pragma unroll
                for (int s = 0; s < NS; ++s) wide_store_factor<NS>(slot, l, 1, s, kWTW * wave + T, lane, act[T][s]);
        }
        const u32x4* wp = wpc + ((size_t)(l * kWT + kWTW * wave) * (kWT / 2) * 3) * 64 + lane;
        WideRing wb;
        wide_wp_first(wp, wb);
        f32x4 bias[kWTW];
#pragma unroll
        for (int T = 0; T < kWTW; ++T) bias[T] = *reinterpret_cast<const f32x4*>(params + L.b(l + 1) + kWF * wave + 16 * T + 4 * q);
        __syncthreads();
        f32x4 a4[kWTW][NS];
        wide_matmul_b16<NS>(xb, wp, wb, lane, a4);
#pragma unroll
        for (int T = 0; T < kWTW; ++T) {
            const int lo = 16 * T + 4 * q;
            const f32x4 g = *reinterpret_cast<const f32x4*>(&gate[l + 1][0][lo]);
            const f32x4 c = *reinterpret_cast<const f32x4*>(&gate[l + 1][1][lo]);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float x = act_f(fmaf(g[r], a4[T][0][r], c[r]));
                const float dg = act_d(x) * g[r];
                xs[l + 1][T][r] = x;
                ub[l + 1][T][r] = a4[T][0][r] + bias[T][r];
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
            w4[m][T] = *reinterpret_cast<const f32x4*>(&sh.w4[m][kWF * wave + 16 * T + 4 * q]);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                pf = fmaf(w4[m][T][r], act[T][0][r], pf);
                pd = fmaf(w4[m][T][r], act[T][HUTCH ? 1 : 1 + m][r], pd);
            }
        }
        pf = xq_sum(pf);
        pd = xq_sum(pd);
        if (q == 0) {
            sh.part[wave][m][j] = pf;
            sh.part[wave][D + m][j] = pd;
        }
    }
    __syncthreads();
    float g4[D], u4[D], U4[D];
    div = 0.f;
#pragma unroll
    for (int m = 0; m < D; ++m) {
        g4[m] = gate4[0][m];
        u4[m] = wsum4(sh.part[0][m][j], sh.part[1][m][j], sh.part[2][m][j], sh.part[3][m][j]);
        U4[m] = wsum4(sh.part[0][D + m][j], sh.part[1][D + m][j], sh.part[2][D + m][j], sh.part[3][D + m][j]);
        f[m] = fmaf(g4[m], u4[m], gate4[1][m]);
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
                float xb_ = 0.f, Xh = 0.f;
#pragma unroll
                for (int m = 0; m < D; ++m) {
                    xb_ = fmaf(ubar[m], w4[m][T][r], xb_);
                    if (HUTCH) Xh = fmaf(Ubar[m], w4[m][T][r], Xh);
                    else cot[1 + (HUTCH ? 0 : m)][T][r] = Ubar[m] * w4[m][T][r];
                }
                cot[0][T][r] = xb_;
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
                const float gb = row16_allsum_w(lf[m] * (u4[m] + sh.vec4[0][m]) + ld * (HUTCH ? e[m] : 1.0f) * U4[m]);
                const float gg = g4[m] * (1.0f - g4[m]);
                const float val = qq < 2 ? s4 : gb;
                const float fac = qq == 0 ? g4[m] : (qq == 1 ? t : (qq == 2 ? t * gg : gg));
                if (wave == 0 && lane < 4) slot_small[SI::l4 + kWH * D + qq * D + m] = w * fac * val;
            }
        }
    }

#pragma unroll
    for (int l = 1; l >= -1; --l) {
        const int Lh = l + 1;
#pragma unroll
        for (int T = 0; T < kWTW; ++T) {
            const int lo = 16 * T + 4 * q, fo = kWF * wave + lo;
            const f32x4 g = *reinterpret_cast<const f32x4*>(&gate[Lh][0][lo]);
            float s1[4], s2[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float x = xs[Lh][T][r];
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
                s2[r] = fmaf(pre, ub[Lh][T][r], q2);
                cot[0][T][r] = g[r] * pre;
            }
            if (Lh == 2 ? th_hi : th_lo) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float st = row16_allsum_w(s1[r]);
                    const float gt = row16_allsum_w(s2[r]);
                    const float gg = g[r] * (1.0f - g[r]);
                    const float val = qq < 2 ? st : gt;
                    const float fac = qq == 0 ? g[r] : (qq == 1 ? t : (qq == 2 ? t * gg : gg));
                    if (j < 4) slot_small[SI::vec(Lh) + qq * kWH + fo + r] = w * fac * val;
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
                for (int s = 0; s < NS; ++s) wide_publish(sh.xb16[0][s], kWTW * wave + T, lane, cot[s][T]);
            if (FACT && (lc == 1 ? th_hi : th_lo)) {
#pragma unroll
                for (int T = 0; T < kWTW; ++T)
#pragma unroll
                    for (int s = 0; s < NS; ++s) {
                        const f32x4 cw = w * cot[s][T];   // w = +-1: the time direction
                        wide_store_factor<NS>(slot, lc, 0, s, kWTW * wave + T, lane, cw);
                    }
            }
            if (!FACT && lc == 0 && th_lo) {
#pragma unroll
                for (int T = 0; T < kWTW; ++T) {
                    const f32x4 gi = *reinterpret_cast<const f32x4*>(&gate[0][0][16 * T + 4 * q]);
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
                        wide_publish(sh.xb16[1][s], kWTW * wave + T, lane, xin);
                    }
                }
            }
            // transposed pieces, rows = this wave's INPUT features, streamed like the forward weights
            const u32x4* wp = wpc + kWideSetVecs + ((size_t)(lc * kWT + kWTW * wave) * (kWT / 2) * 3) * 64 + lane;
            WideRing wb;
            wide_wp_first(wp, wb);
            __syncthreads();
#ifdef CNF_EXP_WIDE_NO_OUTER   // timing experiment: no outer products at all
            if (w > 2.f)
#else
            if (!FACT && (lc == 1 ? th_hi : th_lo))
#endif
                wide_outer_b16<NS>(sh.xb16[0], sh.xb16[1], wave, lane, w,
                                   slot + ((size_t)(lc * kWT + kWTW * wave) * kWT) * 256 + lane * 4);
            f32x4 o4[kWTW][NS];
            wide_matmul_b16<NS>(sh.xb16[0], wp, wb, lane, o4);
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
                if (q == 0) sh.part2[wave][m][j] = pz;
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
            for (int m = 0; m < D; ++m)
                vz[m] = wsum4(sh.part2[0][m][j], sh.part2[1][m][j], sh.part2[2][m][j], sh.part2[3][m][j]);
        }
    }
}

}  // namespace cnf
