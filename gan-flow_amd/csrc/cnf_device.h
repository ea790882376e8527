// cnf_device.h -- device-side building blocks shared by the forward and adjoint kernels (gfx950 only).
//
// Work decomposition (DESIGN.md "Data layout"):
//   * one wave64 owns 16 batch rows ("a tile") and ALL streams of those rows: the value stream and
//     the forward-mode tangent streams (d of them for the exact divergence, 1 for Hutchinson);
//   * a hidden activation vector lives in registers in the C/D layout of v_mfma_f32_16x16x4_f32:
//     lane = (q, j), q = lane>>4, j = lane&15 = row inside the tile; register (T, r) of the lane holds
//     feature 16*T + 4*q + r of row j.  That layout is at the same time a valid B operand of the next
//     layer's MFMA (k index = q), so activations chain from layer to layer with no data movement;
//   * the weights are the A operand.  They are staged once per workgroup into LDS in operand order
//     ("images": one ds_read_b128 per lane yields the A operands of four consecutive MFMAs);
//   * f32-input MFMA is an exact k-ordered fmaf chain (cdna_hip_programming.md section 3), so the
//     result is an ordinary fp32 matmul.  The forward throughput kernels run the hidden layers on the bf16 pipe
//     instead, with every fp32 operand split in three bf16 pieces (all 24 bits) and fp32 sums (field_fwd_b16
//     below): no reduced precision anywhere either way.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/cnf_hip.h"

namespace cnf {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kRows = 16;          // batch rows per wave
constexpr int kH = 64;             // hidden width handled by these kernels
constexpr int kNW = 4;             // waves per workgroup
constexpr int kThreads = kNW * 64;
constexpr int kRowsPerWG = kNW * kRows;
constexpr int kMaxWG = 256;        // grid-wide reductions need every workgroup resident: <= 1 per CU

// ---- flat parameter layout (include/cnf_hip.h) -------------------------------------------------
// Offsets are closed forms (no arrays): a runtime-indexed private array would be promoted to LDS by the compiler.
struct ParamLayout {
    int D, H, total;
    __host__ __device__ int din(int l) const { return l == 0 ? D : H; }
    __host__ __device__ int dout(int l) const { return l == 3 ? D : H; }
    __host__ __device__ int w(int l) const { return l == 0 ? 0 : (H * D + 4 * H) + (l - 1) * (H * H + 4 * H); }
    __host__ __device__ int b(int l) const { return w(l) + dout(l) * din(l); }
    __host__ __device__ int hb(int l) const { return b(l) + dout(l); }
    __host__ __device__ int wg(int l) const { return b(l) + 2 * dout(l); }
    __host__ __device__ int bg(int l) const { return b(l) + 3 * dout(l); }
};
__host__ __device__ inline ParamLayout make_layout(int D, int H) {
    ParamLayout L{D, H, 0};
    L.total = L.bg(3) + D;
    return L;
}

// ---- scalar math -------------------------------------------------------------------------------
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ float sigmoid_f(float a) {
    return __builtin_amdgcn_rcpf(1.0f + __expf(-a));
}
// tanh accurate to ~1 ulp: odd minimax polynomial below 0.625 (Cephes tanhf), 1 - 2/(e^{2|x|}+1) above.
__device__ __forceinline__ float tanh_f(float x) {
#ifdef CNF_EXP_FAST_TANH   // timing experiment: 1 - 2 / (e^{2x} + 1) alone (absolute error ~2e-7, 5 instructions instead of ~18)
    return fmaf(-2.0f, __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(x * 2.885390081777927f) + 1.0f), 1.0f);
#endif
    const float ax = fabsf(x);
    const float x2 = x * x;
    float p = fmaf(x2, -5.70498872745e-3f, 2.06390887954e-2f);
    p = fmaf(p, x2, -5.37397155531e-2f);
    p = fmaf(p, x2, 1.33314422036e-1f);
    p = fmaf(p, x2, -3.33332819422e-1f);
    const float small = fmaf(p * x2, x, x);
    const float big = copysignf(1.0f - 2.0f * __builtin_amdgcn_rcpf(__expf(2.0f * ax) + 1.0f), x);
    return ax < 0.625f ? small : big;
}

// ---- the ODEnet's nonlinearity ---------------------------------------------------------------------------------------
// One library per nonlinearity (-DCNF_ACT=..., Makefile): the kernels are written against act_f (the function), act_d
// (its derivative, from the OUTPUT x = act(u) -- the adjoint keeps x, not u) and act_dd (second derivative, from x and
// act_d(x)), so that tanh pays nothing for the others.  cnf_activation() reports which one a library was built for.
#ifndef CNF_ACT
#define CNF_ACT 0
#endif
#if CNF_ACT == 0   // tanh (the reference's runs and checkpoints)
__device__ __forceinline__ float act_f(float u) { return tanh_f(u); }
__device__ __forceinline__ float act_d(float x) { return 1.0f - x * x; }
__device__ __forceinline__ float act_dd(float x, float dx) { return -2.0f * x * dx; }
#elif CNF_ACT == 1   // softplus, nn.Softplus(beta=1, threshold=20): x = log(1 + e^u); act' = sigmoid(u) = 1 - e^{-x}
__device__ __forceinline__ float act_f(float u) { return u > 20.0f ? u : log1pf(expf(u)); }
__device__ __forceinline__ float act_d(float x) { return -expm1f(-x); }
__device__ __forceinline__ float act_dd(float x, float dx) { return dx * (1.0f - dx); }
#elif CNF_ACT == 2   // elu, nn.ELU(alpha=1): x = u (u > 0), e^u - 1 otherwise; act' = 1 or x + 1; act'' = 0 or x + 1
__device__ __forceinline__ float act_f(float u) { return u > 0.0f ? u : expm1f(u); }
__device__ __forceinline__ float act_d(float x) { return x > 0.0f ? 1.0f : x + 1.0f; }
__device__ __forceinline__ float act_dd(float x, float dx) { return x > 0.0f ? 0.0f : dx; }
#elif CNF_ACT == 3   // relu, nn.ReLU: x = max(u, 0); act' = 1 where the unit is on (x > 0: at u = 0 torch's derivative is 0 too); act'' = 0
__device__ __forceinline__ float act_f(float u) { return fmaxf(u, 0.0f); }
__device__ __forceinline__ float act_d(float x) { return x > 0.0f ? 1.0f : 0.0f; }
__device__ __forceinline__ float act_dd(float, float) { return 0.0f; }
#else
#error "CNF_ACT: 0 = tanh, 1 = softplus, 2 = elu, 3 = relu"
#endif

// wave-private LDS hand-off between lanes of ONE wave: program order is enough for the hardware
// (one in-order LDS queue per wave); this only stops the compiler from moving accesses across it.
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Cross-lane sums stay on the vector ALU: gfx950's v_permlane{16,32}_swap exchange 16- / 32-lane halves between two
// registers and DPP moves rotate within a 16-lane row; __shfl_xor compiles to ds_bpermute, an LDS round trip per step.
__device__ __forceinline__ float xq_sum(float v) {        // sum over the 4 lane quarters (same row j), same in all lanes
    const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(a[0]) + __uint_as_float(a[1]);     // quarters {0,1} and {2,3}
    const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}
template <int CTRL>
__device__ __forceinline__ double dpp_mov_f64(double v) {
    const unsigned long long u = (unsigned long long)__double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)u, CTRL, 0xF, 0xF, true);
    const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(u >> 32), CTRL, 0xF, 0xF, true);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
__device__ __forceinline__ double xq_sum_f64(double v) {   // sum over the 4 lane quarters, same in all lanes
#pragma unroll
    for (int step = 0; step < 2; ++step) {
        const unsigned long long u = (unsigned long long)__double_as_longlong(v);
        const unsigned lo = (unsigned)u, hi = (unsigned)(u >> 32);
        const auto a = step == 0 ? __builtin_amdgcn_permlane16_swap(lo, lo, false, false) : __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
        const auto b = step == 0 ? __builtin_amdgcn_permlane16_swap(hi, hi, false, false) : __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
        v = __longlong_as_double((long long)(((unsigned long long)b[0] << 32) | a[0])) +
            __longlong_as_double((long long)(((unsigned long long)b[1] << 32) | a[1]));
    }
    return v;
}
__device__ __forceinline__ double row16_sum(double v) {   // sum over the 16 rows of a quarter; callers use lane 0's (or one lane's) value
    v += dpp_mov_f64<0xB1>(v);    // quad_perm [1,0,3,2]
    v += dpp_mov_f64<0x4E>(v);    // quad_perm [2,3,0,1]
    v += dpp_mov_f64<0x124>(v);   // row_ror:4
    v += dpp_mov_f64<0x128>(v);   // row_ror:8
    return v;
}

// ---- LDS residents -----------------------------------------------------------------------------
template <int D>
struct alignas(16) NetLds {
    float wimg[2][16][64][4];  // layers 2,3: [layer][Tout*4+Tin][lane][r] = W[16Tout+(lane&15)][16Tin+4(lane>>4)+r]
    float w1[D][kH];           // layer 1 weight, [input m][feature i]
    float w4[D][kH];           // layer 4 weight, [output m][feature i]
    float vec[3][4][kH];       // layers 1..3: bias, hyper_bias, hyper_gate.weight, hyper_gate.bias
    float vec4[4][16];         // layer 4: the same four vectors (first D entries used)
};

template <int D>
struct alignas(16) WaveLds {
    float gate[3][2][kH];      // per evaluation: g = sigmoid(wg t + bg), c = g*b + hb*t
    float gate4[2][16];
    float kst[7][D + 1][kRows];  // Runge-Kutta slopes k1..k7 of (z, logp) for the 16 rows
};

struct alignas(16) RedLds {
    double part[kNW][4];
    double tot[4];
    int status;
    int last;   // grid_cleanup: this workgroup took the last ticket
    float keep[2];   // CNF_FLAG_ADJ_CONTROL_ALL: d1 of a_z between the two evaluations of the initial step (written by thread 0)
};

// everything of the net except the two 64x64 matrices: first / last layer weights and the per-layer vectors
template <int D, class N>
__device__ inline void stage_net_vectors(N& n, const float* __restrict__ p, int tid, int nthreads = kThreads) {
    const ParamLayout L = make_layout(D, kH);
    for (int idx = tid; idx < D * kH; idx += nthreads) {
        const int m = idx / kH, i = idx % kH;
        n.w1[m][i] = p[L.w(0) + i * D + m];
        n.w4[m][i] = p[L.w(3) + m * kH + i];
    }
    for (int idx = tid; idx < 3 * kH; idx += nthreads) {
        const int l = idx / kH, i = idx % kH;
        n.vec[l][0][i] = p[L.b(l) + i];
        n.vec[l][1][i] = p[L.hb(l) + i];
        n.vec[l][2][i] = p[L.wg(l) + i];
        n.vec[l][3][i] = p[L.bg(l) + i];
    }
    if (tid < 16) {
        const bool in = tid < D;
        n.vec4[0][tid] = in ? p[L.b(3) + tid] : 0.f;
        n.vec4[1][tid] = in ? p[L.hb(3) + tid] : 0.f;
        n.vec4[2][tid] = in ? p[L.wg(3) + tid] : 0.f;
        n.vec4[3][tid] = in ? p[L.bg(3) + tid] : 0.f;
    }
}

template <int D>
__device__ inline void stage_net(NetLds<D>& n, const float* __restrict__ p, int tid, int nthreads = kThreads) {
    const ParamLayout L = make_layout(D, kH);
    float* img = &n.wimg[0][0][0][0];
    if (nthreads == kThreads && (reinterpret_cast<uintptr_t>(p) & 15) == 0) {
        // an image entry is 4 consecutive floats of a weight row: one 16-byte load each (both matrices start at multiples of
        // 4 floats), and all 8 loads of a thread are in flight before the first store -- the staging of a launch is one
        // round trip to L2 instead of 32 dependent ones (measured: ~8 us of every solve kernel)
        f32x4 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int e = tid + kThreads * k, lane = e & 63, tt = (e >> 6) & 15, l = e >> 10;
            v[k] = *reinterpret_cast<const f32x4*>(&p[L.w(1 + l) + (16 * (tt >> 2) + (lane & 15)) * kH + 16 * (tt & 3) + 4 * (lane >> 4)]);
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) *reinterpret_cast<f32x4*>(&img[4 * (tid + kThreads * k)]) = v[k];
    } else {
        for (int idx = tid; idx < 2 * 16 * 64 * 4; idx += nthreads) {
            const int r = idx & 3, lane = (idx >> 2) & 63, tt = (idx >> 8) & 15, l = idx >> 12;
            const int row = 16 * (tt >> 2) + (lane & 15);
            const int col = 16 * (tt & 3) + 4 * (lane >> 4) + r;
            img[idx] = p[L.w(1 + l) + row * kH + col];
        }
    }
    for (int idx = tid; idx < D * kH; idx += nthreads) {
        const int m = idx / kH, i = idx % kH;
        n.w1[m][i] = p[L.w(0) + i * D + m];
        n.w4[m][i] = p[L.w(3) + m * kH + i];
    }
    for (int idx = tid; idx < 3 * kH; idx += nthreads) {
        const int l = idx / kH, i = idx % kH;
        n.vec[l][0][i] = p[L.b(l) + i];
        n.vec[l][1][i] = p[L.hb(l) + i];
        n.vec[l][2][i] = p[L.wg(l) + i];
        n.vec[l][3][i] = p[L.bg(l) + i];
    }
    if (tid < 16) {
        const bool in = tid < D;
        n.vec4[0][tid] = in ? p[L.b(3) + tid] : 0.f;
        n.vec4[1][tid] = in ? p[L.hb(3) + tid] : 0.f;
        n.vec4[2][tid] = in ? p[L.wg(3) + tid] : 0.f;
        n.vec4[3][tid] = in ? p[L.bg(3) + tid] : 0.f;
    }
}

// gates of all four layers at time t, one feature per lane, into the wave's scratch
template <class N, class WS>
__device__ __forceinline__ void compute_gates(const N& n, WS& ws, int lane, float t) {
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

// ---- the dynamics: f(t, z) and its divergence for the 16 rows of this wave -----------------------
// Restates ODEnet.forward + divergence_bf / divergence_approx with
// forward-mode tangents instead of autograd passes.  z, e: this lane's row (replicated in the 4 quarters).
// Returns f[D] and div = tr(df/dz) (exact) or e^T (df/dz) e (Hutchinson), identical in all 4 quarters.
template <int D, bool HUTCH>
__device__ __forceinline__ void field_fwd(const NetLds<D>& n, WaveLds<D>& ws, int lane, float t,
                                          const float (&z)[D], const float (&e)[D], float (&f)[D], float& div) {
    constexpr int NT = HUTCH ? 1 : D;
    constexpr int NS = NT + 1;
    const int q = lane >> 4;
    compute_gates(n, ws, lane, t);

    float act[NS][16];
    // layer 1: d -> 64 on the vector ALU (K = d is too thin for a matrix op)
#pragma unroll
    for (int T = 0; T < 4; ++T) {
        const int fo = 16 * T + 4 * q;
        const f32x4 g = *reinterpret_cast<const f32x4*>(&ws.gate[0][0][fo]);
        const f32x4 c = *reinterpret_cast<const f32x4*>(&ws.gate[0][1][fo]);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f}, accu = {0.f, 0.f, 0.f, 0.f};
        f32x4 wcol[D];
#pragma unroll
        for (int m = 0; m < D; ++m) {
            wcol[m] = *reinterpret_cast<const f32x4*>(&n.w1[m][fo]);
            acc += wcol[m] * z[m];
            if (HUTCH) accu += wcol[m] * e[m];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float x = act_f(fmaf(g[r], acc[r], c[r]));
            const float dg = act_d(x) * g[r];
            act[0][4 * T + r] = x;
            if (HUTCH) {
                act[1][4 * T + r] = dg * accu[r];
            } else {
#pragma unroll
                for (int m = 0; m < NT; ++m) act[1 + m][4 * T + r] = dg * wcol[m][r];
            }
        }
    }
    // layers 2, 3: 64 -> 64 on the matrix cores, all streams share the weight operand
#pragma unroll
    for (int l = 0; l < 2; ++l) {
        f32x4 acc[NS][4];
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int To = 0; To < 4; ++To) acc[s][To] = f32x4{0.f, 0.f, 0.f, 0.f};
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
                    for (int s = 0; s < NS; ++s) acc[s][To] = mfma16(a[To][r], act[s][4 * Ti + r], acc[s][To]);
        }
#pragma unroll
        for (int To = 0; To < 4; ++To) {
            const int fo = 16 * To + 4 * q;
            const f32x4 g = *reinterpret_cast<const f32x4*>(&ws.gate[l + 1][0][fo]);
            const f32x4 c = *reinterpret_cast<const f32x4*>(&ws.gate[l + 1][1][fo]);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float x = act_f(fmaf(g[r], acc[0][To][r], c[r]));
                const float dg = act_d(x) * g[r];
                act[0][4 * To + r] = x;
#pragma unroll
                for (int s = 1; s < NS; ++s) act[s][4 * To + r] = dg * acc[s][To][r];
            }
        }
    }
    // layer 4: 64 -> d on the vector ALU; each lane holds 16 of the 64 features of its row
    float pf[D], pd[D];
#pragma unroll
    for (int m = 0; m < D; ++m) pf[m] = pd[m] = 0.f;
#pragma unroll
    for (int T = 0; T < 4; ++T) {
        const int fo = 16 * T + 4 * q;
#pragma unroll
        for (int m = 0; m < D; ++m) {
            const f32x4 w = *reinterpret_cast<const f32x4*>(&n.w4[m][fo]);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                pf[m] = fmaf(w[r], act[0][4 * T + r], pf[m]);
                pd[m] = fmaf(w[r], act[HUTCH ? 1 : 1 + m][4 * T + r], pd[m]);
            }
        }
    }
    div = 0.f;
#pragma unroll
    for (int m = 0; m < D; ++m) {
        const float g4 = ws.gate4[0][m];
        f[m] = fmaf(g4, xq_sum(pf[m]), ws.gate4[1][m]);
        const float dm = g4 * xq_sum(pd[m]);
        div += HUTCH ? dm * e[m] : dm;
    }
}

// ---- the same dynamics with the hidden layers on the bf16 matrix pipe, operands split in three -----------------------
// Throughput evaluation (tile-local dopri5, d <= 2).  On gfx950 the fp32 MFMA occupies the vector ALU for its whole
// duration (tools/mfma_valu_overlap.hip); v_mfma_f32_16x16x32_bf16 runs beside it and does 8x the FLOP per clock.  Every
// fp32 operand is written as hi + mid + lo, three TRUNCATED bf16 pieces (24 significant bits, every subtraction exact);
// of the nine piece products the six of weight >= 2^-16 are kept (hh, hm, mh, hl, mm, lh: the dropped ones amount to
// 4e-8 of the product on average, < 2^-21 at worst) and summed in fp32 by the matrix core, smallest first.  Measured on one 64 -> 64 layer against
// float64 (tools/bf16_split_layer.hip): rms error 6.3e-8 of the rms value -- the k-ordered fp32 fmaf chain of
// v_mfma_f32_16x16x4_f32 has 1.5e-7 -- in 993 instead of 1 649 clocks per wave, split included.
// The C/D register layout of a layer's output is a valid B operand of the 16x16x32 instruction as well: lane (q, j)
// supplies 8 consecutive k, and k order is free as long as the weight image uses the same one -- k-step ks takes the
// lane's features 16T + 4q + r of T = 2ks, 2ks+1, i.e. act[8ks .. 8ks+7].
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void split3_bf16(float x, uint32_t (&p)[3]) {   // pieces in the upper 16 bits
    p[0] = __float_as_uint(x) & 0xffff0000u;
    const float r1 = x - __uint_as_float(p[0]);
    p[1] = __float_as_uint(r1) & 0xffff0000u;
    const float r2 = r1 - __uint_as_float(p[1]);
    p[2] = __float_as_uint(r2) & 0xffff0000u;
}
__device__ __forceinline__ bf16x8 mfma_b16_operand(const uint32_t (&pc)[8][3], int p) {
    const u32x4 v = {(pc[0][p] >> 16) | pc[1][p], (pc[2][p] >> 16) | pc[3][p], (pc[4][p] >> 16) | pc[5][p],
                     (pc[6][p] >> 16) | pc[7][p]};
    return __builtin_bit_cast(bf16x8, v);
}
__device__ __forceinline__ f32x4 mfma_b16(bf16x8 a, bf16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

template <int D>
struct alignas(16) NetLdsB16 {
    uint32_t wimg16[2][4][2][3][64][4];   // layers 2,3: [layer][Tout][k-step][piece][lane] = 8 bf16 of W[16Tout+(lane&15)][k order above]
    float w1[D][kH];
    float w4[D][kH];
    float vec[3][4][kH];
    float vec4[4][16];
};

template <int D>
__device__ inline void stage_net_b16(NetLdsB16<D>& n, const float* __restrict__ p, int tid, int nthreads) {
    const ParamLayout L = make_layout(D, kH);
    for (int idx = tid; idx < 2 * 4 * 2 * 64 * 4; idx += nthreads) {
        const int v = idx & 3, lane = (idx >> 2) & 63, ks = (idx >> 8) & 1, To = (idx >> 9) & 3, l = idx >> 11;
        const int row = 16 * To + (lane & 15);
        uint32_t lo[3], hi[3];
        const int e0 = 2 * v, e1 = 2 * v + 1;
        split3_bf16(p[L.w(1 + l) + row * kH + 16 * (2 * ks + (e0 >> 2)) + 4 * (lane >> 4) + (e0 & 3)], lo);
        split3_bf16(p[L.w(1 + l) + row * kH + 16 * (2 * ks + (e1 >> 2)) + 4 * (lane >> 4) + (e1 & 3)], hi);
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) n.wimg16[l][To][ks][pc][lane][v] = (lo[pc] >> 16) | hi[pc];
    }
    for (int idx = tid; idx < D * kH; idx += nthreads) {
        const int m = idx / kH, i = idx % kH;
        n.w1[m][i] = p[L.w(0) + i * D + m];
        n.w4[m][i] = p[L.w(3) + m * kH + i];
    }
    for (int idx = tid; idx < 3 * kH; idx += nthreads) {
        const int l = idx / kH, i = idx % kH;
        n.vec[l][0][i] = p[L.b(l) + i];
        n.vec[l][1][i] = p[L.hb(l) + i];
        n.vec[l][2][i] = p[L.wg(l) + i];
        n.vec[l][3][i] = p[L.bg(l) + i];
    }
    if (tid < 16) {
        const bool in = tid < D;
        n.vec4[0][tid] = in ? p[L.b(3) + tid] : 0.f;
        n.vec4[1][tid] = in ? p[L.hb(3) + tid] : 0.f;
        n.vec4[2][tid] = in ? p[L.wg(3) + tid] : 0.f;
        n.vec4[3][tid] = in ? p[L.bg(3) + tid] : 0.f;
    }
}

template <int D, bool HUTCH>
__device__ __forceinline__ void field_fwd_b16(const NetLdsB16<D>& n, WaveLds<D>& ws, int lane, float t,
                                              const float (&z)[D], const float (&e)[D], float (&f)[D], float& div) {
    constexpr int NT = HUTCH ? 1 : D;
    constexpr int NS = NT + 1;
    const int q = lane >> 4;
    compute_gates(n, ws, lane, t);

    float act[NS][16];
#pragma unroll
    for (int T = 0; T < 4; ++T) {   // layer 1, as field_fwd
        const int fo = 16 * T + 4 * q;
        const f32x4 g = *reinterpret_cast<const f32x4*>(&ws.gate[0][0][fo]);
        const f32x4 c = *reinterpret_cast<const f32x4*>(&ws.gate[0][1][fo]);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f}, accu = {0.f, 0.f, 0.f, 0.f};
        f32x4 wcol[D];
#pragma unroll
        for (int m = 0; m < D; ++m) {
            wcol[m] = *reinterpret_cast<const f32x4*>(&n.w1[m][fo]);
            acc += wcol[m] * z[m];
            if (HUTCH) accu += wcol[m] * e[m];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float x = act_f(fmaf(g[r], acc[r], c[r]));
            const float dg = act_d(x) * g[r];
            act[0][4 * T + r] = x;
            if (HUTCH) {
                act[1][4 * T + r] = dg * accu[r];
            } else {
#pragma unroll
                for (int m = 0; m < NT; ++m) act[1 + m][4 * T + r] = dg * wcol[m][r];
            }
        }
    }
#pragma unroll
    for (int l = 0; l < 2; ++l) {   // layers 2, 3: six bf16 piece products per k-step, all streams share the weight pieces
        f32x4 acc[NS][4];
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int To = 0; To < 4; ++To) acc[s][To] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 b[NS][3];
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                uint32_t pc[8][3];
#pragma unroll
                for (int k = 0; k < 8; ++k) split3_bf16(act[s][8 * ks + k], pc[k]);
#pragma unroll
                for (int p = 0; p < 3; ++p) b[s][p] = mfma_b16_operand(pc, p);
            }
#pragma unroll
            for (int To = 0; To < 4; ++To) {
                bf16x8 a[3];
#pragma unroll
                for (int p = 0; p < 3; ++p)
                    a[p] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(&n.wimg16[l][To][ks][p][lane][0]));
                // smallest products first; consecutive MFMAs go to different accumulators
#pragma unroll
                for (int s = 0; s < NS; ++s) acc[s][To] = mfma_b16(a[0], b[s][2], acc[s][To]);
#pragma unroll
                for (int s = 0; s < NS; ++s) acc[s][To] = mfma_b16(a[2], b[s][0], acc[s][To]);
#pragma unroll
                for (int s = 0; s < NS; ++s) acc[s][To] = mfma_b16(a[1], b[s][1], acc[s][To]);
#pragma unroll
                for (int s = 0; s < NS; ++s) acc[s][To] = mfma_b16(a[0], b[s][1], acc[s][To]);
#pragma unroll
                for (int s = 0; s < NS; ++s) acc[s][To] = mfma_b16(a[1], b[s][0], acc[s][To]);
#pragma unroll
                for (int s = 0; s < NS; ++s) acc[s][To] = mfma_b16(a[0], b[s][0], acc[s][To]);
            }
        }
#pragma unroll
        for (int To = 0; To < 4; ++To) {
            const int fo = 16 * To + 4 * q;
            const f32x4 g = *reinterpret_cast<const f32x4*>(&ws.gate[l + 1][0][fo]);
            const f32x4 c = *reinterpret_cast<const f32x4*>(&ws.gate[l + 1][1][fo]);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float x = act_f(fmaf(g[r], acc[0][To][r], c[r]));
                const float dg = act_d(x) * g[r];
                act[0][4 * To + r] = x;
#pragma unroll
                for (int s = 1; s < NS; ++s) act[s][4 * To + r] = dg * acc[s][To][r];
            }
        }
    }
    float pf[D], pd[D];   // layer 4, as field_fwd
#pragma unroll
    for (int m = 0; m < D; ++m) pf[m] = pd[m] = 0.f;
#pragma unroll
    for (int T = 0; T < 4; ++T) {
        const int fo = 16 * T + 4 * q;
#pragma unroll
        for (int m = 0; m < D; ++m) {
            const f32x4 w = *reinterpret_cast<const f32x4*>(&n.w4[m][fo]);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                pf[m] = fmaf(w[r], act[0][4 * T + r], pf[m]);
                pd[m] = fmaf(w[r], act[HUTCH ? 1 : 1 + m][4 * T + r], pd[m]);
            }
        }
    }
    div = 0.f;
#pragma unroll
    for (int m = 0; m < D; ++m) {
        const float g4 = ws.gate4[0][m];
        f[m] = fmaf(g4, xq_sum(pf[m]), ws.gate4[1][m]);
        const float dm = g4 * xq_sum(pd[m]);
        div += HUTCH ? dm * e[m] : dm;
    }
}

// ---- Dormand-Prince tableau, rounded to fp32 where the state is fp32 ------
__device__ const float kAlpha[6] = {1.f / 5, 3.f / 10, 4.f / 5, 8.f / 9, 1.f, 1.f};
__device__ const float kBeta[6][6] = {
    {(float)(1.0 / 5), 0, 0, 0, 0, 0},
    {(float)(3.0 / 40), (float)(9.0 / 40), 0, 0, 0, 0},
    {(float)(44.0 / 45), (float)(-56.0 / 15), (float)(32.0 / 9), 0, 0, 0},
    {(float)(19372.0 / 6561), (float)(-25360.0 / 2187), (float)(64448.0 / 6561), (float)(-212.0 / 729), 0, 0},
    {(float)(9017.0 / 3168), (float)(-355.0 / 33), (float)(46732.0 / 5247), (float)(49.0 / 176), (float)(-5103.0 / 18656), 0},
    {(float)(35.0 / 384), 0.f, (float)(500.0 / 1113), (float)(125.0 / 192), (float)(-2187.0 / 6784), (float)(11.0 / 84)},
};
__device__ const float kCErr[7] = {
    (float)(35.0 / 384 - 1951.0 / 21600), 0.f, (float)(500.0 / 1113 - 22642.0 / 50085),
    (float)(125.0 / 192 - 451.0 / 720), (float)(-2187.0 / 6784 + 12231.0 / 42400),
    (float)(11.0 / 84 - 649.0 / 6300), (float)(-1.0 / 60)};
__device__ const float kCMid[7] = {
    (float)(6025192743.0 / 30085553152.0 / 2), 0.f, (float)(51252292925.0 / 65400821598.0 / 2),
    (float)(-2691868925.0 / 45128329728.0 / 2), (float)(187940372067.0 / 1594534317056.0 / 2),
    (float)(-1776094331.0 / 19743644256.0 / 2), (float)(11237099.0 / 235043384.0 / 2)};

// Row `stage` (1..6) of the tableau as compile-time literals behind a wave-uniform switch: no memory access on the
// critical path between two evaluations (a run-time index into the __device__ tables above is a dependent load).
__device__ __forceinline__ void stage_coeffs(int stage, float& alpha, float (&b)[6]) {
#pragma unroll
    for (int s = 0; s < 6; ++s) b[s] = 0.f;
    switch (stage) {
        case 1: alpha = (float)(1.0 / 5); b[0] = (float)(1.0 / 5); break;
        case 2: alpha = (float)(3.0 / 10); b[0] = (float)(3.0 / 40); b[1] = (float)(9.0 / 40); break;
        case 3: alpha = (float)(4.0 / 5); b[0] = (float)(44.0 / 45); b[1] = (float)(-56.0 / 15); b[2] = (float)(32.0 / 9); break;
        case 4: alpha = (float)(8.0 / 9); b[0] = (float)(19372.0 / 6561); b[1] = (float)(-25360.0 / 2187);
                b[2] = (float)(64448.0 / 6561); b[3] = (float)(-212.0 / 729); break;
        case 5: alpha = 1.f; b[0] = (float)(9017.0 / 3168); b[1] = (float)(-355.0 / 33); b[2] = (float)(46732.0 / 5247);
                b[3] = (float)(49.0 / 176); b[4] = (float)(-5103.0 / 18656); break;
        default: alpha = 1.f; b[0] = (float)(35.0 / 384); b[1] = 0.f; b[2] = (float)(500.0 / 1113); b[3] = (float)(125.0 / 192);
                b[4] = (float)(-2187.0 / 6784); b[5] = (float)(11.0 / 84); break;
    }
}

// The state at which stage `stage` (1..6) evaluates the dynamics, y0 + dt * sum_{s < stage} a[stage][s] k_s, reading
// exactly the slopes that stage uses (3.5 of 6 on average; stage_coeffs + a six-term loop reads and multiplies all six,
// the unused ones by zero).  k(s, c): slope of stage s, component c.  Returns the stage's time fraction alpha.
// Same sums as the six-term form up to the sign of a zero.
template <int N, int C, class K>
__device__ __forceinline__ void stage_sum(const float (&w)[N], float dtf, const float (&y0)[C], K& k, float (&yi)[C]) {
    float dw[N];
#pragma unroll
    for (int s = 0; s < N; ++s) dw[s] = dtf * w[s];
#pragma unroll
    for (int c = 0; c < C; ++c) {
        float acc = dw[0] * k(0, c);
#pragma unroll
        for (int s = 1; s < N; ++s) acc += dw[s] * k(s, c);
        yi[c] = y0[c] + acc;
    }
}

template <int C, class K>
__device__ __forceinline__ float stage_state(int stage, float dtf, const float (&y0)[C], K k, float (&yi)[C]) {
    switch (stage) {
        case 1: { const float w[1] = {(float)(1.0 / 5)}; stage_sum(w, dtf, y0, k, yi); return (float)(1.0 / 5); }
        case 2: { const float w[2] = {(float)(3.0 / 40), (float)(9.0 / 40)}; stage_sum(w, dtf, y0, k, yi); return (float)(3.0 / 10); }
        case 3: { const float w[3] = {(float)(44.0 / 45), (float)(-56.0 / 15), (float)(32.0 / 9)}; stage_sum(w, dtf, y0, k, yi);
                  return (float)(4.0 / 5); }
        case 4: { const float w[4] = {(float)(19372.0 / 6561), (float)(-25360.0 / 2187), (float)(64448.0 / 6561), (float)(-212.0 / 729)};
                  stage_sum(w, dtf, y0, k, yi); return (float)(8.0 / 9); }
        case 5: { const float w[5] = {(float)(9017.0 / 3168), (float)(-355.0 / 33), (float)(46732.0 / 5247), (float)(49.0 / 176),
                                      (float)(-5103.0 / 18656)};
                  stage_sum(w, dtf, y0, k, yi); return 1.f; }
        default: { const float w[6] = {(float)(35.0 / 384), 0.f, (float)(500.0 / 1113), (float)(125.0 / 192), (float)(-2187.0 / 6784),
                                       (float)(11.0 / 84)};
                   stage_sum(w, dtf, y0, k, yi); return 1.f; }
    }
}

// the time node of a stage, as stage_state returns it
__device__ __forceinline__ float stage_node(int stage) {
    switch (stage) {
        case 1: return (float)(1.0 / 5);
        case 2: return (float)(3.0 / 10);
        case 3: return (float)(4.0 / 5);
        case 4: return (float)(8.0 / 9);
        default: return 1.f;
    }
}

// stage_coeffs multiplies the slopes of stages that have not been evaluated yet by a zero weight, so those slots must
// hold finite numbers from the start: LDS keeps whatever the previous kernel on the CU left there, and 0 * NaN = NaN.
// Call before the kernel's first workgroup barrier; `n_ws` = number of per-wave blocks in `wl`.
template <class WS>
__device__ __forceinline__ void zero_slopes(WS* wl, int n_ws, int tid, int nthreads = kThreads) {
    constexpr int n = (int)(sizeof(wl[0].kst) / sizeof(float));
    for (int i = 0; i < n_ws; ++i) {
        float* k = &wl[i].kst[0][0][0];
        for (int idx = tid; idx < n; idx += nthreads) k[idx] = 0.f;
    }
}

// x^(1/5) in double for x > 0: single-precision seed (2^(log2(x)/5), ~1e-6 relative), then two Newton steps
// y <- (4y + x / y^4) / 5, which square the error twice (2e-12, 8e-24): the result is within an ulp or two of the library's
// pow(x, 0.2) in ~60 instructions instead of several hundred -- every lane of every workgroup executes this once per
// attempt, on the dependent chain between two Runge-Kutta steps (measured: pow() was ~1.7 us of a ~20 us attempt).
__device__ __forceinline__ double fifth_root(double x) {
    double y = (double)__builtin_amdgcn_exp2f(0.2f * __builtin_amdgcn_logf((float)x));
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const double y2 = y * y;
        y = (4.0 * y + x / (y2 * y2)) * 0.2;
    }
    return y;
}

// step-size update of the controller (Appendix B.5); time bookkeeping is float64 like the original
__device__ __forceinline__ double next_step_size(double dt, float worst) {
    if (worst == 0.f) return dt * 10.0;
    const double inv_dfactor = worst < 1.f ? 1.0 : 5.0;
    const double root = (double)sqrtf(worst);
    // outside [1e-30, 1e30] the factor is clamped anyway (0.1 below, inv_dfactor above): no need for a root there
#ifdef CNF_EXP_POW_STEP   // experiment: the library's double-precision pow(), as round 1 had it (DESIGN.md section 4, 3-wave build)
    const double r5 = pow(root, 0.2);
#else
    const double r5 = root < 1e-30 ? 0.0 : (root > 1e30 ? 1e6 : fifth_root(root));
#endif
    const double factor = fmax(0.1, fmin(r5 / 0.9, inv_dfactor));
    return dt / factor;
}

// python's max(a, b) over a 2-tuple: keeps `a` unless b > a (so a NaN in `a` sticks, one in `b` is ignored)
__device__ __forceinline__ float pymax(float a, float b) { return b > a ? b : a; }

// ---- regularised dynamics (include/cnf_hip.h, CnfRegDesc) ---------------------------------------------------------------
// integrand of regulariser `kind` for one row's dz/dt: l1 = mean_i |dx_i|, l2 = ||dx||_2 / sqrt(d)
template <int D>
__device__ __forceinline__ float reg_integrand(int kind, const float (&dx)[D]) {
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int m = 0; m < D; ++m) {
        s1 += fabsf(dx[m]);
        s2 = fmaf(dx[m], dx[m], s2);
    }
    return kind == CNF_REG_L1 ? s1 / (float)D : sqrtf(s2) / sqrtf((float)D);
}
// c[k] * d integrand_k / d dx, summed over the regularisers (autograd's conventions: |x|' = sign(x) with sign(0) = 0,
// ||x||' = x / ||x|| with 0 at the origin)
template <int D>
__device__ __forceinline__ void reg_cotangent(int nreg, const int (&kind)[CNF_MAX_REG], const float (&c)[CNF_MAX_REG],
                                              const float (&dx)[D], float (&out)[D]) {
    float s2 = 0.f;
#pragma unroll
    for (int m = 0; m < D; ++m) {
        s2 = fmaf(dx[m], dx[m], s2);
        out[m] = 0.f;
    }
    const float nrm = sqrtf(s2);
    const float inv = nrm > 0.f ? 1.0f / (nrm * sqrtf((float)D)) : 0.f;
#pragma unroll
    for (int k = 0; k < CNF_MAX_REG; ++k) {
        if (k >= nreg) break;
#pragma unroll
        for (int m = 0; m < D; ++m) {
            const float sg = dx[m] > 0.f ? 1.f : (dx[m] < 0.f ? -1.f : 0.f);
            out[m] += kind[k] == CNF_REG_L1 ? c[k] * (sg / (float)D) : c[k] * (dx[m] * inv);
        }
    }
}
// The first-step heuristic (Appendix B.2) over (z, logp, r_1 .. r_nreg): python's max() walks the tensors in the order of
// the state tuple.  nreg = 0: initial_h0 / initial_dt below with all = false.
__device__ __forceinline__ float initial_h0_reg(float d0z, float d0l, float d1z, float d1l, int nreg,
                                                const float (&d0r)[CNF_MAX_REG], const float (&d1r)[CNF_MAX_REG]) {
    float m0 = pymax(d0z, d0l), m1 = pymax(d1z, d1l), mr = pymax(d0z / d1z, d0l / d1l);
#pragma unroll
    for (int k = 0; k < CNF_MAX_REG; ++k) {
        if (k >= nreg) break;
        m0 = pymax(m0, d0r[k]);
        m1 = pymax(m1, d1r[k]);
        mr = pymax(mr, d0r[k] / d1r[k]);
    }
    return (m0 < 1e-5f || m1 < 1e-5f) ? 1e-6f : 0.01f * mr;
}
__device__ __forceinline__ float initial_dt_reg(float h0, float d1z, float d1l, float d2z, float d2l, int nreg,
                                                const float (&d1r)[CNF_MAX_REG], const float (&d2r)[CNF_MAX_REG]) {
    float m1 = pymax(d1z, d1l), m2 = pymax(d2z, d2l);
#pragma unroll
    for (int k = 0; k < CNF_MAX_REG; ++k) {
        if (k >= nreg) break;
        m1 = pymax(m1, d1r[k]);
        m2 = pymax(m2, d2r[k]);
    }
    float mm = pymax(pymax(m1, d2z), d2l);   // max(d1 + d2): the concatenated tuple, d1's first
#pragma unroll
    for (int k = 0; k < CNF_MAX_REG; ++k) {
        if (k >= nreg) break;
        mm = pymax(mm, d2r[k]);
    }
    const float h1 = (m1 <= 1e-15f && m2 <= 1e-15f) ? fmaxf(1e-6f, h0 * 1e-3f) : powf(0.01f / mm, 0.2f);
    return fminf(100.f * h0, h1);
}
// dopri5's quartic through one component of an accepted step, at x in [0, 1] (B.6): u, v = the step's ends, k = its slopes
__device__ __forceinline__ float dense_quartic(float dtf, float x, float u, float v, const float (&k)[7]) {
    float mid = 0.f;
#pragma unroll
    for (int s = 0; s < 7; ++s) mid += (dtf * kCMid[s]) * k[s];
    mid = u + mid;
    const float f0 = k[0], f1 = k[6];
    const float ca = (-2 * dtf) * f0 + (2 * dtf) * f1 + -8 * u + -8 * v + 16 * mid;
    const float cb = (5 * dtf) * f0 + (-3 * dtf) * f1 + 18 * u + 14 * v + -32 * mid;
    const float cc = (-4 * dtf) * f0 + dtf * f1 + -11 * u + -5 * v + 16 * mid;
    const float cd = dtf * f0;
    const float x2 = x * x, x3 = x2 * x, x4 = x3 * x;
    return ca * x4 + cb * x3 + cc * x2 + cd * x + u;
}

// ---- CNF_FLAG_ADJ_CONTROL_ALL: which tensors of the adjoint state the step controller looks at ------------------------
// The reference hands odeint_adjoint tolerance LISTS of length 2, and torchdiffeq 0.0.1 zips tolerances with the state
// tuple: in the backward solve only (z, logp) drive the error ratios (recalled, unverifiable
// offline) -- the default here.  With the flag, the per-row tensors of the adjoint state, a_z and a_logp, are controlled
// like z and logp (what scalar tolerances would do to them): each contributes its own mean-square error ratio and its own
// terms to the initial-step heuristic, in the state tuple's order (z, logp, a_z, a_logp).  a_logp never changes, so its
// error ratio is 0 and its d1 is 0 -- which makes d0 / d1 = +inf (or 0 / 0 = NaN when the cotangent is zero) in the
// heuristic exactly as it does in the original's arithmetic: h0 = inf, the second evaluation sees a non-finite state, its
// d2 terms come out as 0 or NaN, python's max() skips NaNs that are not its first argument, and the first step is
// (0.01 / max(d1))^(1/5).  The kernels reproduce that outcome without evaluating the field at infinity (the evaluation is
// still run, at the initial state, and counted).  a_t and a_theta (batch-level quadratures, not per-row state) stay
// outside the controller in both modes: see DESIGN.md section 5.
template <int D, int C>
__device__ __forceinline__ void adj_all_f0_sums(const float (&y0)[C], const float (&k)[C], float al, float rtol, float atol,
                                                bool counted, double (&S)[4]) {
#pragma unroll
    for (int m = 0; m < D; ++m) {
        const int c = D + 1 + m;
        const float s = atol + fabsf(y0[c]) * rtol;
        const float u = y0[c] / s, v = k[c] / s;
        if (counted) {
            S[0] += (double)u * u;
            S[1] += (double)v * v;
        }
    }
    const float ua = al / (atol + fabsf(al) * rtol);
    if (counted) S[2] += (double)ua * ua;
}
// h0 of the heuristic (Appendix B.2) from the d0 / d1 of (z, logp[, a_z, a_logp]); d1 of a_logp is identically zero
__device__ __forceinline__ float initial_h0(float d0z, float d0l, float d1z, float d1l, bool all, float d0a, float d1a, float d0al) {
    float m0 = pymax(d0z, d0l), m1 = pymax(d1z, d1l), mr = pymax(d0z / d1z, d0l / d1l);
    if (all) {
        const float zero = 0.f;
        m0 = pymax(pymax(m0, d0a), d0al);
        m1 = pymax(pymax(m1, d1a), zero);
        mr = pymax(pymax(mr, d0a / d1a), d0al / zero);
    }
    return (m0 < 1e-5f || m1 < 1e-5f) ? 1e-6f : 0.01f * mr;
}
// the first step size from h0 and the d1 / d2 terms, in the tuple order python's max() walks them
__device__ __forceinline__ float initial_dt(float h0, float d1z, float d1l, float d2z, float d2l, bool all, float d1a, float d2a) {
    const bool inf0 = all && !(h0 < INFINITY);   // see above: the d2 terms are 0 or NaN then, never the maximum
    if (inf0) d2z = d2l = d2a = 0.f;
    float m1 = pymax(d1z, d1l), m2 = pymax(d2z, d2l), mm;
    if (all) {
        m1 = pymax(pymax(m1, d1a), 0.f);
        m2 = pymax(pymax(m2, d2a), 0.f);
        mm = pymax(pymax(pymax(m1, d2z), d2l), d2a);
    } else {
        mm = pymax(pymax(m1, d2z), d2l);
    }
    float h1;
    if (!inf0 && m1 <= 1e-15f && m2 <= 1e-15f) h1 = fmaxf(1e-6f, h0 * 1e-3f);
    else h1 = powf(0.01f / mm, 0.2f);
    return fminf(100.f * h0, h1);
}

// ---- grid-wide reduction of up to 4 doubles -----------------------------------------------------
// Every workgroup publishes its partial as four 8-byte {epoch, value} granules (one sc1 store each: the
// data is its own flag, cdna_hip_programming.md Guideline 16 form R2) and wave 0 of every workgroup
// sweeps all granules until every tag equals the epoch, then adds them in a fixed order: all
// workgroups obtain bit-identical totals, so they take identical accept/reject decisions.  Two
// granule banks alternate by epoch parity (a workgroup can be at most one reduction ahead).
struct GridSync {
    unsigned long long* gran;  // [2][nwg*4]
    int nwg;
    unsigned epoch;
#ifdef CNF_PROF_SYNC   // timing experiment (tools/build_variants.sh sync "-DCNF_PROF_SYNC"): wave 0's view of the reductions
    unsigned long long t_total = 0, t_first = 0, t_pre = 0, t_post = 0;
    unsigned n_sweeps = 0, n_red = 0, n_self_first = 0;
#endif
};

constexpr unsigned kSpinLimit = 1u << 24;
// Before its first sweep the polling wave sleeps CNF_PRE_POLL_SLEEP x 64 clocks (~0.55 us).  An agent-scope store needs about that
// long to become visible to the other compute units, so a sweep issued right behind the workgroup's own publish cannot succeed
// when the workgroups arrive together (they do: they run the same evaluations) -- and 128..256 pollers reading the same 4..8 KB
// of granules are a hot spot at the memory side, so the futile sweep also delays the stores it is waiting for.  Measured on
// the forward solve (B = 2 048, 128 workgroups; tools/time_kernels.py, time_prologue.py): stiff field 2.65 -> 2.55 us per
// evaluation and NFE-14 launch 52.1 -> 49.4 us at 20; 8 / 12 / 16 / 24 / 32: 2.62 / 2.62 / 2.59 / 2.56 / 2.61.  Two sweeps in
// flight instead (more polling, earlier detection) measured SLOWER (2.70), a longer sleep between sweeps neutral.
// Applied where it was measured: the eight- / twelve-wave forward solve (tw_allreduce4, cnf_split_fwd_tw.h).  wg_allreduce4
// below keeps its immediate first sweep (CNF_PRE_POLL_SLEEP_WG, default 0): with it on, the H = 256 figures of the bench came
// out 2-3 % lower (long evaluations, workgroups that do not arrive together) and the d = 16 ones unchanged.
#ifndef CNF_PRE_POLL_SLEEP
#define CNF_PRE_POLL_SLEEP 20
#endif
#ifndef CNF_PRE_POLL_SLEEP_WG
#define CNF_PRE_POLL_SLEEP_WG 0
#endif

// returns false on timeout (status set in red.status).  All threads of the workgroup must call it.
// ALLQ: all four lane quarters of a wave carry contributions (cnf_split16.hip), not just quarter 0.
template <bool ALLQ = false, bool PART = true>   // PART: see wg_reduce_publish
__device__ inline bool wg_allreduce4(GridSync& gs, RedLds& red, double (&v)[4], int w, int lane) {
#ifdef CNF_PROF_SYNC
    const unsigned long long t_in = clock64();
    unsigned long long t_done = t_in;
#endif
    if constexpr (PART) {
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = ALLQ ? xq_sum_f64(row16_sum(v[i])) : row16_sum(v[i]);
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) red.part[w][i] = v[i];
    }
    }
    __syncthreads();
    if (PART && w == 0) {
        double s = 0.0;
        if (lane < 4) s = red.part[0][lane] + red.part[1][lane] + red.part[2][lane] + red.part[3][lane];
#ifdef CNF_PROF_SYNC
        const unsigned long long tp0 = clock64();
        gs.t_pre += tp0 - t_in;
        bool first_sweep = true;
#endif
#ifdef CNF_EXP_NO_GRIDSYNC
        if (false) {
#else
        if (gs.nwg > 1) {
#endif
            gs.epoch += 1;
            unsigned long long* bank = gs.gran + (size_t)(gs.epoch & 1u) * (size_t)(gs.nwg * 4);
            if (lane < 4) {
                const unsigned long long pk = ((unsigned long long)gs.epoch << 32) | (unsigned long long)__float_as_uint((float)s);
                __hip_atomic_store(bank + blockIdx.x * 4 + lane, pk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            const int ng = gs.nwg * 4;
            double acc = 0.0;
            bool done = false;
            if (CNF_PRE_POLL_SLEEP_WG > 0) __builtin_amdgcn_s_sleep(CNF_PRE_POLL_SLEEP_WG);   // note above
            // every lane polls its granules (g = lane, lane + 64, ...) with ALL loads of a sweep in flight at once: a
            // sweep then costs one L2 round trip instead of one per granule
            constexpr int kPerLane = kMaxWG * 4 / 64;
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
                    acc += (double)__uint_as_float((unsigned)x[k]);   // padding entries carry +0.0f
                }
#ifdef CNF_PROF_SYNC
                gs.n_sweeps += 1;
                if (first_sweep) {
                    first_sweep = false;
                    gs.t_first += clock64() - tp0;
                    // was this workgroup's own first granule (slot blockIdx.x * 4) already visible to it?
                    const int g0 = blockIdx.x * 4;
                    unsigned long long mine = 0;
#pragma unroll
                    for (int k = 0; k < kPerLane; ++k) mine = (g0 / 64 == k) ? x[k] : mine;
                    const unsigned tag = (unsigned)(__shfl(mine, g0 % 64) >> 32);
                    gs.n_self_first += tag == gs.epoch;
                }
#endif
                if (__all(ok)) { done = true; break; }
                __builtin_amdgcn_s_sleep(1);
            }
#ifdef CNF_PROF_SYNC
            t_done = clock64();
            gs.t_total += t_done - tp0;
            gs.n_red += 1;
#endif
            if (!done && lane == 0) red.status = CNF_ERR_SYNC_TIMEOUT;
            // lanes with equal (lane & 3) hold partial sums of the same quantity; lanes 0..3 need the totals
            acc += dpp_mov_f64<0x124>(acc);   // row_ror:4
            acc += dpp_mov_f64<0x128>(acc);   // row_ror:8
            s = xq_sum_f64(acc);
        }
        if (lane < 4) red.tot[lane] = s;
    } else {
        gs.epoch += (gs.nwg > 1);
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = red.tot[i];
#ifdef CNF_PROF_SYNC
    if (w == 0) gs.t_post += clock64() - t_done;
#endif
    return red.status == 0;
}

// The same reduction in two halves, for callers that have work to do while the granules travel: wg_reduce_publish sums the
// workgroup's partials and stores its granules; wg_reduce_collect polls and totals.  Arithmetic and summation order are those
// of wg_allreduce4 (bit-identical totals).  red.part must stay untouched in between, and no other reduction may run.
// All threads of the workgroup call both.
// PART = false: a wave beyond the kNW that carry rows (the theta waves of an eight-wave workgroup, cnf_split.hip): it
// contributes nothing and polls nothing, but takes part in the barriers, keeps the epoch and reads the totals.
template <bool ALLQ = false, bool PART = true>
__device__ inline void wg_reduce_publish(GridSync& gs, RedLds& red, double (&v)[4], int w, int lane) {
    if constexpr (PART) {
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = ALLQ ? xq_sum_f64(row16_sum(v[i])) : row16_sum(v[i]);
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) red.part[w][i] = v[i];
    }
    }
    __syncthreads();
    if (gs.nwg > 1) {
        gs.epoch += 1;
        if (PART && w == 0 && lane < 4) {
            const double s = red.part[0][lane] + red.part[1][lane] + red.part[2][lane] + red.part[3][lane];
            unsigned long long* bank = gs.gran + (size_t)(gs.epoch & 1u) * (size_t)(gs.nwg * 4);
            const unsigned long long pk = ((unsigned long long)gs.epoch << 32) | (unsigned long long)__float_as_uint((float)s);
            __hip_atomic_store(bank + blockIdx.x * 4 + lane, pk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}
template <bool PART = true>
__device__ inline bool wg_reduce_collect(GridSync& gs, RedLds& red, double (&v)[4], int w, int lane) {
    if (PART && w == 0) {
        double s = 0.0;
        if (lane < 4) s = red.part[0][lane] + red.part[1][lane] + red.part[2][lane] + red.part[3][lane];
        if (gs.nwg > 1) {
            const unsigned long long* bank = gs.gran + (size_t)(gs.epoch & 1u) * (size_t)(gs.nwg * 4);
            const int ng = gs.nwg * 4;
            double acc = 0.0;
            bool done = false;
            constexpr int kPerLane = kMaxWG * 4 / 64;
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

// End of a batch-global kernel: leave the granules as the next launch needs them (all zero) without a memset node.
// Every workgroup takes a ticket after its last reduction; the one that takes the last ticket knows all the others
// are done reading, clears both banks and resets the ticket (the word right after the banks).  All threads call it.
constexpr int kGranWords = 2 * kMaxWG * 4;   // 8-byte granules in the two banks; the ticket is granule kGranWords
// The ticket is a device-scope atomic whose answer takes ~2 us to come back: grid_ticket() takes it as soon as the
// workgroup is past its last reduction (before it computes and stores its outputs), grid_cleanup() looks at the answer
// when everything else is done.  Both are called by all threads; kernels that call only grid_cleanup() get both.
__device__ inline unsigned grid_ticket(GridSync& gs, int tid) {
    if (gs.nwg <= 1 || tid != 0) return 0u;
    unsigned* ticket = reinterpret_cast<unsigned*>(gs.gran + kGranWords);
    // relaxed: this follows the workgroup's last poll in program order (wave 0 polled, and its loads have returned), and the
    // granules are only ever written with agent-scope atomic stores that every poller has already seen; an agent-scope
    // release here would write the whole L2 back at the end of every solve
    return __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ inline void grid_cleanup(GridSync& gs, RedLds& red, int tid, unsigned taken) {
    if (gs.nwg <= 1) return;
    unsigned* ticket = reinterpret_cast<unsigned*>(gs.gran + kGranWords);
    if (tid == 0) red.last = taken == (unsigned)gs.nwg - 1u;
    __syncthreads();
    if (red.last) {
        for (int i = tid; i < 2 * gs.nwg * 4; i += kThreads) gs.gran[i] = 0ull;
        if (tid == 0) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
__device__ inline void grid_cleanup(GridSync& gs, RedLds& red, int tid) {
    __syncthreads();   // wave 0 polls for the workgroup: nobody takes the ticket before it is back from the last reduction
    grid_cleanup(gs, red, tid, grid_ticket(gs, tid));
}

// Solver statistics of a batch-global solve: workgroup 0 stores the counters (no atomics, so the block need not be
// zeroed before the launch); the status is the one sticky word: any workgroup that failed raises it with atomicMax.
__device__ inline void write_stats_global(CnfStats* st, int status, int nfe, int nacc, int nrej, int nrun, float last_dt,
                                          long rows, int tid) {
    if (tid != 0) return;
    if (status) atomicMax(&st->status, status);
    if (blockIdx.x == 0) {
        st->nfe = nfe;
        st->n_accepted = nacc;
        st->n_rejected = nrej;
        st->last_dt = last_dt;
        st->evals_run = nrun;
        st->row_evals = (int64_t)rows * nrun;
    }
}

// tile-local variant: the 16 rows of this wave are the whole control group
__device__ __forceinline__ void wave_allreduce4(double (&v)[4]) {
    // only quarter 0 carries counted rows: reduce its 16 lanes, then hand lane 0's total to every lane
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = __shfl(row16_sum(v[i]), 0);
}

}  // namespace cnf
