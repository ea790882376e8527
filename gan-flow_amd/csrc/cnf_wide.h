// cnf_wide.h -- pieces shared by the hidden-width-256 kernels (cnf_wide.hip: forward path, cnf_wide_bwd.hip: adjoint).
#pragma once
#include "cnf_launch.h"

namespace cnf {

constexpr int kWH = 256;                 // hidden width
constexpr int kWT = kWH / 16;            // feature tiles per layer (16)
constexpr int kWTW = kWT / kNW;          // output tiles per wave (4)
constexpr int kWF = kWH / kNW;           // features per wave (64)
constexpr size_t kWidePieceBytes = (size_t)2 * kWT * (kWT / 2) * 3 * 64 * 16;   // wide_split_weights_kernel's output (786 432)

template <int D>
struct alignas(16) WideNetLds {
    float w1[D][kWH];
    float w4[D][kWH];
    float vec[3][4][kWH];
    float vec4[4][16];
};

template <int C>
struct alignas(16) WideWaveLds {
    float gate[3][2][kWF];   // gates of this wave's 64 features
    float gate4[2][16];
    float kst[7][C][kRows];
};

// NB = 2: the two hidden layers' inputs in separate buffers (no barrier between a layer's reads and the next layer's
// writes; latency-bound batch-global solves, one workgroup per CU anyway).  NB = 1: one buffer and one more barrier
// per evaluation, which brings a workgroup under 80 KB so that TWO are resident per CU (tile-local throughput solves).
template <int D, int NS, int NB = 2>
struct alignas(16) WideXchg {
    static constexpr int kBuffers = NB;
    static constexpr bool kB16 = false;
    float xb[NB][NS][kWT][64][4];
    float part[kNW][2 * D][kRows];
};
// Forward dopri5 solves: activations travel as bf16 pieces for v_mfma_f32_16x16x32_bf16 (cnf_device.h, field_fwd_b16;
// layout as SplitXchgB16, 8 k-steps of two input tiles each).  1.5x the bytes of the fp32 buffer: one buffer only.
template <int D, int NS, int NB = 1>
struct alignas(16) WideXchgB16 {
    static constexpr int kBuffers = NB;
    static constexpr bool kB16 = true;
    uint32_t xb16[NB][NS][3][kWT / 2][64][4];   // [buffer][stream][piece][k-step][lane][tile & 1][2]
    float part[kNW][2 * D][kRows];
};

template <int D>
__device__ inline void wide_stage_net(WideNetLds<D>& n, const float* __restrict__ p, int tid) {
    const ParamLayout L = make_layout(D, kWH);
    for (int idx = tid; idx < D * kWH; idx += kThreads) {
        const int m = idx / kWH, i = idx % kWH;
        n.w1[m][i] = p[L.w(0) + i * D + m];
        n.w4[m][i] = p[L.w(3) + m * kWH + i];
    }
    for (int idx = tid; idx < 3 * kWH; idx += kThreads) {
        const int l = idx / kWH, i = idx % kWH;
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

__device__ __forceinline__ float wsum4(float a, float b, float c, float d) { return ((a + b) + c) + d; }

}  // namespace cnf
