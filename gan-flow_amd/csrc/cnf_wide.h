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

typedef uint32_t WideStreamBlk[3][kWT / 2][64][4];   // one stream of a piece buffer


// A wave's weight pieces of k-step ks + 1 (4 tiles x 3 pieces, 12 KB per wave from L2) and the buffer's operands of k-step
// ks + 1 are REQUESTED BEFORE the 72 MFMAs of k-step ks are issued, into the other half of a two-deep register ring, and the
// order is pinned with sched_group_barrier: left to itself the scheduler, at this kernel's register pressure, sinks the
// loads to just before their first use and every k-step then waits out an L2 round trip (measured on the first version
// of this file: 2.9 k clocks per k-step against 1.15 k of matrix work).
typedef u32x4 WideRing[2][kWTW][3];
// first k-step (issued before the barrier that releases the buffer they multiply)
__device__ __forceinline__ void wide_wp_first(const u32x4* __restrict__ wp, WideRing& wb) {
#pragma unroll
    for (int T = 0; T < kWTW; ++T)
#pragma unroll
        for (int p = 0; p < 3; ++p) wb[0][T][p] = wp[((size_t)(T * (kWT / 2) + 0) * 3 + p) * 64];
}

// acc[T][s] = sum over the 256 inputs of (this wave's 4 output tiles of the matrix whose pieces `wp` streams) x (stream s of
// the buffer): six piece products per k-step, smallest first, as field_fwd_wide's bf16 path
template <int NS>
__device__ __forceinline__ void wide_matmul_b16(const WideStreamBlk* xb, const u32x4* __restrict__ wp, WideRing& wb, int lane,
                                                f32x4 (&acc)[kWTW][NS]) {
#pragma unroll
    for (int T = 0; T < kWTW; ++T)
#pragma unroll
        for (int s = 0; s < NS; ++s) acc[T][s] = f32x4{0.f, 0.f, 0.f, 0.f};
#ifdef CNF_EXP_WIDE_KSTEPS   // timing experiment: fewer k-steps (wrong results)
    constexpr int kSteps = CNF_EXP_WIDE_KSTEPS;
#else
    constexpr int kSteps = kWT / 2;
#endif
    u32x4 bq[2][NS][3];
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int p = 0; p < 3; ++p) bq[0][s][p] = *reinterpret_cast<const u32x4*>(&xb[s][p][0][lane][0]);
#pragma unroll 2
    for (int ks = 0; ks < kSteps; ++ks) {
        const int kn = ks + 1 < kWT / 2 ? ks + 1 : kWT / 2 - 1;   // past the end: a redundant load instead of a branch
        const int cur = ks & 1, nxt = cur ^ 1;
#pragma unroll
        for (int T = 0; T < kWTW; ++T)
#pragma unroll
            for (int p = 0; p < 3; ++p) wb[nxt][T][p] = wp[((size_t)(T * (kWT / 2) + kn) * 3 + p) * 64];
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int p = 0; p < 3; ++p) bq[nxt][s][p] = *reinterpret_cast<const u32x4*>(&xb[s][p][kn][lane][0]);
#pragma unroll
        for (int T = 0; T < kWTW; ++T) {
            bf16x8 aw[3], b[NS][3];
#pragma unroll
            for (int p = 0; p < 3; ++p) aw[p] = __builtin_bit_cast(bf16x8, wb[cur][T][p]);
#pragma unroll
            for (int s = 0; s < NS; ++s)
#pragma unroll
                for (int p = 0; p < 3; ++p) b[s][p] = __builtin_bit_cast(bf16x8, bq[cur][s][p]);
#pragma unroll
            for (int s = 0; s < NS; ++s) acc[T][s] = mfma_b16(aw[0], b[s][2], acc[T][s]);
#pragma unroll
            for (int s = 0; s < NS; ++s) acc[T][s] = mfma_b16(aw[2], b[s][0], acc[T][s]);
#pragma unroll
            for (int s = 0; s < NS; ++s) acc[T][s] = mfma_b16(aw[1], b[s][1], acc[T][s]);
#pragma unroll
            for (int s = 0; s < NS; ++s) acc[T][s] = mfma_b16(aw[0], b[s][1], acc[T][s]);
#pragma unroll
            for (int s = 0; s < NS; ++s) acc[T][s] = mfma_b16(aw[1], b[s][0], acc[T][s]);
#pragma unroll
            for (int s = 0; s < NS; ++s) acc[T][s] = mfma_b16(aw[0], b[s][0], acc[T][s]);
        }
        // the k-step as the scheduler must lay it out: the 12 + 3 NS requests first, then the 24 NS MFMAs
        __builtin_amdgcn_sched_group_barrier(0x020, kWTW * 3, 0);      // VMEM reads
        __builtin_amdgcn_sched_group_barrier(0x100, NS * 3, 0);        // LDS reads
        __builtin_amdgcn_sched_group_barrier(0x008, kWTW * 6 * NS, 0); // MFMA
    }
}

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
