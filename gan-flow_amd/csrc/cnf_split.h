// cnf_split.h -- helpers shared by the feature-split adjoint kernels (cnf_split.hip, cnf_split16.hip).
#pragma once
#include "cnf_launch.h"

namespace cnf {

__device__ __forceinline__ float row16_allsum_s(float v) {
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, true));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, true));
    return v;
}

struct BigAccS {
    f32x4 gw[2][4];   // [layer][Tin]: lane (q, j') reg r = dW[16*wave+4q+r][16Tin+j']
    __device__ __forceinline__ void zero() {
#pragma unroll
        for (int l = 0; l < 2; ++l)
#pragma unroll
            for (int b = 0; b < 4; ++b) gw[l][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
};

// ---- cross-row reductions for the small-parameter gradients -----------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}

// Sums of 8 values (a[0..3], b[0..3]) over the 16 rows of a tile (the 16 lanes of a DPP row), "transposed": every
// exchange step halves the number of values a lane carries instead of reducing each value separately.  On return lane
// j holds the two totals of index r* = j & 3:  ta = sum_rows a[r*],  tb = sum_rows b[r*]  (11 DPP adds + 12 selects
// against 64 DPP adds for eight separate butterflies).  b0 = j & 1, b1 = j & 2.
__device__ __forceinline__ void reduce8(const float (&a)[4], const float (&b)[4], bool b0, bool b1, float& ta, float& tb) {
    const float a0 = (b0 ? a[1] : a[0]) + dpp_mov<0xB1>(b0 ? a[0] : a[1]);   // quad_perm [1,0,3,2]: rows j ^ 1
    const float a1 = (b0 ? a[3] : a[2]) + dpp_mov<0xB1>(b0 ? a[2] : a[3]);
    const float c0 = (b0 ? b[1] : b[0]) + dpp_mov<0xB1>(b0 ? b[0] : b[1]);
    const float c1 = (b0 ? b[3] : b[2]) + dpp_mov<0xB1>(b0 ? b[2] : b[3]);
    float aa = (b1 ? a1 : a0) + dpp_mov<0x4E>(b1 ? a0 : a1);                 // quad_perm [2,3,0,1]: rows j ^ 2
    float cc = (b1 ? c1 : c0) + dpp_mov<0x4E>(b1 ? c0 : c1);
    aa += dpp_mov<0x124>(aa);                                                // row_ror:4, row_ror:8: the lanes with equal j & 3
    cc += dpp_mov<0x124>(cc);
    aa += dpp_mov<0x128>(aa);
    cc += dpp_mov<0x128>(cc);
    ta = aa;
    tb = cc;
}

// fire-and-forget LDS accumulate (ds_add_f32): every address has ONE owner lane per evaluation and a wave's LDS
// operations complete in order, so the sums are deterministic
__device__ __forceinline__ void lds_add(float* p, float v) {
    (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

template <int D>
__device__ inline void stage_wimgT(float* img, const float* __restrict__ p, int tid) {
    const ParamLayout L = make_layout(D, kH);
    for (int idx = tid; idx < 2 * 16 * 64 * 4; idx += kThreads) {
        const int r = idx & 3, lane = (idx >> 2) & 63, tt = (idx >> 8) & 15, l = idx >> 12;
        const int row = 16 * (tt & 3) + 4 * (lane >> 4) + r;
        const int col = 16 * (tt >> 2) + (lane & 15);
        img[idx] = p[L.w(1 + l) + row * kH + col];
    }
}

// this wave's weight operands of both hidden layers, held in registers for the whole solve: [layer][piece][k-step]
struct SplitWeightsB16 {
    bf16x8 a[2][3][2];
    template <int D>
    __device__ __forceinline__ void load(const NetLds<D>& n, int wave, int lane) {
#pragma unroll
        for (int l = 0; l < 2; ++l)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                uint32_t pc[8][3];
#pragma unroll
                for (int k = 0; k < 8; ++k) split3_bf16(n.wimg[l][wave * 4 + 2 * ks + (k >> 2)][lane][k & 3], pc[k]);
#pragma unroll
                for (int p = 0; p < 3; ++p) a[l][p][ks] = mfma_b16_operand(pc, p);
            }
    }
};

// A wave publishes its own 4 values per stream as bf16 pieces: per (stream, piece) 4 bf16 = 8 bytes, placed so that the 16
// bytes a lane reads back for k-step ks are the pieces of tiles 2ks and 2ks+1 -- the B operand of
// v_mfma_f32_16x16x32_bf16 as it is.  xb16: one buffer, [stream][piece][k-step][lane][tile & 1][2].
template <int NS>
__device__ __forceinline__ void publish_b16(uint32_t (*xb16)[3][2][64][4], const f32x4 (&act)[NS], int wave, int lane) {
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        uint32_t pc[4][3];
#pragma unroll
        for (int r = 0; r < 4; ++r) split3_bf16(act[s][r], pc[r]);
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            uint2 v;
            v.x = (pc[0][p] >> 16) | pc[1][p];
            v.y = (pc[2][p] >> 16) | pc[3][p];
            *reinterpret_cast<uint2*>(&xb16[s][p][wave >> 1][lane][2 * (wave & 1)]) = v;
        }
    }
}
// this wave's 16 output features of hidden layer l for NS streams: six piece products per k-step, smallest first,
// consecutive MFMAs to different accumulators
template <int NS>
__device__ __forceinline__ void hidden_b16(const uint32_t (*xb16)[3][2][64][4], const SplitWeightsB16& wb, int l, int lane,
                                           f32x4 (&acc)[NS]) {
#pragma unroll
    for (int s = 0; s < NS; ++s) acc[s] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        bf16x8 b[NS][3];
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int p = 0; p < 3; ++p)
                b[s][p] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(&xb16[s][p][ks][lane][0]));
#pragma unroll
        for (int s = 0; s < NS; ++s) acc[s] = mfma_b16(wb.a[l][0][ks], b[s][2], acc[s]);
#pragma unroll
        for (int s = 0; s < NS; ++s) acc[s] = mfma_b16(wb.a[l][2][ks], b[s][0], acc[s]);
#pragma unroll
        for (int s = 0; s < NS; ++s) acc[s] = mfma_b16(wb.a[l][1][ks], b[s][1], acc[s]);
#pragma unroll
        for (int s = 0; s < NS; ++s) acc[s] = mfma_b16(wb.a[l][0][ks], b[s][1], acc[s]);
#pragma unroll
        for (int s = 0; s < NS; ++s) acc[s] = mfma_b16(wb.a[l][1][ks], b[s][0], acc[s]);
#pragma unroll
        for (int s = 0; s < NS; ++s) acc[s] = mfma_b16(wb.a[l][0][ks], b[s][0], acc[s]);
    }
}

// sum of the workgroups' gradient slabs in a fixed order (kernel in cnf_split.hip)
hipError_t launch_reduce_slab(const float* slab, int nwg, int P, float* out, hipStream_t s, int split_at = 0);
// slab sum (as launch_reduce_slab) + Adam update of the P parameters in one kernel; grad_out may be null; guard (nullable):
// the stats block of the adjoint solve that wrote the slabs -- a non-zero status there makes the kernel a no-op
hipError_t launch_adam_slabs(const float* slab, int ntiles, int P, int split_at, float* grad_out, float* p, float* m, float* v,
                             int* state, const CnfAdamOpts& o, const CnfStats* guard, hipStream_t s);

}  // namespace cnf
