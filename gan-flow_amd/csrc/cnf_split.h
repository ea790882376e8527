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

// NL = 2: both hidden matrices; NL = 1: the one matrix a workgroup of a two-workgroups-per-tile launch keeps (slot 0)
template <int NL>
struct BigAccT {
    static constexpr int kNL = NL;
    f32x4 gw[NL][4];   // [layer][Tin]: lane (q, j') reg r = dW[16*wave+4q+r][16Tin+j']
    __device__ __forceinline__ void zero() {
#pragma unroll
        for (int l = 0; l < NL; ++l)
#pragma unroll
            for (int b = 0; b < 4; ++b) gw[l][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
};
typedef BigAccT<2> BigAccS;

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

// The same operands read straight from the parameter vector (no fp32 LDS image): lane (q, j') of wave w holds, per layer
// and k-step, W[16w + j'][16(2ks + h) + 4q + r], h = 0, 1, r = 0..3 -- two 16-byte loads.
template <int D>
__device__ __forceinline__ void load_weights_b16_global(SplitWeightsB16& wb, const float* __restrict__ p, int wave, int lane) {
    const ParamLayout L = make_layout(D, kH);
    const int q = lane >> 4, j = lane & 15;
#pragma unroll
    for (int l = 0; l < 2; ++l)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            uint32_t pc[8][3];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const float* src = p + L.w(1 + l) + (16 * wave + j) * kH + 16 * (2 * ks + h) + 4 * q;
#pragma unroll
                for (int r = 0; r < 4; ++r) split3_bf16(src[r], pc[4 * h + r]);
            }
#pragma unroll
            for (int pi = 0; pi < 3; ++pi) wb.a[l][pi][ks] = mfma_b16_operand(pc, pi);
        }
}

// The same operands for a workgroup in which G waves (one producer, G - 1 helpers: cnf_split_fwd_tw.h) work on one feature
// tile and need IDENTICAL operands: the conversion (two loads, eight three-piece splits and the packing per (layer, k-step)
// unit) is divided among them -- wave `role` of the tile takes units role, role + G, ... -- and handed over through LDS
// across the barrier that ends the kernel's staging; every wave then reads all twelve operands back (twelve 16-byte LDS
// reads per lane).  Three waves converting everything each spent ~1 us of a launch on the SIMD's shared vector issue.
// sh: [tile][unit = 2 l + ks][piece][lane].
typedef u32x4 SplitWeightsShare[kNW][4][3][64];
template <int D, int G>
__device__ __forceinline__ void convert_weights_b16_shared(SplitWeightsShare& sh, const float* __restrict__ p, int tile, int role, int lane) {
    const ParamLayout L = make_layout(D, kH);
    const int q = lane >> 4, j = lane & 15;
#pragma unroll 1
    for (int u = role; u < 4; u += G) {   // role is uniform across the wave
        const int l = u >> 1, ks = u & 1;
        uint32_t pc[8][3];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const float* src = p + L.w(1 + l) + (16 * tile + j) * kH + 16 * (2 * ks + h) + 4 * q;
#pragma unroll
            for (int r = 0; r < 4; ++r) split3_bf16(src[r], pc[4 * h + r]);
        }
#pragma unroll
        for (int pi = 0; pi < 3; ++pi) sh[tile][u][pi][lane] = __builtin_bit_cast(u32x4, mfma_b16_operand(pc, pi));
    }
}
// ... behind the barrier
__device__ __forceinline__ void collect_weights_b16_shared(SplitWeightsB16& wb, const SplitWeightsShare& sh, int tile, int lane) {
#pragma unroll
    for (int l = 0; l < 2; ++l)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int pi = 0; pi < 3; ++pi) wb.a[l][pi][ks] = __builtin_bit_cast(bf16x8, sh[tile][2 * l + ks][pi][lane]);
}

// Transposed weight pieces in LDS, one slice per wave: [layer][wave][k-step][piece][lane] = 8 bf16 of
// W[16(2ks + h) + 4q + r][16 wave + j'] (the A operand of W^T . cotangent for the wave's own INPUT-feature tile, in the k
// order of publish_b16).  48 KB.
typedef uint32_t WimgT16[2][kNW][2][3][64][4];
template <int D, int NTH = kThreads>   // NTH: threads of the workgroup, kThreads or 2 * kThreads
__device__ inline void stage_wimgT16(WimgT16& img, const float* __restrict__ p, int tid) {
    constexpr int NK = 2 * kThreads / NTH;   // 512 (row-group, column-group) elements in all
    const ParamLayout L = make_layout(D, kH);
    if ((reinterpret_cast<uintptr_t>(p) & 15) == 0) {
        // a thread takes FOUR consecutive rows i of a weight matrix (the k index) and four columns: 4 x 16-byte loads (all
        // eight of a thread's loads are in flight at once), and the four rows' pieces of one column are 8 adjacent bytes of
        // the image -- 24 eight-byte LDS stores per thread (16-bit scatter stores: 96).
        // W[i][c]: i = 16 (2 ks + h) + 4 q + r is the k index, c = 16 wv + j' the output
        uint2* img64 = reinterpret_cast<uint2*>(&img[0][0][0][0][0][0]);
        f32x4 v[NK][4];
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            const int e = tid + NTH * k, c4 = e & 15, iq = (e >> 4) & 15, l = e >> 8;
#pragma unroll
            for (int r = 0; r < 4; ++r) v[k][r] = *reinterpret_cast<const f32x4*>(&p[L.w(1 + l) + (4 * iq + r) * kH + 4 * c4]);
        }
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            const int e = tid + NTH * k, c4 = e & 15, iq = (e >> 4) & 15, l = e >> 8;
            const int T = iq >> 2, q = iq & 3, ks = T >> 1;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int c = 4 * c4 + u, lane = 16 * q + (c & 15), wv = c >> 4;
                uint32_t pc[4][3];
#pragma unroll
                for (int r = 0; r < 4; ++r) split3_bf16(v[k][r][u], pc[r]);
#pragma unroll
                for (int pi = 0; pi < 3; ++pi) {
                    uint2 o;
                    o.x = (pc[0][pi] >> 16) | pc[1][pi];
                    o.y = (pc[2][pi] >> 16) | pc[3][pi];
                    img64[((((l * kNW + wv) * 2 + ks) * 3 + pi) * 64 + lane) * 2 + (T & 1)] = o;
                }
            }
        }
        return;
    }
    for (int idx = tid; idx < 2 * kNW * 2 * 64 * 4; idx += NTH) {
        const int v = idx & 3, lane = (idx >> 2) & 63, ks = (idx >> 8) & 1, wv = (idx >> 9) & 3, l = idx >> 11;
        const int col = 16 * wv + (lane & 15);
        const int e0 = 2 * v, e1 = 2 * v + 1;
        uint32_t lo[3], hi[3];
        split3_bf16(p[L.w(1 + l) + (16 * (2 * ks + (e0 >> 2)) + 4 * (lane >> 4) + (e0 & 3)) * kH + col], lo);
        split3_bf16(p[L.w(1 + l) + (16 * (2 * ks + (e1 >> 2)) + 4 * (lane >> 4) + (e1 & 3)) * kH + col], hi);
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) img[l][wv][ks][pc][lane][v] = (lo[pc] >> 16) | hi[pc];
    }
}

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
__device__ __forceinline__ void hidden_b16(const uint32_t (*xb16)[3][2][64][4], const bf16x8 (&a)[3][2], int lane,
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
        for (int s = 0; s < NS; ++s) acc[s] = mfma_b16(a[0][ks], b[s][2], acc[s]);
#pragma unroll
        for (int s = 0; s < NS; ++s) acc[s] = mfma_b16(a[2][ks], b[s][0], acc[s]);
#pragma unroll
        for (int s = 0; s < NS; ++s) acc[s] = mfma_b16(a[1][ks], b[s][1], acc[s]);
#pragma unroll
        for (int s = 0; s < NS; ++s) acc[s] = mfma_b16(a[0][ks], b[s][1], acc[s]);
#pragma unroll
        for (int s = 0; s < NS; ++s) acc[s] = mfma_b16(a[1][ks], b[s][0], acc[s]);
#pragma unroll
        for (int s = 0; s < NS; ++s) acc[s] = mfma_b16(a[0][ks], b[s][0], acc[s]);
    }
}
template <int NS>
__device__ __forceinline__ void hidden_b16(const uint32_t (*xb16)[3][2][64][4], const SplitWeightsB16& wb, int l, int lane,
                                           f32x4 (&acc)[NS]) {
    hidden_b16<NS>(xb16, wb.a[l], lane, acc);
}
// the transposed product: this wave's A operands come from its slice of the LDS image (stage_wimgT16)
template <int NS>
__device__ __forceinline__ void hidden_b16_lds(const uint32_t (*xb16)[3][2][64][4], const uint32_t (&slice)[2][3][64][4], int lane,
                                               f32x4 (&acc)[NS]) {
    bf16x8 a[3][2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int p = 0; p < 3; ++p) a[p][ks] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(&slice[ks][p][lane][0]));
    hidden_b16<NS>(xb16, a, lane, acc);
}

// ---- outer products over the batch rows on the bf16 pipe, operands read TRANSPOSED from the published piece buffers ------
// dW[i][k] = sum_rows u[i][row] x[k][row]: the contraction runs over the rows, which live on the lanes.  gfx950's
// ds_read_b64_tr_b16 reads a 4 x 16 block of 16-bit elements per 16-lane group and hands lane i the block's column i:
// with block rows = 4 batch rows and block columns = the 16 features of a tile, the piece buffers publish_b16 wrote
// ([piece][k-step][lane (q_l, row)][tile & 1][4 bf16 = features 4 q_l .. 4 q_l + 3]) are read as MFMA operands with the rows
// along k -- no transposing stores, no second copy.  K = 32 holds the 16 rows TWICE, so one v_mfma_f32_16x16x32_bf16 adds
// two piece products: A = (a_p0 | a_p1), B = (b_r0 | b_r1) gives a_p0 b_r0 + a_p1 b_r1.
// Byte offset (inside one stream's [3][2][64][4] block) of the 8 bytes lane (q, j) must SUPPLY for: piece, tile T, read rr.
__device__ __forceinline__ uint32_t tr_piece_offset(int piece, int T, int q, int j, int rr) {
    const int row = 8 * (q & 1) + 4 * rr + (j >> 2);          // lane 4q' + p' of the group supplies block row q' = j >> 2 ...
    return (uint32_t)(piece * 2048 + (T >> 1) * 1024 + (16 * (j & 3) + row) * 16 + (T & 1) * 8);   // ... columns 4p', p' = j & 3
}
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;
// EXEC must be all ones (every lane supplies an address).  The builtin, not inline asm: the compiler then counts the read
// in lgkmcnt itself (an asm read followed by a separate asm wait lets it schedule the consumer BEFORE the wait).
__device__ __forceinline__ s16x4 lds_read_tr16(const char* base, uint32_t off) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(base + off));
}
// one 16x16x32 operand: rows 8(q&1) .. +7 of piece (q >> 1 ? p1 : p0), tile T, from the stream block at `base`
__device__ __forceinline__ bf16x8 tr_operand(const char* base, int p0, int p1, int T, int q, int j) {
    const int piece = (q >> 1) ? p1 : p0;
    const s16x4 lo = lds_read_tr16(base, tr_piece_offset(piece, T, q, j, 0));
    const s16x4 hi = lds_read_tr16(base, tr_piece_offset(piece, T, q, j, 1));
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
}
// tmp[Ti] = sum over NS streams of (cotangent tile `wave`) x (layer input tile Ti), all six significant piece products.
// cot / xin: [stream][3][2][64][4] piece buffers (publish_b16).  All 64 lanes must call it (no divergence).
template <int NS>
__device__ __forceinline__ void outer_b16(const uint32_t (*cot)[3][2][64][4], const uint32_t (*xin)[3][2][64][4], int wave, int lane,
                                          f32x4 (&tmp)[4]) {
    const int q = lane >> 4, j = lane & 15;
#pragma unroll
    for (int Ti = 0; Ti < 4; ++Ti) tmp[Ti] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const char* ca = reinterpret_cast<const char*>(&cot[s][0][0][0][0]);
        const char* xa = reinterpret_cast<const char*>(&xin[s][0][0][0][0]);
        const bf16x8 A_hh = tr_operand(ca, 0, 0, wave, q, j), A_mm = tr_operand(ca, 1, 1, wave, q, j), A_hl = tr_operand(ca, 0, 2, wave, q, j);
        bf16x8 B_hm[4], B_lh[4];
#pragma unroll
        for (int Ti = 0; Ti < 4; ++Ti) {
            B_hm[Ti] = tr_operand(xa, 0, 1, Ti, q, j);
            B_lh[Ti] = tr_operand(xa, 2, 0, Ti, q, j);
        }
        // smallest products first; consecutive MFMAs go to different accumulators
#pragma unroll
        for (int Ti = 0; Ti < 4; ++Ti) tmp[Ti] = mfma_b16(A_hl, B_lh[Ti], tmp[Ti]);   // h.l + l.h
#pragma unroll
        for (int Ti = 0; Ti < 4; ++Ti) tmp[Ti] = mfma_b16(A_mm, B_hm[Ti], tmp[Ti]);   // m.h + m.m
#pragma unroll
        for (int Ti = 0; Ti < 4; ++Ti) tmp[Ti] = mfma_b16(A_hh, B_hm[Ti], tmp[Ti]);   // h.h + h.m
    }
}

// sum of the workgroups' gradient slabs in a fixed order (kernel in cnf_split.hip)
hipError_t launch_reduce_slab(const float* slab, int nwg, int P, float* out, hipStream_t s, int split_at = 0);
// slab sum (as launch_reduce_slab) + Adam update of the P parameters in one kernel; grad_out may be null; guard (nullable):
// the stats block of the adjoint solve that wrote the slabs -- a non-zero status there makes the kernel a no-op
hipError_t launch_adam_slabs(const float* slab, int ntiles, int P, int split_at, float* grad_out, float* p, float* m, float* v,
                             int* state, const CnfAdamOpts& o, const CnfStats* guard, hipStream_t s);

}  // namespace cnf
