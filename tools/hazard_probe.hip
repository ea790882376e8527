// hazard_probe.hip -- instruction-level probes for two anomalies of round 2:
//   (A) low halves of v_pk_fma_f32 results wrong in the last lane quarter, eight-wave workgroups only;
//   (B) a GPU memory fault of the H = 256 elu library built with the max-ILP scheduler.
// Each probe is a fixed instruction sequence in inline asm on hard-coded registers, run by "victim" waves 0..3 of ONE
// workgroup, optionally beside "aggressor" waves 4..7 on the same SIMDs, with a sweep over the number of wait states
// between producer and consumer.  A probe compares the value the consumer saw with the value read again after a long wait.
//   P_MFMA_PK : v_mfma_f32_16x16x32_bf16 (chain of 12 on 3 accumulators, as hidden_b16) -> v_pk_mul_f32 of the last one's result
//   P_MFMA_S  : the same -> v_mul_f32 (scalar) reads
//   P_SWAP_WAW: v_permlane16/32_swap vdst, src0 -> VALU write of src0 / vdst (write after write)
//   P_SWAP_PK : the same, the later write being the LOW half of a v_pk_mul_f32 destination pair
// Build: hipcc -O2 --offload-arch=gfx950 tools/hazard_probe.hip -o tools/hazard_probe.bin ; run on the GPU box: tools/hazard_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

constexpr int kIters = 20000;
constexpr int kMaxN = 12;

// aggressor kinds: 0 none (waves 4..7 exit at once), 1 back-to-back bf16 MFMAs, 2 VALU stream, 3 ds_read + MFMA mix
__device__ __forceinline__ void aggressor(int kind, volatile int* done, float* sink, const char* lds) {
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
    f32x4 acc0 = {0, 0, 0, 0}, acc1 = acc0, acc2 = acc0, acc3 = acc0;
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x & 3); b[i] = (__bf16)1.0f; }
    float v0 = threadIdx.x, v1 = 1.f, v2 = 2.f, v3 = 3.f;
    for (int guard = 0; guard < (1 << 20) && *done < 4; ++guard) {
        if (kind == 1 || kind == 3) {
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc1, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc2, 0, 0, 0);
                acc3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc3, 0, 0, 0);
                if (kind == 3) {
                    const uint4 r = *reinterpret_cast<const uint4*>(lds + ((threadIdx.x & 63) * 16 + k * 1024));
                    a = __builtin_bit_cast(bf16x8, r);
                }
            }
        } else if (kind == 7) {   // fp32 MFMA stream (v_mfma_f32_16x16x4_f32)
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(v1, v2, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(v1, v2, acc1, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(v1, v2, acc2, 0, 0, 0);
                acc3 = __builtin_amdgcn_mfma_f32_16x16x4f32(v1, v2, acc3, 0, 0, 0);
            }
        } else if (kind == 4) {   // DPP stream (the theta waves' reduce8)
#pragma unroll
            for (int k = 0; k < 32; ++k) {
                v0 += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v1), 0xB1, 0xF, 0xF, true));
                v1 += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v2), 0x4E, 0xF, 0xF, true));
                v2 += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v3), 0x124, 0xF, 0xF, true));
                v3 += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v0), 0x128, 0xF, 0xF, true));
            }
        } else if (kind == 5) {   // what a theta wave issues: transposed LDS reads, bf16 MFMAs, packed multiply-adds, DPP
            typedef short s16x4 __attribute__((ext_vector_type(4)));
            typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(lds + (threadIdx.x & 63) * 16 + k * 1024));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(lds + (threadIdx.x & 63) * 16 + k * 1024 + 8));
                typedef short s16x8 __attribute__((ext_vector_type(8)));
                const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                a = __builtin_bit_cast(bf16x8, v);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b, a, acc1, 0, 0, 0);
                acc2 += v0 * acc0;
                acc3 += v1 * acc1;
                v2 += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(acc2[0]), 0x124, 0xF, 0xF, true));
            }
        } else if (kind == 6) {   // packed multiply-add stream
#pragma unroll
            for (int k = 0; k < 32; ++k) {
                acc0 = acc0 * v0 + acc1;
                acc1 = acc1 * v1 + acc2;
                acc2 = acc2 * v2 + acc3;
                acc3 = acc3 * v3 + acc0;
            }
        } else {
#pragma unroll
            for (int k = 0; k < 64; ++k) {
                v0 = fmaf(v0, 1.0001f, v1);
                v1 = fmaf(v1, 0.9999f, v2);
                v2 = fmaf(v2, 1.0002f, v3);
                v3 = fmaf(v3, 0.9998f, v0);
            }
        }
    }
    sink[threadIdx.x] = acc0[0] + acc1[1] + acc2[2] + acc3[3] + v0 + v1 + v2 + v3;
}

#define NOPS_0 ""
#define NOPS_1 "s_nop 0\n"
#define NOPS_2 "s_nop 1\n"
#define NOPS_3 "s_nop 2\n"
#define NOPS_4 "s_nop 3\n"
#define NOPS_5 "s_nop 4\n"
#define NOPS_6 "s_nop 5\n"
#define NOPS_7 "s_nop 6\n"
#define NOPS_8 "s_nop 7\n"
#define NOPS_9 "s_nop 7\ns_nop 0\n"
#define NOPS_10 "s_nop 7\ns_nop 1\n"
#define NOPS_11 "s_nop 7\ns_nop 2\n"

// ---- MFMA -> VALU read ---------------------------------------------------------------------------------------------------
// v[40:43] = A, v[44:47] = B (bf16 pairs), v[48:51], v[52:55], v[56:59] = accumulators, v[60:63] = what the consumer saw
// (consumer: v_pk_mul_f32 by 1.0 held in v[64:65], or v_mul_f32), v[66:69] = the last accumulator read again after a long wait.
#define MFMA_CHAIN \
    "v_mov_b32 v48, 0x7fc00001\nv_mov_b32 v49, 0x7fc00001\nv_mov_b32 v50, 0x7fc00001\nv_mov_b32 v51, 0x7fc00001\n" \
    "v_mov_b32 v52, 0\nv_mov_b32 v53, 0\nv_mov_b32 v54, 0\nv_mov_b32 v55, 0\n" \
    "v_mov_b32 v56, 0\nv_mov_b32 v57, 0\nv_mov_b32 v58, 0\nv_mov_b32 v59, 0\n" \
    "v_mov_b32 v70, 0\nv_mov_b32 v71, 0\nv_mov_b32 v72, 0\nv_mov_b32 v73, 0\n" \
    "s_nop 7\n" \
    "v_mfma_f32_16x16x32_bf16 v[52:55], v[40:43], v[44:47], v[52:55]\n" \
    "v_mfma_f32_16x16x32_bf16 v[56:59], v[40:43], v[44:47], v[56:59]\n" \
    "v_mfma_f32_16x16x32_bf16 v[70:73], v[40:43], v[44:47], v[70:73]\n" \
    "v_mfma_f32_16x16x32_bf16 v[52:55], v[44:47], v[40:43], v[52:55]\n" \
    "v_mfma_f32_16x16x32_bf16 v[56:59], v[44:47], v[40:43], v[56:59]\n" \
    "v_mfma_f32_16x16x32_bf16 v[70:73], v[44:47], v[40:43], v[70:73]\n" \
    "v_mfma_f32_16x16x32_bf16 v[52:55], v[40:43], v[44:47], v[52:55]\n" \
    "v_mfma_f32_16x16x32_bf16 v[56:59], v[40:43], v[44:47], v[56:59]\n" \
    "v_mfma_f32_16x16x32_bf16 v[48:51], v[40:43], v[44:47], v[70:73]\n"
#define MFMA_PK(N) \
    asm volatile(MFMA_CHAIN NOPS_##N \
                 "v_pk_mul_f32 v[60:61], v[48:49], v[64:65]\nv_pk_mul_f32 v[62:63], v[50:51], v[64:65]\n" \
                 "s_nop 7\ns_nop 7\ns_nop 7\ns_nop 7\n" \
                 "v_mov_b32 v66, v48\nv_mov_b32 v67, v49\nv_mov_b32 v68, v50\nv_mov_b32 v69, v51\n" ::: CLOB)
#define MFMA_S(N) \
    asm volatile(MFMA_CHAIN NOPS_##N \
                 "v_mul_f32 v60, v48, v64\nv_mul_f32 v61, v49, v64\nv_mul_f32 v62, v50, v64\nv_mul_f32 v63, v51, v64\n" \
                 "s_nop 7\ns_nop 7\ns_nop 7\ns_nop 7\n" \
                 "v_mov_b32 v66, v48\nv_mov_b32 v67, v49\nv_mov_b32 v68, v50\nv_mov_b32 v69, v51\n" ::: CLOB)
#define CLOB "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59", \
             "v60","v61","v62","v63","v64","v65","v66","v67","v68","v69","v70","v71","v72","v73","memory"

// ---- v_permlane*_swap -> VALU write of one of its operands -----------------------------------------------------------------
// v60 = vdst, v62 = src0 of the swap (v63 pairs with v62); afterwards v62 (or v60) is overwritten with v66 (pair v[66:67]);
// the probe reads v62 / v60 back after a long wait: any lane that does not hold the overwriting value lost the later write.
#define SWAP_WAW(SW, N) \
    asm volatile("v_mov_b32 v60, v70\nv_mov_b32 v62, v71\nv_mov_b32 v63, v71\nv_mov_b32 v61, v70\ns_nop 3\n" \
                 SW " v60, v62\n" NOPS_##N \
                 "v_mov_b32 v62, v66\nv_mov_b32 v60, v66\n" \
                 "s_nop 7\ns_nop 7\ns_nop 7\n" ::: CLOB)
#define SWAP_PK(SW, N) \
    asm volatile("v_mov_b32 v60, v70\nv_mov_b32 v62, v71\nv_mov_b32 v63, v71\nv_mov_b32 v61, v70\ns_nop 3\n" \
                 SW " v60, v62\n" NOPS_##N \
                 "v_pk_mul_f32 v[62:63], v[66:67], v[64:65]\nv_pk_mul_f32 v[60:61], v[66:67], v[64:65]\n" \
                 "s_nop 7\ns_nop 7\ns_nop 7\n" ::: CLOB)

// ---- v_pk_fma_f32 with op_sel on the broadcast operand -------------------------------------------------------------------------
// v[80:81], v[82:83] = x pairs, v[84:85] = (z1, z0), v[86:87], v[88:89] = addends; packed results v[90:93]; the same four
// multiply-adds as scalar v_fma_f32 after a long wait: v[94:97].  OPS: the op_sel text; ZR: the register the scalar form reads.
#define PKFMA_BARE(OPS, ZR) \
    asm volatile("v_pk_fma_f32 v[90:91], v[80:81], v[84:85], v[86:87] " OPS "\nv_pk_fma_f32 v[92:93], v[82:83], v[84:85], v[88:89] " OPS "\n" \
                 "s_nop 7\ns_nop 7\n" \
                 "v_fma_f32 v94, v80, " ZR ", v86\nv_fma_f32 v95, v81, " ZR ", v87\nv_fma_f32 v96, v82, " ZR ", v88\nv_fma_f32 v97, v83, " ZR ", v89\n" \
                 "s_nop 3\n" ::: CLOB2)
// the kernel's neighbourhood (cnf_split.hip, W1 sums of an eight-wave producer): the addends come from v_pk_mul_f32 two
// instructions earlier, scalar multiply-adds of the other column sit in between, the results go to LDS as ds_write_b128 at once
#define PKFMA_CTX(OPS, ZR) \
    asm volatile("v_pk_mul_f32 v[88:89], v[98:99], v[88:89]\nv_pk_mul_f32 v[86:87], v[86:87], v[100:101]\n" \
                 "v_fmac_f32 v102, v80, v84\nv_fmac_f32 v103, v81, v84\nv_fmac_f32 v104, v82, v84\n" \
                 "v_pk_fma_f32 v[90:91], v[80:81], v[84:85], v[86:87] " OPS "\nv_pk_fma_f32 v[92:93], v[82:83], v[84:85], v[88:89] " OPS "\n" \
                 "v_fmac_f32 v105, v83, v84\n" \
                 "ds_write_b128 %0, v[90:93]\nds_write_b128 %0, v[102:105] offset:4096\ns_waitcnt lgkmcnt(0)\n" \
                 "s_nop 7\ns_nop 7\n" \
                 "v_fma_f32 v94, v80, " ZR ", v86\nv_fma_f32 v95, v81, " ZR ", v87\nv_fma_f32 v96, v82, " ZR ", v88\nv_fma_f32 v97, v83, " ZR ", v89\n" \
                 "ds_read_b128 v[90:93], %0\ns_waitcnt lgkmcnt(0)\ns_nop 3\n" :: "v"(lds_off) : CLOB2)
// the same neighbourhood around ANY packed instruction INS (destinations v[90:91], v[92:93]) with its scalar equivalent TRUTH (v94..v97)
#define PK_CTX(INS, TRUTH) \
    asm volatile("v_pk_mul_f32 v[88:89], v[98:99], v[88:89]\nv_pk_mul_f32 v[86:87], v[86:87], v[100:101]\n" \
                 "v_fmac_f32 v102, v80, v84\nv_fmac_f32 v103, v81, v84\nv_fmac_f32 v104, v82, v84\n" \
                 INS \
                 "v_fmac_f32 v105, v83, v84\n" \
                 "ds_write_b128 %0, v[90:93]\nds_write_b128 %0, v[102:105] offset:4096\ns_waitcnt lgkmcnt(0)\n" \
                 "s_nop 7\ns_nop 7\n" \
                 TRUTH \
                 "ds_read_b128 v[90:93], %0\ns_waitcnt lgkmcnt(0)\ns_nop 3\n" :: "v"(lds_off) : CLOB2)
#define CLOB2 "v80","v81","v82","v83","v84","v85","v86","v87","v88","v89","v90","v91","v92","v93","v94","v95","v96","v97","v98","v99","v100","v101", \
              "v102","v103","v104","v105","memory"

#define FOR_N(M, ...) \
    switch (n) { case 0: M(__VA_ARGS__ 0); break; case 1: M(__VA_ARGS__ 1); break; case 2: M(__VA_ARGS__ 2); break; case 3: M(__VA_ARGS__ 3); break; \
                 case 4: M(__VA_ARGS__ 4); break; case 5: M(__VA_ARGS__ 5); break; case 6: M(__VA_ARGS__ 6); break; case 7: M(__VA_ARGS__ 7); break; \
                 case 8: M(__VA_ARGS__ 8); break; case 9: M(__VA_ARGS__ 9); break; case 10: M(__VA_ARGS__ 10); break; default: M(__VA_ARGS__ 11); break; }
#define SW16(N) SWAP_WAW("v_permlane16_swap_b32", N)
#define SW32(N) SWAP_WAW("v_permlane32_swap_b32", N)
#define SWP16(N) SWAP_PK("v_permlane16_swap_b32", N)
#define SWP32(N) SWAP_PK("v_permlane32_swap_b32", N)

// probe: 0 MFMA_PK, 1 MFMA_S, 2 SW16, 3 SW32, 4 SWP16, 5 SWP32.  bad[probe][n][quarter][reg 0..3] counts lanes x iterations
__global__ __launch_bounds__(512) void probe_kernel(int probe, int n, int agg, unsigned long long* bad, float* sink) {
    __shared__ int done;
    __shared__ __attribute__((aligned(16))) char lds[16384];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, q = lane >> 4;
    if (tid == 0) done = 0;
    for (int i = tid; i < 16384 / 4; i += blockDim.x) reinterpret_cast<float*>(lds)[i] = 1.0f;
    __syncthreads();
    if (wave >= 4) {
        aggressor(agg, &done, sink, lds);
        return;
    }
    unsigned cnt[4] = {0, 0, 0, 0};
    for (int it = 0; it < kIters; ++it) {
        // operands: small integers as bf16 pairs, different every iteration so that a stale accumulator differs from a fresh one
        const unsigned ua = 0x3f803f80u + (((unsigned)(it + lane) & 3u) << 22);          // 1.0 .. in both halves (exponent nudged)
        const unsigned ub = 0x40004000u - (((unsigned)(it * 7 + (lane >> 2)) & 1u) << 23);  // 2.0 or 1.0
        const float fresh = (float)(it & 1023) + 0.5f * (float)lane;
        asm volatile("v_mov_b32 v40, %0\nv_mov_b32 v41, %0\nv_mov_b32 v42, %0\nv_mov_b32 v43, %0\n"
                     "v_mov_b32 v44, %1\nv_mov_b32 v45, %1\nv_mov_b32 v46, %1\nv_mov_b32 v47, %1\n"
                     "v_mov_b32 v64, 1.0\nv_mov_b32 v65, 1.0\nv_mov_b32 v66, %2\nv_mov_b32 v67, %2\n"
                     "v_mov_b32 v70, %3\nv_mov_b32 v71, %4\ns_nop 3\n"
                     :: "v"(ua), "v"(ub), "v"(fresh), "v"(-1.0f - (float)it), "v"(-2.0f - (float)lane) : CLOB);
        unsigned seen[4], truth[4];
        if (probe >= 6) { }
        else if (probe == 0) { FOR_N(MFMA_PK, ) }
        else if (probe == 1) { FOR_N(MFMA_S, ) }
        else if (probe == 2) { FOR_N(SW16, ) }
        else if (probe == 3) { FOR_N(SW32, ) }
        else if (probe == 4) { FOR_N(SWP16, ) }
        else { FOR_N(SWP32, ) }
        if (probe >= 6) {
            const float fa = 1.0f + 0.001f * (float)((it * 13 + lane) & 255), fb = -0.5f + 0.003f * (float)((it + 3 * lane) & 127);
            const unsigned lds_off = 8192u + (unsigned)tid * 16u - (unsigned)wave * 1024u + (unsigned)wave * 1024u;
            asm volatile("v_mov_b32 v80, %0\nv_mul_f32 v81, %0, %1\nv_add_f32 v82, %0, %1\nv_sub_f32 v83, %1, %0\n"
                         "v_mov_b32 v84, %2\nv_mov_b32 v85, %3\n"
                         "v_mul_f32 v86, %1, %1\nv_mul_f32 v87, %0, %0\nv_add_f32 v88, %1, %2\nv_add_f32 v89, %0, %3\n"
                         "v_mov_b32 v98, %1\nv_mov_b32 v99, %0\nv_mov_b32 v100, %3\nv_mov_b32 v101, %2\n"
                         "v_mov_b32 v102, %0\nv_mov_b32 v103, %1\nv_mov_b32 v104, %2\nv_mov_b32 v105, %3\ns_nop 3\n"
                         :: "v"(fa), "v"(fb), "v"(fresh), "v"(0.25f * fresh - 3.0f) : CLOB2);
            if (probe == 6) PKFMA_BARE("op_sel:[0,1,0]", "v85");
            else if (probe == 7) PKFMA_BARE("op_sel_hi:[1,0,1]", "v84");
            else if (probe == 8) PKFMA_CTX("op_sel:[0,1,0]", "v85");
            else if (probe == 9) PKFMA_CTX("op_sel_hi:[1,0,1]", "v84");
            else if (probe == 10)   // crossed low lane on src0
                PK_CTX("v_pk_fma_f32 v[90:91], v[84:85], v[80:81], v[86:87] op_sel:[1,0,0]\nv_pk_fma_f32 v[92:93], v[84:85], v[82:83], v[88:89] op_sel:[1,0,0]\n",
                       "v_fma_f32 v94, v85, v80, v86\nv_fma_f32 v95, v85, v81, v87\nv_fma_f32 v96, v85, v82, v88\nv_fma_f32 v97, v85, v83, v89\n");
            else if (probe == 11)   // crossed low lane on src2
                PK_CTX("v_pk_fma_f32 v[90:91], v[80:81], v[86:87], v[84:85] op_sel:[0,0,1]\nv_pk_fma_f32 v[92:93], v[82:83], v[88:89], v[84:85] op_sel:[0,0,1]\n",
                       "v_fma_f32 v94, v80, v86, v85\nv_fma_f32 v95, v81, v87, v85\nv_fma_f32 v96, v82, v88, v85\nv_fma_f32 v97, v83, v89, v85\n");
            else if (probe == 12)   // v_pk_mul_f32, crossed low lane on src1
                PK_CTX("v_pk_mul_f32 v[90:91], v[80:81], v[84:85] op_sel:[0,1]\nv_pk_mul_f32 v[92:93], v[82:83], v[84:85] op_sel:[0,1]\n",
                       "v_mul_f32 v94, v80, v85\nv_mul_f32 v95, v81, v85\nv_mul_f32 v96, v82, v85\nv_mul_f32 v97, v83, v85\n");
            else if (probe == 13)   // v_pk_add_f32, crossed low lane on src1
                PK_CTX("v_pk_add_f32 v[90:91], v[80:81], v[84:85] op_sel:[0,1]\nv_pk_add_f32 v[92:93], v[82:83], v[84:85] op_sel:[0,1]\n",
                       "v_add_f32 v94, v80, v85\nv_add_f32 v95, v81, v85\nv_add_f32 v96, v82, v85\nv_add_f32 v97, v83, v85\n");
            else if (probe == 14)   // no op_sel at all (control)
                PK_CTX("v_pk_fma_f32 v[90:91], v[80:81], v[84:85], v[86:87]\nv_pk_fma_f32 v[92:93], v[82:83], v[84:85], v[88:89]\n",
                       "v_fma_f32 v94, v80, v84, v86\nv_fma_f32 v95, v81, v85, v87\nv_fma_f32 v96, v82, v84, v88\nv_fma_f32 v97, v83, v85, v89\n");
            else if (probe == 15)   // crossed low lane on src1 with an independent instruction right before each packed one
                PK_CTX("s_nop 0\nv_pk_fma_f32 v[90:91], v[80:81], v[84:85], v[86:87] op_sel:[0,1,0]\ns_nop 0\nv_pk_fma_f32 v[92:93], v[82:83], v[84:85], v[88:89] op_sel:[0,1,0]\n",
                       "v_fma_f32 v94, v80, v85, v86\nv_fma_f32 v95, v81, v85, v87\nv_fma_f32 v96, v82, v85, v88\nv_fma_f32 v97, v83, v85, v89\n");
            else if (probe == 16)   // high lane crossed on src0 and src2 (op_sel_hi 0), low lane straight
                PK_CTX("v_pk_fma_f32 v[90:91], v[84:85], v[80:81], v[84:85] op_sel_hi:[0,1,0]\nv_pk_fma_f32 v[92:93], v[84:85], v[82:83], v[84:85] op_sel_hi:[0,1,0]\n",
                       "v_fma_f32 v94, v84, v80, v84\nv_fma_f32 v95, v84, v81, v84\nv_fma_f32 v96, v84, v82, v84\nv_fma_f32 v97, v84, v83, v84\n");
            else   // 17: both lanes read the high dword of src1 through op_sel, src1 = the destination pair of the previous packed op (fresh write)
                PK_CTX("v_pk_mul_f32 v[84:85], v[84:85], v[64:65]\nv_pk_fma_f32 v[90:91], v[80:81], v[84:85], v[86:87] op_sel:[0,1,0]\nv_pk_fma_f32 v[92:93], v[82:83], v[84:85], v[88:89] op_sel:[0,1,0]\n",
                       "v_fma_f32 v94, v80, v85, v86\nv_fma_f32 v95, v81, v85, v87\nv_fma_f32 v96, v82, v85, v88\nv_fma_f32 v97, v83, v85, v89\n");
            asm volatile("v_mov_b32 %0, v90\nv_mov_b32 %1, v91\nv_mov_b32 %2, v92\nv_mov_b32 %3, v93\n"
                         "v_mov_b32 %4, v94\nv_mov_b32 %5, v95\nv_mov_b32 %6, v96\nv_mov_b32 %7, v97\n"
                         : "=v"(seen[0]), "=v"(seen[1]), "=v"(seen[2]), "=v"(seen[3]), "=v"(truth[0]), "=v"(truth[1]), "=v"(truth[2]), "=v"(truth[3]) :: CLOB2);
        } else if (probe <= 1) {
            asm volatile("v_mov_b32 %0, v60\nv_mov_b32 %1, v61\nv_mov_b32 %2, v62\nv_mov_b32 %3, v63\n"
                         "v_mov_b32 %4, v66\nv_mov_b32 %5, v67\nv_mov_b32 %6, v68\nv_mov_b32 %7, v69\n"
                         : "=v"(seen[0]), "=v"(seen[1]), "=v"(seen[2]), "=v"(seen[3]), "=v"(truth[0]), "=v"(truth[1]), "=v"(truth[2]), "=v"(truth[3]) :: CLOB);
        } else {
            // the later write must have won in every lane: v62 (src0) -> seen[0], v60 (vdst) -> seen[1]
            asm volatile("v_mov_b32 %0, v62\nv_mov_b32 %1, v60\n" : "=v"(seen[0]), "=v"(seen[1]) :: CLOB);
            truth[0] = truth[1] = __float_as_uint(fresh);
            seen[2] = seen[3] = truth[2] = truth[3] = 0u;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) cnt[r] += seen[r] != truth[r];
    }
    if (lane == 0) atomicAdd(&done, 1);   // the aggressors run until all four victims are through (no barrier: they never arrive at one)
#pragma unroll
    for (int r = 0; r < 4; ++r)
        if (cnt[r]) atomicAdd(&bad[((probe * kMaxN + n) * 4 + q) * 4 + r], (unsigned long long)cnt[r]);
}

int main(int argc, char** argv) {
    unsigned long long* bad;
    float* sink;
    const size_t nb = 18 * kMaxN * 16;
    CHECK(hipMalloc(&bad, nb * sizeof(unsigned long long)));
    CHECK(hipMalloc(&sink, 512 * sizeof(float)));
    const char* names[18] = {"mfma16x16x32_bf16 -> v_pk_mul_f32", "mfma16x16x32_bf16 -> v_mul_f32", "permlane16_swap -> v_mov of operand",
                            "permlane32_swap -> v_mov of operand", "permlane16_swap -> v_pk_mul low half = operand", "permlane32_swap -> v_pk_mul low half = operand",
                             "v_pk_fma_f32 op_sel:[0,1,0], bare", "v_pk_fma_f32 op_sel_hi:[1,0,1], bare", "v_pk_fma_f32 op_sel:[0,1,0], kernel neighbourhood + ds_write_b128",
                             "v_pk_fma_f32 op_sel_hi:[1,0,1], kernel neighbourhood + ds_write_b128",
                             "neighbourhood: v_pk_fma_f32 op_sel:[1,0,0] (src0 crossed)", "neighbourhood: v_pk_fma_f32 op_sel:[0,0,1] (src2 crossed)",
                             "neighbourhood: v_pk_mul_f32 op_sel:[0,1]", "neighbourhood: v_pk_add_f32 op_sel:[0,1]", "neighbourhood: v_pk_fma_f32 without op_sel",
                             "neighbourhood: op_sel:[0,1,0] with s_nop 0 before each packed op", "neighbourhood: op_sel_hi:[0,1,0] (high lane crossed)",
                             "neighbourhood: op_sel:[0,1,0], src1 written by the packed op right before"};
    const char* aggs[8] = {"no second wave", "second wave: bf16 MFMA stream", "second wave: VALU stream", "second wave: ds_read + MFMA",
                           "second wave: DPP stream", "second wave: ds_read_tr + MFMA + packed fma + DPP", "second wave: packed fma stream", "second wave: fp32 MFMA stream"};
    const bool pk_only = argc > 1 && !strcmp(argv[1], "pk");
    unsigned long long host[nb];
    for (int agg = 0; agg < 8; ++agg) {
        if (pk_only && agg != 0 && agg != 1 && agg != 5 && agg != 7) continue;
        CHECK(hipMemset(bad, 0, nb * sizeof(unsigned long long)));
        for (int probe = pk_only ? 6 : 0; probe < (pk_only ? 18 : 6); ++probe)
            for (int n = 0; n < (probe >= 6 ? 3 : kMaxN); ++n) {
                hipLaunchKernelGGL(probe_kernel, dim3(1), dim3(agg == 0 ? 256 : 512), 0, 0, probe, n, agg, bad, sink);
                CHECK(hipGetLastError());
            }
        CHECK(hipDeviceSynchronize());
        CHECK(hipMemcpy(host, bad, nb * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        printf("== %s (%d iterations x 4 victim waves; counts = lane-iterations with a wrong value, per lane quarter q0..q3, registers r0..r3)\n", aggs[agg], kIters);
        for (int probe = pk_only ? 6 : 0; probe < (pk_only ? 18 : 6); ++probe) {
            printf("  %s\n", names[probe]);
            for (int n = 0; n < (probe >= 6 ? 3 : kMaxN); ++n) {
                unsigned long long tot = 0;
                for (int k = 0; k < 16; ++k) tot += host[(probe * kMaxN + n) * 16 + k];
                printf("    wait states %2d: total %10llu |", n, tot);
                if (tot) for (int q = 0; q < 4; ++q) { printf(" q%d:", q); for (int r = 0; r < 4; ++r) printf(" %llu", host[((probe * kMaxN + n) * 4 + q) * 4 + r]); }
                printf("\n");
            }
        }
    }
    return 0;
}
