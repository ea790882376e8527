// mfma_valu_overlap.hip -- does vector-ALU work of ONE wave run in the shadow of its own fp32 MFMAs on gfx950?
// Each variant runs ITERS iterations of a fixed instruction pattern (inline asm, so the compiler cannot reorder it):
//   0: 48 x v_mfma_f32_16x16x4_f32 (three accumulator chains, as a hidden layer of the split kernels issues them)
//   1: the same 48 MFMAs with 4 independent v_fma_f32 behind each one
//   2: the 192 v_fma_f32 alone
//   3: the 48 MFMAs with one v_exp_f32 + 2 v_fma_f32 behind each one
//   4: the 48 v_exp_f32 + 96 v_fma_f32 alone
//   5: waves 0-3 run pattern 0 (MFMAs), waves 4-7 pattern 2 (v_fma): do two WAVES of one SIMD overlap the two kinds?
//   6: waves 0-3 run pattern 0, waves 4-7 pattern 4 (v_exp + v_fma)
//   7: 48 x v_mfma_f32_16x16x16_bf16 (the XDL matrix pipe; 4x the FLOP of a pattern-0 MFMA each)
//   8: those 48 bf16 MFMAs with 4 independent v_fma_f32 behind each one
// launched as one workgroup of 4 waves (one per SIMD) and of 8 waves (two per SIMD).  Prints shader clocks per iteration
// for the slowest of waves 0-3 and of waves 4-7 (the arbiter favours the older wave: one wave's clock says little).
// build: hipcc -O2 --offload-arch=gfx950 tools/mfma_valu_overlap.hip -o tools/mfma_valu_overlap.bin
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int ITERS = 2000;

#define MFMA(acc) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
#define MFMAB(acc) asm volatile("v_mfma_f32_16x16x16_bf16 %0, %1, %1, %0" : "+v"(acc) : "v"(ab))
#define FMA(x) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b))
#define EXP(x) asm volatile("v_exp_f32 %0, %0" : "+v"(x))

template <int V0>
__global__ void k(float* out, long long* cycles) {
    const int V = (V0 != 5 && V0 != 6) ? V0 : (threadIdx.x < 256 ? 0 : (V0 == 5 ? 2 : 4));
    float a = 1.0f + threadIdx.x * 1e-6f, b = 0.5f;
    f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0;
    const double ab = (double)threadIdx.x;   // any 64 bits: four bf16 per lane
    float x0 = 0.1f, x1 = 0.2f, x2 = 0.3f, x3 = 0.4f;
    __syncthreads();
    const long long t0 = clock64();
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (V == 0 || V == 1 || V == 3) MFMA(c0);
            if (V == 7 || V == 8) { MFMAB(c0); }
            if (V == 8) { FMA(x0); FMA(x1); FMA(x2); FMA(x3); }
            if (V == 7 || V == 8) { MFMAB(c1); }
            if (V == 8) { FMA(x0); FMA(x1); FMA(x2); FMA(x3); }
            if (V == 7 || V == 8) { MFMAB(c2); }
            if (V == 8) { FMA(x0); FMA(x1); FMA(x2); FMA(x3); }
            if (V == 1 || V == 2) { FMA(x0); FMA(x1); FMA(x2); FMA(x3); }
            if (V == 3 || V == 4) { EXP(x0); FMA(x1); FMA(x2); }
            if (V == 0 || V == 1 || V == 3) MFMA(c1);
            if (V == 1 || V == 2) { FMA(x0); FMA(x1); FMA(x2); FMA(x3); }
            if (V == 3 || V == 4) { EXP(x3); FMA(x1); FMA(x2); }
            if (V == 0 || V == 1 || V == 3) MFMA(c2);
            if (V == 1 || V == 2) { FMA(x0); FMA(x1); FMA(x2); FMA(x3); }
            if (V == 3 || V == 4) { EXP(x0); FMA(x1); FMA(x2); }
        }
    }
    asm volatile("s_nop 15\n s_nop 15");
    const long long t1 = clock64();
    if ((threadIdx.x & 63) == 0) cycles[threadIdx.x >> 6] = t1 - t0;
    out[threadIdx.x] = c0[0] + c1[1] + c2[2] + x0 + x1 + x2 + x3;
}

template <int V>
void run(const char* what, float* out, long long* cyc) {
    for (int threads : {256, 512}) {
        if ((V == 5 || V == 6) && threads == 256) continue;
        long long h[8] = {};
        hipLaunchKernelGGL(k<V>, dim3(1), dim3(threads), 0, 0, out, cyc);
        hipLaunchKernelGGL(k<V>, dim3(1), dim3(threads), 0, 0, out, cyc);
        if (hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return;
        long long lo = 0, hi = 0;
        for (int w = 0; w < 4; ++w) lo = h[w] > lo ? h[w] : lo;
        for (int w = 4; w < threads / 64; ++w) hi = h[w] > hi ? h[w] : hi;
        printf("%-44s %d waves/SIMD: waves 0-3 %8.1f, waves 4-7 %8.1f clocks per iteration\n", what, threads / 256,
               (double)lo / ITERS, (double)hi / ITERS);
    }
}

int main() {
    float* out;
    long long* cyc;
    if (hipMalloc(&out, 512 * sizeof(float)) != hipSuccess || hipMalloc(&cyc, 8 * sizeof(long long)) != hipSuccess) return 1;
    run<0>("48 MFMA 16x16x4 f32", out, cyc);
    run<1>("48 MFMA + 192 v_fma interleaved", out, cyc);
    run<2>("192 v_fma alone", out, cyc);
    run<3>("48 MFMA + 48 v_exp + 96 v_fma interleaved", out, cyc);
    run<4>("48 v_exp + 96 v_fma alone", out, cyc);
    run<5>("waves 0-3: 48 MFMA | waves 4-7: 192 v_fma", out, cyc);
    run<6>("waves 0-3: 48 MFMA | waves 4-7: 48 v_exp + 96 v_fma", out, cyc);
    run<7>("48 MFMA 16x16x16 bf16", out, cyc);
    run<8>("48 MFMA bf16 + 192 v_fma interleaved", out, cyc);
    (void)hipFree(out);
    (void)hipFree(cyc);
    return 0;
}
