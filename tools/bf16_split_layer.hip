// bf16_split_layer.hip -- prototype for DESIGN.md section 9 item 1: one 64 -> 64 hidden layer of a 16-row tile (three streams:
// value + two tangents), this wave's 16 output features, computed
//   (a) as the split kernels do now: 48 x v_mfma_f32_16x16x4_f32 on fp32 operands, and
//   (b) on the bf16 matrix pipe: every fp32 operand written as hi + mid + lo (three truncated bf16 pieces, 24 bits), NP of the
//       nine piece products kept (6: weight >= 2^-16 ... hh hm mh hl mm lh; 8: + ml lm), v_mfma_f32_16x16x32_bf16, fp32 sums.
// The inputs arrive in the C/D register layout of the previous layer (lane (q, j): features 16T + 4q + r of row j), which is
// also a valid B operand of the 16x16x32 MFMA once the weight image uses the same k order: k-step s takes T = 2s, 2s+1.
// Prints the error of both against float64 and shader clocks per layer in a loop (the split of this wave's own 12 values
// is inside the timed loop, as it would be before publishing them; weights are split once, outside).
// build: hipcc -O3 --offload-arch=gfx950 tools/bf16_split_layer.hip -o tools/bf16_split_layer.bin
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
constexpr int H = 64, ROWS = 16, NS = 3, ITERS = 2000;

__host__ __device__ inline uint32_t fbits(float x) { union { float f; uint32_t u; } v; v.f = x; return v.u; }
__host__ __device__ inline float bitsf(uint32_t u) { union { float f; uint32_t u; } v; v.u = u; return v.f; }
// truncated bf16 pieces: x = p0 + p1 + p2 + (|rest| <= 2^-24 |x|); every subtraction is exact
__host__ __device__ inline void split3(float x, uint32_t (&p)[3]) {
    p[0] = fbits(x) & 0xffff0000u;
    const float r1 = x - bitsf(p[0]);
    p[1] = fbits(r1) & 0xffff0000u;
    const float r2 = r1 - bitsf(p[1]);
    p[2] = fbits(r2) & 0xffff0000u;
}
__device__ inline uint32_t pack2(uint32_t lo_elem, uint32_t hi_elem) { return (lo_elem >> 16) | hi_elem; }   // both: bf16 in the upper half

// wimg32: [T][lane][r] = W[16*wave + (lane & 15)][16T + 4(lane >> 4) + r]               (fp32 A operands, wave = 0 here)
// wimg16: [piece][s][lane][8] bf16 = pieces of W[lane & 15][16T + 4(lane >> 4) + r], T = 2s + (e >> 2), r = e & 3
// x:      [stream][T][lane][r]  fp32 inputs in C/D layout
template <int MODE>   // 0 fp32, 6 / 8: bf16 with that many piece products
__global__ void layer(const float* __restrict__ wimg32, const uint32_t* __restrict__ wimg16, const float* __restrict__ x,
                      float* __restrict__ out, long long* cycles, int iters) {
    const int lane = threadIdx.x & 63;
    f32x4 in[NS][4];
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int T = 0; T < 4; ++T) in[s][T] = *reinterpret_cast<const f32x4*>(&x[((s * 4 + T) * 64 + lane) * 4]);
    f32x4 a32[4];
    bf16x8 a16[3][2];
    if (MODE == 0) {
#pragma unroll
        for (int T = 0; T < 4; ++T) a32[T] = *reinterpret_cast<const f32x4*>(&wimg32[(T * 64 + lane) * 4]);
    } else {
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const u32x4 v = *reinterpret_cast<const u32x4*>(&wimg16[((p * 2 + s) * 64 + lane) * 4]);
                a16[p][s] = __builtin_bit_cast(bf16x8, v);
            }
    }
    f32x4 acc[NS];
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < NS; ++s) acc[s] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (MODE == 0) {
#pragma unroll
            for (int T = 0; T < 4; ++T)
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int s = 0; s < NS; ++s) acc[s] = __builtin_amdgcn_mfma_f32_16x16x4f32(a32[T][r], in[s][T][r], acc[s], 0, 0, 0);
        } else {
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                // split the 16 input values of this lane (in the kernel: 4 of them per wave, before publishing)
                bf16x8 b[3][2];
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    uint32_t pc[8][3];
#pragma unroll
                    for (int e = 0; e < 8; ++e) split3(in[s][2 * ks + (e >> 2)][e & 3], pc[e]);
#pragma unroll
                    for (int p = 0; p < 3; ++p) {
                        const u32x4 v = {pack2(pc[0][p], pc[1][p]), pack2(pc[2][p], pc[3][p]), pack2(pc[4][p], pc[5][p]), pack2(pc[6][p], pc[7][p])};
                        b[p][ks] = __builtin_bit_cast(bf16x8, v);
                    }
                }
                // smallest products first
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    if (MODE == 8) {
                        acc[s] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a16[1][ks], b[2][ks], acc[s], 0, 0, 0);
                        acc[s] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a16[2][ks], b[1][ks], acc[s], 0, 0, 0);
                    }
                    acc[s] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a16[0][ks], b[2][ks], acc[s], 0, 0, 0);
                    acc[s] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a16[2][ks], b[0][ks], acc[s], 0, 0, 0);
                    acc[s] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a16[1][ks], b[1][ks], acc[s], 0, 0, 0);
                }
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    acc[s] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a16[0][ks], b[1][ks], acc[s], 0, 0, 0);
                    acc[s] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a16[1][ks], b[0][ks], acc[s], 0, 0, 0);
                }
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) acc[s] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a16[0][ks], b[0][ks], acc[s], 0, 0, 0);
            }
        }
        // feed the result back so that no iteration can be hoisted or dropped (adds 0 in exact arithmetic: 1e-30 * acc)
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int r = 0; r < 4; ++r) in[s][0][r] += 1e-30f * acc[s][r];
    }
    const long long t1 = clock64();
    if (threadIdx.x == 0) cycles[0] = t1 - t0;
#pragma unroll
    for (int s = 0; s < NS; ++s) *reinterpret_cast<f32x4*>(&out[(s * 64 + lane) * 4]) = acc[s];
}

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

int main() {
    std::vector<float> W(H * H), X(NS * H * ROWS);
    srand(1);
    for (auto& w : W) w = (rand() / (float)RAND_MAX * 2.f - 1.f) / 8.f;
    for (auto& v : X) v = tanhf((rand() / (float)RAND_MAX * 2.f - 1.f) * 2.f);   // activations in (-1, 1); X[s][feature][row]
    std::vector<float> wimg32(4 * 64 * 4), ximg(NS * 4 * 64 * 4);
    std::vector<uint32_t> wimg16(3 * 2 * 64 * 4);
    for (int lane = 0; lane < 64; ++lane) {
        const int q = lane >> 4, i = lane & 15;
        for (int T = 0; T < 4; ++T)
            for (int r = 0; r < 4; ++r) {
                wimg32[(T * 64 + lane) * 4 + r] = W[i * H + 16 * T + 4 * q + r];
                for (int s = 0; s < NS; ++s) ximg[((s * 4 + T) * 64 + lane) * 4 + r] = X[(s * H + 16 * T + 4 * q + r) * ROWS + i];
            }
        for (int ks = 0; ks < 2; ++ks) {
            uint32_t pc[8][3];
            for (int e = 0; e < 8; ++e) split3(W[i * H + 16 * (2 * ks + (e >> 2)) + 4 * q + (e & 3)], pc[e]);
            for (int p = 0; p < 3; ++p)
                for (int v = 0; v < 4; ++v) wimg16[((p * 2 + ks) * 64 + lane) * 4 + v] = (pc[2 * v][p] >> 16) | pc[2 * v + 1][p];
        }
    }
    float *dw32, *dx, *dout;
    uint32_t* dw16;
    long long* dcyc;
    CHECK(hipMalloc(&dw32, wimg32.size() * 4));
    CHECK(hipMalloc(&dw16, wimg16.size() * 4));
    CHECK(hipMalloc(&dx, ximg.size() * 4));
    CHECK(hipMalloc(&dout, NS * 64 * 4 * 4));
    CHECK(hipMalloc(&dcyc, 8));
    CHECK(hipMemcpy(dw32, wimg32.data(), wimg32.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dw16, wimg16.data(), wimg16.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dx, ximg.data(), ximg.size() * 4, hipMemcpyHostToDevice));
    const char* names[3] = {"fp32: 48 x mfma_f32_16x16x4_f32", "bf16 x 6 products: 36 x mfma_f32_16x16x32_bf16", "bf16 x 8 products: 48 x mfma_f32_16x16x32_bf16"};
    for (int m = 0; m < 3; ++m) {
        std::vector<float> out(NS * 64 * 4);
        long long cyc = 0;
        for (int pass = 0; pass < 2; ++pass) {   // pass 0: one iteration, for the numbers; pass 1: the timed loop
            const int iters = pass == 0 ? 1 : ITERS;
            if (m == 0) hipLaunchKernelGGL(layer<0>, dim3(1), dim3(64), 0, 0, dw32, dw16, dx, dout, dcyc, iters);
            if (m == 1) hipLaunchKernelGGL(layer<6>, dim3(1), dim3(64), 0, 0, dw32, dw16, dx, dout, dcyc, iters);
            if (m == 2) hipLaunchKernelGGL(layer<8>, dim3(1), dim3(64), 0, 0, dw32, dw16, dx, dout, dcyc, iters);
            CHECK(hipDeviceSynchronize());
            if (pass == 0) CHECK(hipMemcpy(out.data(), dout, out.size() * 4, hipMemcpyDeviceToHost));
            CHECK(hipMemcpy(&cyc, dcyc, 8, hipMemcpyDeviceToHost));
        }
        double worst = 0, rms = 0, scale = 0;
        for (int s = 0; s < NS; ++s)
            for (int lane = 0; lane < 64; ++lane)
                for (int r = 0; r < 4; ++r) {
                    const int o = 4 * (lane >> 4) + r, row = lane & 15;   // C/D layout: output feature o of row `row`
                    double ref = 0;
                    for (int k = 0; k < H; ++k) ref += (double)W[o * H + k] * (double)X[(s * H + k) * ROWS + row];
                    const double err = fabs((double)out[(s * 64 + lane) * 4 + r] - ref);
                    worst = err > worst ? err : worst;
                    rms += err * err;
                    scale += ref * ref;
                }
        printf("%-48s max |err| %.3e, rms err / rms value %.3e, %8.1f clocks per layer (3 streams, 16 features x 16 rows)\n", names[m], worst,
               sqrt(rms / scale), (double)cyc / ITERS);
    }
    return 0;
}
