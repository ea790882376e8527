// cnf_data.hip -- the training data of the study drawn ON the device: the Gaussian + uniform
// mixture of /root/reference/gu.py:35-68 (get_sample), which the reference draws with numpy on the host and copies over
// every iteration.  Same distribution, another random stream: the reference's rows are i.i.d. (component uniform over the
// mixtures, then uniform part with probability unif_ratio / (unif_ratio + 1), else Gaussian; its per-component binomial
// split + final shuffle is exactly that), so every row is drawn independently here from a counter-based generator
// (Philox4x32-10 keyed by the seed, counter = (row, call number)): no state per row, reproducible, graph-capturable --
// the call number lives in device memory and is advanced by the kernel itself.
#include <hip/hip_runtime.h>

#include "../../include/cnf_hip.h"

namespace cnf {

__device__ __forceinline__ void philox4x32_10(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c[0]), lo0 = 0xD2511F53u * c[0];
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c[2]), lo1 = 0xCD9E8D57u * c[2];
        const uint32_t n0 = hi1 ^ c[1] ^ k0, n2 = hi0 ^ c[3] ^ k1;
        c[0] = n0;
        c[1] = lo1;
        c[2] = n2;
        c[3] = lo0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
}

__device__ __forceinline__ float u01(uint32_t r) { return (float)(r >> 8) * 5.9604644775390625e-08f; }           // [0, 1)
__device__ __forceinline__ float u01_open(uint32_t r) { return (float)((r >> 8) + 1u) * 5.9604644775390625e-08f; }   // (0, 1]

constexpr int kGuThreads = 256;

// one row of the mixture: counter = (row, call number), second Philox block (word 3 = 1) for the nuisance column
__device__ __forceinline__ void gu_draw_row(const CnfGuDesc& d, unsigned long long seed, uint32_t call, unsigned long long row,
                                            float* __restrict__ dst) {
    uint32_t c[4] = {(uint32_t)row, (uint32_t)(row >> 32), call, 0u};
    philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    const int comp = (int)__umulhi(c[0], (uint32_t)d.n_mixture);          // uniform over the components
    const float mean = d.mean[comp], sigma = d.sigma[comp];
    float v;
    if (u01(c[1]) < d.unif_p[comp]) {
        const float half = d.unif_intsect[comp] * sigma;                  // U(mean - half, mean + half)
        v = fmaf(2.0f * half, u01(c[2]), mean - half);
    } else {                                                             // N(mean, sigma^2), Box-Muller
        const float rad = sqrtf(-2.0f * logf(u01_open(c[2])));
        v = fmaf(sigma * rad, cospif(2.0f * u01(c[3])), mean);
    }
    if (d.extend_dim) {
        uint32_t e[4] = {(uint32_t)row, (uint32_t)(row >> 32), call, 1u};
        philox4x32_10(e, (uint32_t)seed, (uint32_t)(seed >> 32));
        dst[0] = v;
        dst[1] = u01(e[0]);                                               // the U(0, 1) nuisance column (gu.py:63-66)
    } else {
        dst[0] = v;
    }
}

__global__ __launch_bounds__(kGuThreads) void gu_sample_kernel(CnfGuDesc d, unsigned long long seed, int* __restrict__ state,
                                                               long rows, float* __restrict__ out) {
    const int call = __hip_atomic_load(&state[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const long row = (long)blockIdx.x * kGuThreads + threadIdx.x;
    if (row < rows) gu_draw_row(d, seed, (uint32_t)call, (unsigned long long)row, out + (d.extend_dim ? 2 : 1) * row);
    // every workgroup has read the call number: the last one to finish advances it (as adam_end does for the step count)
    __syncthreads();
    if (threadIdx.x == 0) {
        const int done = __hip_atomic_fetch_add(&state[1], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (done == (int)gridDim.x - 1) {
            __hip_atomic_store(&state[1], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&state[0], call + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// Rows row0 .. row0 + rows - 1 of draw number `call`: the same generator addressed by the GLOBAL row index, stateless.  A
// shard of a data set drawn this way holds the same values whatever the number of shards (sharded evaluation, config 3).
__global__ __launch_bounds__(kGuThreads) void gu_sample_rows_kernel(CnfGuDesc d, unsigned long long seed, unsigned call,
                                                                    long row0, long rows, float* __restrict__ out) {
    const long i = (long)blockIdx.x * kGuThreads + threadIdx.x;
    if (i < rows) gu_draw_row(d, seed, call, (unsigned long long)(row0 + i), out + (d.extend_dim ? 2 : 1) * i);
}

hipError_t launch_gu_sample(const CnfGuDesc& d, unsigned long long seed, int* state, long rows, float* out, hipStream_t s) {
    const int blocks = (int)((rows + kGuThreads - 1) / kGuThreads);
    hipLaunchKernelGGL(gu_sample_kernel, dim3(blocks), dim3(kGuThreads), 0, s, d, seed, state, rows, out);
    return hipGetLastError();
}

hipError_t launch_gu_sample_rows(const CnfGuDesc& d, unsigned long long seed, unsigned call, long row0, long rows, float* out,
                                 hipStream_t s) {
    const int blocks = (int)((rows + kGuThreads - 1) / kGuThreads);
    hipLaunchKernelGGL(gu_sample_rows_kernel, dim3(blocks), dim3(kGuThreads), 0, s, d, seed, call, row0, rows, out);
    return hipGetLastError();
}

}  // namespace cnf
