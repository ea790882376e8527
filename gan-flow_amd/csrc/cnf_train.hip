// cnf_train.hip -- the two small pieces of a training iteration that sit between the solves: the flow's
// negative log-likelihood with its gradient w.r.t. the solver outputs, and the Adam
// update of a flat parameter vector (section 8f row N2).  With them an iteration is forward solve -> loss ->
// adjoint solve -> update: four kernels and no framework glue in between.
#include <hip/hip_runtime.h>

#include "../../include/cnf_hip.h"
#include "cnf_adam.h"

namespace cnf {

constexpr int kLossThreads = 1024;

// loss = -(1/B) sum_i [ sum_d (-0.5 log 2pi - z_id^2 / 2) - dlogp_i ];  dL/dz = z / B;  dL/d(dlogp) = 1 / B.
// One workgroup: every thread sums its rows in a fixed order, then a fixed LDS tree -> the same bits on every run.
__global__ __launch_bounds__(kLossThreads) void nll_loss_kernel(const float* __restrict__ z, const float* __restrict__ dlogp,
                                                                long rows, int dim, float* __restrict__ loss_out,
                                                                float* __restrict__ grad_z, float* __restrict__ grad_dlogp) {
    __shared__ float part[kLossThreads];
    const float kHalfLog2Pi = 0.91893853320467274178f;
    const float inv = 1.0f / (float)rows;
    float acc = 0.f;
    for (long i = threadIdx.x; i < rows; i += kLossThreads) {
        float lp = 0.f;
        for (int m = 0; m < dim; ++m) {
            const float v = z[i * dim + m];
            lp += -kHalfLog2Pi - v * v * 0.5f;
            if (grad_z) grad_z[i * dim + m] = v * inv;
        }
        acc += lp - dlogp[i];
        if (grad_dlogp) grad_dlogp[i] = inv;
    }
    part[threadIdx.x] = acc;
    __syncthreads();
    for (int s = kLossThreads / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) part[threadIdx.x] += part[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) *loss_out = -(part[0] * inv);
}

// torch.optim.Adam on a flat vector (arithmetic: cnf_adam.h).  `state` = {step, ticket}: every workgroup reads step, the
// last one to finish stores step + 1.
constexpr int kAdamThreads = 256;

__global__ __launch_bounds__(kAdamThreads) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                            float* __restrict__ m, float* __restrict__ v,
                                                            int* __restrict__ state, long n, CnfAdamOpts o,
                                                            const CnfStats* __restrict__ guard) {
    if (guard && guard->status != 0) return;   // the solve that produced g failed: no update, no step (uniform exit)
    __shared__ float corr[2];
    AdamCoef c;
    const int step = adam_begin(state, o, corr, c);
    const long i = (long)blockIdx.x * kAdamThreads + threadIdx.x;
    if (i < n) adam_update(c, g[i], p, m, v, i);
    adam_end(state, step);
}

// Diagnostics: overwrite the whole LDS of every CU with a bit pattern (cnf_debug_fill_lds).  LDS keeps what the last
// kernel on the CU left behind; the tests poison it with NaNs to prove no solve depends on that content.
__global__ __launch_bounds__(256) void fill_lds_kernel(unsigned pattern, unsigned* __restrict__ sink) {
    __shared__ unsigned buf[160 * 1024 / 4];
    for (int i = threadIdx.x; i < 160 * 1024 / 4; i += 256) buf[i] = pattern;
    __syncthreads();
    if (buf[(threadIdx.x * 97 + blockIdx.x) % (160 * 1024 / 4)] != pattern) *sink = 1u;   // keeps the stores alive
}

hipError_t launch_fill_lds(unsigned pattern, unsigned* sink, hipStream_t s) {
    hipLaunchKernelGGL(fill_lds_kernel, dim3(4096), dim3(256), 0, s, pattern, sink);
    return hipGetLastError();
}

hipError_t launch_nll_loss(const float* z, const float* dlogp, long rows, int dim, float* loss_out, float* grad_z,
                           float* grad_dlogp, hipStream_t s) {
    hipLaunchKernelGGL(nll_loss_kernel, dim3(1), dim3(kLossThreads), 0, s, z, dlogp, rows, dim, loss_out, grad_z, grad_dlogp);
    return hipGetLastError();
}

hipError_t launch_adam(float* p, const float* g, float* m, float* v, int* state, long n, const CnfAdamOpts& o,
                       const CnfStats* guard, hipStream_t s) {
    const int blocks = (int)((n + kAdamThreads - 1) / kAdamThreads);
    hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(kAdamThreads), 0, s, p, g, m, v, state, n, o, guard);
    return hipGetLastError();
}

}  // namespace cnf
