// cnf_train.hip -- the two small pieces of a training iteration that sit between the solves: the flow's
// negative log-likelihood with its gradient w.r.t. the solver outputs, and the Adam
// update of a flat parameter vector (section 8f row N2).  With them an iteration is forward solve -> loss ->
// adjoint solve -> update: four kernels and no framework glue in between.
#include <hip/hip_runtime.h>

#include "../../include/cnf_hip.h"

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

// torch.optim.Adam (amsgrad=False, maximize=False; weight decay added to the gradient):
//   g += wd * p;  m += (g - m)(1 - b1);  v = b2 v + (1 - b2) g^2;
//   p -= lr / (1 - b1^t) * m / (sqrt(v) / sqrt(1 - b2^t) + eps),     t = step + 1.
// `state` = {step, ticket}: every workgroup reads step, the last one to finish (ticket) stores step + 1.
constexpr int kAdamThreads = 256;

__global__ __launch_bounds__(kAdamThreads) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                            float* __restrict__ m, float* __restrict__ v,
                                                            int* __restrict__ state, long n, CnfAdamOpts o) {
    __shared__ float corr[2];
    const float b2 = (float)o.beta2, w1 = (float)(1.0 - o.beta1), w2 = (float)(1.0 - o.beta2);
    const float wd = (float)o.weight_decay, eps = (float)o.eps;
    const int step = __hip_atomic_load(&state[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (threadIdx.x == 0) {
        const double t = (double)(step + 1);
        const double bc1 = 1.0 - pow(o.beta1, t), bc2 = 1.0 - pow(o.beta2, t);
        corr[0] = (float)(o.lr / bc1);
        corr[1] = (float)sqrt(bc2);
    }
    __syncthreads();
    const float step_size = corr[0], bc2_sqrt = corr[1];
    const long i = (long)blockIdx.x * kAdamThreads + threadIdx.x;
    if (i < n) {
        const float pi = p[i];
        const float gi = fmaf(wd, pi, g[i]);
        const float mi = fmaf(gi - m[i], w1, m[i]);
        const float vi = fmaf(b2, v[i], w2 * gi * gi);
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        p[i] = pi - step_size * (mi / denom);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const int done = __hip_atomic_fetch_add(&state[1], 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (done == (int)gridDim.x - 1) {
            __hip_atomic_store(&state[1], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&state[0], step + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
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

hipError_t launch_adam(float* p, const float* g, float* m, float* v, int* state, long n, const CnfAdamOpts& o, hipStream_t s) {
    const int blocks = (int)((n + kAdamThreads - 1) / kAdamThreads);
    hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(kAdamThreads), 0, s, p, g, m, v, state, n, o);
    return hipGetLastError();
}

}  // namespace cnf
