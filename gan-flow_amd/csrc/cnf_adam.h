// cnf_adam.h -- the arithmetic of one torch.optim.Adam update, shared by adam_kernel (cnf_train.hip) and the kernel
// that sums the adjoint's gradient slabs and applies the update in one pass (adam_slab_kernel, cnf_split.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/cnf_hip.h"

namespace cnf {

// torch.optim.Adam (amsgrad=False, maximize=False; weight decay added to the gradient):
//   g += wd * p;  m += (g - m)(1 - b1);  v = b2 v + (1 - b2) g^2;
//   p -= lr / (1 - b1^t) * m / (sqrt(v) / sqrt(1 - b2^t) + eps),     t = step + 1.
struct AdamCoef {
    float b2, w1, w2, wd, eps, step_size, bc2_sqrt;
};

// `state` = {step, ticket}.  Thread 0 of the workgroup computes the bias corrections into corr[2] (LDS); call from all
// threads, it ends with a barrier.  Returns the step the update is for.
__device__ __forceinline__ int adam_begin(const int* state, const CnfAdamOpts& o, float (&corr)[2], AdamCoef& c) {
    const int step = __hip_atomic_load(&state[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (threadIdx.x == 0) {
        const double t = (double)(step + 1);
        const double bc1 = 1.0 - pow(o.beta1, t), bc2 = 1.0 - pow(o.beta2, t);
        corr[0] = (float)(o.lr / bc1);
        corr[1] = (float)sqrt(bc2);
    }
    __syncthreads();
    c.b2 = (float)o.beta2;
    c.w1 = (float)(1.0 - o.beta1);
    c.w2 = (float)(1.0 - o.beta2);
    c.wd = (float)o.weight_decay;
    c.eps = (float)o.eps;
    c.step_size = corr[0];
    c.bc2_sqrt = corr[1];
    return step;
}

__device__ __forceinline__ void adam_update(const AdamCoef& c, float g, float* __restrict__ p, float* __restrict__ m,
                                            float* __restrict__ v, long i) {
    const float pi = p[i];
    const float gi = fmaf(c.wd, pi, g);
    const float mi = fmaf(gi - m[i], c.w1, m[i]);
    const float vi = fmaf(c.b2, v[i], c.w2 * gi * gi);
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / c.bc2_sqrt + c.eps;
    p[i] = pi - c.step_size * (mi / denom);
}

// every workgroup has applied its part: the last one to finish (ticket) stores step + 1.  Call from all threads.
__device__ __forceinline__ void adam_end(int* state, int step) {
    __syncthreads();
    if (threadIdx.x == 0) {
        const int done = __hip_atomic_fetch_add(&state[1], 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (done == (int)gridDim.x - 1) {
            __hip_atomic_store(&state[1], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&state[0], step + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

}  // namespace cnf
