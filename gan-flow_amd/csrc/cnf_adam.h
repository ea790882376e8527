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

// b^t for integer t >= 1 by repeated squaring in double: a few dozen dependent operations where the library pow() is a
// few hundred -- on ONE lane, which every other thread of the kernel waits for (measured: 3.4 of the Adam kernel's 4.4 us).
// Error: <= ~2 log2(t) ulp of a double, invisible after the cast of the corrections to float.
__device__ __forceinline__ double ipow(double b, int t) {
    double r = 1.0;
    while (t > 0) {
        if (t & 1) r *= b;
        b *= b;
        t >>= 1;
    }
    return r;
}

__device__ __forceinline__ void adam_corrections(int step, const CnfAdamOpts& o, float (&corr)[2]) {
    const double bc1 = 1.0 - ipow(o.beta1, step + 1), bc2 = 1.0 - ipow(o.beta2, step + 1);
    corr[0] = (float)(o.lr / bc1);
    corr[1] = (float)sqrt(bc2);
}

__device__ __forceinline__ AdamCoef adam_coef(const CnfAdamOpts& o, const float (&corr)[2]) {
    AdamCoef c;
    c.b2 = (float)o.beta2;
    c.w1 = (float)(1.0 - o.beta1);
    c.w2 = (float)(1.0 - o.beta2);
    c.wd = (float)o.weight_decay;
    c.eps = (float)o.eps;
    c.step_size = corr[0];
    c.bc2_sqrt = corr[1];
    return c;
}

// `state` = {step, ticket}.  Thread 0 of the workgroup computes the bias corrections into corr[2] (LDS); call from all
// threads, it ends with a barrier.  Returns the step the update is for.
__device__ __forceinline__ int adam_begin(const int* state, const CnfAdamOpts& o, float (&corr)[2], AdamCoef& c) {
    const int step = __hip_atomic_load(&state[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (threadIdx.x == 0) adam_corrections(step, o, corr);
    __syncthreads();
    c = adam_coef(o, corr);
    return step;
}

// the update of entry i from its loaded values (pi, mi, vi) and gradient g
__device__ __forceinline__ void adam_apply(const AdamCoef& c, float g, float pi, float mi_, float vi_, float* __restrict__ p,
                                           float* __restrict__ m, float* __restrict__ v, long i) {
    const float gi = fmaf(c.wd, pi, g);
    const float mi = fmaf(gi - mi_, c.w1, mi_);
    const float vi = fmaf(c.b2, vi_, c.w2 * gi * gi);
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / c.bc2_sqrt + c.eps;
    p[i] = pi - c.step_size * (mi / denom);
}

__device__ __forceinline__ void adam_update(const AdamCoef& c, float g, float* __restrict__ p, float* __restrict__ m,
                                            float* __restrict__ v, long i) {
    adam_apply(c, g, p[i], m[i], v[i], p, m, v, i);
}

// every workgroup has applied its part: the last one to finish (ticket) stores step + 1.  Call from all threads.
// Relaxed atomics are enough: a workgroup takes its ticket after the barrier that follows its use of the loaded step, so
// the one holding the last ticket overwrites the step only when every workgroup has consumed it; the next kernel sees
// the new value by stream order.  (An agent-scope release here would write the whole L2 back: ~1.5 us per launch.)
__device__ __forceinline__ void adam_end(int* state, int step) {
    __syncthreads();
    if (threadIdx.x == 0) {
        const int done = __hip_atomic_fetch_add(&state[1], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (done == (int)gridDim.x - 1) {
            __hip_atomic_store(&state[1], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&state[0], step + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

}  // namespace cnf
