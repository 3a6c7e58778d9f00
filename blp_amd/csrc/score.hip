// score.hip -- score_fn(heads, tails, rels) for any broadcast the reference uses (models.py:222-248;
// call sites train.py:146-147, models.py:57,67), plus its backward, plus the generic-width ranking
// fallback.  Output element (i0, i1) reads three operand rows addressed by two strides each
// (stride 0 = broadcast), one lane per output element, sequential over D in the reference's order.
// These kernels are the API-complete path; the fused rank / in-batch-loss kernels are the fast one.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "launch.h"
#include "score_direct.h"

#pragma clang fp contract(off)

namespace blp {

template <int MODEL>
__global__ __launch_bounds__(256) void score_fwd_kernel(int D, int64_t M0, int64_t M1, StridedRows h,
                                                        StridedRows t, StridedRows r,
                                                        float* __restrict__ out) {
    const int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (idx >= M0 * M1) return;
    const int64_t i0 = idx / M1, i1 = idx % M1;
    out[idx] = score_direct<MODEL>(h.base + i0 * h.s0 + i1 * h.s1, t.base + i0 * t.s0 + i1 * t.s1,
                                   r.base + i0 * r.s0 + i1 * r.s1, D);
}

template <int MODEL>
__global__ __launch_bounds__(256) void score_bwd_kernel(int D, int64_t M0, int64_t M1, StridedRows h,
                                                        StridedRows t, StridedRows r,
                                                        const float* __restrict__ grad_out,
                                                        float* __restrict__ gh, float* __restrict__ gt,
                                                        float* __restrict__ gr) {
    const int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;  // over (i0, i1, d)
    if (idx >= M0 * M1 * D) return;
    const int d = (int)(idx % D);
    const int64_t pair = idx / D, i0 = pair / M1, i1 = pair % M1;
    const float* hp = h.base + i0 * h.s0 + i1 * h.s1;
    const float* tp = t.base + i0 * t.s0 + i1 * t.s1;
    const float* rp = r.base + i0 * r.s0 + i1 * r.s1;
    const float g = grad_out[pair];
    if (gh) gh[idx] = g * dscore<MODEL>(0, hp, tp, rp, d, D);
    if (gt) gt[idx] = g * dscore<MODEL>(1, hp, tp, rp, d, D);
    if (gr) gr[idx] = g * dscore<MODEL>(2, hp, tp, rp, d, D);
}

template <class F>
static hipError_t dispatch_model(int model, F f) {
    switch (model) {
    case TRANSE:   return f(std::integral_constant<int, TRANSE>{});
    case DISTMULT: return f(std::integral_constant<int, DISTMULT>{});
    case COMPLEX:  return f(std::integral_constant<int, COMPLEX>{});
    case SIMPLE:   return f(std::integral_constant<int, SIMPLE>{});
    default:       return hipErrorInvalidValue;
    }
}

hipError_t launch_score_fwd(int model, int D, int64_t M0, int64_t M1, StridedRows h, StridedRows t,
                            StridedRows r, float* out, hipStream_t stream) {
    const int64_t total = M0 * M1;
    if (total == 0) return hipSuccess;
    return dispatch_model(model, [&](auto m) {
        score_fwd_kernel<decltype(m)::value><<<(unsigned)((total + 255) / 256), 256, 0, stream>>>(D, M0, M1, h, t, r, out);
        return hipGetLastError();
    });
}

hipError_t launch_score_bwd(int model, int D, int64_t M0, int64_t M1, StridedRows h, StridedRows t,
                            StridedRows r, const float* grad_out, float* grad_h, float* grad_t,
                            float* grad_r, hipStream_t stream) {
    const int64_t total = M0 * M1 * D;
    if (total == 0) return hipSuccess;
    return dispatch_model(model, [&](auto m) {
        score_bwd_kernel<decltype(m)::value><<<(unsigned)((total + 255) / 256), 256, 0, stream>>>(
            D, M0, M1, h, t, r, grad_out, grad_h, grad_t, grad_r);
        return hipGetLastError();
    });
}

}  // namespace blp
