// inbatch_loss.hip -- LinkPrediction.compute_loss (models.py:51-70) with margin / nll loss
// (models.py:251-258) and L2 regularisation (models.py:261-266), forward and backward, for
// in-batch negatives.  The reference materialises ent_embs.view(2B, D)[neg_idx] as a (B, K, 2, D)
// temporary and runs ~40 small torch kernels per step; here the gather is implicit (rows are read
// from the 2B x D matrix, which is 64 KB at B = 64 and lives in L2) and the whole thing is
//   fwd: pair scores (one lane per pair)  ->  one-block deterministic reduction
//   bwd: one wave per entity row / relation row, gathering its contributions in a fixed order
// so gradients are bit-reproducible run to run (no float atomics).
// Storage types: ent_embs / grad_ent in TE, rel_vecs / grad_rel in TR, each f32, f16 or bf16 (TR = TE or
// f32: under autocast the encoder output is half while nn.Embedding rows stay f32).  Half operands are
// widened exactly and every operation is the f32 one of the reference; gradients are rounded once on store.
// This path is launch/latency-bound (tens of KB of data): no roofline applies; see DESIGN.md.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "launch.h"
#include "score_direct.h"

#pragma clang fp contract(off)

namespace blp {

enum : int { LOSS_MARGIN = 0, LOSS_NLL = 1 };

// ---------------------------------------------------------------- forward
template <int MODEL, class TE, class TR>
__global__ __launch_bounds__(64) void inbatch_scores_kernel(const TE* __restrict__ ent,
                                                           const TR* __restrict__ rel,
                                                           const int64_t* __restrict__ neg_idx, int B, int K,
                                                           int D, float* __restrict__ pos,
                                                           float* __restrict__ neg) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;  // over B * (K + 1)
    if (idx >= B * (K + 1)) return;
    const int b = idx / (K + 1), k = idx % (K + 1);
    const RowF<TR> r{rel + (size_t)b * D};
    if (k == K) {  // positive pair: models.py:56-57
        pos[b] = score_direct<MODEL>(RowF<TE>{ent + (size_t)(2 * b) * D}, RowF<TE>{ent + (size_t)(2 * b + 1) * D}, r, D);
    } else {       // negative pair: models.py:65-67
        const int64_t ih = neg_idx[((size_t)b * K + k) * 2], it = neg_idx[((size_t)b * K + k) * 2 + 1];
        neg[(size_t)b * K + k] = score_direct<MODEL>(RowF<TE>{ent + ih * D}, RowF<TE>{ent + it * D}, r, D);
    }
}

__device__ __forceinline__ float softplus_torch(float x) {  // F.softplus, beta = 1, threshold = 20
    return x > 20.0f ? x : log1pf(expf(x));
}

__device__ __forceinline__ double block_sum(double v, double* sh) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    if (lane == 0) sh[wave] = v;
    __syncthreads();
    double total = 0.0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) total += sh[w];  // fixed order
    return total;
}

template <class TE, class TR>
__global__ __launch_bounds__(1024) void inbatch_reduce_kernel(int loss, const TE* __restrict__ ent,
                                                             const TR* __restrict__ rel,
                                                             const float* __restrict__ pos,
                                                             const float* __restrict__ neg, int B, int K, int D,
                                                             float regularizer, float* __restrict__ out) {
    __shared__ double sh[16];
    const int tid = threadIdx.x, nt = blockDim.x;
    double model_loss;
    if (loss == LOSS_MARGIN) {  // models.py:251-254
        double s = 0.0;
        for (int i = tid; i < B * K; i += nt) {
            float l = 1.0f - pos[i / K];
            l = l + neg[i];
            s += l < 0.0f ? 0.0f : l;
        }
        model_loss = block_sum(s, sh) / ((double)B * K);
    } else {                    // models.py:257-258
        double sp = 0.0, sn = 0.0;
        for (int i = tid; i < B; i += nt) sp += softplus_torch(-pos[i]);
        for (int i = tid; i < B * K; i += nt) sn += softplus_torch(neg[i]);
        const double a = block_sum(sp, sh) / B;
        const double c = block_sum(sn, sh) / ((double)B * K);
        model_loss = (a + c) / 2.0;
    }
    double reg = 0.0;
    if (regularizer > 0.0f) {  // models.py:59-60, 261-266: mean(h^2) + mean(t^2) + mean(r^2) over positives
        double s = 0.0;
        for (int i = tid; i < 2 * B * D; i += nt) s += (double)(float)ent[i] * (double)(float)ent[i];
        for (int i = tid; i < B * D; i += nt) s += (double)(float)rel[i] * (double)(float)rel[i];
        reg = (double)regularizer * block_sum(s, sh) / ((double)B * D) / 3.0;
    }
    if (tid == 0) out[0] = (float)(model_loss + reg);
}

// ---------------------------------------------------------------- backward
// d loss / d neg[b, k] and d loss / d pos[b], recomputed from the saved scores.
__device__ __forceinline__ float dloss_dneg(int loss, float pos_b, float neg_bk, int B, int K) {
    if (loss == LOSS_MARGIN) {
        float l = 1.0f - pos_b;
        l = l + neg_bk;
        return l < 0.0f ? 0.0f : 1.0f / ((float)B * K);  // masked in place: gradient passes at l == 0
    }
    const float z = expf(neg_bk);
    return (neg_bk > 20.0f ? 1.0f : z / (z + 1.0f)) / (2.0f * B * K);
}

__device__ __forceinline__ float dloss_dpos(int loss, float pos_b, const float* __restrict__ neg_b, int B, int K) {
    if (loss == LOSS_MARGIN) {
        int cnt = 0;
        for (int k = 0; k < K; ++k) {
            float l = 1.0f - pos_b;
            l = l + neg_b[k];
            cnt += !(l < 0.0f);
        }
        return -(float)cnt / ((float)B * K);
    }
    const float x = -pos_b, z = expf(x);
    return -(x > 20.0f ? 1.0f : z / (z + 1.0f)) / (2.0f * B);
}

// One wave per row j of ent_embs.view(2B, D): positive-pair term, then every negative pair that
// references row j, found by scanning neg_idx in order (8 KB of int64 per K = 64 row -- L2 hits).
template <int MODEL, class TE, class TR>
__global__ __launch_bounds__(64) void inbatch_grad_ent_kernel(
    int loss, const TE* __restrict__ ent, const TR* __restrict__ rel, const int64_t* __restrict__ neg_idx,
    int B, int K, int D, float regularizer, const float* __restrict__ grad_loss,
    const float* __restrict__ pos, const float* __restrict__ neg, TE* __restrict__ grad_ent) {
    const int j = blockIdx.x, lane = threadIdx.x;
    const int b = j >> 1, slot = j & 1;
    const float gl = grad_loss[0];
    constexpr int MAXR = 12;  // d handled by this lane per sweep: d0 + lane, d0 + lane + 64, ... (768 per sweep)
    for (int d0 = 0; d0 < D; d0 += 64 * MAXR) {  // one sweep up to D = 768; wider rows repeat the scan
    float g[MAXR];
#pragma unroll
    for (int i = 0; i < MAXR; ++i) g[i] = 0.0f;

    {   // positive pair (2b, 2b+1, rel b)
        const float gp = dloss_dpos(loss, pos[b], neg + (size_t)b * K, B, K);
        const RowF<TE> h{ent + (size_t)(2 * b) * D}, t{ent + (size_t)(2 * b + 1) * D};
        const RowF<TR> r{rel + (size_t)b * D};
#pragma unroll
        for (int i = 0; i < MAXR; ++i) {
            const int d = d0 + lane + 64 * i;
            if (d < D) g[i] += gp * dscore<MODEL>(slot, h, t, r, d, D);
        }
    }
    const int total = 2 * B * K;
    for (int base = 0; base < total; base += 64) {
        const int i = base + lane;
        const bool hit = i < total && neg_idx[i] == j;
        unsigned long long mask = __ballot(hit);
        while (mask) {
            const int bit = __builtin_ctzll(mask);
            mask &= mask - 1;
            const int e = base + bit, pair = e >> 1, s = e & 1, pb = pair / K;
            const float gn = dloss_dneg(loss, pos[pb], neg[pair], B, K);
            if (gn != 0.0f) {
                const RowF<TE> h{ent + neg_idx[2 * (size_t)pair] * D}, t{ent + neg_idx[2 * (size_t)pair + 1] * D};
                const RowF<TR> r{rel + (size_t)pb * D};
#pragma unroll
                for (int q = 0; q < MAXR; ++q) {
                    const int d = d0 + lane + 64 * q;
                    if (d < D) g[q] += gn * dscore<MODEL>(s, h, t, r, d, D);
                }
            }
        }
    }
    const float reg_scale = regularizer > 0.0f ? regularizer * 2.0f / (3.0f * B * D) : 0.0f;
#pragma unroll
    for (int i = 0; i < MAXR; ++i) {
        const int d = d0 + lane + 64 * i;
        if (d < D) grad_ent[(size_t)j * D + d] = (TE)(gl * (g[i] + reg_scale * (float)ent[(size_t)j * D + d]));
    }
    }
}

// One wave per relation row b: positive pair + its K negatives (rels broadcast over K, models.py:67).
template <int MODEL, class TE, class TR>
__global__ __launch_bounds__(64) void inbatch_grad_rel_kernel(
    int loss, const TE* __restrict__ ent, const TR* __restrict__ rel, const int64_t* __restrict__ neg_idx,
    int B, int K, int D, float regularizer, const float* __restrict__ grad_loss,
    const float* __restrict__ pos, const float* __restrict__ neg, TR* __restrict__ grad_rel) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const float gl = grad_loss[0];
    const RowF<TR> r{rel + (size_t)b * D};
    const float gp = dloss_dpos(loss, pos[b], neg + (size_t)b * K, B, K);
    const float reg_scale = regularizer > 0.0f ? regularizer * 2.0f / (3.0f * B * D) : 0.0f;
    for (int d = lane; d < D; d += 64) {
        float g = gp * dscore<MODEL>(2, RowF<TE>{ent + (size_t)(2 * b) * D}, RowF<TE>{ent + (size_t)(2 * b + 1) * D}, r, d, D);
        for (int k = 0; k < K; ++k) {
            const size_t pair = (size_t)b * K + k;
            const float gn = dloss_dneg(loss, pos[b], neg[pair], B, K);
            if (gn != 0.0f)
                g += gn * dscore<MODEL>(2, RowF<TE>{ent + neg_idx[2 * pair] * D}, RowF<TE>{ent + neg_idx[2 * pair + 1] * D}, r, d, D);
        }
        grad_rel[(size_t)b * D + d] = (TR)(gl * (g + reg_scale * r[d]));
    }
}

// ---------------------------------------------------------------- launchers
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

template <class T> struct TypeTag { using type = T; };

// (ent dtype, rel dtype) with 0 = f32, 1 = f16, 2 = bf16; rel is either ent's type or f32
template <class F>
static hipError_t dispatch_types(int ent_dtype, int rel_dtype, F f) {
    if (ent_dtype == 0 && rel_dtype == 0) return f(TypeTag<float>{}, TypeTag<float>{});
    if (ent_dtype == 1 && rel_dtype == 1) return f(TypeTag<_Float16>{}, TypeTag<_Float16>{});
    if (ent_dtype == 1 && rel_dtype == 0) return f(TypeTag<_Float16>{}, TypeTag<float>{});
    if (ent_dtype == 2 && rel_dtype == 2) return f(TypeTag<__bf16>{}, TypeTag<__bf16>{});
    if (ent_dtype == 2 && rel_dtype == 0) return f(TypeTag<__bf16>{}, TypeTag<float>{});
    return hipErrorInvalidValue;
}

hipError_t launch_inbatch_loss_fwd(int model, int loss, int ent_dtype, int rel_dtype, const void* ent, const void* rel,
                                   const int64_t* neg_idx, int B, int K, int D, float regularizer,
                                   float* out_loss, float* save_pos, float* save_neg, hipStream_t stream) {
    return dispatch_model(model, [&](auto m) {
        return dispatch_types(ent_dtype, rel_dtype, [&](auto te, auto tr) {
            using TE = typename decltype(te)::type;
            using TR = typename decltype(tr)::type;
            const TE* e = static_cast<const TE*>(ent);
            const TR* r = static_cast<const TR*>(rel);
            const int pairs = B * (K + 1);
            inbatch_scores_kernel<decltype(m)::value, TE, TR><<<(pairs + 63) / 64, 64, 0, stream>>>(e, r, neg_idx, B, K, D, save_pos, save_neg);
            inbatch_reduce_kernel<TE, TR><<<1, 1024, 0, stream>>>(loss, e, r, save_pos, save_neg, B, K, D, regularizer, out_loss);
            return hipGetLastError();
        });
    });
}

hipError_t launch_inbatch_loss_bwd(int model, int loss, int ent_dtype, int rel_dtype, const void* ent, const void* rel,
                                   const int64_t* neg_idx, int B, int K, int D, float regularizer,
                                   const float* grad_loss, const float* save_pos, const float* save_neg,
                                   void* grad_ent, void* grad_rel, hipStream_t stream) {
    return dispatch_model(model, [&](auto m) {
        return dispatch_types(ent_dtype, rel_dtype, [&](auto te, auto tr) {
            using TE = typename decltype(te)::type;
            using TR = typename decltype(tr)::type;
            const TE* e = static_cast<const TE*>(ent);
            const TR* r = static_cast<const TR*>(rel);
            inbatch_grad_ent_kernel<decltype(m)::value, TE, TR><<<2 * B, 64, 0, stream>>>(
                loss, e, r, neg_idx, B, K, D, regularizer, grad_loss, save_pos, save_neg, static_cast<TE*>(grad_ent));
            inbatch_grad_rel_kernel<decltype(m)::value, TE, TR><<<B, 64, 0, stream>>>(
                loss, e, r, neg_idx, B, K, D, regularizer, grad_loss, save_pos, save_neg, static_cast<TR*>(grad_rel));
            return hipGetLastError();
        });
    });
}

}  // namespace blp
