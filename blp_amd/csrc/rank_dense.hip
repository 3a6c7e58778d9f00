// rank_dense.hip -- utils.get_metrics' counting (utils.py:103-105) and the filtered re-rank
// (train.py:159-167) on a DENSE (Q, N) score matrix that already sits in HBM.  This is the generic
// route: callers that still materialise `pred_ents`, and embedding widths the fused ranking kernels are
// not compiled for (e.g. the 300 / 768-wide bag-of-words encoders), where the matrix comes from
// blp_score_fwd (runtime width, order-exact).  One wave per query row; HBM-bound on the matrix.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "launch.h"

namespace blp {

__global__ __launch_bounds__(256) void rank_from_scores_kernel(const float* __restrict__ scores, int64_t Q, int64_t N,
                                                               int64_t ld, const int64_t* __restrict__ true_idx,
                                                               const float* __restrict__ true_score,
                                                               const int64_t* __restrict__ rowptr,
                                                               const int64_t* __restrict__ col,
                                                               int32_t* __restrict__ counts) {
    const int64_t q = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (q >= Q) return;
    const float* row = scores + q * ld;
    const float t = true_idx ? row[true_idx[q]] : true_score[q];
    unsigned gt = 0, ge = 0, fgt = 0, fge = 0;
    for (int64_t n = lane; n < N; n += 64) {
        const float s = row[n];
        gt += s > t;
        ge += s >= t;
    }
    if (rowptr) {
        for (int64_t k = rowptr[q] + lane; k < rowptr[q + 1]; k += 64) {
            const int64_t c = col[k];
            if ((uint64_t)c >= (uint64_t)N) continue;  // not a column of this matrix (another candidate shard's)
            const float s = row[c];
            fgt += s > t;
            fge += s >= t;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        gt += __shfl_down(gt, off);
        ge += __shfl_down(ge, off);
        fgt += __shfl_down(fgt, off);
        fge += __shfl_down(fge, off);
    }
    if (lane == 0) reinterpret_cast<int4*>(counts)[q] = make_int4((int)gt, (int)ge, (int)(gt - fgt), (int)(ge - fge));
}

hipError_t launch_rank_from_scores(const float* scores, int64_t Q, int64_t N, int64_t ld, const int64_t* true_idx,
                                   const float* true_score, const int64_t* rowptr, const int64_t* col,
                                   int32_t* counts, hipStream_t stream) {
    if (Q == 0) return hipSuccess;
    rank_from_scores_kernel<<<(unsigned)((Q + 3) / 4), 256, 0, stream>>>(scores, Q, N, ld, true_idx, true_score, rowptr,
                                                                          col, counts);
    return hipGetLastError();
}

}  // namespace blp
