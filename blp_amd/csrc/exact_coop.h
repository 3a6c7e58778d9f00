// exact_coop.h -- exact (reference-order) scores of individual (candidate, query) pairs computed by COOPERATING lanes.
// Used wherever single pairs are re-scored: the refinement of the pre-pass paths' undecided pairs (rank_sad.hip,
// rank_gemm.hip) and the filtered setting's removed candidates (rank_all.hip: filter_finalize_kernel).  A lane that
// reads its own candidate row and coefficient rows 16 bytes at a time makes every load instruction of the wave touch
// 64 different cache lines; here every load instruction reads whole 128-byte lines.  Same arithmetic, operation for
// operation, as Scorer<MODEL, SIDE, D>::score (score_core.h), so the keys are bit-identical to the true-entity keys.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rank_common.h"
#include "score_core.h"

#pragma clang fp contract(off)

namespace blp {

// ---------------------------------------------------------------- TransE: 64 pairs per wave, one lane per pair
// The L1 sum of a pair is one sequential chain of D additions: it cannot be spread over lanes without changing the
// rounding.  What is shared is the memory traffic: the wave fetches the 64 rows the way the exact kernel fetches a
// tile (rank_all.hip: load_tile) -- 8 rows x 128 B per load instruction -- 32 columns at a time, and transposes them
// through a wave-private LDS slab (64 x kRefStride floats) so that lane p ends up with row p; the rows are gathered
// through per-lane pointers instead of being consecutive.
constexpr int kRefStride = 36;  // dwords per slab row: conflict-free for the 16-byte reads of 16 consecutive lanes

__device__ __forceinline__ const float* shfl_ptr(const float* p, int src) {
    const unsigned long long v = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = __shfl((unsigned)v, src), hi = __shfl((unsigned)(v >> 32), src);
    return reinterpret_cast<const float*>(((unsigned long long)hi << 32) | lo);
}

// columns [32 s, 32 s + 32) of the 64 gathered rows: g[i] = start of row (8 i + sub_row) + sub_col, x[k] <- own row
__device__ __forceinline__ void gather_chunk(float (&x)[32], const float* const (&g)[8], int s, float* slab, int lane) {
    // the loads land in x[] in the coalesced layout first (plain scalars: they stay in registers across the fences)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float4 v = *reinterpret_cast<const float4*>(g[i] + 32 * s);
        x[4 * i] = v.x; x[4 * i + 1] = v.y; x[4 * i + 2] = v.z; x[4 * i + 3] = v.w;
    }
    float* wr = slab + (lane >> 3) * kRefStride + (lane & 7) * 4;
    wave_lds_sync();  // the previous chunk's reads are done before the slab is rewritten
#pragma unroll
    for (int i = 0; i < 8; ++i)
        *reinterpret_cast<float4*>(wr + 8 * i * kRefStride) = make_float4(x[4 * i], x[4 * i + 1], x[4 * i + 2], x[4 * i + 3]);
    wave_lds_sync();
    const float* rd = slab + lane * kRefStride;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float4 w = *reinterpret_cast<const float4*>(rd + 4 * j);
        x[4 * j] = w.x; x[4 * j + 1] = w.y; x[4 * j + 2] = w.z; x[4 * j + 3] = w.w;
    }
}


// key of lane's pair: pe = its candidate row; head-replacing query: pa = r, pb = t ((e + r) - t); tail-replacing:
// pa = h + r (hoisted), pb = any readable row ((h + r) - e).  Every lane must pass readable pointers.
template <int D>
__device__ __forceinline__ float transe_key_64(const float* pe, const float* pa, const float* pb, bool head, float* slab, int lane) {
    const int sub_row = lane >> 3, sub_col = (lane & 7) * 4;
    const float* ge[8];
    const float* ga[8];
    const float* gb[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        ge[k] = shfl_ptr(pe, 8 * k + sub_row) + sub_col;
        ga[k] = shfl_ptr(pa, 8 * k + sub_row) + sub_col;
        gb[k] = shfl_ptr(pb, 8 * k + sub_row) + sub_col;
    }
    float sum = 0.0f;
#pragma unroll
    for (int s = 0; s < D / 32; ++s) {
        float e[32], a[32], b[32];
        gather_chunk(e, ge, s, slab, lane);
        gather_chunk(a, ga, s, slab, lane);
        gather_chunk(b, gb, s, slab, lane);
#pragma unroll
        for (int k = 0; k < 32; ++k) {
            const float y = e[k] + a[k];  // head: (e + r) - t
            const float xh = y - b[k];
            const float xt = a[k] - e[k];  // tail: (h + r) - e
            sum = sum + fabsf(head ? xh : xt);
        }
    }
    return -sum;
}

// The same at a run-time width D (D % 4 == 0; the bag-of-words / DKRL widths 300, 768), straight from the entity and
// relation vectors: head-replacing query (e + r) - f with f = the tail, tail-replacing (f + r) - e with f = the head
// (models.py:222-223).  The last chunk of a width that is not a multiple of 32 loads only the columns that exist.
__device__ __forceinline__ void gather_chunk_rt(float (&x)[32], const float* const (&g)[8], int s, int cols, float* slab, int lane) {
    const bool mine = (lane & 7) * 4 < cols;  // cols is a multiple of 4: a lane's four columns exist together
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (mine) v = *reinterpret_cast<const float4*>(g[i] + 32 * s);
        x[4 * i] = v.x; x[4 * i + 1] = v.y; x[4 * i + 2] = v.z; x[4 * i + 3] = v.w;
    }
    float* wr = slab + (lane >> 3) * kRefStride + (lane & 7) * 4;
    wave_lds_sync();
#pragma unroll
    for (int i = 0; i < 8; ++i)
        *reinterpret_cast<float4*>(wr + 8 * i * kRefStride) = make_float4(x[4 * i], x[4 * i + 1], x[4 * i + 2], x[4 * i + 3]);
    wave_lds_sync();
    const float* rd = slab + lane * kRefStride;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float4 w = *reinterpret_cast<const float4*>(rd + 4 * j);
        x[4 * j] = w.x; x[4 * j + 1] = w.y; x[4 * j + 2] = w.z; x[4 * j + 3] = w.w;
    }
}

__device__ __forceinline__ float transe_key_64_rt(const float* pe, const float* pf, const float* pr, int D, bool head,
                                                  float* slab, int lane) {
    const int sub_row = lane >> 3, sub_col = (lane & 7) * 4;
    const float* ge[8];
    const float* gf[8];
    const float* gr[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        ge[k] = shfl_ptr(pe, 8 * k + sub_row) + sub_col;
        gf[k] = shfl_ptr(pf, 8 * k + sub_row) + sub_col;
        gr[k] = shfl_ptr(pr, 8 * k + sub_row) + sub_col;
    }
    float sum = 0.0f;
    for (int s = 0; s * 32 < D; ++s) {
        const int cols = D - 32 * s < 32 ? D - 32 * s : 32;
        float e[32], f[32], r[32];
        gather_chunk_rt(e, ge, s, cols, slab, lane);
        gather_chunk_rt(f, gf, s, cols, slab, lane);
        gather_chunk_rt(r, gr, s, cols, slab, lane);
#pragma unroll
        for (int k = 0; k < 32; ++k) {
            float x = (head ? e[k] : f[k]) + r[k];
            x = x - (head ? f[k] : e[k]);
            if (k < cols) sum = sum + fabsf(x);
        }
    }
    return -sum;
}

// ---------------------------------------------------------------- bilinear models: 32 lanes per pair
// Term i of the reference's sum for candidate row e and query coefficients c: the arithmetic of
// Scorer<MODEL, SIDE, D>::score (score_core.h) for one summand, addressed at run time.
template <int MODEL, int SIDE, int D>
__device__ __forceinline__ float score_term(const float* __restrict__ e, const float* __restrict__ c, int i) {
    constexpr int H = D / 2;
    if constexpr (MODEL == DISTMULT) {
        if constexpr (SIDE == TAIL) {
            return c[i] * e[i];
        } else {
            const float x = e[i] * c[i];
            return x * c[D + i];
        }
    } else if constexpr (MODEL == COMPLEX) {
        if constexpr (SIDE == TAIL) {
            const float a = c[i] * e[i];
            const float b = c[H + i] * e[H + i];
            const float cc = c[2 * H + i] * e[H + i];
            const float d = c[3 * H + i] * e[i];
            float s = a + b;
            s = s + cc;
            return s - d;
        } else {
            float a = c[i] * e[i];          a = a * c[D + i];
            float b = c[i] * e[H + i];      b = b * c[D + H + i];
            float cc = c[H + i] * e[i];     cc = cc * c[D + H + i];
            float d = c[H + i] * e[H + i];  d = d * c[D + i];
            float s = a + b;
            s = s + cc;
            return s - d;
        }
    } else {
        if constexpr (SIDE == TAIL) {
            const float a = c[i] * e[H + i];
            float b = e[i] * c[H + i];
            b = b * c[2 * H + i];
            return a + b;
        } else {
            float a = e[i] * c[i];
            a = a * c[H + i];
            const float b = c[2 * H + i] * e[H + i];
            return a + b;
        }
    }
}

// The exact score of one (candidate, query) pair by 32 cooperating lanes, in the reference's summation order
// (torch_inner_sum, score_core.h): lane j IS accumulator A[j] -- it adds the terms j, 32 + j, 64 + j, ... in that
// order -- then V[l] = ((A[l] + A[8 + l]) + A[16 + l]) + A[24 + l] and the eight V left to right.  Every load is 32
// consecutive floats (one 128-B line) instead of 64 lanes gathering 16 bytes each from 64 different rows.
// `sub` = lane & 31; both 32-lane halves of a wave work on their own pair.  Result valid in every lane of the half.
template <int MODEL, int SIDE, int D>
__device__ __forceinline__ float coop_score(const float* __restrict__ e, const float* __restrict__ c, int sub) {
    constexpr int NT = MODEL == DISTMULT ? D : D / 2;
    float a = score_term<MODEL, SIDE, D>(e, c, sub);
#pragma unroll
    for (int k = 1; k < NT / 32; ++k) a = a + score_term<MODEL, SIDE, D>(e, c, 32 * k + sub);
    float v = a + __shfl_down(a, 8, 32);
    v = v + __shfl_down(a, 16, 32);
    v = v + __shfl_down(a, 24, 32);
    float s = __shfl(v, 0, 32);
#pragma unroll
    for (int l = 1; l < 8; ++l) s = s + __shfl(v, l, 32);
    return MODEL == SIMPLE ? s / 2.0f : s;
}


}  // namespace blp
