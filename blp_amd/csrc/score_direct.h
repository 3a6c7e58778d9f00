// score_direct.h -- the four score functions evaluated straight from three operand rows in memory
// (runtime width D), plus their elementwise partial derivatives.  Used by the generic
// score_fn(heads, tails, rels) kernels and by the in-batch loss kernels, where nothing is reused
// across pairs and hoisting buys nothing.  Same operation order as score_core.h / the oracle
// (models.py:222-248), so the values are bit-identical to Scorer<>::score and to the reference.
#pragma once
#include <hip/hip_runtime.h>

#include "score_core.h"

#pragma clang fp contract(off)

namespace blp {

// torch.sum(dim=-1) order at ANY width n (ATen SumKernel.cpp: vectorized_inner_sum -> row_sum -> multi_row_sum;
// restated in oracle/blp_oracle.c: blp_oracle_torch_inner_sum): rows of 4 vectors x 8 lanes accumulate into
// acc[0]; every level_step rows the partial cascades one level up (four levels); then the leftover whole vectors
// go to the first accumulator vector, the four vectors fold into one, the scalar tail (n % 8 elements) is added
// FIRST and the eight lanes left to right.  Slow (the accumulators live in scratch memory) and rare: the widths
// the reference's scripts use for the bilinear models (128) take the register version below.
template <class Term>
__device__ __noinline__ float torch_inner_sum_any(int n, Term term) {
    const int vec_size = n / 8, size_ilp = vec_size / 4;
    int log2_rows = 1;  // ceil(log2(size_ilp)), 1 for size_ilp <= 2 (utils::CeilLog2)
    if (size_ilp > 2) {
        log2_rows = 0;
        for (int v = size_ilp - 1; v > 0; v >>= 1) ++log2_rows;
    }
    const int level_power = log2_rows / 4 > 4 ? log2_rows / 4 : 4;
    const int level_step = 1 << level_power, level_mask = level_step - 1;
    float acc[4][32];
    for (int j = 0; j < 4; ++j)
        for (int e = 0; e < 32; ++e) acc[j][e] = 0.0f;
    int i = 0;
    while (i + level_step <= size_ilp) {
        for (int j = 0; j < level_step; ++j, ++i)
            for (int e = 0; e < 32; ++e) acc[0][e] = acc[0][e] + term(i * 32 + e);
        for (int j = 1; j < 4; ++j) {
            for (int e = 0; e < 32; ++e) {
                acc[j][e] = acc[j][e] + acc[j - 1][e];
                acc[j - 1][e] = 0.0f;
            }
            if ((i & (level_mask << (j * level_power))) != 0) break;
        }
    }
    for (; i < size_ilp; ++i)
        for (int e = 0; e < 32; ++e) acc[0][e] = acc[0][e] + term(i * 32 + e);
    for (int j = 1; j < 4; ++j)
        for (int e = 0; e < 32; ++e) acc[0][e] = acc[0][e] + acc[j][e];
    for (int v = size_ilp * 4; v < vec_size; ++v)
        for (int l = 0; l < 8; ++l) acc[0][l] = acc[0][l] + term(v * 8 + l);
    for (int k = 1; k < 4; ++k)
        for (int l = 0; l < 8; ++l) acc[0][l] = acc[0][l] + acc[0][8 * k + l];
    float s = 0.0f;
    for (int k = vec_size * 8; k < n; ++k) s = s + term(k);
    for (int l = 0; l < 8; ++l) s = s + acc[0][l];
    return s;
}

// The same order for n % 32 == 0, 32 <= n < 512 (no cascade level is reached, no leftovers): 32 register
// accumulators; literal "0 + x" first adds.  Other widths go to torch_inner_sum_any.
template <class Term>
__device__ __forceinline__ float torch_inner_sum_rt(int n, Term term) {
    if (n % 32 != 0 || n >= 512 || n < 32) return torch_inner_sum_any(n, term);
    float A[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) A[i] = 0.0f + term(i);
    for (int c = 1; c < n / 32; ++c) {
#pragma unroll
        for (int i = 0; i < 32; ++i) A[i] = A[i] + term(32 * c + i);
    }
    float s = 0.0f;
#pragma unroll
    for (int l = 0; l < 8; ++l) {
        float v = A[l] + A[8 + l];
        v = v + A[16 + l];
        v = v + A[24 + l];
        s = s + v;
    }
    return s;
}

// A row of storage type T read as f32 (f16 / bf16 -> f32 is exact): lets the routines below take
// half-precision operands while every operation stays the reference's f32 one.
template <class T>
struct RowF {
    const T* p;
    __device__ __forceinline__ float operator[](int i) const { return (float)p[i]; }
};

template <int MODEL, class PH, class PT, class PR>
__device__ __forceinline__ float score_direct(PH h, PT t, PR r, int D) {
    if constexpr (MODEL == TRANSE) {
        float acc = 0.0f;
        for (int d = 0; d < D; ++d) {
            float x = h[d] + r[d];
            x = x - t[d];
            acc = acc + fabsf(x);
        }
        return -acc;
    } else if constexpr (MODEL == DISTMULT) {
        return torch_inner_sum_rt(D, [&](int i) {
            const float x = h[i] * r[i];
            return x * t[i];
        });
    } else if constexpr (MODEL == COMPLEX) {
        const int H = D / 2;
        return torch_inner_sum_rt(H, [&](int j) {
            float a = r[j] * h[j];          a = a * t[j];
            float b = r[j] * h[H + j];      b = b * t[H + j];
            float c = r[H + j] * h[j];      c = c * t[H + j];
            float d = r[H + j] * h[H + j];  d = d * t[j];
            float s = a + b;
            s = s + c;
            return s - d;
        });
    } else {
        const int H = D / 2;
        const float s = torch_inner_sum_rt(H, [&](int j) {
            float a = h[j] * r[j];      a = a * t[H + j];
            float b = t[j] * r[H + j];  b = b * h[H + j];
            return a + b;
        });
        return s / 2.0f;
    }
}

__device__ __forceinline__ float sign0(float x) { return x > 0.0f ? 1.0f : (x < 0.0f ? -1.0f : 0.0f); }

// d score / d operand[d] for operand in {0: heads, 1: tails, 2: rels}.
template <int MODEL, class PH, class PT, class PR>
__device__ __forceinline__ float dscore(int operand, PH h, PT t, PR r, int d, int D) {
    if constexpr (MODEL == TRANSE) {
        const float sg = sign0(h[d] + r[d] - t[d]);  // -|x|' = -sign(x); sign(0) = 0 like torch
        return operand == 1 ? sg : -sg;
    } else if constexpr (MODEL == DISTMULT) {
        return operand == 0 ? r[d] * t[d] : (operand == 1 ? h[d] * r[d] : h[d] * t[d]);
    } else if constexpr (MODEL == COMPLEX) {
        const int H = D / 2;
        const bool im = d >= H;
        const int j = im ? d - H : d;
        const float hr = h[j], hi = h[H + j], tr = t[j], ti = t[H + j], rr = r[j], ri = r[H + j];
        // s = rr*hr*tr + rr*hi*ti + ri*hr*ti - ri*hi*tr
        if (operand == 0) return im ? rr * ti - ri * tr : rr * tr + ri * ti;
        if (operand == 1) return im ? rr * hi + ri * hr : rr * hr - ri * hi;
        return im ? hr * ti - hi * tr : hr * tr + hi * ti;
    } else {
        const int H = D / 2;
        const bool second = d >= H;
        const int j = second ? d - H : d;
        // s = (hh*ra*tt + th*rb*ht) / 2 ; h = [hh|ht], t = [th|tt], r = [ra|rb]
        if (operand == 0) return 0.5f * (second ? t[j] * r[H + j] : r[j] * t[H + j]);
        if (operand == 1) return 0.5f * (second ? h[j] * r[j] : r[H + j] * h[H + j]);
        return 0.5f * (second ? t[j] * h[H + j] : h[j] * t[H + j]);
    }
}

}  // namespace blp
