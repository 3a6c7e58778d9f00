// score_core.h -- order-exact device arithmetic of the four BLP relational score functions.
//
// Every kernel in this library scores a (candidate entity, query) pair through Scorer<>::score(),
// so the instruction sequence -- and therefore every rounding -- is the same whether the candidate
// sits in registers (rank kernel), comes from a CSR-listed row (filter kernel) or is the true
// entity (true-score kernel).  The sequence restates what the reference computes on the torch CPU
// backend (models.py:222-248; order pinned in oracle/blp_oracle.c):
//   TransE    -sum_d |(h_d + r_d) - t_d|          strict left-to-right f32 sum
//   DistMult  sum (h_d * r_d) * t_d               torch.sum order: 8 lanes x 4 accumulators
//   ComplEx   sum ((a + b) + c) - d over D/2      (same sum order)
//   SimplE    (sum (a + b)) / 2 over D/2          (same sum order)
// The file must be compiled with -ffp-contract=off: a fused multiply-add changes the last bit.
//
// A query is reduced once (prep kernel) to C "coefficients": whatever part of the expression does
// not involve the candidate (e.g. h + r for TransE tail queries) is hoisted, with the same
// rounding the reference applies, so the per-candidate work is 2-3 VALU ops per element.
#pragma once
#include <hip/hip_runtime.h>

#pragma clang fp contract(off)

namespace blp {

enum : int { TRANSE = 0, DISTMULT = 1, COMPLEX = 2, SIMPLE = 3 };
enum : int { HEAD = 0, TAIL = 1 };  // which position the candidates replace

// torch.sum(dim=-1) order for n = NT terms, NT % 32 == 0, NT < 512 (no cascade level is reached):
// A[k][l] += p[32c + 8k + l] over c;  V[l] = ((A0+A1)+A2)+A3;  result = ((V0+V1)+...)+V7.
// LITERAL_ZERO keeps the "0 + first term" additions of the reference (they only matter for the
// sign of an all-zero sum, i.e. for bitwise score output, never for > / >= comparisons).
template <int NT, bool LITERAL_ZERO, class Term>
__device__ __forceinline__ float torch_inner_sum(Term term) {
    static_assert(NT % 32 == 0 && NT >= 32 && NT < 512, "unsupported reduction width");
    float A[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) A[i] = LITERAL_ZERO ? 0.0f + term(i) : term(i);
#pragma unroll
    for (int c = 1; c < NT / 32; ++c) {
#pragma unroll
        for (int i = 0; i < 32; ++i) A[i] = A[i] + term(32 * c + i);
    }
    float V[8];
#pragma unroll
    for (int l = 0; l < 8; ++l) {
        float v = A[l] + A[8 + l];
        v = v + A[16 + l];
        V[l] = v + A[24 + l];
    }
    float s = LITERAL_ZERO ? 0.0f + V[0] : V[0];
#pragma unroll
    for (int l = 1; l < 8; ++l) s = s + V[l];
    return s;
}

template <int MODEL, int SIDE, int D>
struct Scorer;

// ---------------------------------------------------------------- TransE (models.py:222-223)
template <int D>
struct Scorer<TRANSE, TAIL, D> {  // candidates are tails: (h + r) - e, h + r hoisted
    static constexpr int C = D;
    __device__ static float coef(const float* f, const float* r, int i) { return f[i] + r[i]; }
    template <bool LZ>
    __device__ __forceinline__ static float score(const float (&e)[D], const float* __restrict__ c) {
        float acc = LZ ? 0.0f + fabsf(c[0] - e[0]) : fabsf(c[0] - e[0]);
#pragma unroll
        for (int d = 1; d < D; ++d) acc = acc + fabsf(c[d] - e[d]);
        return -acc;
    }
};
template <int D>
struct Scorer<TRANSE, HEAD, D> {  // candidates are heads: (e + r) - t, nothing hoistable
    static constexpr int C = 2 * D;
    __device__ static float coef(const float* f, const float* r, int i) { return i < D ? r[i] : f[i - D]; }
    template <bool LZ>
    __device__ __forceinline__ static float score(const float (&e)[D], const float* __restrict__ c) {
        float acc = 0.0f;
#pragma unroll
        for (int d = 0; d < D; ++d) {
            float x = e[d] + c[d];
            x = x - c[D + d];
            acc = (d == 0 && !LZ) ? fabsf(x) : acc + fabsf(x);
        }
        return -acc;
    }
};

// ---------------------------------------------------------------- DistMult (models.py:226-227)
template <int D>
struct Scorer<DISTMULT, TAIL, D> {  // (h * r) * e
    static constexpr int C = D;
    __device__ static float coef(const float* f, const float* r, int i) { return f[i] * r[i]; }
    template <bool LZ>
    __device__ __forceinline__ static float score(const float (&e)[D], const float* __restrict__ c) {
        return torch_inner_sum<D, LZ>([&](int i) { return c[i] * e[i]; });
    }
};
template <int D>
struct Scorer<DISTMULT, HEAD, D> {  // (e * r) * t
    static constexpr int C = 2 * D;
    __device__ static float coef(const float* f, const float* r, int i) { return i < D ? r[i] : f[i - D]; }
    template <bool LZ>
    __device__ __forceinline__ static float score(const float (&e)[D], const float* __restrict__ c) {
        return torch_inner_sum<D, LZ>([&](int i) {
            float x = e[i] * c[i];
            return x * c[D + i];
        });
    }
};

// ---------------------------------------------------------------- ComplEx (models.py:230-239)
// halves: re = [0, H), im = [H, D).  term = ((rr*hr)*tr + (rr*hi)*ti + (ri*hr)*ti) - (ri*hi)*tr
template <int D>
struct Scorer<COMPLEX, TAIL, D> {  // e = tail; the four r*h products are hoisted
    static constexpr int H = D / 2;
    static constexpr int C = 4 * H;
    __device__ static float coef(const float* f, const float* r, int i) {
        const int j = i % H, which = i / H;  // f = head
        switch (which) {
        case 0: return r[j] * f[j];            // rr * hr
        case 1: return r[j] * f[H + j];        // rr * hi
        case 2: return r[H + j] * f[j];        // ri * hr
        default: return r[H + j] * f[H + j];   // ri * hi
        }
    }
    template <bool LZ>
    __device__ __forceinline__ static float score(const float (&e)[D], const float* __restrict__ c) {
        return torch_inner_sum<H, LZ>([&](int j) {
            const float a = c[j] * e[j];
            const float b = c[H + j] * e[H + j];
            const float cc = c[2 * H + j] * e[H + j];
            const float d = c[3 * H + j] * e[j];
            float s = a + b;
            s = s + cc;
            return s - d;
        });
    }
};
template <int D>
struct Scorer<COMPLEX, HEAD, D> {  // e = head; coefficients are r then t, unchanged
    static constexpr int H = D / 2;
    static constexpr int C = 2 * D;
    __device__ static float coef(const float* f, const float* r, int i) { return i < D ? r[i] : f[i - D]; }
    template <bool LZ>
    __device__ __forceinline__ static float score(const float (&e)[D], const float* __restrict__ c) {
        return torch_inner_sum<H, LZ>([&](int j) {
            float a = c[j] * e[j];          a = a * c[D + j];          // (rr*hr)*tr
            float b = c[j] * e[H + j];      b = b * c[D + H + j];      // (rr*hi)*ti
            float cc = c[H + j] * e[j];     cc = cc * c[D + H + j];    // (ri*hr)*ti
            float d = c[H + j] * e[H + j];  d = d * c[D + j];          // (ri*hi)*tr
            float s = a + b;
            s = s + cc;
            return s - d;
        });
    }
};

// ---------------------------------------------------------------- SimplE (models.py:242-248)
// halves: head-role = [0, H), tail-role = [H, D).  term = (hh*ra)*tt + (th*rb)*ht ; sum / 2
template <int D>
struct Scorer<SIMPLE, TAIL, D> {  // e = tail = [th | tt]
    static constexpr int H = D / 2;
    static constexpr int C = 3 * H;
    __device__ static float coef(const float* f, const float* r, int i) {
        const int j = i % H, which = i / H;  // f = head = [hh | ht]
        switch (which) {
        case 0: return f[j] * r[j];   // hh * ra
        case 1: return r[H + j];      // rb
        default: return f[H + j];     // ht
        }
    }
    template <bool LZ>
    __device__ __forceinline__ static float score(const float (&e)[D], const float* __restrict__ c) {
        const float s = torch_inner_sum<H, LZ>([&](int j) {
            const float a = c[j] * e[H + j];
            float b = e[j] * c[H + j];
            b = b * c[2 * H + j];
            return a + b;
        });
        return s / 2.0f;
    }
};
template <int D>
struct Scorer<SIMPLE, HEAD, D> {  // e = head = [hh | ht]
    static constexpr int H = D / 2;
    static constexpr int C = 3 * H;
    __device__ static float coef(const float* f, const float* r, int i) {
        const int j = i % H, which = i / H;  // f = tail = [th | tt]
        switch (which) {
        case 0: return r[j];               // ra
        case 1: return f[H + j];           // tt
        default: return f[j] * r[H + j];   // th * rb
        }
    }
    template <bool LZ>
    __device__ __forceinline__ static float score(const float (&e)[D], const float* __restrict__ c) {
        const float s = torch_inner_sum<H, LZ>([&](int j) {
            float a = e[j] * c[j];
            a = a * c[H + j];
            const float b = c[2 * H + j] * e[H + j];
            return a + b;
        });
        return s / 2.0f;
    }
};

// Maximum coefficient count per query over both sides, used to size the workspace.
__host__ __device__ constexpr int max_coef(int D) { return 2 * D; }

}  // namespace blp
