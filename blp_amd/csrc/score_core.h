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
//
// All element loops are compile-time (static_for over integral constants), not "#pragma unroll":
// the candidate row must stay in registers and the DPP lane selectors must be immediates.
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>
#include <utility>

#pragma clang fp contract(off)

namespace blp {

enum : int { TRANSE = 0, DISTMULT = 1, COMPLEX = 2, SIMPLE = 3 };
enum : int { HEAD = 0, TAIL = 1 };  // which position the candidates replace

template <int I>
using ic = std::integral_constant<int, I>;

template <class F, int... Is>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, Is...>) {
    (f(ic<Is>{}), ...);  // comma fold: evaluated left to right
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    static_for_impl(f, std::make_integer_sequence<int, N>{});
}

// torch.sum(dim=-1) order for n = NT terms, NT % 32 == 0, NT < 512 (no cascade level is reached):
// A[k][l] += p[32c + 8k + l] over c;  V[l] = ((A0+A1)+A2)+A3;  result = ((V0+V1)+...)+V7.
// LITERAL_ZERO keeps the "0 + first term" additions of the reference (they only matter for the
// sign of an all-zero sum, i.e. for bitwise score output, never for > / >= comparisons).
// term(ic<i>) yields the i-th summand.
template <int NT, bool LITERAL_ZERO, class Term>
__device__ __forceinline__ float torch_inner_sum(Term term) {
    static_assert(NT % 32 == 0 && NT >= 32 && NT < 512, "unsupported reduction width");
    float A[32];
    static_for<32>([&](auto i) { A[i] = LITERAL_ZERO ? 0.0f + term(i) : term(i); });
    static_for<NT - 32>([&](auto k) {
        constexpr int i = decltype(k)::value + 32;
        A[i % 32] = A[i % 32] + term(ic<i>{});
    });
    float V[8];
    static_for<8>([&](auto l) {
        float v = A[l] + A[8 + l];
        v = v + A[16 + l];
        V[l] = v + A[24 + l];
    });
    float s = LITERAL_ZERO ? 0.0f + V[0] : V[0];
    static_for<7>([&](auto l) { s = s + V[l + 1]; });
    return s;
}

template <int MODEL, int SIDE, int D>
struct Scorer;

// In every Scorer: e is the candidate row (registers), c(ic<i>) the query's i-th coefficient.

// ---------------------------------------------------------------- TransE (models.py:222-223)
// The L1 sum is one dependent chain of D adds per (candidate, query); the differences feeding it are
// independent.  They are computed kPipe elements ahead of the add that consumes them, into a small
// ring of temporaries, so an add never waits on the subtract issued just before it (and no DPP
// instruction re-uses a register written two instructions earlier, which costs an s_nop).
constexpr int kPipe = 4;

template <int D>
struct Scorer<TRANSE, TAIL, D> {  // candidates are tails: (h + r) - e, h + r hoisted
    static constexpr int C = D;
    __device__ static float coef(const float* f, const float* r, int i) { return f[i] + r[i]; }
    template <bool LZ, class CF>
    __device__ __forceinline__ static float score(const float (&e)[D], const CF& c) {
        float x[kPipe];
        static_for<kPipe>([&](auto k) { x[k] = c(k) - e[k]; });
        float acc = 0.0f;
        static_for<D>([&](auto k) {
            constexpr int d = decltype(k)::value;
            const float cur = fabsf(x[d % kPipe]);
            if constexpr (d + kPipe < D) x[d % kPipe] = c(ic<d + kPipe>{}) - e[d + kPipe];
            acc = (d == 0 && !LZ) ? cur : acc + cur;
        });
        return -acc;
    }
};
template <int D>
struct Scorer<TRANSE, HEAD, D> {  // candidates are heads: (e + r) - t, nothing hoistable
    static constexpr int C = 2 * D;
    __device__ static float coef(const float* f, const float* r, int i) { return i < D ? r[i] : f[i - D]; }
    template <bool LZ, class CF>
    __device__ __forceinline__ static float score(const float (&e)[D], const CF& c) {
        float x[kPipe];
        static_for<kPipe>([&](auto k) {
            constexpr int d = decltype(k)::value;
            const float y = e[d] + c(ic<d>{});
            x[d] = y - c(ic<D + d>{});
        });
        float acc = 0.0f;
        static_for<D>([&](auto k) {
            constexpr int d = decltype(k)::value;
            const float cur = fabsf(x[d % kPipe]);
            if constexpr (d + kPipe < D) {
                const float y = e[d + kPipe] + c(ic<d + kPipe>{});
                x[d % kPipe] = y - c(ic<D + d + kPipe>{});
            }
            acc = (d == 0 && !LZ) ? cur : acc + cur;
        });
        return -acc;
    }
};

// ---------------------------------------------------------------- DistMult (models.py:226-227)
template <int D>
struct Scorer<DISTMULT, TAIL, D> {  // (h * r) * e
    static constexpr int C = D;
    __device__ static float coef(const float* f, const float* r, int i) { return f[i] * r[i]; }
    template <bool LZ, class CF>
    __device__ __forceinline__ static float score(const float (&e)[D], const CF& c) {
        return torch_inner_sum<D, LZ>([&](auto i) { return c(i) * e[i]; });
    }
};
template <int D>
struct Scorer<DISTMULT, HEAD, D> {  // (e * r) * t
    static constexpr int C = 2 * D;
    __device__ static float coef(const float* f, const float* r, int i) { return i < D ? r[i] : f[i - D]; }
    template <bool LZ, class CF>
    __device__ __forceinline__ static float score(const float (&e)[D], const CF& c) {
        return torch_inner_sum<D, LZ>([&](auto i) {
            constexpr int d = decltype(i)::value;
            float x = e[d] * c(ic<d>{});
            return x * c(ic<D + d>{});
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
    template <bool LZ, class CF>
    __device__ __forceinline__ static float score(const float (&e)[D], const CF& c) {
        return torch_inner_sum<H, LZ>([&](auto jj) {
            constexpr int j = decltype(jj)::value;
            const float a = c(ic<j>{}) * e[j];
            const float b = c(ic<H + j>{}) * e[H + j];
            const float cc = c(ic<2 * H + j>{}) * e[H + j];
            const float d = c(ic<3 * H + j>{}) * e[j];
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
    template <bool LZ, class CF>
    __device__ __forceinline__ static float score(const float (&e)[D], const CF& c) {
        return torch_inner_sum<H, LZ>([&](auto jj) {
            constexpr int j = decltype(jj)::value;
            // (every coefficient is read ONCE: an accessor that prefetches on the first touch of a chunk -- rank_stream.hip's
            //  SgprStreamCoef -- would otherwise see two first touches, request the next chunk twice and wait for it at once)
            const float rr = c(ic<j>{}), ri = c(ic<H + j>{}), tr = c(ic<D + j>{}), ti = c(ic<D + H + j>{});
            float a = rr * e[j];       a = a * tr;    // (rr*hr)*tr
            float b = rr * e[H + j];   b = b * ti;    // (rr*hi)*ti
            float cc = ri * e[j];      cc = cc * ti;  // (ri*hr)*ti
            float d = ri * e[H + j];   d = d * tr;    // (ri*hi)*tr
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
    template <bool LZ, class CF>
    __device__ __forceinline__ static float score(const float (&e)[D], const CF& c) {
        const float s = torch_inner_sum<H, LZ>([&](auto jj) {
            constexpr int j = decltype(jj)::value;
            const float a = c(ic<j>{}) * e[H + j];
            float b = e[j] * c(ic<H + j>{});
            b = b * c(ic<2 * H + j>{});
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
    template <bool LZ, class CF>
    __device__ __forceinline__ static float score(const float (&e)[D], const CF& c) {
        const float s = torch_inner_sum<H, LZ>([&](auto jj) {
            constexpr int j = decltype(jj)::value;
            float a = e[j] * c(ic<j>{});
            a = a * c(ic<H + j>{});
            const float b = c(ic<2 * H + j>{}) * e[H + j];
            return a + b;
        });
        return s / 2.0f;
    }
};

// ---------------------------------------------------------------- coefficient accessors
// PtrCoef: plain memory (true-score and filter kernels; per-lane pointer).
struct PtrCoef {
    const float* __restrict__ p;
    template <int I>
    __device__ __forceinline__ float operator()(ic<I>) const { return p[I]; }
};

// LazyCoef: coefficient i computed on use from the query's two vectors (f = the entity kept fixed, r = the relation),
// by the same Scorer<>::coef expression the prep kernel materialises -- so a key scored through it is bit-identical to
// one scored from the materialised row.  Used wherever single pairs are re-scored (true-entity keys, refinement,
// flag sweeps, filter): those paths then need no (Q, C) coefficient array at all.
template <int MODEL, int SIDE, int D>
struct LazyCoef {
    const float* __restrict__ f;
    const float* __restrict__ r;
    template <int I>
    __device__ __forceinline__ float operator()(ic<I>) const { return Scorer<MODEL, SIDE, D>::coef(f, r, I); }
};

template <int K>
__device__ __forceinline__ float quad_bcast(float x) {  // lane K of each quad -> the whole quad
    constexpr int ctrl = K | (K << 2) | (K << 4) | (K << 6);
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), ctrl, 0xf, 0xf, true));
}

// DppCoef: the rank kernel's path.  The query's coefficient row sits in LDS; lane l reads the 16 bytes
// at float4 index 4*(i/16) + (l & 3), so one conflict-free ds_read_b128 hands every quad 16 consecutive
// coefficients (a 4-periodic register set), and coefficient i is then the DPP quad_perm broadcast of
// component i%4 from quad-lane (i/4)%4 -- folded by the compiler into the consuming VALU instruction
// (v_sub_f32_dpp / v_mul_f32_dpp), so the broadcast costs no instruction and no SGPRs.  LDS returns
// in order, so the reads can be issued ahead of their use (scalar loads return out of order and can
// only be waited for all at once, which made the scalar-cache version of this kernel stall).
struct DppCoef {
    const float4* p;  // LDS: (const float4*)row + (lane & 3)
    template <int I>
    __device__ __forceinline__ float operator()(ic<I>) const {
        const float4 v = p[(I >> 4) * 4];
        constexpr int comp = I & 3, src = (I >> 2) & 3;
        const float x = comp == 0 ? v.x : comp == 1 ? v.y : comp == 2 ? v.z : v.w;
        return quad_bcast<src>(x);
    }
};

// Maximum coefficient count per query over both sides, used to size the workspace.
__host__ __device__ constexpr int max_coef(int D) { return 2 * D; }

}  // namespace blp
