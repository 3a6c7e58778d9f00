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
#include "table_elem.h"

#pragma clang fp contract(off)

namespace blp {

// ---------------------------------------------------------------- TransE: 64 pairs per wave, one lane per pair
// The L1 sum of a pair is one sequential chain of D additions: it cannot be spread over lanes without changing the
// rounding.  What is shared is the memory traffic: the wave fetches the 64 rows the way the exact kernel fetches a
// tile (rank_all.hip: load_tile) -- 8 rows x 128 B per load instruction -- 32 columns at a time, and transposes them
// through a wave-private LDS slab (64 x kRefStride floats) so that lane p ends up with row p; the rows are gathered
// through per-lane pointers instead of being consecutive.
constexpr int kRefStride = 36;  // dwords per slab row: conflict-free for the 16-byte reads of 16 consecutive lanes

// A candidate row in its storage type TE (float, _Float16, __bf16: table_elem.h): element k widened to f32, exactly.
template <class TE>
struct RowOf {
    const TE* p;
    __device__ __forceinline__ float operator[](int k) const { return (float)p[k]; }
};

template <class T>
__device__ __forceinline__ const T* shfl_ptr(const T* p, int src) {
    const unsigned long long v = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = __shfl((unsigned)v, src), hi = __shfl((unsigned)(v >> 32), src);
    return reinterpret_cast<const T*>(((unsigned long long)hi << 32) | lo);
}

// columns [32 s, 32 s + 32) of the 64 gathered rows: g[i] = start of row (8 i + sub_row) + sub_col, x[k] <- own row
template <class T>
__device__ __forceinline__ void gather_chunk(float (&x)[32], const T* const (&g)[8], int s, float* slab, int lane) {
    // the loads land in x[] in the coalesced layout first (plain scalars: they stay in registers across the fences)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float4 v = load4<T>(g[i] + 32 * s);  // (a 16-bit row: widened here, exactly)
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


// key of lane's pair straight from the vectors: pe = its candidate row, pf = the query's fixed entity, pr = its
// relation; head-replacing query (e + r) - f, tail-replacing (f + r) - e (models.py:222-223; Scorer<TRANSE, *, D>
// computes the same operations, with f + r hoisted).  Every lane must pass readable pointers.
template <int D, class TE = float>
__device__ __forceinline__ float transe_key_64(const TE* pe, const float* pf, const float* pr, bool head, float* slab, int lane) {
    const int sub_row = lane >> 3, sub_col = (lane & 7) * 4;
    const TE* ge[8];
    const float* gf[8];
    const float* gr[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        ge[k] = shfl_ptr(pe, 8 * k + sub_row) + sub_col;
        gf[k] = shfl_ptr(pf, 8 * k + sub_row) + sub_col;
        gr[k] = shfl_ptr(pr, 8 * k + sub_row) + sub_col;
    }
    float sum = 0.0f;
#pragma unroll
    for (int s = 0; s < D / 32; ++s) {
        float e[32], f[32], r[32];
        gather_chunk(e, ge, s, slab, lane);
        gather_chunk(f, gf, s, slab, lane);
        gather_chunk(r, gr, s, slab, lane);
#pragma unroll
        for (int k = 0; k < 32; ++k) {
            float x = (head ? e[k] : f[k]) + r[k];
            x = x - (head ? f[k] : e[k]);
            sum = sum + fabsf(x);
        }
    }
    return -sum;
}

// The same at a run-time width D (D % 4 == 0; the bag-of-words / DKRL widths 300, 768), straight from the entity and
// relation vectors: head-replacing query (e + r) - f with f = the tail, tail-replacing (f + r) - e with f = the head
// (models.py:222-223).  The last chunk of a width that is not a multiple of 32 loads only the columns that exist.
__device__ __forceinline__ void gather_issue_rt(float (&x)[32], const float* const (&g)[8], int s, int cols, int lane) {
    const bool mine = (lane & 7) * 4 < cols;  // cols is a multiple of 4: a lane's four columns exist together
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (mine) v = *reinterpret_cast<const float4*>(g[i] + 32 * s);
        x[4 * i] = v.x; x[4 * i + 1] = v.y; x[4 * i + 2] = v.z; x[4 * i + 3] = v.w;
    }
}
__device__ __forceinline__ void gather_transpose(float (&x)[32], float* slab, int lane) {
    float* wr = slab + (lane >> 3) * kRefStride + (lane & 7) * 4;
    wave_lds_sync();  // the previous trip's reads are done before the slab is rewritten
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
        // the 24 row loads of the chunk go out together, then the three trips through the slab: one memory round trip
        // per chunk instead of three (each gather's loads otherwise wait behind the previous one's LDS fences); at
        // D = 768 that is worth 7 %, at D <= 256 the extra live registers cost more occupancy than the round trips
        float e[32], f[32], r[32];
        gather_issue_rt(e, ge, s, cols, lane);
        gather_issue_rt(f, gf, s, cols, lane);
        gather_issue_rt(r, gr, s, cols, lane);
        gather_transpose(e, slab, lane);
        gather_transpose(f, slab, lane);
        gather_transpose(r, slab, lane);
#pragma unroll
        for (int k = 0; k < 32; ++k) {
            float x = (head ? e[k] : f[k]) + r[k];
            x = x - (head ? f[k] : e[k]);
            if (k < cols) sum = sum + fabsf(x);
        }
    }
    return -sum;
}

// 64 candidates against ONE query at a run-time width: what a refinement that has its pairs grouped by query calls
// (rank_sad_wide.hip).  The query side is staged ONCE per wave in LDS by the caller (stage_query_rt: qa = h + r for a
// tail-replacing query; qa = r, qb = t for a head-replacing one) and read back as broadcasts -- through the scalar cache the
// two 3-KB vectors of a new query every task were a cold miss per 32-column chunk (5.7 ms for the D = 768 block's pairs).  The
// candidate rows go through a ring of four 32-column chunks, three requested ahead of the one being summed.  Same operations
// in the same order as transe_key_64_rt -- models.py:222-223 -- so the keys are the true-entity keys' bit for bit.
__device__ __forceinline__ void stage_query_rt(const float* __restrict__ f, const float* __restrict__ r, int D, bool head,
                                               float* qa, float* qb, int lane) {
    for (int c = 4 * lane; c < D; c += 256) {
        const float4 fv = *reinterpret_cast<const float4*>(f + c);
        const float4 rv = *reinterpret_cast<const float4*>(r + c);
        if (head) {
            *reinterpret_cast<float4*>(qa + c) = rv;
            *reinterpret_cast<float4*>(qb + c) = fv;
        } else {  // (h + r), rounded once, exactly as the reference's first operation
            *reinterpret_cast<float4*>(qa + c) = make_float4(fv.x + rv.x, fv.y + rv.y, fv.z + rv.z, fv.w + rv.w);
        }
    }
}

__device__ __forceinline__ float transe_key_64_one_query_rt(const float* pe, const float* qa, const float* qb, int D, bool head,
                                                            float* slab, int lane) {
    const int sub_row = lane >> 3, sub_col = (lane & 7) * 4;
    const float* ge[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) ge[k] = shfl_ptr(pe, 8 * k + sub_row) + sub_col;
    float sum = 0.0f;
    const int n_chunks = (D + 31) / 32;  // the last one may be partial (D = 300: 12 columns)
    float ring[4][32];
    auto issue = [&](float (&x)[32], int s) {
        const int rest = D - 32 * s;
        gather_issue_rt(x, ge, s, rest < 32 ? rest : 32, lane);
    };
    auto consume = [&](float (&x)[32], int s) {
        gather_transpose(x, slab, lane);
        const float* a = qa + 32 * s;
        const float* b = qb + 32 * s;
        const int cols = D - 32 * s;
#pragma unroll
        for (int k4 = 0; k4 < 8; ++k4) {
            if (4 * k4 < cols) {  // wave-uniform (D % 4 == 0: four columns exist together)
                const float4 av = *reinterpret_cast<const float4*>(a + 4 * k4);  // every lane the same address: a broadcast
                const float as[4] = {av.x, av.y, av.z, av.w};
                if (head) {
                    const float4 bv = *reinterpret_cast<const float4*>(b + 4 * k4);
                    const float bs[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        float v = x[4 * k4 + i] + as[i];  // (e + r) - t
                        v = v - bs[i];
                        sum = sum + fabsf(v);
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i) sum = sum + fabsf(as[i] - x[4 * k4 + i]);  // (h + r) - e
                }
            }
        }
    };
    if (0 < n_chunks) issue(ring[0], 0);
    if (1 < n_chunks) issue(ring[1], 1);
    if (2 < n_chunks) issue(ring[2], 2);
    for (int s0 = 0; s0 < n_chunks; s0 += 4) {
        static_for<4>([&](auto ii) {
            constexpr int i = decltype(ii)::value;
            const int s = s0 + i;
            if (s < n_chunks) {
                if (s + 3 < n_chunks) issue(ring[(i + 3) % 4], s + 3);
                consume(ring[i], s);
            }
        });
    }
    return -sum;
}

// ---------------------------------------------------------------- bilinear models: 32 lanes per pair
// Term i of the reference's sum for candidate row e and the query's vectors f (the entity kept fixed) and r (the
// relation): the arithmetic of Scorer<MODEL, SIDE, D>::score (score_core.h) for one summand with its coefficients
// (Scorer<>::coef) computed in place, addressed at run time.
template <int MODEL, int SIDE, int D, class TE>
__device__ __forceinline__ float score_term(const TE* __restrict__ ep, const float* __restrict__ f,
                                            const float* __restrict__ r, int i) {
    constexpr int H = D / 2;
    (void)H;
    const RowOf<TE> e{ep};  // e[k]: element k of the candidate row, widened (exact) from its storage type
    if constexpr (MODEL == DISTMULT) {
        if constexpr (SIDE == TAIL) {  // (h * r) * e
            const float c = f[i] * r[i];
            return c * e[i];
        } else {                       // (e * r) * t
            const float x = e[i] * r[i];
            return x * f[i];
        }
    } else if constexpr (MODEL == COMPLEX) {
        if constexpr (SIDE == TAIL) {  // f = head: the four r * h products are the coefficients
            const float c0 = r[i] * f[i], c1 = r[i] * f[H + i], c2 = r[H + i] * f[i], c3 = r[H + i] * f[H + i];
            const float a = c0 * e[i];
            const float b = c1 * e[H + i];
            const float cc = c2 * e[H + i];
            const float d = c3 * e[i];
            float s = a + b;
            s = s + cc;
            return s - d;
        } else {                       // f = tail
            float a = r[i] * e[i];          a = a * f[i];
            float b = r[i] * e[H + i];      b = b * f[H + i];
            float cc = r[H + i] * e[i];     cc = cc * f[H + i];
            float d = r[H + i] * e[H + i];  d = d * f[i];
            float s = a + b;
            s = s + cc;
            return s - d;
        }
    } else {
        if constexpr (SIDE == TAIL) {  // f = head = [hh | ht], e = tail = [th | tt]
            const float c0 = f[i] * r[i];
            const float a = c0 * e[H + i];
            float b = e[i] * r[H + i];
            b = b * f[H + i];
            return a + b;
        } else {                       // f = tail = [th | tt], e = head = [hh | ht]
            float a = e[i] * r[i];
            a = a * f[H + i];
            const float c2 = f[i] * r[H + i];
            const float b = c2 * e[H + i];
            return a + b;
        }
    }
}

// The exact score of one (candidate, query) pair by 32 cooperating lanes, in the reference's summation order
// (torch_inner_sum, score_core.h): lane j IS accumulator A[j] -- it adds the terms j, 32 + j, 64 + j, ... in that
// order -- then V[l] = ((A[l] + A[8 + l]) + A[16 + l]) + A[24 + l] and the eight V left to right.  Every load is 32
// consecutive floats (one 128-B line) instead of 64 lanes gathering 16 bytes each from 64 different rows.
// `sub` = lane & 31; both 32-lane halves of a wave work on their own pair.  Result valid in every lane of the half.
template <int MODEL, int SIDE, int D, class TE = float>
__device__ __forceinline__ float coop_score(const TE* __restrict__ e, const float* __restrict__ f,
                                            const float* __restrict__ r, int sub) {
    constexpr int NT = MODEL == DISTMULT ? D : D / 2;
    float a = score_term<MODEL, SIDE, D>(e, f, r, sub);
#pragma unroll
    for (int k = 1; k < NT / 32; ++k) a = a + score_term<MODEL, SIDE, D>(e, f, r, 32 * k + sub);
    float v = a + __shfl_down(a, 8, 32);
    v = v + __shfl_down(a, 16, 32);
    v = v + __shfl_down(a, 24, 32);
    float s = __shfl(v, 0, 32);
#pragma unroll
    for (int l = 1; l < 8; ++l) s = s + __shfl(v, l, 32);
    return MODEL == SIMPLE ? s / 2.0f : s;
}


// One lane's true-entity key (and its query's zeroed accumulator): the row in registers, all of its loads in flight at
// once -- what a launch with few queries wants (rank_all.hip: true_key_lane_kernel; rank_gemm.hip: the bilinear prelude).
template <int MODEL, int D>
__device__ __forceinline__ void true_key_lane(const QRows& q_true, const QRows& q_fixed, const QRows& q_rel,
                                              int64_t q, int64_t q_head, float* __restrict__ key_true,
                                              unsigned long long* __restrict__ acc) {
    acc[q] = 0;
    float e[D];
    load_row<D>(e, q_true.row(q));
    const float* f = q_fixed.row(q);
    const float* r = q_rel.row(q);
    key_true[q] = q < q_head ? Scorer<MODEL, HEAD, D>::template score<false>(e, LazyCoef<MODEL, HEAD, D>{f, r})
                             : Scorer<MODEL, TAIL, D>::template score<false>(e, LazyCoef<MODEL, TAIL, D>{f, r});
}

// the same with the query's side given (the reference loop's layout, where the sides alternate batch by batch)
template <int MODEL, int D>
__device__ __forceinline__ void true_key_lane_side(const QRows& q_true, const QRows& q_fixed, const QRows& q_rel, int64_t q, bool head,
                                                   float* __restrict__ key_true, unsigned long long* __restrict__ acc) {
    acc[q] = 0;
    float e[D];
    load_row<D>(e, q_true.row(q));
    const float* f = q_fixed.row(q);
    const float* r = q_rel.row(q);
    key_true[q] = head ? Scorer<MODEL, HEAD, D>::template score<false>(e, LazyCoef<MODEL, HEAD, D>{f, r})
                       : Scorer<MODEL, TAIL, D>::template score<false>(e, LazyCoef<MODEL, TAIL, D>{f, r});
}

// rank_common.h: FallbackPrep -- called by ALL threads of a refinement kernel's grid when the gate says heavy
template <int MODEL, int D>
__device__ __forceinline__ void fallback_prep(const FallbackPrep& fp, const QRows& q_fixed, const QRows& q_rel, int64_t q_head,
                                              int64_t q_tail) {
    using SH = Scorer<MODEL, HEAD, D>;
    using ST = Scorer<MODEL, TAIL, D>;
    float* coef_head = fp.coef;
    float* coef_tail = fp.coef + fallback_coef_tail_offset(D, q_head);
    const int64_t stride = (int64_t)gridDim.x * blockDim.x, i0 = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    for (int64_t q = i0; q < q_head + q_tail; q += stride) fp.acc[q] = 0;  // what the pre-pass counted is recounted
    const int64_t n_head = q_head * SH::C, total = n_head + q_tail * ST::C;
    for (int64_t i = i0; i < total; i += stride) {
        if (i < n_head) {
            const int64_t q = i / SH::C;
            coef_head[i] = SH::coef(q_fixed.row(q), q_rel.row(q), (int)(i % SH::C));
        } else {
            const int64_t k = i - n_head, q = k / ST::C;
            coef_tail[k] = ST::coef(q_fixed.row(q_head + q), q_rel.row(q_head + q), (int)(k % ST::C));
        }
    }
}

}  // namespace blp
