// rank_all.hip -- all-entities ranking on gfx950: stream the (N, D) candidate table once and turn
// every query into its rank counts without materialising the (Q, N) score matrix.
// Replaces train.py:146-171 + utils.py:103-105 of the reference (see include/blp_hip.h).
//
// This file: the dispatch (launch_rank_all: which kernel family ranks a block), the exact
// f32 kernels, and the building blocks the pre-pass paths (rank_sad*.hip, rank_gemm.hip) share with them.
//
// Kernel pipeline of one blp_rank_all call on the exact path (all on the caller's stream, no host sync):
//   1. prep_coef    (q, i) elementwise: hoist the query-only part of the score into C coefficients
//   2. true_key     score of the true entity, by the same arithmetic (one lane per query up to 2 048 queries -- 1 and 2
//                   are then one launch --, the cooperative routines of exact_coop.h above)
//   3. rank_tiles   the hot kernel, below  (small blocks: rank_small.hip instead of 1 and 3)
//   4. filter_finalize  64 queries per workgroup: scores of the filtered rows vs the true score, then
//                       counts[q] = {gt, ge, gt - fgt, ge - fge}  (no filter: a plain unpack of the accumulators)
// Queries arrive as QRows (launch.h): dense (Q, D) vectors or rows of the table / of rel_emb by index.
//
// rank_tiles layout.  One wavefront owns a tile of 64 consecutive table rows, one row per lane, held
// in D VGPRs (the sequential f32 sum of the reference forces one lane per (candidate, query) chain).
// The tile is fetched with coalesced 16-B/lane loads (8 rows x 128 B per wave instruction, full
// cache lines), transposed through a wave-private, bank-conflict-free LDS slab (row stride 36
// dwords), and then every query of the wave's query chunk is applied to it: the query coefficients
// are wave-uniform, so they arrive through the scalar cache as SGPR operands and cost no VGPRs and
// no LDS bandwidth.  Per query the wave does 2-3 VALU ops per element, two v_cmp + s_bcnt1 for the
// counts, and adds them to a per-wave LDS counter; counters are flushed once per wave with 64-bit
// atomics (gt in the low word, ge in the high word).
//   - few queries, huge table (Wikidata5M scale): the kernel is HBM-bound; waves grid-stride over
//     tiles and the table is read exactly once.
//   - many queries, small table (FB15k-237 test set): the table is L2/MALL-resident and the kernel
//     is f32-VALU-bound; parallelism comes from (tile x query chunk).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/blp_hip.h"  // BLP_METRIC_SUMS_DOUBLES
#include "dot_band.h"
#include "exact_coop.h"
#include "knobs.h"
#include "launch.h"
#include "rank_common.h"
#include "score_core.h"
#include "tile.h"

#pragma clang fp contract(off)

namespace blp {

constexpr int kQueryChunk = 128;     // queries per workgroup pass over its tiles
constexpr int kQB = 4;               // queries staged per LDS coefficient batch
__host__ __device__ constexpr int kMaxCoef(int D) { return 2 * D; }

// ------------------------------------------------------------------------------------------------
template <int MODEL, int D>
__global__ void prep_coef_kernel(const QRows q_fixed, const QRows q_rel,
                                 int64_t q_head, int64_t q_tail, float* __restrict__ coef_head,
                                 float* __restrict__ coef_tail) {
    using SH = Scorer<MODEL, HEAD, D>;
    using ST = Scorer<MODEL, TAIL, D>;
    const int64_t n_head = q_head * SH::C;
    const int64_t total = n_head + q_tail * ST::C;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        if (i < n_head) {
            const int64_t q = i / SH::C;
            coef_head[i] = SH::coef(q_fixed.row(q), q_rel.row(q), (int)(i % SH::C));
        } else {
            const int64_t k = i - n_head, q = k / ST::C;
            coef_tail[k] = ST::coef(q_fixed.row(q_head + q), q_rel.row(q_head + q), (int)(k % ST::C));
        }
    }
}

// The true entities' keys, by the cooperative exact routines of exact_coop.h (operation for operation the Scorer<>
// arithmetic: the same routines re-score the undecided pairs, whose keys are compared with these for ties).  A wave of
// 64 queries: TransE one lane per query with the three rows of every query fetched 8 rows x 128 B per instruction and
// transposed through LDS; the bilinear models 32 lanes per query, four queries per (one-wave) workgroup.  (One lane per query loading
// its own rows 16 bytes at a time: 96 load instructions x 64 cache lines each -- 28 us for the FB15k-237 test set.)
// The queries' accumulators start at zero.
__host__ __device__ constexpr int true_key_queries_per_block(int model) { return model == TRANSE ? 64 : 4; }
template <int MODEL, int D>
__global__ __launch_bounds__(64) void true_key_kernel(const QRows q_true,
                                const QRows q_fixed, const QRows q_rel,
                                int64_t q_head, int64_t q_tail, float* __restrict__ key_true,
                                unsigned long long* __restrict__ acc) {
    constexpr int QB = true_key_queries_per_block(MODEL);
    const int64_t Q = q_head + q_tail, q0 = blockIdx.x * (int64_t)QB;
    const int lane = threadIdx.x;
    if (lane < QB && q0 + lane < Q) acc[q0 + lane] = 0;
    auto true_vec = [&](int64_t q) { return q_true.row(q); };
    if constexpr (MODEL == TRANSE) {
        __shared__ __attribute__((aligned(16))) float slab[64 * kRefStride];
        const int64_t q = q0 + lane < Q ? q0 + lane : Q - 1;
        // (the run-time-width routine: its chunk loop is not unrolled -- unrolled, the compiler hoists every chunk's
        //  24 row loads to the top, 390 registers at D = 128 and spills at 256)
        const float key = transe_key_64_rt(true_vec(q), q_fixed.row(q), q_rel.row(q), D, q < q_head, slab, lane);
        if (q0 + lane < Q) key_true[q] = key;
    } else {
        const int half = lane >> 5, sub = lane & 31;
        for (int i = 0; i < QB / 2; ++i) {  // wave-uniform
            const int64_t qq = q0 + 2 * i + half, q = qq < Q ? qq : Q - 1;
            const float* e = true_vec(q);
            const float* f = q_fixed.row(q);
            const float* r = q_rel.row(q);
            const float key = q < q_head ? coop_score<MODEL, HEAD, D>(e, f, r, sub) : coop_score<MODEL, TAIL, D>(e, f, r, sub);
            if (sub == 0 && qq < Q) key_true[qq] = key;
        }
    }
}

// The bilinear models' approximate keys (rank_stream.hip, DOT): the GEMM operand row W_q of one query and its two band
// factors, by eight neighbouring lanes (`sub` = 0 .. 7; all eight must call).  row: where W_q goes (D floats); band2: eq =
// C u ||B_q|| -- times ||e|| it bounds |<W_q, e> - S_ref| for any summation order, rank_gemm.hip's header -- and et = the
// same bound for rows whose squares underflow (||e||^2 < 1e-30: every |e_k| <= 1.0001e-15, T <= ||B_q||_1 max |e_k|).
template <int MODEL, int D>
__device__ __forceinline__ void dot_prepare(const float* __restrict__ f, const float* __restrict__ r, bool head, bool live, int sub,
                                            float* __restrict__ row, float* __restrict__ band2) {
    float bsq = 0.f, bmax = 0.f, b1 = 0.f;
    bool bad = false;
#pragma unroll
    for (int i = 0; i < D / 32; ++i) {
        float w[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int k = 32 * i + 4 * sub + c;
            float b;
            if (head) gemm_operand<MODEL, HEAD>(f, r, k, D, w[c], b);
            else gemm_operand<MODEL, TAIL>(f, r, k, D, w[c], b);
            bsq += b * b;
            b1 += b;
            bad |= !(b <= 3.0e38f);  // NaN too
            bmax = b > bmax ? b : bmax;
        }
        if (live) *reinterpret_cast<float4*>(row + 32 * i + 4 * sub) = make_float4(w[0], w[1], w[2], w[3]);
    }
#pragma unroll
    for (int off = 1; off < 8; off <<= 1) {
        bsq += __shfl_xor(bsq, off);
        b1 += __shfl_xor(b1, off);
        const float m2 = __shfl_xor(bmax, off);
        bmax = m2 > bmax ? m2 : bmax;
        bad |= (bool)__shfl_xor((int)bad, off);
    }
    if (sub == 0 && live) {
        band2[0] = kBandC * 5.9604645e-8f * band_norm(bsq, bad ? __builtin_inff() : bmax);
        band2[1] = bad ? __builtin_inff() : kBandC * 5.9604645e-8f * 1.1e-15f * b1;
    }
}

// Few queries: latency, not throughput, is what a true-key launch costs, and one lane per query with all 96 of its
// row loads in flight at once is a single memory round trip (the cooperative kernel above walks four chunks one after
// the other: 128-query call 40 -> 49 us when it replaced this one outright).  Same Scorer<> arithmetic, bit for bit.
// Workgroups [key_blocks, gridDim.x), if any: the coefficient rows of prep_coef_kernel -- the exact path's two
// query-side launches in one (a 4-query pass over a 1/8 shard of the Wikidata5M table is ~50 us of table read; every
// launch of the chain is 4-5 us on top).
template <int MODEL, int D>
__global__ __launch_bounds__(64) void true_key_lane_kernel(const QRows q_true,
                                const QRows q_fixed, const QRows q_rel,
                                int64_t q_head, int64_t q_tail, float* __restrict__ key_true,
                                unsigned long long* __restrict__ acc, unsigned key_blocks,
                                float* __restrict__ coef_head, float* __restrict__ coef_tail, int64_t zero_slots,
                                unsigned dot_blocks, float* __restrict__ wq, float* __restrict__ band) {
    if (blockIdx.x >= gridDim.x - dot_blocks) {  // the last dot_blocks workgroups: eight queries' operand rows each (one pass)
        if constexpr (MODEL != TRANSE) {
            const int64_t Q = q_head + q_tail, qq = (blockIdx.x - (gridDim.x - dot_blocks)) * 8ll + (threadIdx.x >> 3);
            const bool live = qq < Q;
            const int64_t q = live ? qq : Q - 1;
            dot_prepare<MODEL, D>(q_fixed.row(q), q_rel.row(q), q < q_head, live, threadIdx.x & 7, wq + stream_dot_row((int)q) * D,
                                  band + 2 * q);
        }
        return;
    }
    if (blockIdx.x >= key_blocks) {
        using SH = Scorer<MODEL, HEAD, D>;
        using ST = Scorer<MODEL, TAIL, D>;
        const int64_t n_head = q_head * SH::C, total = n_head + q_tail * ST::C;
        // (partial counts of slots 1 .. zero_slots - 1, for a kernel that ADDS to them; slot 0 = acc[q]: the key lanes)
        const int64_t n_fill = gridDim.x - dot_blocks - key_blocks;
        for (int64_t i = (q_head + q_tail) + (blockIdx.x - key_blocks) * 64ll + threadIdx.x; i < zero_slots * (q_head + q_tail);
             i += n_fill * 64ll)
            acc[i] = 0;
        if (!coef_head) return;
        for (int64_t i = (blockIdx.x - key_blocks) * 64ll + threadIdx.x; i < total; i += n_fill * 64ll) {
            if (i < n_head) {
                const int64_t q = i / SH::C;
                coef_head[i] = SH::coef(q_fixed.row(q), q_rel.row(q), (int)(i % SH::C));
            } else {
                const int64_t k = i - n_head, q = k / ST::C;
                coef_tail[k] = ST::coef(q_fixed.row(q_head + q), q_rel.row(q_head + q), (int)(k % ST::C));
            }
        }
        return;
    }
    const int64_t q = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (q >= q_head + q_tail) return;
    true_key_lane<MODEL, D>(q_true, q_fixed, q_rel, q, q_head, key_true, acc);
}


// True keys (and zeroed accumulators); with coef_head / coef_tail also the coefficient rows of the exact path.
template <int MODEL, int D>
static void launch_true_key(const QRows q_true, const QRows& q_fixed,
                            const QRows& q_rel, int64_t q_head, int64_t q_tail, float* key_true, unsigned long long* acc,
                            hipStream_t stream, float* coef_head = nullptr, float* coef_tail = nullptr, int64_t zero_slots = 0,
                            float* wq = nullptr, float* band = nullptr) {
    const int64_t Q = q_head + q_tail;
    const int64_t n_coef = coef_head ? q_head * Scorer<MODEL, HEAD, D>::C + q_tail * Scorer<MODEL, TAIL, D>::C : 0;
    if (Q <= kTrueKeyLaneMaxQueries) {
        const int64_t key_blocks = (Q + 63) / 64, work = n_coef > zero_slots * Q ? n_coef : zero_slots * Q, want = (work + 63) / 64,
                      coef_blocks = want < 4096 ? want : 4096, dot_blocks = wq ? (Q + 7) / 8 : 0;  // (wq: one pass, Q <= 8)
        true_key_lane_kernel<MODEL, D><<<(unsigned)(key_blocks + coef_blocks + dot_blocks), 64, 0, stream>>>(
            q_true, q_fixed, q_rel, q_head, q_tail, key_true, acc, (unsigned)key_blocks, coef_head, coef_tail,
            zero_slots, (unsigned)dot_blocks, wq, band);
    } else {
        if (zero_slots > 1) (void)hipMemsetAsync(acc + Q, 0, (size_t)(zero_slots - 1) * Q * 8, stream);
        if (n_coef) {
            const int64_t blocks = (n_coef + 255) / 256;
            prep_coef_kernel<MODEL, D><<<(int)(blocks < 8192 ? blocks : 8192), 256, 0, stream>>>(q_fixed, q_rel, q_head, q_tail,
                                                                                                 coef_head, coef_tail);
        }
        constexpr int QB = true_key_queries_per_block(MODEL);
        true_key_kernel<MODEL, D><<<(unsigned)((Q + QB - 1) / QB), 64, 0, stream>>>(q_true, q_fixed, q_rel,
                                                                                     q_head, q_tail, key_true, acc);
    }
}

// Stage `count` floats (a multiple of 4, at most 2 * 4 * 256) from global memory into LDS with
// LDS-DMA (global_load_lds_dwordx4: 16 B per lane straight into LDS at wave-uniform base + lane * 16,
// no VGPR round trip).  The copy stays in flight until the caller's `s_waitcnt vmcnt(0)` + barrier.
__device__ __forceinline__ void stage_dma(const float* __restrict__ src, float* dst, int count, int wave, int lane) {
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int base = (k * kWaves + wave) * 256;  // floats: 64 lanes x 4 per wave instruction
        if (base + lane * 4 < count)
            __builtin_amdgcn_global_load_lds((global_cptr)(src + base + lane * 4), (lds_ptr)(dst + base), 16, 0, 0);
    }
}

// ------------------------------------------------------------------------------------------------
// TransE with scalar-cache coefficients, hand-pipelined (many-query mode).  Scalar loads return out of
// order, so every s_waitcnt on them is a full drain; the compiler issues the next chunk's load in the
// middle of a chunk and waits a few instructions later, exposing the scalar-cache latency 8-16 times
// per query.  Here the 64-byte chunk k+1 (and, on the last chunk, chunk 0 of the NEXT query) is
// requested before chunk k's 32-48 VALU instructions and drained after them, and one dword of every
// line of the query after next is touched once per query so that those requests hit the scalar cache.
template <int D, int CHUNK0, class Get>
__device__ __forceinline__ void l1_chunk(const float (&e)[D], float& acc, bool first, Get x_of) {
    // 16 elements starting at CHUNK0: differences kPipe ahead of the dependent |x| adds
    float x[kPipe];
    static_for<kPipe>([&](auto k) { x[k] = x_of(k); });
    static_for<16>([&](auto k) {
        constexpr int i = decltype(k)::value;
        const float cur = fabsf(x[i % kPipe]);
        if constexpr (i + kPipe < 16) x[i % kPipe] = x_of(ic<i + kPipe>{});
        acc = (first && i == 0) ? cur : acc + cur;
    });
}

// tail-replacing query: coefficient row = h + r (D floats).  `cur` holds chunk 0 on entry and chunk 0
// of `next_row` on exit.
template <int D>
__device__ __forceinline__ float transe_tail_sgpr(const float (&e)[D], sf16& cur, const float* row,
                                                  const float* next_row, const float* touch_row) {
    float acc = 0.f;
    static_for<D / 16>([&](auto kk) {
        constexpr int k = decltype(kk)::value;
        sf16 nxt;
        if constexpr (k + 1 < D / 16) nxt = sload16<(k + 1) * 64>(row); else nxt = sload16<0>(next_row);
        if constexpr (k == 0) stouch<D * 4>(touch_row);
        l1_chunk<D, 16 * k>(e, acc, k == 0, [&](auto ii) { return cur[decltype(ii)::value] - e[16 * k + decltype(ii)::value]; });
        sdrain(nxt);
        cur = nxt;
    });
    return -acc;
}

// head-replacing query: coefficient row = r (D floats) then t (D floats): (e + r) - t
template <int D>
__device__ __forceinline__ float transe_head_sgpr(const float (&e)[D], sf16& cur_r, sf16& cur_t, const float* row,
                                                  const float* next_row, const float* touch_row) {
    float acc = 0.f;
    static_for<D / 16>([&](auto kk) {
        constexpr int k = decltype(kk)::value;
        sf16 nr, nt;
        if constexpr (k + 1 < D / 16) { nr = sload16<(k + 1) * 64>(row); nt = sload16<D * 4 + (k + 1) * 64>(row); }
        else { nr = sload16<0>(next_row); nt = sload16<D * 4>(next_row); }
        if constexpr (k == 0) stouch<2 * D * 4>(touch_row);
        l1_chunk<D, 16 * k>(e, acc, k == 0, [&](auto ii) {
            constexpr int i = decltype(ii)::value;
            const float y = e[16 * k + i] + cur_r[i];
            return y - cur_t[i];
        });
        sdrain(nr, nt);
        cur_r = nr;
        cur_t = nt;
    });
    return -acc;
}

template <int SIDE, int D>
__device__ __forceinline__ void score_batch_transe_sgpr(const float (&e)[D], bool valid, const float* rows, int nq,
                                                        const float* __restrict__ key_true, unsigned* cnt, int wave,
                                                        int lane) {
    constexpr int C = Scorer<TRANSE, SIDE, D>::C;
    if (nq <= 0) return;
    // (no copy of `a` before its wait: between a hand-issued scalar load and its s_waitcnt the registers hold nothing yet --
    //  tests/test_abi.py reads the disassembly for exactly that)
    sf16 a = sload16<0>(rows), b;
    if constexpr (SIDE == HEAD) {
        b = sload16<D * 4>(rows);
        sdrain(a, b);
    } else {
        sdrain(a);
        b = a;  // (unused on this side)
    }
    for (int j = 0; j < nq; ++j) {
        const float* row = rows + (size_t)j * C;
        const float* next_row = rows + (size_t)(j + 1 < nq ? j + 1 : j) * C;
        // the four waves of a workgroup walk the same rows: they take turns touching the lines of the
        // query three ahead, so each wave pays the scalar-cache miss drain once every four queries
        const float* touch_row = ((j & 3) == wave && j + 3 < nq) ? rows + (size_t)(j + 3) * C : row;
        const float key = SIDE == HEAD ? transe_head_sgpr<D>(e, a, b, row, next_row, touch_row)
                                       : transe_tail_sgpr<D>(e, a, row, next_row, touch_row);
        const float kt = key_true[j];
        const unsigned gt = __popcll(__ballot(valid && key > kt));
        const unsigned ge = __popcll(__ballot(valid && key >= kt));
        if (lane == 0) {
            __hip_atomic_fetch_add(cnt + 2 * j, gt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            __hip_atomic_fetch_add(cnt + 2 * j + 1, ge, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        }
    }
}

// Score nq queries of one side against the wave's tile and add the two rank counts of each to the
// wave's LDS counters.  USE_SGPR = false: coefficient rows are staged in LDS at `cur` (DppCoef).
// USE_SGPR = true: `cur` is the wave-uniform global address of the rows, so the compiler fetches them
// with scalar loads and they are SGPR operands of full-rate VALU instructions -- the right choice
// when the whole query block (a few KB) stays resident in the scalar cache (STATIC mode); DPP
// operands issue at roughly a third of the plain VALU rate on gfx950 (tools/valu_ubench.hip).
template <int MODEL, int SIDE, int D, bool USE_SGPR>
__device__ __forceinline__ void score_batch(const float (&e)[D], bool valid, const float* cur, int nq,
                                            const float* __restrict__ key_true, unsigned* cnt, int lane) {
    using S = Scorer<MODEL, SIDE, D>;
    for (int j = 0; j < nq; ++j) {
        float key;
        if constexpr (USE_SGPR) {
            key = S::template score<false>(e, PtrCoef{cur + (size_t)j * S::C});
        } else {
            key = S::template score<false>(e, DppCoef{reinterpret_cast<const float4*>(cur + j * S::C) + (lane & 3)});
        }
        const float kt = key_true[j];
        const unsigned gt = __popcll(__ballot(valid && key > kt));
        const unsigned ge = __popcll(__ballot(valid && key >= kt));
        if (lane == 0) {  // ds_add_u32 without return: fire and forget
            __hip_atomic_fetch_add(cnt + 2 * j, gt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            __hip_atomic_fetch_add(cnt + 2 * j + 1, ge, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        }
    }
}

// Apply the n queries of one side (coefficient rows coef[0..n), C floats each, contiguous in global
// memory) to the wave's tile.  The workgroup stages kQB queries at a time into a double-buffered LDS
// block shared by its four waves (LDS-DMA, in flight under the previous batch's arithmetic); one
// barrier per batch.  Every wave of the workgroup calls this with the same n.
template <int MODEL, int SIDE, int D>
__device__ __forceinline__ void apply_queries(const float (&e)[D], bool valid, const float* __restrict__ coef,
                                              const float* __restrict__ key_true, int n, float* cbuf,
                                              unsigned* cnt, int wave, int lane) {
    using S = Scorer<MODEL, SIDE, D>;
    constexpr int C = S::C;
    static_assert(C % 16 == 0 && kQB * C <= 2 * kWaves * 256, "coefficient batch does not fit the staging plan");
    if (n <= 0) return;
    const int nb = (n + kQB - 1) / kQB;
    stage_dma(coef, cbuf, (n < kQB ? n : kQB) * C, wave, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int b = 0; b < nb; ++b) {
        const float* cur = cbuf + (b & 1) * (kQB * kMaxCoef(D));
        const int nq = n - b * kQB < kQB ? n - b * kQB : kQB;
        const int next = n - (b + 1) * kQB < kQB ? n - (b + 1) * kQB : kQB;  // <= 0 on the last batch
        // the other buffer was last read in batch b-1, which every wave left through the barrier below
        if (next > 0) stage_dma(coef + (size_t)(b + 1) * kQB * C, cbuf + ((b + 1) & 1) * (kQB * kMaxCoef(D)), next * C, wave, lane);
        score_batch<MODEL, SIDE, D, false>(e, valid, cur, nq, key_true + b * kQB, cnt + 2 * b * kQB, lane);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's share of the next batch has landed
        __syncthreads();
    }
}

// STATIC = a handful of queries (<= kQB head- and <= kQB tail-replacing, e.g. the reference's
// Wikidata5M eval batch of 2 triples): their coefficients (a few KB) stay in the scalar cache and are
// read as SGPR operands; the waves stream tiles independently, with no workgroup barrier in the loop.
template <int MODEL, int D, bool STATIC>
// TransE keeps D + ~35 VGPRs -> 3 waves/SIMD (fewer waves measured slower: 12.2 vs 9.9 ms); the bilinear
// fallback needs 32 more accumulators -> 2 waves/SIMD.  D = 256: a row alone is 256 registers -- one wave per SIMD and the
// accumulator half of the unified register file instead of scratch memory (round 2: 51-732 spilled registers).
__global__ __launch_bounds__(kWaves * 64, (D == 256 ? 1 : (MODEL == TRANSE ? 3 : 2))) void rank_tiles_kernel(
    const float* __restrict__ table, int64_t N, int64_t ld, const float* __restrict__ coef_head,
    const float* __restrict__ coef_tail, const float* __restrict__ key_true, int q_head, int q_tail,
    int n_tiles, int n_quad_groups, int q_chunk,
    unsigned long long* __restrict__ acc, const Gate gate) {
    if (gate.counter != nullptr && !gate_heavy(gate)) return;  // (the pre-pass paths' exact fallback: only when the lists ran full)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    float* slab = smem + wave * kSlabFloats;
    float* cbuf = smem + kWaves * kSlabFloats;
    unsigned* cnt = reinterpret_cast<unsigned*>(cbuf + 2 * kQB * kMaxCoef(D)) + wave * (2 * kQueryChunk);

    // workgroup -> (query chunk, group of tile quads); its four waves take four consecutive tiles
    const int chunk = blockIdx.x / n_quad_groups;
    const int qg = blockIdx.x % n_quad_groups;
    const int Q = q_head + q_tail;
    const int qa = chunk * q_chunk;  // q_chunk <= kQueryChunk: short chunks spread a small block over the chip
    const int qb = qa + q_chunk < Q ? qa + q_chunk : Q;
    // [qa, qb) split at q_head: head-replacing queries first (train.py:149 order)
    const int h_lo = qa < q_head ? qa : q_head, h_hi = qb < q_head ? qb : q_head;
    const int t_lo = qa > q_head ? qa : q_head, t_hi = qb > q_head ? qb : q_head;
    const float* ch = coef_head + (size_t)h_lo * Scorer<MODEL, HEAD, D>::C;
    const float* ct = coef_tail + (size_t)(t_lo - q_head) * Scorer<MODEL, TAIL, D>::C;

    for (int i = lane; i < 2 * kQueryChunk; i += 64) cnt[i] = 0;

    wave_lds_sync();

    const int n_quads = (n_tiles + kWaves - 1) / kWaves;
    for (int quad = qg; quad < n_quads; quad += n_quad_groups) {
        float e[D];
        const int64_t row0 = ((int64_t)quad * kWaves + wave) * kTileRows;
        // STATIC streams the table once (non-temporal loads); the many-query mode re-reads it from L2
        load_tile<D, STATIC>(e, table, N, ld, row0, slab, lane);  // rows past the end are clamped, then masked
        const bool valid = row0 + lane < N;
        if (STATIC) {
            score_batch<MODEL, HEAD, D, true>(e, valid, ch, h_hi - h_lo, key_true + h_lo, cnt + 2 * (h_lo - qa), lane);
            score_batch<MODEL, TAIL, D, true>(e, valid, ct, t_hi - t_lo, key_true + t_lo, cnt + 2 * (t_lo - qa), lane);
        } else {
            if constexpr (MODEL == TRANSE) {  // scalar-cache coefficients, hand-pipelined (no barriers)
                score_batch_transe_sgpr<TAIL, D>(e, valid, ct, t_hi - t_lo, key_true + t_lo, cnt + 2 * (t_lo - qa), wave, lane);
                // (until round 3 head queries of one relation could share e + r per tile through q_rel_id: the second copy
                //  of the tile it kept cost this kernel 147 spilled registers for a case no caller of the package hits)
                score_batch_transe_sgpr<HEAD, D>(e, valid, ch, h_hi - h_lo, key_true + h_lo, cnt + 2 * (h_lo - qa), wave, lane);
            } else {                          // LDS-staged coefficients + DPP broadcast
                apply_queries<MODEL, HEAD, D>(e, valid, ch, key_true + h_lo, h_hi - h_lo, cbuf, cnt + 2 * (h_lo - qa), wave, lane);
                apply_queries<MODEL, TAIL, D>(e, valid, ct, key_true + t_lo, t_hi - t_lo, cbuf, cnt + 2 * (t_lo - qa), wave, lane);
            }
        }
    }

    wave_lds_sync();
    for (int j = lane; j < qb - qa; j += 64) {
        const unsigned long long v = (unsigned long long)cnt[2 * j] | ((unsigned long long)cnt[2 * j + 1] << 32);
        if (v) atomicAdd(acc + qa + j, v);
    }
}

// ------------------------------------------------------------------------------------------------
// Packed counts of query q_base + (threadIdx.x & 63) summed over the n_partials slots acc[p * Q + q] -- workgroup of four
// waves: wave w adds the slots p = w (mod 4), the first wave gets the total (the others: garbage).  Ends with a barrier.
__device__ __forceinline__ unsigned long long sum_partials(const unsigned long long* __restrict__ acc, int n_partials,
                                                           int64_t Q, int64_t q_base,
                                                           unsigned long long (&sums)[3][kSweepQueries]) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t q = q_base + lane;
    unsigned long long a = 0;
    if (q < Q) {
        int p = wave;
#pragma unroll 1
        for (; p + 28 < n_partials; p += 32) {  // eight independent loads in flight
            unsigned long long v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = acc[(int64_t)(p + 4 * k) * Q + q];
#pragma unroll
            for (int k = 0; k < 8; ++k) a += v[k];
        }
        for (; p < n_partials; p += 4) a += acc[(int64_t)p * Q + q];
    }
    if (wave > 0) sums[wave - 1][lane] = a;
    __syncthreads();
    if (wave == 0) a += sums[0][lane] + sums[1][lane] + sums[2][lane];
    return a;
}

// ------------------------------------------------------------------------------------------------
// Last kernel of a call.  Filtered setting (train.py:159-171): a workgroup of four waves owns 64 consecutive queries.
// Its first wave reads their filter segments and scans the segment lengths; the workgroup's entries -- the rows the
// filter removes, usually few: most evaluation queries have none -- are then numbered 0 .. total - 1 across the 64
// queries and scored by the cooperative exact routines of exact_coop.h (TransE: 64 entries per wave and step, one
// lane each, rows gathered in whole lines; bilinear models: one entry per 32-lane half-wave), each entry finding its
// query by a binary search in the scanned lengths.  Entries at or above the true entity are counted per query in
// LDS; finally thread q writes query q's four counts.  (One wave per query, one lane per entry with its own
// row-by-row loads: 127 us for the 105 740 queries of the FB15k-237 block, of which 9 % have an entry.)
// Which side a query replaces: plain blocks -- queries [0, q_head) the head --, or the reference loop's layout
// (blp_rank_all_batches: batch after batch of `batch` triples, each batch as [its head queries | its tail queries]).
__host__ __device__ inline bool replaces_head(int64_t q, int64_t q_head, int64_t Q, int64_t batch) {
    if (batch <= 0) return q < q_head;
    const int64_t first = q / (2 * batch) * batch, n = Q / 2, nb = n - first < batch ? n - first : batch;
    return q - 2 * first < nb;
}

template <int MODEL, int D, class TE = float>  // TE: the candidate table's storage type (table_elem.h)
__global__ __launch_bounds__(256) void filter_finalize_kernel(
    const TE* __restrict__ table, int64_t N, int64_t ld, const QRows q_fixed,
    const QRows q_rel, const float* __restrict__ key_true, int64_t q_head,
    int64_t q_tail, const FilterSpec filter, const unsigned long long* __restrict__ acc, int n_partials,
    int32_t* __restrict__ counts, int64_t batch) {
    __shared__ unsigned long long partial_sums[3][kSweepQueries];
    __shared__ int prefix[kSweepQueries + 1];
    __shared__ unsigned removed[kSweepQueries][2];
    __shared__ __attribute__((aligned(16))) float slabs[MODEL == TRANSE ? 4 * 64 * kRefStride : 4];
    const int64_t Q = q_head + q_tail, q_base = (int64_t)blockIdx.x * kSweepQueries;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __shared__ const float* frow[kSweepQueries];  // the queries' two vectors: looked up once (indexed queries: a
    __shared__ const float* rrow[kSweepQueries];  // dependent load each), not once per filter entry
    if (threadIdx.x < 64) {
        const int64_t q = q_base + lane;
        frow[lane] = q_fixed.row(q < Q ? q : 0);
        rrow[lane] = q_rel.row(q < Q ? q : 0);
        int n = q < Q ? (int)(filter.hi[q] - filter.lo[q]) : 0;
        n = n > 0 ? n : 0;
        removed[lane][0] = removed[lane][1] = 0;
        int incl = n;  // inclusive scan over the wave
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int up = __shfl_up(incl, off);
            if (lane >= off) incl += up;
        }
        prefix[lane + 1] = incl;
        if (lane == 0) prefix[0] = 0;
    }
    __syncthreads();
    const int total = prefix[kSweepQueries];
    auto locate = [&](int x, int64_t& q, int64_t& row, int& slot) {  // entry number x of this workgroup
        int lo = 0, hi = kSweepQueries;                                // largest slot with prefix[slot] <= x
#pragma unroll
        for (int step = 0; step < 6; ++step) {
            const int mid = (lo + hi) >> 1;
            if (prefix[mid] <= x) lo = mid; else hi = mid;
        }
        slot = lo;
        q = q_base + slot;
        row = filter_row(filter, q, filter.lo[q] + (x - prefix[slot]), N);
    };
    if constexpr (MODEL == TRANSE) {
        float* slab = slabs + wave * 64 * kRefStride;
        for (int x0 = wave * 64; x0 < total; x0 += 4 * 64) {  // wave-uniform
            const int x = x0 + lane;
            int64_t q = q_base < Q ? q_base : 0, row = -1;
            int slot = 0;
            if (x < total) locate(x, q, row, slot);
            const bool live = row >= 0, head = replaces_head(q, q_head, Q, batch);
            const float key = transe_key_64<D, TE>(table + (live ? row : 0) * ld, frow[slot], rrow[slot], head, slab, lane);
            const float kt = key_true[q];
            if (live && key > kt) atomicAdd(&removed[slot][0], 1u);
            if (live && key >= kt) atomicAdd(&removed[slot][1], 1u);
        }
    } else {
        const int half = lane >> 5, sub = lane & 31;
        for (int x0 = wave * 2; x0 < total; x0 += 4 * 2) {  // wave-uniform; each half-wave takes one entry
            const int x = x0 + half;
            int64_t q = q_base < Q ? q_base : 0, row = -1;
            int slot = 0;
            if (x < total) locate(x, q, row, slot);
            const bool live = row >= 0, head = replaces_head(q, q_head, Q, batch);
            const TE* e = table + (live ? row : 0) * ld;
            float key;
            if (head) key = coop_score<MODEL, HEAD, D>(e, frow[slot], rrow[slot], sub);
            else key = coop_score<MODEL, TAIL, D>(e, frow[slot], rrow[slot], sub);
            const float kt = key_true[q];
            if (sub == 0 && live && key > kt) atomicAdd(&removed[slot][0], 1u);
            if (sub == 0 && live && key >= kt) atomicAdd(&removed[slot][1], 1u);
        }
    }
    const unsigned long long a = sum_partials(acc, n_partials, Q, q_base, partial_sums);  // ends with a barrier
    if (threadIdx.x < 64 && q_base + lane < Q) {
        const int64_t q = q_base + lane;
        const int32_t all_gt = (int32_t)(a & 0xffffffffull), all_ge = (int32_t)(a >> 32);
        reinterpret_cast<int4*>(counts)[q] =
            make_int4(all_gt, all_ge, all_gt - (int32_t)removed[lane][0], all_ge - (int32_t)removed[lane][1]);
    }
}

__global__ __launch_bounds__(256) void finalize_counts_kernel(const unsigned long long* __restrict__ acc, int n_partials,
                                                              int64_t Q, int32_t* __restrict__ counts) {
    __shared__ unsigned long long partial_sums[3][kSweepQueries];
    const int64_t q_base = (int64_t)blockIdx.x * kSweepQueries, q = q_base + threadIdx.x;
    const unsigned long long a = sum_partials(acc, n_partials, Q, q_base, partial_sums);
    if (threadIdx.x < 64 && q < Q) {
        const int32_t gt = (int32_t)(a & 0xffffffffull), ge = (int32_t)(a >> 32);
        reinterpret_cast<int4*>(counts)[q] = make_int4(gt, ge, gt, ge);
    }
}

// utils.py:104-109 from the counts
__global__ void rank_metrics_kernel(const int32_t* __restrict__ counts, int64_t Q, int k0, int k1, int k2,
                                    float* __restrict__ rr, uint8_t* __restrict__ hits) {
    const int64_t q = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (q >= Q) return;
    const int4 c = reinterpret_cast<const int4*>(counts)[q];
#pragma unroll
    for (int v = 0; v < 2; ++v) {
        const int64_t best = (int64_t)(v ? c.z : c.x) + 1, worst = v ? c.w : c.y;
        const float avg = (float)(best + worst) * 0.5f;
        rr[2 * q + v] = __fdiv_rn(1.0f, avg);
        hits[6 * q + 3 * v + 0] = avg <= (float)k0;
        hits[6 * q + 3 * v + 1] = avg <= (float)k1;
        hits[6 * q + 3 * v + 2] = avg <= (float)k2;
    }
}

// train.py:152-157 from the counts: sums[0..1] = sum of reciprocal ranks (raw, filtered), sums[2 + 3 v + j]
// = number of queries with avg rank <= k_j (v: raw, filtered), as f64.  Block b reduces queries
// [b per_block, (b + 1) per_block) in a fixed order into out[8 b ..]; with more than one block a second
// launch adds the partial results in block order: reproducible run to run (no floating-point atomics).
// (One block for 105 740 queries was a 58 us latency-bound tail of every evaluation step; 64 blocks: 12 us.)
__global__ __launch_bounds__(1024) void rank_metric_sums_kernel(const int32_t* __restrict__ counts_all, int64_t Q_all,
                                                               int64_t per_block, int k0, int k1, int k2,
                                                               double* __restrict__ out) {
    const int64_t first = blockIdx.x * per_block;
    const int32_t* counts = counts_all + 4 * first;
    const int64_t Q = Q_all - first < per_block ? Q_all - first : per_block;
    double* sums = out + 8 * blockIdx.x;
    __shared__ double sh[16][8];
    double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    auto add = [&](const int4& c) {
#pragma unroll
        for (int v = 0; v < 2; ++v) {
            const int64_t best = (int64_t)(v ? c.z : c.x) + 1, worst = v ? c.w : c.y;
            const float avg = (float)(best + worst) * 0.5f;
            acc[v] += (double)__fdiv_rn(1.0f, avg);
            acc[2 + 3 * v + 0] += avg <= (float)k0;
            acc[2 + 3 * v + 1] += avg <= (float)k1;
            acc[2 + 3 * v + 2] += avg <= (float)k2;
        }
    };
    const int4* rows = reinterpret_cast<const int4*>(counts);
    int64_t q = threadIdx.x;
    for (; q + 7 * (int64_t)blockDim.x < Q; q += 8 * (int64_t)blockDim.x) {  // eight loads in flight (one block: latency-bound)
        int4 c[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) c[j] = rows[q + j * (int64_t)blockDim.x];
#pragma unroll
        for (int j = 0; j < 8; ++j) add(c[j]);
    }
    for (; q < Q; q += blockDim.x) add(rows[q]);
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) acc[i] += __shfl_down(acc[i], off);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0)
        for (int i = 0; i < 8; ++i) sh[wave][i] = acc[i];
    __syncthreads();
    if (threadIdx.x < 8) {
        double total = 0.0;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) total += sh[w][threadIdx.x];
        sums[threadIdx.x] = total;
    }
}

// ------------------------------------------------------------------------------------------------
struct RankWorkspace {
    float* coef_head;
    float* coef_tail;
    float* key_true;
    unsigned long long* acc;
    float* wq;    // one pass's operand rows + bands of the bilinear models' approximate keys (rank_stream.hip, DOT)
    float* band;
    size_t bytes;
};

// floats of one pass's operand rows / bands (rank_common.h: StreamDot)
constexpr size_t dot_rows_floats(int D) { return (size_t)kStreamDotRows * D; }
constexpr size_t dot_band_floats() { return (size_t)2 * kStreamDotRows; }

// partial_slots > 1: the small-block kernel's per-slot partial counts, partial_slots x Q accumulators
static RankWorkspace carve_workspace(void* base, int D, int64_t q_head, int64_t q_tail, int partial_slots = 1) {
    // Both sides are sized for the larger coefficient layout (2 * D floats per query).
    RankWorkspace w;
    const int64_t Q = q_head + q_tail;
    char* p = static_cast<char*>(base);
    size_t off = 0;
    w.coef_head = reinterpret_cast<float*>(p + off); off = align_up(off + (size_t)q_head * max_coef(D) * 4, 256);
    w.coef_tail = reinterpret_cast<float*>(p + off); off = align_up(off + (size_t)q_tail * max_coef(D) * 4, 256);
    w.key_true = reinterpret_cast<float*>(p + off);  off = align_up(off + (size_t)Q * 4, 256);
    w.acc = reinterpret_cast<unsigned long long*>(p + off);   off = align_up(off + (size_t)Q * 8 * partial_slots, 256);
    w.wq = reinterpret_cast<float*>(p + off);   off = align_up(off + dot_rows_floats(D) * 4, 256);
    w.band = reinterpret_cast<float*>(p + off); off = align_up(off + dot_band_floats() * 4, 256);
    w.bytes = off;
    return w;
}

size_t rank_all_workspace_bytes(int model, int D, int64_t N, int64_t q_head, int64_t q_tail) {
    if (rank_sad_wide_applicable(model, D, q_head, q_tail)) return rank_sad_wide_workspace_bytes(model, D, N, q_head, q_tail);
    const int slots = rank_small_applicable(model, D, N, q_head, q_tail) ? rank_small_slots(N) : 1;
    size_t bytes = carve_workspace(nullptr, D, q_head, q_tail, slots).bytes;
    const size_t alt[2] = {rank_gemm_workspace_bytes(model, D, N, q_head, q_tail),
                           rank_sad_workspace_bytes(model, D, N, q_head, q_tail)};
    for (size_t a : alt) bytes = a > bytes ? a : bytes;
    return bytes;
}

// One ranking pass of <= 4 + 4 queries over the table (the reference's Wikidata5M eval batch): the streaming kernels of
// rank_stream.hip, or rank_tiles<STATIC> where they do not apply (bilinear models at D = 256; knob stream_kernel = 2).
// Coefficients, true keys and zeroed accumulators are the caller's.
template <int MODEL, int D>
static hipError_t launch_static_pass(const float* table, int64_t N, int64_t ld, const float* coef_head, const float* coef_tail,
                                     const float* key_true, int q_head, int q_tail, unsigned long long* acc, const StreamDot& dot,
                                     int n_cu, hipStream_t stream) {
    if (knob(KNOB_STREAM_KERNEL) != 2 && rank_stream_applicable(MODEL, D, N, ld, q_head, q_tail))
        return launch_rank_stream(MODEL, D, table, N, ld, coef_head, coef_tail, key_true, q_head, q_tail, acc, dot, n_cu, stream);
    // one resident set of persistent workgroups that grid-stride over the tile quads with no barrier in the loop, so waves
    // drift apart and one wave's arithmetic overlaps the others' loads
    const int64_t n_tiles = (N + kTileRows - 1) / kTileRows, n_quads = (n_tiles + kWaves - 1) / kWaves;
    const int64_t resident = (int64_t)n_cu * 3, groups = n_quads < resident ? n_quads : resident;
    const size_t lds = (size_t)kWaves * kSlabFloats * 4 + (size_t)2 * kQB * kMaxCoef(D) * 4 + (size_t)kWaves * 2 * kQueryChunk * 4;
    rank_tiles_kernel<MODEL, D, true><<<dim3((unsigned)groups), kWaves * 64, lds, stream>>>(
        table, N, ld, coef_head, coef_tail, key_true, q_head, q_tail, (int)n_tiles, (int)groups, kQueryChunk, acc, Gate{nullptr, 0u});
    return hipGetLastError();
}

template <int MODEL, int D>
static hipError_t rank_all_impl(const float* table, int64_t N, int64_t ld, const QRows q_fixed,
                                const QRows q_rel, const QRows q_true,
                                int64_t q_head, int64_t q_tail, const FilterSpec& filter, int32_t* counts, void* workspace,
                                int n_cu, hipStream_t stream, hipEvent_t ev_start, hipEvent_t ev_stop) {
    const int64_t Q = q_head + q_tail;
    if (Q == 0) return hipSuccess;
    if (rank_small_applicable(MODEL, D, N, q_head, q_tail)) {  // small block: 3 launches, no coefficient array
        const int slots = rank_small_slots(N);
        RankWorkspace w = carve_workspace(workspace, D, q_head, q_tail, slots);
        const bool coef = rank_small_wants_coef(MODEL, D, N);
        // (tiles sharing slots -- more tiles than slots, the scalar-register kernel only -- add to zeroed counts)
        const bool shared_slots = coef && (N + kTileRows - 1) / kTileRows > slots;
        launch_true_key<MODEL, D>(q_true, q_fixed, q_rel, q_head, q_tail, w.key_true, w.acc, stream,
                                  coef ? w.coef_head : nullptr, coef ? w.coef_tail : nullptr, shared_slots ? slots : 0);
        if (ev_start) (void)hipEventRecord(ev_start, stream);
        const hipError_t err = launch_rank_small(MODEL, D, table, N, ld, q_fixed, q_rel, w.coef_head, w.coef_tail, w.key_true, q_head,
                                                 q_tail, w.acc, n_cu, stream);
        if (err != hipSuccess) return err;
        if (ev_stop) (void)hipEventRecord(ev_stop, stream);
        return launch_filter_finalize(MODEL, D, table, N, ld, q_fixed, q_rel, w.key_true, q_head, q_tail, filter, w.acc, counts,
                                      stream, slots);
    }
    RankWorkspace w = carve_workspace(workspace, D, q_head, q_tail);
    if (q_head <= kQB && q_tail <= kQB) w.coef_tail = w.coef_head + q_head * Scorer<MODEL, HEAD, D>::C;  // (static mode: one array, as a pass of blp_rank_all_batches has it -- the streaming kernels' layout)
    // <= 4 + 4 queries of a bilinear model: the streaming kernels take approximate keys first (operand rows + bands)
    const bool dot_keys = q_head <= kQB && q_tail <= kQB && N > 0 && rank_stream_wants_dot(MODEL, D, N, ld, q_head, q_tail);
    launch_true_key<MODEL, D>(q_true, q_fixed, q_rel, q_head, q_tail, w.key_true, w.acc, stream, dot_keys ? nullptr : w.coef_head,
                              w.coef_tail, 0, dot_keys ? w.wq : nullptr, w.band);  // (approximate keys: no coefficient rows)

    if (N > 0) {
        const int64_t n_tiles = (N + kTileRows - 1) / kTileRows;
        const int64_t n_quads = (n_tiles + kWaves - 1) / kWaves;
        // Queries per workgroup: kQueryChunk amortises a tile fetch best, but a small block then occupies only
        // n_quads workgroups (57 for the FB15k-237 table: a fifth of the chip, one wave per SIMD): halve the chunk
        // while the grid would not even fill the resident set once.
        int q_chunk = kQueryChunk;
        while (q_chunk > 16 && n_quads * ((Q + q_chunk - 1) / q_chunk) < (int64_t)n_cu * 3) q_chunk >>= 1;
        if (const long long forced = knob(KNOB_EXACT_QUERY_CHUNK); forced == 16 || forced == 32 || forced == 64 || forced == 128)
            q_chunk = (int)forced;
        const int64_t n_chunks = (Q + q_chunk - 1) / q_chunk;
        // Resident workgroups: 3 per CU (12 waves at <= 168 VGPRs, ~48 KB of LDS each).  With plenty
        // of (tile quad, chunk) pairs give every workgroup one quad (many short workgroups -> no
        // tail); otherwise grid-stride ~2 resident sets of workgroups over the quads.
        const int64_t resident = (int64_t)n_cu * 3;
        const bool static_mode = q_head <= kQB && q_tail <= kQB;
        int64_t groups;
        if (static_mode) {
            // few queries, long table (HBM-bound): one resident set of persistent workgroups that
            // grid-stride over the tile quads with no barrier in the loop, so waves drift apart and
            // one wave's arithmetic overlaps the others' loads.  (One quad per workgroup ran every
            // wave of the chip in lock-step: load phase, then compute phase, never overlapped.)
            groups = n_quads < resident ? n_quads : resident;
        } else if (n_quads * n_chunks > 8 * resident) {
            groups = n_quads;
        } else {
            const int64_t want = (2 * resident + n_chunks - 1) / n_chunks;
            groups = n_quads < want ? n_quads : want;
        }
        const int64_t blocks = groups * n_chunks;
        const size_t lds = (size_t)kWaves * kSlabFloats * 4 + (size_t)2 * kQB * kMaxCoef(D) * 4 +
                     (size_t)kWaves * 2 * kQueryChunk * 4;

        if (ev_start) (void)hipEventRecord(ev_start, stream);
        if (static_mode) {
            StreamDot dot;
            if (dot_keys) { dot.wq = w.wq; dot.band = w.band; dot.q_fixed = q_fixed; dot.q_rel = q_rel; }
            const hipError_t err = launch_static_pass<MODEL, D>(table, N, ld, w.coef_head, w.coef_tail, w.key_true, (int)q_head,
                                                                (int)q_tail, w.acc, dot, n_cu, stream);
            if (err != hipSuccess) return err;
        } else
            rank_tiles_kernel<MODEL, D, false><<<dim3((unsigned)blocks), kWaves * 64, lds, stream>>>(
                table, N, ld, w.coef_head, w.coef_tail, w.key_true, (int)q_head, (int)q_tail, (int)n_tiles,
                (int)groups, q_chunk, w.acc, Gate{nullptr, 0u});
        if (ev_stop) (void)hipEventRecord(ev_stop, stream);
    }

    return launch_filter_finalize(MODEL, D, table, N, ld, q_fixed, q_rel, w.key_true, q_head, q_tail, filter, w.acc, counts,
                                  stream);
}

template <int MODEL>
static hipError_t rank_all_dim(int D, const float* table, int64_t N, int64_t ld, const QRows q_fixed,
                               const QRows q_rel, const QRows q_true,
                               int64_t q_head, int64_t q_tail, const FilterSpec& filter, int32_t* counts, void* workspace, int n_cu,
                               hipStream_t stream, hipEvent_t ev_start, hipEvent_t ev_stop) {
#define BLP_RANK_CASE(DD)                                                                              \
    case DD:                                                                                           \
        return rank_all_impl<MODEL, DD>(table, N, ld, q_fixed, q_rel, q_true, q_head, q_tail, \
                                        filter, counts, workspace, n_cu, stream, ev_start, ev_stop);
    switch (D) {
        BLP_RANK_CASE(64)
        BLP_RANK_CASE(128)
        BLP_RANK_CASE(256)
    default:
        return hipErrorInvalidValue;
    }
#undef BLP_RANK_CASE
}

hipError_t launch_rank_all(int model, int D, const float* table, int64_t N, int64_t ld,
                           const QRows q_fixed, const QRows q_rel,
                           const QRows q_true, int64_t q_head, int64_t q_tail,
                           const FilterSpec& filter, int32_t* counts,
                           void* workspace, int n_cu, hipStream_t stream, hipEvent_t ev_start,
                           hipEvent_t ev_stop) {
    const bool small = rank_small_applicable(model, D, N, q_head, q_tail);  // a small block: rank_small.hip, below
    if (!small && rank_gemm_applicable(model, D, q_head, q_tail))
        return launch_rank_all_gemm(model, D, table, N, ld, q_fixed, q_rel, q_true, q_head, q_tail,
                                    filter, counts, workspace, n_cu, stream, ev_start, ev_stop);
    if (!small && rank_sad_wide_applicable(model, D, q_head, q_tail))
        return launch_rank_all_sad_wide(D, table, N, ld, q_fixed, q_rel, q_true, q_head, q_tail, filter,
                                        counts, workspace, n_cu, stream, ev_start, ev_stop);
    if (!small && rank_sad_applicable(model, D, N, q_head, q_tail))
        return launch_rank_all_sad(D, table, N, ld, q_fixed, q_rel, q_true, q_head, q_tail, filter,
                                   counts, workspace, n_cu, stream, ev_start, ev_stop);
    // everything else: the exact f32 kernels (few queries, D = 256 bilinear, or the rank_kernel knob)
    switch (model) {
    case TRANSE:   return rank_all_dim<TRANSE>(D, table, N, ld, q_fixed, q_rel, q_true, q_head, q_tail, filter, counts, workspace, n_cu, stream, ev_start, ev_stop);
    case DISTMULT: return rank_all_dim<DISTMULT>(D, table, N, ld, q_fixed, q_rel, q_true, q_head, q_tail, filter, counts, workspace, n_cu, stream, ev_start, ev_stop);
    case COMPLEX:  return rank_all_dim<COMPLEX>(D, table, N, ld, q_fixed, q_rel, q_true, q_head, q_tail, filter, counts, workspace, n_cu, stream, ev_start, ev_stop);
    case SIMPLE:   return rank_all_dim<SIMPLE>(D, table, N, ld, q_fixed, q_rel, q_true, q_head, q_tail, filter, counts, workspace, n_cu, stream, ev_start, ev_stop);
    default:       return hipErrorInvalidValue;
    }
}

// ------------------------------------------------------------------------------------------------
// The pre-pass paths' bounded worst case (rank_common.h: Gate): when the lists ran full, the exact kernel re-ranks the block.
size_t exact_fallback_coef_floats(int D, int64_t q_head, int64_t q_tail) {
    return (size_t)(q_head + q_tail) * max_coef(D) + 64;
}

template <int MODEL, int D>
static hipError_t exact_fallback_impl(const float* table, int64_t N, int64_t ld, const QRows q_fixed, const QRows q_rel, int64_t q_head,
                                      int64_t q_tail, float* coef, const float* key_true, unsigned long long* acc, Gate gate, int n_cu,
                                      hipStream_t stream) {
    const int64_t Q = q_head + q_tail;
    if (Q == 0 || N == 0) return hipSuccess;
    float* coef_head = coef;  // (written, with the zeroed counts, by the path's last refinement kernel: exact_coop.h: fallback_prep)
    float* coef_tail = coef + fallback_coef_tail_offset(D, q_head);
    // one workgroup per query chunk (or a few), grid-striding over the tile quads: as many workgroups as the chip holds a few
    // times over, not one per (quad, chunk) -- tens of thousands of workgroups that return at once would cost the common
    // case ~20 us
    const int64_t n_tiles = (N + kTileRows - 1) / kTileRows, n_quads = (n_tiles + kWaves - 1) / kWaves;
    const int64_t n_chunks = (Q + kQueryChunk - 1) / kQueryChunk, resident = (int64_t)n_cu * 3;
#ifndef BLP_FALLBACK_GRID
#define BLP_FALLBACK_GRID 16  // x the resident workgroups ([measured] FB15k-237 TransE block at 5 % ties, 4 / 16 / one workgroup per tile quad: 17.5 / 15.6 / 13.2 ms; what the gated launch costs the random-data DistMult step: 0 / +0.2 / +0.9 %)
#endif
    int64_t groups = (BLP_FALLBACK_GRID * resident + n_chunks - 1) / n_chunks;
    groups = groups < 1 ? 1 : (groups > n_quads ? n_quads : groups);
    const size_t lds = (size_t)kWaves * kSlabFloats * 4 + (size_t)2 * kQB * kMaxCoef(D) * 4 + (size_t)kWaves * 2 * kQueryChunk * 4;
    if (groups * n_chunks > 0x7fffffff) return hipErrorInvalidValue;
    rank_tiles_kernel<MODEL, D, false><<<dim3((unsigned)(groups * n_chunks)), kWaves * 64, lds, stream>>>(
        table, N, ld, coef_head, coef_tail, key_true, (int)q_head, (int)q_tail, (int)n_tiles, (int)groups, kQueryChunk, acc, gate);
    return hipGetLastError();
}

hipError_t launch_exact_fallback(int model, int D, const float* table, int64_t N, int64_t ld, const QRows q_fixed, const QRows q_rel,
                                 int64_t q_head, int64_t q_tail, float* coef, const float* key_true, unsigned long long* acc, Gate gate,
                                 int n_cu, hipStream_t stream) {
#define BLP_FALLBACK_CASE(M, DD)                                                                                        \
    if (model == M && D == DD)                                                                                          \
        return exact_fallback_impl<M, DD>(table, N, ld, q_fixed, q_rel, q_head, q_tail, coef, key_true, acc, gate, n_cu, stream);
    BLP_FALLBACK_CASE(TRANSE, 64) BLP_FALLBACK_CASE(TRANSE, 128) BLP_FALLBACK_CASE(TRANSE, 256)
    BLP_FALLBACK_CASE(DISTMULT, 64) BLP_FALLBACK_CASE(DISTMULT, 128)
    BLP_FALLBACK_CASE(COMPLEX, 64) BLP_FALLBACK_CASE(COMPLEX, 128)
    BLP_FALLBACK_CASE(SIMPLE, 64) BLP_FALLBACK_CASE(SIMPLE, 128)
#undef BLP_FALLBACK_CASE
    return hipErrorInvalidValue;
}

// ------------------------------------------------------------------------------------------------
// Measurement: how much a pre-pass left to the exact path (blp_rank_all_prepass_stats).  Synchronous, allocates 32 bytes.
__global__ __launch_bounds__(256) void count_bits_kernel(const unsigned* __restrict__ flags, int64_t n_words,
                                                         const uint2* __restrict__ pairs, const unsigned* __restrict__ n_pairs,
                                                         int mask_entries, unsigned max_entries, unsigned long long* __restrict__ out) {
    unsigned long long f = 0, m = 0;
    const int64_t stride = (int64_t)gridDim.x * 256, i0 = (int64_t)blockIdx.x * 256 + threadIdx.x;
    for (int64_t i = i0; i < n_words; i += stride) f += __popc(flags[i]);
    const int64_t n = *n_pairs < max_entries ? *n_pairs : max_entries;
    if (mask_entries)
        for (int64_t i = i0; i < n; i += stride) m += __popc(pairs[i].y & 0xffffu);
    for (int off = 32; off > 0; off >>= 1) {
        f += __shfl_xor(f, off);
        m += __shfl_xor(m, off);
    }
    if ((threadIdx.x & 63) == 0) {
        if (f) atomicAdd(out, f);
        if (m) atomicAdd(out + 1, m);
    }
    if (i0 == 0) out[2] = (unsigned long long)n;
}

hipError_t launch_count_bits(const unsigned* flags, int64_t n_words, const uint2* pairs, const unsigned* n_pairs, bool mask_entries,
                             unsigned long long host_out[3], hipStream_t stream, unsigned max_entries) {
    unsigned long long* dev = nullptr;
    hipError_t err = hipMalloc(&dev, 32);
    if (err != hipSuccess) return err;
    err = hipMemsetAsync(dev, 0, 32, stream);
    if (err == hipSuccess) {
        count_bits_kernel<<<dim3(1024), 256, 0, stream>>>(flags, n_words, pairs, n_pairs, mask_entries ? 1 : 0, max_entries, dev);
        err = hipGetLastError();
    }
    if (err == hipSuccess) err = hipMemcpyAsync(host_out, dev, 24, hipMemcpyDeviceToHost, stream);
    if (err == hipSuccess) err = hipStreamSynchronize(stream);
    (void)hipFree(dev);
    return err;
}

hipError_t prepass_stats(int model, int D, int64_t N, int64_t q_head, int64_t q_tail, const void* workspace, PrepassStats* out,
                         hipStream_t stream) {
    out->pairs = (long long)(q_head + q_tail) * (long long)N;
    out->listed = out->flagged_rows = 0;
    out->path = 0;
    if (rank_small_applicable(model, D, N, q_head, q_tail)) return hipSuccess;  // (the order of launch_rank_all)
    if (rank_gemm_applicable(model, D, q_head, q_tail)) {
        int device = 0;
        (void)hipGetDevice(&device);
        return gemm_prepass_stats(D, N, q_head, q_tail, workspace, device, out, stream);
    }
    if (rank_sad_wide_applicable(model, D, q_head, q_tail)) {
        out->path = 4;
        out->listed = out->flagged_rows = -1;
        return hipSuccess;
    }
    if (rank_sad_applicable(model, D, N, q_head, q_tail)) return sad_prepass_stats(D, N, q_head, q_tail, workspace, out, stream);
    return hipSuccess;
}

// ------------------------------------------------------------------------------------------------
// Building blocks the pre-pass paths (rank_gemm.hip, rank_sad.hip) share with the exact path: true-entity keys (+ zeroed
// accumulators) and the filter / finalize step.  They read the query vectors directly: no coefficient array.
template <int MODEL, int D>
static hipError_t true_keys_impl(const QRows q_fixed, const QRows q_rel, int64_t q_head, int64_t q_tail,
                                 const QRows q_true,
                                 float* key_true, unsigned long long* acc, hipStream_t stream) {
    launch_true_key<MODEL, D>(q_true, q_fixed, q_rel, q_head, q_tail, key_true, acc, stream);
    return hipGetLastError();
}

template <int MODEL, int D, class TE = float>
static hipError_t filter_finalize_impl(const TE* table, int64_t N, int64_t ld, const QRows q_fixed,
                                       const QRows q_rel, const float* key_true, int64_t q_head, int64_t q_tail,
                                       const FilterSpec& filter, const unsigned long long* acc, int n_partials,
                                       int32_t* counts, hipStream_t stream, int64_t batch = 0) {
    filter_finalize_kernel<MODEL, D, TE><<<(int)((q_head + q_tail + kSweepQueries - 1) / kSweepQueries), 256, 0, stream>>>(
        table, N, ld, q_fixed, q_rel, key_true, q_head, q_tail, filter, acc, n_partials, counts, batch);
    return hipGetLastError();
}

#define BLP_DISPATCH_MODEL_DIM(FN, ...)                                                     \
    switch (model * 1000 + D) {                                                             \
    case TRANSE * 1000 + 64: return FN<TRANSE, 64>(__VA_ARGS__);                            \
    case TRANSE * 1000 + 128: return FN<TRANSE, 128>(__VA_ARGS__);                          \
    case TRANSE * 1000 + 256: return FN<TRANSE, 256>(__VA_ARGS__);                          \
    case DISTMULT * 1000 + 64: return FN<DISTMULT, 64>(__VA_ARGS__);                        \
    case DISTMULT * 1000 + 128: return FN<DISTMULT, 128>(__VA_ARGS__);                      \
    case DISTMULT * 1000 + 256: return FN<DISTMULT, 256>(__VA_ARGS__);                      \
    case COMPLEX * 1000 + 64: return FN<COMPLEX, 64>(__VA_ARGS__);                          \
    case COMPLEX * 1000 + 128: return FN<COMPLEX, 128>(__VA_ARGS__);                        \
    case COMPLEX * 1000 + 256: return FN<COMPLEX, 256>(__VA_ARGS__);                        \
    case SIMPLE * 1000 + 64: return FN<SIMPLE, 64>(__VA_ARGS__);                            \
    case SIMPLE * 1000 + 128: return FN<SIMPLE, 128>(__VA_ARGS__);                          \
    case SIMPLE * 1000 + 256: return FN<SIMPLE, 256>(__VA_ARGS__);                          \
    default: return hipErrorInvalidValue;                                                   \
    }

hipError_t launch_true_keys(int model, int D, const QRows q_fixed, const QRows q_rel, int64_t q_head, int64_t q_tail,
                            const QRows q_true,
                            float* key_true, unsigned long long* acc, hipStream_t stream) {
    BLP_DISPATCH_MODEL_DIM(true_keys_impl, q_fixed, q_rel, q_head, q_tail, q_true, key_true, acc, stream)
}

hipError_t launch_filter_finalize(int model, int D, const float* table, int64_t N, int64_t ld, const QRows q_fixed,
                                  const QRows q_rel, const float* key_true, int64_t q_head, int64_t q_tail,
                                  const FilterSpec& filter, const unsigned long long* acc, int32_t* counts,
                                  hipStream_t stream, int n_partials) {
    const int64_t Q = q_head + q_tail;
    if (!filter.on()) {
        finalize_counts_kernel<<<(int)((Q + kSweepQueries - 1) / kSweepQueries), 256, 0, stream>>>(acc, n_partials, Q, counts);
        return hipGetLastError();
    }
    BLP_DISPATCH_MODEL_DIM(filter_finalize_impl, table, N, ld, q_fixed, q_rel, key_true, q_head, q_tail, filter, acc,
                           n_partials, counts, stream)
}

// ------------------------------------------------------------------------------------------------
// Many passes of <= 4 + 4 queries each over one table, back to back (blp_rank_all_batches with a pass per batch: the
// reference's Wikidata5M evaluation, eval_batch_size = 2 -- scripts/blp-*-wikidata5m.sh:18, train.py:128-171).  A pass
// on its own is a chain of three launches (true keys + coefficients -> streaming kernel -> filter + finalize): on a 1/8
// shard of the 4.6 M-row table the streaming kernel takes 49 us of a 62 us pass.  Here the first and the last stage are
// done ONCE for all passes -- one launch writes every pass's coefficient rows (in the layout its streaming kernel reads),
// every query's true key and zeroed accumulator; one launch at the end applies the filter and writes all counts -- and
// the streaming kernels follow each other with nothing in between.  Queries in the reference loop's layout.
template <int MODEL, int D>
__global__ __launch_bounds__(64) void prep_passes_kernel(const QRows q_fixed, const QRows q_rel, const QRows q_true, int64_t n,
                                                        int64_t batch, float* __restrict__ coef, float* __restrict__ key_true,
                                                        unsigned long long* __restrict__ acc, unsigned key_blocks,
                                                        unsigned dot_blocks, float* __restrict__ wq, float* __restrict__ band,
                                                        int acc_slots) {
    using SH = Scorer<MODEL, HEAD, D>;
    using ST = Scorer<MODEL, TAIL, D>;
    constexpr int CP = SH::C + ST::C, CM = SH::C > ST::C ? SH::C : ST::C;  // floats per triple of a pass / widest row
    const int64_t Q = 2 * n;
    if (blockIdx.x < key_blocks) {  // one lane per query: its true key (all of its row loads in flight at once), acc = 0
        const int64_t q = (int64_t)blockIdx.x * 64 + threadIdx.x;
        if (q < Q) {
            true_key_lane_side<MODEL, D>(q_true, q_fixed, q_rel, q, replaces_head(q, 0, Q, batch), key_true, acc);
            for (int s = 1; s < acc_slots; ++s) acc[(int64_t)s * Q + q] = 0;  // (all passes in one launch: replicated accumulators)
        }
        return;
    }
    if (blockIdx.x >= gridDim.x - dot_blocks) {  // the approximate keys' operand rows, eight queries per workgroup
        if constexpr (MODEL != TRANSE) {
            const int64_t qq = (blockIdx.x - (gridDim.x - dot_blocks)) * 8ll + (threadIdx.x >> 3);
            const bool live = qq < Q;
            const int64_t q = live ? qq : Q - 1, pass = q / (2 * batch), o = q - pass * 2 * batch;  // query o of its pass
            dot_prepare<MODEL, D>(q_fixed.row(q), q_rel.row(q), replaces_head(q, 0, Q, batch), live, threadIdx.x & 7,
                                  wq + (pass * kStreamDotRows + stream_dot_row((int)o)) * D, band + 2 * q);
        }
        return;
    }
    const int64_t n_fill = gridDim.x - dot_blocks - key_blocks;
    for (int64_t i = (int64_t)(blockIdx.x - key_blocks) * 64 + threadIdx.x; i < Q * CM; i += n_fill * 64) {
        const int64_t q = i / CM;
        const int c = (int)(i - q * CM);
        const int64_t first = q / (2 * batch) * batch, nb = n - first < batch ? n - first : batch, o = q - 2 * first;
        float* pass = coef + first * CP;  // this pass: nb head rows of SH::C floats, then nb tail rows of ST::C
        if (o < nb) {
            if (c < SH::C) pass[o * SH::C + c] = SH::coef(q_fixed.row(q), q_rel.row(q), c);
        } else if (c < ST::C) {
            pass[nb * SH::C + (o - nb) * ST::C + c] = ST::coef(q_fixed.row(q), q_rel.row(q), c);
        }
    }
}

struct PassesWorkspace {
    float* coef;
    float* key_true;
    unsigned long long* acc;
    float* wq;    // per pass kStreamDotRows operand rows; two band factors per query (rank_stream.hip, DOT)
    float* band;
    size_t bytes;
};
static PassesWorkspace carve_passes(void* base, int D, int64_t n, int64_t batch) {
    PassesWorkspace w;
    char* p = static_cast<char*>(base);
    const int64_t passes = batch > 0 ? (n + batch - 1) / batch : 0;
    size_t off = 0;
    w.coef = reinterpret_cast<float*>(p + off);     off = align_up(off + (size_t)n * 2 * max_coef(D) * 4, 256);
    w.key_true = reinterpret_cast<float*>(p + off); off = align_up(off + (size_t)n * 2 * 4, 256);
    w.acc = reinterpret_cast<unsigned long long*>(p + off); off = align_up(off + (size_t)n * 2 * 8 * kStreamAccSlots, 256);
    w.wq = reinterpret_cast<float*>(p + off);       off = align_up(off + (size_t)passes * dot_rows_floats(D) * 4, 256);
    w.band = reinterpret_cast<float*>(p + off);     off = align_up(off + (size_t)n * 2 * 2 * 4, 256);
    w.bytes = off;
    return w;
}

bool rank_static_passes_applicable(int model, int D, int64_t N, int64_t batch) {
    if (batch <= 0 || batch > kQB || N <= 0 || !(D == 64 || D == 128 || D == 256)) return false;
    return !rank_small_applicable(model, D, N, batch, batch) && knob(KNOB_RANK_KERNEL) != 1;  // i.e. rank_all_impl's static mode
}
size_t rank_static_passes_workspace_bytes(int D, int64_t n, int64_t batch) { return carve_passes(nullptr, D, n, batch).bytes; }

template <int MODEL, int D>
static hipError_t static_passes_impl(const void* table_any, int dtype, int64_t N, int64_t ld, const QRows q_fixed, const QRows q_rel,
                                     const QRows q_true, int64_t n, int64_t batch, const FilterSpec& filter, int32_t* counts,
                                     void* workspace, int n_cu, hipStream_t stream, hipEvent_t ev_start, hipEvent_t ev_stop) {
    using SH = Scorer<MODEL, HEAD, D>;
    using ST = Scorer<MODEL, TAIL, D>;
    const PassesWorkspace w = carve_passes(workspace, D, n, batch);
    const int64_t Q = 2 * n, key_blocks = (Q + 63) / 64;
    int64_t coef_blocks = (Q * (SH::C > ST::C ? SH::C : ST::C) + 63) / 64;
    coef_blocks = coef_blocks > 8192 ? 8192 : coef_blocks;
    const float* table = static_cast<const float*>(table_any);  // (an f32 table; a 16-bit one: the branch right below)
    if (dtype != kTableF32) {
        // A 16-BIT TABLE (table_elem.h): the ring kernels of rank_stream16.hip, all passes in one launch, the bilinear models
        // with approximate keys; the queries' own vectors (q_fixed / q_true index an f32 `source`) and everything derived
        // from them -- coefficient rows, operand rows, true keys -- are what the f32 path makes; the filter's rows are read
        // from the 16-bit table and widened.
        if constexpr (D == 64) {
            return hipErrorInvalidValue;
        } else {
            if (!rank_stream16_takes_passes(MODEL, D, N, ld, batch, n)) return hipErrorInvalidValue;
            const bool dot16 = MODEL != TRANSE;
            prep_passes_kernel<MODEL, D><<<dim3((unsigned)(key_blocks + (dot16 ? 0 : coef_blocks) + (dot16 ? (Q + 7) / 8 : 0))), 64, 0, stream>>>(
                q_fixed, q_rel, q_true, n, batch, w.coef, w.key_true, w.acc, (unsigned)key_blocks, (unsigned)(dot16 ? (Q + 7) / 8 : 0), w.wq,
                w.band, kStreamAccSlots);
            StreamPasses passes;
            passes.n_passes = (int)((n + batch - 1) / batch);
            passes.batch = (int)batch;
            passes.n = n;
            passes.acc_slots = kStreamAccSlots;
            StreamDot dot;
            if (dot16) { dot.wq = w.wq; dot.band = w.band; dot.q_fixed = q_fixed; dot.q_rel = q_rel; dot.q0 = 0; }
            if (ev_start) (void)hipEventRecord(ev_start, stream);
            const int64_t per_side = n < batch ? n : batch;  // (a single short pass is described by these, not by passes.batch)
            const hipError_t err = launch_rank_stream16(MODEL, D, dtype, table_any, N, ld, w.coef, w.key_true, per_side, per_side, w.acc, dot,
                                                        n_cu, stream, passes);
            if (err != hipSuccess) return err;
            if (ev_stop) (void)hipEventRecord(ev_stop, stream);
            if (filter.on()) {
                if (dtype == kTableF16)
                    return filter_finalize_impl<MODEL, D, _Float16>(static_cast<const _Float16*>(table_any), N, ld, q_fixed, q_rel, w.key_true,
                                                                    n, n, filter, w.acc, kStreamAccSlots, counts, stream, batch);
                return filter_finalize_impl<MODEL, D, __bf16>(static_cast<const __bf16*>(table_any), N, ld, q_fixed, q_rel, w.key_true, n, n,
                                                              filter, w.acc, kStreamAccSlots, counts, stream, batch);
            }
            finalize_counts_kernel<<<(int)((Q + kSweepQueries - 1) / kSweepQueries), 256, 0, stream>>>(w.acc, kStreamAccSlots, Q, counts);
            return hipGetLastError();
        }
    }
    // all passes in one launch of a ring kernel (rank_stream.hip: StreamPasses), or a launch per pass
    const bool one_launch = knob(KNOB_STREAM_KERNEL) != 2 && rank_stream_takes_passes(MODEL, D, N, ld, batch, n);
    const bool dot_keys = rank_stream_wants_dot(MODEL, D, N, ld, batch, batch, one_launch);
    const int64_t dot_blocks = dot_keys ? (Q + 7) / 8 : 0;
    if (dot_keys) coef_blocks = 0;  // the approximate-key kernel reads operand rows, not coefficient rows
    const int acc_slots = one_launch ? kStreamAccSlots : 1;
    prep_passes_kernel<MODEL, D><<<dim3((unsigned)(key_blocks + coef_blocks + dot_blocks)), 64, 0, stream>>>(
        q_fixed, q_rel, q_true, n, batch, w.coef, w.key_true, w.acc, (unsigned)key_blocks, (unsigned)dot_blocks, w.wq, w.band, acc_slots);
    if (one_launch) {
        StreamPasses passes;
        passes.n_passes = (int)((n + batch - 1) / batch);
        passes.batch = (int)batch;
        passes.n = n;
        passes.acc_slots = acc_slots;
        StreamDot dot;
        if (dot_keys) { dot.wq = w.wq; dot.band = w.band; dot.q_fixed = q_fixed; dot.q_rel = q_rel; dot.q0 = 0; }
        if (ev_start) (void)hipEventRecord(ev_start, stream);
        const hipError_t err = launch_rank_stream(MODEL, D, table, N, ld, w.coef, w.coef, w.key_true, batch, batch, w.acc, dot, n_cu, stream, passes);
        if (err != hipSuccess) return err;
        if (ev_stop) (void)hipEventRecord(ev_stop, stream);
    } else {
        for (int64_t first = 0; first < n; first += batch) {
            const int nb = (int)(n - first < batch ? n - first : batch);
            const float* coef_head = w.coef + first * (SH::C + ST::C);
            StreamDot dot;
            if (dot_keys) {
                dot.wq = w.wq + (first / batch) * dot_rows_floats(D);
                dot.band = w.band + 2 * 2 * first;
                dot.q_fixed = q_fixed; dot.q_rel = q_rel; dot.q0 = 2 * first;
            }
            if (first == 0 && ev_start) (void)hipEventRecord(ev_start, stream);
            const hipError_t err = launch_static_pass<MODEL, D>(table, N, ld, coef_head, coef_head + (size_t)nb * SH::C, w.key_true + 2 * first,
                                                                nb, nb, w.acc + 2 * first, dot, n_cu, stream);
            if (err != hipSuccess) return err;
            if (first == 0 && ev_stop) (void)hipEventRecord(ev_stop, stream);
        }
    }
    if (filter.on())
        return filter_finalize_impl<MODEL, D>(table, N, ld, q_fixed, q_rel, w.key_true, n, n, filter, w.acc, acc_slots, counts, stream, batch);
    finalize_counts_kernel<<<(int)((Q + kSweepQueries - 1) / kSweepQueries), 256, 0, stream>>>(w.acc, acc_slots, Q, counts);
    return hipGetLastError();
}

hipError_t launch_rank_static_passes(int model, int D, const void* table, int dtype, int64_t N, int64_t ld, const QRows q_fixed,
                                     const QRows q_rel, const QRows q_true, int64_t n, int64_t batch, const FilterSpec& filter,
                                     int32_t* counts, void* workspace, int n_cu, hipStream_t stream, hipEvent_t ev_start,
                                     hipEvent_t ev_stop) {
    BLP_DISPATCH_MODEL_DIM(static_passes_impl, table, dtype, N, ld, q_fixed, q_rel, q_true, n, batch, filter, counts, workspace, n_cu, stream,
                           ev_start, ev_stop)
}

hipError_t launch_rank_metrics(const int32_t* counts, int64_t Q, const int32_t* k, float* rr,
                               uint8_t* hits, hipStream_t stream) {
    if (Q == 0) return hipSuccess;
    rank_metrics_kernel<<<(int)((Q + 255) / 256), 256, 0, stream>>>(counts, Q, k[0], k[1], k[2], rr, hits);
    return hipGetLastError();
}

__global__ __launch_bounds__(64) void rank_metric_sums_finish_kernel(const double* __restrict__ partial, int n,
                                                                     double* __restrict__ sums) {
    if (threadIdx.x < 8) {
        double total = 0.0;
        for (int b = 0; b < n; ++b) total += partial[8 * b + threadIdx.x];
        sums[threadIdx.x] = total;
    }
}

// sums: BLP_METRIC_SUMS_DOUBLES doubles -- the 8 results, then scratch for the per-block partial sums
hipError_t launch_rank_metric_sums(const int32_t* counts, int64_t Q, const int32_t* k, double* sums,
                                   hipStream_t stream) {
    constexpr int64_t kMaxBlocks = (BLP_METRIC_SUMS_DOUBLES - 8) / 8, kMinPerBlock = 8192;
    int64_t blocks = (Q + kMinPerBlock - 1) / kMinPerBlock;
    blocks = blocks < 1 ? 1 : blocks > kMaxBlocks ? kMaxBlocks : blocks;
    const int64_t per_block = blocks > 1 ? (Q + blocks - 1) / blocks : (Q > 0 ? Q : 1);
    if (blocks == 1) {
        rank_metric_sums_kernel<<<1, 1024, 0, stream>>>(counts, Q, per_block, k[0], k[1], k[2], sums);
    } else {
        rank_metric_sums_kernel<<<dim3((unsigned)blocks), 1024, 0, stream>>>(counts, Q, per_block, k[0], k[1], k[2], sums + 8);
        rank_metric_sums_finish_kernel<<<1, 64, 0, stream>>>(sums + 8, (int)blocks, sums);
    }
    return hipGetLastError();
}

}  // namespace blp
