// rank_stream16.hip -- the ring kernels of rank_stream.hip over a 16-BIT candidate table (IEEE half or bfloat16: the copy the
// table build can emit next to the f32 table, SURVEY 8f row 2 / table_elem.h).  The few-queries passes over a long table
// (the reference's Wikidata5M evaluation batch: eval_batch_size 2 = 4 queries per pass, scripts/blp-*-wikidata5m.sh:18;
// train.py:128-171) are one read of the table, HBM-bound: half the bytes per row is half the pass.  Nothing else changes:
// every element is widened to f32 -- exactly -- before it is used, the TransE keys are Scorer<TRANSE, SIDE, D>::score<false>'s
// operations in its order on the widened row, the bilinear models' approximate keys are decided within the same band and the
// undecided rows re-scored by coop_score on the widened row: the counts are the oracle's on the widened table, bit for bit.
//
// Layout: a row-piece is still one 128-byte line -- now 64 columns -- so the loads (8 rows x 128 B per instruction, a ring of two
// pieces refilled two steps ahead across tile and pass boundaries), the transposing LDS slab and its conflict-free reads are
// rank_stream.hip's word for word; a lane ends up with the 32 words of its row-piece and consumes them in HALVES of 32 columns
// (16 words read back from the slab at a time: the ring, a whole piece and its widened columns do not fit 128 registers), a
// half in two chunks of 16 columns: a chunk is widened ONCE (16 conversions) and then walked by every query of the pass -- the
// coefficient units (16 columns of one query, requested by hand one unit ahead) therefore run chunk-major here, query-major in
// the f32 kernel; a query's own chain is left to right in both.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "knobs.h"
#include "launch.h"
#include "rank_common.h"
#include "stream_common.h"

#pragma clang fp contract(off)

namespace blp {

// 16 columns (8 words of the lane's row) widened
template <class T>
__device__ __forceinline__ void widen_chunk(float (&w)[16], const unsigned (&xw)[16], int c) {
#pragma unroll
    for (int i = 0; i < 8; ++i) widen_pair<T>(xw[8 * c + i], w[2 * i], w[2 * i + 1]);
}

// The units of one HALF piece (32 columns; HP = its index in the row, 0 .. D / 32 - 1): chunk c = 0, 1, and per chunk the NQ
// tail-side queries, then the NQ head-side queries.  The unit after a half's last is the next half's first, after a row's last
// the first of the same coefficient rows again (the next tile; a new pass reloads in enter()).  See rank_stream.hip:
// transe_units for the request / drain discipline and for why a side with fewer queries repeats its last row.
template <int D, int NQ, class T, int HP, int U>
__device__ __forceinline__ void transe_units16(float (&sum)[2 * NQ], float (&w)[16], const unsigned (&xw)[16],
                                               const float* const (&row)[2 * NQ], sf16& cur_a, sf16& cur_b) {
    if constexpr (U < 4 * NQ) {
        constexpr int NH = D / 32;
        constexpr int c = U / (2 * NQ), v = U % (2 * NQ);
        constexpr bool head = v >= NQ;
        constexpr int q = head ? v - NQ : v, slot = head ? q : NQ + q;  // sum[0 .. NQ): head, [NQ .. 2 NQ): tail
        constexpr int nu = (U + 1) % (4 * NQ), nhp = U + 1 < 4 * NQ ? HP : (HP + 1) % NH;  // the unit after this one
        constexpr int nc = nu / (2 * NQ), nv = nu % (2 * NQ);
        constexpr bool nhead = nv >= NQ;
        constexpr int nq = nhead ? nv - NQ : nv, nslot = nhead ? nq : NQ + nq, noff = (nhp * 32 + 16 * nc) * 4;
        if constexpr (v == 0) widen_chunk<T>(w, xw, c);
        sf16 nxt_a, nxt_b;
        sload16_pinned<noff>(nxt_a, row[nslot], sum[slot]);
        if constexpr (nhead) sload16_pinned<noff + D * 4>(nxt_b, row[nslot], sum[slot]);
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            float d;
            if constexpr (!head) {
                d = cur_a[k] - w[k];              // (h + r) - e, h + r hoisted
            } else {
                const float y = w[k] + cur_a[k];  // (e + r) - t
                d = y - cur_b[k];
            }
            sum[slot] = sum[slot] + fabsf(d);
        }
        if constexpr (nhead) sdrain_pinned2(nxt_a, nxt_b, sum[slot]);
        else sdrain_pinned(nxt_a, sum[slot]);
        cur_a = nxt_a;
        if constexpr (nhead) cur_b = nxt_b;
        transe_units16<D, NQ, T, HP, U + 1>(sum, w, xw, row, cur_a, cur_b);
    }
}

// what the two ring kernels share: this lane's byte offsets into a tile of a 16-bit table, and a half piece read back
struct Ring16 {
    int sub_row, lane;
    __device__ __forceinline__ void offsets(unsigned (&boff)[8], int64_t N, int64_t ld, int t) const {
        const int64_t left = N - (int64_t)t * kTileRows;
        const int last = left < kTileRows ? (int)left - 1 : kTileRows - 1;  // wave-uniform
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int r = 8 * i + sub_row;
            boff[i] = (unsigned)((r < last ? r : last) * (int)ld * 2 + (lane & 7) * 16);
        }
    }
};
__device__ __forceinline__ void read_half(unsigned (&xw)[16], const float* rd, int half) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint4 v = *reinterpret_cast<const uint4*>(rd + 16 * half + 4 * j);
        xw[4 * j] = v.x; xw[4 * j + 1] = v.y; xw[4 * j + 2] = v.z; xw[4 * j + 3] = v.w;
    }
}

// (4 + 4 queries: the eight sums, the widened chunk and the ring do not fit 128 registers -- three waves per SIMD there; such
//  a pass is bound by its arithmetic, not by the table read)
template <int D, int NQ, class T>
__global__ __launch_bounds__(kWaves * 64, NQ <= 3 ? 4 : 3) void rank_stream16_kernel(
    const void* __restrict__ table, int64_t N, int64_t ld, const float* __restrict__ coef,
    const float* __restrict__ key_true, int q_head, int q_tail, int n_tiles,
    unsigned long long* __restrict__ acc, const StreamPasses passes) {
    constexpr int NP = D / 64;  // pieces (128-byte row parts) per tile
    static_assert(NP % 2 == 0, "the ring of two pieces assumes an even number of pieces per tile");
    __shared__ __attribute__((aligned(16))) float slabs[kWaves * kSlabFloats];
    __shared__ unsigned long long wg_cnt[2 * kStreamQ];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* slab = slabs + wave * kSlabFloats;
    const Ring16 rg{lane >> 3, lane};
    float* wr = slab + rg.sub_row * kLdsStride + (lane & 7) * 4;
    const float* rd = slab + lane * kLdsStride;
    if (tid < 2 * kStreamQ) wg_cnt[tid] = 0;

    // rounds of four consecutive tiles (one per wave) over all passes of the launch: rank_stream.hip: rank_stream_kernel
    const unsigned n_rounds = (unsigned)(n_tiles + kWaves - 1) / kWaves, total = n_rounds * (unsigned)passes.n_passes;
    const unsigned long long slot_off = passes.acc_slots > 1 ? (unsigned long long)(blockIdx.x % passes.acc_slots) * (2ull * passes.n) : 0ull;
    auto tile_of = [&](unsigned r) { return (int)((r % n_rounds) * kWaves) + wave; };  // may be >= n_tiles
    unsigned boff[8];
    auto tile_base = [&](int t) { return static_cast<const char*>(table) + (int64_t)t * kTileRows * ld * 2; };
    auto clamp_tile = [&](int t) { return t < n_tiles ? t : n_tiles - 1; };

    static_assert(NQ >= 1 && NQ <= kStreamQ, "queries per side");
    unsigned n_gt[2 * kStreamQ] = {}, n_ge[2 * kStreamQ] = {};  // wave-uniform: scalar registers
    int cur_p = -1, qh = 0, qt = 0;
    const float *ch = coef, *ct = coef, *kt = key_true;
    unsigned long long* acc_p = acc;
    const float* row[2 * NQ];
    sf16 cur_a, cur_b;
    auto enter = [&](int p) {
        const PassView v = pass_view(passes, p, q_head, q_tail);
        qh = v.q_head; qt = v.q_tail;
        ch = coef + (v.first2 / 2) * (3 * D);
        ct = ch + (size_t)qh * (2 * D);
        kt = key_true + v.first2;
        acc_p = acc + slot_off + v.first2;
        cur_p = p;
        static_for<NQ>([&](auto jj) {
            constexpr int j = decltype(jj)::value;
            row[j] = qh > 0 ? ch + (size_t)(j < qh ? j : qh - 1) * (2 * D) : ct;
            row[NQ + j] = qt > 0 ? ct + (size_t)(j < qt ? j : qt - 1) * D : ch;
        });
        cur_a = sload16<0>(row[NQ]);  // unit 0 of half 0: the first tail-side row's columns 0 .. 15
        sdrain(cur_a);
    };
    auto flush_wave = [&]() {
        if (lane < 2 * kStreamQ) {
            unsigned gt = 0, ge = 0;
            static_for<2 * kStreamQ>([&](auto jj) {
                constexpr int j = decltype(jj)::value;
                if (lane == j) { gt = n_gt[j]; ge = n_ge[j]; }
            });
            const int side_q = lane < kStreamQ ? lane : lane - kStreamQ;
            const bool live = lane < kStreamQ ? side_q < qh : side_q < qt;
            const unsigned long long v = (unsigned long long)gt | ((unsigned long long)ge << 32);
            if (live && v) atomicAdd(acc_p + (lane < kStreamQ ? side_q : qh + side_q), v);
        }
        static_for<2 * kStreamQ>([&](auto jj) { n_gt[decltype(jj)::value] = 0; n_ge[decltype(jj)::value] = 0; });
    };

    unsigned r = blockIdx.x;
    f32x4 ring[2][8];
    if (r < total) {
        const int t0 = clamp_tile(tile_of(r));
        rg.offsets(boff, N, ld, t0);
        piece_fetch(ring[0], tile_base(t0), boff, 0);
        piece_fetch(ring[1], tile_base(t0), boff, 1);
        enter((int)(r / n_rounds));
    }
    for (; r < total; r += gridDim.x) {
        const int p = (int)(r / n_rounds);
        if (p != cur_p) {  // wave-uniform
            flush_wave();
            enter(p);
        }
        const int tile_real = tile_of(r), tile = clamp_tile(tile_real);
        const unsigned next_r = r + gridDim.x;
        const bool more = next_r < total;  // wave-uniform
        const int next = more ? clamp_tile(tile_of(next_r)) : 0;
        const char* base = tile_base(tile);
        float sum[2 * NQ] = {};
        float w[16];
        static_for<NP>([&](auto ss) {
            constexpr int s = decltype(ss)::value, pp = s & 1;
#pragma unroll
            for (int i = 0; i < 8; ++i)
                *reinterpret_cast<f32x4*>(wr + 8 * i * kLdsStride) = ring[pp][i];
            if constexpr (s + 2 < NP) {
                piece_fetch(ring[pp], base, boff, s + 2);
            } else {
                if constexpr (s + 2 == NP) {
                    if (more) rg.offsets(boff, N, ld, next);
                }
                if (more) piece_fetch(ring[pp], tile_base(next), boff, s + 2 - NP);
            }
            wave_lds_sync();
            unsigned xw[16];
            read_half(xw, rd, 0);
            transe_units16<D, NQ, T, 2 * s, 0>(sum, w, xw, row, cur_a, cur_b);
            read_half(xw, rd, 1);
            wave_lds_sync();  // the reads are out before the next piece overwrites the slab
            transe_units16<D, NQ, T, 2 * s + 1, 0>(sum, w, xw, row, cur_a, cur_b);
        });
        const bool valid = tile_real < n_tiles && (int64_t)tile * kTileRows + lane < N;
        static_for<NQ>([&](auto jj) {
            constexpr int j = decltype(jj)::value;
            if (j < qh) {
                const float key = -sum[j], k_true = kt[j];
                n_gt[j] += __popcll(__ballot(valid && key > k_true));
                n_ge[j] += __popcll(__ballot(valid && key >= k_true));
            }
            if (j < qt) {
                const float key = -sum[NQ + j], k_true = kt[qh + j];
                n_gt[kStreamQ + j] += __popcll(__ballot(valid && key > k_true));
                n_ge[kStreamQ + j] += __popcll(__ballot(valid && key >= k_true));
            }
        });
    }

    __syncthreads();  // wg_cnt is zero
    if (lane < 2 * kStreamQ) {
        unsigned gt = 0, ge = 0;
        static_for<2 * kStreamQ>([&](auto jj) {
            constexpr int j = decltype(jj)::value;
            if (lane == j) { gt = n_gt[j]; ge = n_ge[j]; }
        });
        const unsigned long long v = (unsigned long long)gt | ((unsigned long long)ge << 32);
        if (v) atomicAdd(&wg_cnt[lane], v);
    }
    __syncthreads();
    if (tid < 2 * kStreamQ && cur_p >= 0) {
        const int side_q = tid < kStreamQ ? tid : tid - kStreamQ;
        const bool live = tid < kStreamQ ? side_q < qh : side_q < qt;
        const unsigned long long v = wg_cnt[tid];
        if (live && v) atomicAdd(acc_p + (tid < kStreamQ ? side_q : qh + side_q), v);
    }
}

// ---- the bilinear models: approximate keys (rank_stream.hip: rank_stream_dot_kernel) over a 16-bit table ----------------
// Units of a half piece: chunk c = 0, 1, and per chunk the pass's NQ queries; the squared norm rides with query 0's unit.
template <int D, int NQ, class T, int HP, int U>
__device__ __forceinline__ void dot_units16(float (&sum)[NQ], float& ssq, float (&w)[16], const unsigned (&xw)[16], const float* wq,
                                            sf16& cur) {
    if constexpr (U < 2 * NQ) {
        constexpr int NH = D / 32;
        constexpr int c = U / NQ, q = U % NQ;
        constexpr int nu = (U + 1) % (2 * NQ), nhp = U + 1 < 2 * NQ ? HP : (HP + 1) % NH;  // the unit after this one
        constexpr int nc = nu / NQ, nq = nu % NQ;
        if constexpr (q == 0) widen_chunk<T>(w, xw, c);
        sf16 nxt;
        sload16_pinned<(stream_dot_row(nq) * D + nhp * 32 + 16 * nc) * 4>(nxt, wq, sum[q]);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            sum[q] = __builtin_fmaf(w[i], cur[i], sum[q]);
            if constexpr (q == 0) ssq = __builtin_fmaf(w[i], w[i], ssq);
        }
        if constexpr (q == 0) sdrain_pinned(nxt, sum[q], ssq);
        else sdrain_pinned(nxt, sum[q]);
        cur = nxt;
        dot_units16<D, NQ, T, HP, U + 1>(sum, ssq, w, xw, wq, cur);
    }
}

template <int MODEL, int D, int NQ, class T>
__global__ __launch_bounds__(kWaves * 64, NQ <= 4 ? 4 : 3) void rank_stream_dot16_kernel(
    const T* __restrict__ table, int64_t N, int64_t ld, const float* __restrict__ wq, const float* __restrict__ band,
    const float* __restrict__ key_true, const QRows q_fixed, const QRows q_rel, int64_t q0, int q_head, int q_tail,
    int n_tiles, unsigned long long* __restrict__ acc, const StreamPasses passes) {
    constexpr int NP = D / 64;
    static_assert(NP % 2 == 0, "the ring of two pieces assumes an even number of pieces per tile");
    static_assert(NQ == 4 || NQ == 2 * kStreamQ, "a pass of up to 4, or up to 4 + 4, queries");
    __shared__ __attribute__((aligned(16))) float slabs[kWaves * kSlabFloats];
    __shared__ unsigned long long wg_cnt[NQ];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* slab = slabs + wave * kSlabFloats;
    const Ring16 rg{lane >> 3, lane};
    float* wr = slab + rg.sub_row * kLdsStride + (lane & 7) * 4;
    const float* rd = slab + lane * kLdsStride;
    if (tid < NQ) wg_cnt[tid] = 0;

    const unsigned n_rounds = (unsigned)(n_tiles + kWaves - 1) / kWaves, total = n_rounds * (unsigned)passes.n_passes;
    const unsigned long long slot_off = passes.acc_slots > 1 ? (unsigned long long)(blockIdx.x % passes.acc_slots) * (2ull * passes.n) : 0ull;
    auto tile_of = [&](unsigned r) { return (int)((r % n_rounds) * kWaves) + wave; };  // may be >= n_tiles
    unsigned boff[8];
    auto tile_base = [&](int t) { return reinterpret_cast<const char*>(table) + (int64_t)t * kTileRows * ld * 2; };
    auto clamp_tile = [&](int t) { return t < n_tiles ? t : n_tiles - 1; };

    unsigned n_gt[NQ] = {};           // certainly above the true key (gt and ge alike): wave-uniform, scalar registers
    unsigned long long ex_cnt = 0;    // lane q: query q's undecided rows at or above it, gt | ge << 32
    int cur_p = -1, qh = 0, Q = 0;
    const float *wq_p = wq, *band_p = band, *kt = key_true;
    int64_t q0_p = q0;
    unsigned long long* acc_p = acc;
    sf16 cur;
    auto enter = [&](int p) {
        const PassView v = pass_view(passes, p, q_head, q_tail);
        qh = v.q_head; Q = v.q_head + v.q_tail;
        if (passes.n_passes > 1) {
            wq_p = wq + (size_t)p * (kStreamDotRows * D);
            band_p = band + 2 * v.first2;
            kt = key_true + v.first2;
            q0_p = q0 + v.first2;
            acc_p = acc + slot_off + v.first2;
        }
        cur = sload16<stream_dot_row(0) * D * 4>(wq_p);  // unit 0 of half 0 of this pass's operands
        sdrain(cur);
        cur_p = p;
    };
    auto flush_wave = [&]() {
        if (lane < NQ) {
            unsigned above = 0;
            static_for<NQ>([&](auto jj) {
                constexpr int j = decltype(jj)::value;
                if (lane == j) above = n_gt[j];
            });
            const unsigned long long v = ((unsigned long long)above | ((unsigned long long)above << 32)) + ex_cnt;
            if (lane < Q && v) atomicAdd(acc_p + lane, v);
        }
        ex_cnt = 0;
        static_for<NQ>([&](auto jj) { n_gt[decltype(jj)::value] = 0; });
    };

    unsigned r = blockIdx.x;
    f32x4 ring[2][8];
    if (r < total) {
        const int t0 = clamp_tile(tile_of(r));
        rg.offsets(boff, N, ld, t0);
        piece_fetch(ring[0], tile_base(t0), boff, 0);
        piece_fetch(ring[1], tile_base(t0), boff, 1);
        enter((int)(r / n_rounds));
    }
    for (; r < total; r += gridDim.x) {
        const int p = (int)(r / n_rounds);
        if (p != cur_p) {  // wave-uniform
            flush_wave();
            enter(p);
        }
        const int tile_real = tile_of(r), tile = clamp_tile(tile_real);
        const unsigned next_r = r + gridDim.x;
        const bool more = next_r < total;  // wave-uniform
        const int next = more ? clamp_tile(tile_of(next_r)) : 0;
        const char* base = tile_base(tile);
        float sum[NQ] = {}, ssq = 0.f;
        float w[16];
        static_for<NP>([&](auto ss) {
            constexpr int s = decltype(ss)::value, pp = s & 1;
#pragma unroll
            for (int i = 0; i < 8; ++i)
                *reinterpret_cast<f32x4*>(wr + 8 * i * kLdsStride) = ring[pp][i];
            if constexpr (s + 2 < NP) {
                piece_fetch(ring[pp], base, boff, s + 2);
            } else {
                if constexpr (s + 2 == NP) {
                    if (more) rg.offsets(boff, N, ld, next);
                }
                if (more) piece_fetch(ring[pp], tile_base(next), boff, s + 2 - NP);
            }
            wave_lds_sync();
            unsigned xw[16];
            read_half(xw, rd, 0);
            dot_units16<D, NQ, T, 2 * s, 0>(sum, ssq, w, xw, wq_p, cur);
            read_half(xw, rd, 1);
            wave_lds_sync();  // the reads are out before the next piece overwrites the slab
            dot_units16<D, NQ, T, 2 * s + 1, 0>(sum, ssq, w, xw, wq_p, cur);
        });
        const bool valid = tile_real < n_tiles && (int64_t)tile * kTileRows + lane < N;
        const float nrow = sqrtf(ssq) * 1.0001f;
        const bool tiny = ssq < 1e-30f;
        unsigned und_lo = 0, und_hi = 0;
        bool any_und = false;
        static_for<NQ>([&](auto qq) {
            constexpr int q = decltype(qq)::value;
            if (q < Q) {
                const DotBand b = dot_band_of(kt, band_p, q);
                bool gt, lt;
                dot_decide(sum[q], b, nrow, tiny, gt, lt);
                const unsigned long long above = __ballot(valid && gt), und = __ballot(valid && !(gt || lt));
                n_gt[q] += __popcll(above);
                if (lane == q) { und_lo = (unsigned)und; und_hi = (unsigned)(und >> 32); }
                any_und |= und != 0;
            }
        });
        if (any_und) {  // wave-uniform, rare
            for (int q = 0; q < Q; ++q) {
                const unsigned long long und = (unsigned long long)(unsigned)__builtin_amdgcn_readlane(und_lo, q) |
                                               ((unsigned long long)(unsigned)__builtin_amdgcn_readlane(und_hi, q) << 32);
                if (!und) continue;
                const unsigned long long c = exact_undecided<MODEL, D, T>(table, ld, (int64_t)tile * kTileRows, q_fixed.row(q0_p + q),
                                                                          q_rel.row(q0_p + q), q < qh, und, kt[q], lane);
                if (lane == q) ex_cnt += c;
            }
        }
    }

    __syncthreads();  // wg_cnt is zero
    if (lane < NQ) {
        unsigned above = 0;
        static_for<NQ>([&](auto jj) {
            constexpr int j = decltype(jj)::value;
            if (lane == j) above = n_gt[j];
        });
        const unsigned long long v = ((unsigned long long)above | ((unsigned long long)above << 32)) + ex_cnt;
        if (v) atomicAdd(&wg_cnt[lane], v);
    }
    __syncthreads();
    if (tid < Q && cur_p >= 0) {
        const unsigned long long v = wg_cnt[tid];
        if (v) atomicAdd(acc_p + tid, v);
    }
}

// A pass (or all passes of a reference-batched call) of <= 4 + 4 queries over a 16-bit table: D = 128 or 256 (a 128-byte
// row-piece is 64 columns, two pieces in the ring), 16-byte aligned rows, 32-bit byte offsets inside a tile.
bool rank_stream16_applicable(int model, int D, int64_t N, int64_t ld, int64_t q_head, int64_t q_tail) {
    // (both sides non-empty: with q_head == 0 the TransE kernel's head units would still request coefficient rows past the tail
    //  rows' -- every caller ranks a batch's head AND tail queries, so the one-sided case goes to the other kernels)
    if (q_head > kStreamQ || q_tail > kStreamQ || N <= 0 || q_head <= 0 || q_tail <= 0) return false;
    if (!(D == 128 || D == 256) || ld % 8 != 0 || ld >= (1 << 22)) return false;
    if (model != TRANSE && model != DISTMULT && model != COMPLEX && model != SIMPLE) return false;
    return ((N + kTileRows - 1) / kTileRows + kWaves - 1) / kWaves < (int64_t)0x7fffffff;  // (the kernels' 32-bit round index)
}
bool rank_stream16_takes_passes(int model, int D, int64_t N, int64_t ld, int64_t batch, int64_t n) {
    if (!rank_stream16_applicable(model, D, N, ld, batch, batch)) return false;
    const int64_t n_rounds = ((N + kTileRows - 1) / kTileRows + kWaves - 1) / kWaves, n_passes = (n + batch - 1) / batch;
    return n_passes >= 1 && n_rounds * n_passes < (int64_t)0x7fffffff;
}

template <class T>
static hipError_t launch_stream16_typed(int model, int D, const T* table, int64_t N, int64_t ld, const float* coef,
                                        const float* key_true, int64_t q_head, int64_t q_tail, unsigned long long* acc,
                                        const StreamDot& dot, int n_cu, hipStream_t stream, const StreamPasses& passes) {
    const int64_t n_tiles = (N + kTileRows - 1) / kTileRows;
    const int64_t n_rounds = ((n_tiles + kWaves - 1) / kWaves) * (passes.n_passes > 1 ? passes.n_passes : 1);
    if (model != TRANSE) {
        if (dot.wq == nullptr) return hipErrorInvalidValue;  // the bilinear models come with approximate keys here
        const int per_pass = passes.n_passes > 1 ? 2 * passes.batch : (int)(q_head + q_tail);
#define BLP_STREAM_DOT16(MM, DD)                                                                                          \
    if (model == MM && D == DD) {                                                                                        \
        if (per_pass <= 4) {                                                                                             \
            const int64_t resident = (int64_t)n_cu * 4;                                                                  \
            rank_stream_dot16_kernel<MM, DD, 4, T><<<(unsigned)(n_rounds < resident ? n_rounds : resident), kWaves * 64, 0, stream>>>( \
                table, N, ld, dot.wq, dot.band, key_true, dot.q_fixed, dot.q_rel, dot.q0, (int)q_head, (int)q_tail, (int)n_tiles, acc, passes); \
        } else {                                                                                                         \
            const int64_t resident = (int64_t)n_cu * 3;                                                                  \
            rank_stream_dot16_kernel<MM, DD, 2 * kStreamQ, T><<<(unsigned)(n_rounds < resident ? n_rounds : resident), kWaves * 64, 0, stream>>>( \
                table, N, ld, dot.wq, dot.band, key_true, dot.q_fixed, dot.q_rel, dot.q0, (int)q_head, (int)q_tail, (int)n_tiles, acc, passes); \
        }                                                                                                                \
        return hipGetLastError();                                                                                        \
    }
        BLP_STREAM_DOT16(DISTMULT, 128) BLP_STREAM_DOT16(DISTMULT, 256) BLP_STREAM_DOT16(COMPLEX, 128) BLP_STREAM_DOT16(COMPLEX, 256)
        BLP_STREAM_DOT16(SIMPLE, 128) BLP_STREAM_DOT16(SIMPLE, 256)
#undef BLP_STREAM_DOT16
        return hipErrorInvalidValue;
    }
    const int64_t per_side = passes.n_passes > 1 ? passes.batch : (q_head > q_tail ? q_head : q_tail);
    const int64_t resident = (int64_t)n_cu * (per_side <= 3 ? 4 : 3);  // workgroups of four waves per CU = waves per SIMD
    const unsigned blocks = (unsigned)(n_rounds < resident ? n_rounds : resident);
#define BLP_STREAM16_CASE(DD, QQ)                                                                                          \
    case DD * 8 + QQ:                                                                                                    \
        rank_stream16_kernel<DD, QQ, T><<<blocks, kWaves * 64, 0, stream>>>(table, N, ld, coef, key_true, (int)q_head,    \
                                                                            (int)q_tail, (int)n_tiles, acc, passes);     \
        break;
    switch (D * 8 + (int)per_side) {
        BLP_STREAM16_CASE(128, 1) BLP_STREAM16_CASE(128, 2) BLP_STREAM16_CASE(128, 3) BLP_STREAM16_CASE(128, 4)
        BLP_STREAM16_CASE(256, 1) BLP_STREAM16_CASE(256, 2) BLP_STREAM16_CASE(256, 3) BLP_STREAM16_CASE(256, 4)
    default: return hipErrorInvalidValue;
    }
#undef BLP_STREAM16_CASE
    return hipGetLastError();
}

hipError_t launch_rank_stream16(int model, int D, int dtype, const void* table, int64_t N, int64_t ld, const float* coef,
                                const float* key_true, int64_t q_head, int64_t q_tail, unsigned long long* acc,
                                const StreamDot& dot, int n_cu, hipStream_t stream, const StreamPasses& passes) {
    if (!rank_stream16_applicable(model, D, N, ld, passes.n_passes > 1 ? passes.batch : q_head,
                                  passes.n_passes > 1 ? passes.batch : q_tail))
        return hipErrorInvalidValue;
    if (dtype == kTableF16)
        return launch_stream16_typed(model, D, static_cast<const _Float16*>(table), N, ld, coef, key_true, q_head, q_tail, acc, dot,
                                     n_cu, stream, passes);
    if (dtype == kTableBF16)
        return launch_stream16_typed(model, D, static_cast<const __bf16*>(table), N, ld, coef, key_true, q_head, q_tail, acc, dot,
                                     n_cu, stream, passes);
    return hipErrorInvalidValue;
}

}  // namespace blp
