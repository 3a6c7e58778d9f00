// rank_stream.hip -- a handful of queries against a LONG table (the reference's Wikidata5M evaluation batch:
// eval_batch_size 2 = 4 queries per pass over 4.6 M rows, scripts/blp-*-wikidata5m.sh:18; train.py:128-171): the
// pass is one read of the table, HBM-bound, and what matters is that the read never stops.  Three kernels:
//   rank_stream_kernel       TransE, a per-wave ring of 32-column pieces (single calls on tables of up to 1.7 M rows; every
//                            reference-batched call: all its passes in one launch, StreamPasses);
//   rank_stream_wg_kernel    order-exact keys from a workgroup's double-buffered LDS tile (single calls: TransE on longer
//                            tables, DistMult / SimplE on tables of 1.7 M rows and more; the A/B reference of the tests);
//   rank_stream_dot_kernel   the bilinear models on the ring: approximate keys (a chain of fused multiply-adds) decided within
//                            a provable band, undecided rows re-scored in the reference's order on the spot (single calls
//                            below 1.7 M rows, ComplEx and D = 256 at any length; every reference-batched call).
//
// rank_tiles<STATIC> (rank_all.hip) fetches a whole tile (64 rows x D floats, all D/4 loads of a lane at once), then
// scores it: a wave alternates between a load phase and an arithmetic phase and only the drift between the waves of a
// SIMD overlaps the two.  On the full table the waves have 23 tiles to drift apart (6.0 TB/s); on a 1/8 shard of it --
// what a rank of an 8-GPU evaluation holds -- they have three, and the chip runs load phase / arithmetic phase in lock-step
// (3.8 TB/s: [measured] 77 us for 294 MB).
//
// TransE: a wave consumes its tile 32 columns at a time -- TransE's sum runs strictly left to right over the row, so the
// partial sums of the <= 4 + 4 queries are all that crosses a step -- from a ring of two 32-column pieces (2 x 8 loads of
// 16 B per lane) that is refilled two steps ahead, ACROSS tile boundaries: every wave has 8-16 KB in flight at all
// times and the arithmetic of a step runs under the loads of the next two.  64 + 32 + 8 live data registers instead of D
// + 40: four waves per SIMD.  Keys: the operations of Scorer<TRANSE, SIDE, D>::score<false> in its order (score_core.h),
// coefficients as SGPR operands from the scalar cache as in rank_tiles<STATIC>: counts are bit-identical.
// Counts: scalar registers per (wave, query), summed over the workgroup's waves in LDS, one 64-bit atomic per
// (workgroup, query) at the end.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "dot_band.h"
#include "exact_coop.h"
#include "stream_common.h"
#include "knobs.h"
#include "launch.h"
#include "rank_common.h"
#include "score_core.h"
#include "tile.h"

#pragma clang fp contract(off)

namespace blp {

// The coefficient rows come through the scalar cache in UNITS of 16 columns of one query (tail side: one s_load_dwordx16 of
// h + r; head side: two, r and t), requested by hand one unit ahead of their use and drained after the arithmetic of the unit
// before -- as the approximate-key kernel further down does (sload16_pinned / sdrain_pinned).  Left to the compiler (rounds
// 2-3) every one of a tile's 32 scalar loads was followed by its wait on the spot: 32 exposed scalar-cache round trips per
// tile and wave, and a pass of 2 + 2 queries ran 2 % behind the bilinear models' ring (346 against 339 us per 4.6 M rows).
// Units of a piece: the NQ tail-side queries' (two each), then the NQ head-side queries'; the unit after a piece's last is the
// next piece's first, after a tile's last the first of the same rows again (the next tile; a new pass reloads in enter()).
// NQ = queries per side the kernel is compiled for: a side with fewer repeats its last row (sums computed, never counted).
// Sums start at 0: 0 + |d| == |d| bit for bit (|d| is +0, positive or NaN), so the chain is score<false>'s.
template <int D, int NQ, int S, int U>
__device__ __forceinline__ void transe_units(float (&sum)[2 * NQ], const float (&x)[kSubCols], const float* const (&row)[2 * NQ],
                                             sf16& cur_a, sf16& cur_b) {
    if constexpr (U < 4 * NQ) {
        constexpr int NP = D / kSubCols;
        constexpr bool head = U >= 2 * NQ;
        constexpr int q = (head ? U - 2 * NQ : U) / 2, c0 = 16 * (U % 2), slot = head ? q : NQ + q;  // sum[0 .. NQ): head, [NQ .. 2 NQ): tail
        constexpr int nu = (U + 1) % (4 * NQ), ns = U + 1 < 4 * NQ ? S : (S + 1) % NP;                // the unit after this one
        constexpr bool nhead = nu >= 2 * NQ;
        constexpr int nq = (nhead ? nu - 2 * NQ : nu) / 2, nslot = nhead ? nq : NQ + nq, noff = (ns * kSubCols + 16 * (nu % 2)) * 4;
        sf16 nxt_a, nxt_b;
        sload16_pinned<noff>(nxt_a, row[nslot], sum[slot]);
        if constexpr (nhead) sload16_pinned<noff + D * 4>(nxt_b, row[nslot], sum[slot]);
#ifdef BLP_STREAM_NULL  // experiment (tools/step_ab.py): the ring with (almost) no arithmetic -- what the access pattern alone streams at
        sum[slot] = sum[slot] + x[c0] * cur_a[0];
#else
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            float d;
            if constexpr (!head) {
                d = cur_a[k] - x[c0 + k];              // (h + r) - e, h + r hoisted
            } else {
                const float y = x[c0 + k] + cur_a[k];  // (e + r) - t
                d = y - cur_b[k];
            }
            sum[slot] = sum[slot] + fabsf(d);
        }
#endif
        if constexpr (nhead) sdrain_pinned2(nxt_a, nxt_b, sum[slot]);
        else sdrain_pinned(nxt_a, sum[slot]);
        cur_a = nxt_a;
        if constexpr (nhead) cur_b = nxt_b;
        transe_units<D, NQ, S, U + 1>(sum, x, row, cur_a, cur_b);
    }
}

template <int D, int NQ>
__global__ __launch_bounds__(kWaves * 64, D == 256 ? 3 : 4) void rank_stream_kernel(
    const float* __restrict__ table, int64_t N, int64_t ld, const float* __restrict__ coef,
    const float* __restrict__ key_true, int q_head, int q_tail, int n_tiles,
    unsigned long long* __restrict__ acc, const StreamPasses passes) {
    constexpr int NP = D / kSubCols;  // pieces per tile
    static_assert(NP % 2 == 0, "the ring of two pieces assumes an even number of pieces per tile");
    __shared__ __attribute__((aligned(16))) float slabs[kWaves * kSlabFloats];
    __shared__ unsigned long long wg_cnt[2 * kStreamQ];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* slab = slabs + wave * kSlabFloats;
    const int sub_row = lane >> 3, sub_col = (lane & 7) * 4;
    float* wr = slab + sub_row * kLdsStride + sub_col;
    const float* rd = slab + lane * kLdsStride;
    if (tid < 2 * kStreamQ) wg_cnt[tid] = 0;

    // rounds of four consecutive tiles (one per wave); round r of the launch = round r % n_rounds of pass r / n_rounds.
    // A wave whose tile of a pass's last round does not exist walks the last tile again, masked when counting.
    const unsigned n_rounds = (unsigned)(n_tiles + kWaves - 1) / kWaves, total = n_rounds * (unsigned)passes.n_passes;
    const unsigned long long slot_off = passes.acc_slots > 1 ? (unsigned long long)(blockIdx.x % passes.acc_slots) * (2ull * passes.n) : 0ull;
    auto tile_of = [&](unsigned r) { return (int)((r % n_rounds) * kWaves) + wave; };  // may be >= n_tiles

    // byte offsets of this lane's parts of a tile's rows (rows past the end of the table: the last row, masked later)
    unsigned boff[8];
    auto offsets = [&](int t) {
        const int64_t left = N - (int64_t)t * kTileRows;
        const int last = left < kTileRows ? (int)left - 1 : kTileRows - 1;  // wave-uniform
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int r = 8 * i + sub_row;
            boff[i] = (unsigned)(((r < last ? r : last) * (int)ld + sub_col) * 4);
#ifdef BLP_STREAM_CONTIG
            boff[i] = (unsigned)((i * 256 + lane * 4) * 4);  // (experiment: full tiles only)
#endif
        }
    };
    auto tile_base = [&](int t) { return table + (int64_t)t * kTileRows * ld; };
    auto clamp_tile = [&](int t) { return t < n_tiles ? t : n_tiles - 1; };

    static_assert(NQ >= 1 && NQ <= kStreamQ, "queries per side");
    unsigned n_gt[2 * kStreamQ] = {}, n_ge[2 * kStreamQ] = {};  // wave-uniform: scalar registers (slots past NQ per side stay 0)
    // the pass the wave is in: its queries' coefficient rows, true keys, accumulators (prep_passes_kernel's layout)
    // (one base pointer for the coefficient rows: the pass's head rows, 2 D floats each, then its tail rows, D each -- with
    //  two, the pass-dependent choice between them costs the compiler the no-alias reasoning that keeps these loads scalar)
    int cur_p = -1, qh = 0, qt = 0;
    const float *ch = coef, *ct = coef, *kt = key_true;
    unsigned long long* acc_p = acc;
    const float* row[2 * NQ];  // the coefficient rows the units walk: [0, NQ) head side, [NQ, 2 NQ) tail side
    sf16 cur_a, cur_b;         // the unit about to be used (cur_b: the t half of a head-side unit)
    auto enter = [&](int p) {
        const PassView v = pass_view(passes, p, q_head, q_tail);
        qh = v.q_head; qt = v.q_tail;
        ch = coef + (v.first2 / 2) * (3 * D);  // per triple of the earlier passes: a head row and a tail row
        ct = ch + (size_t)qh * (2 * D);
        kt = key_true + v.first2;
        acc_p = acc + slot_off + v.first2;
        cur_p = p;
        // a side with fewer than NQ queries repeats its last row; a side with none walks a row of the other side (in bounds:
        // both sides' rows are carved 2 D floats wide, rank_all.hip: carve_workspace / carve_passes) -- never counted
        static_for<NQ>([&](auto jj) {
            constexpr int j = decltype(jj)::value;
            row[j] = qh > 0 ? ch + (size_t)(j < qh ? j : qh - 1) * (2 * D) : ct;
            row[NQ + j] = qt > 0 ? ct + (size_t)(j < qt ? j : qt - 1) * D : ch;
        });
        cur_a = sload16<0>(row[NQ]);  // unit 0 of piece 0: the first tail-side row's columns 0 .. 15
        sdrain(cur_a);
    };
    auto flush_wave = [&]() {  // this wave's counts of pass cur_p -> its accumulators; slot j < 4: head query j; 4 + j: tail query j
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
        offsets(t0);
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
        const float* base = tile_base(tile);
        float sum[2 * NQ] = {};
        static_for<NP>([&](auto ss) {
            constexpr int s = decltype(ss)::value, pp = s & 1;
            // piece s: registers -> slab (transposing) ...
#pragma unroll
            for (int i = 0; i < 8; ++i)
                *reinterpret_cast<f32x4*>(wr + 8 * i * kLdsStride) = ring[pp][i];
            // ... and its ring slot is refilled with the piece two steps ahead: this tile's, or the next tile's
            if constexpr (s + 2 < NP) {
                piece_fetch(ring[pp], base, boff, s + 2);
            } else {
                if constexpr (s + 2 == NP) {
                    if (more) offsets(next);
                }
                if (more) piece_fetch(ring[pp], tile_base(next), boff, s + 2 - NP);
            }
            wave_lds_sync();
            float x[kSubCols];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(rd + 4 * j);
                x[4 * j] = v.x; x[4 * j + 1] = v.y; x[4 * j + 2] = v.z; x[4 * j + 3] = v.w;
            }
            wave_lds_sync();  // the reads are out before the next piece overwrites the slab
            transe_units<D, NQ, s, 0>(sum, x, row, cur_a, cur_b);
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

    // the last pass of the workgroup (all of its waves walk the same rounds): summed over its waves in LDS, one atomic per query
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
    if (tid < 2 * kStreamQ && cur_p >= 0) {  // slot j < 4: head query j; slot 4 + j: tail query j
        const int side_q = tid < kStreamQ ? tid : tid - kStreamQ;
        const bool live = tid < kStreamQ ? side_q < qh : side_q < qt;
        const unsigned long long v = wg_cnt[tid];
        if (live && v) atomicAdd(acc_p + (tid < kStreamQ ? side_q : qh + side_q), v);
    }
}

// ---- the bilinear models ---------------------------------------------------------------------------------------------
// torch.sum's order (score_core.h: 32 running sums per key, every one fed by every 32-column piece of the row) leaves
// nothing small to carry across pieces, so a key needs its whole row at once -- in registers, next to the 32 sums: no
// room for a load ring as well at more than one wave per SIMD.  Here the WORKGROUP owns the tile instead of the wave
// (as in rank_small.hip): its four waves fetch the next tile together (16 B per thread x 8, in flight in 32 registers
// while the current tile is scored), hand it over through a double-buffered LDS tile (row stride D + 4 floats: lane l
// reads its row with conflict-free ds_read_b128), and split the QUERIES -- wave w scores queries w and w + 4 against
// the whole tile with Scorer<>::score, coefficients as SGPR operands (SgprStreamCoef, above).  One barrier per tile; two
// workgroups per CU.
// Coefficients as SGPR operands WITHOUT the scalar-cache latency on the critical path.  Plain scalar loads (PtrCoef) are
// issued by the compiler right where a chunk is first needed and waited for on the spot: 8-16 exposed round trips per
// key, which two waves per SIMD do not hide ([measured] ComplEx, 4 queries, 4.6 M rows: 491 us = 4.8 TB/s with PtrCoef,
// no better than rank_tiles<STATIC>).  Every bilinear Scorer<> walks NS parallel coefficient streams (index I: stream
// I / L, position I % L) front to back, term by term, touching stream 0 first -- so the accessor itself keeps a ring of
// two CH-float chunks per stream in SGPRs: the first touch of chunk k waits for it and requests chunk k + 1.
typedef float sf8 __attribute__((ext_vector_type(8)));
template <int CH> struct SVec;
template <> struct SVec<8> { typedef sf8 type; };
template <> struct SVec<16> { typedef sf16 type; };

template <int OFF>
__device__ __forceinline__ void sload(sf8& v, const float* base) {
    asm volatile("s_load_dwordx8 %0, %1, %2" : "=s"(v) : "s"(base), "i"(OFF) : "memory");
}
template <int OFF>
__device__ __forceinline__ void sload(sf16& v, const float* base) {
    asm volatile("s_load_dwordx16 %0, %1, %2" : "=s"(v) : "s"(base), "i"(OFF) : "memory");
}
template <class V> __device__ __forceinline__ void sdrain_n(V (&b)[1]) { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(b[0]) : : "memory"); }
template <class V> __device__ __forceinline__ void sdrain_n(V (&b)[2]) { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(b[0]), "+s"(b[1]) : : "memory"); }
template <class V> __device__ __forceinline__ void sdrain_n(V (&b)[3]) { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(b[0]), "+s"(b[1]), "+s"(b[2]) : : "memory"); }
template <class V> __device__ __forceinline__ void sdrain_n(V (&b)[4]) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(b[0]), "+s"(b[1]), "+s"(b[2]), "+s"(b[3]) : : "memory");
}

template <int L, int NS, int CH>
struct SgprStreamCoef {
    typedef typename SVec<CH>::type V;
    const float* base;   // the query's coefficient row (wave-uniform)
    V (&ring)[2][NS];
    template <int K>
    __device__ __forceinline__ void request() const {  // chunk K of every stream -> ring[K & 1]
        static_for<NS>([&](auto ss) {
            constexpr int st = decltype(ss)::value;
            sload<(st * L + K * CH) * 4>(ring[K & 1][st], base);
        });
    }
    template <int I>
    __device__ __forceinline__ float operator()(ic<I>) const {
        constexpr int st = I / L, j = I % L, k = j / CH, i = j % CH;
        static_assert(st < NS, "coefficient index outside the streams");
        if constexpr (st == 0 && i == 0) {  // first touch of chunk k (the Scorer reads stream 0 first in every term)
            sdrain_n(ring[k & 1]);
            if constexpr ((k + 1) * CH < L) request<k + 1>();
        }
        return ring[k & 1][st][i];
    }
};

// streams of Scorer<MODEL, SIDE, D>: count, length, and the chunk that keeps 2 x NS x CH floats within the SGPR file
template <int MODEL, int SIDE, int D> struct StreamPlan;
template <int D> struct StreamPlan<TRANSE, TAIL, D> { static constexpr int NS = 1, L = D, CH = 16; };
template <int D> struct StreamPlan<TRANSE, HEAD, D> { static constexpr int NS = 2, L = D, CH = 16; };
template <int D> struct StreamPlan<DISTMULT, TAIL, D> { static constexpr int NS = 1, L = D, CH = 16; };
template <int D> struct StreamPlan<DISTMULT, HEAD, D> { static constexpr int NS = 2, L = D, CH = 16; };
template <int SIDE, int D> struct StreamPlan<COMPLEX, SIDE, D> { static constexpr int NS = 4, L = D / 2, CH = 8; };
template <int SIDE, int D> struct StreamPlan<SIMPLE, SIDE, D> { static constexpr int NS = 3, L = D / 2, CH = 8; };

template <int MODEL, int SIDE, int D>
__device__ __forceinline__ float score_sgpr_stream(const float (&e)[D], const float* row) {
    using P = StreamPlan<MODEL, SIDE, D>;
    static_assert(Scorer<MODEL, SIDE, D>::C == P::NS * P::L, "stream plan does not cover the coefficient row");
    typename SVec<P::CH>::type ring[2][P::NS];
    const SgprStreamCoef<P::L, P::NS, P::CH> c{row, ring};
    c.template request<0>();
    return Scorer<MODEL, SIDE, D>::template score<false>(e, c);
}

// The operand rows W_q come through the scalar cache, loads issued by hand in units of 16 columns of one query: the next
// unit is requested before the current one's 16 multiply-adds and drained after them (rank_common.h: sload16 / sdrain;
// scalar loads return out of order, every wait is a full drain).  Left alone the compiler hoists the loop-invariant
// values out of the tile loop and parks them in VGPR lanes -- a v_readlane per multiply-add -- or, once the pointer is
// made opaque, falls back to per-lane vector loads; an SGPR ring behind an accessor (SgprStreamCoef) spills the same way.
// Both asm statements carry the unit's running sum as an operand: the request stands before the unit's first
// multiply-add, the wait behind its last (volatile asm keeps its own order, not its place among the arithmetic --
// unpinned, the compiler issues every request and wait of a piece up front and parks the values in VGPR lanes all the same).
template <int MODEL, int D>
__global__ __launch_bounds__(kWaves * 64, 2) void rank_stream_wg_kernel(
    const float* __restrict__ table, int64_t N, int64_t ld, const float* __restrict__ coef_head,
    const float* __restrict__ coef_tail, const float* __restrict__ key_true, int q_head, int q_tail, int n_tiles,
    unsigned long long* __restrict__ acc) {
    using SH = Scorer<MODEL, HEAD, D>;
    using ST = Scorer<MODEL, TAIL, D>;
    constexpr int TS = D + 4;  // row stride of an LDS tile, floats
    constexpr int kPieces = kTileRows * (D / 4) / (kWaves * 64);
    extern __shared__ __attribute__((aligned(16))) float smem[];  // 2 x kTileRows x TS
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int Q = q_head + q_tail;

    // thread t moves 16-byte pieces t, t + 256, ... of the 64 x D block (whole rows per D / 4 threads); rows past the end
    // of the table: the last row, masked when counting.  TWO tiles in flight per workgroup (a register ring): with one,
    // the memory pipeline drains while the workgroup hands a tile over, and the pass runs at load latency + arithmetic
    // per tile instead of the larger of the two.
    f32x4 ring[2][kPieces];
    auto fetch = [&](f32x4 (&piece)[kPieces], int t) {
        const int64_t row0 = (int64_t)t * kTileRows;
        static_for<kPieces>([&](auto kk) {
            constexpr int k = decltype(kk)::value;
            const int idx = tid + k * kWaves * 64, r = idx / (D / 4), c = idx % (D / 4);
            int64_t row = row0 + r;
            row = row < N ? row : N - 1;
            piece[k] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(table + row * ld + 4 * c));
        });
    };
    auto put = [&](float* tile, const f32x4 (&piece)[kPieces]) {
        static_for<kPieces>([&](auto kk) {
            constexpr int k = decltype(kk)::value;
            const int idx = tid + k * kWaves * 64, r = idx / (D / 4), c = idx % (D / 4);
            *reinterpret_cast<f32x4*>(tile + r * TS + 4 * c) = piece[k];
        });
    };
    unsigned n_gt[2] = {0, 0}, n_ge[2] = {0, 0};  // wave-uniform
    const int qw = wave;  // wave w scores queries w and w + 4
    // score the tile in LDS buffer `cur` (tile index t) with this wave's queries
    auto score_tile = [&](int t, int cur) {
        const float* tile = smem + cur * (kTileRows * TS);
        float e[D];
        static_for<D / 4>([&](auto jj) {
            constexpr int j = decltype(jj)::value;
            const float4 w = *reinterpret_cast<const float4*>(tile + lane * TS + 4 * j);
            e[4 * j] = w.x; e[4 * j + 1] = w.y; e[4 * j + 2] = w.z; e[4 * j + 3] = w.w;
        });
        const bool valid = (int64_t)t * kTileRows + lane < N;
        static_for<2>([&](auto ss) {
            constexpr int s = decltype(ss)::value;
            const int q = qw + kWaves * s;
            if (q < Q) {
                const float key = q < q_head ? score_sgpr_stream<MODEL, HEAD, D>(e, coef_head + (size_t)q * SH::C)
                                             : score_sgpr_stream<MODEL, TAIL, D>(e, coef_tail + (size_t)(q - q_head) * ST::C);
                const float kt = key_true[q];
                n_gt[s] += __popcll(__ballot(valid && key > kt));
                n_ge[s] += __popcll(__ballot(valid && key >= kt));
            }
        });
    };

    // tiles t0, t0 + G, t0 + 2G, ...: in LDS tile i (buffer i & 1), in ring[(i + 1) & 1] tile i + 1, in ring[i & 1] tile i + 2
    const int G = gridDim.x;
    int t = blockIdx.x;  // (the grid has at most n_tiles workgroups)
    fetch(ring[0], t);
    if (t + G < n_tiles) fetch(ring[1], t + G);
    put(smem, ring[0]);
    if (t + 2 * G < n_tiles) fetch(ring[0], t + 2 * G);
    __syncthreads();
    for (;;) {  // two tiles per trip, so that the ring slots and LDS buffers are compile-time
        score_tile(t, 0);
        if (t + G >= n_tiles) break;
        put(smem + kTileRows * TS, ring[1]);
        if (t + 3 * G < n_tiles) fetch(ring[1], t + 3 * G);
        __syncthreads();  // tile t + G is in LDS buffer 1; every wave is done with buffer 0
        score_tile(t + G, 1);
        if (t + 2 * G >= n_tiles) break;
        put(smem, ring[0]);
        if (t + 4 * G < n_tiles) fetch(ring[0], t + 4 * G);
        __syncthreads();
        t += 2 * G;
    }
    if (lane == 0) {
        static_for<2>([&](auto ss) {
            constexpr int s = decltype(ss)::value;
            const int q = qw + kWaves * s;
            const unsigned long long v = (unsigned long long)n_gt[s] | ((unsigned long long)n_ge[s] << 32);
            if (q < Q && v) atomicAdd(acc + q, v);
        });
    }
}

// ---- the bilinear models on the ring: approximate keys ----------------------------------------------------------------
// The dot product <W_q, e> can be summed in any order -- the band does not care -- so it CAN be carried across 32-column
// pieces: the bilinear models take the ring kernel's structure (a wave's two pieces in flight across tile boundaries, the
// pass ramps up and down in a few microseconds: what a 1/8 shard of the Wikidata5M table needs) with D fused multiply-adds
// per (row, query) and D for the row's squared norm.  Undecided lanes (the true entity, near-ties, non-finite values:
// about one (tile, query) in a hundred has one) are re-scored on the spot by coop_score -- 32 lanes per pair, the row
// re-read from the cache it has just passed through, the order-exact arithmetic of every other exact path.
// Operand units as above (sload16_pinned / sdrain_pinned).  Unit u of piece S: query u / 2, columns 32 S + 16 (u % 2) ..; the
// unit after a tile's last is the next tile's first.  `cur` holds the unit about to be used.
// (the squared norm's chain rides along with query 0's two units -- on its own it drifts behind the following pieces and
//  keeps three pieces' columns alive)
template <int D, int NQ, int S, int U>
__device__ __forceinline__ void dot_units(float (&sum)[NQ], float& ssq, const float (&x)[kSubCols], const float* wq, sf16& cur) {
    if constexpr (U < 2 * NQ) {
        constexpr int NP = D / kSubCols;
        constexpr int nu = (U + 1) % (2 * NQ), ns = U + 1 < 2 * NQ ? S : (S + 1) % NP;  // the unit after this one
        constexpr int q = U / 2, c0 = 16 * (U % 2);
        sf16 nxt;
        sload16_pinned<(stream_dot_row(nu / 2) * D + ns * kSubCols + 16 * (nu % 2)) * 4>(nxt, wq, sum[q]);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            sum[q] = __builtin_fmaf(x[c0 + i], cur[i], sum[q]);
            if constexpr (q == 0) ssq = __builtin_fmaf(x[c0 + i], x[c0 + i], ssq);
        }
        if constexpr (q == 0) sdrain_pinned(nxt, sum[q], ssq);
        else sdrain_pinned(nxt, sum[q]);
        cur = nxt;
        dot_units<D, NQ, S, U + 1>(sum, ssq, x, wq, cur);
    }
}

template <int MODEL, int D, int NQ>
__global__ __launch_bounds__(kWaves * 64, NQ <= 4 ? 4 : 3) void rank_stream_dot_kernel(
    const float* __restrict__ table, int64_t N, int64_t ld, const float* __restrict__ wq, const float* __restrict__ band,
    const float* __restrict__ key_true, const QRows q_fixed, const QRows q_rel, int64_t q0, int q_head, int q_tail,
    int n_tiles, unsigned long long* __restrict__ acc, const StreamPasses passes) {
    constexpr int NP = D / kSubCols;  // NQ: a pass has at most NQ queries (4: the reference's Wikidata5M batch of two triples)
    static_assert(NP % 2 == 0, "the ring of two pieces assumes an even number of pieces per tile");
    static_assert(NQ == 4 || NQ == 2 * kStreamQ, "a pass of up to 4, or up to 4 + 4, queries");
    __shared__ __attribute__((aligned(16))) float slabs[kWaves * kSlabFloats];
    __shared__ unsigned long long wg_cnt[NQ];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* slab = slabs + wave * kSlabFloats;
    const int sub_row = lane >> 3, sub_col = (lane & 7) * 4;
    float* wr = slab + sub_row * kLdsStride + sub_col;
    const float* rd = slab + lane * kLdsStride;
    if (tid < NQ) wg_cnt[tid] = 0;

    // rounds of four tiles over all passes of the launch: see rank_stream_kernel
    const unsigned n_rounds = (unsigned)(n_tiles + kWaves - 1) / kWaves, total = n_rounds * (unsigned)passes.n_passes;
    const unsigned long long slot_off = passes.acc_slots > 1 ? (unsigned long long)(blockIdx.x % passes.acc_slots) * (2ull * passes.n) : 0ull;
    auto tile_of = [&](unsigned r) { return (int)((r % n_rounds) * kWaves) + wave; };  // may be >= n_tiles
    unsigned boff[8];
    auto offsets = [&](int t) {
        const int64_t left = N - (int64_t)t * kTileRows;
        const int last = left < kTileRows ? (int)left - 1 : kTileRows - 1;  // wave-uniform
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int r = 8 * i + sub_row;
            boff[i] = (unsigned)(((r < last ? r : last) * (int)ld + sub_col) * 4);
        }
    };
    auto tile_base = [&](int t) { return table + (int64_t)t * kTileRows * ld; };
    auto clamp_tile = [&](int t) { return t < n_tiles ? t : n_tiles - 1; };

    unsigned n_gt[NQ] = {};           // certainly above the true key (gt and ge alike): wave-uniform, scalar registers
    unsigned long long ex_cnt = 0;    // lane q: query q's undecided rows at or above it, gt | ge << 32
    // the pass the wave is in (queries heads first: query q < Q of the pass is q0_p + q of the call)
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
        cur = sload16<stream_dot_row(0) * D * 4>(wq_p);  // unit 0 of piece 0 of this pass's operands
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
        offsets(t0);
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
        const float* base = tile_base(tile);
        float sum[NQ] = {}, ssq = 0.f;
        static_for<NP>([&](auto ss) {
            constexpr int s = decltype(ss)::value, pp = s & 1;
#pragma unroll
            for (int i = 0; i < 8; ++i)
                *reinterpret_cast<f32x4*>(wr + 8 * i * kLdsStride) = ring[pp][i];
            if constexpr (s + 2 < NP) {
                piece_fetch(ring[pp], base, boff, s + 2);
            } else {
                if constexpr (s + 2 == NP) {
                    if (more) offsets(next);
                }
                if (more) piece_fetch(ring[pp], tile_base(next), boff, s + 2 - NP);
            }
            wave_lds_sync();
            float x[kSubCols];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(rd + 4 * j);
                x[4 * j] = v.x; x[4 * j + 1] = v.y; x[4 * j + 2] = v.z; x[4 * j + 3] = v.w;
            }
            wave_lds_sync();  // the reads are out before the next piece overwrites the slab
            // (no branch per query: a slot the pass has no query for walks its -- unwritten -- operand row, never counted)
            dot_units<D, NQ, s, 0>(sum, ssq, x, wq_p, cur);
        });
        const bool valid = tile_real < n_tiles && (int64_t)tile * kTileRows + lane < N;
        const float nrow = sqrtf(ssq) * 1.0001f;
        const bool tiny = ssq < 1e-30f;
        // lane q parks query q's undecided rows (a mask); the exact routine then runs once, over run-time q
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
        if (any_und) {  // wave-uniform, rare (about one (tile, query) in a hundred has an undecided row)
            for (int q = 0; q < Q; ++q) {
                const unsigned long long und = (unsigned long long)(unsigned)__builtin_amdgcn_readlane(und_lo, q) |  // (returns int)
                                               ((unsigned long long)(unsigned)__builtin_amdgcn_readlane(und_hi, q) << 32);
                if (!und) continue;
                const unsigned long long c = exact_undecided<MODEL, D>(table, ld, (int64_t)tile * kTileRows, q_fixed.row(q0_p + q),
                                                                       q_rel.row(q0_p + q), q < qh, und, kt[q], lane);
                if (lane == q) ex_cnt += c;
            }
        }
    }

    // the last pass of the workgroup (all of its waves walk the same rounds): summed over its waves in LDS, one atomic per query
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

bool rank_stream_applicable(int model, int D, int64_t N, int64_t ld, int64_t q_head, int64_t q_tail) {
    if (q_head > kStreamQ || q_tail > kStreamQ || N <= 0) return false;
    if (model == TRANSE) return (D == 64 || D == 128 || D == 256) && ld < (1 << 22);  // (32-bit byte offsets inside a tile)
    if (D == 256) return ld < (1 << 22) && knob(KNOB_STREAM_KERNEL) != 3 && knob(KNOB_STREAM_KERNEL) != 5;  // the ring with approximate keys only
    return D == 64 || D == 128;  // (the workgroup-tile kernel: a whole row + 32 sums in registers)
}

// Which kernel takes a pass of a bilinear model ([measured] 4-query pass, us, ring with approximate keys / workgroup tile
// with order-exact keys: 575 k rows DistMult 54 / 70, ComplEx 55 / 80, SimplE 53 / 69; 1.15 M rows 101 / 108, 102 / 132,
// 101 / 108; 4.6 M rows 368 / 355, 369 / 471, 367 / 352): the ring below kStreamWgMinRows rows and for ComplEx at any
// length; knob stream_kernel = 3 (or 5) / 4 forces the workgroup-tile / the ring kernel.
// `passes`: the call is many passes, which the ring kernels take in one launch (below) -- ramps paid once per call, not per
// pass: [measured, ComplEx] 369 -> 338 us per 4.6 M-row pass, past the workgroup tile's 350 - 355 for DistMult / SimplE.
static bool bilinear_takes_ring(int model, int D, int64_t N, int64_t ld, bool passes) {
    const long long forced = knob(KNOB_STREAM_KERNEL);
    if (ld >= (1 << 22)) return false;  // (32-bit byte offsets inside a tile)
    if (forced == 3 || forced == 5) return false;
    return forced == 4 || passes || model == COMPLEX || D == 256 || N < kStreamWgMinRows;
}
// ... and so needs the operand rows and bands of rank_all.hip's preparation launch
bool rank_stream_wants_dot(int model, int D, int64_t N, int64_t ld, int64_t q_head, int64_t q_tail, bool passes) {
    return model != TRANSE && knob(KNOB_STREAM_KERNEL) != 2 && rank_stream_applicable(model, D, N, ld, q_head, q_tail) &&
           bilinear_takes_ring(model, D, N, ld, passes);
}

template <int MODEL, int D>
static hipError_t launch_stream_wg(const float* table, int64_t N, int64_t ld, const float* coef_head, const float* coef_tail,
                                   const float* key_true, int q_head, int q_tail, int n_tiles, unsigned long long* acc,
                                   int n_cu, hipStream_t stream) {
    const size_t lds = (size_t)2 * kTileRows * (D + 4) * 4;
    if (lds > 64 * 1024) {  // beyond the default dynamic-LDS limit
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&rank_stream_wg_kernel<MODEL, D>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    const int resident = 2 * n_cu;
    rank_stream_wg_kernel<MODEL, D><<<(unsigned)(n_tiles < resident ? n_tiles : resident), kWaves * 64, lds, stream>>>(
        table, N, ld, coef_head, coef_tail, key_true, q_head, q_tail, n_tiles, acc);
    return hipGetLastError();
}

template <int MODEL, int D>
static hipError_t launch_stream_dot(const float* table, int64_t N, int64_t ld, const StreamDot& dot, const float* key_true,
                                    int q_head, int q_tail, int n_tiles, unsigned long long* acc, int n_cu, hipStream_t stream,
                                    const StreamPasses& passes) {
    const int64_t n_rounds = (((int64_t)n_tiles + kWaves - 1) / kWaves) * (passes.n_passes > 1 ? passes.n_passes : 1);
    const int per_pass = passes.n_passes > 1 ? 2 * passes.batch : q_head + q_tail;
    if (per_pass <= 4) {  // four waves per SIMD
        const int64_t resident = (int64_t)n_cu * 4;
        rank_stream_dot_kernel<MODEL, D, 4><<<(unsigned)(n_rounds < resident ? n_rounds : resident), kWaves * 64, 0, stream>>>(
            table, N, ld, dot.wq, dot.band, key_true, dot.q_fixed, dot.q_rel, dot.q0, q_head, q_tail, n_tiles, acc, passes);
    } else {  // eight running sums: three
        const int64_t resident = (int64_t)n_cu * 3;
        rank_stream_dot_kernel<MODEL, D, 2 * kStreamQ><<<(unsigned)(n_rounds < resident ? n_rounds : resident), kWaves * 64, 0, stream>>>(
            table, N, ld, dot.wq, dot.band, key_true, dot.q_fixed, dot.q_rel, dot.q0, q_head, q_tail, n_tiles, acc, passes);
    }
    return hipGetLastError();
}

// TransE: the ring or the workgroup-tile kernel ([measured] 4-query pass, ring / workgroup tile: 1.15 M rows 103 / 109,
// 2.3 M rows 198 / 188, 4.6 M rows 384 / 353 us = 6.7 TB/s; a 1/8 Wikidata5M shard: 58 against 72); knob stream_kernel = 3 / 4
// forces the workgroup-tile / the ring kernel
// (many passes in one launch: [measured] ring / workgroup tile, a launch per pass: 2.3 M rows 176 - 179 / 180 - 183, 4.6 M rows
//  346 - 355 / 346 - 348 us per pass; the workgroup-tile kernel walking all passes in one launch was tried: 390)
static bool transe_takes_ring(int D, int64_t N, bool passes) {
    const long long forced = knob(KNOB_STREAM_KERNEL);
    return !(D != 256 && forced != 4 && (forced == 3 || (!passes && N >= kStreamWgMinRows)));
}

// All passes of a call in one launch: the ring kernels only (the workgroup-tile kernel's passes are long: their ramps
// are a per cent of them), and while the rounds of all passes fit the kernels' 32-bit round index.
bool rank_stream_takes_passes(int model, int D, int64_t N, int64_t ld, int64_t batch, int64_t n) {
    if (knob(KNOB_STREAM_KERNEL) == 2 || !rank_stream_applicable(model, D, N, ld, batch, batch)) return false;
    const int64_t n_rounds = ((N + kTileRows - 1) / kTileRows + kWaves - 1) / kWaves, n_passes = (n + batch - 1) / batch;
    if (n_passes <= 1 || n_rounds * n_passes >= (int64_t)0x7fffffff) return false;
    return model == TRANSE ? transe_takes_ring(D, N, true) : bilinear_takes_ring(model, D, N, ld, true);
}

hipError_t launch_rank_stream(int model, int D, const float* table, int64_t N, int64_t ld, const float* coef_head,
                              const float* coef_tail, const float* key_true, int64_t q_head, int64_t q_tail,
                              unsigned long long* acc, const StreamDot& dot, int n_cu, hipStream_t stream,
                              const StreamPasses& passes) {
    const int64_t n_tiles = (N + kTileRows - 1) / kTileRows;
    if (n_tiles > 0x7fffffff) return hipErrorInvalidValue;
    if (passes.n_passes > 1 && !rank_stream_takes_passes(model, D, N, ld, passes.batch, passes.n)) return hipErrorInvalidValue;
    if (model != TRANSE) {
        if (dot.wq != nullptr && bilinear_takes_ring(model, D, N, ld, passes.n_passes > 1)) {
#define BLP_STREAM_DOT(MM, DD)                                                                                       \
    if (model == MM && D == DD)                                                                                      \
        return launch_stream_dot<MM, DD>(table, N, ld, dot, key_true, (int)q_head, (int)q_tail, (int)n_tiles, acc, n_cu, stream, passes);
            BLP_STREAM_DOT(DISTMULT, 64) BLP_STREAM_DOT(DISTMULT, 128) BLP_STREAM_DOT(DISTMULT, 256) BLP_STREAM_DOT(COMPLEX, 64)
            BLP_STREAM_DOT(COMPLEX, 128) BLP_STREAM_DOT(COMPLEX, 256) BLP_STREAM_DOT(SIMPLE, 64) BLP_STREAM_DOT(SIMPLE, 128) BLP_STREAM_DOT(SIMPLE, 256)
#undef BLP_STREAM_DOT
            return hipErrorInvalidValue;
        }
#define BLP_STREAM_WG(MM, DD)                                                                                        \
    if (model == MM && D == DD)                                                                                      \
        return launch_stream_wg<MM, DD>(table, N, ld, coef_head, coef_tail, key_true, (int)q_head, (int)q_tail,       \
                                        (int)n_tiles, acc, n_cu, stream);
        BLP_STREAM_WG(DISTMULT, 64) BLP_STREAM_WG(DISTMULT, 128) BLP_STREAM_WG(COMPLEX, 64) BLP_STREAM_WG(COMPLEX, 128)
        BLP_STREAM_WG(SIMPLE, 64) BLP_STREAM_WG(SIMPLE, 128)
        return hipErrorInvalidValue;
    }
    if (!transe_takes_ring(D, N, passes.n_passes > 1)) {
        BLP_STREAM_WG(TRANSE, 64) BLP_STREAM_WG(TRANSE, 128)
    }
#undef BLP_STREAM_WG
    // (the ring kernel takes ONE coefficient array: a pass's head rows, then its tail rows)
    if (passes.n_passes <= 1 && q_tail > 0 && coef_tail != coef_head + q_head * 2 * D) return hipErrorInvalidValue;
    const int64_t n_rounds = ((n_tiles + kWaves - 1) / kWaves) * (passes.n_passes > 1 ? passes.n_passes : 1);
    const int64_t resident = (int64_t)n_cu * (D == 256 ? 3 : 4);  // workgroups of four waves per CU: 4 (3) waves per SIMD
    const unsigned blocks = (unsigned)(n_rounds < resident ? n_rounds : resident);
    // queries per side the kernel is compiled for: a pass's batch, or the larger side of a single call
    const int64_t per_side = passes.n_passes > 1 ? passes.batch : (q_head > q_tail ? q_head : q_tail);
#define BLP_STREAM_CASE(DD, QQ)                                                                                      \
    case DD * 8 + QQ:                                                                                                \
        rank_stream_kernel<DD, QQ><<<blocks, kWaves * 64, 0, stream>>>(table, N, ld, coef_head, key_true, (int)q_head, \
                                                                       (int)q_tail, (int)n_tiles, acc, passes);      \
        break;
    switch (D * 8 + (int)per_side) {
        BLP_STREAM_CASE(64, 1) BLP_STREAM_CASE(64, 2) BLP_STREAM_CASE(64, 3) BLP_STREAM_CASE(64, 4)
        BLP_STREAM_CASE(128, 1) BLP_STREAM_CASE(128, 2) BLP_STREAM_CASE(128, 3) BLP_STREAM_CASE(128, 4)
        BLP_STREAM_CASE(256, 1) BLP_STREAM_CASE(256, 2) BLP_STREAM_CASE(256, 3) BLP_STREAM_CASE(256, 4)
    default: return hipErrorInvalidValue;
    }
#undef BLP_STREAM_CASE
    return hipGetLastError();
}

}  // namespace blp
