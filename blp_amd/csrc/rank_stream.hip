// rank_stream.hip -- a handful of queries against a LONG table (the reference's Wikidata5M evaluation batch:
// eval_batch_size 2 = 4 queries per pass over 4.6 M rows, scripts/blp-*-wikidata5m.sh:18; train.py:128-171): the
// pass is one read of the table, HBM-bound, and what matters is that the read never stops.  TransE's per-wave ring first
// (tables of up to 1.7 M rows); the workgroup-tile kernel (the bilinear models; TransE on longer tables) is further down.
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

#include "knobs.h"
#include "launch.h"
#include "rank_common.h"
#include "score_core.h"
#include "tile.h"

#pragma clang fp contract(off)

namespace blp {

constexpr int kStreamQ = 4;  // queries per side (== rank_all.hip's kQB: what its static mode takes)
constexpr int64_t kStreamWgMinRows = 1700000;  // TransE: from here on the workgroup-tile kernel (launch_rank_stream)

typedef float f32x4 __attribute__((ext_vector_type(4)));

// The 8 loads of one 32-column piece: piece s of the tile whose first row is `base` (byte offsets boff[i] of this lane's
// part of rows 8i .. 8i + 7, clamped to the table by the caller).  Streamed once: non-temporal.
template <class Off>
__device__ __forceinline__ void piece_fetch(f32x4 (&b)[8], const float* __restrict__ base, const Off (&boff)[8], int s) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
        b[i] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(base) + boff[i] + s * (kSubCols * 4)));
}

template <int D>
__global__ __launch_bounds__(kWaves * 64, D == 256 ? 3 : 4) void rank_stream_kernel(
    const float* __restrict__ table, int64_t N, int64_t ld, const float* __restrict__ coef_head,
    const float* __restrict__ coef_tail, const float* __restrict__ key_true, int q_head, int q_tail, int n_tiles,
    unsigned long long* __restrict__ acc) {
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

    // the wave's tiles: four consecutive tiles per workgroup and round, workgroups grid-stride over the rounds
    const int stride = gridDim.x * kWaves;
    int tile = blockIdx.x * kWaves + wave;

    // byte offsets of this lane's parts of a tile's rows (rows past the end of the table: the last row, masked later)
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

    unsigned n_gt[2 * kStreamQ] = {}, n_ge[2 * kStreamQ] = {};  // wave-uniform: scalar registers
    f32x4 ring[2][8];
    if (tile < n_tiles) {
        offsets(tile);
        piece_fetch(ring[0], tile_base(tile), boff, 0);
        piece_fetch(ring[1], tile_base(tile), boff, 1);
    }
    for (; tile < n_tiles; tile += stride) {
        const int next = tile + stride;
        const bool more = next < n_tiles;  // wave-uniform
        const float* base = tile_base(tile);
        float sum[2 * kStreamQ] = {};
        static_for<NP>([&](auto ss) {
            constexpr int s = decltype(ss)::value, p = s & 1;
            // piece s: registers -> slab (transposing) ...
#pragma unroll
            for (int i = 0; i < 8; ++i)
                *reinterpret_cast<f32x4*>(wr + 8 * i * kLdsStride) = ring[p][i];
            // ... and its ring slot is refilled with the piece two steps ahead: this tile's, or the next tile's
            if constexpr (s + 2 < NP) {
                piece_fetch(ring[p], base, boff, s + 2);
            } else {
                if constexpr (s + 2 == NP) {
                    if (more) offsets(next);
                }
                if (more) piece_fetch(ring[p], tile_base(next), boff, s + 2 - NP);
            }
            wave_lds_sync();
            float x[kSubCols];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(rd + 4 * j);
                x[4 * j] = v.x; x[4 * j + 1] = v.y; x[4 * j + 2] = v.z; x[4 * j + 3] = v.w;
            }
            wave_lds_sync();  // the reads are out before the next piece overwrites the slab
            static_for<kStreamQ>([&](auto jj) {
                constexpr int j = decltype(jj)::value;
                if (j < q_tail) sum[kStreamQ + j] = transe_piece_sum<TAIL, D, s == 0>(sum[kStreamQ + j], x, coef_tail + j * D + s * kSubCols);
            });
            static_for<kStreamQ>([&](auto jj) {
                constexpr int j = decltype(jj)::value;
                if (j < q_head) sum[j] = transe_piece_sum<HEAD, D, s == 0>(sum[j], x, coef_head + j * 2 * D + s * kSubCols);
            });
        });
        const bool valid = (int64_t)tile * kTileRows + lane < N;
        static_for<kStreamQ>([&](auto jj) {
            constexpr int j = decltype(jj)::value;
            if (j < q_head) {
                const float key = -sum[j], kt = key_true[j];
                n_gt[j] += __popcll(__ballot(valid && key > kt));
                n_ge[j] += __popcll(__ballot(valid && key >= kt));
            }
            if (j < q_tail) {
                const float key = -sum[kStreamQ + j], kt = key_true[q_head + j];
                n_gt[kStreamQ + j] += __popcll(__ballot(valid && key > kt));
                n_ge[kStreamQ + j] += __popcll(__ballot(valid && key >= kt));
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
    if (tid < 2 * kStreamQ) {  // slot j < 4: head query j; slot 4 + j: tail query j
        const int side_q = tid < kStreamQ ? tid : tid - kStreamQ;
        const bool live = tid < kStreamQ ? side_q < q_head : side_q < q_tail;
        const unsigned long long v = wg_cnt[tid];
        if (live && v) atomicAdd(acc + (tid < kStreamQ ? side_q : q_head + side_q), v);
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

bool rank_stream_applicable(int model, int D, int64_t N, int64_t ld, int64_t q_head, int64_t q_tail) {
    if (q_head > kStreamQ || q_tail > kStreamQ || N <= 0) return false;
    if (model == TRANSE) return (D == 64 || D == 128 || D == 256) && ld < (1 << 22);  // (32-bit byte offsets inside a tile)
    return D == 64 || D == 128;  // a whole row + 32 sums in registers
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

hipError_t launch_rank_stream(int model, int D, const float* table, int64_t N, int64_t ld, const float* coef_head,
                              const float* coef_tail, const float* key_true, int64_t q_head, int64_t q_tail,
                              unsigned long long* acc, int n_cu, hipStream_t stream) {
    const int64_t n_tiles = (N + kTileRows - 1) / kTileRows;
    if (n_tiles > 0x7fffffff) return hipErrorInvalidValue;
    // TransE: the per-wave ring ramps up and down faster (1/8 Wikidata5M shard: 58 us against 72), the workgroup-tile kernel
    // streams a long table better ([measured] 4-query pass: 1.15 M rows 103 / 109, 2.3 M rows 198 / 188, 4.6 M rows 384 / 353 us
    // = 6.7 TB/s); knob stream_kernel = 3 / 4 forces the workgroup-tile / the ring kernel
    const long long forced = knob(KNOB_STREAM_KERNEL);
    const bool wg_tile = model != TRANSE || (D != 256 && forced != 4 && (forced == 3 || N >= kStreamWgMinRows));
    if (wg_tile) {
#define BLP_STREAM_WG(MM, DD)                                                                                        \
    if (model == MM && D == DD)                                                                                      \
        return launch_stream_wg<MM, DD>(table, N, ld, coef_head, coef_tail, key_true, (int)q_head, (int)q_tail,       \
                                        (int)n_tiles, acc, n_cu, stream);
        BLP_STREAM_WG(TRANSE, 64) BLP_STREAM_WG(TRANSE, 128) BLP_STREAM_WG(DISTMULT, 64) BLP_STREAM_WG(DISTMULT, 128) BLP_STREAM_WG(COMPLEX, 64) BLP_STREAM_WG(COMPLEX, 128)
        BLP_STREAM_WG(SIMPLE, 64) BLP_STREAM_WG(SIMPLE, 128)
#undef BLP_STREAM_WG
        return hipErrorInvalidValue;
    }
    const int64_t n_rounds = (n_tiles + kWaves - 1) / kWaves;
    const int64_t resident = (int64_t)n_cu * (D == 256 ? 3 : 4);  // workgroups of four waves per CU: 4 (3) waves per SIMD
    const unsigned blocks = (unsigned)(n_rounds < resident ? n_rounds : resident);
#define BLP_STREAM_CASE(DD)                                                                                          \
    case DD:                                                                                                         \
        rank_stream_kernel<DD><<<blocks, kWaves * 64, 0, stream>>>(table, N, ld, coef_head, coef_tail, key_true,      \
                                                                   (int)q_head, (int)q_tail, (int)n_tiles, acc);     \
        break;
    switch (D) {
        BLP_STREAM_CASE(64)
        BLP_STREAM_CASE(128)
        BLP_STREAM_CASE(256)
    default: return hipErrorInvalidValue;
    }
#undef BLP_STREAM_CASE
    return hipGetLastError();
}

}  // namespace blp
