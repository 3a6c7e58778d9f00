// tile.h -- a wavefront's tile of 64 consecutive table rows, one row per lane, in D VGPRs (the layout of the exact
// f32 kernels: rank_all.hip, rank_small.hip).  The rows are fetched with coalesced 16-B/lane loads (8 rows x 128 B
// per wave instruction, whole cache lines) and transposed through a wave-private, bank-conflict-free LDS slab.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rank_common.h"
#include "score_core.h"

namespace blp {

constexpr int kWaves = 4;            // waves per workgroup
constexpr int kTileRows = 64;        // candidates per wave tile (one per lane)
constexpr int kSubCols = 32;         // floats per row per LDS pass (128 B = one cache line)
constexpr int kLdsStride = 36;       // dwords; 36*l mod 64 is conflict-free for ds_read_b128
constexpr int kSlabFloats = kTileRows * kLdsStride;

// 1. every global load of the tile up front (D/4 x 1 KiB in flight per wave), landing in e[] in the coalesced layout:
//    e[32s + 4i .. +3] = row (row0 + 8i + lane/8), cols 32s + 4(lane%8) .. +3.  Rows past the table end are clamped to
//    the last row (the caller masks their counts).
template <int D, bool NT>
__device__ __forceinline__ void tile_fetch(float (&e)[D], const float* __restrict__ table, int64_t N, int64_t ld,
                                           int64_t row0, int lane) {
    const int sub_row = lane >> 3;        // 8 rows per wave instruction
    const int sub_col = (lane & 7) * 4;   // 8 x 16 B = one 128-B line per row
    static_for<8>([&](auto ii) {
        constexpr int i = decltype(ii)::value;
        int64_t row = row0 + 8 * i + sub_row;
        row = row < N ? row : N - 1;
        const float* src = table + row * ld + sub_col;
        static_for<D / kSubCols>([&](auto ss) {
            constexpr int s = decltype(ss)::value;
            typedef float floatx4 __attribute__((ext_vector_type(4)));
            const floatx4 v = NT ? __builtin_nontemporal_load(reinterpret_cast<const floatx4*>(src + s * kSubCols))
                                 : *reinterpret_cast<const floatx4*>(src + s * kSubCols);
            e[32 * s + 4 * i] = v.x; e[32 * s + 4 * i + 1] = v.y;
            e[32 * s + 4 * i + 2] = v.z; e[32 * s + 4 * i + 3] = v.w;
        });
    });
}

// 2. transpose 32 columns at a time through the wave's slab, in place in e[]: afterwards lane l holds row row0 + l
template <int D>
__device__ __forceinline__ void tile_transpose(float (&e)[D], float* slab, int lane) {
    const int sub_row = lane >> 3, sub_col = (lane & 7) * 4;
    float* wr = slab + sub_row * kLdsStride + sub_col;
    const float* rd = slab + lane * kLdsStride;
    static_for<D / kSubCols>([&](auto ss) {
        constexpr int s = decltype(ss)::value;
        if (s > 0) wave_lds_sync();  // previous pass' reads are done before the slab is rewritten
        static_for<8>([&](auto ii) {
            constexpr int i = decltype(ii)::value;
            *reinterpret_cast<float4*>(wr + 8 * i * kLdsStride) =
                make_float4(e[32 * s + 4 * i], e[32 * s + 4 * i + 1], e[32 * s + 4 * i + 2], e[32 * s + 4 * i + 3]);
        });
        wave_lds_sync();
        static_for<8>([&](auto jj) {
            constexpr int j = decltype(jj)::value;
            const float4 w = *reinterpret_cast<const float4*>(rd + 4 * j);
            e[32 * s + 4 * j] = w.x; e[32 * s + 4 * j + 1] = w.y;
            e[32 * s + 4 * j + 2] = w.z; e[32 * s + 4 * j + 3] = w.w;
        });
    });
}

// Fetch rows [row0, row0 + 64) of the table into e[] (lane l <- row row0 + l) through the wave's LDS slab.
template <int D, bool NT>
__device__ __forceinline__ void load_tile(float (&e)[D], const float* __restrict__ table, int64_t N,
                                          int64_t ld, int64_t row0, float* slab, int lane) {
    tile_fetch<D, NT>(e, table, N, ld, row0, lane);
    tile_transpose<D>(e, slab, lane);
}

// TransE, one query's partial sum over the 32 columns x[] = columns 32 s .. 32 s + 31 of the lane's row; c: the query's
// coefficient row (wave-uniform -> scalar loads, SGPR operands).  FIRST: column 0 starts the sum (score<false>).
template <int SIDE, int D, bool FIRST>
__device__ __forceinline__ float transe_piece_sum(float acc, const float (&x)[kSubCols], const float* __restrict__ c) {
#pragma unroll
    for (int k = 0; k < kSubCols; ++k) {
        float d;
        if constexpr (SIDE == TAIL) {
            d = c[k] - x[k];              // (h + r) - e, h + r hoisted
        } else {
            const float y = x[k] + c[k];  // (e + r) - t
            d = y - c[D + k];
        }
        acc = (FIRST && k == 0) ? fabsf(d) : acc + fabsf(d);
    }
    return acc;
}

}  // namespace blp
