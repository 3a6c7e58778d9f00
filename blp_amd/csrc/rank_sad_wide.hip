// rank_sad_wide.hip -- the fixed-point TransE pre-pass (rank_sad.hip) for ANY embedding width.
//
// The reference's bag-of-words and DKRL encoders score with TransE at the width of their word vectors
// (300 for GloVe, 768 for BERT word embeddings: models.py:118-135, 165-172, scripts/{glove,bert}-{bow,dkrl}-*.sh),
// which the register-resident kernels (D = 64 / 128 / 256) do not cover.  Same scheme, same bounds and
// the same exactness argument as rank_sad.hip; what changes:
//   * the width is a run-time value (D % 4 == 0, D <= 1024); the 2-byte images are padded with zeros to
//     a multiple of 128 elements (padding contributes |0 - 0| = 0 to every SAD and has no residual);
//   * a wave keeps 128 elements of its candidate tile in registers at a time and walks the row in
//     chunks: for each chunk it runs over the workgroup's queries, carrying the partial SADs in LDS;
//   * the exact scores (true keys, pair / tile refinement, CSR filter) come from one run-time-width
//     routine in the reference's operation order instead of the Scorer<> templates.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "exact_coop.h"
#include "knobs.h"
#include "launch.h"
#include "rank_common.h"
#include "sad_common.h"

#pragma clang fp contract(off)

namespace blp {

constexpr int kWChunk = 32;    // queries per workgroup (their partial SADs live in LDS: 8 KB per wave)
constexpr int kWQuota = 128;   // undecided pairs a workgroup can list (4 per query)
// Which workgroups run together on an XCD (its L2 holds 4 MB): windows of kWideGroupBlock candidate groups x kWideChunkBlock
// query chunks -- at D = 768 a group's four tile images are 393 KB, a chunk's query images 49 KB.  Rounds 2-3 walked ALL
// groups of a block of 4 chunks in turn: the ~128 workgroups an XCD holds at a time covered 32 groups x 4 chunks = 12.8 MB of
// distinct data, and 19 GB per launch went over the fabric for 83 GB read from the L2s (algorithmic: 0.7 GB).  A window of
// 4 groups x 32 chunks is 3.1 MB of distinct data for the same 128 workgroups.
#ifndef BLP_WIDE_CHUNK_BLOCK
#define BLP_WIDE_CHUNK_BLOCK 32
#endif
#ifndef BLP_WIDE_GROUP_BLOCK
#define BLP_WIDE_GROUP_BLOCK 4
#endif
constexpr unsigned kWideChunkBlock = BLP_WIDE_CHUNK_BLOCK;  // query chunks ...
constexpr unsigned kWideGroupBlock = BLP_WIDE_GROUP_BLOCK;  // ... x candidate groups whose workgroups run together on an XCD
constexpr int kWMaxD = 1024;

// models.py:222-223 for one (candidate, query) pair, any width: ((h + r) - t), |.|, sequential f32 sum
// from 0, negated.  head: the candidate replaces the head (fixed = t); else the tail (fixed = h).
__device__ __forceinline__ float transe_key_rt(const float* __restrict__ e, const float* __restrict__ f,
                                               const float* __restrict__ r, int D, bool head) {
    float acc = 0.0f;
    auto piece = [&](int d) {  // 16 elements: all twelve 16-byte loads are issued before the dependent adds
        float4 ev[4], fv[4], rv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            ev[j] = *reinterpret_cast<const float4*>(e + d + 4 * j);
            fv[j] = *reinterpret_cast<const float4*>(f + d + 4 * j);
            rv[j] = *reinterpret_cast<const float4*>(r + d + 4 * j);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float es[4] = {ev[j].x, ev[j].y, ev[j].z, ev[j].w}, fs[4] = {fv[j].x, fv[j].y, fv[j].z, fv[j].w},
                        rs[4] = {rv[j].x, rv[j].y, rv[j].z, rv[j].w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float x = (head ? es[i] : fs[i]) + rs[i];
                x = x - (head ? fs[i] : es[i]);
                acc = acc + fabsf(x);
            }
        }
    };
    int d = 0;
    for (; d + 16 <= D; d += 16) piece(d);
    for (; d < D; d += 4) {
        const float4 ev = *reinterpret_cast<const float4*>(e + d);
        const float4 fv = *reinterpret_cast<const float4*>(f + d);
        const float4 rv = *reinterpret_cast<const float4*>(r + d);
        const float es[4] = {ev.x, ev.y, ev.z, ev.w}, fs[4] = {fv.x, fv.y, fv.z, fv.w}, rs[4] = {rv.x, rv.y, rv.z, rv.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float x = (head ? es[i] : fs[i]) + rs[i];
            x = x - (head ? fs[i] : es[i]);
            acc = acc + fabsf(x);
        }
    }
    return -acc;
}

__global__ __launch_bounds__(256) void wide_true_key_kernel(int D,
                                                            const QRows q_fixed,
                                                            const QRows q_rel,
                                                            const QRows q_true, int64_t q_head, int64_t Q,
                                                            float* __restrict__ key_true,
                                                            unsigned long long* __restrict__ acc) {
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= Q) return;
    const float* e = q_true.row(q);
    key_true[q] = transe_key_rt(e, q_fixed.row(q), q_rel.row(q), D, q < q_head);
    acc[q] = 0;
}

// The same keys for a SMALL call, one WAVE per query: a lane that walks three rows on its own is one memory round trip
// per 16 elements (13 / 31 us at D = 300 / 768 whatever the number of queries).  Here the wave's lanes fetch the rows
// coalesced and compute the |differences| -- elementwise, any lane gets the reference's bits -- into LDS, and lane 0 adds
// them up in the reference's order, left to right.
constexpr int64_t kWTrueKeyWaveMaxQueries = 2048;
__global__ __launch_bounds__(256) void wide_true_key_wave_kernel(int D,
                                                                 const QRows q_fixed, const QRows q_rel,
                                                                 const QRows q_true, int64_t q_head, int64_t Q,
                                                                 float* __restrict__ key_true,
                                                                 unsigned long long* __restrict__ acc) {
    __shared__ __attribute__((aligned(16))) float terms[4][kWMaxD];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t q = (int64_t)blockIdx.x * 4 + wave;
    if (q >= Q) return;  // (wave-uniform; nothing below synchronises the workgroup)
    const float* e = q_true.row(q);
    const float* f = q_fixed.row(q);
    const float* r = q_rel.row(q);
    const bool head = q < q_head;
    for (int c = 4 * lane; c < D; c += 256) {
        const float4 ev = *reinterpret_cast<const float4*>(e + c);
        const float4 fv = *reinterpret_cast<const float4*>(f + c);
        const float4 rv = *reinterpret_cast<const float4*>(r + c);
        const float es[4] = {ev.x, ev.y, ev.z, ev.w}, fs[4] = {fv.x, fv.y, fv.z, fv.w}, rs[4] = {rv.x, rv.y, rv.z, rv.w};
        float t[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float x = (head ? es[i] : fs[i]) + rs[i];
            x = x - (head ? fs[i] : es[i]);
            t[i] = fabsf(x);
        }
        *reinterpret_cast<float4*>(&terms[wave][c]) = make_float4(t[0], t[1], t[2], t[3]);
    }
    wave_lds_sync();
    if (lane == 0) {
        float sum = 0.0f;
        for (int d = 0; d < D; d += 4) {
            const float4 t = *reinterpret_cast<const float4*>(&terms[wave][d]);
            sum = sum + t.x; sum = sum + t.y; sum = sum + t.z; sum = sum + t.w;
        }
        key_true[q] = -sum;
        acc[q] = 0;
    }
}

__global__ __launch_bounds__(256) void wide_range_kernel(const float* __restrict__ table, int64_t N, int64_t ld, int D,
                                                         const QRows q_fixed,
                                                         const QRows q_rel, int64_t q_head, int64_t Q,
                                                         SadParams* __restrict__ partial) {
    SadRange range;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x, t0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int quads = D / 4;
    for (int64_t i = t0; i < N * quads; i += stride) {
        const float4 v = *reinterpret_cast<const float4*>(table + (i / quads) * ld + (i % quads) * 4);
        range.see(v.x); range.see(v.y); range.see(v.z); range.see(v.w);
    }
    for (int64_t u = t0; u < Q * quads; u += stride) {  // four coefficients per step: one row lookup each side
        const bool head = u / quads < q_head;
        const float4 f = *reinterpret_cast<const float4*>(q_fixed.flat(4 * u, D));
        const float4 r = *reinterpret_cast<const float4*>(q_rel.flat(4 * u, D));
        range.see(sad_coef(f.x, r.x, head)); range.see(sad_coef(f.y, r.y, head));
        range.see(sad_coef(f.z, r.z, head)); range.see(sad_coef(f.w, r.w, head));
    }
    sad_range_block_store(range, partial);
}

// Candidate tile image as in rank_sad.hip, Dp / 8 uint4 per lane (Dp = D rounded up to 128, zero padded).
__device__ __forceinline__ void wide_quantize_table_tile(int64_t tile, const float* __restrict__ table, int64_t N, int64_t ld,
                                                         int D, int Dp, const SadParams* __restrict__ p,
                                                         uint4* __restrict__ cimg, unsigned* __restrict__ resid) {
    const SadScale sc = sad_scale(p);
    if (!sc.ok) return;
    const int row_in = threadIdx.x >> 2, part = threadIdx.x & 3;
    const int64_t row = tile * 64 + row_in;
    const float* src = table + (row < N ? row : 0) * ld;
    float res = 0.f;
    bool outside = false;  // a value the map does not cover: the row is left to the exact path
    for (int j4 = part; j4 < Dp / 8; j4 += 4) {
        unsigned w[4] = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int d = 8 * j4 + 4 * h;
            if (row < N && d < D) {
                const float4 v = *reinterpret_cast<const float4*>(src + d);
                w[2 * h] = sad_quant(v.x, sc, res, outside);
                w[2 * h] |= sad_quant(v.y, sc, res, outside) << 16;
                w[2 * h + 1] = sad_quant(v.z, sc, res, outside);
                w[2 * h + 1] |= sad_quant(v.w, sc, res, outside) << 16;
            }
        }
        cimg[(tile * (Dp / 8) + j4) * 64 + row_in] = make_uint4(w[0], w[1], w[2], w[3]);
    }
    res += __shfl_xor(res, 1);
    res += __shfl_xor(res, 2);
    outside |= (bool)__shfl_xor((int)outside, 1);
    outside |= (bool)__shfl_xor((int)outside, 2);
    if (part == 0) resid[row] = outside ? kSRowExact : (unsigned)(res * 1.0001f + kSResidSlack * D) + 1u;
}

// One wave per query: its 2-byte image (Dp / 2 dwords) and the two thresholds; the first candidate slab's
// flag bitmap is zeroed on the way.
__device__ __forceinline__ void wide_quantize_queries_block(int64_t vblock, int64_t vgrid, const QRows& q_fixed,
                                                            const QRows& q_rel, int64_t q_head,
                                                            int64_t Q, int D, int Dp,
                                                            const float* __restrict__ key_true,
                                                            const SadParams* __restrict__ p,
                                                            unsigned* __restrict__ qimg, int2* __restrict__ thr,
                                                            unsigned* __restrict__ flags, int64_t n_flag_words) {
    for (int64_t j = vblock * 256 + threadIdx.x; j < n_flag_words; j += vgrid * 256) flags[j] = 0;
    const SadScale sc = sad_scale(p);
    if (!sc.ok) return;
    const int64_t q = vblock * 4 + (threadIdx.x >> 6);
    if (q >= Q) return;
    const int lane = threadIdx.x & 63;
    const bool head = q < q_head;
    const float* f = q_fixed.row(q);
    const float* r = q_rel.row(q);
    float res = 0.f, qmax = 0.f;
    bool outside = false;  // a coefficient the map does not cover: the query is left to the exact path
    for (int j = lane; j < Dp / 2; j += 64) {
        unsigned w = 0;
        if (2 * j < D) {  // D is even: both elements exist or neither
            w = sad_quant(sad_coef(f[2 * j], r[2 * j], head), sc, res, outside);
            w |= sad_quant(sad_coef(f[2 * j + 1], r[2 * j + 1], head), sc, res, outside) << 16;
            qmax = fmaxf(qmax, fmaxf(fmaxf(fabsf(f[2 * j]), fabsf(f[2 * j + 1])), fmaxf(fabsf(r[2 * j]), fabsf(r[2 * j + 1]))));
        }
        qimg[q * (Dp / 2) + j] = w;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        res += __shfl_xor(res, off);
        qmax = fmaxf(qmax, __shfl_xor(qmax, off));
        outside |= (bool)__shfl_xor((int)outside, off);
    }
    if (lane == 0) {
        const double u = 5.9604644775390625e-8;
        const double dt = -(double)key_true[q], s = (double)sc.scale;
        const double M = (double)fmaxf(sad_range_maxabs(sc), qmax);  // every value a decided pair of this query involves
        const double band = (double)res * 1.0001 + 2.0 * kSResidSlack * D + 1.0;  // E_q + both sides' slack
        const double gamma = 1.05 * (D + 2) * u, rho = 6.0 * D * u * M;
        int t_lo = -1, t_hi = (int)kSThrMax;  // nothing decided
        if (!outside && M < 3.0e38 && dt >= 0.0 && dt < 1.0e300) {  // false for NaN
            const double lo_v = s * (dt - rho) / (1.0 + gamma) - band;   // SAD + E_c below this: certainly above
            const double hi_v = s * (dt + rho) / (1.0 - gamma) + band;   // SAD - E_c above this: certainly below
            if (lo_v > 2.0) t_lo = lo_v - 1.0 < (double)kSThrMax ? (int)(lo_v - 1.0) : (int)kSThrMax;
            if (hi_v + 2.0 < (double)kSThrMax) t_hi = (int)(hi_v + 2.0);
        }
        thr[q] = make_int2(t_lo, t_hi);
    }
}

// Both images in one launch (as rank_sad.hip's sad_quantize_kernel): workgroups [0, table_blocks) a candidate tile each,
// the rest four queries each; with n_partial > 0 (small calls) every workgroup first finishes the range pass itself.
__global__ __launch_bounds__(256) void wide_quantize_kernel(const float* __restrict__ table, int64_t N, int64_t ld, int D, int Dp,
                                                            SadParams* __restrict__ p, const SadParams* __restrict__ partial,
                                                            int n_partial, uint4* __restrict__ cimg,
                                                            unsigned* __restrict__ resid, unsigned table_blocks,
                                                            const QRows q_fixed, const QRows q_rel, int64_t q_head, int64_t Q,
                                                            const float* __restrict__ key_true, unsigned* __restrict__ qimg,
                                                            int2* __restrict__ thr, unsigned* __restrict__ flags,
                                                            int64_t n_flag_words) {
    __shared__ SadParams p_block;
    const SadParams* pp = p;
    if (n_partial > 0) {
        if (threadIdx.x < 64) {
            const SadParams r = sad_range_finish_wave(partial, n_partial, threadIdx.x);
            if (threadIdx.x == 0) {
                p_block = r;
                if (blockIdx.x == 0) *p = r;
            }
        }
        __syncthreads();
        pp = &p_block;
    }
    if (blockIdx.x < table_blocks)
        wide_quantize_table_tile(blockIdx.x, table, N, ld, D, Dp, pp, cimg, resid);
    else
        wide_quantize_queries_block(blockIdx.x - table_blocks, gridDim.x - table_blocks, q_fixed, q_rel, q_head, Q, D, Dp, key_true,
                                    pp, qimg, thr, flags, n_flag_words);
}

// ------------------------------------------------------------------------------------------------
// Pass 1.  One candidate tile per wave; the row is walked in chunks of 128 elements (64 VGPRs), the
// workgroup's queries run inside each chunk and their partial SADs wait in LDS between chunks.
__global__ __launch_bounds__(kSW * 64, 4) void wide_rank_sad_kernel(
    const uint4* __restrict__ cimg, const unsigned* __restrict__ resid, int64_t n_rows, int n_groups,
    const unsigned* __restrict__ qimg, const int2* __restrict__ thr, int64_t Q, int Dp, int words_per_query,
    unsigned long long* __restrict__ acc, unsigned* __restrict__ flags, uint2* __restrict__ pairs,
    unsigned* __restrict__ pair_cnt, SadParams* __restrict__ params, int q_per_wg) {  // q_per_wg <= kWChunk: wide_chunk_queries()
    if (!sad_scale(params).ok) return;
    __shared__ unsigned psum[kSW][kWChunk][64];
    __shared__ int2 thr_s[kWChunk];
    __shared__ unsigned cnt[kWChunk];
    __shared__ uint2 pair_s[kWQuota];
    __shared__ unsigned pair_n;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    // Consecutive workgroups share the candidate group: its four tile images (64 rows x Dp x 2 B each, 393 KB at
    // D = 768) are what a workgroup streams chunk by chunk, and with the query chunk as the fast index every one of the
    // 3 305 query chunks of the FB15k-237 block fetched them again from the fabric (77 GB per launch, 111 x the
    // algorithmic bytes); now they stay in the L2s while the group's chunks go by.
    // (round 2, later: each XCD -- blocks go to them round-robin, b % 8 -- walks a contiguous range of a logical index
    // laid out as (block of kWideChunkBlock query chunks) x (candidate group) x (chunk of the block), as in rank_sad.hip:
    // a query chunk's image is read by one XCD only, a group's tile images once per block of chunks)
    const unsigned n_blocks = gridDim.x, xcd = blockIdx.x & 7u, per_xcd = n_blocks >> 3, rem_b = n_blocks & 7u;
    const unsigned logical = (xcd < rem_b ? xcd * (per_xcd + 1) : rem_b * (per_xcd + 1) + (xcd - rem_b) * per_xcd) + (blockIdx.x >> 3);
    const unsigned n_chunks_all = n_blocks / (unsigned)n_groups;
    // logical index = (block of kWideChunkBlock chunks) x (block of kWideGroupBlock groups) x (group of the block) x (chunk of the block)
    const unsigned per_cb = (unsigned)n_groups * kWideChunkBlock, cb = logical / per_cb, in_cb = logical % per_cb;
    const unsigned chunks_here = n_chunks_all - cb * kWideChunkBlock < kWideChunkBlock ? n_chunks_all - cb * kWideChunkBlock : kWideChunkBlock;
    const unsigned per_gb = kWideGroupBlock * chunks_here, gb = in_cb / per_gb, in_gb = in_cb % per_gb;
    const int group = (int)(gb * kWideGroupBlock + in_gb / chunks_here);  // (the last block of groups may be short: in_gb stops early)
    const int64_t q0 = (int64_t)(cb * kWideChunkBlock + in_gb % chunks_here) * q_per_wg;
    const int nq = (int)(Q - q0 < q_per_wg ? Q - q0 : q_per_wg);

    if (tid < kWChunk) {
        thr_s[tid] = tid < nq ? thr[q0 + tid] : make_int2(0, 0);
        cnt[tid] = 0;
    }
    if (tid < kWQuota) pair_s[tid] = make_uint2(kSNoPair, 0u);
    if (tid == 0) pair_n = 0;
    __syncthreads();

    const int64_t n_tiles = (n_rows + 63) / 64;
    const int64_t tile_id = (int64_t)group * kSW + wave;
    const unsigned bias = tile_id * 64 + lane < n_rows ? 0u : kSInvalid;
    const int64_t tile = tile_id < n_tiles ? tile_id : n_tiles - 1;
    const unsigned ec_raw = resid[tile * 64 + lane];
    const unsigned long long exact_rows = __ballot(bias == 0u && ec_raw == kSRowExact);  // rows marked exact-only
    const unsigned ec = ec_raw == kSRowExact ? 0u : ec_raw;
    const int n_ch = Dp / 128, row_dwords = Dp / 2;

    for (int ch = 0; ch < n_ch; ++ch) {
        unsigned v[64];
        const uint4* src = cimg + (tile * (Dp / 8) + ch * 16) * 64 + lane;
        static_for<16>([&](auto jj) {
            constexpr int j = decltype(jj)::value;
            const uint4 x = src[j * 64];
            v[4 * j] = x.x; v[4 * j + 1] = x.y; v[4 * j + 2] = x.z; v[4 * j + 3] = x.w;
        });
        const bool first = ch == 0, last = ch + 1 == n_ch;
        // (scalar work per query kept small -- a CU has one scalar ALU for four SIMDs, see rank_sad.hip: pointers step
        //  by a constant over an image padded by four rows, counts go to lane j of a register)
        const float* row = reinterpret_cast<const float*>(qimg + q0 * row_dwords + ch * 64);
        sf16 cur = sload16<0>(row);
        sdrain(cur);
        unsigned cnt_reg = 0;  // lane j: certainly-above count of query q0 + j for this wave's tile
        unsigned long long bit = 1;
        unsigned carried = first ? bias : psum[wave][0][lane];  // the next query's partial SAD, read one query ahead
        for (int j = 0; j < nq; ++j, row += row_dwords, bit <<= 1) {
            const float* next_row = row + row_dwords;
            unsigned sad = carried;
            if (!first && j + 1 < nq) carried = psum[wave][j + 1][lane];  // its LDS latency hides behind this query's SADs
            static_for<4>([&](auto kk) {
                constexpr int k = decltype(kk)::value;
                sf16 nxt;
                if constexpr (k + 1 < 4) nxt = sload16<(k + 1) * 64>(row); else nxt = sload16<0>(next_row);
                if constexpr (k == 0) { if ((j & 3) == wave) stouch<256>(row + 3 * (size_t)row_dwords); }
                static_for<16>([&](auto ii) {
                    constexpr int i = decltype(ii)::value;
                    const float qf = cur[i];
                    sad = __builtin_amdgcn_sad_u16(__float_as_uint(qf), v[16 * k + i], sad);
                });
                sdrain(nxt);
                cur = nxt;
            });
            if (!last) {
                psum[wave][j][lane] = sad;
                continue;
            }
            const int2 th = thr_s[j];
            const unsigned long long above = __ballot((int)(sad + ec) < th.x) & ~exact_rows;
            const unsigned long long und = (__ballot((int)(sad - ec) <= th.y) | exact_rows) & ~above;
            unsigned n_above = __popcll(above);
            if (und) {  // wave-uniform
                const unsigned n = __popcll(und);
                unsigned slot = 0;
                if (lane == 0) slot = atomicAdd(&pair_n, n);
                slot = __builtin_amdgcn_readfirstlane(slot);
                if (slot + n <= kWQuota) {
                    if ((und >> lane) & 1ull)
                        pair_s[slot + __popcll(und & ((1ull << lane) - 1ull))] =
                            make_uint2((unsigned)(q0 + j), (unsigned)(tile_id * 64 + lane));
                } else {
                    if (lane == 0)
                        atomicOr(flags + (size_t)(q0 + j) * words_per_query + (tile_id >> 5), 1u << (tile_id & 31));
                    n_above = 0;
                }
            }
            unsigned long long saved;  // cnt_reg[lane j] += n_above: one VALU instruction under a one-lane exec mask
            asm volatile("s_mov_b64 %1, exec\n\ts_mov_b64 exec, %3\n\tv_add_u32 %0, %0, %2\n\ts_mov_b64 exec, %1"
                         : "+v"(cnt_reg), "=&s"(saved) : "s"(n_above), "s"(bit));
        }
        if (last && lane < kWChunk && cnt_reg) atomicAdd(&cnt[lane], cnt_reg);
    }
    __syncthreads();
    if (tid < nq) {
        const unsigned long long c = cnt[tid];
        if (c) atomicAdd(acc + q0 + tid, c | (c << 32));  // certainly above: gt and ge
    }
    // the workgroup's pairs go to ITS region of the list -- (query chunk, candidate group), kWQuota slots, the fill beside it --
    // so that the refinement finds a chunk's pairs together and can group them by query (wide_refine_chunks_kernel)
    const unsigned used = pair_n < (unsigned)kWQuota ? pair_n : (unsigned)kWQuota;
    const size_t region = (size_t)(q0 / q_per_wg) * n_groups + group;
    if (tid == 0) pair_cnt[region] = used;
    for (unsigned i = tid; i < used; i += kSW * 64) pairs[region * kWQuota + i] = pair_s[i];
}

// Pass 2a: the listed pairs, grouped by query.  Rounds 2-3 re-scored the pairs in the order the pre-pass listed them, one
// lane per pair, every lane gathering its candidate row AND its query's two vectors: three 3-KB rows per pair at D = 768,
// 26 GB over the fabric per launch -- 3.5 ms at 7.4 TB/s, bandwidth-bound.  A workgroup now takes the pairs of ONE query chunk
// (<= 32 queries) from kWSliceGroups candidate groups' regions, sorts them by query in LDS (a histogram, a scan, a scatter of
// the candidate rows), and its waves then re-score 64 candidates of one query at a time: the query's vectors come through
// the scalar cache once per wave (exact_coop.h: transe_key_64_one_query_rt), only the candidate rows are gathered.
constexpr int kWSliceGroups = 32;                       // regions a workgroup sorts at a time ...
constexpr int kWSliceCap = kWSliceGroups * kWQuota;     // ... i.e. at most this many pairs (16 KB of LDS)
__global__ __launch_bounds__(256) void wide_refine_chunks_kernel(const float* __restrict__ table, int64_t ld, int D,
                                                                const QRows q_fixed, const QRows q_rel,
                                                                const float* __restrict__ key_true, int64_t q_head, int64_t Q,
                                                                const uint2* __restrict__ pairs, const unsigned* __restrict__ pair_cnt,
                                                                int n_groups, int n_slices, int q_per_wg,
                                                                const SadParams* __restrict__ params,
                                                                unsigned long long* __restrict__ acc) {
    if (!sad_scale(params).ok) return;  // the pre-pass did not run: nothing was listed (the tile sweep takes every query)
    __shared__ __attribute__((aligned(16))) float slabs[4][64 * kRefStride];
    extern __shared__ __attribute__((aligned(16))) float qvec[];  // per wave: 2 x Dp floats, the query side of its current task
    __shared__ unsigned sorted[kWSliceCap];
    __shared__ unsigned hist[kWChunk], start[kWChunk + 1], cnt_s[kWSliceGroups];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t chunk = blockIdx.x / n_slices;
    const int g0 = (int)(blockIdx.x % n_slices) * kWSliceGroups;
    const int ng = n_groups - g0 < kWSliceGroups ? n_groups - g0 : kWSliceGroups;
    const int64_t q0 = chunk * q_per_wg;
    const size_t region0 = (size_t)chunk * n_groups + g0;
    if (tid < kWChunk) hist[tid] = 0;
    if (tid < kWSliceGroups) cnt_s[tid] = tid < ng ? pair_cnt[region0 + tid] : 0u;
    __syncthreads();
    // 1. histogram over the chunk's queries; a thread keeps its <= 16 pairs and their ranks inside their bins
    constexpr int kSlots = kWSliceCap / 256;
    unsigned cand[kSlots], where[kSlots];
#pragma unroll
    for (int k = 0; k < kSlots; ++k) {
        const int slot = tid + 256 * k, r = slot / kWQuota, i = slot % kWQuota;
        where[k] = 0xFFFFFFFFu;
        cand[k] = 0;
        if (i < (int)cnt_s[r]) {  // (cnt_s[r] == 0 for r >= ng)
            const uint2 p = pairs[(region0 + r) * kWQuota + i];
            const unsigned ql = p.x - (unsigned)q0;
            if (ql < (unsigned)kWChunk) {
                cand[k] = p.y;
                where[k] = (ql << 16) | atomicAdd(&hist[ql], 1u);  // (at most kWSliceCap = 4 096 per bin)
            }
        }
    }
    __syncthreads();
    // 2. exclusive scan of the 32 bins
    if (tid < 64) {
        unsigned incl = lane < kWChunk ? hist[lane] : 0u;
#pragma unroll
        for (int off = 1; off < kWChunk; off <<= 1) {
            const unsigned up = __shfl_up(incl, off);
            if (lane >= off) incl += up;
        }
        if (lane < kWChunk) start[lane + 1] = incl;
        if (lane == 0) start[0] = 0;
    }
    __syncthreads();
    // 3. the candidate rows, sorted by query
#pragma unroll
    for (int k = 0; k < kSlots; ++k)
        if (where[k] != 0xFFFFFFFFu) sorted[start[where[k] >> 16] + (where[k] & 0xFFFFu)] = cand[k];
    __syncthreads();
    // 4. tasks (query, 64 of its candidates), round-robin over the waves
    float* slab = slabs[wave];
    const int Dp = (D + 127) / 128 * 128;
    float* qa = qvec + (size_t)wave * 2 * Dp;
    float* qb = qa + Dp;
    int task = 0;
    for (int j = 0; j < kWChunk; ++j) {  // wave-uniform throughout
        const unsigned lo = start[j], hi = start[j + 1];
        if (lo == hi) continue;
        const int64_t q = q0 + j;
        for (unsigned b = lo; b < hi; b += 64, ++task) {
            if ((task & 3) != wave) continue;
            const bool live = b + lane < hi;
            const unsigned row = sorted[live ? b + lane : lo];
            const bool head = q < q_head;
            wave_lds_sync();  // the previous task's reads of the query side are done
            stage_query_rt(q_fixed.row(q), q_rel.row(q), D, head, qa, qb, lane);
            wave_lds_sync();
            const float key = transe_key_64_one_query_rt(table + (int64_t)row * ld, qa, qb, D, head, slab, lane);
            const float kt = key_true[q];
            const unsigned long long gt = __popcll(__ballot(live && key > kt)), ge = __popcll(__ballot(live && key >= kt));
            if (lane == 0 && ge) atomicAdd(acc + q, gt | (ge << 32));
        }
    }
}

// Pass 2b: one wave per query over its flagged 64-candidate tiles (all tiles when the pre-pass did not run).
__global__ __launch_bounds__(256) void wide_refine_tiles_kernel(const float* __restrict__ table, int64_t n_rows,
                                                                int64_t ld, int D, const QRows q_fixed,
                                                                const QRows q_rel,
                                                                const float* __restrict__ key_true, int64_t q_head,
                                                                int64_t Q, int words_per_query,
                                                                const unsigned* __restrict__ flags,
                                                                const SadParams* __restrict__ params,
                                                                unsigned long long* __restrict__ acc) {
    __shared__ int list[kSweepQueries], n_list;
    const bool all = !sad_scale(params).ok;
    const int lane = threadIdx.x & 63;
    const int64_t q_base = (int64_t)blockIdx.x * kSweepQueries;
    const int n = flagged_queries(flags, q_base, Q, words_per_query, all, list, &n_list);
    for (int i = threadIdx.x >> 6; i < n; i += 4) {
    const int64_t q = q_base + list[i];
    const unsigned* row = flags + q * words_per_query;
    const float kt = key_true[q];
    const float* f = q_fixed.row(q);
    const float* r = q_rel.row(q);
    const bool head = q < q_head;
    unsigned gt = 0, ge = 0;
    for (int w0 = 0; w0 < words_per_query; w0 += 64) {
        const unsigned mine = w0 + lane < words_per_query ? (all ? 0xFFFFFFFFu : row[w0 + lane]) : 0u;
        unsigned long long nonzero = __ballot(mine != 0);
        while (nonzero) {
            const int src = __builtin_ctzll(nonzero);
            nonzero &= nonzero - 1;
            unsigned bits = __shfl(mine, src);
            const int64_t tile_base = (int64_t)(w0 + src) * 32;
            while (bits) {
                const int b = __builtin_ctz(bits);
                bits &= bits - 1;
                const int64_t cand = (tile_base + b) * 64 + lane;
                const bool ok = cand < n_rows;
                if (__ballot(ok) == 0) continue;
                const float key = transe_key_rt(table + (ok ? cand : 0) * ld, f, r, D, head);
                gt += ok && key > kt;
                ge += ok && key >= kt;
            }
        }
    }
    if (__ballot(gt | ge) == 0) continue;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        gt += __shfl_down(gt, off);
        ge += __shfl_down(ge, off);
    }
    if (lane == 0) atomicAdd(acc + q, (unsigned long long)gt | ((unsigned long long)ge << 32));
    }
}

// Last kernel of a call (as rank_all.hip's filter_finalize_kernel, run-time width): a workgroup owns 64 queries, its
// waves take the ones that have filter entries in turn, thread q writes query q's four counts.
__global__ __launch_bounds__(256) void wide_filter_finalize_kernel(const float* __restrict__ table, int64_t N, int64_t ld,
                                                                   int D, const QRows q_fixed,
                                                                   const QRows q_rel,
                                                                   const float* __restrict__ key_true, int64_t q_head,
                                                                   int64_t Q, const FilterSpec filter,
                                                                   const unsigned long long* __restrict__ acc,
                                                                   int32_t* __restrict__ counts) {
    __shared__ int list[kSweepQueries], n_list;
    __shared__ unsigned removed[kSweepQueries][2];
    const int64_t q_base = (int64_t)blockIdx.x * kSweepQueries;
    const int lane = threadIdx.x & 63;
    if (threadIdx.x < 64) {
        const int64_t q = q_base + lane;
        const bool any = filter.on() && q < Q && filter.hi[q] > filter.lo[q];
        removed[lane][0] = removed[lane][1] = 0;
        const unsigned long long mask = __ballot(any);
        if (any) list[__popcll(mask & ((1ull << lane) - 1ull))] = lane;
        if (lane == 0) n_list = __popcll(mask);
    }
    __syncthreads();
    for (int i = threadIdx.x >> 6; i < n_list; i += 4) {
        const int slot = list[i];
        const int64_t q = q_base + slot;
        const float kt = key_true[q];
        unsigned gt = 0, ge = 0;
        for (int64_t k = filter.lo[q] + lane; k < filter.hi[q]; k += 64) {
            const int64_t row = filter_row(filter, q, k, N);
            if (row < 0) continue;
            const float key = transe_key_rt(table + row * ld, q_fixed.row(q), q_rel.row(q), D, q < q_head);
            gt += key > kt;
            ge += key >= kt;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            gt += __shfl_down(gt, off);
            ge += __shfl_down(ge, off);
        }
        if (lane == 0) { removed[slot][0] = gt; removed[slot][1] = ge; }
    }
    __syncthreads();
    if (threadIdx.x < 64 && q_base + lane < Q) {
        const int64_t q = q_base + lane;
        const unsigned long long a = acc[q];
        const int32_t all_gt = (int32_t)(a & 0xffffffffull), all_ge = (int32_t)(a >> 32);
        reinterpret_cast<int4*>(counts)[q] =
            make_int4(all_gt, all_ge, all_gt - (int32_t)removed[lane][0], all_ge - (int32_t)removed[lane][1]);
    }
}

// ------------------------------------------------------------------------------------------------
struct WideWorkspace {
    float* key_true;
    unsigned long long* acc;
    SadParams* params; SadParams* partial; int2* thr; unsigned* qimg; uint4* cimg; unsigned* resid; unsigned* flags;
    uint2* pairs;
    unsigned* pair_cnt;  // fill of every (query chunk, candidate group) region of `pairs`
    int64_t pass_groups;
    size_t bytes;
};

static int padded_width(int D) { return (D + 127) / 128 * 128; }

// Queries per workgroup of the pre-pass: kWChunk, fewer when (candidate groups) x (query chunks) would not fill the chip
// -- the BOW / DKRL scripts' eval batches: 64 queries against 14 541 rows were 114 workgroups ([measured] D = 300:
// 38.6 us for the pre-pass kernel of a 110 us call).
static int wide_chunk_queries(int64_t N, int64_t Q) {
    const int64_t n_groups = ((N + 63) / 64 + kSW - 1) / kSW;
    int cs = kWChunk;
    while (cs > 8 && n_groups * ((Q + cs - 1) / cs) < 2 * 256) cs >>= 1;
    return cs;
}

static int64_t wide_groups_per_pass(int64_t N, int64_t Q) {
    const int64_t n_groups = ((N + 63) / 64 + kSW - 1) / kSW;
    const int cs = wide_chunk_queries(N, Q);
    const int64_t n_chunks = (Q + cs - 1) / cs;
    const int64_t cap = (int64_t)256 << 20;
    const int64_t by_pairs = cap / (kWQuota * 8) / (n_chunks > 0 ? n_chunks : 1);
    const int64_t by_flags = cap / 4 / (Q > 0 ? Q : 1) * 32 / kSW;
    int64_t g = n_groups;
    if (g > by_pairs) g = by_pairs;
    if (g > by_flags) g = by_flags;
    if (const int64_t forced = knob(KNOB_SAD_PASS_GROUPS))  // test knob: force the multi-slab path
        if (forced > 0 && forced < g) g = forced;
    return g < 1 ? 1 : g;
}

static WideWorkspace carve_wide(void* base, int D, int64_t N, int64_t Q) {
    WideWorkspace w;
    const int Dp = padded_width(D);
    char* p = static_cast<char*>(base);
    size_t off = 0;
    w.key_true = reinterpret_cast<float*>(p + off);  off = align_up(off + (size_t)Q * 4, 256);
    w.acc = reinterpret_cast<unsigned long long*>(p + off);   off = align_up(off + (size_t)Q * 8, 256);
    w.params = reinterpret_cast<SadParams*>(p + off);  off = align_up(off + sizeof(SadParams), 256);
    w.partial = reinterpret_cast<SadParams*>(p + off); off = align_up(off + sizeof(SadParams) * kSRangeBlocks, 256);
    w.thr = reinterpret_cast<int2*>(p + off);          off = align_up(off + (size_t)Q * 8, 256);
    // + 4 rows: the pre-pass reads ahead of its query (next row, the row three ahead) without clamping at the end
    w.qimg = reinterpret_cast<unsigned*>(p + off);     off = align_up(off + (size_t)(Q + 4) * (Dp / 2) * 4 + 64, 256);
    w.cimg = reinterpret_cast<uint4*>(p + off);        off = align_up(off + (size_t)((N + 63) / 64) * 64 * (Dp / 2) * 4, 256);
    w.resid = reinterpret_cast<unsigned*>(p + off);    off = align_up(off + (size_t)((N + 63) / 64) * 64 * 4, 256);
    w.pass_groups = wide_groups_per_pass(N, Q);
    const int64_t words = (w.pass_groups * kSW + 31) / 32;
    w.flags = reinterpret_cast<unsigned*>(p + off); off = align_up(off + (size_t)Q * words * 4, 256);
    w.pairs = reinterpret_cast<uint2*>(p + off);
    const int cs = wide_chunk_queries(N, Q);
    off = align_up(off + (size_t)w.pass_groups * ((Q + cs - 1) / cs) * kWQuota * 8, 256);
    w.pair_cnt = reinterpret_cast<unsigned*>(p + off);
    off = align_up(off + (size_t)w.pass_groups * ((Q + cs - 1) / cs) * 4, 256);
    w.bytes = off;
    return w;
}

// TransE at a width the register-resident kernels are not compiled for.  Any number of queries: the alternative for a
// small block is the dense route (blp_score_fwd + blp_rank_from_scores), 0.7 ms for 32 queries x 14 541 rows at D = 300
// against 0.1 ms here (tools/wide_small_probe.py) -- round 1 drew the line at 256 queries without measuring it.
bool rank_sad_wide_applicable(int model, int D, int64_t q_head, int64_t q_tail) {
    return model == TRANSE && D != 64 && D != 128 && D != 256 && D >= 4 && D % 4 == 0 && D <= kWMaxD && q_head + q_tail >= 1;
}

size_t rank_sad_wide_workspace_bytes(int model, int D, int64_t N, int64_t q_head, int64_t q_tail) {
    if (!rank_sad_wide_applicable(model, D, q_head, q_tail)) return 0;
    return carve_wide(nullptr, D, N, q_head + q_tail).bytes;
}

hipError_t launch_rank_all_sad_wide(int D, const float* table, int64_t N, int64_t ld, const QRows q_fixed,
                                    const QRows q_rel, const QRows q_true, int64_t q_head,
                                    int64_t q_tail, const FilterSpec& filter,
                                    int32_t* counts, void* workspace, int n_cu, hipStream_t stream,
                                    hipEvent_t ev_start, hipEvent_t ev_stop) {
    const int64_t Q = q_head + q_tail;
    const int Dp = padded_width(D);
    WideWorkspace w = carve_wide(workspace, D, N, Q);
    if (Q <= kWTrueKeyWaveMaxQueries)
        wide_true_key_wave_kernel<<<dim3((unsigned)((Q + 3) / 4)), 256, 0, stream>>>(D, q_fixed, q_rel, q_true,
                                                                                  q_head, Q, w.key_true, w.acc);
    else
        wide_true_key_kernel<<<dim3((unsigned)((Q + 255) / 256)), 256, 0, stream>>>(D, q_fixed, q_rel, q_true,
                                                                                   q_head, Q, w.key_true, w.acc);
    if (ev_start) (void)hipEventRecord(ev_start, stream);
    const int64_t n_tiles = (N + 63) / 64, query_blocks = (Q + 3) / 4;
    if (n_tiles + query_blocks > 0x7fffffff) return hipErrorInvalidValue;
    // small calls: no launch for the range's last step -- the quantising workgroups do it themselves (from fewer partials)
    const bool fold_finish = n_tiles + query_blocks <= 1024;
    int64_t range_blocks;
    {
        const int64_t items = N * (D / 4) > Q * D ? N * (D / 4) : Q * D, cap = fold_finish ? 512 : kSRangeBlocks;
        range_blocks = (items + 255) / 256;
        range_blocks = range_blocks < cap ? (range_blocks > 0 ? range_blocks : 1) : cap;
        wide_range_kernel<<<dim3((unsigned)range_blocks), 256, 0, stream>>>(table, N, ld, D, q_fixed, q_rel, q_head, Q, w.partial);
        if (!fold_finish) sad_range_finish_kernel<<<1, 64, 0, stream>>>(w.partial, (int)range_blocks, w.params);
    }
    const int64_t pass_rows = w.pass_groups * kSW * 64;
    const int64_t first_rows = N < pass_rows ? N : pass_rows;
    const int64_t first_words = ((((first_rows + 63) / 64 + kSW - 1) / kSW) * kSW + 31) / 32;
    wide_quantize_kernel<<<dim3((unsigned)(n_tiles + query_blocks)), 256, 0, stream>>>(
        table, N, ld, D, Dp, w.params, w.partial, fold_finish ? (int)range_blocks : 0, w.cimg, w.resid, (unsigned)n_tiles, q_fixed,
        q_rel, q_head, Q, w.key_true, w.qimg, w.thr, w.flags, first_words * Q);

    const int q_per_wg = wide_chunk_queries(N, Q);
    const int64_t n_chunks = (Q + q_per_wg - 1) / q_per_wg;
    hipError_t err = hipSuccess;
    for (int64_t slab0 = 0; slab0 < N; slab0 += pass_rows) {  // one iteration unless the caps bind
        const int64_t n_rows = N - slab0 < pass_rows ? N - slab0 : pass_rows;
        const int64_t n_groups = ((n_rows + 63) / 64 + kSW - 1) / kSW;
        const int words = (int)((n_groups * kSW + 31) / 32);
        const int64_t n_blocks = n_groups * n_chunks;
        if (slab0 > 0) {
            err = hipMemsetAsync(w.flags, 0, (size_t)Q * words * 4, stream);
            if (err != hipSuccess) return err;
            err = hipMemsetAsync(&w.params->n_pairs, 0, 4, stream);
            if (err != hipSuccess) return err;
        }
        const float* slab = table + slab0 * ld;
        wide_rank_sad_kernel<<<dim3((unsigned)n_blocks), kSW * 64, 0, stream>>>(
            w.cimg + (slab0 / 64) * (Dp / 8) * 64, w.resid + slab0, n_rows, (int)n_groups, w.qimg, w.thr, Q, Dp, words, w.acc,
            w.flags, w.pairs, w.pair_cnt, w.params, q_per_wg);
        const int64_t n_slices = (n_groups + kWSliceGroups - 1) / kWSliceGroups;
        if (n_chunks * n_slices > 0x7fffffff) return hipErrorInvalidValue;
        wide_refine_chunks_kernel<<<dim3((unsigned)(n_chunks * n_slices)), 256, (size_t)4 * 2 * Dp * 4, stream>>>(
            slab, ld, D, q_fixed, q_rel, w.key_true, q_head, Q, w.pairs, w.pair_cnt, (int)n_groups, (int)n_slices, q_per_wg, w.params,
            w.acc);
        wide_refine_tiles_kernel<<<dim3((unsigned)((Q + kSweepQueries - 1) / kSweepQueries)), 256, 0, stream>>>(
            slab, n_rows, ld, D, q_fixed, q_rel, w.key_true, q_head, Q, words, w.flags, w.params, w.acc);
    }
    if (ev_stop) (void)hipEventRecord(ev_stop, stream);
    wide_filter_finalize_kernel<<<dim3((unsigned)((Q + kSweepQueries - 1) / kSweepQueries)), 256, 0, stream>>>(table, N, ld, D, q_fixed, q_rel, w.key_true,
                                                                                  q_head, Q, filter, w.acc, counts);
    return hipGetLastError();
}

}  // namespace blp
