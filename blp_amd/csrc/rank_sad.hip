// rank_sad.hip -- all-entities ranking for TransE with many queries: a 16-bit fixed-point pre-pass on
// v_sad_u16 with a rigorous error band, then exact refinement of the undecided pairs.
// (SURVEY.md 8a "K1"; replaces train.py:146-171 + utils.py:103-105 of the reference for rel_model=transe.)
//
// The exact TransE key costs two f32 VALU instructions per element (subtract, |x| accumulate) and the
// f32 VALU pipe is the roof of the exact kernel (rank_all.hip: 61 % of it at FB15k-237).  There is no
// matrix-core form of an L1 norm, but gfx950 has v_sad_u16: D = S2 + |S0.lo - S1.lo| + |S0.hi - S1.hi|,
// two elements per instruction at 4.2 cycles (tools/sad_ubench.hip) against 2 x 2 x 2.43 for the f32
// pair: 2.3x fewer VALU cycles per element.  So, as for the bilinear models (rank_gemm.hip):
//   prep    one affine map x^ = rint((x - lo) s) in [0, 65535] for the whole call (lo, hi = range of the
//           table and of the query coefficients c = h + r | t - r); candidate rows are packed two
//           elements per dword into a lane-major tile image, query coefficients into 2-byte rows.
//   pass 1  SAD(q, c) = sum_k |c^_k - e^_k| (exact integer), so |SAD - s X| <= E_q + E_c for the real-number
//           distance X = sum_k |c_k - e_k|, where E_row = sum_k |x^_k - (x_k - lo) s| is the row's own
//           rounding residual (~D/4, at most 0.508 D; measured while quantising, rounded up, kept per
//           candidate row and folded into the thresholds per query).  The reference score S_ref is -X
//           up to f32 rounding:
//           |S_ref + X| <= gamma X + rho (gamma = (D+2) u, rho = 6 D u M, M = max |value|; the head side's
//           extra rounding of e + r and of t - r is inside rho).  Against the EXACT true-entity key
//           s_true = -d_t this gives two integer thresholds per query:
//              SAD + E_c < T_lo  -> certainly ranked above (count it)
//              SAD - E_c > T_hi  -> certainly below        (ignore it)
//              otherwise   -> undecided: listed as a (query, row) pair
//           Undecided pairs are ~0.2 % for FB15k-237-like data.  A workgroup lists up to kSQuota pairs;
//           beyond that the (query, 64-candidate tile) segment contributes nothing and its bit is set in
//           a flag bitmap.
//   pass 2  listed pairs and flagged segments are re-scored with the order-exact Scorer<TRANSE> (the
//           code that produced s_true) and counted exactly.
// Non-finite values anywhere, or a degenerate range, make every kernel of the pre-pass a no-op and the
// segment refinement sweep all tiles: the result is exact in every case, just slower.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "exact_coop.h"
#include "knobs.h"
#include "launch.h"
#include "rank_common.h"
#include "sad_common.h"
#include "score_core.h"

#pragma clang fp contract(off)

namespace blp {


template <int D>
__global__ __launch_bounds__(256) void sad_range_kernel(const float* __restrict__ table, int64_t N, int64_t ld,
                                                        const QRows q_fixed,
                                                        const QRows q_rel, int64_t q_head, int64_t Q,
                                                        SadParams* __restrict__ partial) {
    SadRange range;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x, t0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (int64_t i = t0; i < N * (D / 4); i += stride) {
        const float4 v = *reinterpret_cast<const float4*>(table + (i / (D / 4)) * ld + (i % (D / 4)) * 4);
        range.see(v.x); range.see(v.y); range.see(v.z); range.see(v.w);
    }
    for (int64_t u = t0; u < Q * (D / 4); u += stride) {  // four coefficients per step: one row lookup each side
        const bool head = u / (D / 4) < q_head;
        const float4 f = *reinterpret_cast<const float4*>(q_fixed.flat(4 * u, D));
        const float4 r = *reinterpret_cast<const float4*>(q_rel.flat(4 * u, D));
        range.see(sad_coef(f.x, r.x, head)); range.see(sad_coef(f.y, r.y, head));
        range.see(sad_coef(f.z, r.z, head)); range.see(sad_coef(f.w, r.w, head));
    }
    sad_range_block_store(range, partial);
}

// Candidate tile image: 64 rows per tile, uint4 index ((tile * D/8 + j4) * 64 + lane) holds dwords
// 4 j4 .. 4 j4 + 3 of row (tile * 64 + lane); dword j packs elements 2j (low half) and 2j + 1.
// resid[row] = the row's rounding residual E_c, rounded up to an integer.
constexpr int64_t kSFoldFinishBlocks = 1024;  // quantising workgroups up to which they finish the range themselves ...
constexpr int64_t kSFoldRangeBlocks = 256;    // ... from at most this many partial results

template <int D>
__device__ __forceinline__ void sad_quantize_table_tile(int64_t tile, const float* __restrict__ table, int64_t N,
                                                        int64_t ld, const SadParams* __restrict__ p,
                                                        uint4* __restrict__ cimg, unsigned* __restrict__ resid) {
    const SadScale sc = sad_scale(p);
    if (!sc.ok) return;
    const int row_in = threadIdx.x >> 2, part = threadIdx.x & 3;
    const int64_t row = tile * 64 + row_in;
    constexpr int F = D / 4;  // floats per thread
    unsigned w[F / 2];
    float res = 0.f;
    bool outside = false;  // a value the map does not cover: the row is left to the exact path
    if (row < N) {
        const float* src = table + row * ld + part * F;
#pragma unroll
        for (int i = 0; i < F / 4; ++i) {
            const float4 v = *reinterpret_cast<const float4*>(src + 4 * i);
            w[2 * i] = sad_quant(v.x, sc, res, outside);
            w[2 * i] |= sad_quant(v.y, sc, res, outside) << 16;
            w[2 * i + 1] = sad_quant(v.z, sc, res, outside);
            w[2 * i + 1] |= sad_quant(v.w, sc, res, outside) << 16;
        }
    } else {
#pragma unroll
        for (int i = 0; i < F / 2; ++i) w[i] = 0;
    }
    res += __shfl_xor(res, 1);
    res += __shfl_xor(res, 2);
    outside |= (bool)__shfl_xor((int)outside, 1);
    outside |= (bool)__shfl_xor((int)outside, 2);
    if (part == 0)  // the image is padded to whole tiles
        resid[row] = outside ? kSRowExact : (unsigned)(res * 1.0001f + kSResidSlack * D) + 1u;
    uint4* dst = cimg + (tile * (D / 8) + part * (F / 8)) * 64 + row_in;
#pragma unroll
    for (int i = 0; i < F / 8; ++i) dst[i * 64] = make_uint4(w[4 * i], w[4 * i + 1], w[4 * i + 2], w[4 * i + 3]);
}

// Query image (D/2 dwords per query, same packing) and the two signed thresholds {T_lo, T_hi}.
// D/2 consecutive threads per query (32, 64 or 128: whole waves or half-waves).
// (workgroup vblock of vgrid)
template <int D>
__device__ __forceinline__ void sad_quantize_queries_block(int64_t vblock, int64_t vgrid, const QRows& q_fixed,
                                                           const QRows& q_rel, int64_t q_head,
                                                           int64_t Q, const float* __restrict__ key_true,
                                                           const SadParams* __restrict__ p,
                                                           unsigned* __restrict__ qimg,
                                                           int2* __restrict__ thr, unsigned* __restrict__ flags,
                                                           int64_t n_flag_words) {
    const int64_t i = vblock * 256 + threadIdx.x;
    for (int64_t j = i; j < n_flag_words; j += vgrid * 256) flags[j] = 0;  // first slab's bitmap
    const SadScale sc = sad_scale(p);
    if (!sc.ok) return;
    __shared__ float part_res[4], part_max[4];
    __shared__ int part_out[4];
    const bool live = i < Q * (D / 2);
    const int64_t q = live ? i / (D / 2) : Q - 1;
    const bool head = q < q_head;
    float res = 0.f, qmax = 0.f;  // qmax: the query's largest |fixed|, |rel| (its share of the rounding term rho)
    bool outside = false;         // a coefficient the map does not cover: the query is left to the exact path
    if (live) {
        const float2 f = *reinterpret_cast<const float2*>(q_fixed.flat(2 * i, D));
        const float2 r = *reinterpret_cast<const float2*>(q_rel.flat(2 * i, D));
        unsigned w = sad_quant(sad_coef(f.x, r.x, head), sc, res, outside);
        w |= sad_quant(sad_coef(f.y, r.y, head), sc, res, outside) << 16;
        qimg[i] = w;
        qmax = fmaxf(fmaxf(fabsf(f.x), fabsf(f.y)), fmaxf(fabsf(r.x), fabsf(r.y)));  // (NaN: outside is set anyway)
    }
#pragma unroll
    for (int off = 1; off < (D / 2 < 64 ? D / 2 : 64); off <<= 1) {
        res += __shfl_xor(res, off);
        qmax = fmaxf(qmax, __shfl_xor(qmax, off));
        outside |= (bool)__shfl_xor((int)outside, off);
    }
    if constexpr (D / 2 > 64) {  // D = 256: two waves per query
        const int w = threadIdx.x >> 6;
        if ((threadIdx.x & 63) == 0) { part_res[w] = res; part_max[w] = qmax; part_out[w] = outside; }
        __syncthreads();
        res = part_res[w & ~1] + part_res[w | 1];
        qmax = fmaxf(part_max[w & ~1], part_max[w | 1]);
        outside = part_out[w & ~1] || part_out[w | 1];
    }
    if (live && i % (D / 2) == 0) {
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

// Both images in one launch: workgroups [0, table_blocks) a candidate tile each, the rest the queries.  n_partial > 0
// (small calls: a few hundred workgroups): every workgroup first reduces the range pass's partial results itself -- in
// the fixed order of sad_range_finish_wave, so all of them quantise with the same map -- instead of waiting for a launch of
// its own to do it once; workgroup 0 leaves the result in *p for the kernels that follow.
template <int D>
__global__ __launch_bounds__(256) void sad_quantize_kernel(const float* __restrict__ table, int64_t N, int64_t ld,
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
        sad_quantize_table_tile<D>(blockIdx.x, table, N, ld, pp, cimg, resid);
    else
        sad_quantize_queries_block<D>(blockIdx.x - table_blocks, gridDim.x - table_blocks, q_fixed, q_rel, q_head, Q, key_true, pp,
                                      qimg, thr, flags, n_flag_words);
}

// ------------------------------------------------------------------------------------------------
template <int D, int TPW>
struct SadTiles {
    unsigned v[TPW][D / 2];
    unsigned bias[TPW];  // accumulator start: the row's rounding residual E_c, or kSInvalid for padding rows
    unsigned long long exact_rows[TPW];  // lanes whose row is marked exact-only (kSRowExact): undecided for every query
};

#ifndef BLP_SAD_CHUNK_BLOCK
#define BLP_SAD_CHUNK_BLOCK 4  // FB15k-237 block, ranking pass ms | fabric MB per launch: 2: 3.07 | -, 4: 3.06 | 319, 8: 3.08 | 261, 16: 3.09 | 170, 32: 3.09 | 136 (was 3.24 | 1390)
#endif
constexpr unsigned kSadChunkBlock = BLP_SAD_CHUNK_BLOCK;  // query chunks whose workgroups run together on an XCD

// -DBLP_TIMING: per wave of the kernel below, shader-clock ticks and ticks of the constant 100 MHz counter, summed: their
// ratio is the clock the kernel ran at (tools/gemm_ab.py --sad)
#ifdef BLP_TIMING
__device__ unsigned long long g_sad_timing[4];
extern "C" int blp_debug_read_sad_timing(unsigned long long* out) {
    hipError_t err = hipMemcpyFromSymbol(out, HIP_SYMBOL(g_sad_timing), sizeof(g_sad_timing));
    unsigned long long zero[4] = {0, 0, 0, 0};
    if (err == hipSuccess) err = hipMemcpyToSymbol(HIP_SYMBOL(g_sad_timing), zero, sizeof(zero));
    return (int)err;
}
#endif

template <int D, int TPW>
__global__ __launch_bounds__(kSW * 64, (TPW * D / 2 <= 64 ? 5 : TPW * D / 2 <= 128 ? 3 : 2)) void rank_sad_kernel(
    const uint4* __restrict__ cimg, const unsigned* __restrict__ resid, int64_t n_rows, int n_groups, int q_per_group,
    const unsigned* __restrict__ qimg, const int2* __restrict__ thr, int64_t Q, int words_per_query,
    unsigned long long* __restrict__ acc,
    unsigned* __restrict__ flags, uint2* __restrict__ pairs, SadParams* __restrict__ params) {
    if (!sad_scale(params).ok) return;
#ifdef BLP_TIMING
    const unsigned long long clk0 = __builtin_readcyclecounter(), wall0 = __builtin_amdgcn_s_memrealtime();
#endif
    __shared__ int2 thr_s[kSChunk];
    __shared__ unsigned cnt[kSChunk];
    __shared__ uint2 pair_s[kSQuota];
    __shared__ unsigned wave_used[kSW], wave_base[kSW], ec_max_s;  // per-wave slices of pair_s: fill counts, global offsets
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    // Workgroup order.  Blocks are dispatched to the 8 XCDs round-robin (block b -> XCD b % 8), each with its own 4 MB L2.
    // With b = chunk * n_groups + group every XCD saw every query chunk and, with ~160 workgroups resident, ~22 chunks'
    // images at a time next to the whole candidate image: 1.39 GB of fabric reads per launch for 118 MB of data.  Each
    // XCD now walks a contiguous range of a logical index laid out as (block of kSadChunkBlock query chunks) x (candidate
    // group) x (chunk of the block): a query chunk's image is read by one XCD only, the candidate image once per block
    // of chunks.
    const unsigned n_blocks = gridDim.x, xcd = blockIdx.x & 7u, per_xcd = n_blocks >> 3, rem_b = n_blocks & 7u;
    const unsigned logical = (xcd < rem_b ? xcd * (per_xcd + 1) : rem_b * (per_xcd + 1) + (xcd - rem_b) * per_xcd) + (blockIdx.x >> 3);
    const unsigned n_chunks_all = n_blocks / (unsigned)n_groups;  // the grid is n_groups x chunks
    const unsigned per_cb = (unsigned)n_groups * kSadChunkBlock, cb = logical / per_cb, in_cb = logical % per_cb;
    const unsigned chunks_here = n_chunks_all - cb * kSadChunkBlock < kSadChunkBlock ? n_chunks_all - cb * kSadChunkBlock : kSadChunkBlock;
    const int group = (int)(in_cb / chunks_here);
    const int64_t q0 = (int64_t)(cb * kSadChunkBlock + in_cb % chunks_here) * q_per_group;  // q_per_group <= kSChunk
    const int nq = (int)(Q - q0 < q_per_group ? Q - q0 : q_per_group);

    for (int i = tid; i < kSChunk; i += kSW * 64) cnt[i] = 0;
    if (tid == 0) ec_max_s = 0;
    __syncthreads();

    // this wave's TPW candidate tiles: one row per lane, D/2 packed dwords each.  The row's residual E_c is
    // folded into the accumulator's start value (SAD + E_c is what the lower threshold is compared with);
    // the upper threshold takes 2 x the workgroup's largest E_c instead of the row's own (a slightly wider
    // band for the other rows, two VALU instructions fewer per (query, tile)).
    SadTiles<D, TPW> c;
    const int64_t n_tiles = (n_rows + 63) / 64;
    const int64_t tile0 = ((int64_t)group * kSW + wave) * TPW;
    unsigned ec_max = 0;
    static_for<TPW>([&](auto tt) {
        constexpr int t = decltype(tt)::value;
        int64_t tile = tile0 + t;
        const bool exists = tile * 64 + lane < n_rows;
        tile = tile < n_tiles ? tile : n_tiles - 1;
        const unsigned ec = resid[tile * 64 + lane];
        const bool marked = exists && ec == kSRowExact;
        c.exact_rows[t] = __ballot(marked);
        c.bias[t] = exists ? (marked ? 0u : ec) : kSInvalid;
        ec_max = exists && !marked && ec > ec_max ? ec : ec_max;
        const uint4* src = cimg + tile * (D / 8) * 64 + lane;
        static_for<D / 8>([&](auto jj) {
            constexpr int j = decltype(jj)::value;
            const uint4 x = src[j * 64];
            c.v[t][4 * j] = x.x; c.v[t][4 * j + 1] = x.y; c.v[t][4 * j + 2] = x.z; c.v[t][4 * j + 3] = x.w;
        });
    });
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned o = __shfl_xor(ec_max, off);
        ec_max = o > ec_max ? o : ec_max;
    }
    if (lane == 0) atomicMax(&ec_max_s, ec_max);
    __syncthreads();
    for (int i = tid; i < kSChunk; i += kSW * 64) {
        int2 t2 = i < nq ? thr[q0 + i] : make_int2(0, 0);
        const long long hi2 = (long long)t2.y + 2ll * ec_max_s;       // SAD + E_c above this: certainly below
        t2.y = hi2 < (long long)kSThrMax ? (int)hi2 : (int)kSThrMax;
        thr_s[i] = t2;
    }
    __syncthreads();

    // The query loop.  A CU has ONE scalar ALU for its four SIMDs: with five waves per SIMD walking 64 v_sad_u16 per
    // query, ~70 scalar instructions per query (64-bit address arithmetic for three row pointers with end-of-block
    // clamps, the atomic optimiser's lane loop around the LDS count) kept the scalar pipe as busy as the SAD pipe and
    // the kernel at 78 % of the SAD rate.  So: pointers advance by a constant (the image is padded by four rows, no
    // clamps), and a query's count goes to lane j % 64 of a register with one masked add; the register is flushed to
    // the LDS counters once per 64 queries by all lanes.
    unsigned my_pairs = 0;  // entries in this wave's slice of pair_s (wave-uniform)
    const float* row = reinterpret_cast<const float*>(qimg + q0 * (D / 2));
    sf16 cur = sload16<0>(row);
    sdrain(cur);
    for (int jb = 0; jb < nq; jb += 64) {  // 64 queries per flush of the count register
    const int jn = nq - jb < 64 ? nq - jb : 64;
    unsigned cnt_reg = 0;  // lane l: certainly-above count of query jb + l
    unsigned long long bit = 1;
    for (int j = jb; j < jb + jn; ++j, row += D / 2, bit <<= 1) {
        const float* next_row = row + D / 2;  // rows past the block's last query: the next block's, or the padding
        const int2 th = thr_s[j];
        unsigned sad[TPW];
        static_for<TPW>([&](auto tt) { sad[decltype(tt)::value] = c.bias[decltype(tt)::value]; });
        static_for<D / 32>([&](auto kk) {
            constexpr int k = decltype(kk)::value;
            sf16 nxt;
            if constexpr (k + 1 < D / 32) nxt = sload16<(k + 1) * 64>(row); else nxt = sload16<0>(next_row);
            // the four waves walk the same rows: they take turns touching the lines of the query three ahead
            if constexpr (k == 0) { if ((j & 3) == wave) stouch<D * 2>(row + 3 * (D / 2)); }
            static_for<16>([&](auto ii) {
                constexpr int i = decltype(ii)::value;
                const float qf = cur[i];  // (bit_cast straight from the vector element reads element 0)
                const unsigned qv = __float_as_uint(qf);
                static_for<TPW>([&](auto tt) {
                    constexpr int t = decltype(tt)::value;
                    sad[t] = __builtin_amdgcn_sad_u16(qv, c.v[t][16 * k + i], sad[t]);
                });
            });
            sdrain(nxt);
            cur = nxt;
        });
        unsigned n_above = 0;
        static_for<TPW>([&](auto tt) {
            constexpr int t = decltype(tt)::value;
            const unsigned long long above = __ballot((int)sad[t] < th.x) & ~c.exact_rows[t];
            const unsigned long long und = (__ballot((int)sad[t] <= th.y) | c.exact_rows[t]) & ~above;
            unsigned n_t = __popcll(above);
            if (und) {  // wave-uniform.  The wave fills its own slice of the list: no LDS round trip for a slot
                const unsigned n = __popcll(und);
                if (my_pairs + n <= kSQuota / kSW) {
                    if ((und >> lane) & 1ull)
                        pair_s[wave * (kSQuota / kSW) + my_pairs + __popcll(und & ((1ull << lane) - 1ull))] =
                            make_uint2((unsigned)(q0 + j), (unsigned)((tile0 + t) * 64 + lane));
                    my_pairs += n;
                } else {
                    const int64_t tile = tile0 + t;
                    if (lane == 0) atomicOr(flags + (size_t)(q0 + j) * words_per_query + (tile >> 5), 1u << (tile & 31));
                    n_t = 0;
                }
            }
            n_above += n_t;
        });
        {   // cnt_reg[lane j - jb] += n_above: one VALU instruction under a one-lane exec mask
            unsigned long long saved;
            asm volatile("s_mov_b64 %1, exec\n\ts_mov_b64 exec, %3\n\tv_add_u32 %0, %0, %2\n\ts_mov_b64 exec, %1"
                         : "+v"(cnt_reg), "=&s"(saved) : "s"(n_above), "s"(bit));
        }
    }
    if (cnt_reg) atomicAdd(&cnt[jb + lane], cnt_reg);  // all lanes: distinct counters, plain ds_add_u32
    }
    __syncthreads();
    for (int i = tid; i < nq; i += kSW * 64) {
        const unsigned long long v = cnt[i];
        if (v) atomicAdd(acc + q0 + i, v | (v << 32));  // certainly above: gt and ge
    }
    if (lane == 0) wave_used[wave] = my_pairs;
    __syncthreads();
    if (tid == 0) {
        unsigned total = 0;
        for (int w = 0; w < kSW; ++w) { wave_base[w] = total; total += wave_used[w]; }
        const unsigned base = total ? atomicAdd(&params->n_pairs, total) : 0u;
        for (int w = 0; w < kSW; ++w) wave_base[w] += base;
    }
    __syncthreads();
    for (int w = 0; w < kSW; ++w)
        for (unsigned i = tid; i < wave_used[w]; i += kSW * 64) pairs[wave_base[w] + i] = pair_s[w * (kSQuota / kSW) + i];
#ifdef BLP_TIMING
    if (lane == 0) {
        atomicAdd(&g_sad_timing[0], __builtin_readcyclecounter() - clk0);
        atomicAdd(&g_sad_timing[1], __builtin_amdgcn_s_memrealtime() - wall0);
        atomicAdd(&g_sad_timing[2], 1ull);
    }
#endif
}

// Pass 2a: the listed pairs, 64 per wave, one lane per pair (the L1 sum of a pair is one sequential chain of D
// additions: it cannot be spread over lanes without changing the rounding).  What CAN be shared is the memory
// traffic: a lane reading its own candidate row and its own coefficient rows 16 bytes at a time makes every load
// instruction touch 64 different cache lines.  Instead the wave fetches the 64 rows the way the exact kernel
// fetches a tile (rank_all.hip: load_tile) -- 8 rows x 128 B per load instruction, whole lines -- 32 columns at a
// time, and transposes them through a wave-private LDS slab so that lane p ends up with row p; the rows are
// gathered through per-lane pointers instead of being consecutive.  Same arithmetic as Scorer<TRANSE, *, D>::score.
template <int D>
__global__ __launch_bounds__(64) void sad_refine_pairs_kernel(const float* __restrict__ table, int64_t ld,
                                                              const QRows q_fixed,
                                                              const QRows q_rel,
                                                              const float* __restrict__ key_true, int64_t q_head,
                                                              const uint2* __restrict__ pairs,
                                                              const SadParams* __restrict__ params,
                                                              unsigned long long* __restrict__ acc, const Gate gate) {
    if (gate_heavy(gate)) return;  // the lists ran full: the exact kernel re-ranks the block (rank_common.h: Gate)
    __shared__ __attribute__((aligned(16))) float slab[64 * kRefStride];
    const int lane = threadIdx.x;
    const int64_t n = params->n_pairs;
    for (int64_t base = (int64_t)blockIdx.x * 64; base < n; base += (int64_t)gridDim.x * 64) {  // wave-uniform
        const int64_t i = base + lane;
        const uint2 p = i < n ? pairs[i] : make_uint2(kSNoPair, 0u);
        const bool live = p.x != kSNoPair;
        const int64_t q = live ? p.x : 0;
        const bool head = q < q_head;
        const float key = transe_key_64<D>(table + (live ? (int64_t)p.y : 0) * ld, q_fixed.row(q), q_rel.row(q), head, slab, lane);
        const float kt = key_true[q];
        const unsigned long long gt = live && key > kt, ge_ = live && key >= kt;
        if (gt | ge_) atomicAdd(acc + q, gt | (ge_ << 32));
    }
}

// Pass 2b: one wave per query sweeps the query's flag words; every flagged 64-candidate tile is
// re-scored exactly.  When the pre-pass did not run (non-finite input, degenerate range) all tiles are.
template <int D>
__global__ __launch_bounds__(256) void sad_refine_tiles_kernel(const float* __restrict__ table, int64_t n_rows,
                                                               int64_t ld, const QRows q_fixed,
                                                               const QRows q_rel,
                                                               const float* __restrict__ key_true, int64_t q_head,
                                                               int64_t Q, int words_per_query,
                                                               const unsigned* __restrict__ flags,
                                                               const SadParams* __restrict__ params,
                                                               unsigned long long* __restrict__ acc, const Gate gate,
                                                               const FallbackPrep prep, int64_t q_tail) {
    if (gate_heavy(gate)) {  // the lists ran full: the exact kernel re-ranks the block (rank_common.h: Gate); this grid prepares it
        fallback_prep<TRANSE, D>(prep, q_fixed, q_rel, q_head, q_tail);
        return;
    }
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
    const float* rl = q_rel.row(q);
    unsigned gt = 0, ge = 0;
    for (int w0 = 0; w0 < words_per_query; w0 += 64) {
        const unsigned mine = w0 + lane < words_per_query ? (all ? 0xFFFFFFFFu : row[w0 + lane]) : 0u;
        unsigned long long nonzero = __ballot(mine != 0);
        while (nonzero) {  // wave-uniform: next non-empty flag word
            const int src = __builtin_ctzll(nonzero);
            nonzero &= nonzero - 1;
            unsigned bits = __shfl(mine, src);
            const int64_t tile_base = (int64_t)(w0 + src) * 32;
            while (bits) {
                const int b = __builtin_ctz(bits);
                bits &= bits - 1;
                const int64_t r = (tile_base + b) * 64 + lane;
                const bool ok = r < n_rows;
                if (__ballot(ok) == 0) continue;
                float e[D];
                load_row<D>(e, table + (ok ? r : 0) * ld);
                const float key = q < q_head ? Scorer<TRANSE, HEAD, D>::template score<false>(e, LazyCoef<TRANSE, HEAD, D>{f, rl})
                                             : Scorer<TRANSE, TAIL, D>::template score<false>(e, LazyCoef<TRANSE, TAIL, D>{f, rl});
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

// ------------------------------------------------------------------------------------------------
static constexpr int sad_tiles_per_wave(int D) { return D <= 64 ? 2 : 1; }

struct SadWorkspace {
    float* key_true;
    unsigned long long* acc;
    SadParams* params; SadParams* partial; int2* thr; unsigned* qimg; uint4* cimg; unsigned* resid; unsigned* flags;
    uint2* pairs;
    float* fallback_coef;  // coefficient rows of the exact re-ranking (rank_common.h: Gate), or nullptr: this block has no fallback
    int64_t pass_groups;  // candidate groups (kSW * TPW tiles) per pre-pass + refine pass
    size_t bytes;
};

// Candidate groups per pass: the flag bitmap (one bit per (query, tile)) and the pair list (kSQuota
// entries per workgroup, worst case) are each capped at ~256 MB; larger (Q x N) problems are
// processed in candidate slabs.
// Queries per workgroup, 16 .. kSChunk: long chunks amortise a workgroup's set-up (its candidate tile, thresholds,
// two barriers: about as much as 45 queries of work), short ones keep the (candidate group, query chunk) grid
// balanced over the resident workgroups.  Picks the length with the smallest estimated time
// rounds(grid / resident) x (set-up + queries x cycles per query).  Measured on the FB15k-237 table: 128
// queries (the reference's eval batch) 52 us at 128 per workgroup, 16 us at 16; 1 024 queries: 32; an 8-way
// shard of the test set (13 k queries): 128 (0.54 -> 0.52 ms); the whole test set: 256 (3.79 -> 3.75 ms).
static int sad_queries_per_group(int D, int64_t N, int64_t Q) {
    const int64_t tiles_per_group = kSW * sad_tiles_per_wave(D);
    const int64_t n_groups = ((N + 63) / 64 + tiles_per_group - 1) / tiles_per_group;
    const int64_t resident = D >= 256 ? 256 * 3 : 256 * 5;  // workgroups the chip holds (launch bounds of rank_sad_kernel)
    const int64_t query_cycles = (int64_t)(D / 2) * sad_tiles_per_wave(D) * 17 / 4, setup_cycles = 12000;
    if (const int forced = (int)knob(KNOB_SAD_QUERIES_PER_GROUP))  // test knob
        if (forced >= 16 && forced <= kSChunk && (forced & (forced - 1)) == 0) return forced;
    int best = kSChunk;
    int64_t best_cost = INT64_MAX;
    for (int per_group = kSChunk; per_group >= 16; per_group >>= 1) {
        const int64_t grid = n_groups * ((Q + per_group - 1) / per_group);
        const int64_t cost = ((grid + resident - 1) / resident) * (setup_cycles + per_group * query_cycles);
        if (cost < best_cost) { best_cost = cost; best = per_group; }
    }
    return best;
}

static int64_t sad_groups_per_pass(int D, int64_t N, int64_t Q) {
    const int64_t tiles_per_group = kSW * sad_tiles_per_wave(D);
    const int64_t n_groups = ((N + 63) / 64 + tiles_per_group - 1) / tiles_per_group;
    const int64_t per_group = sad_queries_per_group(D, N, Q);
    const int64_t n_chunks = (Q + per_group - 1) / per_group;
    const int64_t cap = (int64_t)256 << 20;
    int64_t by_pairs = cap / (kSQuota * 8) / (n_chunks > 0 ? n_chunks : 1);
    int64_t by_flags = cap / 4 / (Q > 0 ? Q : 1) * 32 / tiles_per_group;
    int64_t g = n_groups;
    if (g > by_pairs) g = by_pairs;
    if (g > by_flags) g = by_flags;
    if (const int64_t forced = knob(KNOB_SAD_PASS_GROUPS))  // test knob: force the multi-slab path
        if (forced > 0 && forced < g) g = forced;
    return g < 1 ? 1 : g;
}

static SadWorkspace carve_sad(void* base, int D, int64_t N, int64_t q_head, int64_t q_tail) {
    SadWorkspace w;
    const int64_t Q = q_head + q_tail;
    char* p = static_cast<char*>(base);
    size_t off = 0;
    w.key_true = reinterpret_cast<float*>(p + off);  off = align_up(off + (size_t)Q * 4, 256);
    w.acc = reinterpret_cast<unsigned long long*>(p + off);   off = align_up(off + (size_t)Q * 8, 256);
    w.params = reinterpret_cast<SadParams*>(p + off); off = align_up(off + sizeof(SadParams), 256);
    w.partial = reinterpret_cast<SadParams*>(p + off); off = align_up(off + sizeof(SadParams) * kSRangeBlocks, 256);
    w.thr = reinterpret_cast<int2*>(p + off);         off = align_up(off + (size_t)Q * 8, 256);
    // + 4 rows: the pre-pass reads ahead of its query (next row, the row three ahead) without clamping at the end
    w.qimg = reinterpret_cast<unsigned*>(p + off);    off = align_up(off + (size_t)(Q + 4) * (D / 2) * 4 + 64, 256);
    w.cimg = reinterpret_cast<uint4*>(p + off);       off = align_up(off + (size_t)((N + 63) / 64) * 64 * (D / 2) * 4, 256);
    w.resid = reinterpret_cast<unsigned*>(p + off);   off = align_up(off + (size_t)((N + 63) / 64) * 64 * 4, 256);
    w.pass_groups = sad_groups_per_pass(D, N, Q);
    const int64_t tiles_per_group = kSW * sad_tiles_per_wave(D);
    const int64_t words = (w.pass_groups * tiles_per_group + 31) / 32;
    w.flags = reinterpret_cast<unsigned*>(p + off); off = align_up(off + (size_t)Q * words * 4, 256);
    w.pairs = reinterpret_cast<uint2*>(p + off);
    const int64_t per_group = sad_queries_per_group(D, N, Q);
    off = align_up(off + (size_t)w.pass_groups * ((Q + per_group - 1) / per_group) * kSQuota * 8, 256);
    // the exact fallback: blocks of one candidate slab and at least kFallbackMinPairs pairs
    w.fallback_coef = nullptr;
    const int64_t n_groups_all = ((N + 63) / 64 + tiles_per_group - 1) / tiles_per_group;
    if (w.pass_groups >= n_groups_all && Q * N >= kFallbackMinPairs && Q <= 0x7fffffff / 2) {
        w.fallback_coef = reinterpret_cast<float*>(p + off);
        off = align_up(off + exact_fallback_coef_floats(D, q_head, q_tail) * 4, 256);
    }
    w.bytes = off;
    return w;
}

// Worth it once its longer launch chain (range, quantisation, pre-pass, two refinement kernels: 8 launches against
// the exact path's 4) is paid for by the 2.3x cheaper pair-element: measured on the FB15k-237 table (14 541 rows) the
// exact f32 kernels win up to ~256 queries (128 queries: 42 vs 59 us), on the 4.6 M-row table the pre-pass wins from 64
// queries on -- i.e. from about 4 million (query, candidate) pairs (tools/bench_small_blocks.py).
bool rank_sad_applicable(int model, int D, int64_t N, int64_t q_head, int64_t q_tail) {
    if (knob(KNOB_RANK_KERNEL) == 1) return false;  // test knob: the exact f32 kernels
    const int64_t Q = q_head + q_tail;
    int64_t min_queries = kSadMinQueries;
    bool forced = false;
    if (const int64_t v = knob(KNOB_SAD_MIN_QUERIES)) { min_queries = v > 0 ? v : 1; forced = true; }  // A/B knob
    if (model != TRANSE || !(D == 64 || D == 128 || D == 256) || Q < min_queries) return false;
    return forced || Q * N >= kSadMinPairs;
}

hipError_t sad_prepass_stats(int D, int64_t N, int64_t q_head, int64_t q_tail, const void* workspace, PrepassStats* out,
                             hipStream_t stream) {
    const SadWorkspace w = carve_sad(const_cast<void*>(workspace), D, N, q_head, q_tail);
    out->path = 1;
    const int64_t tiles_per_group = kSW * sad_tiles_per_wave(D);
    const int64_t words = (w.pass_groups * tiles_per_group + 31) / 32;
    unsigned long long host[3] = {0, 0, 0};
    const hipError_t err = launch_count_bits(w.flags, (q_head + q_tail) * words, w.pairs, &w.params->n_pairs, false, host, stream);
    out->flagged_rows = (long long)host[0] * 64;  // a flag = one (query, 64-candidate tile)
    out->listed = (long long)host[2];             // one entry per undecided pair
    return err;
}

size_t rank_sad_workspace_bytes(int model, int D, int64_t N, int64_t q_head, int64_t q_tail) {
    if (!rank_sad_applicable(model, D, N, q_head, q_tail)) return 0;
    return carve_sad(nullptr, D, N, q_head, q_tail).bytes;
}

template <int D>
static hipError_t rank_sad_impl(const float* table, int64_t N, int64_t ld, const QRows q_fixed, const QRows q_rel,
                                const QRows q_true, int64_t q_head, int64_t q_tail,
                                const FilterSpec& filter, int32_t* counts, void* workspace,
                                int n_cu, hipStream_t stream, hipEvent_t ev_start, hipEvent_t ev_stop) {
    constexpr int TPW = sad_tiles_per_wave(D);
    const int64_t Q = q_head + q_tail;
    SadWorkspace w = carve_sad(workspace, D, N, q_head, q_tail);
    // (the true keys of a small call folded into the range launch, one lane per query: tried -- 2 us slower from 256
    //  queries on, where this path starts)
    hipError_t err = launch_true_keys(TRANSE, D, q_fixed, q_rel, q_head, q_tail, q_true, w.key_true, w.acc,
                                      stream);
    if (err != hipSuccess) return err;
    if (ev_start) (void)hipEventRecord(ev_start, stream);
    const int64_t n_tiles = (N + 63) / 64;
    const int64_t query_blocks = (Q * (D / 2) + 255) / 256;
    if (n_tiles + query_blocks > 0x7fffffff) return hipErrorInvalidValue;
    // small calls: no launch for the range's last step -- the quantising workgroups do it themselves (from fewer partials)
    const bool fold_finish = n_tiles + query_blocks <= kSFoldFinishBlocks;
    int64_t range_blocks;
    {
        const int64_t items = N * (D / 4) > Q * D ? N * (D / 4) : Q * D, cap = fold_finish ? kSFoldRangeBlocks : kSRangeBlocks;
        range_blocks = (items + 255) / 256;
        range_blocks = range_blocks < cap ? (range_blocks > 0 ? range_blocks : 1) : cap;
        sad_range_kernel<D><<<dim3((unsigned)range_blocks), 256, 0, stream>>>(table, N, ld, q_fixed, q_rel, q_head, Q, w.partial);
        if (!fold_finish) sad_range_finish_kernel<<<1, 64, 0, stream>>>(w.partial, (int)range_blocks, w.params);
    }
    // the first slab's flag bitmap and pair counter are zeroed by the prep kernels (later slabs: memsets)
    const int64_t first_rows = N < w.pass_groups * kSW * TPW * 64 ? N : w.pass_groups * kSW * TPW * 64;
    const int64_t first_groups = ((first_rows + 63) / 64 + kSW * TPW - 1) / (kSW * TPW);
    const int64_t first_words = (first_groups * kSW * TPW + 31) / 32;
    sad_quantize_kernel<D><<<dim3((unsigned)(n_tiles + query_blocks)), 256, 0, stream>>>(
        table, N, ld, w.params, w.partial, fold_finish ? (int)range_blocks : 0, w.cimg, w.resid, (unsigned)n_tiles, q_fixed, q_rel,
        q_head, Q, w.key_true, w.qimg, w.thr, w.flags, first_words * Q);

    const int64_t tiles_per_group = kSW * TPW;
    const int per_group = sad_queries_per_group(D, N, Q);
    const int64_t n_chunks = (Q + per_group - 1) / per_group;
    const int64_t pass_rows = w.pass_groups * tiles_per_group * 64;
    for (int64_t slab0 = 0; slab0 < N; slab0 += pass_rows) {  // one iteration unless the caps bind
        const int64_t n_rows = N - slab0 < pass_rows ? N - slab0 : pass_rows;
        const int64_t slab_tiles = (n_rows + 63) / 64;
        const int64_t n_groups = (slab_tiles + tiles_per_group - 1) / tiles_per_group;
        const int words = (int)((n_groups * tiles_per_group + 31) / 32);
        const int64_t n_blocks = n_groups * n_chunks;
        if (slab0 > 0) {
            err = hipMemsetAsync(w.flags, 0, (size_t)Q * words * 4, stream);
            if (err != hipSuccess) return err;
            err = hipMemsetAsync(&w.params->n_pairs, 0, 4, stream);
            if (err != hipSuccess) return err;
        }
        const float* slab = table + slab0 * ld;
        rank_sad_kernel<D, TPW><<<dim3((unsigned)n_blocks), kSW * 64, 0, stream>>>(
            w.cimg + (slab0 / 64) * (D / 8) * 64, w.resid + slab0, n_rows, (int)n_groups, per_group, w.qimg, w.thr, Q, words, w.acc,
            w.flags, w.pairs, w.params);
        const int64_t pair_blocks = (n_blocks * kSQuota + 63) / 64;  // 64 pairs per single-wave workgroup and iteration
        // heavy <=> the workgroups' lists are >= 90 % full (capacity: kSQuota entries each): exact ties on whole percents of the
        // table -- the flagged tiles would then cost 64 exact scores per flag (132 ms for the FB15k-237 block at 5 % ties) where
        // the exact kernel re-ranks everything in 8
        const Gate gate{w.fallback_coef ? &w.params->n_pairs : nullptr, (unsigned)((n_blocks * kSQuota / 10 * 9) < 0xffffffffll ? n_blocks * kSQuota / 10 * 9 : 0xffffffffll)};
        sad_refine_pairs_kernel<D><<<dim3((unsigned)(pair_blocks < (int64_t)n_cu * 40 ? pair_blocks : (int64_t)n_cu * 40)), 64, 0, stream>>>(
            slab, ld, q_fixed, q_rel, w.key_true, q_head, w.pairs, w.params, w.acc, gate);
        sad_refine_tiles_kernel<D><<<dim3((unsigned)((Q + kSweepQueries - 1) / kSweepQueries)), 256, 0, stream>>>(
            slab, n_rows, ld, q_fixed, q_rel, w.key_true, q_head, Q, words, w.flags, w.params, w.acc, gate,
            FallbackPrep{w.fallback_coef, w.acc}, q_tail);
        if (gate.counter) {
            err = launch_exact_fallback(TRANSE, D, table, N, ld, q_fixed, q_rel, q_head, q_tail, w.fallback_coef, w.key_true, w.acc, gate,
                                        n_cu, stream);
            if (err != hipSuccess) return err;
        }
    }
    if (ev_stop) (void)hipEventRecord(ev_stop, stream);
    err = launch_filter_finalize(TRANSE, D, table, N, ld, q_fixed, q_rel, w.key_true, q_head, q_tail, filter,
                                 w.acc, counts, stream);
    return err != hipSuccess ? err : hipGetLastError();
}

hipError_t launch_rank_all_sad(int D, const float* table, int64_t N, int64_t ld, const QRows q_fixed,
                               const QRows q_rel, const QRows q_true, int64_t q_head,
                               int64_t q_tail, const FilterSpec& filter, int32_t* counts,
                               void* workspace, int n_cu, hipStream_t stream, hipEvent_t ev_start,
                               hipEvent_t ev_stop) {
#define BLP_SAD_CASE(DD)                                                                                          \
    if (D == DD)                                                                                                  \
        return rank_sad_impl<DD>(table, N, ld, q_fixed, q_rel, q_true, q_head, q_tail, filter,          \
                                 counts, workspace, n_cu, stream, ev_start, ev_stop);
    BLP_SAD_CASE(64) BLP_SAD_CASE(128) BLP_SAD_CASE(256)
#undef BLP_SAD_CASE
    return hipErrorInvalidValue;
}

}  // namespace blp
