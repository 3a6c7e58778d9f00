// rank_gemm.hip -- all-entities ranking for the bilinear models (DistMult / ComplEx / SimplE) as a dense
// MFMA GEMM with a fused rank-count epilogue, made bit-exact by a rigorous error band and an exact
// refinement pass.  ("K2".)
//
// Scoring every entity against a query block IS a GEMM for these models: score(q, c) = <W_q, e_c> with
// a query-side vector W_q (DistMult: h*r; ComplEx: the complex product of r with h, or with conj t;
// SimplE: the two half-products).  But the reference's result is not that dot product rounded once:
// it rounds every elementwise product and sums in torch's vectorised order, and the ranks must be
// bit-identical.  So:
//   pass 1  S~ = W E^T on the matrix cores and per pair a three-way decision against the EXACT
//           true-entity score s_true:
//              S~ > s_true + eps  -> certainly ranked above   (count it)
//              S~ < s_true - eps  -> certainly below          (ignore it)
//              otherwise          -> undecided
//           eps(q, c) = C u ||B_q|| ||e_c|| bounds |S~ - S_ref| for ANY evaluation order: both the GEMM and
//           the reference are within a small multiple of u sum_k B_q[k] |e_c[k]| of the real-number score,
//           where B_q[k] >= |W_q[k]| is the sum of the absolute values of the products W_q[k] is made of
//           (Cauchy-Schwarz turns the sum into the two norms); u = 2^-24.  Two kernels:
//             rank_gemm_bf16_kernel (default)  every f32 operand split into bf16 hi + lo, three
//                                   v_mfma_f32_32x32x16_bf16 products per K-step, C = 1150 (see its header)
//             rank_gemm_kernel (BLP_GEMM_KERNEL=f32)  v_mfma_f32_32x32x2_f32, an exact f32 fma chain, 64 per
//                                   32 x 32 tile at K = 128, C = 320 = 2 (n + 2) + slack for n <= 128 terms
//           Undecided pairs (the true entity itself plus near-ties, a few per query on FB15k-237) go to a
//           per-workgroup pair list (a fixed region per workgroup: no global atomics, nothing to
//           overflow).  When a workgroup's quota is used up, the lane's whole (query, 16-candidate)
//           half-segment contributes nothing and its bit is set in a flag bitmap.
//   pass 2  a) every listed pair and b) every flagged half-segment is re-scored with the order-exact
//           Scorer<> routine (the same code that produced s_true) and counted exactly.  If everything
//           is flagged the result is still exact, just slower.
// NaN / Inf anywhere makes a pair undecided (both comparisons are false), so it takes the exact path.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <atomic>
#include <mutex>

#include "dot_band.h"
#include "exact_coop.h"
#include "knobs.h"
#include "launch.h"
#include "rank_common.h"
#include "score_core.h"

#pragma clang fp contract(off)

namespace blp {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kGW = 4;              // waves per workgroup (one 32-candidate tile each)
constexpr int kGQT = 32;            // queries per tile
constexpr int kGCT = 32;            // candidates per tile
constexpr int kGTilesPerChunk = 8;  // query tiles per workgroup
constexpr int kGSlab = 36;          // dwords, transpose slab row stride
constexpr int kPairQuota = 32;      // undecided (query, candidate) pairs a workgroup can list
constexpr unsigned kNoPair = 0xFFFFFFFFu;

// candidate row (within its 32-row tile) held by accumulator register r of a lane in half `half`
__device__ __forceinline__ int acc_row(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

// LDS floats shared by the two query-tile buffers and (before any staging) the transpose slabs.
__host__ __device__ constexpr int gemm_buf_floats(int D) {
    return 2 * (D / 8) * 64 * 4 > kGW * kGCT * kGSlab ? 2 * (D / 8) * 64 * 4 : kGW * kGCT * kGSlab;
}

// B-operand image of one 32-query tile: float4 index (g * 64 + l), l = 64-lane id, holds
// W_{q=l&31}[8g + (l>>5) + 2i], i = 0..3  ->  one linear ds_read_b128 per 4 MFMAs.
// One workgroup per query tile: 8 threads per query build W (and the band factor C u ||B_q||, rounded
// up) from coalesced reads, the image goes out through LDS so that the stores are linear too.
template <int MODEL, int D>
__global__ __launch_bounds__(256) void prep_gemm_kernel(const QRows q_fixed,
                                                        const QRows q_rel, int64_t q_head,
                                                        int64_t q_tail, float4* __restrict__ img_head,
                                                        float4* __restrict__ img_tail, float* __restrict__ eps_q) {
    constexpr int F4 = (D / 8) * 64;  // float4 per tile
    __shared__ float w_s[kGQT][D + 1];
    const int64_t th = (q_head + kGQT - 1) / kGQT;
    const bool head = blockIdx.x < th;
    const int64_t tile = head ? blockIdx.x : blockIdx.x - th;
    const int64_t n_side = head ? q_head : q_tail;
    const int ql = threadIdx.x >> 3, sub = threadIdx.x & 7;
    const int64_t q_local = tile * kGQT + ql;
    const bool q_ok = q_local < n_side;
    const int64_t q = (head ? 0 : q_head) + (q_ok ? q_local : n_side - 1);
    const float* f = q_fixed.row(q);
    const float* r = q_rel.row(q);
    float bsq = 0.f;
    for (int i = 0; i < D / 32; ++i) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int k = 32 * i + 4 * sub + c;
            float w, b;
            if (head) gemm_operand<MODEL, HEAD>(f, r, k, D, w, b);
            else gemm_operand<MODEL, TAIL>(f, r, k, D, w, b);
            w_s[ql][k] = q_ok ? w : 0.f;
            bsq += b * b;
        }
    }
    bsq += __shfl_xor(bsq, 1);
    bsq += __shfl_xor(bsq, 2);
    bsq += __shfl_xor(bsq, 4);
    if (sub == 0 && q_ok) eps_q[q] = kBandC * 5.9604645e-8f * sqrtf(bsq) * 1.0001f;
    __syncthreads();
    float4* out = (head ? img_head : img_tail) + tile * F4;
    for (int idx = threadIdx.x; idx < F4; idx += 256) {
        const int g = idx >> 6, l = idx & 63;
        const float* w = &w_s[l & 31][8 * g + (l >> 5)];
        out[idx] = make_float4(w[0], w[2], w[4], w[6]);
    }
}

// 32 candidate rows -> A operand a[s] (lanes 0-31: e[2s], lanes 32-63: e[2s+1]) + ||e|| of row (lane & 31).
template <int D>
__device__ __forceinline__ float load_a_gemm(float (&a)[D / 2], const float* __restrict__ table, int64_t N, int64_t ld,
                                             int64_t row0, float* slab, int lane) {
    float e[D];
    const int sub_row = lane >> 3, sub_col = (lane & 7) * 4;
    static_for<4>([&](auto ii) {
        constexpr int i = decltype(ii)::value;
        int64_t row = row0 + 8 * i + sub_row;
        row = row < N ? row : N - 1;
        const float* src = table + row * ld + sub_col;
        static_for<D / 32>([&](auto ss) {
            constexpr int s = decltype(ss)::value;
            const float4 v = *reinterpret_cast<const float4*>(src + s * 32);
            e[32 * s + 4 * i] = v.x; e[32 * s + 4 * i + 1] = v.y; e[32 * s + 4 * i + 2] = v.z; e[32 * s + 4 * i + 3] = v.w;
        });
    });
    float* wr = slab + sub_row * kGSlab + sub_col;
    const float* rd = slab + (lane & 31) * kGSlab;
    static_for<D / 32>([&](auto ss) {
        constexpr int s = decltype(ss)::value;
        if (s > 0) wave_lds_sync();
        static_for<4>([&](auto ii) {
            constexpr int i = decltype(ii)::value;
            *reinterpret_cast<float4*>(wr + 8 * i * kGSlab) =
                make_float4(e[32 * s + 4 * i], e[32 * s + 4 * i + 1], e[32 * s + 4 * i + 2], e[32 * s + 4 * i + 3]);
        });
        wave_lds_sync();
        static_for<8>([&](auto jj) {
            constexpr int j = decltype(jj)::value;
            const float4 w = *reinterpret_cast<const float4*>(rd + 4 * j);
            e[32 * s + 4 * j] = w.x; e[32 * s + 4 * j + 1] = w.y; e[32 * s + 4 * j + 2] = w.z; e[32 * s + 4 * j + 3] = w.w;
        });
    });
    float ss = 0.f;
    static_for<D>([&](auto d) { ss += e[d] * e[d]; });
    const bool hi = lane >= 32;
    static_for<D / 2>([&](auto s) { a[s] = hi ? e[2 * s + 1] : e[2 * s]; });
    return sqrtf(ss) * 1.0001f;
}

template <int BYTES, int WAVES = kGW>
__device__ __forceinline__ void stage_gemm_tile(const float4* __restrict__ src, float* dst, int wave, int lane) {
    constexpr int ROUNDS = (BYTES + WAVES * 1024 - 1) / (WAVES * 1024);
    static_for<ROUNDS>([&](auto kk) {
        constexpr int k = decltype(kk)::value;
        const int seg = (k * WAVES + wave) * 1024;
        if (seg + lane * 16 < BYTES)
            __builtin_amdgcn_global_load_lds((global_cptr)(reinterpret_cast<const char*>(src) + seg + lane * 16),
                                             (lds_ptr)(reinterpret_cast<char*>(dst) + seg), 16, 0, 0);
    });
}

// The same staging with the LDS-DMA issued through inline asm.  After a global_load_lds the compiler puts
// `s_waitcnt vmcnt(0)` in front of the wave's NEXT LDS access of any kind (it cannot tell the DMA's target from the
// buffer being read): issued at the top of a pipeline stage, the copy of tile t + 2 was waited for -- an L2 round trip --
// before the stage's first operand read, every stage.  The asm form is invisible to that bookkeeping; the caller waits
// (vmcnt(0)) and synchronises itself before anyone reads the target.  M0 = the LDS byte address of the wave's 1 KB
// segment (lane l lands at + 16 l).
template <int BYTES, int WAVES>
__device__ __forceinline__ void stage_gemm_tile_async(const float4* __restrict__ src, float* dst, int wave, int lane) {
    constexpr int ROUNDS = (BYTES + WAVES * 1024 - 1) / (WAVES * 1024);
    static_assert(BYTES % (WAVES * 1024) == 0, "whole 1 KB segments per wave and round");
    const unsigned lds0 = (unsigned)(size_t)(lds_ptr)dst;
    static_for<ROUNDS>([&](auto kk) {
        constexpr int k = decltype(kk)::value;
        const int seg = (k * WAVES + wave) * 1024;  // wave-uniform
        const char* g = reinterpret_cast<const char*>(src) + seg + lane * 16;
        const unsigned target = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)seg);
        unsigned saved_m0;  // (M0 is the compiler's own: put back, so that nothing it believes about M0 is wrong afterwards)
        asm volatile("s_mov_b32 %0, m0\n\t"
                     "s_mov_b32 m0, %1\n\t"
                     "s_nop 0\n\t"
                     "global_load_lds_dwordx4 %2, off\n\t"
                     "s_mov_b32 m0, %0"
                     : "=&s"(saved_m0)
                     : "s"(target), "v"(g)
                     : "memory");
    });
}

template <int MODEL, int D>
__global__ __launch_bounds__(kGW * 64, 2) void rank_gemm_kernel(
    const float* __restrict__ table, int64_t N, int64_t ld, const float4* __restrict__ img_head,
    const float4* __restrict__ img_tail, const float* __restrict__ key_true, const float* __restrict__ eps_q,
    int q_head, int q_tail, int n_quads, int chunks_head, int words_per_query,
    unsigned long long* __restrict__ acc, unsigned* __restrict__ flags, uint2* __restrict__ pairs) {
    constexpr int TILE_FLOATS = (D / 8) * 64 * 4;  // 16 KB at D = 128
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* buf0 = smem;
    float* buf1 = smem + TILE_FLOATS;
    unsigned* cnt = reinterpret_cast<unsigned*>(smem + gemm_buf_floats(D));  // [kGTilesPerChunk * 32]
    float* nrm = reinterpret_cast<float*>(cnt + kGTilesPerChunk * kGQT);  // [kGW][32] candidate norms
    uint2* pair_s = reinterpret_cast<uint2*>(nrm + kGW * 32);             // [kPairQuota] (query, row)
    unsigned* pair_n = reinterpret_cast<unsigned*>(pair_s + kPairQuota);
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;

    const int quad = blockIdx.x % n_quads, chunk = blockIdx.x / n_quads;
    const bool head = chunk < chunks_head;
    const int side_chunk = head ? chunk : chunk - chunks_head;
    const int n_side = head ? q_head : q_tail, q_base = head ? 0 : q_head;
    const int tile0 = side_chunk * kGTilesPerChunk;
    const int n_side_tiles = (n_side + kGQT - 1) / kGQT;
    const int n_tiles = n_side_tiles - tile0 < kGTilesPerChunk ? n_side_tiles - tile0 : kGTilesPerChunk;

    // candidate tile -> A operand + row norms (slabs alias the query buffers)
    float a[D / 2];
    const int ctile = quad * kGW + wave;
    const int64_t row0 = (int64_t)ctile * kGCT;
    const float my_norm = load_a_gemm<D>(a, table, N, ld, row0, smem + wave * (kGCT * kGSlab), lane);
    __syncthreads();  // slabs done before anything is staged over them
    if (lane < 32) nrm[wave * 32 + lane] = my_norm;
    for (int i = tid; i < kGTilesPerChunk * kGQT; i += kGW * 64) cnt[i] = 0;
    if (tid < kPairQuota) pair_s[tid] = make_uint2(kNoPair, 0u);
    if (tid == 0) *pair_n = 0;
    __syncthreads();
    float nrow[16];       // ||e|| of the candidate held by accumulator register r of this lane
    unsigned row_mask = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = acc_row(r, lane >> 5);
        nrow[r] = nrm[wave * 32 + row];
        row_mask |= (unsigned)(row0 + row < N) << r;
    }

    const float4* img = (head ? img_head : img_tail) + (int64_t)tile0 * (TILE_FLOATS / 4);
    stage_gemm_tile<TILE_FLOATS * 4>(img, buf0, wave, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int t = 0; t < n_tiles; ++t) {
        const float* cur = (t & 1) ? buf1 : buf0;
        if (t + 1 < n_tiles)
            stage_gemm_tile<TILE_FLOATS * 4>(img + (int64_t)(t + 1) * (TILE_FLOATS / 4), (t & 1) ? buf0 : buf1, wave, lane);
        const int q_local = (tile0 + t) * kGQT + (lane & 31);
        const bool q_ok = q_local < n_side;
        const int q = q_base + (q_ok ? q_local : n_side - 1);
        const float kt = key_true[q];
        const float eq = eps_q[q];

        // S~ tile = A (32 candidates x D) . B (D x 32 queries): D/2 MFMAs chained on one accumulator
        f32x16 s = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        const float4* bp = reinterpret_cast<const float4*>(cur) + lane;
        static_for<D / 8>([&](auto gg) {
            constexpr int g = decltype(gg)::value;
            const float4 b4 = bp[g * 64];
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(a[4 * g + 0], b4.x, s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(a[4 * g + 1], b4.y, s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(a[4 * g + 2], b4.z, s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(a[4 * g + 3], b4.w, s, 0, 0, 0);
        });

        // epilogue: three-way decision per pair; each lane owns a (query, 16-candidate) half-segment
        unsigned above = 0, und = 0;
        const float guard = fabsf(kt) * 2.4e-7f + 1e-36f;  // rounding of kt +- eps itself, underflow slack
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float eps = eq * nrow[r] + guard;
            const float v = s[r];
            const bool gt = v > kt + eps, lt = v < kt - eps, ok = (row_mask >> r) & 1u;
            above += ok && gt;
            und |= (unsigned)(ok && !(gt || lt)) << r;
        }
        if (q_ok) {
            bool listed = true;
            if (und) {
                const unsigned n_und = __popc(und);
                unsigned slot = atomicAdd(pair_n, n_und);
                listed = slot + n_und <= kPairQuota;
                if (listed) {
                    for (unsigned m = und; m; m &= m - 1)
                        pair_s[slot++] = make_uint2((unsigned)q, (unsigned)(row0 + acc_row(__builtin_ctz(m), lane >> 5)));
                } else {
                    const int hseg = ctile * 2 + (lane >> 5);
                    atomicOr(flags + (size_t)q * words_per_query + (hseg >> 5), 1u << (hseg & 31));
                }
            }
            if (listed && above) atomicAdd(cnt + t * kGQT + (lane & 31), above);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    for (int i = tid; i < n_tiles * kGQT; i += kGW * 64) {
        const int q_local = tile0 * kGQT + i;
        const unsigned long long c = cnt[i];
        if (q_local < n_side && c) atomicAdd(acc + q_base + q_local, c | (c << 32));  // certainly above: gt and ge
    }
    if (tid < kPairQuota) pairs[(size_t)blockIdx.x * kPairQuota + tid] = pair_s[tid];
}

// ------------------------------------------------------------------------------------------------
// bf16 x 3 variant of pass 1 (default): every f32 operand x is split into x = hi + lo + r with
// hi = bf16(x), lo = bf16(x - hi), |r| <= 2^-18 |x|, and W . e is taken as w_hi e_hi + w_hi e_lo + w_lo e_hi
// on v_mfma_f32_32x32x16_bf16 (products of bf16 pairs are exact in f32, the accumulator is f32).  Three
// bf16 MFMAs of K = 16 replace eight f32 MFMAs of K = 2: 5.3x less matrix-pipe time per tile.  The price
// is a wider band (kBandCBf16 u ||B_q|| ||e_c||): dropped terms 3.02 * 2^-18 = 194 u; the reference's
// own (n + 2) u = 130 u; operand rounding 3 u; and the accumulation inside the matrix pipe, whose internal
// rounding is not documented -- ASSUMED no worse than adding the products one at a time with a truncating f32
// addition (2 u each).  The MFMAs of a tile are ordered cross terms first (256 additions while the partial sum is
// <= 2^-7 of the final magnitude: 4 u), main products last (128 additions: 258 u): 262 u instead of the 775 u of an
// interleaved order.  Sum 589 u, taken as 620 u.
// Operands whose magnitudes would break the relative bounds (bf16 denormal flush below 1e-18, bf16
// overflow above 3e38, non-finite) get an infinite band factor and so take the exact path.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr float kBandCBf16 = 620.0f;
constexpr int kBfW = 4;          // waves per workgroup
constexpr int kBfTPW = 2;        // candidate tiles per wave (4 at one wave/SIMD spills and runs 1.8x slower)
constexpr int kBfWavesPerSimd = 2;  // launch bound: <= 256 VGPRs (128 of them hold the A operands)
constexpr int kBfMinTilesPerChunk = 16;          // query tiles per workgroup: 16 / 32 / 64, chosen per launch by a small cost model
constexpr int kBfMaxTilesPerChunk = 96;          // (rank_gemm_impl); the test knob takes any multiple of 4 up to 96 (from 92 on: one workgroup per CU)
constexpr int kBfResident = 512;                 // workgroups the chip holds (2 per CU)
constexpr int kBfSetupTiles = 5;                 // a workgroup's set-up, in query tiles of work
constexpr int kBfQuotaPerTile = 16 * kBfTPW;     // entries a workgroup can list, per query tile of its chunk (16 KB of LDS at 64 tiles: two workgroups
                                                 // per CU use 148 of the 160 KB; round 4: 8 per tile -- duplicate rows ran half the workgroups out of it)

// (x0, x1) -> packed bf16 pairs hi, lo (x0 in the low half)
__device__ __forceinline__ void split_bf16(float x0, float x1, unsigned& hi, unsigned& lo) {
    const f32x2 x = {x0, x1};
    const bf16x2 h = __builtin_convertvector(x, bf16x2);
    hi = __builtin_bit_cast(unsigned, h);
    const f32x2 rest = {x0 - __uint_as_float(hi << 16), x1 - __uint_as_float(hi & 0xffff0000u)};
    const bf16x2 l = __builtin_convertvector(rest, bf16x2);
    lo = __builtin_bit_cast(unsigned, l);
}

// Query tile image for the bf16 kernel: uint4 index (s * 64 + l) of the hi part, then of the lo part,
// holds W_{q = l & 31}[16 s + 8 (l >> 5) + j], j = 0..7, as four packed bf16 pairs.  A workgroup of 256 threads, query
// tile `vblock` of the th head-side + tt tail-side tiles.
template <int MODEL, int D>
__device__ __forceinline__ void prep_query_tile_bf16(int64_t vblock, const QRows& q_fixed, const QRows& q_rel, int64_t q_head,
                                                     int64_t q_tail, uint4* __restrict__ img_head,
                                                     uint4* __restrict__ img_tail, float* __restrict__ eps_q,
                                                     float (&w_s)[kGQT][D + 1]) {
    constexpr int STEPS = D / 16, U4 = STEPS * 64;  // uint4 per part
    const int64_t th = (q_head + kGQT - 1) / kGQT;
    const bool head = vblock < th;
    const int64_t tile = head ? vblock : vblock - th;
    const int64_t n_side = head ? q_head : q_tail;
    const int ql = threadIdx.x >> 3, sub = threadIdx.x & 7;
    const int64_t q_local = tile * kGQT + ql;
    const bool q_ok = q_local < n_side;
    const int64_t q = (head ? 0 : q_head) + (q_ok ? q_local : n_side - 1);
    const float* f = q_fixed.row(q);
    const float* r = q_rel.row(q);
    float bsq = 0.f, bmax = 0.f;
    bool bad = false;
    for (int i = 0; i < D / 32; ++i) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int k = 32 * i + 4 * sub + c;
            float w, b;
            if (head) gemm_operand<MODEL, HEAD>(f, r, k, D, w, b);
            else gemm_operand<MODEL, TAIL>(f, r, k, D, w, b);
            w_s[ql][k] = q_ok ? w : 0.f;
            bsq += b * b;
            bad |= !(b <= 3.0e38f);  // NaN too
            bmax = b > bmax ? b : bmax;
        }
    }
#pragma unroll
    for (int off = 1; off < 8; off <<= 1) {
        bsq += __shfl_xor(bsq, off);
        const float m2 = __shfl_xor(bmax, off);
        bmax = m2 > bmax ? m2 : bmax;
        bad |= (bool)__shfl_xor((int)bad, off);
    }
    if (sub == 0 && q_ok) eps_q[q] = kBandCBf16 * 5.9604645e-8f * band_norm(bsq, bad ? __builtin_inff() : bmax);
    __syncthreads();
    uint4* out = (head ? img_head : img_tail) + tile * (2 * U4);
    for (int idx = threadIdx.x; idx < U4; idx += 256) {
        const int st = idx >> 6, l = idx & 63;
        const float* w = &w_s[l & 31][16 * st + 8 * (l >> 5)];
        uint4 hi, lo;
        split_bf16(w[0], w[1], hi.x, lo.x);
        split_bf16(w[2], w[3], hi.y, lo.y);
        split_bf16(w[4], w[5], hi.z, lo.z);
        split_bf16(w[6], w[7], hi.w, lo.w);
        out[idx] = hi;
        out[U4 + idx] = lo;
    }
}

// Candidate image for the bf16 kernel (built once per call by prep_cand_bf16_kernel): per 32-row tile the hi parts of
// the D/16 K-steps, then the lo parts; uint4 index ((tile * 2 + part) * STEPS + st) * 64 + l holds
// e_{row = tile * 32 + (l & 31)}[16 st + 8 (l >> 5) + j], j = 0..7, as four packed bf16 pairs: a wave's load of one
// (part, K-step) is 1 KB contiguous, and lands in the registers as the MFMA A operand.  Rows past the table end are
// NaN (S~ = NaN is never decided and their bit of row_mask drops them).  cnmax[tile * 2 + half] = the largest band
// factor (||e|| rounded up, inf for rows the relative bounds do not cover) among the 16 rows the accumulator
// registers of a lane in that half hold.  Tiles are padded to whole workgroup loads (kBfW * kBfTPW tiles).
template <int D>
__device__ __forceinline__ void prep_cand_tile_bf16(int64_t tile, int lane, const float* __restrict__ table, int64_t N,
                                                    int64_t ld, uint4* __restrict__ cimg, float* __restrict__ cnmax,
                                                    float (&nrm)[32]) {  // one wave per tile; nrm: the wave's own
    constexpr int STEPS = D / 16;
    const int half = lane >> 5;
    const int64_t row = tile * kGCT + (lane & 31);
    const bool exists = row < N;
    const float* src = table + (exists ? row : 0) * ld + 8 * half;
    uint4* dst = cimg + (size_t)tile * 2 * STEPS * 64 + lane;
    float ss = 0.f, mx = 0.f;
    bool bad = false;
    static_for<STEPS>([&](auto kk) {
        constexpr int st = decltype(kk)::value;
        uint4 hi = make_uint4(0x7FC07FC0u, 0x7FC07FC0u, 0x7FC07FC0u, 0x7FC07FC0u), lo = make_uint4(0u, 0u, 0u, 0u);
        if (exists) {
            const float4 x = *reinterpret_cast<const float4*>(src + 16 * st);
            const float4 y = *reinterpret_cast<const float4*>(src + 16 * st + 4);
            split_bf16(x.x, x.y, hi.x, lo.x);
            split_bf16(x.z, x.w, hi.y, lo.y);
            split_bf16(y.x, y.y, hi.z, lo.z);
            split_bf16(y.z, y.w, hi.w, lo.w);
            const float v[8] = {x.x, x.y, x.z, x.w, y.x, y.y, y.z, y.w};
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                ss += v[j] * v[j];
                const float a = fabsf(v[j]);
                bad |= !(a <= 3.0e38f);  // NaN too
                mx = a > mx ? a : mx;
            }
        }
        dst[st * 64] = hi;
        dst[(STEPS + st) * 64] = lo;
    });
    ss += __shfl_xor(ss, 32);
    const float m2 = __shfl_xor(mx, 32);
    mx = m2 > mx ? m2 : mx;
    bad |= (bool)__shfl_xor((int)bad, 32);
    if (lane < 32) nrm[lane] = exists ? band_norm(ss, bad ? __builtin_inff() : mx) : 0.f;
    wave_lds_sync();
    if (lane < 2) {  // lane = half
        float m = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float nr = nrm[acc_row(r, lane)];
            m = nr > m ? nr : m;  // inf propagates; norms are never NaN
        }
        cnmax[tile * 2 + lane] = m;
    }
}

// Everything the bf16 pre-pass needs before its first MFMA, in ONE launch (they were four: a call with the reference's
// 128-query batch is a chain of ~5 us launches around an 11 us GEMM):
//   workgroups [0, q_blocks): a query tile each -- its operand image and band factors; its queries' rows of the flag
//     bitmap zeroed (slab 0; `words` per query); and, for calls of up to kTrueKeyLaneMaxQueries queries (true_keys), the
//     true-entity keys of the tile's queries, one lane each (above that the cooperative kernel does them, launched before);
//   workgroups after them: four candidate tiles each, one per wave -- the table's operand images.
template <int MODEL, int D>
__global__ __launch_bounds__(256) void gemm_prelude_bf16_kernel(
    const QRows q_fixed, const QRows q_rel, int64_t q_head, int64_t q_tail, uint4* __restrict__ img_head,
    uint4* __restrict__ img_tail, float* __restrict__ eps_q, unsigned q_blocks, const float* __restrict__ table, int64_t N,
    int64_t ld, int64_t cand_tiles, uint4* __restrict__ cimg, float* __restrict__ cnmax, unsigned* __restrict__ n_pairs,
    unsigned* __restrict__ flags, int words, int true_keys, const QRows q_true, float* __restrict__ key_true, unsigned long long* __restrict__ acc) {
    __shared__ float w_s[kGQT][D + 1];
    __shared__ float nrm[4][32];
    if (blockIdx.x >= q_blocks) {
        const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
        const int64_t tile = (int64_t)(blockIdx.x - q_blocks) * 4 + wave;
        if (tile < cand_tiles) prep_cand_tile_bf16<D>(tile, lane, table, N, ld, cimg, cnmax, nrm[wave]);
        return;
    }
    if (blockIdx.x == 0 && threadIdx.x < 64) n_pairs[threadIdx.x] = 0;  // the entry counter (and its padding)
    const int64_t th = (q_head + kGQT - 1) / kGQT;
    const bool head = blockIdx.x < th;
    const int64_t q0 = head ? (int64_t)blockIdx.x * kGQT : q_head + ((int64_t)blockIdx.x - th) * kGQT;
    const int64_t q1 = head ? (q0 + kGQT < q_head ? q0 + kGQT : q_head) : (q0 + kGQT < q_head + q_tail ? q0 + kGQT : q_head + q_tail);
    for (int64_t i = q0 * words + threadIdx.x; i < q1 * words; i += 256) flags[i] = 0;
    if (true_keys && threadIdx.x >= 192 && q0 + (threadIdx.x - 192) < q1)  // (the last wave: kGQT = 32 of its lanes)
        true_key_lane<MODEL, D>(q_true, q_fixed, q_rel, q0 + (threadIdx.x - 192), q_head, key_true, acc);
    prep_query_tile_bf16<MODEL, D>(blockIdx.x, q_fixed, q_rel, q_head, q_tail, img_head, img_tail, eps_q, w_s);
}

template <int D>
struct BfTile {
    uint4 hi[D / 16], lo[D / 16];  // A operands of the D/16 K-steps
};
// per (wave, candidate tile, half), kept in LDS (the registers are all taken): nmax = the largest band factor among the
// 16 candidates a lane's accumulators hold; row_mask bit 15 - r = the candidate of accumulator register r exists
struct BfHalfInfo { float nmax; unsigned row_mask; };

// -DBLP_TIMING: per-phase cycle counts of the kernel below, summed over waves (tools/gemm_phase_timing.py)
#ifdef BLP_TIMING
__device__ unsigned long long g_bf16_timing[8];
__device__ unsigned long long g_bf16_trace[3 * 4096];  // per workgroup: start, end (100 MHz wall ticks), HW_ID | XCC_ID << 32
#define BLP_T(i) do { const unsigned long long now_ = __builtin_readcyclecounter(); tacc[i] += now_ - tlast; tlast = now_; } while (0)
#else
#define BLP_T(i) do { } while (0)
#endif

// Two accumulator registers against the two thresholds: gm = 4 gm + {s0 > hi, s1 > hi}, lm likewise with < lo.
// Hand-written: the compiler turns `m = 2 m + (s > thr)` into compare + select + shift + or (7 VALU per register with
// both masks); here every compare shifts its result into the mask with one add-with-carry (m + m + carry): 4 VALU per
// register, no scalar instruction.  Two registers per block so that every carry is read >= 2 instructions after the
// compare that wrote it (gfx950: a VALU reading an SGPR a VALU wrote needs 2 wait states, and the assembler does not
// insert them inside inline asm).
__device__ __forceinline__ void decide_pair(unsigned& gm, unsigned& lm, float s0, float s1, float hi, float lo) {
    unsigned long long c0, c1, c2, c3;  // the four carries: one SGPR pair each
    asm("v_cmp_gt_f32_e64 %2, %6, %8\n\t"
        "v_cmp_lt_f32_e64 %3, %6, %9\n\t"
        "v_cmp_gt_f32_e64 %4, %7, %8\n\t"
        "v_cmp_lt_f32_e64 %5, %7, %9\n\t"
        "v_addc_co_u32_e64 %0, %2, %0, %0, %2\n\t"
        "v_addc_co_u32_e64 %1, %3, %1, %1, %3\n\t"
        "v_addc_co_u32_e64 %0, %4, %0, %0, %4\n\t"
        "v_addc_co_u32_e64 %1, %5, %1, %1, %5"
        : "+v"(gm), "+v"(lm), "=&s"(c0), "=&s"(c1), "=&s"(c2), "=&s"(c3)
        : "v"(s0), "v"(s1), "v"(hi), "v"(lo));
}

// The pipelined stage with the instruction order fixed by hand.  A query tile's 48 MFMAs (two candidate tiles, two
// accumulator chains alternating so that no MFMA reads the accumulator the previous one writes; bf16 products are
// exact in f32) are issued in TWO PHASES: first the 32 cross-term MFMAs (n += lo x bh, n += hi x bl for the eight
// K-steps), then the 16 main ones (n += hi x bh).  The cross terms are 2^-8 of the main ones, so the accumulator stays
// small while they are added and the matrix pipe's internal rounding of those 256 additions is negligible; only the
// 128 additions of the main phase see the full magnitude (this order is what the band constant kBandCBf16 prices).
// Behind the MFMAs, in their shadow, runs the decision of the CURRENT query tile's accumulators (decide_pair's eight
// instructions per register pair): a cross K-step decides one pair of candidate tile 0 (8 VALU behind 4 MFMAs), a
// main K-step one pair of candidate tile 1 (8 VALU behind 2 MFMAs).  Left to itself the compiler issues the 48
// MFMAs back to back and the 128 decision instructions after them.  Carries live in s[84:91] (clobbered): every
// carry is read >= 2 instructions after the compare that wrote it.  A and B operands are register quadruples of four
// packed bf16 pairs.  FIRST: the K-step that starts the accumulation (SrcC = 0: no register zeroing).
// BLP_GEMM_ASM_STAGE=0 builds the compiler-scheduled variant (A/B measurements).
#ifndef BLP_GEMM_ASM_STAGE
#define BLP_GEMM_ASM_STAGE 1
#endif
typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));  // (HIP's uint4 is a struct: not an asm operand)
__device__ __forceinline__ u32x4_t as_quad(const uint4& x) { return __builtin_bit_cast(u32x4_t, x); }
#define BLP_KSTEP_CLOBBERS "s84", "s85", "s86", "s87", "s88", "s89", "s90", "s91"
// -DBLP_GEMM_PRIO=1: the wave raises its issue priority for the K-steps of a stage (s_setprio 3 with its first MFMA, back to 0
// behind its last); =2: the other way round.  An experiment (tools/step_ab.py), not the shipped build.
#ifndef BLP_GEMM_PRIO
#define BLP_GEMM_PRIO 0
#endif
#if BLP_GEMM_PRIO == 1
#define BLP_PRIO_ENTER "s_setprio 3\n\t"
#define BLP_PRIO_LEAVE "\n\ts_setprio 0"
#elif BLP_GEMM_PRIO == 2
#define BLP_PRIO_ENTER "s_setprio 0\n\t"
#define BLP_PRIO_LEAVE "\n\ts_setprio 3"
#else
#define BLP_PRIO_ENTER ""
#define BLP_PRIO_LEAVE ""
#endif
#define BLP_KCROSS_BODY(C0, C1)                                                                             \
    asm("v_mfma_f32_32x32x16_bf16 %0, %4, %8, " C0 "\n\t"                                                    \
        "v_cmp_gt_f32_e64 s[84:85], %10, %12\n\t"                                                           \
        "v_cmp_lt_f32_e64 s[86:87], %10, %13\n\t"                                                           \
        "v_mfma_f32_32x32x16_bf16 %1, %6, %8, " C1 "\n\t"                                                    \
        "v_cmp_gt_f32_e64 s[88:89], %11, %12\n\t"                                                           \
        "v_cmp_lt_f32_e64 s[90:91], %11, %13\n\t"                                                           \
        "v_mfma_f32_32x32x16_bf16 %0, %5, %9, %0\n\t"                                                       \
        "v_addc_co_u32_e64 %2, s[84:85], %2, %2, s[84:85]\n\t"                                              \
        "v_addc_co_u32_e64 %3, s[86:87], %3, %3, s[86:87]\n\t"                                              \
        "v_mfma_f32_32x32x16_bf16 %1, %7, %9, %1\n\t"                                                       \
        "v_addc_co_u32_e64 %2, s[88:89], %2, %2, s[88:89]\n\t"                                              \
        "v_addc_co_u32_e64 %3, s[90:91], %3, %3, s[90:91]"
// cross K-step: n0 += lo0 x bh, n1 += lo1 x bh, n0 += hi0 x bl, n1 += hi1 x bl; decides (c0, c1) of candidate tile 0
template <bool FIRST>
__device__ __forceinline__ void kstep_cross(f32x16_t& n0, f32x16_t& n1, u32x4_t lo0, u32x4_t hi0, u32x4_t lo1, u32x4_t hi1,
                                            u32x4_t bh, u32x4_t bl, float c0, float c1, float th, float tl, unsigned& gm,
                                            unsigned& lm) {
    if constexpr (FIRST) {
#if BLP_GEMM_PRIO
        asm volatile(BLP_PRIO_ENTER "s_nop 0");
#endif
        BLP_KCROSS_BODY("0", "0")
            : "=&v"(n0), "=&v"(n1), "+v"(gm), "+v"(lm)
            : "v"(lo0), "v"(hi0), "v"(lo1), "v"(hi1), "v"(bh), "v"(bl), "v"(c0), "v"(c1), "v"(th), "v"(tl)
            : BLP_KSTEP_CLOBBERS);
    } else {
        BLP_KCROSS_BODY("%0", "%1")
            : "+v"(n0), "+v"(n1), "+v"(gm), "+v"(lm)
            : "v"(lo0), "v"(hi0), "v"(lo1), "v"(hi1), "v"(bh), "v"(bl), "v"(c0), "v"(c1), "v"(th), "v"(tl)
            : BLP_KSTEP_CLOBBERS);
    }
}
// main K-step: n0 += hi0 x bh, n1 += hi1 x bh; decides (c0, c1) of candidate tile 1
__device__ __forceinline__ void kstep_main(f32x16_t& n0, f32x16_t& n1, u32x4_t hi0, u32x4_t hi1, u32x4_t bh, float c0,
                                           float c1, float th, float tl, unsigned& gm, unsigned& lm) {
    asm("v_mfma_f32_32x32x16_bf16 %0, %4, %6, %0\n\t"
        "v_cmp_gt_f32_e64 s[84:85], %7, %9\n\t"
        "v_cmp_lt_f32_e64 s[86:87], %7, %10\n\t"
        "v_cmp_gt_f32_e64 s[88:89], %8, %9\n\t"
        "v_cmp_lt_f32_e64 s[90:91], %8, %10\n\t"
        "v_mfma_f32_32x32x16_bf16 %1, %5, %6, %1\n\t"
        "v_addc_co_u32_e64 %2, s[84:85], %2, %2, s[84:85]\n\t"
        "v_addc_co_u32_e64 %3, s[86:87], %3, %3, s[86:87]\n\t"
        "v_addc_co_u32_e64 %2, s[88:89], %2, %2, s[88:89]\n\t"
        "v_addc_co_u32_e64 %3, s[90:91], %3, %3, s[90:91]"
        : "+v"(n0), "+v"(n1), "+v"(gm), "+v"(lm)
        : "v"(hi0), "v"(hi1), "v"(bh), "v"(c0), "v"(c1), "v"(th), "v"(tl)
        : BLP_KSTEP_CLOBBERS);
}
#undef BLP_KCROSS_BODY
#undef BLP_KSTEP_CLOBBERS

// Undecided entries of the bf16 kernel: one per (query, 16-candidate half-tile) with at least one undecided pair.
//   x = query * 2 + half ;  y = (candidate tile within the slab) << 16 | mask, bit 15 - r of the mask = accumulator
//   register r = row acc_row(r, half) of the tile is undecided.
constexpr int kBfMaxSlabTiles = 65536 - 16;  // the tile index must fit 16 bits
#ifndef BLP_CHUNK_BLOCK
#define BLP_CHUNK_BLOCK 3  // measured fabric reads per launch (FB15k-237 DistMult): 1: 493 MB, 2: 335, 3: 331, 4: 359, 5: 410, 8: 608
#endif
constexpr unsigned kBfChunkBlock = BLP_CHUNK_BLOCK;       // query chunks whose workgroups share a candidate group's L2-resident image

// DUMP = true (tests only, blp_debug_gemm_dump): the same MFMA sequence and band arithmetic, but instead of deciding the
// kernel stores S~ and the band half-width eps of every (query, candidate) pair into dense (Q, n_rows) matrices.
template <int MODEL, int D, bool DUMP>
__global__ __launch_bounds__(kBfW * 64, kBfWavesPerSimd) void rank_gemm_bf16_kernel(
    const uint4* __restrict__ cimg, const float* __restrict__ cnmax, int64_t n_rows, const uint4* __restrict__ img_head,
    const uint4* __restrict__ img_tail, const float* __restrict__ key_true, const float* __restrict__ eps_q,
    int q_head, int q_tail, int n_groups, int chunks_head, int tiles_per_chunk, int words_per_query,
    unsigned long long* __restrict__ acc, unsigned* __restrict__ flags, uint2* __restrict__ pairs,
    unsigned* __restrict__ n_pairs, float* __restrict__ dump_s, float* __restrict__ dump_eps) {
    constexpr int STEPS = D / 16;
    constexpr int TILE_BYTES = 2 * STEPS * 64 * 16;  // hi + lo parts: 16 KB at D = 128
    const int quota = kBfQuotaPerTile * tiles_per_chunk;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* buf0 = smem;
    float* buf1 = smem + TILE_BYTES / 4;
    unsigned* cnt = reinterpret_cast<unsigned*>(smem + 2 * (TILE_BYTES / 4));   // [tiles_per_chunk * 32]
    uint2* pair_s = reinterpret_cast<uint2*>(cnt + tiles_per_chunk * kGQT);     // [quota]
    float2* kq_s = reinterpret_cast<float2*>(pair_s + quota);                    // [tiles_per_chunk * 32] {s_true, eps_q}
    BfHalfInfo* half_info = reinterpret_cast<BfHalfInfo*>(kq_s + tiles_per_chunk * kGQT);  // [kBfW][kBfTPW][2]
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63, half = lane >> 5;

    // Workgroups are dispatched to the 8 XCDs round-robin (block b runs on XCD b % 8), each with its own L2.  The
    // logical (candidate group, query chunk) index is laid out so that every XCD walks a contiguous range of it: the
    // workgroups that share a query chunk (its 1 MB image is what a workgroup streams through LDS) run on one XCD
    // and find it in that L2, instead of every XCD fetching every chunk.
    const unsigned n_blocks = gridDim.x, xcd = blockIdx.x & 7u, per_xcd = n_blocks >> 3, rem = n_blocks & 7u;
    const unsigned logical = (xcd < rem ? xcd * (per_xcd + 1) : rem * (per_xcd + 1) + (xcd - rem) * per_xcd) + (blockIdx.x >> 3);
    // Inside an XCD's range the workgroups go (block of kBfChunkBlock query chunks) x (candidate group) x (chunk of the
    // block): the ~64 workgroups resident on the XCD are then ~21 groups x 3 chunks, so a group's operand image (128 KB;
    // all groups together 7.4 MB -- more than the 4 MB L2) is fetched from the fabric once per three chunks instead of
    // once per chunk, while each chunk's query image is still streamed by workgroups that run together (larger blocks
    // thrash the L2 with query images).  Fabric reads 493 -> 331 MB per launch = 2.8 x the algorithmic bytes; the
    // kernel is MFMA / VALU-issue bound, its time does not change.
    const unsigned n_chunks_all = (unsigned)((n_blocks + n_groups - 1) / n_groups);
    const unsigned per_block = (unsigned)n_groups * kBfChunkBlock, cb = logical / per_block, in_block = logical % per_block;
    const unsigned chunks_here = n_chunks_all - cb * kBfChunkBlock < kBfChunkBlock ? n_chunks_all - cb * kBfChunkBlock : kBfChunkBlock;
    const int group = (int)(in_block / chunks_here), chunk = (int)(cb * kBfChunkBlock + in_block % chunks_here);
    const bool head = chunk < chunks_head;
    const int side_chunk = head ? chunk : chunk - chunks_head;
    const int n_side = head ? q_head : q_tail, q_base = head ? 0 : q_head;
    const int tile0 = side_chunk * tiles_per_chunk;
    const int n_side_tiles = (n_side + kGQT - 1) / kGQT;
    const int n_tiles = n_side_tiles - tile0 < tiles_per_chunk ? n_side_tiles - tile0 : tiles_per_chunk;

    for (int i = tid; i < tiles_per_chunk * kGQT; i += kBfW * 64) {
        cnt[i] = 0;
        const int q_local = tile0 * kGQT + i;
        const int q = q_base + (q_local < n_side ? q_local : n_side - 1);
        kq_s[i] = make_float2(key_true[q], eps_q[q]);
    }

    // this wave's candidate tiles: the pre-split A operands straight from the image (1 KB per load instruction)
    BfTile<D> c[kBfTPW];
    const int ctile0 = (group * kBfW + wave) * kBfTPW;
    static_for<kBfTPW>([&](auto tt) {
        constexpr int t = decltype(tt)::value;
        const uint4* src = cimg + (size_t)(ctile0 + t) * 2 * STEPS * 64 + lane;
        static_for<STEPS>([&](auto kk) {
            constexpr int st = decltype(kk)::value;
            c[t].hi[st] = src[st * 64];
            c[t].lo[st] = src[(STEPS + st) * 64];
        });
        if ((lane & 31) == 0) {
            const int64_t row0 = (int64_t)(ctile0 + t) * kGCT;
            unsigned mask = 0;
#pragma unroll
            for (int r = 0; r < 16; ++r) mask |= (unsigned)(row0 + acc_row(r, half) < n_rows) << (15 - r);
            half_info[(wave * kBfTPW + t) * 2 + half] = BfHalfInfo{cnmax[(ctile0 + t) * 2 + half], mask};
        }
    });
    const BfHalfInfo* my_info = half_info + wave * kBfTPW * 2 + half;  // [ti * 2]; written by this wave's own lanes
    // What a stage needs of it, wave-uniform (scalar registers: the vector registers are all taken, and an LDS read at the
    // top of every stage and two more in its settle step were three exposed round trips per query tile): the larger of a
    // tile's two band factors serves both halves (a wider band, never a narrower one); a tile with all of its 32 rows
    // needs no row mask.
    float nmax_tile[kBfTPW];
    bool tile_full[kBfTPW];
    static_for<kBfTPW>([&](auto tt) {
        constexpr int t = decltype(tt)::value;
        const float a = cnmax[(ctile0 + t) * 2], b = cnmax[(ctile0 + t) * 2 + 1];
        nmax_tile[t] = a > b ? a : b;  // (band factors are never NaN: band_norm)
        tile_full[t] = (int64_t)(ctile0 + t + 1) * kGCT <= n_rows;
    });

    const uint4* img = (head ? img_head : img_tail) + (int64_t)tile0 * (TILE_BYTES / 16);
    unsigned my_pairs = 0;  // entries this wave has listed (wave-uniform)
    const unsigned wave_quota = (unsigned)quota / kBfW;
    // K-step st of a query tile against both candidate tiles (bh / bl = its hi / lo B operands); the two
    // accumulator chains are interleaved so that no MFMA reads the accumulator the previous one writes
    auto load_b = [&](auto kk, const float* buf, bf16x8& bh, bf16x8& bl) {
        constexpr int st = decltype(kk)::value;
        const uint4* bp = reinterpret_cast<const uint4*>(buf) + lane;
        bh = __builtin_bit_cast(bf16x8, bp[st * 64]);
        bl = __builtin_bit_cast(bf16x8, bp[(STEPS + st) * 64]);
    };
    // A query tile without anything in the MFMAs' shadow (the first tile of a workgroup, dump mode, the
    // compiler-scheduled variant): the same two-phase order as kstep_cross / kstep_main -- all cross terms, then the
    // main products -- so that every accumulator sees the same sequence of additions whichever routine filled it.
    auto mfma_tile = [&](f32x16 (&s)[kBfTPW], const float* buf) {
        static_for<STEPS>([&](auto kk) {
            constexpr int st = decltype(kk)::value;
            bf16x8 bh, bl;
            load_b(kk, buf, bh, bl);
            static_for<kBfTPW>([&](auto tt) {
                constexpr int ti = decltype(tt)::value;
                s[ti] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, c[ti].lo[st]), bh, s[ti], 0, 0, 0);
            });
            static_for<kBfTPW>([&](auto tt) {
                constexpr int ti = decltype(tt)::value;
                s[ti] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, c[ti].hi[st]), bl, s[ti], 0, 0, 0);
            });
        });
        static_for<STEPS>([&](auto kk) {
            constexpr int st = decltype(kk)::value;
            bf16x8 bh, bl;
            load_b(kk, buf, bh, bl);
            static_for<kBfTPW>([&](auto tt) {
                constexpr int ti = decltype(tt)::value;
                s[ti] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, c[ti].hi[st]), bh, s[ti], 0, 0, 0);
            });
        });
    };
    // Decision for accumulator registers [R0, R0 + NR) of both tiles against the two thresholds of this
    // (query, 16-candidate half-tile): s_true +- (eps_q * max ||e|| + guard).  Using the half-tile's largest
    // band factor for all of its rows only widens the band.  Each compare shifts its result into a per-lane bit
    // mask with one add-with-carry (m = m + m + carry): after the 16 registers gm holds "certainly above", lm
    // "certainly below", register r at bit 15 - r; what is in neither is undecided (NaN: neither compare holds).
    auto decide_chunk = [&](auto r0, auto nr, const f32x16 (&s)[kBfTPW], const float (&thr_hi)[kBfTPW],
                            const float (&thr_lo)[kBfTPW], unsigned (&gm)[kBfTPW], unsigned (&lm)[kBfTPW]) {
        constexpr int R0 = decltype(r0)::value, NR = decltype(nr)::value;
        static_assert(NR % 2 == 0, "registers are decided in pairs");
        static_for<kBfTPW>([&](auto tt) {
            constexpr int ti = decltype(tt)::value;
            static_for<NR / 2>([&](auto jj) {
                constexpr int r = R0 + 2 * decltype(jj)::value;
                decide_pair(gm[ti], lm[ti], s[ti][r], s[ti][r + 1], thr_hi[ti], thr_lo[ti]);
            });
        });
    };
    // Counts and undecided entries of one (query tile, wave).  A lane whose 16 registers hold an undecided pair
    // lists ONE entry (query, half-tile, mask): the slot is the wave's scalar fill count plus the lane's rank among
    // the listing lanes (a ballot and a bit count, no loop, no LDS atomic).  When the wave's slice of the list
    // is full the lane's half-segment is flagged instead and contributes no count (the sweep recounts all of it).
    auto settle = [&](int t, int q, bool q_ok, const unsigned (&gm)[kBfTPW], const unsigned (&lm)[kBfTPW]) {
        unsigned total_above = 0;
        static_for<kBfTPW>([&](auto tt) {
            constexpr int ti = decltype(tt)::value;
            unsigned above = q_ok ? __popc(gm[ti]) : 0u;  // rows that do not exist score NaN: never above
            unsigned rows = 0xffffu;
            if (!tile_full[ti]) rows = my_info[ti * 2].row_mask;  // wave-uniform: the table's last tile only
            const unsigned und = q_ok ? ~(gm[ti] | lm[ti]) & rows : 0u;
            const unsigned long long listing = __ballot(und != 0);
            if (listing) {  // wave-uniform
                const unsigned n = __popcll(listing);
                const bool fits = my_pairs + n <= wave_quota;
                if (und) {
                    const int ctile = ctile0 + ti;
                    if (fits) {
                        const unsigned slot = my_pairs + __popcll(listing & ((1ull << lane) - 1ull));
                        pair_s[wave * wave_quota + slot] = make_uint2((unsigned)q * 2u + (unsigned)half, ((unsigned)ctile << 16) | und);
                    } else {
                        const int hseg = ctile * 2 + half;
                        atomicOr(flags + (size_t)q * words_per_query + (hseg >> 5), 1u << (hseg & 31));
                        above = 0;
                    }
                }
                my_pairs += fits ? n : 0u;
            }
            total_above += above;
        });
        if (total_above) atomicAdd(cnt + t * kGQT + (lane & 31), total_above);
    };
    auto zero_acc = [&](f32x16 (&s)[kBfTPW]) {
        static_for<kBfTPW>([&](auto tt) {
            s[decltype(tt)::value] = (f32x16){0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        });
    };
    // One pipeline stage: the MFMAs of query tile t + 1 (into `nxt`) are issued between the chunks of
    // tile t's decision arithmetic (on `cur`), so the vector ALU works in the matrix pipe's shadow.
#ifdef BLP_TIMING
    unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = __builtin_readcyclecounter();
    const unsigned long long wall0 = __builtin_amdgcn_s_memrealtime();  // the constant 100 MHz counter: [6] = wall ticks per wave
#endif
    auto stage = [&](int t, f32x16 (&cur)[kBfTPW], f32x16 (&nxt)[kBfTPW]) {
        const bool more = t + 1 < n_tiles;
        BLP_T(0);
        if (t + 2 < n_tiles)  // buf[t & 1] was last read by the MFMAs of tile t, one barrier ago
            stage_gemm_tile_async<TILE_BYTES, kBfW>(reinterpret_cast<const float4*>(img + (int64_t)(t + 2) * (TILE_BYTES / 16)),
                                                    (t & 1) ? buf1 : buf0, wave, lane);
        const int q_local = (tile0 + t) * kGQT + (lane & 31);
        const bool q_ok = q_local < n_side;
        const int q = q_base + (q_ok ? q_local : n_side - 1);
        const float2 kq = kq_s[t * kGQT + (lane & 31)];
        const float kt = kq.x, eq = kq.y;
        const float guard = fabsf(kt) * 2.4e-7f + 1e-35f;  // rounding of the thresholds themselves, product underflow
        unsigned gm[kBfTPW], lm[kBfTPW];
        float thr_hi[kBfTPW], thr_lo[kBfTPW], eps[kBfTPW];
        static_for<kBfTPW>([&](auto tt) {
            constexpr int ti = decltype(tt)::value;
            eps[ti] = __builtin_fmaf(eq, nmax_tile[ti], guard);
            thr_hi[ti] = kt + eps[ti];
            thr_lo[ti] = kt - eps[ti];
            gm[ti] = 0;
            lm[ti] = 0;
        });
        if (more) {
            const float* nbuf = ((t + 1) & 1) ? buf1 : buf0;
            if constexpr (!DUMP && BLP_GEMM_ASM_STAGE && STEPS == 8) {  // one register pair per K-step and phase: 16 = 2 x 8
                static_assert(kBfTPW == 2, "kstep_cross / kstep_main are written for two candidate tiles per wave");
                const u32x4_t* bp = reinterpret_cast<const u32x4_t*>(nbuf) + lane;
                u32x4_t bh[2], bl[2];
                bh[0] = bp[0];
                bl[0] = bp[STEPS * 64];
                static_for<STEPS>([&](auto kk) {  // cross terms; candidate tile 0 is decided
                    constexpr int st = decltype(kk)::value;
                    if constexpr (st + 1 < STEPS) {
                        bh[(st + 1) & 1] = bp[(st + 1) * 64];
                        bl[(st + 1) & 1] = bp[(STEPS + st + 1) * 64];
                    } else {
                        bh[(st + 1) & 1] = bp[0];  // the main phase starts over at K-step 0
                    }
                    kstep_cross<st == 0>(nxt[0], nxt[1], as_quad(c[0].lo[st]), as_quad(c[0].hi[st]), as_quad(c[1].lo[st]),
                                         as_quad(c[1].hi[st]), bh[st & 1], bl[st & 1], cur[0][2 * st], cur[0][2 * st + 1],
                                         thr_hi[0], thr_lo[0], gm[0], lm[0]);
                });
                static_for<STEPS>([&](auto kk) {  // main products; candidate tile 1 is decided
                    constexpr int st = decltype(kk)::value;
                    if constexpr (st + 1 < STEPS) bh[(st + 1) & 1] = bp[(st + 1) * 64];
                    kstep_main(nxt[0], nxt[1], as_quad(c[0].hi[st]), as_quad(c[1].hi[st]), bh[st & 1], cur[1][2 * st],
                               cur[1][2 * st + 1], thr_hi[1], thr_lo[1], gm[1], lm[1]);
                });
#if BLP_GEMM_PRIO
                asm volatile("s_nop 0" BLP_PRIO_LEAVE : "+v"(gm[1]));  // (behind the last K-step: it produces gm[1])
#endif
            } else {
                zero_acc(nxt);
                mfma_tile(nxt, nbuf);
                if constexpr (!DUMP) decide_chunk(ic<0>{}, ic<16>{}, cur, thr_hi, thr_lo, gm, lm);
            }
        } else if constexpr (!DUMP) {
            decide_chunk(ic<0>{}, ic<16>{}, cur, thr_hi, thr_lo, gm, lm);
        }
        BLP_T(1);
        if constexpr (DUMP) {
            static_for<kBfTPW>([&](auto tt) {
                constexpr int ti = decltype(tt)::value;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int64_t row = (int64_t)(ctile0 + ti) * kGCT + acc_row(r, half);
                    if (q_ok && row < n_rows) {
                        dump_s[(size_t)q * n_rows + row] = cur[ti][r];
                        dump_eps[(size_t)q * n_rows + row] = eps[ti];
                    }
                }
            });
        } else {
            settle(t, q, q_ok, gm, lm);
        }
        BLP_T(2);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        BLP_T(3);
        __syncthreads();
        BLP_T(4);
    };

    stage_gemm_tile<TILE_BYTES, kBfW>(reinterpret_cast<const float4*>(img), buf0, wave, lane);
    if (n_tiles > 1)
        stage_gemm_tile<TILE_BYTES, kBfW>(reinterpret_cast<const float4*>(img + TILE_BYTES / 16), buf1, wave, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    f32x16 acc_a[kBfTPW], acc_b[kBfTPW];
    zero_acc(acc_a);
    mfma_tile(acc_a, buf0);
    __syncthreads();  // buf0 is read: stage(0) may overwrite it with tile 2
    for (int t = 0; t < n_tiles; t += 2) {
        stage(t, acc_a, acc_b);
        if (t + 1 < n_tiles) stage(t + 1, acc_b, acc_a);
    }
    if constexpr (!DUMP) {
        for (int i = tid; i < n_tiles * kGQT; i += kBfW * 64) {
            const int q_local = tile0 * kGQT + i;
            const unsigned long long v = cnt[i];
            if (q_local < n_side && v) atomicAdd(acc + q_base + q_local, v | (v << 32));  // certainly above: gt and ge
        }
        // this wave's entries go to a slice of the global list reserved with one atomic (its own LDS writes are
        // visible to it in program order: no barrier)
        unsigned base = 0;
        if (lane == 0 && my_pairs) base = atomicAdd(n_pairs, my_pairs);
        base = __shfl(base, 0);
        for (unsigned i = lane; i < my_pairs; i += 64) pairs[base + i] = pair_s[wave * wave_quota + i];
    }
#ifdef BLP_TIMING
    BLP_T(5);
    if (lane == 0) {
        for (int i = 0; i < 6; ++i) atomicAdd(&g_bf16_timing[i], tacc[i]);
        atomicAdd(&g_bf16_timing[6], __builtin_amdgcn_s_memrealtime() - wall0);
        if (wave == 0 && blockIdx.x < 4096) {
            unsigned hw, xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            g_bf16_trace[3 * blockIdx.x] = wall0;
            g_bf16_trace[3 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
            g_bf16_trace[3 * blockIdx.x + 2] = (unsigned long long)hw | ((unsigned long long)xcc << 32);
        }
        atomicAdd(&g_bf16_timing[7], 1ull);
    }
#endif
}

// Between the passes (blocks of >= kFlagsToEntriesMinQueries queries): every flagged (query, 16-candidate) half-segment --
// a workgroup's slice of the list was full when a lane wanted to list it -- becomes an entry of the SPILL region with all of
// its existing rows marked, and its flag is cleared; the pair pass then re-scores it like any listed entry, spread over
// every half-wave of the chip.  Why: flags come in runs (the LAST queries of a chunk whose workgroups ran out of quota), and
// the sweep below, one wave per query and 64 consecutive queries per workgroup, took 3.7 ms for the 186 000 flags of the
// clustered FB15k-237 block (duplicate rows: ~29 ties per query) where the same rows cost the pair pass 0.4 ms.  A flag
// that finds the region full stays a flag (the sweep is still exact, only slower).  One thread per flag word.
constexpr int64_t kFlagsToEntriesMinQueries = 2048;
__global__ __launch_bounds__(256) void flags_to_entries_kernel(unsigned* __restrict__ flags, int64_t Q, int words_per_query,
                                                               int64_t n_rows, unsigned* __restrict__ counter,
                                                               uint2* __restrict__ spill, unsigned capacity) {
    const int64_t total = Q * words_per_query, stride = (int64_t)gridDim.x * 256;
    const int lane = threadIdx.x & 63;
    for (int64_t i0 = (int64_t)blockIdx.x * 256 + threadIdx.x - lane; i0 < total; i0 += stride) {  // wave-uniform trip count
        const int64_t i = i0 + lane;
        const unsigned w = i < total ? flags[i] : 0u;
        if (__ballot(w != 0) == 0) continue;
        // room for the wave's flags with one atomic: an exclusive prefix sum of the lanes' bit counts
        const unsigned mine = __popc(w);
        unsigned incl = mine;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const unsigned up = __shfl_up(incl, off);
            incl += lane >= off ? up : 0u;
        }
        const unsigned all = __shfl(incl, 63);
        unsigned base = 0;
        if (lane == 0) base = atomicAdd(counter, all);
        base = __shfl(base, 0);
        if (base + all > capacity) {  // (the counter overshoots from here on: readers clamp it; reserved slots below the capacity read as empty)
            for (unsigned k = base + lane; k < capacity && k < base + all; k += 64) spill[k] = make_uint2(0u, 0u);
            continue;
        }
        if (w) {
            const int64_t q = i / words_per_query;
            const int word = (int)(i - q * words_per_query);
            unsigned slot = base + incl - mine, bits = w;
            while (bits) {
                const int b = __builtin_ctz(bits);
                bits &= bits - 1;
                const int hseg = word * 32 + b, half = hseg & 1;
                const int64_t row0 = (int64_t)(hseg >> 1) * kGCT;
                unsigned mask = 0;  // bit 15 - r = accumulator register r = row acc_row(r, half) of the tile (the entries' convention)
#pragma unroll
                for (int r = 0; r < 16; ++r) mask |= (unsigned)(row0 + acc_row(r, half) < n_rows) << (15 - r);
                spill[slot++] = make_uint2((unsigned)q * 2u + (unsigned)half, ((unsigned)(hseg >> 1) << 16) | mask);
            }
            flags[i] = 0;
        }
    }
}

// Pass 2a: the listed entries, one per 32-lane half-wave; every undecided pair of the entry is re-scored exactly
// (coop_score) and counted.
template <int MODEL, int D>
__global__ __launch_bounds__(256) void refine_pairs_kernel(const float* __restrict__ table, int64_t ld,
                                                           const QRows q_fixed,
                                                           const QRows q_rel,
                                                           const float* __restrict__ key_true, int64_t q_head,
                                                           const uint2* __restrict__ pairs,
                                                           const unsigned* __restrict__ n_pairs,
                                                           const uint2* __restrict__ spill, unsigned spill_capacity,
                                                           unsigned long long* __restrict__ acc, const Gate gate) {
    if (gate_heavy(gate)) return;  // more flags than the spill region holds: the exact kernel re-ranks the block (rank_common.h: Gate)
    // the workgroups' list, then the entries made of flagged half-segments (flags_to_entries_kernel; counter n_pairs[1], clamped:
    // a wave that found the region full left it overshooting)
    const unsigned n_final = n_pairs[0], n_spill = n_pairs[1] < spill_capacity ? n_pairs[1] : spill_capacity;
    const unsigned long long n = (unsigned long long)n_final + n_spill;
    const int lane = threadIdx.x & 63, sub = lane & 31;
    const unsigned wave_id = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = (gridDim.x * blockDim.x) >> 6;
    for (unsigned long long e0 = (unsigned long long)wave_id * 2; e0 < n; e0 += (unsigned long long)n_waves * 2) {  // wave-uniform trip count
        const unsigned long long ei = e0 + (lane >> 5);
        const bool live = ei < n;
        const uint2 p = !live ? make_uint2(0u, 0u) : ei < n_final ? pairs[ei] : spill[ei - n_final];
        const int64_t q = p.x >> 1;
        const int half = p.x & 1;
        const int64_t row0 = (int64_t)(p.y >> 16) * kGCT;
        unsigned und = live ? p.y & 0xffffu : 0u;
        const bool is_head = q < q_head;
        const float* f = q_fixed.row(q);
        const float* r = q_rel.row(q);
        const float kt = key_true[q];
        unsigned gt = 0, ge = 0;
        while (__ballot(und != 0)) {  // the halves walk their own masks; an exhausted half idles
            const int bit = und ? 31 - __builtin_clz(und) : 0;
            const bool work = und != 0;
            und &= ~(1u << bit);
            const float* e = table + (row0 + acc_row(15 - bit, half)) * ld;
            float key;
            if (is_head) key = coop_score<MODEL, HEAD, D>(work ? e : table, f, r, sub);
            else key = coop_score<MODEL, TAIL, D>(work ? e : table, f, r, sub);
            gt += work && key > kt;
            ge += work && key >= kt;
        }
        if (sub == 0 && (gt | ge)) atomicAdd(acc + q, (unsigned long long)gt | ((unsigned long long)ge << 32));
    }
}

// Pass 2a of the f32-chain kernel: one lane per slot of the workgroups' fixed (query, row) pair regions.
template <int MODEL, int D>
__global__ __launch_bounds__(256) void refine_pair_slots_kernel(const float* __restrict__ table, int64_t ld,
                                                                const QRows q_fixed,
                                                                const QRows q_rel,
                                                                const float* __restrict__ key_true, int64_t q_head,
                                                                const uint2* __restrict__ pairs, int64_t n_entries,
                                                                unsigned long long* __restrict__ acc) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_entries) return;
    const uint2 p = pairs[i];
    if (p.x == kNoPair) return;
    const int64_t q = p.x;
    float e[D];
    load_row<D>(e, table + (int64_t)p.y * ld);
    float key;
    const float* f = q_fixed.row(q);
    const float* r = q_rel.row(q);
    if (q < q_head) key = Scorer<MODEL, HEAD, D>::template score<false>(e, LazyCoef<MODEL, HEAD, D>{f, r});
    else key = Scorer<MODEL, TAIL, D>::template score<false>(e, LazyCoef<MODEL, TAIL, D>{f, r});
    const float kt = key_true[q];
    const unsigned long long gt = key > kt, ge = key >= kt;
    if (gt | ge) atomicAdd(acc + q, gt | (ge << 32));
}

// Pass 2b: one wave per query.  The wave sweeps the query's flag words 64 at a time; every flagged (query, 16-candidate)
// half-segment is re-scored exactly, two half-segments per iteration: each 32-lane half of the wave walks the 16 rows of its
// half-segment with the cooperative routine of the pair pass (coop_score: lane j is accumulator A[j] of torch.sum's 32, every
// load one 128-byte line).  Round 4 gave every LANE a row of its own (load_row: 64 lanes gathering 16 bytes each from 64
// different rows per instruction, 128 registers of row per lane): fine while flags were rare, 3.7 ms for the 186 000 flags of
// the clustered FB15k-237 block (duplicate rows: ~29 ties per query) against 0.39 ms for as many rows on the pair list.
template <int MODEL, int D>
__global__ __launch_bounds__(256) void refine_kernel(const float* __restrict__ table, int64_t N, int64_t ld,
                                                     const QRows q_fixed,
                                                     const QRows q_rel,
                                                     const float* __restrict__ key_true, int64_t q_head, int64_t Q,
                                                     int words_per_query, const unsigned* __restrict__ flags,
                                                     unsigned long long* __restrict__ acc, const Gate gate, const FallbackPrep prep) {
    if (gate_heavy(gate)) {  // (rank_common.h: Gate) the exact kernel re-ranks the block; this grid prepares it
        fallback_prep<MODEL, D>(prep, q_fixed, q_rel, q_head, Q - q_head);
        return;
    }
    __shared__ int list[kSweepQueries], n_list;
    const int lane = threadIdx.x & 63, part = lane >> 5, sub = lane & 31;
    const int64_t q_base = (int64_t)blockIdx.x * kSweepQueries;
    const int n = flagged_queries(flags, q_base, Q, words_per_query, false, list, &n_list);
    for (int i = threadIdx.x >> 6; i < n; i += 4) {
    const int64_t q = q_base + list[i];
    const unsigned* row = flags + q * words_per_query;
    const float kt = key_true[q];
    const float* f = q_fixed.row(q);
    const float* rl = q_rel.row(q);
    const bool is_head = q < q_head;
    unsigned gt = 0, ge = 0;
    for (int w0 = 0; w0 < words_per_query; w0 += 64) {
        const unsigned mine = w0 + lane < words_per_query ? row[w0 + lane] : 0u;
        unsigned long long nonzero = __ballot(mine != 0);
        while (nonzero) {  // wave-uniform: next non-empty flag word
            const int src = __builtin_ctzll(nonzero);
            nonzero &= nonzero - 1;
            unsigned bits = __shfl(mine, src);
            const int hseg_base = (w0 + src) * 32;
            while (bits) {  // wave-uniform: the next two set bits, one per 32-lane half
                const int b0 = __builtin_ctz(bits);
                bits &= bits - 1;
                const int b1 = bits ? __builtin_ctz(bits) : -1;
                bits &= bits - 1;  // 0 stays 0
                const int b = part == 0 ? b0 : b1;
                const int hseg = hseg_base + (b < 0 ? 0 : b);
                const int64_t row0 = (int64_t)(hseg >> 1) * kGCT;
                const int hh = hseg & 1;
                for (int rr = 0; rr < 16; ++rr) {  // (both halves walk 16 rows; a half without a segment, or past the table, idles)
                    const int64_t r = row0 + acc_row(rr, hh);
                    const bool ok = b >= 0 && r < N;
                    const float* e = table + (ok ? r : 0) * ld;
                    float key;
                    if (is_head) key = coop_score<MODEL, HEAD, D>(e, f, rl, sub);
                    else key = coop_score<MODEL, TAIL, D>(e, f, rl, sub);
                    gt += ok && key > kt;
                    ge += ok && key >= kt;
                }
            }
        }
    }
    // every lane of a half holds that half's counts: lanes 0 and 32 carry them
    const unsigned gt2 = gt + __shfl(gt, 32), ge2 = ge + __shfl(ge, 32);
    if (lane == 0 && (gt2 | ge2)) atomicAdd(acc + q, (unsigned long long)gt2 | ((unsigned long long)ge2 << 32));
    }
}

// ------------------------------------------------------------------------------------------------
// Run-time guard behind assumption (A) of the bf16 band: the ISA does not say how
// v_mfma_f32_32x32x16_bf16 rounds its 16-term accumulation; the band prices it as "no worse than one truncating f32
// addition per product", which with the kernel's order (all cross terms, then the main products) bounds |S~ - S3| by
// 262 u T.  Tests measure that on the bench data -- the PRODUCT must not depend on a test having run on this stepping /
// firmware.  So, once per device, before the first bilinear pre-pass: a self-test kernel pushes adversarial operand
// sets through the very MFMA sequence of rank_gemm_bf16_kernel (same instruction, same operand layout, cross terms first)
// and compares every accumulator with the split sum evaluated exactly (f64: products of bf16 pairs have 16-bit
// significands, 384 of them add exactly enough in 53 bits) against HALF the bound the band uses.  Any violation (or a
// non-finite accumulator where the exact value is finite) routes every bilinear block of this device to the f32-chain
// pre-pass (rank_gemm_kernel: an exact fma chain, C = 320, provable without (A)); blp_device_caps reports the verdict.
// Operand sets, chosen per candidate row (A operand) and per query column (B operand) of a 32 x 32 tile, so that one
// tile holds all 64 combinations 16 times over, with the per-block seed varying the values:
//   0 same sign, magnitudes in [1, 2)              4 random signs, exponents spread over 2^-20 .. 2^20
//   1 one element of 2^20 among same-sign 2^-4s    5 tiny: 2^-60 scale (products at the edge of the f32 denormals)
//   2 alternating signs, equal magnitudes          6 huge: 2^50 scale
//   3 geometric decay 2^-(k/4)                     7 all-ones significands (longest carries), same sign
struct MfmaSelftest { float worst; unsigned bad; };  // per block: max |S~ - S3| / (262 u T + slack), #non-finite accumulators

__device__ __forceinline__ unsigned selftest_hash(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
// element k of operand vector `idx` (row of A / column of B) of pattern `idx & 7`, an f32 value
__device__ __forceinline__ float selftest_value(unsigned seed, unsigned idx, int k) {
    const unsigned h = selftest_hash(seed * 0x9e3779b9u + idx * 131u + (unsigned)k * 2654435761u);
    const float frac = 1.0f + (float)(h & 0xffffu) * (1.0f / 65536.0f);  // [1, 2): 16 significant bits -> hi and lo parts both live
    const float sign = (h >> 31) ? -1.0f : 1.0f;
    switch (idx & 7u) {
    case 0: return frac;
    case 1: return k == (int)((idx * 5u + seed) & 127u) ? 1048576.0f * frac : 0.0625f * frac;
    case 2: return (k & 1) ? -(1.0f + (float)((idx + seed) & 255u) * (1.0f / 256.0f)) : (1.0f + (float)((idx + seed) & 255u) * (1.0f / 256.0f));
    case 3: return frac * __builtin_ldexpf(1.0f, -(k >> 2));
    case 4: return sign * frac * __builtin_ldexpf(1.0f, (int)((h >> 16) & 31u) - 16);
    case 5: return sign * frac * __builtin_ldexpf(1.0f, -60);
    case 6: return sign * frac * __builtin_ldexpf(1.0f, 50);
    default: return 1.9921875f * 1.001953125f;  // hi = 1.1111111b, lo = 1.1111111b x 2^-9: all-ones significands in both parts
    }
}
__device__ __forceinline__ void selftest_split(float x, float& hi, float& lo) {
    unsigned h, l;
    split_bf16(x, 0.f, h, l);
    hi = __uint_as_float(h << 16);
    lo = __uint_as_float(l << 16);
}

__global__ __launch_bounds__(64) void mfma_selftest_kernel(MfmaSelftest* __restrict__ out) {
    constexpr int D = 128, STEPS = D / 16;
    const int lane = threadIdx.x, half = lane >> 5, col = lane & 31;
    const unsigned seed = blockIdx.x + 1u;
    // operands in the pre-pass kernel's layout: lane l holds, for K-step st, elements k = 16 st + 8 (l >> 5) + j, j = 0..7, of
    // candidate row (l & 31) (A) and of query column (l & 31) (B), as four packed bf16 pairs each, hi and lo parts
    uint4 a_hi[STEPS], a_lo[STEPS], b_hi[STEPS], b_lo[STEPS];
    static_for<STEPS>([&](auto kk) {
        constexpr int st = decltype(kk)::value;
        float av[8], bv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            av[j] = selftest_value(seed, (unsigned)col, 16 * st + 8 * half + j);
            bv[j] = selftest_value(seed ^ 0x5bd1e995u, (unsigned)col + 32u * (seed & 1u) + (seed >> 1), 16 * st + 8 * half + j);
        }
        split_bf16(av[0], av[1], a_hi[st].x, a_lo[st].x); split_bf16(av[2], av[3], a_hi[st].y, a_lo[st].y);
        split_bf16(av[4], av[5], a_hi[st].z, a_lo[st].z); split_bf16(av[6], av[7], a_hi[st].w, a_lo[st].w);
        split_bf16(bv[0], bv[1], b_hi[st].x, b_lo[st].x); split_bf16(bv[2], bv[3], b_hi[st].y, b_lo[st].y);
        split_bf16(bv[4], bv[5], b_hi[st].z, b_lo[st].z); split_bf16(bv[6], bv[7], b_hi[st].w, b_lo[st].w);
    });
    // the pre-pass's sequence for one accumulator: all cross terms (lo x bh, hi x bl per K-step), then the main products
    f32x16 s = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    static_for<STEPS>([&](auto kk) {
        constexpr int st = decltype(kk)::value;
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a_lo[st]), __builtin_bit_cast(bf16x8, b_hi[st]), s, 0, 0, 0);
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a_hi[st]), __builtin_bit_cast(bf16x8, b_lo[st]), s, 0, 0, 0);
    });
    static_for<STEPS>([&](auto kk) {
        constexpr int st = decltype(kk)::value;
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a_hi[st]), __builtin_bit_cast(bf16x8, b_hi[st]), s, 0, 0, 0);
    });
    // the exact split sum of each of this lane's 16 accumulators (column `col`, rows acc_row(r, half)), from the same values
    float worst = 0.f;
    unsigned bad = 0;
    const unsigned bidx = (unsigned)col + 32u * (seed & 1u) + (seed >> 1);
#pragma unroll 1
    for (int r = 0; r < 16; ++r) {
        const unsigned row = (unsigned)acc_row(r, half);
        double s3 = 0.0, t = 0.0;
#pragma unroll 4
        for (int k = 0; k < D; ++k) {
            float ah, al, bh, bl;
            selftest_split(selftest_value(seed, row, k), ah, al);
            selftest_split(selftest_value(seed ^ 0x5bd1e995u, bidx, k), bh, bl);
            const double p = (double)ah * (double)bh, c1 = (double)al * (double)bh, c2 = (double)ah * (double)bl;
            s3 += p + c1 + c2;
            t += fabs(p) + fabs(c1) + fabs(c2);
        }
        const float got = s[r];
        if (!(fabsf(got) <= 3.0e38f)) { ++bad; continue; }  // NaN / Inf where every exact value is finite
        // 262 u T is what the band allows the matrix pipe; 2^-118 covers flushed denormal products (the band's own
        // guard is 1e-35 = 2^-116); the verdict fails at HALF of the sum (the margin the GPU tests keep, too)
        const double bound = 262.0 * 5.9604644775390625e-8 * t + 3.009265538105056e-36;
        const float ratio = (float)(fabs((double)got - s3) / bound);
        worst = ratio > worst ? ratio : worst;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float w2 = __shfl_xor(worst, off);
        worst = w2 > worst ? w2 : worst;
        bad += __shfl_xor(bad, off);
    }
    if (lane == 0) out[blockIdx.x] = MfmaSelftest{worst, bad};
}

constexpr int kSelftestBlocks = 64;
constexpr int kMaxDevices = 64;
// The one piece of process-wide state of the product library (SURVEY 8b allows "a per-device lazily-built constant table
// guarded by a mutex"): per device 0 = not tested yet, 1 = (A) holds, 2 = violated.  Written once per device under the mutex.
static std::atomic<int> g_mfma_accum_state[kMaxDevices];
static std::atomic<float> g_mfma_accum_worst[kMaxDevices];
static std::mutex g_mfma_selftest_mutex;

int mfma_accum_state(int device) {
    return device >= 0 && device < kMaxDevices ? g_mfma_accum_state[device].load(std::memory_order_acquire) : 0;
}
float mfma_accum_worst(int device) {
    return device >= 0 && device < kMaxDevices ? g_mfma_accum_worst[device].load(std::memory_order_relaxed) : 0.f;
}
#ifdef BLP_TEST_HOOKS
void mfma_accum_reset(int device) {  // blp_debug_reset_selftest: the next bilinear block of this device tests again
    if (device >= 0 && device < kMaxDevices) g_mfma_accum_state[device].store(0, std::memory_order_release);
}
#endif

// Runs the self-test on `stream` with `scratch` (>= kSelftestBlocks * 8 bytes of device memory) and WAITS for it: the one
// host synchronisation of the library, once per device.  A stream that is being captured into a graph cannot be waited
// for: the verdict then stays open (returns 0) and the caller takes the provable path for this call.
int mfma_accum_selftest(int device, void* scratch, hipStream_t stream, hipError_t* err_out) {
    *err_out = hipSuccess;
    if (device < 0 || device >= kMaxDevices) return 2;  // (no slot to remember a verdict in: stay on the provable path)
    int state = g_mfma_accum_state[device].load(std::memory_order_acquire);
    if (state) return state;
    hipStreamCaptureStatus capture = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &capture) != hipSuccess || capture != hipStreamCaptureStatusNone) {
        (void)hipGetLastError();
        return 0;
    }
    std::lock_guard<std::mutex> lock(g_mfma_selftest_mutex);
    state = g_mfma_accum_state[device].load(std::memory_order_acquire);
    if (state) return state;
    MfmaSelftest host[kSelftestBlocks];
    mfma_selftest_kernel<<<dim3(kSelftestBlocks), 64, 0, stream>>>(static_cast<MfmaSelftest*>(scratch));
    hipError_t err = hipGetLastError();
    if (err == hipSuccess) err = hipMemcpyAsync(host, scratch, sizeof(host), hipMemcpyDeviceToHost, stream);
    if (err == hipSuccess) err = hipStreamSynchronize(stream);
    if (err != hipSuccess) { *err_out = err; return 0; }
    float worst = 0.f;
    unsigned bad = 0;
    for (const MfmaSelftest& b : host) {
        worst = b.worst > worst || b.worst != b.worst ? b.worst : worst;
        bad += b.bad;
    }
    if (knob(KNOB_MFMA_SELFTEST) == 1) worst = __builtin_inff();  // test knob: as if the matrix pipe had broken the bound
    state = (bad == 0 && worst < 0.5f) ? 1 : 2;
    g_mfma_accum_worst[device].store(worst, std::memory_order_relaxed);
    g_mfma_accum_state[device].store(state, std::memory_order_release);
    return state;
}

// ------------------------------------------------------------------------------------------------
struct GemmWorkspace {
    float* key_true; float* eps_q;
    float4* img_head; float4* img_tail;
    uint4* cimg; float* cnmax;  // bf16 candidate image (whole table) and its per-half-tile band factors
    unsigned long long* acc; unsigned* n_pairs; unsigned* flags; uint2* pairs;
    uint2* spill; int64_t spill_entries;  // bf16 kernel: entries made of flagged half-segments (flags_to_entries_kernel; counter n_pairs[1])
    float* fallback_coef;  // coefficient rows of the exact re-ranking (rank_common.h: Gate), or nullptr: this block has no fallback
    int64_t pass_ctiles;  // candidate tiles per GEMM + refine pass
    size_t bytes;
};

// Pair-list entries one pass over T candidate tiles can write: every workgroup owns a fixed region of its
// kernel's quota, so this is (candidate groups) x (query chunks) x quota -- for whichever of the two
// pass-1 kernels needs more (their groupings of tiles and of query tiles differ).
// Spill region of the bf16 path for a pass over T candidate tiles (where flagged half-segments become entries): room for one
// entry per 8 (query, tile) combinations = 1 / 16 of all half-segments (500 clusters of duplicate rows on the FB15k-237 block
// flag 1 / 500 of them), at most 2^31 entries.
static int64_t spill_entries(int64_t T, int64_t q_head, int64_t q_tail) {
    const int64_t e = (q_head + q_tail) * T / 8;
    return e < ((int64_t)1 << 31) - 64 ? e : ((int64_t)1 << 31) - 64;
}

static int64_t pair_entries(int64_t T, int64_t q_head, int64_t q_tail) {
    const int64_t th = (q_head + kGQT - 1) / kGQT, tt = (q_tail + kGQT - 1) / kGQT;
    auto chunks = [&](int64_t per) { return (th + per - 1) / per + (tt + per - 1) / per; };
    const int64_t f32_kernel = (T + kGW - 1) / kGW * chunks(kGTilesPerChunk) * kPairQuota;
    int64_t most = f32_kernel;
    // whichever chunking a launch picks: chunks(per) * per <= th + tt + 2 * per
    const int64_t bf16_kernel = (T + kBfW * kBfTPW - 1) / (kBfW * kBfTPW) * (th + tt + 2 * kBfMaxTilesPerChunk) * kBfQuotaPerTile;
    most = bf16_kernel > most ? bf16_kernel : most;
    return most;
}

static bool gemm_use_f32() { return knob(KNOB_GEMM_KERNEL) == 1; }  // the exact-f32-chain MFMA kernel instead of bf16 x 3

// Candidate tiles handled per GEMM + refine pass (a multiple of 16 = one flag word per query): the flag
// bitmap (2 bits per (query, tile)) and the pair regions (kPairQuota entries per workgroup) are each
// capped at ~256 MB; larger (Q x N) problems are processed in candidate slabs.
static int64_t tiles_per_pass(int64_t N, int64_t q_head, int64_t q_tail) {
    const int64_t Q = q_head + q_tail > 0 ? q_head + q_tail : 1;
    const int64_t cap = (int64_t)256 << 20;
    int64_t words = ((N + kGCT - 1) / kGCT + 15) / 16;
    const int64_t by_flags = cap / 4 / Q;
    // (entries grow linearly in whole 16-tile words; list + spill region together get three times the bitmap's cap)
    const int64_t by_pairs = 3 * cap / 8 / (pair_entries(16, q_head, q_tail) + spill_entries(16, q_head, q_tail));
    if (words > by_flags) words = by_flags;
    if (words > by_pairs) words = by_pairs;
    if (words * 16 > kBfMaxSlabTiles) words = kBfMaxSlabTiles / 16;  // an entry holds the slab-local tile in 16 bits
    if (const int64_t forced = knob(KNOB_GEMM_PASS_WORDS))  // test knob: force the multi-slab path
        if (forced > 0 && forced < words) words = forced;
    return (words < 1 ? 1 : words) * 16;
}

static GemmWorkspace carve_gemm(void* base, int D, int64_t N, int64_t q_head, int64_t q_tail) {
    GemmWorkspace w;
    const int64_t Q = q_head + q_tail;
    char* p = static_cast<char*>(base);
    size_t off = 0;
    const size_t tile_bytes = (size_t)(D / 8) * 64 * 16;
    w.key_true = reinterpret_cast<float*>(p + off);  off = align_up(off + (size_t)Q * 4, 256);
    w.eps_q = reinterpret_cast<float*>(p + off);     off = align_up(off + (size_t)Q * 4, 256);
    w.img_head = reinterpret_cast<float4*>(p + off); off = align_up(off + (size_t)((q_head + kGQT - 1) / kGQT) * tile_bytes, 256);
    w.img_tail = reinterpret_cast<float4*>(p + off); off = align_up(off + (size_t)((q_tail + kGQT - 1) / kGQT) * tile_bytes, 256);
    w.acc = reinterpret_cast<unsigned long long*>(p + off);   off = align_up(off + (size_t)Q * 8, 256);
    {   // candidate tiles padded to whole workgroup loads of the bf16 kernel
        const int64_t per_group = kBfW * kBfTPW;
        const int64_t tiles = ((N + kGCT - 1) / kGCT + per_group - 1) / per_group * per_group;
        w.cimg = reinterpret_cast<uint4*>(p + off);  off = align_up(off + (size_t)tiles * kGCT * D * 4, 256);
        w.cnmax = reinterpret_cast<float*>(p + off); off = align_up(off + (size_t)tiles * 2 * 4, 256);
    }
    w.pass_ctiles = tiles_per_pass(N, q_head, q_tail);
    w.n_pairs = reinterpret_cast<unsigned*>(p + off); off += 256;  // directly before the flags: one memset clears both
    w.flags = reinterpret_cast<unsigned*>(p + off);
    off = align_up(off + (size_t)Q * (size_t)(w.pass_ctiles / 16) * 4, 256);
    w.pairs = reinterpret_cast<uint2*>(p + off);
    off = align_up(off + (size_t)pair_entries(w.pass_ctiles, q_head, q_tail) * 8, 256);
    w.spill = reinterpret_cast<uint2*>(p + off);
    w.spill_entries = spill_entries(w.pass_ctiles, q_head, q_tail);
    off = align_up(off + (size_t)w.spill_entries * 8, 256);
    // the exact fallback: blocks of one candidate slab, of the size flags are turned into entries at, and >= kFallbackMinPairs pairs
    w.fallback_coef = nullptr;
    if (w.pass_ctiles * kGCT >= N && Q >= kFlagsToEntriesMinQueries && Q * N >= kFallbackMinPairs) {
        w.fallback_coef = reinterpret_cast<float*>(p + off);
        off = align_up(off + exact_fallback_coef_floats(D, q_head, q_tail) * 4, 256);
    }
    w.bytes = off;
    return w;
}

hipError_t gemm_prepass_stats(int D, int64_t N, int64_t q_head, int64_t q_tail, const void* workspace, int device, PrepassStats* out,
                              hipStream_t stream) {
    const GemmWorkspace w = carve_gemm(const_cast<void*>(workspace), D, N, q_head, q_tail);
    if (gemm_use_f32() || mfma_accum_state(device) != 1) {  // the f32-chain kernel's fixed pair regions are not counted
        out->path = 3;
        out->listed = out->flagged_rows = -1;
        return hipSuccess;
    }
    out->path = 2;
    const int64_t rows = N < w.pass_ctiles * kGCT ? N : w.pass_ctiles * kGCT;  // (the first slab's bitmap width; one slab as a rule)
    const int64_t words = ((rows + kGCT - 1) / kGCT + 15) / 16;
    unsigned long long host[3] = {0, 0, 0};
    hipError_t err = launch_count_bits(w.flags, (q_head + q_tail) * words, w.pairs, w.n_pairs, true, host, stream);
    out->flagged_rows = (long long)host[0] * 16;  // a flag = one (query, 16-candidate half-tile)
    out->listed = (long long)host[1];
    if (err != hipSuccess) return err;
    // ... and the entries made of flagged half-segments: all 16 rows each (slots a wave left empty count nothing; the counter may overshoot)
    err = launch_count_bits(w.flags, 0, w.spill, w.n_pairs + 1, true, host, stream, (unsigned)w.spill_entries);
    out->listed += (long long)host[1];
    return err;
}

constexpr int64_t kGemmMinQueries = 32;  // (one query tile: the chain's five launches are what a call this small costs)
bool rank_gemm_applicable(int model, int D, int64_t q_head, int64_t q_tail) {
    if (knob(KNOB_RANK_KERNEL) == 1) return false;  // test knob: the exact f32 kernels
    return (model == DISTMULT || model == COMPLEX || model == SIMPLE) && (D == 64 || D == 128) && q_head + q_tail >= kGemmMinQueries;
}

size_t rank_gemm_workspace_bytes(int model, int D, int64_t N, int64_t q_head, int64_t q_tail) {
    if (!rank_gemm_applicable(model, D, q_head, q_tail)) return 0;
    return carve_gemm(nullptr, D, N, q_head, q_tail).bytes;
}

// blp_debug_gemm_dump: the next bilinear pre-pass of this thread stores S~ and eps instead of deciding (tests)
#ifdef BLP_TEST_HOOKS
static thread_local float* g_dump_s = nullptr;
static thread_local float* g_dump_eps = nullptr;
void gemm_set_dump(float* s, float* eps) { g_dump_s = s; g_dump_eps = eps; }
#endif

template <int MODEL, int D>
static hipError_t rank_gemm_impl(const float* table, int64_t N, int64_t ld, const QRows q_fixed, const QRows q_rel,
                                 const QRows q_true, int64_t q_head, int64_t q_tail,
                                 const FilterSpec& filter, int32_t* counts, void* workspace, int n_cu,
                                 hipStream_t stream, hipEvent_t ev_start, hipEvent_t ev_stop) {
    const int64_t Q = q_head + q_tail;
    GemmWorkspace w = carve_gemm(workspace, D, N, q_head, q_tail);
    const int64_t pass_tiles = w.pass_ctiles;
    const int64_t th = (q_head + kGQT - 1) / kGQT, tt = (q_tail + kGQT - 1) / kGQT;
    // The bf16 x 3 pre-pass only on a device whose matrix pipe has passed the accumulation self-test (assumption (A) of the
    // band; once per device, the candidate-image region of the workspace as its scratch); a violation, or a verdict that
    // cannot be had now (the stream is being captured), takes the f32-chain pre-pass, whose band is arithmetic alone.
    int device = 0;
    (void)hipGetDevice(&device);
    hipError_t selftest_err = hipSuccess;
    const int verdict = gemm_use_f32() ? 1 : mfma_accum_selftest(device, w.cimg, stream, &selftest_err);
    if (selftest_err != hipSuccess) return selftest_err;
    const bool f32_kernel = gemm_use_f32() || verdict != 1;
    // the bf16 path folds the true keys of a small call into its prelude launch (below)
    const bool keys_in_prelude = !f32_kernel && Q <= kTrueKeyLaneMaxQueries;
    hipError_t err = hipSuccess;
    if (!keys_in_prelude)
        err = launch_true_keys(MODEL, D, q_fixed, q_rel, q_head, q_tail, q_true, w.key_true, w.acc, stream);
    if (err != hipSuccess) return err;
    const size_t lds_f32 = (size_t)gemm_buf_floats(D) * 4 + (size_t)kGTilesPerChunk * kGQT * 4 + (size_t)kGW * 32 * 4 +
                           (size_t)kPairQuota * 8 + 16;
#ifdef BLP_TEST_HOOKS
    float* dump_s = g_dump_s;
    float* dump_eps = g_dump_eps;
    g_dump_s = g_dump_eps = nullptr;  // one-shot: the pointers are not kept beyond this call
#else
    constexpr float* dump_s = nullptr;
    constexpr float* dump_eps = nullptr;
#endif
    if (dump_s && (f32_kernel || D != 128 || N > pass_tiles * kGCT)) return hipErrorInvalidValue;  // one slab of the bf16 kernel
    if (ev_start) (void)hipEventRecord(ev_start, stream);
    auto slab_words = [&](int64_t slab0) {  // flag words per query of the slab that starts at row slab0
        const int64_t n_rows = N - slab0 < pass_tiles * kGCT ? N - slab0 : pass_tiles * kGCT;
        return (int)(((n_rows + kGCT - 1) / kGCT + 15) / 16);
    };
    if (f32_kernel) {
        prep_gemm_kernel<MODEL, D><<<dim3((unsigned)(th + tt)), 256, 0, stream>>>(q_fixed, q_rel, q_head, q_tail,
                                                                                 w.img_head, w.img_tail, w.eps_q);
    } else {
        // query operand images + the table as bf16 hi / lo MFMA operands with band factors (once per call) + slab 0's
        // zeroed flag bitmap and entry counter (+ the true keys of a small call): one launch
        const int64_t per_group = kBfW * kBfTPW;
        const int64_t tiles = ((N + kGCT - 1) / kGCT + per_group - 1) / per_group * per_group;
        const int64_t blocks = th + tt + (tiles + 3) / 4;
        if (blocks > 0x7fffffff) return hipErrorInvalidValue;
        gemm_prelude_bf16_kernel<MODEL, D><<<dim3((unsigned)blocks), 256, 0, stream>>>(
            q_fixed, q_rel, q_head, q_tail, reinterpret_cast<uint4*>(w.img_head), reinterpret_cast<uint4*>(w.img_tail), w.eps_q,
            (unsigned)(th + tt), table, N, ld, tiles, w.cimg, w.cnmax, w.n_pairs, w.flags, slab_words(0), keys_in_prelude ? 1 : 0,
            q_true, w.key_true, w.acc);
    }
    for (int64_t slab0 = 0; slab0 < N; slab0 += pass_tiles * kGCT) {  // one iteration unless the bitmap is capped
        const int64_t n_rows = N - slab0 < pass_tiles * kGCT ? N - slab0 : pass_tiles * kGCT;
        const int64_t n_ctiles = (n_rows + kGCT - 1) / kGCT;
        const int tiles_per_group = f32_kernel ? kGW : kBfW * kBfTPW;
        const int64_t n_groups = (n_ctiles + tiles_per_group - 1) / tiles_per_group;
        const int words = slab_words(slab0);
        int tiles_per_chunk = kGTilesPerChunk;
        if (!f32_kernel) {
            // smallest estimated time: rounds of the resident workgroups (2 per CU) x (set-up + query tiles).  A
            // workgroup's set-up (candidate split, LDS tables, pipeline ramp, list write-out) costs about as much
            // as kBfSetupTiles query tiles (fitted on the FB15k-237 block: 1.64 / 1.49 / 1.47 ms at 16 / 32 / 64).
            int64_t best_cost = INT64_MAX;
            for (int per = 64; per >= kBfMinTilesPerChunk; per /= 2) {  // ([measured] 48 .. 88 in steps of 4: 0.998 - 1.035 ms against 1.004 at 64: flat)
                const int64_t grid = n_groups * ((th + per - 1) / per + (tt + per - 1) / per);
                const int64_t cost = ((grid + kBfResident - 1) / kBfResident) * (kBfSetupTiles + per);
                if (cost < best_cost) { best_cost = cost; tiles_per_chunk = per; }
            }
#ifdef BLP_GEMM_FORCE_TILES  // experiment builds (tools/step_ab.py): query tiles per workgroup whatever the cost model says
            tiles_per_chunk = BLP_GEMM_FORCE_TILES;
#endif
            const int forced = (int)knob(KNOB_GEMM_TILES_PER_CHUNK);  // test knob: long chunks on small problems
            if (forced >= kBfMinTilesPerChunk && forced <= kBfMaxTilesPerChunk && forced % 4 == 0) tiles_per_chunk = forced;
        }
        const int64_t chunks_head = (th + tiles_per_chunk - 1) / tiles_per_chunk;
        const int64_t chunks_tail = (tt + tiles_per_chunk - 1) / tiles_per_chunk;
        const size_t lds_bf16 = (size_t)2 * (2 * (D / 16) * 64 * 16) + (size_t)tiles_per_chunk * kGQT * 4 +
                                (size_t)kBfQuotaPerTile * tiles_per_chunk * 8 + (size_t)tiles_per_chunk * kGQT * 8 +
                                (size_t)kBfW * kBfTPW * 2 * 8 + 16;
        const int64_t n_blocks = n_groups * (chunks_head + chunks_tail);
        if (f32_kernel || slab0 > 0) {  // the entry counter and the flag bitmap (slab 0 of the bf16 path: zeroed by its prelude)
            err = hipMemsetAsync(w.n_pairs, 0, 256 + (size_t)Q * words * 4, stream);
            if (err != hipSuccess) return err;
        }
        const float* slab = table + slab0 * ld;
        Gate gate{nullptr, 0u};
        if (f32_kernel) {
            const int64_t n_entries = n_blocks * kPairQuota;
            rank_gemm_kernel<MODEL, D><<<dim3((unsigned)n_blocks), kGW * 64, lds_f32, stream>>>(
                slab, n_rows, ld, w.img_head, w.img_tail, w.key_true, w.eps_q, (int)q_head, (int)q_tail, (int)n_groups,
                (int)chunks_head, words, w.acc, w.flags, w.pairs);
            refine_pair_slots_kernel<MODEL, D><<<dim3((unsigned)((n_entries + 255) / 256)), 256, 0, stream>>>(
                slab, ld, q_fixed, q_rel, w.key_true, q_head, w.pairs, n_entries, w.acc);
        } else {
            const uint4* cimg = w.cimg + (size_t)(slab0 / kGCT) * (D / 8) * 64;  // 2 parts x D/16 steps x 64 lanes per tile
            const float* cnmax = w.cnmax + (slab0 / kGCT) * 2;
            const uint4* ih = reinterpret_cast<const uint4*>(w.img_head);
            const uint4* it = reinterpret_cast<const uint4*>(w.img_tail);
            if constexpr (D == 128) {
                if (dump_s) {
                    rank_gemm_bf16_kernel<MODEL, D, true><<<dim3((unsigned)n_blocks), kBfW * 64, lds_bf16, stream>>>(
                        cimg, cnmax, n_rows, ih, it, w.key_true, w.eps_q, (int)q_head, (int)q_tail, (int)n_groups,
                        (int)chunks_head, tiles_per_chunk, words, w.acc, w.flags, w.pairs, w.n_pairs, dump_s, dump_eps);
                    continue;
                }
            }
            rank_gemm_bf16_kernel<MODEL, D, false><<<dim3((unsigned)n_blocks), kBfW * 64, lds_bf16, stream>>>(
                cimg, cnmax, n_rows, ih, it, w.key_true, w.eps_q, (int)q_head, (int)q_tail, (int)n_groups, (int)chunks_head,
                tiles_per_chunk, words, w.acc, w.flags, w.pairs, w.n_pairs, nullptr, nullptr);
            // flagged half-segments of a large block become entries of the balanced pair pass (what stays flagged: the sweep below)
            if (Q >= kFlagsToEntriesMinQueries)
                flags_to_entries_kernel<<<dim3(1024), 256, 0, stream>>>(w.flags, Q, words, n_rows, w.n_pairs + 1, w.spill,
                                                                        (unsigned)w.spill_entries);
            // heavy <=> the flags outnumber the spill region (n_pairs[1] counts every flag, also those that found it full): exact
            // ties on whole percents of the table -- every flag costs 16 exact scores where the exact kernel re-ranks everything
            // (ComplEx switches at twice that: its exact kernel takes 27 ms for the FB15k-237 block, the others' 15)
            const int64_t heavy_from = w.spill_entries * (MODEL == COMPLEX ? 2 : 1);
            gate = Gate{w.fallback_coef ? w.n_pairs + 1 : nullptr, (unsigned)(heavy_from < 0xffffffffll ? heavy_from : 0xffffffffll)};
            refine_pairs_kernel<MODEL, D><<<dim3(2048), 256, 0, stream>>>(slab, ld, q_fixed, q_rel, w.key_true, q_head, w.pairs,
                                                                        w.n_pairs, w.spill, (unsigned)w.spill_entries, w.acc, gate);
        }
        refine_kernel<MODEL, D><<<dim3((unsigned)((Q + kSweepQueries - 1) / kSweepQueries)), 256, 0, stream>>>(
            slab, n_rows, ld, q_fixed, q_rel, w.key_true, q_head, Q, words, w.flags, w.acc, gate, FallbackPrep{w.fallback_coef, w.acc});
        if (gate.counter) {
            err = launch_exact_fallback(MODEL, D, table, N, ld, q_fixed, q_rel, q_head, q_tail, w.fallback_coef, w.key_true, w.acc, gate,
                                        n_cu, stream);
            if (err != hipSuccess) return err;
        }
    }
    if (ev_stop) (void)hipEventRecord(ev_stop, stream);
    err = launch_filter_finalize(MODEL, D, table, N, ld, q_fixed, q_rel, w.key_true, q_head, q_tail, filter,
                                 w.acc, counts, stream);
    return err != hipSuccess ? err : hipGetLastError();
}

hipError_t launch_rank_all_gemm(int model, int D, const float* table, int64_t N, int64_t ld,
                                const QRows q_fixed, const QRows q_rel, const QRows q_true, int64_t q_head, int64_t q_tail,
                                const FilterSpec& filter, int32_t* counts,
                                void* workspace, int n_cu, hipStream_t stream, hipEvent_t ev_start,
                                hipEvent_t ev_stop) {
#define BLP_GEMM_CASE(M, DD)                                                                                    \
    if (model == M && D == DD)                                                                                  \
        return rank_gemm_impl<M, DD>(table, N, ld, q_fixed, q_rel, q_true, q_head, q_tail, filter, \
                                     counts, workspace, n_cu, stream, ev_start, ev_stop);
    BLP_GEMM_CASE(DISTMULT, 128) BLP_GEMM_CASE(DISTMULT, 64)
    BLP_GEMM_CASE(COMPLEX, 128) BLP_GEMM_CASE(COMPLEX, 64)
    BLP_GEMM_CASE(SIMPLE, 128) BLP_GEMM_CASE(SIMPLE, 64)
#undef BLP_GEMM_CASE
    return hipErrorInvalidValue;
}

#ifdef BLP_TIMING
extern "C" int blp_debug_read_trace(unsigned long long* out, int n_blocks) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_bf16_trace), sizeof(unsigned long long) * 3 * (n_blocks < 4096 ? n_blocks : 4096));
}
extern "C" int blp_debug_read_timing(unsigned long long* out) {
    hipError_t err = hipMemcpyFromSymbol(out, HIP_SYMBOL(g_bf16_timing), sizeof(g_bf16_timing));
    unsigned long long zero[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (err == hipSuccess) err = hipMemcpyToSymbol(HIP_SYMBOL(g_bf16_timing), zero, sizeof(zero));
    return (int)err;
}
#endif

}  // namespace blp
