// inbatch_loss.hip -- LinkPrediction.compute_loss (models.py:51-70) with margin / nll loss
// (models.py:251-258) and L2 regularisation (models.py:261-266), forward and backward, for
// in-batch negatives.  The reference materialises ent_embs.view(2B, D)[neg_idx] as a (B, K, 2, D)
// temporary and runs ~40 small torch kernels per step; here the gather is implicit (rows are read
// from the 2B x D matrix, which is 64 KB at B = 64 and lives in L2) and a step is THREE launches:
//   fwd 1  inbatch_scores_kernel   16 lanes per (b, k) pair: coalesced row reads, the elementwise terms of the score
//                                  (models.py:222-248, each product / sum rounded as the reference rounds it), a
//                                  4-step wavefront shuffle reduction; the positive pair's lanes also leave the row's
//                                  share of the L2 regulariser;
//   fwd 2  inbatch_reduce_kernel   one block: the loss from the B + B K saved scores (f64 accumulation, fixed order);
//   bwd    inbatch_grad_kernel     entity rows and relation rows in one grid.  A workgroup owns R consecutive rows of
//                                  ent_embs.view(2B, D) and finds the negative pairs that reference them with ONE
//                                  stable compaction of neg_idx into LDS (entry order), so the work is
//                                  O(B K) per step -- not the O(B^2 K) of one full scan per row -- and every row's
//                                  contributions are added in entry order: no float atomics, gradients are
//                                  bit-reproducible run to run.
// Storage types: ent_embs / grad_ent in TE, rel_vecs / grad_rel in TR, each f32, f16 or bf16 (TR = TE or
// f32: under autocast the encoder output is half while nn.Embedding rows stay f32).  Half operands are
// widened exactly and every operation is the f32 one of the reference; gradients are rounded once on store.
// Floating point: a score is the reference's terms summed in a different (tree) order -- the loss agrees with the
// reference to ~1e-6 relative, gradients to ~1e-5 (tests/test_gpu_parity.py states the tolerances); the bit-exact
// score_fn is blp_score_fwd (score.hip).  No kernel here uses private scratch memory (tests/test_abi.py reads the
// code-object notes).  This path is launch / latency-bound (tens of KB of data): no roofline applies; DESIGN.md 4.6.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "launch.h"
#include "score_core.h"

#pragma clang fp contract(off)

namespace blp {

enum : int { LOSS_MARGIN = 0, LOSS_NLL = 1 };

template <class T>
__device__ __forceinline__ float widen(T x) { return (float)x; }

__device__ __forceinline__ float sign0(float x) { return x > 0.0f ? 1.0f : (x < 0.0f ? -1.0f : 0.0f); }

// ---------------------------------------------------------------- forward: scores
constexpr int kPairLanes = 16;                       // lanes that share one (b, k) pair
constexpr int kPairsPerBlock = 256 / kPairLanes;     // 16 pairs per 256-thread workgroup

// One lane's share of a pair's score: elements j = sub, sub + 16, ... of the reduction (width n = D, or D / 2 for the
// models that split the vector).  16 consecutive lanes read 16 consecutive elements: 64-byte (f32) segments of rows that
// sit in L2.  Every product and sum below is one f32 operation of the reference expression, in its order.
template <int MODEL, class TE, class TR>
__device__ __forceinline__ float pair_partial(const TE* __restrict__ h, const TE* __restrict__ t,
                                              const TR* __restrict__ r, int D, int sub) {
    float acc = 0.0f;
    if constexpr (MODEL == TRANSE) {         // -||h + r - t||_1
        for (int d = sub; d < D; d += kPairLanes) {
            float x = widen(h[d]) + widen(r[d]);
            x = x - widen(t[d]);
            acc = acc + fabsf(x);
        }
    } else if constexpr (MODEL == DISTMULT) {  // sum (h r) t
        for (int d = sub; d < D; d += kPairLanes) {
            const float x = widen(h[d]) * widen(r[d]);
            acc = acc + x * widen(t[d]);
        }
    } else if constexpr (MODEL == COMPLEX) {   // models.py:230-239
        const int H = D / 2;
        for (int j = sub; j < H; j += kPairLanes) {
            const float hr = widen(h[j]), hi = widen(h[H + j]), tr = widen(t[j]), ti = widen(t[H + j]);
            const float rr = widen(r[j]), ri = widen(r[H + j]);
            float a = rr * hr;  a = a * tr;
            float b = rr * hi;  b = b * ti;
            float c = ri * hr;  c = c * ti;
            float d = ri * hi;  d = d * tr;
            float s = a + b;
            s = s + c;
            acc = acc + (s - d);
        }
    } else {                                   // models.py:242-248 (the / 2 is applied to the sum)
        const int H = D / 2;
        for (int j = sub; j < H; j += kPairLanes) {
            float a = widen(h[j]) * widen(r[j]);      a = a * widen(t[H + j]);
            float b = widen(t[j]) * widen(r[H + j]);  b = b * widen(h[H + j]);
            acc = acc + (a + b);
        }
    }
    return acc;
}

__device__ __forceinline__ float reduce16(float v) {  // sum over the 16 lanes of a pair (every lane gets it)
#pragma unroll
    for (int off = kPairLanes / 2; off > 0; off >>= 1) v = v + __shfl_xor(v, off);
    return v;
}

// save_pos: (2 B) floats -- pos[b], then reg_part[b] = sum of squares of the head, tail and relation row of pair b.
template <int MODEL, class TE, class TR>
__global__ __launch_bounds__(256) void inbatch_scores_kernel(const TE* __restrict__ ent, const TR* __restrict__ rel,
                                                            const int64_t* __restrict__ neg_idx, int B, int K, int D,
                                                            float* __restrict__ pos, float* __restrict__ neg) {
    const int sub = threadIdx.x & (kPairLanes - 1);
    const int64_t pair = (int64_t)blockIdx.x * kPairsPerBlock + (threadIdx.x >> 4);  // over B * (K + 1), k == K: positive
    if (pair >= (int64_t)B * (K + 1)) return;  // (whole 16-lane groups leave together)
    const int b = (int)(pair / (K + 1)), k = (int)(pair % (K + 1));
    const TR* r = rel + (size_t)b * D;
    const TE* h;
    const TE* t;
    if (k == K) {  // positive pair: models.py:56-57
        h = ent + (size_t)(2 * b) * D;
        t = h + D;
    } else {       // negative pair: models.py:65-67
        const int64_t* idx = neg_idx + ((size_t)b * K + k) * 2;
        h = ent + idx[0] * D;
        t = ent + idx[1] * D;
    }
    float s = reduce16(pair_partial<MODEL>(h, t, r, D, sub));
    if constexpr (MODEL == TRANSE) s = -s;
    if constexpr (MODEL == SIMPLE) s = s / 2.0f;
    if (k == K) {
        float sq = 0.0f;  // models.py:261-266 on the positives' rows (read again: they are in L1 now)
        for (int d = sub; d < D; d += kPairLanes) {
            const float a = widen(h[d]), c = widen(t[d]), e = widen(r[d]);
            sq = sq + a * a;
            sq = sq + c * c;
            sq = sq + e * e;
        }
        sq = reduce16(sq);
        if (sub == 0) {
            pos[b] = s;
            pos[B + b] = sq;
        }
    } else if (sub == 0) {
        neg[(size_t)b * K + k] = s;
    }
}

__device__ __forceinline__ float softplus_torch(float x) {  // F.softplus, beta = 1, threshold = 20
    return x > 20.0f ? x : log1pf(expf(x));
}

__device__ __forceinline__ double block_sum(double v, double* sh) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    if (lane == 0) sh[wave] = v;
    __syncthreads();
    double total = 0.0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) total += sh[w];  // fixed order
    return total;
}

__global__ __launch_bounds__(1024) void inbatch_reduce_kernel(int loss, const float* __restrict__ pos,
                                                             const float* __restrict__ neg, int B, int K, int D,
                                                             float regularizer, float* __restrict__ out) {
    __shared__ double sh[16];
    const int tid = threadIdx.x, nt = blockDim.x;
    const int64_t BK = (int64_t)B * K;
    double model_loss;
    if (loss == LOSS_MARGIN) {  // models.py:251-254
        double s = 0.0;
        for (int64_t i = tid; i < BK; i += nt) {
            float l = 1.0f - pos[i / K];
            l = l + neg[i];
            s += l < 0.0f ? 0.0f : l;
        }
        model_loss = block_sum(s, sh) / (double)BK;
    } else {                    // models.py:257-258
        double sp = 0.0, sn = 0.0;
        for (int i = tid; i < B; i += nt) sp += softplus_torch(-pos[i]);
        for (int64_t i = tid; i < BK; i += nt) sn += softplus_torch(neg[i]);
        const double a = block_sum(sp, sh) / B;
        const double c = block_sum(sn, sh) / (double)BK;
        model_loss = (a + c) / 2.0;
    }
    double reg = 0.0;
    if (regularizer > 0.0f) {  // models.py:59-60, 261-266: (mean(h^2) + mean(t^2) + mean(r^2)) / 3 over the positives
        double s = 0.0;
        for (int i = tid; i < B; i += nt) s += (double)pos[B + i];
        reg = (double)regularizer * block_sum(s, sh) / ((double)B * D) / 3.0;
    }
    if (tid == 0) out[0] = (float)(model_loss + reg);
}

// ---------------------------------------------------------------- backward
// d loss / d neg[b, k] and d loss / d pos[b], recomputed from the saved scores.
__device__ __forceinline__ float dloss_dneg(int loss, float pos_b, float neg_bk, int B, int K) {
    if (loss == LOSS_MARGIN) {
        float l = 1.0f - pos_b;
        l = l + neg_bk;
        return l < 0.0f ? 0.0f : 1.0f / ((float)B * K);  // masked in place: gradient passes at l == 0
    }
    const float z = expf(neg_bk);
    return (neg_bk > 20.0f ? 1.0f : z / (z + 1.0f)) / (2.0f * B * K);
}

// by one wave: lanes over k
__device__ __forceinline__ float dloss_dpos_wave(int loss, float pos_b, const float* __restrict__ neg_b, int B, int K, int lane) {
    if (loss == LOSS_MARGIN) {
        int cnt = 0;
        for (int k = lane; k < K; k += 64) {
            float l = 1.0f - pos_b;
            l = l + neg_b[k];
            cnt += !(l < 0.0f);
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) cnt += __shfl_xor(cnt, off);
        return -(float)cnt / ((float)B * K);
    }
    const float x = -pos_b, z = expf(x);
    return -(x > 20.0f ? 1.0f : z / (z + 1.0f)) / (2.0f * B);
}

// d score / d operand[d] for operand in {0: heads, 1: tails, 2: rels}, elements read through f32-widening pointers.
template <int MODEL, class TE, class TR>
__device__ __forceinline__ float dscore(int operand, const TE* __restrict__ h, const TE* __restrict__ t,
                                        const TR* __restrict__ r, int d, int D) {
    if constexpr (MODEL == TRANSE) {
        const float sg = sign0(widen(h[d]) + widen(r[d]) - widen(t[d]));  // -|x|' = -sign(x); sign(0) = 0 like torch
        return operand == 1 ? sg : -sg;
    } else if constexpr (MODEL == DISTMULT) {
        const float hv = widen(h[d]), tv = widen(t[d]), rv = widen(r[d]);
        return operand == 0 ? rv * tv : (operand == 1 ? hv * rv : hv * tv);
    } else if constexpr (MODEL == COMPLEX) {
        const int H = D / 2;
        const bool im = d >= H;
        const int j = im ? d - H : d;
        const float hr = widen(h[j]), hi = widen(h[H + j]), tr = widen(t[j]), ti = widen(t[H + j]);
        const float rr = widen(r[j]), ri = widen(r[H + j]);
        // s = rr*hr*tr + rr*hi*ti + ri*hr*ti - ri*hi*tr
        if (operand == 0) return im ? rr * ti - ri * tr : rr * tr + ri * ti;
        if (operand == 1) return im ? rr * hi + ri * hr : rr * hr - ri * hi;
        return im ? hr * ti - hi * tr : hr * tr + hi * ti;
    } else {
        const int H = D / 2;
        const bool second = d >= H;
        const int j = second ? d - H : d;
        // s = (hh*ra*tt + th*rb*ht) / 2 ; h = [hh|ht], t = [th|tt], r = [ra|rb]
        if (operand == 0) return 0.5f * (second ? widen(t[j]) * widen(r[H + j]) : widen(r[j]) * widen(t[H + j]));
        if (operand == 1) return 0.5f * (second ? widen(h[j]) * widen(r[j]) : widen(r[H + j]) * widen(h[H + j]));
        return 0.5f * (second ? widen(t[j]) * widen(h[H + j]) : widen(h[j]) * widen(t[H + j]));
    }
}

constexpr int kGradWaves = 4;
constexpr int kGradSweep = 8;                 // elements per lane and sweep: d0 + lane + 64 i, i < 8 (512 per sweep)
constexpr int kHitCap = 4096;                 // hits a workgroup holds in LDS per round (more: further rounds)
constexpr int kScanSlices = 8;                // 64-entry slices of neg_idx per wave and scan step (2 048 entries per step)
constexpr int kMaxRowsPerBlock = 16;          // R: entity rows a workgroup owns
constexpr int kMaxTasks = 16;                 // (row, share) pairs of a workgroup: max(R, kGradWaves)
constexpr int kBatch = 4;                     // hits whose row loads are in flight together

// Rows per workgroup of the entity part: about 128 workgroups on big batches, never more hits expected than a quarter of
// what the LDS list holds (2 K per row on average).
static int grad_rows_per_block(int B, int K) {
    int R = (2 * B + 127) / 128;
    const int cap = kHitCap / (8 * (K > 0 ? K : 1));
    if (R > cap) R = cap;
    if (R > kMaxRowsPerBlock) R = kMaxRowsPerBlock;
    return R < 1 ? 1 : R;
}

// g += gn[n] * d score / d operand op[n] of pair (rows ih[n], it[n], relation row pb[n]), n < cnt, in that order; the
// loads of all cnt pairs are issued before the first dependent addition (cnt is wave-uniform).
template <int MODEL, class TE, class TR>
__device__ __forceinline__ void add_contributions(float (&g)[kGradSweep], const TE* __restrict__ ent, const TR* __restrict__ rel,
                                                  int D, int d0, int lane, int cnt, const int (&ih)[kBatch], const int (&it)[kBatch],
                                                  const int (&pb)[kBatch], const float (&gn)[kBatch], const int (&op)[kBatch]) {
#pragma unroll
    for (int i = 0; i < kGradSweep; ++i) {
        const int d = d0 + lane + 64 * i;
        if (d0 + 64 * i >= D) break;  // (wave-uniform)
        float v[kBatch];
#pragma unroll
        for (int n = 0; n < kBatch; ++n)
            v[n] = (n < cnt && d < D) ? dscore<MODEL>(op[n], ent + (size_t)ih[n] * D, ent + (size_t)it[n] * D, rel + (size_t)pb[n] * D, d, D) : 0.0f;
#pragma unroll
        for (int n = 0; n < kBatch; ++n)
            if (n < cnt) g[i] += gn[n] * v[n];
    }
}

// A wave consumes the set bits of `mask` (lanes holding a contribution: its pair's rows in hv / tv, relation row in bv,
// loss gradient in gv, operand in ov), kBatch at a time, in lane order.
template <int MODEL, class TE, class TR>
__device__ __forceinline__ void consume(float (&g)[kGradSweep], unsigned long long mask, int hv, int tv, int bv, float gv, int ov,
                                        const TE* __restrict__ ent, const TR* __restrict__ rel, int D, int d0, int lane) {
    while (mask) {
        int ih[kBatch], it[kBatch], pb[kBatch], op[kBatch];
        float gn[kBatch];
        int cnt = 0;
#pragma unroll
        for (int n = 0; n < kBatch; ++n) {
            ih[n] = it[n] = pb[n] = op[n] = 0;
            gn[n] = 0.0f;
            if (mask) {
                const int bit = __builtin_ctzll(mask);
                mask &= mask - 1;
                ih[n] = __builtin_amdgcn_readlane(hv, bit);
                it[n] = __builtin_amdgcn_readlane(tv, bit);
                pb[n] = __builtin_amdgcn_readlane(bv, bit);
                op[n] = __builtin_amdgcn_readlane(ov, bit);
                gn[n] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, gv), bit));
                cnt = n + 1;
            }
        }
        add_contributions<MODEL>(g, ent, rel, D, d0, lane, cnt, ih, it, pb, gn, op);
    }
}

// One grid: workgroups [0, ent_blocks) own R rows of ent_embs.view(2B, D) each; workgroups after them own one relation
// row (rel_shares == 4: its K negatives split over the four waves) or four (one per wave).
//
// Entity rows.  The negative pairs that reference row j are the entries e of neg_idx.view(-1) with neg_idx[e] == j
// (pair e >> 1, slot e & 1).  The workgroup scans neg_idx ONCE, 2 048 entries per step (wave w the w-th 512 of them, 64 per
// load instruction), and compacts the entries whose value lies in its row range into `hits` IN ENTRY ORDER (ballots and
// popcounts inside a wave, the waves' totals through LDS).  Then every (row, share) task -- a wave takes tasks w, w + 4, ...;
// a row is cut into `shares` tasks by pair index when the workgroup owns fewer rows than it has waves -- walks the list:
// 64 entries at a time the lanes fetch their pair's rows and loss gradient, the wave then adds gn(pair) * d score / d row
// for the entries of its task in list order, four pairs' row loads in flight together.  A task's partial sum is parked
// in LDS; at the end a row's shares are added in share order.  More than kHitCap hits in the range (every negative
// pointing at a few rows): the scan repeats for the next kHitCap, the parked sums carry over.
template <int MODEL, class TE, class TR>
__global__ __launch_bounds__(kGradWaves * 64) void inbatch_grad_kernel(
    int loss, const TE* __restrict__ ent, const TR* __restrict__ rel, const int64_t* __restrict__ neg_idx, int B, int K,
    int D, float regularizer, const float* __restrict__ grad_loss, const float* __restrict__ pos,
    const float* __restrict__ neg, TE* __restrict__ grad_ent, TR* __restrict__ grad_rel, int ent_blocks, int R,
    int rel_shares) {
    __shared__ int hits[kHitCap];
    __shared__ unsigned short hit_row[kHitCap];
    __shared__ int wave_count[kGradWaves];
    __shared__ float park[kMaxTasks][64 * kGradSweep];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const float gl = grad_loss[0];
    const float reg_scale = regularizer > 0.0f ? regularizer * 2.0f / (3.0f * B * D) : 0.0f;

    if ((int)blockIdx.x >= ent_blocks) {  // ---- relation rows: positive pair + its K negatives (models.py:67 broadcasts rels over K)
        const int rows_here = kGradWaves / rel_shares;  // 1 or 4
        const int b = ((int)blockIdx.x - ent_blocks) * rows_here + wave / rel_shares, my_share = wave % rel_shares;
        const bool active = b < B;
        const int bb = active ? b : B - 1;
        const TR* r = rel + (size_t)bb * D;
        const float pb = pos[bb];
        const float gp = dloss_dpos_wave(loss, pb, neg + (size_t)bb * K, B, K, lane);
        for (int d0 = 0; d0 < D; d0 += 64 * kGradSweep) {
            float g[kGradSweep];
#pragma unroll
            for (int i = 0; i < kGradSweep; ++i) {
                const int d = d0 + lane + 64 * i;
                g[i] = (d < D && my_share == 0)
                           ? gp * dscore<MODEL>(2, ent + (size_t)(2 * bb) * D, ent + (size_t)(2 * bb + 1) * D, r, d, D) : 0.0f;
            }
            // the K negatives, share s the k-ranges [s K / shares, (s + 1) K / shares): lanes fetch 64 pairs' indices and
            // loss gradients at once, the wave then consumes them in k order
            const int k_lo = (int)((int64_t)K * my_share / rel_shares), k_hi = (int)((int64_t)K * (my_share + 1) / rel_shares);
            for (int kb = k_lo; kb < k_hi && active; kb += 64) {
                const int k = kb + lane;
                const bool in = k < k_hi;
                const size_t pair = (size_t)bb * K + (in ? k : k_lo);
                const float gn = in ? dloss_dneg(loss, pb, neg[pair], B, K) : 0.0f;
                const int hv = (int)neg_idx[2 * pair], tv = (int)neg_idx[2 * pair + 1];
                consume<MODEL>(g, __ballot(in && gn != 0.0f), hv, tv, bb, gn, 2, ent, rel, D, d0, lane);
            }
            if (rel_shares > 1) {
                __syncthreads();
#pragma unroll
                for (int i = 0; i < kGradSweep; ++i) park[wave][lane + 64 * i] = g[i];
                __syncthreads();
                if (my_share == 0) {
#pragma unroll
                    for (int i = 0; i < kGradSweep; ++i)
                        for (int s = 1; s < rel_shares; ++s) g[i] += park[wave + s][lane + 64 * i];
                }
            }
            if (active && my_share == 0) {
#pragma unroll
                for (int i = 0; i < kGradSweep; ++i) {
                    const int d = d0 + lane + 64 * i;
                    if (d < D) grad_rel[(size_t)b * D + d] = (TR)(gl * (g[i] + reg_scale * widen(r[d])));
                }
            }
        }
        return;
    }

    // ---- entity rows [row0, row1)
    const int row0 = (int)blockIdx.x * R, row1 = row0 + R < 2 * B ? row0 + R : 2 * B;
    const int n_rows = row1 - row0;
    const int64_t total = 2ll * B * K;
    const int shares = n_rows >= kGradWaves ? 1 : (n_rows == 1 ? kGradWaves : kGradWaves / 2);  // tasks per row
    const int n_tasks = n_rows * shares;
    for (int d0 = 0; d0 < D; d0 += 64 * kGradSweep) {  // one sweep up to D = 512; wider rows repeat the scan
        for (int64_t skip = 0;; skip += kHitCap) {     // rounds of at most kHitCap hits (normally one)
            // -- scan: the entries that name a row of the range, in entry order
            int64_t seen = 0;  // hits of the range before the current scan step (the same in every thread)
            for (int64_t base = 0; base < total; base += (int64_t)kGradWaves * 64 * kScanSlices) {
                const int64_t e0 = base + (int64_t)wave * 64 * kScanSlices + lane;
                unsigned long long m[kScanSlices];
                int v[kScanSlices], mine = 0;
#pragma unroll
                for (int i = 0; i < kScanSlices; ++i) {
                    const int64_t e = e0 + 64 * i;
                    const int64_t x = e < total ? neg_idx[e] : -1;
                    v[i] = (int)(x - row0);
                    m[i] = __ballot(x >= row0 && x < row1);
                    mine += __popcll(m[i]);  // (wave-uniform: the wave's hits in this step)
                }
                __syncthreads();  // (the previous step's / round's readers of wave_count and hits are done)
                if (lane == 0) wave_count[wave] = mine;
                __syncthreads();
                int64_t ord = seen;
                int step_total = 0;
#pragma unroll
                for (int w = 0; w < kGradWaves; ++w) {
                    const int c = wave_count[w];
                    if (w < wave) ord += c;
                    step_total += c;
                }
#pragma unroll
                for (int i = 0; i < kScanSlices; ++i) {
                    const int64_t at = ord + __popcll(m[i] & ((1ull << lane) - 1ull)) - skip;
                    if ((m[i] >> lane & 1) && at >= 0 && at < kHitCap) {
                        hits[at] = (int)(e0 + 64 * i);
                        hit_row[at] = (unsigned short)v[i];
                    }
                    ord += __popcll(m[i]);
                }
                seen += step_total;
            }
            __syncthreads();
            const int n_hits = (int)(seen - skip < kHitCap ? (seen > skip ? seen - skip : 0) : kHitCap);
            const bool last_round = seen <= skip + kHitCap;
            // -- walk: every task adds the contributions of its entries, in list order
            for (int task = wave; task < n_tasks; task += kGradWaves) {
                const int row_local = task / shares, my_share = task % shares, my_row = row0 + row_local;
                float g[kGradSweep];
                if (skip == 0) {
#pragma unroll
                    for (int i = 0; i < kGradSweep; ++i) g[i] = 0.0f;
                    if (my_share == 0) {  // positive pair (2b, 2b + 1, rel b): the row is its head or its tail
                        const int b = my_row >> 1, slot = my_row & 1;
                        const float gp = dloss_dpos_wave(loss, pos[b], neg + (size_t)b * K, B, K, lane);
#pragma unroll
                        for (int i = 0; i < kGradSweep; ++i) {
                            const int d = d0 + lane + 64 * i;
                            if (d < D)
                                g[i] = gp * dscore<MODEL>(slot, ent + (size_t)(2 * b) * D, ent + (size_t)(2 * b + 1) * D,
                                                          rel + (size_t)b * D, d, D);
                        }
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < kGradSweep; ++i) g[i] = park[task][lane + 64 * i];
                }
                for (int base = 0; base < n_hits; base += 64) {
                    const bool in = base + lane < n_hits;
                    const int e = in ? hits[base + lane] : 0;
                    const int pair = e >> 1, pb = pair / K;
                    const bool ours = in && hit_row[base + lane] == row_local && pair % shares == my_share;
                    const float gn = ours ? dloss_dneg(loss, pos[pb], neg[pair], B, K) : 0.0f;
                    const int hv = (int)neg_idx[2 * (size_t)pair], tv = (int)neg_idx[2 * (size_t)pair + 1];
                    consume<MODEL>(g, __ballot(ours && gn != 0.0f), hv, tv, pb, gn, e & 1, ent, rel, D, d0, lane);
                }
#pragma unroll
                for (int i = 0; i < kGradSweep; ++i) park[task][lane + 64 * i] = g[i];
            }
            if (last_round) break;
        }
        __syncthreads();
        // -- a row's shares in share order, the regulariser's term, the store
        for (int row_local = wave; row_local < n_rows; row_local += kGradWaves) {
            const int my_row = row0 + row_local;
#pragma unroll
            for (int i = 0; i < kGradSweep; ++i) {
                const int d = d0 + lane + 64 * i;
                if (d < D) {
                    float g = park[row_local * shares][lane + 64 * i];
                    for (int s = 1; s < shares; ++s) g += park[row_local * shares + s][lane + 64 * i];
                    grad_ent[(size_t)my_row * D + d] = (TE)(gl * (g + reg_scale * widen(ent[(size_t)my_row * D + d])));
                }
            }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------- launchers
template <class F>
static hipError_t dispatch_model(int model, F f) {
    switch (model) {
    case TRANSE:   return f(std::integral_constant<int, TRANSE>{});
    case DISTMULT: return f(std::integral_constant<int, DISTMULT>{});
    case COMPLEX:  return f(std::integral_constant<int, COMPLEX>{});
    case SIMPLE:   return f(std::integral_constant<int, SIMPLE>{});
    default:       return hipErrorInvalidValue;
    }
}

template <class T> struct TypeTag { using type = T; };

// (ent dtype, rel dtype) with 0 = f32, 1 = f16, 2 = bf16; rel is either ent's type or f32
template <class F>
static hipError_t dispatch_types(int ent_dtype, int rel_dtype, F f) {
    if (ent_dtype == 0 && rel_dtype == 0) return f(TypeTag<float>{}, TypeTag<float>{});
    if (ent_dtype == 1 && rel_dtype == 1) return f(TypeTag<_Float16>{}, TypeTag<_Float16>{});
    if (ent_dtype == 1 && rel_dtype == 0) return f(TypeTag<_Float16>{}, TypeTag<float>{});
    if (ent_dtype == 2 && rel_dtype == 2) return f(TypeTag<__bf16>{}, TypeTag<__bf16>{});
    if (ent_dtype == 2 && rel_dtype == 0) return f(TypeTag<__bf16>{}, TypeTag<float>{});
    return hipErrorInvalidValue;
}

hipError_t launch_inbatch_loss_fwd(int model, int loss, int ent_dtype, int rel_dtype, const void* ent, const void* rel,
                                   const int64_t* neg_idx, int B, int K, int D, float regularizer,
                                   float* out_loss, float* save_pos, float* save_neg, hipStream_t stream) {
    return dispatch_model(model, [&](auto m) {
        return dispatch_types(ent_dtype, rel_dtype, [&](auto te, auto tr) {
            using TE = typename decltype(te)::type;
            using TR = typename decltype(tr)::type;
            const int64_t pairs = (int64_t)B * (K + 1), blocks = (pairs + kPairsPerBlock - 1) / kPairsPerBlock;
            if (blocks > 0x7fffffff) return hipErrorInvalidValue;
            inbatch_scores_kernel<decltype(m)::value, TE, TR><<<dim3((unsigned)blocks), 256, 0, stream>>>(
                static_cast<const TE*>(ent), static_cast<const TR*>(rel), neg_idx, B, K, D, save_pos, save_neg);
            inbatch_reduce_kernel<<<1, 1024, 0, stream>>>(loss, save_pos, save_neg, B, K, D, regularizer, out_loss);
            return hipGetLastError();
        });
    });
}

hipError_t launch_inbatch_loss_bwd(int model, int loss, int ent_dtype, int rel_dtype, const void* ent, const void* rel,
                                   const int64_t* neg_idx, int B, int K, int D, float regularizer,
                                   const float* grad_loss, const float* save_pos, const float* save_neg,
                                   void* grad_ent, void* grad_rel, hipStream_t stream) {
    return dispatch_model(model, [&](auto m) {
        return dispatch_types(ent_dtype, rel_dtype, [&](auto te, auto tr) {
            using TE = typename decltype(te)::type;
            using TR = typename decltype(tr)::type;
            const int R = grad_rows_per_block(B, K);
            // relation rows: a workgroup per row (its K negatives over four waves) while that does not flood the chip
            const int rel_shares = B <= 512 ? kGradWaves : 1;
            const int ent_blocks = (2 * B + R - 1) / R, rows_per_rel_block = kGradWaves / rel_shares;
            const int rel_blocks = (B + rows_per_rel_block - 1) / rows_per_rel_block;
            inbatch_grad_kernel<decltype(m)::value, TE, TR><<<dim3((unsigned)(ent_blocks + rel_blocks)), kGradWaves * 64, 0, stream>>>(
                loss, static_cast<const TE*>(ent), static_cast<const TR*>(rel), neg_idx, B, K, D, regularizer, grad_loss,
                save_pos, save_neg, static_cast<TE*>(grad_ent), static_cast<TR*>(grad_rel), ent_blocks, R, rel_shares);
            return hipGetLastError();
        });
    });
}

}  // namespace blp
