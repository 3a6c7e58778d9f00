// inbatch_loss.hip -- LinkPrediction.compute_loss (models.py:51-70) with margin / nll loss
// (models.py:251-258) and L2 regularisation (models.py:261-266), forward and backward, for
// in-batch negatives.  The reference materialises ent_embs.view(2B, D)[neg_idx] as a (B, K, 2, D)
// temporary and runs ~40 small torch kernels per step; here the gather is implicit (rows are read
// from the 2B x D matrix, which is 64 KB at B = 64 and lives in L2) and a step is THREE launches:
//   fwd 1  inbatch_scores_kernel   the B (K + 1) pair scores with coalesced row reads: the bilinear models 32 lanes per
//                                  pair (lane i = running sum i of torch.sum, then its fold by wavefront shuffles), TransE
//                                  one lane per pair (its L1 sum is one sequential chain) with the wave's 64 pairs' rows
//                                  fetched cooperatively through LDS -- scores bit-identical to the reference at the
//                                  scripts' widths; extra workgroups leave the rows' shares of the L2 regulariser;
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
// Floating point: the loss and the gradients are sums in this file's own (fixed) order: they agree with the reference
// to ~1e-6 / ~1e-5 relative (tests/test_gpu_parity.py states the tolerances).  No kernel here uses private scratch
// memory (tests/test_abi.py reads the code-object notes).  This path is launch / latency-bound (tens of KB of data): no roofline applies; DESIGN.md 4.6.
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

__device__ __forceinline__ void wave_lds_fence() {  // LDS accesses of one wave execute in order: only the compiler must not reorder
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ float sign0(float x) { return x > 0.0f ? 1.0f : (x < 0.0f ? -1.0f : 0.0f); }

constexpr int kReduceBlocks = 64, kReduceSlice = 8192;
// save_pos layout in floats: [0, B) positive scores | [B, 2B) regulariser shares | 6 kReduceBlocks: three f64 partial sums
// per block | 1: the ticket counter (zeroed by the scores kernel) -- include/blp_hip.h: BLP_INBATCH_SAVE_POS_FLOATS
__host__ __device__ inline int save_pos_partials_at(int B) { return 2 * B + (2 * B) % 2; }
__host__ __device__ inline int save_pos_ticket_at(int B) { return save_pos_partials_at(B) + 6 * kReduceBlocks; }

// ---------------------------------------------------------------- forward: scores
// Three ways to a pair's score, all with the reference's per-element arithmetic (models.py:222-248: each product / sum
// is one f32 operation, in its order); what differs is who adds the terms up, and in which order:
//   * TransE (any width): torch.norm(p=1) adds strictly left to right -- ONE sequential chain per pair (bit-identical
//     scores).  Widths % 16 == 0 up to 256: four lanes per pair form the |h + r - t| terms of a quarter of the row each
//     and the running sum walks through the quad (transe_quad); other widths: one lane per pair.  (Measured: the
//     cooperative LDS gather of exact_coop.h -- 64 pairs per wave, four dependent memory round trips per 128 columns --
//     took 11 us for the 4 160 pairs of a B = 64 step that keep one wave per CU busy.)
//   * DistMult / ComplEx / SimplE at reduction widths n % 32 == 0, n < 512 (the scripts' 128): torch.sum's order is 32
//     running sums, so 32 LANES per pair -- lane i IS accumulator i -- then the reference's fold (bit-identical scores).
//   * the bilinear models at any other width: 16 lanes per pair and a wavefront shuffle tree (not the reference's order:
//     ~1e-7 relative on scores of order 1; the order-exact any-width routine keeps 128 partial sums in scratch memory).
constexpr int kTreeLanes = 16;

template <class T> struct Vec4 { typedef T type __attribute__((ext_vector_type(4))); };

template <class T>
__device__ __forceinline__ void load4(const T* __restrict__ p, float (&v)[4]) {  // 4 consecutive elements, widened
    const typename Vec4<T>::type x = *reinterpret_cast<const typename Vec4<T>::type*>(p);
    v[0] = (float)x[0]; v[1] = (float)x[1]; v[2] = (float)x[2]; v[3] = (float)x[3];
}

// -(sum of |h + r - t|), left to right, by one lane from its own row pointers
template <class TE, class TR>
__device__ __forceinline__ float transe_lane(const TE* __restrict__ h, const TE* __restrict__ t, const TR* __restrict__ r, int D) {
    float acc = 0.0f;
    int d = 0;
    if ((D & 3) == 0) {
        for (; d < D; d += 4) {
            float hv[4], tv[4], rv[4];
            load4(h + d, hv); load4(t + d, tv); load4(r + d, rv);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float x = hv[k] + rv[k];
                x = x - tv[k];
                acc = acc + fabsf(x);
            }
        }
    }
    for (; d < D; ++d) {
        float x = widen(h[d]) + widen(r[d]);
        x = x - widen(t[d]);
        acc = acc + fabsf(x);
    }
    return -acc;
}

// The same sum by FOUR lanes (D % 16 == 0, D <= 4 * kQuadTerms): lane q of the quad loads columns [q D / 4, (q + 1) D / 4) of
// the three rows and forms their |h + r - t| terms -- elementwise, any lane gets the reference's bits -- and the running
// sum then walks through the quad: lane 0 adds its terms left to right and hands the sum to lane 1, ...  Still ONE chain in
// the reference's order (the result is bit-identical), but the loads and two of the three operations per element are
// spread over four lanes and a lane's rows are 8 x 16 contiguous bytes instead of 32.  Result in every lane of the quad.
constexpr int kQuadTerms = 64;
template <class TE, class TR>
__device__ __forceinline__ float transe_quad(const TE* __restrict__ h, const TE* __restrict__ t, const TR* __restrict__ r, int D,
                                             int lane) {
    const int q = lane & 3, n = D >> 2;  // n terms per lane, a multiple of 4
    float term[kQuadTerms];
#pragma unroll
    for (int k = 0; k < kQuadTerms; k += 4) {
        if (k < n) {
            float hv[4], tv[4], rv[4];
            load4(h + q * n + k, hv); load4(t + q * n + k, tv); load4(r + q * n + k, rv);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float x = hv[j] + rv[j];
                x = x - tv[j];
                term[k + j] = fabsf(x);
            }
        }
    }
    float acc = 0.0f;
#pragma unroll
    for (int step = 0; step < 4; ++step) {
        const float in = __shfl(acc, (lane & ~3) + (step > 0 ? step - 1 : 0));
        if (q == step) {
            if (step > 0) acc = in;
#pragma unroll
            for (int k = 0; k < kQuadTerms; ++k)
                if (k < n) acc = acc + term[k];
        }
    }
    return -__shfl(acc, lane | 3);
}

// term j of the bilinear models' sums (j < D, or j < D / 2 for the models that split the vector)
template <int MODEL, class TE, class TR>
__device__ __forceinline__ float bilinear_term(const TE* __restrict__ h, const TE* __restrict__ t, const TR* __restrict__ r,
                                               int j, int H) {
    if constexpr (MODEL == DISTMULT) {   // (h r) t
        const float x = widen(h[j]) * widen(r[j]);
        return x * widen(t[j]);
    } else if constexpr (MODEL == COMPLEX) {  // models.py:230-239
        const float hr = widen(h[j]), hi = widen(h[H + j]), tr = widen(t[j]), ti = widen(t[H + j]);
        const float rr = widen(r[j]), ri = widen(r[H + j]);
        float a = rr * hr;  a = a * tr;
        float b = rr * hi;  b = b * ti;
        float c = ri * hr;  c = c * ti;
        float d = ri * hi;  d = d * tr;
        float s = a + b;
        s = s + c;
        return s - d;
    } else {                                  // models.py:242-248 (the / 2 is applied to the sum)
        float a = widen(h[j]) * widen(r[j]);      a = a * widen(t[H + j]);
        float b = widen(t[j]) * widen(r[H + j]);  b = b * widen(h[H + j]);
        return a + b;
    }
}

__host__ __device__ inline bool torch_sum_in_registers(int model, int D) {  // torch.sum's 32-accumulator case
    const int n = model == DISTMULT ? D : D / 2;
    return model != TRANSE && n % 32 == 0 && n >= 32 && n < 512;
}
__host__ __device__ inline int lanes_per_pair(int model, int D) {
    if (model == TRANSE) return (D % 16 == 0 && D <= 4 * kQuadTerms) ? 4 : 1;
    return torch_sum_in_registers(model, D) ? 32 : kTreeLanes;
}

// save_pos: (2 B) floats -- pos[b], then (regularizer > 0) the sum of squares of the head, tail and relation row of pair b.
// Workgroups [0, pair_blocks): the scores; workgroups after them (regularizer > 0 only): one wave per positive triple,
// its three rows' squares (models.py:261-266).
template <int MODEL, class TE, class TR>
__global__ __launch_bounds__(256) void inbatch_scores_kernel(const TE* __restrict__ ent, const TR* __restrict__ rel,
                                                            const int64_t* __restrict__ neg_idx, int B, int K, int D,
                                                            float* __restrict__ pos, float* __restrict__ neg,
                                                            unsigned pair_blocks) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (blockIdx.x == 0 && threadIdx.x == 0)  // the reduce kernel's ticket counter (behind the scores in save_pos)
        *reinterpret_cast<unsigned*>(pos + save_pos_ticket_at(B)) = 0u;
    if (blockIdx.x >= pair_blocks) {  // the regulariser's shares
        const int b = (int)(blockIdx.x - pair_blocks) * 4 + wave;
        if (b >= B) return;
        const TE* h = ent + (size_t)(2 * b) * D;
        const TR* r = rel + (size_t)b * D;
        float sq = 0.0f;
        for (int d = lane; d < 2 * D; d += 64) sq = sq + widen(h[d]) * widen(h[d]);  // head and tail rows are adjacent
        for (int d = lane; d < D; d += 64) sq = sq + widen(r[d]) * widen(r[d]);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) sq = sq + __shfl_xor(sq, off);
        if (lane == 0) pos[B + b] = sq;
        return;
    }
    const int64_t n_pairs = (int64_t)B * (K + 1);
    const int per_pair = lanes_per_pair(MODEL, D), sub = threadIdx.x & (per_pair - 1);
    const int64_t slot = (int64_t)blockIdx.x * (256 / per_pair) + threadIdx.x / per_pair;  // over B * (K + 1); k == K: positive
    const bool live = slot < n_pairs;
    const int64_t pair = live ? slot : n_pairs - 1;  // (idle lanes redo the last pair: every lane has readable rows)
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
    float s;
    if constexpr (MODEL == TRANSE) {
        s = per_pair == 4 ? transe_quad(h, t, r, D, lane) : transe_lane(h, t, r, D);
    } else if (per_pair == 32) {  // torch.sum's order: lane i = running sum i, then its fold
        const int H = MODEL == DISTMULT ? D : D / 2;
        float a = 0.0f + bilinear_term<MODEL>(h, t, r, sub, H);
        for (int c = 1; c < H / 32; ++c) a = a + bilinear_term<MODEL>(h, t, r, 32 * c + sub, H);
        float v = a + __shfl_down(a, 8, 32);
        v = v + __shfl_down(a, 16, 32);
        v = v + __shfl_down(a, 24, 32);
        s = 0.0f;
#pragma unroll
        for (int l = 0; l < 8; ++l) s = s + __shfl(v, l, 32);
        if constexpr (MODEL == SIMPLE) s = s / 2.0f;
    } else {                      // any other width: a shuffle tree over 16 lanes
        const int H = MODEL == DISTMULT ? D : D / 2;
        float a = 0.0f;
        for (int j = sub; j < H; j += kTreeLanes) a = a + bilinear_term<MODEL>(h, t, r, j, H);
#pragma unroll
        for (int off = kTreeLanes / 2; off > 0; off >>= 1) a = a + __shfl_xor(a, off);
        s = MODEL == SIMPLE ? a / 2.0f : a;
    }
    if (live && sub == 0) {
        if (k == K) pos[b] = s;
        else neg[(size_t)b * K + k] = s;
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

// The loss from the saved scores.  Up to kReduceBlocks workgroups each add a slice of the B K negatives (block 0 also the
// B positives' and the regulariser's terms) in f64 and leave three partial sums behind the scores in save_pos; the
// workgroup that takes the last ticket adds the partials in block order -- one launch, a fixed summation order, no float
// atomics.  (One block for 65 536 negatives -- B = 1 024 -- was a 17 us latency-bound loop.)

__global__ __launch_bounds__(1024) void inbatch_reduce_kernel(int loss, float* __restrict__ pos,
                                                             const float* __restrict__ neg, int B, int K, int D,
                                                             float regularizer, float* __restrict__ out) {
    __shared__ double sh[16];
    __shared__ unsigned ticket_sh;
    const int tid = threadIdx.x, nt = blockDim.x, G = gridDim.x, g = blockIdx.x;
    const int BK = B * K;  // (the launcher refuses B K >= 2^30)
    const int per = (BK + G - 1) / G, lo = g * per, hi = lo + per < BK ? lo + per : BK;
    constexpr int U = 8;  // loads in flight per thread
    double part[3] = {0.0, 0.0, 0.0};  // negatives' terms | positives' softplus terms | squares
    if (loss == LOSS_MARGIN) {  // models.py:251-254
        double s = 0.0;
        for (int i0 = lo + tid; i0 < hi; i0 += nt * U) {
            float p[U], n[U];
#pragma unroll
            for (int j = 0; j < U; ++j) {
                const int i = i0 + j * nt;
                p[j] = i < hi ? pos[(unsigned)i / (unsigned)K] : 0.0f;
                n[j] = i < hi ? neg[i] : -3.0e38f;  // (a padded slot's hinge is negative: it adds nothing)
            }
#pragma unroll
            for (int j = 0; j < U; ++j) {
                float l = 1.0f - p[j];
                l = l + n[j];
                s += l < 0.0f ? 0.0f : l;
            }
        }
        part[0] = block_sum(s, sh);
    } else {                    // models.py:257-258
        double sp = 0.0, sn = 0.0;
        if (g == 0)
            for (int i = tid; i < B; i += nt) sp += softplus_torch(-pos[i]);
        for (int i0 = lo + tid; i0 < hi; i0 += nt * U) {
            float n[U];
#pragma unroll
            for (int j = 0; j < U; ++j) n[j] = i0 + j * nt < hi ? neg[i0 + j * nt] : 0.0f;
#pragma unroll
            for (int j = 0; j < U; ++j)
                if (i0 + j * nt < hi) sn += softplus_torch(n[j]);
        }
        part[0] = block_sum(sn, sh);
        part[1] = block_sum(sp, sh);
    }
    if (regularizer > 0.0f && g == 0) {  // models.py:59-60, 261-266: the rows' squares, summed per positive by the scores kernel
        double s = 0.0;
        for (int i = tid; i < B; i += nt) s += (double)pos[B + i];
        part[2] = block_sum(s, sh);
    }
    double* partials = reinterpret_cast<double*>(pos + save_pos_partials_at(B));
    if (G > 1) {
        if (tid == 0) {
            for (int j = 0; j < 3; ++j) partials[3 * g + j] = part[j];
            __threadfence();
            ticket_sh = atomicAdd(reinterpret_cast<unsigned*>(pos + save_pos_ticket_at(B)), 1u);
        }
        __syncthreads();
        if (ticket_sh != (unsigned)(G - 1)) return;  // not the last workgroup to finish
        __threadfence();
        if (tid == 0) {
            part[0] = part[1] = part[2] = 0.0;
            for (int b = 0; b < G; ++b)  // block order, whoever finished last
                for (int j = 0; j < 3; ++j) part[j] += __builtin_nontemporal_load(&partials[3 * b + j]);
        }
    }
    if (tid == 0) {
        const double model_loss = loss == LOSS_MARGIN ? part[0] / (double)BK : (part[1] / B + part[0] / (double)BK) / 2.0;
        const double reg = regularizer > 0.0f ? (double)regularizer * part[2] / ((double)B * D) / 3.0 : 0.0;
        out[0] = (float)(model_loss + reg);
    }
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

constexpr int kHitCap = 4096;                 // hits a workgroup holds in LDS per round (more: further rounds)
constexpr int kScanSlices = 8;                // 64-entry slices of neg_idx per wave and scan step (all in flight together)
constexpr int kMaxRowsPerBlock = 16;          // R: entity rows a workgroup owns
constexpr int kMaxTasks = 16;                 // (row, share) tasks of a workgroup

// Two shapes of the backward kernel: rows of up to 128 elements (the scripts' dim) keep 2 elements per lane, which leaves
// room for 16 waves per workgroup (4 per SIMD hide the dependent loads of the walk); wider rows (up to 512 elements per
// sweep, more: the scan repeats) take 4 waves.
template <int SWEEP> struct GradShape;
template <> struct GradShape<2> { static constexpr int kWaves = 16; };
template <> struct GradShape<8> { static constexpr int kWaves = 4; };
// pairs whose row loads are in flight together: every pair in flight holds five wave-uniform values and three row
// addresses in scalar registers -- eight pairs of a model that reads both halves of its rows spill them
__host__ __device__ constexpr int grad_batch(int model, int sweep) { return sweep == 2 && (model == TRANSE || model == DISTMULT) ? 8 : 4; }

// Rows per workgroup of the entity part: about 128 workgroups on big batches (every workgroup scans all of neg_idx: the
// fewer there are, the less is read twice), never more hits expected than a quarter of what the LDS list holds (2 K per
// row on average).
static int grad_rows_per_block(int B, int K) {
    int R = (2 * B + 127) / 128;
    const int cap = kHitCap / (4 * (K > 0 ? K : 1));  // K hits per row on average
    if (R > cap) R = cap;
    if (R > kMaxRowsPerBlock) R = kMaxRowsPerBlock;
    return R < 1 ? 1 : R;
}

// Element i of a lane's share of a sweep.  SWEEP == 2 (rows of up to 128 elements, D % 4 == 0): the lane owns the two
// CONSECUTIVE elements 2 lane, 2 lane + 1, so that every operand of a contribution is one 2-element load per half of the row
// (the walk is bound by the number of load instructions: one lane-strided element per load was 12 loads per ComplEx pair,
// this is 4); SWEEP == 8: lane-strided elements lane + 64 i (any width, scalar loads).
template <int SWEEP>
__device__ __forceinline__ int elem_at(int d0, int lane, int i) { return SWEEP == 2 ? d0 + 2 * lane + i : d0 + lane + 64 * i; }

template <class T> struct Vec2 { typedef T type __attribute__((ext_vector_type(2))); };
template <class T>
__device__ __forceinline__ void load2(const T* __restrict__ p, float (&v)[2]) {  // 2 consecutive elements, widened
    const typename Vec2<T>::type x = *reinterpret_cast<const typename Vec2<T>::type*>(p);
    v[0] = (float)x[0]; v[1] = (float)x[1];
}

// d score / d operand at the lane's element pair (d, d + 1), d even, D % 4 == 0; `operand` is wave-uniform, so only the rows
// the derivative needs are loaded (the derivative with respect to one operand does not read it, except TransE's sign).
template <int MODEL, class TE, class TR>
__device__ __forceinline__ void dscore2(int operand, const TE* __restrict__ h, const TE* __restrict__ t, const TR* __restrict__ r,
                                        int d, int D, float (&out)[2]) {
    if constexpr (MODEL == TRANSE) {
        float hv[2], tv[2], rv[2];
        load2(h + d, hv); load2(t + d, tv); load2(r + d, rv);
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const float sg = sign0(hv[k] + rv[k] - tv[k]);
            out[k] = operand == 1 ? sg : -sg;
        }
    } else if constexpr (MODEL == DISTMULT) {
        float a[2], b[2];
        if (operand == 0) { load2(r + d, a); load2(t + d, b); }
        else if (operand == 1) { load2(h + d, a); load2(r + d, b); }
        else { load2(h + d, a); load2(t + d, b); }
        out[0] = a[0] * b[0];
        out[1] = a[1] * b[1];
    } else {
        const int H = D / 2;
        const bool second = d >= H;  // (H is even: a pair never straddles the halves)
        const int j = second ? d - H : d;
        float a0[2], a1[2], b0[2], b1[2];  // the two other operands' first- and second-half pairs
        if (operand == 0) { load2(r + j, a0); load2(r + H + j, a1); load2(t + j, b0); load2(t + H + j, b1); }
        else if (operand == 1) { load2(h + j, a0); load2(h + H + j, a1); load2(r + j, b0); load2(r + H + j, b1); }
        else { load2(h + j, a0); load2(h + H + j, a1); load2(t + j, b0); load2(t + H + j, b1); }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            if constexpr (MODEL == COMPLEX) {
                // s = rr*hr*tr + rr*hi*ti + ri*hr*ti - ri*hi*tr
                if (operand == 0) out[k] = second ? a0[k] * b1[k] - a1[k] * b0[k] : a0[k] * b0[k] + a1[k] * b1[k];       // a = r, b = t
                else if (operand == 1) out[k] = second ? b0[k] * a1[k] + b1[k] * a0[k] : b0[k] * a0[k] - b1[k] * a1[k];  // a = h, b = r
                else out[k] = second ? a0[k] * b1[k] - a1[k] * b0[k] : a0[k] * b0[k] + a1[k] * b1[k];                    // a = h, b = t
            } else {
                // s = (hh*ra*tt + th*rb*ht) / 2 ; h = [hh|ht], t = [th|tt], r = [ra|rb]
                if (operand == 0) out[k] = 0.5f * (second ? b0[k] * a1[k] : a0[k] * b1[k]);       // a = r, b = t: ht' = th*rb ; hh' = ra*tt
                else if (operand == 1) out[k] = 0.5f * (second ? a0[k] * b0[k] : b1[k] * a1[k]);  // a = h, b = r: tt' = hh*ra ; th' = rb*ht
                else out[k] = 0.5f * (second ? b0[k] * a1[k] : a0[k] * b1[k]);                    // a = h, b = t: rb' = th*ht ; ra' = hh*tt
            }
        }
    }
}

// g += gn[n] * d score / d operand op[n] of pair (rows ih[n], it[n], relation row pb[n]), n < cnt, in that order; the
// loads of all cnt pairs are issued before the first dependent addition (cnt is wave-uniform).
template <int MODEL, int SWEEP, int BATCH, class TE, class TR>
__device__ __forceinline__ void add_contributions(float (&g)[SWEEP], const TE* __restrict__ ent, const TR* __restrict__ rel,
                                                  int D, int d0, int lane, int cnt, const int (&ih)[BATCH], const int (&it)[BATCH],
                                                  const int (&pb)[BATCH], const float (&gn)[BATCH], const int (&op)[BATCH]) {
    if constexpr (SWEEP == 2) {
        const int d = d0 + 2 * lane;
        float v[BATCH][2];
#pragma unroll
        for (int n = 0; n < BATCH; ++n) {
            v[n][0] = v[n][1] = 0.0f;
            if (n < cnt && d < D)  // (D % 4 == 0: d + 1 < D as well)
                dscore2<MODEL>(op[n], ent + (size_t)ih[n] * D, ent + (size_t)it[n] * D, rel + (size_t)pb[n] * D, d, D, v[n]);
        }
#pragma unroll
        for (int n = 0; n < BATCH; ++n)
            if (n < cnt) {
                g[0] += gn[n] * v[n][0];
                g[1] += gn[n] * v[n][1];
            }
    } else {
#pragma unroll
        for (int i = 0; i < SWEEP; ++i) {
            const int d = d0 + lane + 64 * i;
            if (d0 + 64 * i >= D) break;  // (wave-uniform)
            float v[BATCH];
#pragma unroll
            for (int n = 0; n < BATCH; ++n)
                v[n] = (n < cnt && d < D) ? dscore<MODEL>(op[n], ent + (size_t)ih[n] * D, ent + (size_t)it[n] * D, rel + (size_t)pb[n] * D, d, D) : 0.0f;
#pragma unroll
            for (int n = 0; n < BATCH; ++n)
                if (n < cnt) g[i] += gn[n] * v[n];
        }
    }
}

// A wave consumes the set bits of `mask` (lanes holding a contribution: its pair's rows in hv / tv, relation row in bv,
// loss gradient in gv, operand in ov), BATCH at a time, in lane order.
template <int MODEL, int SWEEP, int BATCH, class TE, class TR>
__device__ __forceinline__ void consume(float (&g)[SWEEP], unsigned long long mask, int hv, int tv, int bv, float gv, int ov,
                                        const TE* __restrict__ ent, const TR* __restrict__ rel, int D, int d0, int lane) {
    while (mask) {
        int ih[BATCH], it[BATCH], pb[BATCH], op[BATCH];
        float gn[BATCH];
        int cnt = 0;
#pragma unroll
        for (int n = 0; n < BATCH; ++n) {
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
        add_contributions<MODEL, SWEEP, BATCH>(g, ent, rel, D, d0, lane, cnt, ih, it, pb, gn, op);
    }
}

// One grid: workgroups [0, ent_blocks) own R rows of ent_embs.view(2B, D) each; workgroups after them own kWaves /
// rel_shares relation rows, each row's K negatives split over rel_shares waves.
//
// Entity rows.  The negative pairs that reference row j are the entries e of neg_idx.view(-1) with neg_idx[e] == j
// (pair e >> 1, slot e & 1).  The workgroup compacts the entries whose value lies in its row range into `hits` IN ENTRY
// ORDER: wave w owns the w-th contiguous part of neg_idx, counts its hits, and -- after ONE exchange of the waves' totals
// through LDS -- lists them behind those of the waves before it (ballots and popcounts inside a wave; no barrier inside
// either loop, eight coalesced loads in flight).  Then every (row, share) task -- a wave takes tasks w, w + kWaves, ...; a row is cut into `shares` tasks by
// pair index when the workgroup owns fewer rows than it has waves -- walks the list: 64 entries at a time the lanes fetch
// their pair's rows and loss gradient, the wave then adds gn(pair) * d score / d row for the entries of its task in list
// order, kBatch pairs' row loads in flight together.  A task's partial sum is parked in LDS; at the end a row's shares are
// added in share order.  More than kHitCap hits in the range (every negative pointing at a few rows): the scan repeats
// for the next kHitCap, the parked sums carry over.  Fixed orders everywhere: the gradients are bit-reproducible.
template <int MODEL, class TE, class TR, int SWEEP>
__global__ __launch_bounds__(GradShape<SWEEP>::kWaves * 64) void inbatch_grad_kernel(
    int loss, const TE* __restrict__ ent, const TR* __restrict__ rel, const int64_t* __restrict__ neg_idx, int B, int K,
    int D, float regularizer, const float* __restrict__ grad_loss, const float* __restrict__ pos,
    const float* __restrict__ neg, TE* __restrict__ grad_ent, TR* __restrict__ grad_rel, int ent_blocks, int R,
    int rel_shares) {
    constexpr int WAVES = GradShape<SWEEP>::kWaves, BATCH = grad_batch(MODEL, SWEEP);
    constexpr int kTasks = WAVES > kMaxTasks ? WAVES : kMaxTasks;
    __shared__ int hits[kHitCap];
    __shared__ unsigned short hit_row[kHitCap];
    __shared__ int wave_count[WAVES];
    __shared__ int mine[WAVES][64];
    __shared__ float park[kTasks][64 * SWEEP];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const float gl = grad_loss[0];
    const float reg_scale = regularizer > 0.0f ? regularizer * 2.0f / (3.0f * B * D) : 0.0f;

    if ((int)blockIdx.x >= ent_blocks) {  // ---- relation rows: positive pair + its K negatives (models.py:67 broadcasts rels over K)
        const int rows_here = WAVES / rel_shares;
        const int b = ((int)blockIdx.x - ent_blocks) * rows_here + wave / rel_shares, my_share = wave % rel_shares;
        const bool active = b < B;
        const int bb = active ? b : B - 1;
        const TR* r = rel + (size_t)bb * D;
        const float pb = pos[bb];
        const float gp = dloss_dpos_wave(loss, pb, neg + (size_t)bb * K, B, K, lane);
        for (int d0 = 0; d0 < D; d0 += 64 * SWEEP) {
            float g[SWEEP];
#pragma unroll
            for (int i = 0; i < SWEEP; ++i) {
                const int d = elem_at<SWEEP>(d0, lane, i);
                g[i] = (d < D && my_share == 0)
                           ? gp * dscore<MODEL>(2, ent + (size_t)(2 * bb) * D, ent + (size_t)(2 * bb + 1) * D, r, d, D) : 0.0f;
            }
            // the K negatives, share s the k-range [s K / shares, (s + 1) K / shares): lanes fetch 64 pairs' indices and
            // loss gradients at once, the wave then consumes them in k order
            const int k_lo = (int)((int64_t)K * my_share / rel_shares), k_hi = (int)((int64_t)K * (my_share + 1) / rel_shares);
            for (int kb = k_lo; kb < k_hi && active; kb += 64) {
                const int k = kb + lane;
                const bool in = k < k_hi;
                const size_t pair = (size_t)bb * K + (in ? k : k_lo);
                const float gn = in ? dloss_dneg(loss, pb, neg[pair], B, K) : 0.0f;
                const int hv = (int)neg_idx[2 * pair], tv = (int)neg_idx[2 * pair + 1];
                consume<MODEL, SWEEP, BATCH>(g, __ballot(in && gn != 0.0f), hv, tv, bb, gn, 2, ent, rel, D, d0, lane);
            }
            if (rel_shares > 1) {
                __syncthreads();
#pragma unroll
                for (int i = 0; i < SWEEP; ++i) park[wave][lane + 64 * i] = g[i];
                __syncthreads();
                if (my_share == 0) {
#pragma unroll
                    for (int i = 0; i < SWEEP; ++i)
                        for (int s = 1; s < rel_shares; ++s) g[i] += park[wave + s][lane + 64 * i];
                }
            }
            if (active && my_share == 0) {
#pragma unroll
                for (int i = 0; i < SWEEP; ++i) {
                    const int d = elem_at<SWEEP>(d0, lane, i);
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
    int shares = WAVES / n_rows;  // tasks per row: every wave gets one when the workgroup owns fewer rows than it has waves
    shares = shares < 1 ? 1 : (n_rows * shares > kTasks ? kTasks / n_rows : shares);
    const int n_tasks = n_rows * shares;
    for (int d0 = 0; d0 < D; d0 += 64 * SWEEP) {  // one sweep up to D = 64 SWEEP; wider rows repeat the scan
        for (int64_t skip = 0;; skip += kHitCap) {  // rounds of at most kHitCap hits (normally one)
            // -- scan: the entries that name a row of the range, in entry order.  Wave w owns the w-th contiguous part of
            // neg_idx: it counts its hits (pass 1: 8 coalesced loads in flight, no barrier in the loop), the waves
            // exchange their totals once, and it lists its hits behind those of the waves before it (pass 2).
            const int64_t per_wave = ((total + WAVES - 1) / WAVES + 63) / 64 * 64;
            const int64_t w0 = (int64_t)wave * per_wave, w1 = w0 + per_wave < total ? w0 + per_wave : total;
            int n_mine = 0;  // (wave-uniform)
            for (int64_t base = w0; base < w1; base += 64 * kScanSlices) {
                int x[kScanSlices];  // (row indices: below 2 B < 2^31)
#pragma unroll
                for (int i = 0; i < kScanSlices; ++i) {
                    const int64_t e = base + 64 * i + lane;
                    x[i] = e < w1 ? (int)neg_idx[e] : -1;
                }
#pragma unroll
                for (int i = 0; i < kScanSlices; ++i) n_mine += __popcll(__ballot(x[i] >= row0 && x[i] < row1));
            }
            __syncthreads();  // (the previous round's readers of wave_count and hits are done)
            if (lane == 0) wave_count[wave] = n_mine;
            __syncthreads();
            int64_t seen = 0, ord = 0;  // hits of the range in all of neg_idx / before this wave's part
#pragma unroll
            for (int w = 0; w < WAVES; ++w) {
                const int c = wave_count[w];
                if (w < wave) ord += c;
                seen += c;
            }
            for (int64_t base = w0; base < w1 && n_mine > 0; base += 64 * kScanSlices) {
                int x[kScanSlices];
#pragma unroll
                for (int i = 0; i < kScanSlices; ++i) {
                    const int64_t e = base + 64 * i + lane;
                    x[i] = e < w1 ? (int)neg_idx[e] : -1;
                }
#pragma unroll
                for (int i = 0; i < kScanSlices; ++i) {
                    const bool hit = x[i] >= row0 && x[i] < row1;
                    const unsigned long long m = __ballot(hit);
                    const int64_t at = ord + __popcll(m & ((1ull << lane) - 1ull)) - skip;
                    if (hit && at >= 0 && at < kHitCap) {
                        hits[at] = (int)(base + 64 * i + lane);
                        hit_row[at] = (unsigned short)(x[i] - row0);
                    }
                    ord += __popcll(m);
                }
            }
            __syncthreads();
            const int n_hits = (int)(seen - skip < kHitCap ? (seen > skip ? seen - skip : 0) : kHitCap);
            const bool last_round = seen <= skip + kHitCap;
            // -- walk: every task adds the contributions of its entries, in list order
            for (int task = wave; task < n_tasks; task += WAVES) {
                const int row_local = task / shares, my_share = task % shares, my_row = row0 + row_local;
                float g[SWEEP];
                if (skip == 0) {
#pragma unroll
                    for (int i = 0; i < SWEEP; ++i) g[i] = 0.0f;
                    if (my_share == 0) {  // positive pair (2b, 2b + 1, rel b): the row is its head or its tail
                        const int b = my_row >> 1, slot = my_row & 1;
                        const float gp = dloss_dpos_wave(loss, pos[b], neg + (size_t)b * K, B, K, lane);
#pragma unroll
                        for (int i = 0; i < SWEEP; ++i) {
                            const int d = elem_at<SWEEP>(d0, lane, i);
                            if (d < D)
                                g[i] = gp * dscore<MODEL>(slot, ent + (size_t)(2 * b) * D, ent + (size_t)(2 * b + 1) * D,
                                                          rel + (size_t)b * D, d, D);
                        }
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < SWEEP; ++i) g[i] = park[task][lane + 64 * i];
                }
                // the task's entries, gathered from the list into the wave's own 64-slot buffer (LDS only: no memory round
                // trip per 64 list entries, most of which belong to other tasks); a full buffer -- or the last, partial
                // one -- is then consumed: its lanes fetch their pairs' rows and loss gradients together
                int filled = 0;  // (wave-uniform)
                for (int base = 0;; base += 64) {
                    const bool tail = base >= n_hits;  // one extra trip flushes what is left
                    if (!tail) {
                        const bool in = base + lane < n_hits;
                        const int e = in ? hits[base + lane] : 0;
                        const bool ours = in && hit_row[base + lane] == row_local && (e >> 1) % shares == my_share;
                        const unsigned long long mask = __ballot(ours);
                        const int at = filled + __popcll(mask & ((1ull << lane) - 1ull));
                        if (ours && at < 64) mine[wave][at] = e;
                        const int n_new = __popcll(mask);
                        if (filled + n_new < 64) { filled += n_new; continue; }
                        // buffer full: consume 64, then park the overflow of this trip at the front
                        wave_lds_fence();
                        const int e_full = mine[wave][lane];
                        wave_lds_fence();
                        if (ours && at >= 64) mine[wave][at - 64] = e;
                        filled = filled + n_new - 64;
                        const int pair = e_full >> 1, pb = pair / K;
                        const float gn = dloss_dneg(loss, pos[pb], neg[pair], B, K);
                        const int hv = (int)neg_idx[2 * (size_t)pair], tv = (int)neg_idx[2 * (size_t)pair + 1];
                        consume<MODEL, SWEEP, BATCH>(g, __ballot(gn != 0.0f), hv, tv, pb, gn, e_full & 1, ent, rel, D, d0, lane);
                    } else if (filled > 0) {
                        wave_lds_fence();
                        const bool in = lane < filled;
                        const int e_last = in ? mine[wave][lane] : 0;
                        const int pair = e_last >> 1, pb = pair / K;
                        const float gn = in ? dloss_dneg(loss, pos[pb], neg[pair], B, K) : 0.0f;
                        const int hv = (int)neg_idx[2 * (size_t)pair], tv = (int)neg_idx[2 * (size_t)pair + 1];
                        consume<MODEL, SWEEP, BATCH>(g, __ballot(in && gn != 0.0f), hv, tv, pb, gn, e_last & 1, ent, rel, D, d0, lane);
                        wave_lds_fence();
                    }
                    if (tail) break;
                }
#pragma unroll
                for (int i = 0; i < SWEEP; ++i) park[task][lane + 64 * i] = g[i];
            }
            if (last_round) break;
        }
        __syncthreads();
        // -- a row's shares in share order, the regulariser's term, the store
        for (int row_local = wave; row_local < n_rows; row_local += WAVES) {
            const int my_row = row0 + row_local;
#pragma unroll
            for (int i = 0; i < SWEEP; ++i) {
                const int d = elem_at<SWEEP>(d0, lane, i);
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

template <int MODEL, class TE, class TR, int SWEEP>
static void launch_grad(int loss, const TE* ent, const TR* rel, const int64_t* neg_idx, int B, int K, int D, float regularizer,
                        const float* grad_loss, const float* pos, const float* neg, TE* grad_ent, TR* grad_rel, hipStream_t stream) {
    constexpr int WAVES = GradShape<SWEEP>::kWaves;
    const int R = grad_rows_per_block(B, K);
    // relation rows: a workgroup per row (its K negatives over all the waves) while that does not flood the chip
    int rel_shares = B <= 128 ? WAVES : (B <= 512 ? 4 : 1);
    rel_shares = rel_shares > WAVES ? WAVES : rel_shares;
    const int ent_blocks = (2 * B + R - 1) / R, rows_per_rel_block = WAVES / rel_shares;
    const int rel_blocks = (B + rows_per_rel_block - 1) / rows_per_rel_block;
    inbatch_grad_kernel<MODEL, TE, TR, SWEEP><<<dim3((unsigned)(ent_blocks + rel_blocks)), WAVES * 64, 0, stream>>>(
        loss, ent, rel, neg_idx, B, K, D, regularizer, grad_loss, pos, neg, grad_ent, grad_rel, ent_blocks, R, rel_shares);
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
    if ((int64_t)B * K >= (1ll << 30)) return hipErrorInvalidValue;  // entries of neg_idx are numbered in 32 bits
    return dispatch_model(model, [&](auto m) {
        return dispatch_types(ent_dtype, rel_dtype, [&](auto te, auto tr) {
            using TE = typename decltype(te)::type;
            using TR = typename decltype(tr)::type;
            const int per_block = 256 / lanes_per_pair(decltype(m)::value, D);
            const int64_t pairs = (int64_t)B * (K + 1), pair_blocks = (pairs + per_block - 1) / per_block;
            const int64_t reg_blocks = regularizer > 0.0f ? (B + 3) / 4 : 0;
            if (pair_blocks + reg_blocks > 0x7fffffff) return hipErrorInvalidValue;
            inbatch_scores_kernel<decltype(m)::value, TE, TR><<<dim3((unsigned)(pair_blocks + reg_blocks)), 256, 0, stream>>>(
                static_cast<const TE*>(ent), static_cast<const TR*>(rel), neg_idx, B, K, D, save_pos, save_neg, (unsigned)pair_blocks);
            int reduce_blocks = (int)(((int64_t)B * K + kReduceSlice - 1) / kReduceSlice);
            reduce_blocks = reduce_blocks < 1 ? 1 : (reduce_blocks > kReduceBlocks ? kReduceBlocks : reduce_blocks);
            inbatch_reduce_kernel<<<reduce_blocks, 1024, 0, stream>>>(loss, save_pos, save_neg, B, K, D, regularizer, out_loss);
            return hipGetLastError();
        });
    });
}

hipError_t launch_inbatch_loss_bwd(int model, int loss, int ent_dtype, int rel_dtype, const void* ent, const void* rel,
                                   const int64_t* neg_idx, int B, int K, int D, float regularizer,
                                   const float* grad_loss, const float* save_pos, const float* save_neg,
                                   void* grad_ent, void* grad_rel, hipStream_t stream) {
    if ((int64_t)B * K >= (1ll << 30)) return hipErrorInvalidValue;
    return dispatch_model(model, [&](auto m) {
        return dispatch_types(ent_dtype, rel_dtype, [&](auto te, auto tr) {
            using TE = typename decltype(te)::type;
            using TR = typename decltype(tr)::type;
            if (D <= 128 && D % 4 == 0)  // (element pairs per lane: rows and their halves start on even elements)
                launch_grad<decltype(m)::value, TE, TR, 2>(loss, static_cast<const TE*>(ent), static_cast<const TR*>(rel), neg_idx, B, K, D,
                                                           regularizer, grad_loss, save_pos, save_neg, static_cast<TE*>(grad_ent),
                                                           static_cast<TR*>(grad_rel), stream);
            else
                launch_grad<decltype(m)::value, TE, TR, 8>(loss, static_cast<const TE*>(ent), static_cast<const TR*>(rel), neg_idx, B, K, D,
                                                           regularizer, grad_loss, save_pos, save_neg, static_cast<TE*>(grad_ent),
                                                           static_cast<TR*>(grad_rel), stream);
            return hipGetLastError();
        });
    });
}

}  // namespace blp
