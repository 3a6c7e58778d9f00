// inbatch_loss.hip -- LinkPrediction.compute_loss (models.py:51-70) with margin / nll loss
// (models.py:251-258) and L2 regularisation (models.py:261-266), forward and backward, for
// in-batch negatives.  The reference materialises ent_embs.view(2B, D)[neg_idx] as a (B, K, 2, D)
// temporary and runs ~40 small torch kernels per step; here the gather is implicit (rows are read
// from the 2B x D matrix, which is 64 KB at B = 64 and lives in L2) and a step is TWO launches (three when the forward has
// more than 96 scoring workgroups):
//   fwd    inbatch_forward_kernel  the B (K + 1) pair scores with coalesced row reads: the bilinear models 32 lanes per
//                                  pair (lane i = running sum i of torch.sum, then its fold by wavefront shuffles), TransE
//                                  four lanes per pair (its L1 sum is one sequential chain walking through them) --
//                                  scores bit-identical to the reference at the scripts' widths.  Few scoring workgroups
//                                  (<= 96: TransE at the scripts' batch sizes -- a workgroup per row of the batch, its K
//                                  negatives and its positive side by side): each forms its slots' loss terms and leaves
//                                  their f64 sum, the last one (a ticket) adds the sums in workgroup order.  Many: a second
//                                  launch (inbatch_reduce_kernel) forms the loss from the scores.  Extra workgroups leave
//                                  the rows' shares of the L2 regulariser, and others an INDEX of neg_idx (per chunk: the
//                                  entries grouped by the row they name, stably) for the backward;
//   bwd    inbatch_grad4_kernel    (rows of up to 128 elements; inbatch_grad_kernel: any width) entity rows and relation
//                                  rows in one grid.  S waves share a row of ent_embs.view(2B, D) and walk the negative
//                                  pairs that reference it straight from the forward's index, in entry order, a half-wave
//                                  per contribution: O(B K) work per step whatever the grid (round 5: every workgroup
//                                  scanned all of neg_idx for its rows -- 92 us at B = 1 024, now 34), no float atomics,
//                                  gradients bit-reproducible run to run.
// Storage types: ent_embs / grad_ent in TE, rel_vecs / grad_rel in TR, each f32, f16 or bf16 (TR = TE or
// f32: under autocast the encoder output is half while nn.Embedding rows stay f32).  Half operands are
// widened exactly and every operation is the f32 one of the reference; gradients are rounded once on store.
// Floating point: the loss and the gradients are sums in this file's own (fixed) order: they agree with the reference
// to ~1e-6 / ~1e-5 relative (tests/test_gpu_parity.py states the tolerances).  No kernel here uses private scratch
// memory (tests/test_abi.py reads the code-object notes).  This path is launch / latency-bound (tens of KB of data): no roofline applies; DESIGN.md 4.6.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "knobs.h"
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

// ---------------------------------------------------------------- what the forward leaves for the backward
// save_pos, in 4-byte units (include/blp_hip.h: blp_inbatch_loss_save_floats):
//   [0, B)                          the positives' scores
//   partials  2 doubles per pair / regulariser workgroup of the forward: its slots' loss terms (f64, slot order)
//   off       n_chunks x (2B + 1) int32: chunk c's exclusive offsets of row j's entries in ITS part of `sorted`
//   sorted    2BK x int2 {e, the row the OTHER slot of e's pair names}: the entries e of neg_idx.view(-1) (pair e >> 1, slot
//             e & 1; with the partner row beside it the backward knows both rows of the pair without reading neg_idx),
//             chunk by chunk of `chunk` consecutive
//             entries, inside a chunk grouped by the row they name (neg_idx[e]) and in entry order within a row
// The index is what makes the backward O(BK): the negatives that reference row j are off[c][j] .. off[c][j + 1] of every
// chunk c, and walking the chunks in order visits them in ENTRY order -- a fixed summation order, so gradients are
// bit-reproducible without float atomics (round 5 had every one of ~128 workgroups scan all of neg_idx for its rows).
struct SaveLayout {
    int64_t partials_at, regsh_at, red_at, off_at, sorted_at, total;  // in floats
    int pair_blocks, reg_blocks, chunk, n_chunks;
    bool fused;  // the scoring workgroups are few enough for the last one to add their partial sums itself (one launch)
    bool rows;   // ... and each of them is ONE row of the batch: its K negatives and its positive side by side, in
                 // max(256, (K + 1) x lanes per pair) threads -- no positive is scored twice
    int threads; // of a forward workgroup
};
// Up to this many scoring workgroups finish the loss themselves (a ticket each: same-address device-scope atomics and the
// uncached round trips behind them, ~3 us in all -- what a second launch costs); more take the second launch, which
// then costs less than the tickets would ([measured] 1 072 workgroups: 14 us of tickets against a 3 us kernel).
constexpr int kFusedForwardWgs = 96;
constexpr int kReduceBlocks = 64, kReduceSlice = 8192;  // inbatch_reduce_kernel: workgroups at most, scores per workgroup
constexpr int kIdxBins = 1024;      // rows an index workgroup sorts per pass (its LDS histograms); more rows: more passes
constexpr int kIdxChunk = 1024;     // entries per chunk: 4 waves x 4 slices of 64 (the slices' values live in registers -- the index
                                    // workgroups share the forward kernel, hence its register budget, with the scoring ones)
constexpr int kIdxMaxSlices = kIdxChunk / 256;
__host__ __device__ inline int lanes_per_pair(int model, int D);
__host__ __device__ inline SaveLayout save_layout(int model, int B, int K, int D, bool regularised) {
    SaveLayout L;
    const int per_block = 256 / lanes_per_pair(model, D);
    L.pair_blocks = (int)(((int64_t)B * (K + 1) + per_block - 1) / per_block);
    L.reg_blocks = regularised ? (B + 3) / 4 : 0;
    const int64_t entries = 2ll * B * K;
    L.chunk = kIdxChunk;
    L.n_chunks = (int)((entries + L.chunk - 1) / L.chunk);
    const int packed_blocks = L.pair_blocks, lanes = lanes_per_pair(model, D);
    L.rows = K + 1 <= 256 && (int64_t)(K + 1) * lanes <= 1024 && B + L.reg_blocks <= kFusedForwardWgs;
    if (L.rows) L.pair_blocks = B;
    L.threads = L.rows && (K + 1) * lanes > 256 ? ((K + 1) * lanes + 63) / 64 * 64 : 256;
    L.fused = L.pair_blocks + L.reg_blocks <= kFusedForwardWgs;
    L.partials_at = (B + 1) / 2 * 2;  // (doubles: 8-byte aligned)
    // (sized for a regularised call either way, and for either slot mapping: the size must not depend on a float argument)
    L.regsh_at = L.partials_at + 4ll * ((packed_blocks > B ? packed_blocks : B) + (B + 3) / 4);  // (many workgroups) the rows' shares of the regulariser
    L.red_at = (L.regsh_at + B + 1) / 2 * 2;                           // (many workgroups) inbatch_reduce_kernel's partial sums
    L.off_at = L.red_at + 6 * kReduceBlocks;
    L.sorted_at = L.off_at + (int64_t)L.n_chunks * (2 * B + 1);
    L.sorted_at = (L.sorted_at + 1) / 2 * 2;  // (int2: 8-byte aligned)
    L.total = L.sorted_at + 2 * entries;
    return L;
}

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

__device__ __forceinline__ float softplus_torch(float x) {  // F.softplus, beta = 1, threshold = 20
    return x > 20.0f ? x : log1pf(expf(x));
}

// A value one workgroup hands to another through memory: device-scope atomic store / load (past the caches of this XCD's L2
// to where the other XCDs see it), the store waited for before the ticket that announces it.
__device__ __forceinline__ void publish(double* p, double v) {
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), __builtin_bit_cast(unsigned long long, v), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double observe(const double* p) {
    return __builtin_bit_cast(double, __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED,
                                                        __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ unsigned take_ticket(unsigned* counter) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // this thread's published stores have completed (s_waitcnt), no cache flush
    const unsigned t = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    return t;
}

__device__ __forceinline__ void write_loss(int loss, const double (&t)[3], int B, int K, int D, float regularizer, float* out) {
    const double BK = (double)B * K;  // t: negatives' terms | positives' softplus terms | squares
    const double model_loss = loss == LOSS_MARGIN ? t[0] / BK : (t[1] / B + t[0] / BK) / 2.0;
    const double reg = regularizer > 0.0f ? (double)regularizer * t[2] / ((double)B * D) / 3.0 : 0.0;
    out[0] = (float)(model_loss + reg);
}

// by ONE wave: the workgroups' partial sums (published with device-scope stores) added in a fixed order
__device__ __forceinline__ void finish_loss(int loss, const double* partials, int pair_blocks, int n_blocks, int B, int K, int D,
                                            float regularizer, float* out, int lane, int stride) {
    double acc[3] = {0.0, 0.0, 0.0};
    for (int i = lane; i < n_blocks; i += stride) {
        const double a = observe(&partials[2 * i]), p = observe(&partials[2 * i + 1]);
        if (i < pair_blocks) {
            acc[0] += a;
            acc[1] += p;
        } else {
            acc[2] += a;
        }
    }
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) acc[j] += __shfl_down(acc[j], off);
    if (lane == 0) write_loss(loss, acc, B, K, D, regularizer, out);
}

// One pair's score by its lane group (every lane of the wave takes part: the routines shuffle); result in the group's lane 0.
template <int MODEL, class TE, class TR>
__device__ __forceinline__ float pair_score(const TE* __restrict__ h, const TE* __restrict__ t, const TR* __restrict__ r, int D,
                                            int per_pair, int sub, int lane) {
    if constexpr (MODEL == TRANSE) {
        return per_pair == 4 ? transe_quad(h, t, r, D, lane) : transe_lane(h, t, r, D);
    } else {
        const int H = MODEL == DISTMULT ? D : D / 2;
        if (per_pair == 32) {  // torch.sum's order: lane i = running sum i, then its fold
            float a = 0.0f + bilinear_term<MODEL>(h, t, r, sub, H);
            for (int c = 1; c < H / 32; ++c) a = a + bilinear_term<MODEL>(h, t, r, 32 * c + sub, H);
            float v = a + __shfl_down(a, 8, 32);
            v = v + __shfl_down(a, 16, 32);
            v = v + __shfl_down(a, 24, 32);
            float s = 0.0f;
#pragma unroll
            for (int l = 0; l < 8; ++l) s = s + __shfl(v, l, 32);
            return MODEL == SIMPLE ? s / 2.0f : s;
        }
        float a = 0.0f;        // any other width: a shuffle tree over 16 lanes
        for (int j = sub; j < H; j += kTreeLanes) a = a + bilinear_term<MODEL>(h, t, r, j, H);
#pragma unroll
        for (int off = kTreeLanes / 2; off > 0; off >>= 1) a = a + __shfl_xor(a, off);
        return MODEL == SIMPLE ? a / 2.0f : a;
    }
}

// ---- the index of neg_idx (see SaveLayout): one workgroup sorts one chunk of `chunk` consecutive entries by the row they
// name, STABLY (entry order within a row).  Wave w owns the w-th quarter of the chunk and walks it 64 entries at a time, in
// order, against its own LDS histogram: an entry's rank among the wave's earlier entries of the same row is the histogram's
// value, plus -- when several lanes of one slice name the same row, which the lanes detect with two tagged LDS writes --
// its rank among those lanes (ballots).  The waves' histograms are then prefixed across waves and across rows (one block
// scan) and every entry is written to its place.  Rows beyond kIdxBins: further passes over the same chunk.
__device__ __forceinline__ void index_chunk(const int64_t* __restrict__ neg_idx, int64_t entries, int n_rows, int chunk, int c,
                                            int* __restrict__ off, int2* __restrict__ sorted) {
    __shared__ unsigned short hist[4][kIdxBins];
    __shared__ unsigned char tag[4][kIdxBins];
    __shared__ int start[kIdxBins];
    __shared__ int wave_tot[4];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int per_wave = chunk / 4, n_slices = per_wave / 64;  // (chunk == kIdxChunk: n_slices == kIdxMaxSlices)
    const int64_t e0 = (int64_t)c * chunk + (int64_t)wave * per_wave;
    const int64_t e_end = ((int64_t)c + 1) * chunk < entries ? ((int64_t)c + 1) * chunk : entries;
    const unsigned long long below = (1ull << lane) - 1ull;
    int* off_c = off + (size_t)c * (n_rows + 1);
    int2* sorted_c = sorted + (size_t)c * chunk;
    int placed = 0;  // entries of the rows of earlier passes (block-uniform)
    int raw[kIdxMaxSlices], partner[kIdxMaxSlices];  // the wave's entries: every slice's load in flight before the first LDS round trip
#pragma unroll
    for (int i = 0; i < kIdxMaxSlices; ++i) {
        const int64_t e = e0 + 64 * i + lane;
        raw[i] = (i < n_slices && e < e_end) ? (int)neg_idx[e] : -1;
    }
#pragma unroll
    for (int i = 0; i < kIdxMaxSlices; ++i) partner[i] = __shfl_xor(raw[i], 1);  // (entries 2p, 2p + 1 of pair p sit in neighbouring lanes)
    for (int w0 = 0; w0 < n_rows; w0 += kIdxBins) {
        for (int i = tid; i < 4 * kIdxBins; i += 256) (&hist[0][0])[i] = 0;
        __syncthreads();
        int val[kIdxMaxSlices], rank[kIdxMaxSlices];
#pragma unroll
        for (int i = 0; i < kIdxMaxSlices; ++i) {
            val[i] = -1;
            rank[i] = 0;
            if (i < n_slices) {  // (block-uniform)
                const int v = raw[i];
                const bool in = v >= w0 && v < w0 + kIdxBins;
                const int lv = in ? v - w0 : 0;
                // which lanes share their row with another lane of this slice
                if (in) tag[wave][lv] = (unsigned char)lane;
                wave_lds_fence();
                const bool lost = in && tag[wave][lv] != (unsigned char)lane;
                wave_lds_fence();
                if (lost) tag[wave][lv] = (unsigned char)(0x80 | lane);
                wave_lds_fence();
                const bool dup = in && (tag[wave][lv] & 0x80);
                wave_lds_fence();
                int among = 0, cnt = 1;
                unsigned long long dm = __ballot(dup);
                while (dm) {  // (rare: one trip per row named twice in 64 consecutive entries)
                    const int v0 = __builtin_amdgcn_readlane(lv, __builtin_ctzll(dm));
                    const unsigned long long m = __ballot(dup && lv == v0);
                    if (dup && lv == v0) {
                        among = __popcll(m & below);
                        cnt = __popcll(m);
                    }
                    dm &= ~m;
                }
                const int base = in ? hist[wave][lv] : 0;
                wave_lds_fence();
                if (in && among == cnt - 1) hist[wave][lv] = (unsigned short)(base + cnt);
                wave_lds_fence();
                val[i] = in ? lv : -1;
                rank[i] = base + among;
            }
        }
        __syncthreads();
        // the waves' counts of a row -> exclusive prefix across the waves (in place); the rows' totals -> exclusive scan
        constexpr int kPer = kIdxBins / 256;
        int tot[kPer], mine = 0;
#pragma unroll
        for (int i = 0; i < kPer; ++i) {
            const int bin = tid * kPer + i;
            int run = 0;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const int h = hist[w][bin];
                hist[w][bin] = (unsigned short)run;
                run += h;
            }
            tot[i] = run;
            mine += run;
        }
        int incl = mine;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int up = __shfl_up(incl, d);
            if (lane >= d) incl += up;
        }
        if (lane == 63) wave_tot[wave] = incl;
        __syncthreads();
        int before = placed, pass_total = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            if (w < wave) before += wave_tot[w];
            pass_total += wave_tot[w];
        }
        int at = before + incl - mine;
#pragma unroll
        for (int i = 0; i < kPer; ++i) {
            const int bin = tid * kPer + i;
            start[bin] = at;
            if (w0 + bin < n_rows) off_c[w0 + bin] = at;
            at += tot[i];
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < kIdxMaxSlices; ++i)
            if (i < n_slices && val[i] >= 0)
                sorted_c[start[val[i]] + hist[wave][val[i]] + rank[i]] = make_int2((int)(e0 + 64 * i + lane), partner[i]);
        placed += pass_total;
        __syncthreads();
    }
    if (tid == 0) off_c[n_rows] = placed;
}

// The forward in ONE launch.  Workgroups [0, n_chunks): the index of neg_idx for the backward (the longest: dispatched
// first).  Then pair_blocks workgroups: the scores -- every workgroup first scores the positive pairs of the rows its slots
// belong to (one or two rows at the scripts' sizes; the reference's bits, recomputed rather than waited for), so that it
// can form ITS slots' loss terms (models.py:251-258) right away and leave their f64 sum, slot order, in `partials`.  Then
// (regularizer > 0) one workgroup per four positive triples: the squares of their rows (models.py:261-266).  The workgroup
// that takes the LAST ticket adds the partials in workgroup order and writes the loss: a fixed summation order whoever comes
// last, no float atomics, no second launch.  `ticket`: one counter, zero when the kernel starts, left zero.
template <int MODEL, class TE, class TR>
__global__ __launch_bounds__(1024) void inbatch_forward_kernel(int loss, const TE* __restrict__ ent, const TR* __restrict__ rel,
                                                             const int64_t* __restrict__ neg_idx, int B, int K, int D,
                                                             float regularizer, float* __restrict__ pos, float* __restrict__ neg,
                                                             float* __restrict__ out, unsigned* __restrict__ ticket, SaveLayout L,
                                                             int probe) {
    __shared__ float pos_sh[256];
    __shared__ float term_neg[256], term_pos[256];
    __shared__ unsigned ticket_sh;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double* partials = reinterpret_cast<double*>(pos + L.partials_at);
    if ((int)blockIdx.x < L.n_chunks) {
        if (tid >= 256) return;  // (a row-mode launch has more waves per workgroup: the index workgroups use four)
        index_chunk(neg_idx, 2ll * B * K, 2 * B, L.chunk, (int)blockIdx.x, reinterpret_cast<int*>(pos + L.off_at),
                    reinterpret_cast<int2*>(pos + L.sorted_at));
        return;
    }
    const int blk = (int)blockIdx.x - L.n_chunks, n_blocks = L.pair_blocks + L.reg_blocks;
    double part0 = 0.0, part1 = 0.0;
    if (blk >= L.pair_blocks) {  // the regulariser: four positive triples, a wave each
        if (tid >= 256) return;
        const int b = (blk - L.pair_blocks) * 4 + wave;
        float sq = 0.0f;
        if (b < B) {
            const TE* h = ent + (size_t)(2 * b) * D;
            const TR* r = rel + (size_t)b * D;
            for (int d = lane; d < 2 * D; d += 64) sq = sq + widen(h[d]) * widen(h[d]);  // head and tail rows are adjacent
            for (int d = lane; d < D; d += 64) sq = sq + widen(r[d]) * widen(r[d]);
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) sq = sq + __shfl_xor(sq, off);
        }
        if (!L.fused) {  // many workgroups: inbatch_reduce_kernel adds the rows' shares
            if (lane == 0 && b < B) pos[L.regsh_at + b] = sq;
            return;
        }
        if (lane == 0) term_neg[wave] = sq;
        __syncthreads();
        if (tid == 0)
            for (int i = 0; i < 4; ++i) part0 += (double)term_neg[i];
    } else {
        const int64_t n_pairs = (int64_t)B * (K + 1);
        const int per_pair = lanes_per_pair(MODEL, D), sub = tid & (per_pair - 1), groups = L.threads / per_pair, gid = tid / per_pair;
        // slots: packed -- workgroup blk takes slots [blk groups, (blk + 1) groups) of the B (K + 1) (k == K: the positive) --
        // or, row mode, workgroup blk IS row blk: group k scores negative k, group K the positive
        const int64_t slot0 = L.rows ? (int64_t)blk * (K + 1) : (int64_t)blk * groups;
        const int64_t slot = slot0 + gid;
        const int64_t slot_last = L.rows ? slot0 + K : (slot0 + groups < n_pairs ? slot0 + groups - 1 : n_pairs - 1);
        const int b_first = (int)(slot0 / (K + 1)), b_last = (int)(slot_last / (K + 1));
        const bool live = L.rows ? gid <= K : slot < n_pairs;
        const int64_t pair = live ? slot : slot_last;  // (idle lanes redo the last pair: every lane has readable rows)
        const int b = (int)(pair / (K + 1)), k = (int)(pair % (K + 1));
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
        // group gid also scores the positive of row b_first + gid (at most `groups` rows: a row holds at least one slot; idle
        // groups redo the last row: the routines shuffle) -- independent of its slot's pair, so the two sets of row loads
        // are in flight together
        const int bp = b_first + gid <= b_last ? b_first + gid : b_last;
        const TE* hp = ent + (size_t)(2 * bp) * D;
        // (the positive first: its rows' loads do not wait for neg_idx, so they are in flight while the slot's indices arrive)
        const float sp = (probe & 4) || !L.fused || L.rows ? 0.0f : pair_score<MODEL>(hp, hp + D, rel + (size_t)bp * D, D, per_pair, sub, lane);
        const float s = pair_score<MODEL>(h, t, rel + (size_t)b * D, D, per_pair, sub, lane);
        if (!L.fused) {  // many workgroups: the scores only; inbatch_reduce_kernel forms the loss from them
            if (live && sub == 0) {
                if (k == K) pos[b] = s;
                else neg[(size_t)b * K + k] = s;
            }
            return;
        }
        if (L.rows) {
            if (sub == 0 && live && k == K) pos_sh[0] = s;  // the row's positive, scored once, by its own group
        } else if (sub == 0) {
            pos_sh[gid] = sp;
        }
        __syncthreads();
        if (sub == 0) {
            float tn = 0.0f, tp = 0.0f;
            if (live) {
                if (k == K) {
                    pos[b] = s;
                    if (loss == LOSS_NLL) tp = softplus_torch(-s);
                } else {
                    neg[(size_t)b * K + k] = s;
                    if (loss == LOSS_MARGIN) {  // models.py:251-254
                        float l = 1.0f - pos_sh[b - b_first];
                        l = l + s;
                        tn = l < 0.0f ? 0.0f : l;
                    } else {                    // models.py:257-258
                        tn = softplus_torch(s);
                    }
                }
            }
            term_neg[gid] = tn;
            term_pos[gid] = tp;
        }
        __syncthreads();
        if (wave == 0) {  // the block's terms, slot order within a lane, then a fixed shuffle tree
            double a = 0.0, p = 0.0;
            for (int i = lane; i < groups; i += 64) {
                a += (double)term_neg[i];
                p += (double)term_pos[i];
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                a += __shfl_down(a, off);
                p += __shfl_down(p, off);
            }
            part0 = a;
            part1 = p;
        }
    }
    // Few workgroups: the one that takes the last ticket adds everybody's partials in block order and writes the loss.  What
    // crosses workgroups (partials, the counter) is written and read with device-scope atomics that go to memory; nothing
    // else needs to be visible, so no cache-wide release / acquire (an L2 write-back + invalidate per workgroup) is paid:
    // take_ticket() waits for this thread's published stores first.
    if (probe & 2) return;
    if (tid == 0) {
        publish(&partials[2 * blk], part0);
        publish(&partials[2 * blk + 1], part1);
        ticket_sh = take_ticket(ticket);
    }
    __syncthreads();
    if (ticket_sh != (unsigned)(n_blocks - 1) || wave != 0) return;  // not the last workgroup to finish
    finish_loss(loss, partials, L.pair_blocks, n_blocks, B, K, D, regularizer, out, lane, 64);
    if (lane == 0) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // as found
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

// The second launch of a forward with many scoring workgroups (more than kFusedForwardWgs): the loss from the saved scores.
// Up to kReduceBlocks workgroups each add a slice of the B K negatives (block 0 also the
// B positives' and the regulariser's terms) in f64 and leave three partial sums behind the scores in save_pos; the
// workgroup that takes the last ticket adds the partials in block order -- one launch, a fixed summation order, no float
// atomics.  (One block for 65 536 negatives -- B = 1 024 -- was a 17 us latency-bound loop.)

__global__ __launch_bounds__(1024) void inbatch_reduce_kernel(int loss, const float* __restrict__ pos, const float* __restrict__ regsh,
                                                             double* __restrict__ partials, unsigned* __restrict__ ticket,
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
        for (int i = tid; i < B; i += nt) s += (double)regsh[i];
        part[2] = block_sum(s, sh);
    }
    if (G > 1) {
        // (what crosses workgroups goes through device-scope atomic stores / loads, as in the one-launch forward: no cache-wide
        //  release / acquire per workgroup)
        if (tid == 0) {
            for (int j = 0; j < 3; ++j) publish(&partials[3 * g + j], part[j]);
            ticket_sh = take_ticket(ticket);
        }
        __syncthreads();
        if (ticket_sh != (unsigned)(G - 1)) return;  // not the last workgroup to finish
        if (tid < 64) {  // block order, whoever finished last: lane b fetches block b's sums, lane 0 adds them in order
            double v[3] = {0.0, 0.0, 0.0};
            if (tid < G)
                for (int j = 0; j < 3; ++j) v[j] = observe(&partials[3 * tid + j]);
            part[0] = part[1] = part[2] = 0.0;
            for (int b = 0; b < G; ++b)
                for (int j = 0; j < 3; ++j) part[j] += __shfl(v[j], b);
        }
    }
    if (tid == 0) {
        const double model_loss = loss == LOSS_MARGIN ? part[0] / (double)BK : (part[1] / B + part[0] / (double)BK) / 2.0;
        const double reg = regularizer > 0.0f ? (double)regularizer * part[2] / ((double)B * D) / 3.0 : 0.0;
        out[0] = (float)(model_loss + reg);
        if (G > 1) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // as found
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

// Two shapes of the backward: rows of up to 128 elements (the scripts' dim) -- inbatch_grad4_kernel below: a half-wave per
// contribution, four elements per lane, 16 waves per workgroup -- and wider rows / odd widths -- inbatch_grad_kernel: a wave per
// contribution, lane-strided elements lane + 64 i (up to 512 elements per sweep, more: the walk repeats), 4 waves.
template <int SWEEP> struct GradShape;
template <> struct GradShape<8> { static constexpr int kWaves = 4; };
// pairs whose row loads are in flight together (every pair in flight holds five wave-uniform values and three row addresses
// in scalar registers)
__host__ __device__ constexpr int grad_batch(int, int) { return 4; }

template <int SWEEP>
__device__ __forceinline__ int elem_at(int d0, int lane, int i) { return d0 + lane + 64 * i; }

// g += gn[n] * d score / d operand op[n] of pair (rows ih[n], it[n], relation row pb[n]), n < cnt, in that order; the
// loads of all cnt pairs are issued before the first dependent addition (cnt is wave-uniform).
template <int MODEL, int SWEEP, int BATCH, class TE, class TR>
__device__ __forceinline__ void add_contributions(float (&g)[SWEEP], const TE* __restrict__ ent, const TR* __restrict__ rel,
                                                  int D, int d0, int lane, int cnt, const int (&ih)[BATCH], const int (&it)[BATCH],
                                                  const int (&pb)[BATCH], const float (&gn)[BATCH], const int (&op)[BATCH]) {
    {
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

// One grid: workgroups [0, ent_blocks) own R = kWaves / S rows of ent_embs.view(2B, D) each, S waves ("shares") per row;
// workgroups after them own kWaves / rel_shares relation rows, each row's K negatives split over rel_shares waves.
//
// Entity rows.  The negative pairs that reference row j are the entries e of neg_idx.view(-1) with neg_idx[e] == j (pair
// e >> 1, slot e & 1); the forward left them indexed (SaveLayout): chunk c of neg_idx holds off[c][j + 1] - off[c][j] of
// them, in entry order, at sorted[c * chunk + off[c][j]].  A wave reads its row's two offsets of up to 64 chunks with one
// load per lane, prefixes the counts (the row's list = the chunks' runs, concatenated: entry order), takes share s of S of
// it by position, and walks it 64 entries at a time: a lane finds the chunk its entry lies in by a binary search over the
// prefix (LDS), fetches the entry, its pair's rows and loss gradient; the wave then adds gn(pair) * d score / d row in list
// order, kBatch pairs' row loads in flight together.  A row's shares are added in share order.  Work per row: its own
// entries only -- O(BK) for the batch, whatever the number of workgroups.  Fixed orders everywhere: the gradients are
// bit-reproducible.
template <int MODEL, class TE, class TR, int SWEEP>
__global__ __launch_bounds__(GradShape<SWEEP>::kWaves * 64) void inbatch_grad_kernel(
    int loss, const TE* __restrict__ ent, const TR* __restrict__ rel, const int64_t* __restrict__ neg_idx, int B, int K,
    int D, float regularizer, const float* __restrict__ grad_loss, const float* __restrict__ pos,
    const float* __restrict__ neg, TE* __restrict__ grad_ent, TR* __restrict__ grad_rel, int ent_blocks, int S,
    int rel_shares, const int* __restrict__ off, const int2* __restrict__ sorted, int chunk, int C) {
    constexpr int WAVES = GradShape<SWEEP>::kWaves, BATCH = grad_batch(MODEL, SWEEP);
    __shared__ int run_at[WAVES][64];    // exclusive prefix of the chunks' counts (INT_MAX past the last chunk)
    __shared__ int run_from[WAVES][64];  // where the row's run starts in the chunk's part of `sorted`
    __shared__ float park[WAVES][64 * SWEEP];
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

    // ---- entity rows: wave = (row_local, share)
    const int R = WAVES / S, row_local = wave / S, my_share = wave % S;
    const int my_row = (int)blockIdx.x * R + row_local;
    const bool active = my_row < 2 * B;
    const int j = active ? my_row : 2 * B - 1;
    const int stride = 2 * B + 1;
    // how many entries name row j, and this share's part of the list (by position)
    int n_list = 0;
    for (int cb = 0; cb < C; cb += 64) {
        const int c = cb + lane;
        int cnt = c < C ? off[(size_t)c * stride + j + 1] - off[(size_t)c * stride + j] : 0;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
        n_list += cnt;
    }
    const int my_lo = active ? (int)((int64_t)n_list * my_share / S) : 0, my_hi = active ? (int)((int64_t)n_list * (my_share + 1) / S) : 0;
    for (int d0 = 0; d0 < D; d0 += 64 * SWEEP) {  // one sweep up to D = 64 SWEEP; wider rows walk the list again
        float g[SWEEP];
#pragma unroll
        for (int i = 0; i < SWEEP; ++i) g[i] = 0.0f;
        if (my_share == 0 && active) {  // positive pair (2b, 2b + 1, rel b): the row is its head or its tail
            const int b = j >> 1, slot = j & 1;
            const float gp = dloss_dpos_wave(loss, pos[b], neg + (size_t)b * K, B, K, lane);
#pragma unroll
            for (int i = 0; i < SWEEP; ++i) {
                const int d = elem_at<SWEEP>(d0, lane, i);
                if (d < D)
                    g[i] = gp * dscore<MODEL>(slot, ent + (size_t)(2 * b) * D, ent + (size_t)(2 * b + 1) * D, rel + (size_t)b * D, d, D);
            }
        }
        int at0 = 0;  // list position of the first entry of this block of 64 chunks
        for (int cb = 0; cb < C && at0 < my_hi; cb += 64) {
            const int c = cb + lane;
            const int from = c < C ? off[(size_t)c * stride + j] : 0;
            const int cnt = c < C ? off[(size_t)c * stride + j + 1] - from : 0;
            int incl = cnt;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int up = __shfl_up(incl, o);
                if (lane >= o) incl += up;
            }
            const int blk_n = __shfl(incl, 63);
            wave_lds_fence();
            run_at[wave][lane] = c < C ? incl - cnt : 0x7fffffff;
            run_from[wave][lane] = from;
            wave_lds_fence();
            const int a = my_lo > at0 ? my_lo : at0, z = my_hi < at0 + blk_n ? my_hi : at0 + blk_n;
            for (int w = a; w < z; w += 64) {
                const bool in = w + lane < z;
                const int q = (in ? w + lane : a) - at0;  // position inside this block of chunks
                int l = 0;  // the last chunk whose run starts at or before q: the run that holds q (empty runs share their start with the next)
#pragma unroll
                for (int step = 32; step > 0; step >>= 1)
                    if (run_at[wave][l + step] <= q) l += step;
                const int2 ep = sorted[(size_t)(cb + l) * chunk + run_from[wave][l] + (q - run_at[wave][l])];
                const int e = ep.x, pair = e >> 1, pb = pair / K;
                const float gn = in ? dloss_dneg(loss, pos[pb], neg[pair], B, K) : 0.0f;
                const int hv = (e & 1) ? ep.y : j, tv = (e & 1) ? j : ep.y;  // (the row itself and its pair's other row)
                consume<MODEL, SWEEP, BATCH>(g, __ballot(in && gn != 0.0f), hv, tv, pb, gn, e & 1, ent, rel, D, d0, lane);
            }
            at0 += blk_n;
        }
        // -- a row's shares in share order, the regulariser's term, the store
        if (S > 1) {
            __syncthreads();
#pragma unroll
            for (int i = 0; i < SWEEP; ++i) park[wave][lane + 64 * i] = g[i];
            __syncthreads();
            if (my_share == 0) {
#pragma unroll
                for (int i = 0; i < SWEEP; ++i)
                    for (int s = 1; s < S; ++s) g[i] += park[wave + s][lane + 64 * i];
            }
        }
        if (active && my_share == 0) {
#pragma unroll
            for (int i = 0; i < SWEEP; ++i) {
                const int d = elem_at<SWEEP>(d0, lane, i);
                if (d < D) grad_ent[(size_t)my_row * D + d] = (TE)(gl * (g[i] + reg_scale * widen(ent[(size_t)my_row * D + d])));
            }
        }
    }
}

// ---------------------------------------------------------------- backward, rows of up to 128 elements (the scripts' dim)
// The same grid and the same lists as inbatch_grad_kernel, another lane layout: a HALF-wave per contribution.  Lane (half, l)
// owns elements 4 l .. 4 l + 3 of the row (D <= 128, D % 4 == 0; the split models D % 8 == 0 so that a lane's four elements lie
// in one half of the vector), so one wave-instruction serves TWO list entries and every address is per lane: no scalar
// broadcast of a pair's five values (v_readlane x 5), no scalar address arithmetic per pair -- [counted] ~35 -> ~15
// wave-instructions per contribution.  A window of 64 list entries is fetched one entry per lane (entry, pair, rows, loss
// gradient), the entries with a non-zero loss gradient are compacted into the wave's LDS slots in list order, and the halves
// then take them alternately (half 0 the even slots, half 1 the odd ones), kUnroll slots' row loads in flight together.  A
// half adds its contributions in list order; the two halves' sums are added at the end (half 0 + half 1), then the shares in
// share order: fixed orders, bit-reproducible gradients.
__host__ __device__ constexpr int grad4_rows(int model) { return model == TRANSE ? 3 : (model == DISTMULT ? 2 : 4); }  // row pieces a contribution reads

template <int MODEL, class TE, class TR>
__device__ __forceinline__ void load_ops4(int op, const TE* __restrict__ h, const TE* __restrict__ t, const TR* __restrict__ r, int d,
                                          int D, float (&x)[grad4_rows(MODEL)][4]) {
    if constexpr (MODEL == TRANSE) {
        load4(h + d, x[0]); load4(t + d, x[1]); load4(r + d, x[2]);
    } else if constexpr (MODEL == DISTMULT) {  // the derivative with respect to one operand reads the two others
        if (op == 0) { load4(r + d, x[0]); load4(t + d, x[1]); }
        else if (op == 1) { load4(h + d, x[0]); load4(r + d, x[1]); }
        else { load4(h + d, x[0]); load4(t + d, x[1]); }
    } else {
        const int H = D / 2, j = d >= H ? d - H : d;  // (a lane's four elements lie in one half: H % 4 == 0)
        if (op == 0) { load4(r + j, x[0]); load4(r + H + j, x[1]); load4(t + j, x[2]); load4(t + H + j, x[3]); }
        else if (op == 1) { load4(h + j, x[0]); load4(h + H + j, x[1]); load4(r + j, x[2]); load4(r + H + j, x[3]); }
        else { load4(h + j, x[0]); load4(h + H + j, x[1]); load4(t + j, x[2]); load4(t + H + j, x[3]); }
    }
}

// d score / d operand `op` at the lane's four elements from what load_ops4 fetched (the formulas of dscore / dscore2)
template <int MODEL>
__device__ __forceinline__ void eval_ops4(int op, int d, int D, const float (&x)[grad4_rows(MODEL)][4], float (&out)[4]) {
    const bool second = d >= D / 2;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if constexpr (MODEL == TRANSE) {
            const float sg = sign0(x[0][k] + x[2][k] - x[1][k]);
            out[k] = op == 1 ? sg : -sg;
        } else if constexpr (MODEL == DISTMULT) {
            out[k] = x[0][k] * x[1][k];
        } else if constexpr (MODEL == COMPLEX) {  // s = rr*hr*tr + rr*hi*ti + ri*hr*ti - ri*hi*tr ; x = {a0, a1, b0, b1}
            const float a0 = x[0][k], a1 = x[1][k], b0 = x[2][k], b1 = x[3][k];
            if (op == 1) out[k] = second ? b0 * a1 + b1 * a0 : b0 * a0 - b1 * a1;   // a = h, b = r
            else out[k] = second ? a0 * b1 - a1 * b0 : a0 * b0 + a1 * b1;            // (a, b) = (r, t) or (h, t)
        } else {                                   // s = (hh*ra*tt + th*rb*ht) / 2
            const float a0 = x[0][k], a1 = x[1][k], b0 = x[2][k], b1 = x[3][k];
            if (op == 1) out[k] = 0.5f * (second ? a0 * b0 : b1 * a1);               // a = h, b = r
            else out[k] = 0.5f * (second ? b0 * a1 : a0 * b1);                        // (a, b) = (r, t) or (h, t)
        }
    }
}

constexpr int kGrad4Waves = 16;
__host__ __device__ constexpr int grad4_unroll(int model) { return model == TRANSE || model == DISTMULT ? 4 : 2; }

// Small batches are chains of dependent memory round trips, not arithmetic: there the pairs' rows are requested BEFORE their
// loss gradient is known (which costs the row loads of the pairs whose hinge is inactive, and saves a round trip).
constexpr int64_t kSpeculateEntries = 32768;

// The wave's slots [0, n_live) (LDS: the pair's rows; relation row | operand << 30; loss gradient), the halves alternately:
// issue = the row loads of slots i0 .. i0 + 2 U - 1, finish = their contributions added in slot order.
template <int MODEL, class TE, class TR, int U>
__device__ __forceinline__ void issue_slots4(float (&x)[U][grad4_rows(MODEL)][4], const int2* __restrict__ slot_rows,
                                             const int* __restrict__ slot_pbop, int i0, int n_live, const TE* __restrict__ ent,
                                             const TR* __restrict__ rel, int D, int d, bool active, int half) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int idx = i0 + 2 * u + half;
        const bool ok = active && idx < n_live;
        const int2 rows = slot_rows[ok ? idx : 0];
        const int pbop = slot_pbop[ok ? idx : 0];
        if (ok)
            load_ops4<MODEL>((int)((unsigned)pbop >> 30), ent + (size_t)rows.x * D, ent + (size_t)rows.y * D, rel + (size_t)(pbop & 0x3fffffff) * D, d, D, x[u]);
        else
#pragma unroll
            for (int a = 0; a < grad4_rows(MODEL); ++a)
#pragma unroll
                for (int k = 0; k < 4; ++k) x[u][a][k] = 0.0f;
    }
}
template <int MODEL, int U>
__device__ __forceinline__ void finish_slots4(float (&g)[4], const float (&x)[U][grad4_rows(MODEL)][4], const int* __restrict__ slot_pbop, const float* __restrict__ slot_gn,
                                              int i0, int n_live, int D, int d, bool active, int half) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int idx = i0 + 2 * u + half;
        const bool ok = active && idx < n_live;
        const float gn = ok ? slot_gn[idx] : 0.0f;
        const int op = (int)((unsigned)slot_pbop[ok ? idx : 0] >> 30);  // (read again rather than kept: a per-lane array of them went to scratch)
        float v[4];
        eval_ops4<MODEL>(op, d, D, x[u], v);
#pragma unroll
        for (int k = 0; k < 4; ++k) g[k] += gn * v[k];
    }
}

// One window of <= 64 list entries, a lane each: (rows, relation row, operand) and the scores its loss gradient comes from.
// Default: the entries with a non-zero loss gradient are compacted into the wave's LDS slots in list order, then consumed.
// SPECULATE: every entry takes a slot and the first slots' row loads are issued BEFORE the loss gradients are computed.
template <int MODEL, class TE, class TR, bool SPECULATE>
__device__ __forceinline__ void window4(float (&g)[4], bool in, int hv, int tv, int pb, int operand, float pos_b, float neg_bk,
                                        int2* __restrict__ slot_rows, int* __restrict__ slot_pbop, float* __restrict__ slot_gn,
                                        const TE* __restrict__ ent, const TR* __restrict__ rel, int loss, int B, int K, int D, int d,
                                        bool lane_active, int half, int lane) {
    constexpr int U = SPECULATE ? 2 : grad4_unroll(MODEL);
    wave_lds_fence();  // the previous window's readers are done
    float x[U][grad4_rows(MODEL)][4];
    int n_live;
    if constexpr (SPECULATE) {
        n_live = __popcll(__ballot(in));  // (the `in` lanes are the first n_live lanes)
        slot_rows[lane] = make_int2(hv, tv);
        slot_pbop[lane] = pb | (operand << 30);
        wave_lds_fence();
        issue_slots4<MODEL>(x, slot_rows, slot_pbop, 0, n_live, ent, rel, D, d, lane_active, half);
        slot_gn[lane] = in ? dloss_dneg(loss, pos_b, neg_bk, B, K) : 0.0f;
        wave_lds_fence();
    } else {
        const float gn = in ? dloss_dneg(loss, pos_b, neg_bk, B, K) : 0.0f;
        const bool keep = in && gn != 0.0f;
        const unsigned long long mask = __ballot(keep);
        n_live = __popcll(mask);
        if (keep) {
            const int at = __popcll(mask & ((1ull << lane) - 1ull));
            slot_rows[at] = make_int2(hv, tv);
            slot_pbop[at] = pb | (operand << 30);
            slot_gn[at] = gn;
        }
        wave_lds_fence();
        issue_slots4<MODEL>(x, slot_rows, slot_pbop, 0, n_live, ent, rel, D, d, lane_active, half);
    }
    finish_slots4<MODEL>(g, x, slot_pbop, slot_gn, 0, n_live, D, d, lane_active, half);
    for (int i0 = 2 * U; i0 < n_live; i0 += 2 * U) {
        issue_slots4<MODEL>(x, slot_rows, slot_pbop, i0, n_live, ent, rel, D, d, lane_active, half);
        finish_slots4<MODEL>(g, x, slot_pbop, slot_gn, i0, n_live, D, d, lane_active, half);
    }
}

template <int MODEL, class TE, class TR, bool SPECULATE>
__global__ __launch_bounds__(kGrad4Waves * 64) void inbatch_grad4_kernel(
    int loss, const TE* __restrict__ ent, const TR* __restrict__ rel, const int64_t* __restrict__ neg_idx, int B, int K,
    int D, float regularizer, const float* __restrict__ grad_loss, const float* __restrict__ pos,
    const float* __restrict__ neg, TE* __restrict__ grad_ent, TR* __restrict__ grad_rel, int ent_blocks, int S,
    int rel_shares, const int* __restrict__ off, const int2* __restrict__ sorted, int chunk, int C) {
    constexpr int WAVES = kGrad4Waves;
    __shared__ int run_at[WAVES][64];
    __shared__ int run_from[WAVES][64];
    __shared__ int2 slot_rows[WAVES][64];
    __shared__ int slot_pbop[WAVES][64];
    __shared__ float slot_gn[WAVES][64];
    __shared__ float park[WAVES][128];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, half = lane >> 5, d = 4 * (lane & 31);
    const bool lane_active = d < D;
    const float gl = grad_loss[0];
    const float reg_scale = regularizer > 0.0f ? regularizer * 2.0f / (3.0f * B * D) : 0.0f;
    float g[4] = {0.0f, 0.0f, 0.0f, 0.0f};

    auto window = [&](bool in, int hv, int tv, int pb, int operand, float pos_b, float neg_bk) __attribute__((always_inline)) {
        window4<MODEL, TE, TR, SPECULATE>(g, in, hv, tv, pb, operand, pos_b, neg_bk, slot_rows[wave], slot_pbop[wave], slot_gn[wave], ent, rel,
                                          loss, B, K, D, d, lane_active, half, lane);
    };
    auto fold_halves = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < 4; ++k) g[k] = g[k] + __shfl_xor(g[k], 32);  // (half 0 + half 1: the same bits in both halves)
    };

    if ((int)blockIdx.x >= ent_blocks) {  // ---- relation rows: positive pair + its K negatives (models.py:67 broadcasts rels over K)
        const int rows_here = WAVES / rel_shares;
        const int b = ((int)blockIdx.x - ent_blocks) * rows_here + wave / rel_shares, my_share = wave % rel_shares;
        const bool active = b < B;
        const int bb = active ? b : B - 1;
        const TR* r = rel + (size_t)bb * D;
        const float pb = pos[bb];
        const float gp = dloss_dpos_wave(loss, pb, neg + (size_t)bb * K, B, K, lane);
        if (my_share == 0 && half == 0 && lane_active) {
            float x[grad4_rows(MODEL)][4], v[4];
            load_ops4<MODEL>(2, ent + (size_t)(2 * bb) * D, ent + (size_t)(2 * bb + 1) * D, r, d, D, x);
            eval_ops4<MODEL>(2, d, D, x, v);
#pragma unroll
            for (int k = 0; k < 4; ++k) g[k] = gp * v[k];
        }
        const int k_lo = (int)((int64_t)K * my_share / rel_shares), k_hi = (int)((int64_t)K * (my_share + 1) / rel_shares);
        for (int kb = k_lo; kb < k_hi && active; kb += 64) {
            const int k = kb + lane;
            const bool in = k < k_hi;
            const size_t pair = (size_t)bb * K + (in ? k : k_lo);
            window(in, (int)neg_idx[2 * pair], (int)neg_idx[2 * pair + 1], bb, 2, pb, neg[pair]);
        }
        fold_halves();
        if (rel_shares > 1) {
            __syncthreads();
            if (half == 0)
#pragma unroll
                for (int k = 0; k < 4; ++k) park[wave][d + k] = g[k];
            __syncthreads();
            if (my_share == 0 && half == 0)
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    for (int s = 1; s < rel_shares; ++s) g[k] += park[wave + s][d + k];
        }
        if (active && my_share == 0 && half == 0 && lane_active) {
            float rv[4];
            load4(r + d, rv);
#pragma unroll
            for (int k = 0; k < 4; ++k) grad_rel[(size_t)b * D + d + k] = (TR)(gl * (g[k] + reg_scale * rv[k]));
        }
        return;
    }

    // ---- entity rows: wave = (row_local, share)
    const int R = WAVES / S, row_local = wave / S, my_share = wave % S;
    const int my_row = (int)blockIdx.x * R + row_local;
    const bool active = my_row < 2 * B;
    const int j = active ? my_row : 2 * B - 1;
    const int stride = 2 * B + 1;
    // the row's runs: offsets of up to 64 chunks per lane-load; one block of chunks (2 B K <= 65 536 entries) keeps them in
    // registers for the walk, more blocks are counted first and read again
    const int c0 = lane;
    const int from0 = c0 < C ? off[(size_t)c0 * stride + j] : 0;
    const int cnt0 = c0 < C ? off[(size_t)c0 * stride + j + 1] - from0 : 0;
    int n_list = cnt0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) n_list += __shfl_xor(n_list, o);
    for (int cb = 64; cb < C; cb += 64) {
        const int c = cb + lane;
        int cnt = c < C ? off[(size_t)c * stride + j + 1] - off[(size_t)c * stride + j] : 0;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
        n_list += cnt;
    }
    const int my_lo = active ? (int)((int64_t)n_list * my_share / S) : 0, my_hi = active ? (int)((int64_t)n_list * (my_share + 1) / S) : 0;
    const float gp_row = dloss_dpos_wave(loss, pos[j >> 1], neg + (size_t)(j >> 1) * K, B, K, lane);  // (the whole wave: it shuffles)
    if (my_share == 0 && active && half == 0 && lane_active) {  // positive pair (2b, 2b + 1, rel b): the row is its head or its tail
        const int b = j >> 1, slot = j & 1;
        float x[grad4_rows(MODEL)][4], v[4];
        load_ops4<MODEL>(slot, ent + (size_t)(2 * b) * D, ent + (size_t)(2 * b + 1) * D, rel + (size_t)b * D, d, D, x);
        eval_ops4<MODEL>(slot, d, D, x, v);
#pragma unroll
        for (int k = 0; k < 4; ++k) g[k] = gp_row * v[k];
    }
    int at0 = 0;  // list position of the first entry of this block of 64 chunks
    for (int cb = 0; cb < C && at0 < my_hi; cb += 64) {
        const int c = cb + lane;
        const int from = cb == 0 ? from0 : (c < C ? off[(size_t)c * stride + j] : 0);
        const int cnt = cb == 0 ? cnt0 : (c < C ? off[(size_t)c * stride + j + 1] - from : 0);
        int incl = cnt;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int up = __shfl_up(incl, o);
            if (lane >= o) incl += up;
        }
        const int blk_n = __shfl(incl, 63);
        wave_lds_fence();
        run_at[wave][lane] = c < C ? incl - cnt : 0x7fffffff;
        run_from[wave][lane] = from;
        wave_lds_fence();
        const int a = my_lo > at0 ? my_lo : at0, z = my_hi < at0 + blk_n ? my_hi : at0 + blk_n;
        for (int w = a; w < z; w += 64) {
            const bool in = w + lane < z;
            const int q = (in ? w + lane : a) - at0;
            int l = 0;
#pragma unroll
            for (int step = 32; step > 0; step >>= 1)
                if (run_at[wave][l + step] <= q) l += step;
            const int2 ep = sorted[(size_t)(cb + l) * chunk + run_from[wave][l] + (q - run_at[wave][l])];
            const int e = ep.x, pair = e >> 1, pb = pair / K;
            // (the row itself and the row the other slot of the pair names: no read of neg_idx)
            window(in, (e & 1) ? ep.y : j, (e & 1) ? j : ep.y, pb, e & 1, pos[pb], neg[pair]);
        }
        at0 += blk_n;
    }
    fold_halves();
    if (S > 1) {
        __syncthreads();
        if (half == 0)
#pragma unroll
            for (int k = 0; k < 4; ++k) park[wave][d + k] = g[k];
        __syncthreads();
        if (my_share == 0 && half == 0)
#pragma unroll
            for (int k = 0; k < 4; ++k)
                for (int s = 1; s < S; ++s) g[k] += park[wave + s][d + k];
    }
    if (active && my_share == 0 && half == 0 && lane_active) {
        float ev[4];
        load4(ent + (size_t)my_row * D + d, ev);
#pragma unroll
        for (int k = 0; k < 4; ++k) grad_ent[(size_t)my_row * D + d + k] = (TE)(gl * (g[k] + reg_scale * ev[k]));
    }
}

template <int MODEL, class TE, class TR>
static void launch_grad4(int loss, const TE* ent, const TR* rel, const int64_t* neg_idx, int B, int K, int D, float regularizer,
                         const float* grad_loss, const float* pos, const float* neg, TE* grad_ent, TR* grad_rel, hipStream_t stream) {
    constexpr int WAVES = kGrad4Waves;
    // entity rows: S waves share a row's list while that keeps the grid near 2 048 waves; a power of two, at least two
    int S = WAVES;
    while (S > 2 && (int64_t)2 * B * S > 2048) S >>= 1;
    if (const long long forced = knob(KNOB_INBATCH_SHARES); forced > 0 && forced <= WAVES && (forced & (forced - 1)) == 0) S = (int)forced;
    const int R = WAVES / S;
    int rel_shares = WAVES;
    while (rel_shares > 1 && (int64_t)B * rel_shares > 4096) rel_shares >>= 1;
    const int ent_blocks = (2 * B + R - 1) / R, rows_per_rel_block = WAVES / rel_shares;
    const int rel_blocks = (B + rows_per_rel_block - 1) / rows_per_rel_block;
    const SaveLayout L = save_layout(MODEL, B, K, D, regularizer > 0.0f);
    const int* off = reinterpret_cast<const int*>(pos + L.off_at);
    const int2* sorted = reinterpret_cast<const int2*>(pos + L.sorted_at);
    if ((int64_t)2 * B * K <= kSpeculateEntries)
        inbatch_grad4_kernel<MODEL, TE, TR, true><<<dim3((unsigned)(ent_blocks + rel_blocks)), WAVES * 64, 0, stream>>>(
            loss, ent, rel, neg_idx, B, K, D, regularizer, grad_loss, pos, neg, grad_ent, grad_rel, ent_blocks, S, rel_shares, off, sorted,
            L.chunk, L.n_chunks);
    else
        inbatch_grad4_kernel<MODEL, TE, TR, false><<<dim3((unsigned)(ent_blocks + rel_blocks)), WAVES * 64, 0, stream>>>(
            loss, ent, rel, neg_idx, B, K, D, regularizer, grad_loss, pos, neg, grad_ent, grad_rel, ent_blocks, S, rel_shares, off, sorted,
            L.chunk, L.n_chunks);
}

template <int MODEL, class TE, class TR, int SWEEP>
static void launch_grad(int loss, const TE* ent, const TR* rel, const int64_t* neg_idx, int B, int K, int D, float regularizer,
                        const float* grad_loss, const float* pos, const float* neg, TE* grad_ent, TR* grad_rel, hipStream_t stream) {
    constexpr int WAVES = GradShape<SWEEP>::kWaves;
    // entity rows: S waves share a row's list (a wave adds its entries one batch of row loads after the other: a latency chain)
    // while that keeps the grid near 2 048 waves -- [measured, tools/inbatch_probe.py] B = 64: S = 16 (7.9 us; S = 4: 11.4), B = 128:
    // S = 8 (11.2; 16: 12.5), B = 1 024: S = 2 (40; 1: 48, 4: 45, 16: 72) --; a power of two, at least two
    int S = WAVES;
    while (S > 2 && (int64_t)2 * B * S > 2048) S >>= 1;
    if (const long long forced = knob(KNOB_INBATCH_SHARES); forced > 0 && forced <= WAVES && (forced & (forced - 1)) == 0) S = (int)forced;
    const int R = WAVES / S;
    // relation rows: a workgroup per row (its K negatives over all the waves) while that does not flood the chip
    int rel_shares = WAVES;
    while (rel_shares > 1 && (int64_t)B * rel_shares > 4096) rel_shares >>= 1;
    const int ent_blocks = (2 * B + R - 1) / R, rows_per_rel_block = WAVES / rel_shares;
    const int rel_blocks = (B + rows_per_rel_block - 1) / rows_per_rel_block;
    const SaveLayout L = save_layout(MODEL, B, K, D, regularizer > 0.0f);
    inbatch_grad_kernel<MODEL, TE, TR, SWEEP><<<dim3((unsigned)(ent_blocks + rel_blocks)), WAVES * 64, 0, stream>>>(
        loss, ent, rel, neg_idx, B, K, D, regularizer, grad_loss, pos, neg, grad_ent, grad_rel, ent_blocks, S, rel_shares,
        reinterpret_cast<const int*>(pos + L.off_at), reinterpret_cast<const int2*>(pos + L.sorted_at), L.chunk, L.n_chunks);
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

size_t inbatch_loss_save_floats(int model, int B, int K, int D) { return (size_t)save_layout(model, B, K, D, true).total; }
int inbatch_loss_fwd_launches(int model, int B, int K, int D, bool regularised) { return save_layout(model, B, K, D, regularised).fused ? 1 : 2; }

hipError_t launch_inbatch_loss_fwd(int model, int loss, int ent_dtype, int rel_dtype, const void* ent, const void* rel,
                                   const int64_t* neg_idx, int B, int K, int D, float regularizer,
                                   float* out_loss, float* save_pos, float* save_neg, unsigned* ticket, hipStream_t stream) {
    if ((int64_t)B * K >= (1ll << 30)) return hipErrorInvalidValue;  // entries of neg_idx are numbered in 32 bits
    return dispatch_model(model, [&](auto m) {
        return dispatch_types(ent_dtype, rel_dtype, [&](auto te, auto tr) {
            using TE = typename decltype(te)::type;
            using TR = typename decltype(tr)::type;
            SaveLayout L = save_layout(decltype(m)::value, B, K, D, regularizer > 0.0f);
            const int probe = (int)knob(KNOB_INBATCH_PROBE);  // (0 in the product library)
            if (probe & 1) L.n_chunks = 0;
            const int64_t blocks = (int64_t)L.n_chunks + L.pair_blocks + L.reg_blocks;
            if (blocks > 0x7fffffff) return hipErrorInvalidValue;
            inbatch_forward_kernel<decltype(m)::value, TE, TR><<<dim3((unsigned)blocks), (unsigned)L.threads, 0, stream>>>(
                loss, static_cast<const TE*>(ent), static_cast<const TR*>(rel), neg_idx, B, K, D, regularizer, save_pos, save_neg,
                out_loss, ticket, L, probe);
            if (!L.fused) {
                int reduce_blocks = (int)(((int64_t)B * K + kReduceSlice - 1) / kReduceSlice);
                reduce_blocks = reduce_blocks < 1 ? 1 : (reduce_blocks > kReduceBlocks ? kReduceBlocks : reduce_blocks);
                inbatch_reduce_kernel<<<reduce_blocks, 1024, 0, stream>>>(loss, save_pos, save_pos + L.regsh_at,
                                                                          reinterpret_cast<double*>(save_pos + L.red_at), ticket + 1, save_neg,
                                                                          B, K, D, regularizer, out_loss);
            }
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
            constexpr int MODEL = decltype(m)::value;
            // rows of up to 128 elements (the scripts' dim): a half-wave per contribution, four elements per lane (the split
            // models: a lane's four elements inside one half of the vector); anything else: four waves, lane-strided elements
            if (D <= 128 && D % 4 == 0 && (MODEL == TRANSE || MODEL == DISTMULT || D % 8 == 0))
                launch_grad4<MODEL, TE, TR>(loss, static_cast<const TE*>(ent), static_cast<const TR*>(rel), neg_idx, B, K, D,
                                            regularizer, grad_loss, save_pos, save_neg, static_cast<TE*>(grad_ent),
                                            static_cast<TR*>(grad_rel), stream);
            else
                launch_grad<MODEL, TE, TR, 8>(loss, static_cast<const TE*>(ent), static_cast<const TR*>(rel), neg_idx, B, K, D,
                                              regularizer, grad_loss, save_pos, save_neg, static_cast<TE*>(grad_ent),
                                              static_cast<TR*>(grad_rel), stream);
            return hipGetLastError();
        });
    });
}

}  // namespace blp
